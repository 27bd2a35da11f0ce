#!/bin/bash
# environment settings on another workload: tools/ab_env_w.sh <workload> <steps> "" "RT_OVERLAP=3" ...
w=$1; n=$2; shift 2
for e in "$@"; do
  env $e timeout -k 10 400 python bench.py --workload $w --steps $n --warmup 1 --no-cpu-baseline --no-in-contract --no-walk --roofline-frames 1 2>/dev/null | python -c "
import json,sys; d=json.loads(sys.stdin.read()); r=d['roofline']
print('$w', '[$e]', 'ms', round(d['ms_per_step'],3), 'src', d['config'].get('candidate_source'), 'sub', d['config'].get('sub_pipelines'), 'hash', d['config']['frame_sha256_16'])" || exit 1
done
