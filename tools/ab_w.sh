#!/bin/bash
# A/B builds of librt_hip.so on another workload: tools/ab_w.sh <workload> <steps> lib1.so lib2.so ...
w=$1; n=$2; shift 2
for lib in "$@"; do
  RT_HIP_LIB=$PWD/$lib timeout -k 10 400 python bench.py --workload $w --steps $n --warmup 1 --no-cpu-baseline --no-in-contract --no-walk --roofline-frames 1 2>/dev/null | python -c "
import json,sys; d=json.loads(sys.stdin.read()); r=d['roofline']
print('$lib', '$w', 'ms', round(d['ms_per_step'],2), 'accel', d['config'].get('candidate_source'), 'cand/ray', round(r['candidates_per_ray'],2), 'hash', d['config']['frame_sha256_16'])" || exit 1
done
