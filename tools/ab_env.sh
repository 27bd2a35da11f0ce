#!/bin/bash
# A/B environment settings on the bench workload: tools/ab_env.sh "RT_FUSE=0" "RT_FUSE=1" ...
for e in "$@"; do
  env $e timeout -k 10 300 python bench.py --steps 8 --warmup 2 --no-cpu-baseline 2>/dev/null | python -c "
import json,sys; d=json.loads(sys.stdin.read()); r=d['roofline']
print('$e', 'ms', round(d['ms_per_step'],2), 'broad_ms', round(r['kernel_ms'],2), 'frame_kernels', round(r['frame_kernels_ms'],2), 'frac', round(r['frac'],3), 'whole', round(r['frac_whole_frame'],3))" || exit 1
done
