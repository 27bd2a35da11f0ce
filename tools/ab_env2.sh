#!/bin/bash
# bench line under several environment settings: tools/ab_env2.sh "RT_WG_PER_CU=1" "RT_WG_PER_CU=2 RT_OVERLAP=3" ...
for cfg in "$@"; do
  env $cfg timeout -k 10 300 python bench.py --steps 10 --warmup 3 --no-cpu-baseline --no-walk --roofline-frames 1 2>/dev/null | python -c "
import json,sys; d=json.loads(sys.stdin.read()); r=d['roofline']
print('$cfg', 'ms', round(d['ms_per_step'],3), 'broad_ms', round(r['kernel_ms'],2), 'frame_kernels', round(r['frame_kernels_ms'],3), 'serial', round(r['frame_kernels_serial_ms'],2))" || exit 1
done
