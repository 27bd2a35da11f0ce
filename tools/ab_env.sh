#!/bin/bash
# A/B environment settings on a bench workload: [WORKLOAD=c4] tools/ab_env.sh "RT_BROAD_BLOCK=256" "RT_BROAD_BLOCK=512" ...
wl=${WORKLOAD:-c3}
for e in "$@"; do
  env $e timeout -k 10 300 python bench.py --workload $wl --steps ${STEPS:-8} --warmup 2 --no-cpu-baseline 2>/dev/null | python -c "
import json,sys; d=json.loads(sys.stdin.read()); r=d['roofline']
print('$wl $e', 'ms', round(d['ms_per_step'],2), 'Mrays/s', round(d['value']), 'broad_ms', round(r['kernel_ms'],2), 'serial', round(r['frame_kernels_serial_ms'],2), 'frac', round(r['frac'],3), 'whole', round(r['frac_whole_frame'],3))" || exit 1
done
