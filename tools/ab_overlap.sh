#!/bin/bash
# A/B builds of librt_hip.so on the bench workload with the default sub-pipeline overlap: tools/ab_overlap.sh lib1.so lib2.so ...
for lib in "$@"; do
  RT_HIP_LIB=$PWD/$lib timeout -k 10 300 python bench.py --workload ${WORKLOAD:-c3} --steps 10 --warmup 2 --no-cpu-baseline 2>/dev/null | python -c "
import json,sys; d=json.loads(sys.stdin.read()); r=d['roofline']
print('$lib', 'ms', round(d['ms_per_step'],2), 'Mrays/s', round(d['value']), 'broad_ms', round(r['kernel_ms'],2), 'serial', round(r['frame_kernels_serial_ms'],2), 'whole', round(r['frac_whole_frame'],3), 'walk_ms', round(d.get('grid_walk',{}).get('ms_per_step',0),2))" || exit 1
done
