#!/bin/bash
# A/B several builds of librt_hip.so on the bench workload: tools/ab.sh build_ab/lib1.so build_ab/lib2.so ...
# (kernels serialised: RT_OVERLAP=1, so frame_kernels is the sum of the kernels' own durations)
for lib in "$@"; do
  RT_OVERLAP=1 RT_HIP_LIB=$PWD/$lib timeout -k 10 300 python bench.py --steps 8 --warmup 2 --no-cpu-baseline 2>/dev/null | python -c "
import json,sys; d=json.loads(sys.stdin.read()); r=d['roofline']
print('$lib', 'ms', round(d['ms_per_step'],2), 'broad_ms', round(r['kernel_ms'],2), 'frame_kernels', round(r['frame_kernels_ms'],2), 'rest', round(r['frame_kernels_ms']-r['kernel_ms'],2), 'frac', round(r['frac'],3), 'overflow', r['overflow_rays'], 'cand/ray', round(r['candidates_per_ray'],2), 'walk_ms', round(d.get('grid_walk',{}).get('ms_per_step',0),2))" || exit 1
done
