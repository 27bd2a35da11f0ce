#!/bin/bash
# A/B several builds of librt_hip.so on the bench workload: tools/ab.sh build_ab/lib1.so build_ab/lib2.so ...
# (kernels serialised: RT_OVERLAP=1, so frame_kernels is the sum of the kernels' own durations)
for lib in "$@"; do
  RT_OVERLAP=1 RT_HIP_LIB=$PWD/$lib timeout -k 10 300 python bench.py --steps 8 --warmup 2 --no-cpu-baseline --no-in-contract $AB_ARGS 2>/dev/null | python -c "
import json,sys; d=json.loads(sys.stdin.read()); r=d['roofline']
sh, sc = (r, r['second_kernel']) if r['bound'] == 'fp64_valu' else (r['second_kernel'], r)
print('$lib', 'ms', round(d['ms_per_step'],2), 'scan_ms', round(sc['kernel_ms'],3), 'shade_ms', round(sh['kernel_ms'],3), 'frame_kernels', round(r['frame_kernels_ms'],2), 'serial', round(r['frame_kernels_serial_ms'],2), 'overflow', r['overflow_rays'], 'cand/ray', round(r['candidates_per_ray'],2), 'other_ms', round((d.get('grid_walk') or d.get('scan') or {}).get('ms_per_step',0),2), 'hash', d['config']['frame_sha256_16'], 'same', (d.get('grid_walk') or d.get('scan') or {}).get('same_output_bits'))" || exit 1
done
