#!/bin/bash
# A/B environment settings on the bench workload (default overlap mode): tools/ab_env.sh "RT_SHADE_GRID=3" "RT_SHADE_GRID=4 RT_OVERLAP=3" ...
for e in "$@"; do
  env $e timeout -k 10 300 python bench.py --steps 30 --warmup 3 --no-cpu-baseline --no-in-contract --no-walk --roofline-frames 2 $AB_ARGS 2>/dev/null | python -c "
import json,sys; d=json.loads(sys.stdin.read()); r=d['roofline']
sh, sc = (r, r['second_kernel']) if r['bound'] == 'fp64_valu' else (r['second_kernel'], r)
print('$e', 'ms', round(d['ms_per_step'],3), 'frame_kernels', round(r['frame_kernels_ms'],3), 'scan_ms', round(sc['kernel_ms'],3), 'shade_ms', round(sh['kernel_ms'],3), 'cand/ray', round(r['candidates_per_ray'],3), 'pairs', round(sc.get('pairs_evaluated_frac') or 0,4), 'hash', d['config']['frame_sha256_16'])" || exit 1
done
