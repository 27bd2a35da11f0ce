#!/bin/bash
# builds of librt_hip.so on the bench workload as the driver runs it (default overlap): tools/ab_full.sh lib1.so lib2.so ...
for lib in "$@"; do
  RT_HIP_LIB=$PWD/$lib timeout -k 10 300 python bench.py --steps 60 --warmup 5 --no-cpu-baseline --no-in-contract --no-walk 2>/dev/null | python -c "
import json,sys; d=json.loads(sys.stdin.read()); r=d['roofline']
print('$lib', 'ms', round(d['ms_per_step'],3), 'shade_ms', round(r['kernel_ms'],3), 'scan_ms', round(r['second_kernel']['kernel_ms'],3), 'hash', d['config']['frame_sha256_16'])" || exit 1
done
