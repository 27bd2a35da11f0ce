#!/bin/bash
# Kernel trace of the bench workload, last frame listed launch by launch: gpurun -- tools/trace_frame.sh tag [bench args]
tag=$1; shift
root=${GRAFT_REPO_ROOT:-$PWD}
out=$root/gpurun_out/$tag
mkdir -p $out
cd /tmp && export TMPDIR=/tmp
timeout -s KILL 180 rocprofv3 --kernel-trace --output-format csv -d $out/trace -o vm -- python $root/bench.py --steps 3 --warmup 1 --roofline-frames 0 --no-cpu-baseline "$@" > $out/trace.log 2>&1 || { tail -5 $out/trace.log; exit 1; }
python - $out/trace/vm_kernel_trace.csv <<'PY'
import csv, sys
rows = sorted(csv.DictReader(open(sys.argv[1])), key=lambda r: int(r['Start_Timestamp']))
idx = [i for i, r in enumerate(rows) if 'init_kernel' in r['Kernel_Name']]
i0 = idx[-1] if len(idx) < 2 or idx[-1] - idx[-2] > 2 else idx[-2]
t0 = int(rows[i0]['Start_Timestamp'])
for r in rows[i0:]:
    st, en = int(r['Start_Timestamp']), int(r['End_Timestamp'])
    name = r['Kernel_Name'].replace('(anonymous namespace)::', '').replace('void ', '')[:28]
    print(f"{name:28s} start {(st - t0) / 1e3:9.1f} us  dur {(en - st) / 1e3:8.1f} us  stream/queue {r.get('Queue_Id', '')}")
PY
