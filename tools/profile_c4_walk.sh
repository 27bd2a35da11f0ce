#!/bin/bash
# SQ counters of one C4 frame (walk_kernel is the candidate source there): gpurun -- tools/profile_c4_walk.sh <tag>
set -e
tag=${1:-c4walk}
root=${GRAFT_REPO_ROOT:-$PWD}
out=$root/gpurun_out/$tag
mkdir -p $out
cd /tmp && export TMPDIR=/tmp
export RT_OVERLAP=1
A="--workload c4 --steps 1 --warmup 0 --roofline-frames 0 --no-cpu-baseline --no-walk --no-in-contract"
timeout -s KILL 200 rocprofv3 --kernel-trace --stats --output-format csv -d $out/trace -o vm -- python $root/bench.py $A > $out/trace.log 2>&1
timeout -s KILL 200 rocprofv3 --pmc SQ_INSTS_VALU SQ_ACTIVE_INST_VALU SQ_THREAD_CYCLES_VALU SQ_WAVE_CYCLES SQ_BUSY_CYCLES SQ_WAIT_INST_ANY SQ_INSTS_SALU GRBM_GUI_ACTIVE --output-format csv -d $out/pmc_sq -o vm -- python $root/bench.py $A > $out/pmc_sq.log 2>&1
timeout -s KILL 200 rocprofv3 --pmc SQ_INSTS SQ_INSTS_VMEM_RD SQ_INSTS_VMEM_WR SQ_INSTS_BRANCH SQ_WAIT_ANY SQ_ACTIVE_INST_VMEM SQ_INSTS_SMEM --output-format csv -d $out/pmc_sq2 -o vm -- python $root/bench.py $A > $out/pmc_sq2.log 2>&1
timeout -s KILL 200 rocprofv3 --pmc FETCH_SIZE --output-format csv -d $out/pmc_fetch -o vm -- python $root/bench.py $A > $out/pmc_fetch.log 2>&1
timeout -s KILL 200 rocprofv3 --pmc TCP_TOTAL_ACCESSES_sum TCP_TCC_READ_REQ_sum TCP_PENDING_STALL_CYCLES_sum TCP_TCP_TA_DATA_STALL_CYCLES_sum --output-format csv -d $out/pmc_tcp -o vm -- python $root/bench.py $A > $out/pmc_tcp.log 2>&1
python3 - $out <<'PY'
import csv,sys,glob,collections
out=sys.argv[1]
for d in ('pmc_sq','pmc_sq2','pmc_fetch','pmc_tcp'):
    tot=collections.defaultdict(float)
    for f in glob.glob(f'{out}/{d}/**/*counter_collection.csv', recursive=True):
        for r in csv.DictReader(open(f)):
            k=r['Kernel_Name'].split('(')[0][-40:]
            tot[(k,r['Counter_Name'])]+=float(r['Counter_Value'])
    for k,v in sorted(tot.items()):
        if 'walk' in k[0] or 'shade' in k[0]: print(k[0],k[1],'%.4g'%v)
for f in glob.glob(f'{out}/trace/**/*kernel_stats.csv', recursive=True): print(open(f).read())
PY
