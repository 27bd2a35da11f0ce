#!/bin/bash
# One recorded measurement of the bench workload on the GPU box: bench line (with CPU baseline), rocprofv3 kernel
# trace + stats, and the HBM traffic / SQ counters in passes of their own.  Output under gpurun_out/$1/.
#   gpurun -- tools/profile_run.sh r1h
# The traced and counted runs use RT_OVERLAP=1 (one sub-pipeline: kernels run one after the other), so that a kernel's
# traced duration is its own -- the same condition under which bench.py measures roofline.kernel_ms.
set -e
tag=${1:-rec}
root=${GRAFT_REPO_ROOT:-$PWD}
out=$root/gpurun_out/$tag
mkdir -p $out
cd /tmp && export TMPDIR=/tmp
timeout -k 10 400 python $root/bench.py > $out/bench.json 2> $out/bench.err
export RT_OVERLAP=1
timeout -s KILL 180 rocprofv3 --kernel-trace --stats --output-format csv -d $out/trace -o vm -- python $root/bench.py --steps 5 --warmup 2 --roofline-frames 0 --no-cpu-baseline --no-walk --no-in-contract > $out/trace.log 2>&1
timeout -s KILL 120 rocprofv3 --pmc FETCH_SIZE --output-format csv -d $out/pmc_fetch -o vm -- python $root/bench.py --steps 1 --warmup 0 --roofline-frames 0 --no-cpu-baseline --no-walk --no-in-contract > $out/pmc_fetch.log 2>&1
timeout -s KILL 120 rocprofv3 --pmc WRITE_SIZE --output-format csv -d $out/pmc_write -o vm -- python $root/bench.py --steps 1 --warmup 0 --roofline-frames 0 --no-cpu-baseline --no-walk --no-in-contract > $out/pmc_write.log 2>&1
timeout -s KILL 120 rocprofv3 --pmc SQ_INSTS_VALU SQ_ACTIVE_INST_VALU SQ_THREAD_CYCLES_VALU SQ_WAVE_CYCLES SQ_BUSY_CYCLES SQ_WAIT_INST_ANY SQ_INSTS_LDS SQ_INSTS_SALU GRBM_GUI_ACTIVE --output-format csv -d $out/pmc_sq -o vm -- python $root/bench.py --steps 1 --warmup 0 --roofline-frames 0 --no-cpu-baseline --no-walk --no-in-contract > $out/pmc_sq.log 2>&1
timeout -s KILL 120 rocprofv3 --pmc SQ_INSTS SQ_INSTS_VMEM_RD SQ_INSTS_VMEM_WR SQ_INSTS_BRANCH SQ_WAIT_ANY SQ_INSTS_VALU_FMA_F64 SQ_INSTS_VALU_MUL_F64 SQ_INSTS_VALU_ADD_F64 --output-format csv -d $out/pmc_sq2 -o vm -- python $root/bench.py --steps 1 --warmup 0 --roofline-frames 0 --no-cpu-baseline --no-walk --no-in-contract > $out/pmc_sq2.log 2>&1
timeout -s KILL 120 rocprofv3 --pmc SQ_INSTS_MFMA SQ_VALU_MFMA_BUSY_CYCLES SQ_VALU_MFMA_COEXEC_CYCLES SQ_INSTS_VALU_MFMA_MOPS_F16 SQ_ACTIVE_INST_ANY SQ_ACTIVE_INST_LDS --output-format csv -d $out/pmc_sq3 -o vm -- python $root/bench.py --steps 1 --warmup 0 --roofline-frames 0 --no-cpu-baseline --no-walk --no-in-contract > $out/pmc_sq3.log 2>&1
# the same frames with candidate solids from the grid walk (RT_ACCEL=1 makes it the context's default)
RT_ACCEL=1 timeout -s KILL 180 rocprofv3 --kernel-trace --stats --output-format csv -d $out/trace_walk -o vm -- python $root/bench.py --steps 5 --warmup 2 --roofline-frames 0 --no-cpu-baseline --no-walk --no-in-contract > $out/trace_walk.log 2>&1
unset RT_OVERLAP
cat $out/bench.json
