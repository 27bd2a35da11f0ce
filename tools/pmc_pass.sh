#!/bin/bash
# Counter passes over one frame of the bench workload, one rocprofv3 run per pass (a pass = one quoted list of
# counters that fits the per-block slots).  Output: gpurun_out/$tag/pass<k>/ and a per-kernel summary on stdout.
#   gpurun -- tools/pmc_pass.sh tag "SQ_A SQ_B" "TA_X TCP_Y" ...
tag=$1; shift
root=${GRAFT_REPO_ROOT:-$PWD}
out=$root/gpurun_out/$tag
mkdir -p $out
cd /tmp && export TMPDIR=/tmp
k=0
for pass in "$@"; do
  k=$((k+1))
  timeout -s KILL 120 rocprofv3 --pmc $pass --output-format csv -d $out/pass$k -o vm -- python $root/bench.py --steps 1 --warmup 0 --no-cpu-baseline --no-walk --roofline-frames 0 > $out/pass$k.log 2>&1 || { echo "pass $k failed"; tail -5 $out/pass$k.log; exit 1; }
done
python $root/tools/pmc_summary.py $out | grep -v "at::\|rocclr\|set_count\|elementwise"
