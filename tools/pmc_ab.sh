#!/bin/bash
# The same counters for several builds of librt_hip.so on one C3 frame: tools/pmc_ab.sh "<counters>" lib1.so lib2.so ...
ctr=$1; shift
root=${GRAFT_REPO_ROOT:-$PWD}
cd /tmp && export TMPDIR=/tmp
export RT_OVERLAP=1
for lib in "$@"; do
  tag=$(basename $lib .so)
  export RT_HIP_LIB=$root/$lib
  timeout -s KILL 120 rocprofv3 --pmc $ctr --output-format csv -d $root/gpurun_out/pmc_ab/$tag -o vm -- python $root/bench.py --steps 1 --warmup 0 --roofline-frames 0 --no-cpu-baseline --no-walk --no-in-contract > $root/gpurun_out/pmc_ab/$tag.log 2>&1 || exit 1
  python3 - $root/gpurun_out/pmc_ab/$tag $tag <<'PY'
import csv,sys,glob,collections
tot=collections.defaultdict(float)
for f in glob.glob(sys.argv[1]+'/*counter_collection.csv'):
    for r in csv.DictReader(open(f)):
        n=r['Kernel_Name']
        k='shade_later' if 'shade_kernel<false' in n else ('shade0' if 'shade_kernel<true' in n else ('binned' if 'binned' in n else None))
        if k: tot[(k,r['Counter_Name'])]+=float(r['Counter_Value'])
for k,v in sorted(tot.items()): print(sys.argv[2],k[0],k[1],'%.4g'%v)
PY
done
