#!/bin/bash
# Register and spill figures of the kernels in the current sources: tools/kinfo.sh [extra hipcc flags]
cd $(dirname $0)/../raytracer_amd/csrc
/opt/rocm/bin/hipcc -O3 -std=c++17 --offload-arch=gfx950 -ffp-contract=off -fno-slp-vectorize -fPIC -Wno-unused-function -Wno-unused-value -Wno-unused-result "$@" -S --cuda-device-only -o /tmp/kinfo.s rt_device.hip 2>/dev/null
python3 - <<'PY'
import re
t=open('/tmp/kinfo.s').read()
for m in re.finditer(r'\.name:\s+(\S+)\n(.*?)\.wavefront_size', t, re.S):
    name=m.group(1); b=m.group(2)
    if 'shade' not in name and 'broad_kernelILi512' not in name and 'mfma' not in name and 'binned' not in name and 'walk' not in name and 'init' not in name: continue
    g=lambda k: re.search(k+r':\s+(\d+)', b).group(1)
    print(name[:60], 'vgpr', g(r'\.vgpr_count'), 'vspill', g(r'\.vgpr_spill_count'), 'sgpr', g(r'\.sgpr_count'), 'sspill', g(r'\.sgpr_spill_count'), 'scratch', g(r'\.private_segment_fixed_size'))
lines=t.split('\n')
st=[i for i,l in enumerate(lines) if l.startswith('_ZN12_GLOBAL__N_112shade_kernelILb0ELb0EEEvNS_8WfParamsE:')][0]
en=[i for i,l in enumerate(lines) if i>st and l.startswith('.Lfunc_end')][0]
n=sum(1 for l in lines[st:en] if l.strip() and not l.strip().startswith((';','.')) and not l.strip().endswith(':'))
print('shade static instrs', n)
PY
