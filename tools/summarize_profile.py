#!/usr/bin/env python
"""Condense one tools/profile_run.sh output directory into the files kept under profiles/:
   tools/summarize_profile.py gpurun_out/r1g profiles/r1 g_pipeline
writes <prefix>_bench.json, <prefix>_kernel_stats.csv, <prefix>_summary.md and traffic.json (HBM bytes per frame of
every kernel from the FETCH_SIZE / WRITE_SIZE passes; FETCH_SIZE doubled for the 16-byte-per-lane streaming reads as
gfx950 tallies 128-byte requests at 64 bytes)."""
import csv, glob, json, os, re, shutil, sys, collections

src, dst, prefix = sys.argv[1], sys.argv[2], sys.argv[3]
os.makedirs(dst, exist_ok=True)

def short(name):
    name = name.replace('(anonymous namespace)::', '').replace('void ', '')
    return re.sub(r'\(.*', '', name)[:48]

def pmc(d):
    tot = collections.defaultdict(float); n = collections.defaultdict(int)
    for f in glob.glob(os.path.join(src, d, '*counter_collection.csv')):
        for row in csv.DictReader(open(f)):
            k = (short(row['Kernel_Name']), row['Counter_Name'])
            tot[k] += float(row['Counter_Value']); n[k] += 1
    return tot, n

bench = json.loads(open(os.path.join(src, 'bench.json')).read().strip().splitlines()[-1])
json.dump(bench, open(os.path.join(dst, prefix + '_bench.json'), 'w'), indent=1)
shutil.copy(os.path.join(src, 'trace', 'vm_kernel_stats.csv'), os.path.join(dst, prefix + '_kernel_stats.csv'))
log = open(os.path.join(src, 'trace.log')).read()
m = re.search(r'\{"metric".*\}', log)
frames = json.loads(m.group(0))['steps'] + json.loads(m.group(0))['warmup'] if m else None

lines = ['# ' + prefix, '', 'bench line: `' + json.dumps({k: bench[k] for k in ('value', 'unit', 'ms_per_step', 'n_gpus')}) + '`',
         '', '## rocprofv3 --kernel-trace --stats (%s frames traced, RT_OVERLAP=1: kernels one after the other)' % frames, '',
         '| kernel | calls | avg us | total ms | ms per frame | % |', '|---|---|---|---|---|---|']
for row in csv.DictReader(open(os.path.join(src, 'trace', 'vm_kernel_stats.csv'))):
    tot = float(row['TotalDurationNs']) / 1e6
    lines.append('| %s | %s | %.1f | %.2f | %s | %s |' % (short(row['Name']), row['Calls'], float(row['AverageNs']) / 1e3, tot,
                                                         '%.3f' % (tot / frames) if frames else '-', row['Percentage']))
wk = os.path.join(src, 'trace_walk', 'vm_kernel_stats.csv')
if os.path.exists(wk):
    shutil.copy(wk, os.path.join(dst, prefix + '_walk_kernel_stats.csv'))
    lines += ['', '## the same frames with RT_OPT_ACCEL = 1 (walk_kernel in place of the scan), rocprofv3 --kernel-trace --stats', '',
              '| kernel | calls | avg us | total ms | ms per frame | % |', '|---|---|---|---|---|---|']
    for row in csv.DictReader(open(wk)):
        tot = float(row['TotalDurationNs']) / 1e6
        lines.append('| %s | %s | %.1f | %.2f | %s | %s |' % (short(row['Name']), row['Calls'], float(row['AverageNs']) / 1e3, tot,
                                                             '%.3f' % (tot / frames) if frames else '-', row['Percentage']))
traffic = {}
lines += ['', '## HBM traffic per frame (separate --pmc passes, 1 frame each)', '',
          '| kernel | launches | FETCH_SIZE KB (raw) | WRITE_SIZE KB | bytes per frame (fetch x2 + write) |', '|---|---|---|---|---|']
ft, fn = pmc('pmc_fetch'); wt, _ = pmc('pmc_write')
for (k, c), v in sorted(ft.items()):
    if 'rocclr' in k or k.startswith('at::'): continue
    w = wt.get((k, 'WRITE_SIZE'), 0.0)
    b = (2 * v + w) * 1024
    traffic[k] = {'launches': fn[(k, c)], 'fetch_kb_raw': v, 'write_kb': w, 'bytes_per_frame': b}
    lines.append('| %s | %d | %.0f | %.0f | %.3e |' % (k, fn[(k, c)], v, w, b))
st, sn = pmc('pmc_sq')
st2, _ = pmc('pmc_sq2')
st.update(st2)
st3, _ = pmc('pmc_sq3')
st.update(st3)
if st:
    lines += ['', '## SQ counters per frame', '', '| kernel | counter | total |', '|---|---|---|']
    for (k, c), v in sorted(st.items()):
        if 'rocclr' in k or k.startswith('at::'): continue
        lines.append('| %s | %s | %.4g |' % (k, c, v))
# counters bench.py reads back: VALU wave-instructions (roofline.valu_issue_frac), FP64 add / mul / fma wave-instructions
# (executed FLOP of shade_kernel), lanes active per VALU instruction, SALU share
for k in traffic:
    for key, counter in (('valu_insts', 'SQ_INSTS_VALU'), ('fp64_add', 'SQ_INSTS_VALU_ADD_F64'), ('fp64_mul', 'SQ_INSTS_VALU_MUL_F64'),
                         ('fp64_fma', 'SQ_INSTS_VALU_FMA_F64'), ('thread_cycles_valu', 'SQ_THREAD_CYCLES_VALU'),
                         ('active_inst_valu', 'SQ_ACTIVE_INST_VALU'), ('salu_insts', 'SQ_INSTS_SALU'), ('insts', 'SQ_INSTS'),
                         ('mfma_insts', 'SQ_INSTS_MFMA'), ('wave_cycles', 'SQ_WAVE_CYCLES'), ('wait_inst_any', 'SQ_WAIT_INST_ANY')):
        v = st.get((k, counter))
        if v is not None: traffic[k][key] = v
open(os.path.join(dst, prefix + '_summary.md'), 'w').write('\n'.join(lines) + '\n')
json.dump({'source': prefix, 'workload': bench.get('config', {}).get('workload'), 'note': 'HBM bytes per frame and kernel: (2 x FETCH_SIZE + WRITE_SIZE) KB from rocprofv3 --pmc passes',
           'kernels': traffic}, open(os.path.join(dst, 'traffic.json'), 'w'), indent=1)
print('\n'.join(lines))
