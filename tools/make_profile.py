'''Turn one round's rocprofv3 outputs into the committed files under profiles/.

  python tools/make_profile.py <kernel-trace dir or .db> <pmc dir> <bench json line file> [round tag]

<kernel-trace>: output of `rocprofv3 --kernel-trace --stats -d DIR -o r -- python3 bench.py --no-cpu-baseline` (rocpd .db)
<pmc dir>     : one sub-directory per `rocprofv3 --pmc X --output-format csv -d DIR/X -o b -- python3 bench.py --steps 1 --warmup 1
                --no-cpu-baseline` pass (FETCH_SIZE, WRITE_SIZE, "SQ_VALU_MFMA_BUSY_CYCLES GRBM_GUI_ACTIVE")
Writes profiles/<tag>_bench_kernel_stats.csv, <tag>_pmc_bench.json, <tag>_summary.md, <tag>_bench_line.json.
'''
import collections, csv, glob, json, os, re, sqlite3, sys

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
trace, pmc_dir, line_file = sys.argv[1:4]
tag = sys.argv[4] if len(sys.argv) > 4 else 'r01'
prof = os.path.join(ROOT, 'profiles')

FAMILIES = [   # (bench.py KERNEL_NAMES entry, regular expression on the cleaned kernel name).  The event families of bench.py are kernel-id
    # families: 3x3 stride 1 and 3x3 stride 2 are different ids, so they are different rows here too (SplitCfg's last parameter is LSTEP)
    ('conv_split_kernel 3x3 s1', r'conv_split_kernel<SplitCfg<3, \d+, \d+, \d+, \d+, 1(, \d+)?>'),
    ('conv_split_kernel 3x3 s2', r'conv_split_kernel<SplitCfg<3, \d+, \d+, \d+, \d+, 2(, \d+)?>'),
    ('conv_split_kernel 2x2 phases', r'conv_split_kernel<SplitCfg<2,'),
    ('conv_split_kernel 4x4 stem on the space-to-depth image', r'conv_split_kernel<SplitCfg<4,'),
    ('conv_wgrad_split_kernel', r'conv_wgrad_split_kernel<'),
    ('conv_wgrad_split_kernel 3x3', r'conv_wgrad_split_kernel<WsCfg<\d+, \d+, 3,'),
    ('conv_wgrad_tr_kernel', r'conv_wgrad_tr_kernel<'),
    ('conv_wgrad_tr_kernel 3x3', r'conv_wgrad_tr_kernel<WtCfg<\d+, \d+, 3,'),
    ('conv_fwd_kernel 3x3 s1', r'conv_fwd_kernel<FwdCfg<3, 3, 0, 1'),
    # the kernels north_star's "3x3 encoder convs" run on: 3x3 forward / input gradient (64-co and small-layer tiles, both strides) and
    # the 3x3 weight gradients -- the decoder's launches of the same kernels are in this row too (PMC rows are per kernel, not per layer)
    ('kernels of the encoder 3x3 convolutions', r'conv_split_kernel<SplitCfg<3, 2,|conv_split_kernel<SplitCfg<3, 1, \d+, 2,|conv_split_kernel<SplitCfg<3, \d+, \d+, \d+, \d+, 2(, \d+)?>|conv_wgrad_split_kernel<WsCfg<\d+, \d+, 3,|conv_wgrad_tr_kernel<WtCfg<\d+, \d+, 3,'),
]


def clean(n):
    n = re.sub(r'\(anonymous namespace\)::|void ', '', n)
    return re.sub(r'\(ConvArgs\)|\(.*', '', n).strip()


# ---- kernel trace
db = trace if trace.endswith('.db') else sorted(glob.glob(os.path.join(trace, '**', '*.db'), recursive=True))[0]
cur = sqlite3.connect(db).cursor()
rows = list(cur.execute('select name, start, end from kernels order by start'))
agg = collections.defaultdict(lambda: [0, 0.0])
for n, s, e in rows:
    a = agg[clean(n)]
    a[0] += 1
    a[1] += (e - s) / 1e3
line = json.loads([l for l in open(line_file) if l.startswith('{')][-1])
nstep = sum(1 for n, s_, e_ in rows if 'head_fwd' in n) or (line['steps'] + line['warmup'])   # the head's forward kernel runs once per step
tot = sum(v[1] for v in agg.values())
with open(os.path.join(prof, tag + '_bench_kernel_stats.csv'), 'w') as f:
    w = csv.writer(f)
    w.writerow(['Name', 'Calls', 'TotalDurationUs', 'AverageUs', 'Percentage'])
    for n, (cnt, us) in sorted(agg.items(), key=lambda kv: -kv[1][1]):
        w.writerow([n, cnt, '%.1f' % us, '%.2f' % (us / cnt), '%.2f' % (100 * us / tot)])

# ---- PMC passes
# every dispatch of the PMC run counts; launches are reported PER STEP = total / (launches of the head's forward kernel, which runs once
# per step) -- bench.py's run has warm-up, timed and a few untimed steps (host-time probe), all with the same launches (RCF_BATCH_PACK=0)
pm = collections.defaultdict(lambda: collections.defaultdict(lambda: [0, 0.0]))
n_steps_pmc = 0
for f in glob.glob(os.path.join(pmc_dir, '**', '*counter_collection.csv'), recursive=True):
    rr = list(csv.DictReader(open(f)))
    first = rr[0]['Counter_Name'] if rr else None
    n_steps_pmc = max(n_steps_pmc, sum(1 for r in rr if r['Counter_Name'] == first and re.search(r'head_fwd', r['Kernel_Name'])))
    for r in rr:
        a = pm[clean(r['Kernel_Name'])][r['Counter_Name']]
        a[0] += 1
        a[1] += float(r['Counter_Value'])
n_steps_pmc = max(n_steps_pmc, 1)
kernels = {}
for n, cs in pm.items():
    nl = max(v[0] for v in cs.values())
    e = {'launches': nl / float(n_steps_pmc)}
    for cn, (k, s) in cs.items():
        e[cn] = s / n_steps_pmc            # per step
    if 'FETCH_SIZE' in e and 'WRITE_SIZE' in e:
        e['hbm_bytes_per_launch'] = (2.0 * e['FETCH_SIZE'] + e['WRITE_SIZE']) * 1024.0 / e['launches']
    kernels[n] = e
import hashlib, subprocess
def _csrc_sha():
    h = hashlib.sha256()
    d = os.path.join(ROOT, 'radar-camera-fusion-depth_amd', 'csrc')
    for name in sorted(os.listdir(d)):
        if not name.endswith(('.h', '.hip')):
            continue      # (a stray cache directory is not a kernel source)
        h.update(open(os.path.join(d, name), 'rb').read())
    return h.hexdigest()[:16]
try:
    head = subprocess.run(['git', '-C', ROOT, 'rev-parse', '--short', 'HEAD'], capture_output=True, text=True).stdout.strip() or 'unknown'
except Exception:
    head = 'unknown'
if len(sys.argv) > 5:
    head = sys.argv[5]      # the GPU box has no .git: the caller passes the commit the tree was built from
out = {'_meta': {'head': head, 'csrc_sha': _csrc_sha(), 'round': tag},
       'kernels': kernels,
       'corrections': 'FETCH_SIZE and WRITE_SIZE are KiB; FETCH_SIZE doubled (gfx950 counts 128-B requests of 16-B/lane coalesced reads as '
                      '64 B, MI355X_MICROARCH.md HBM section); MFMA-busy fraction = SQ_VALU_MFMA_BUSY_CYCLES / (GRBM_GUI_ACTIVE / 8 XCDs x 1024 SIMDs)',
       'source': 'rocprofv3 --pmc {FETCH_SIZE | WRITE_SIZE | SQ_VALU_MFMA_BUSY_CYCLES GRBM_GUI_ACTIVE} -- python3 bench.py --steps 1 --warmup 1 '
                 '--no-cpu-baseline (separate passes; per-step figures = totals / launches of the head forward kernel)', 'pmc_steps_in_run': n_steps_pmc}
fam_rows = []
for key, prefix in FAMILIES:
    ks = [k for k in kernels if re.match(prefix, k)]
    if not ks:
        continue
    n = sum(kernels[k]['launches'] for k in ks)
    e = {'launches': n}
    if all('hbm_bytes_per_launch' in kernels[k] for k in ks):
        e['hbm_bytes_per_launch'] = sum(kernels[k]['hbm_bytes_per_launch'] * kernels[k]['launches'] for k in ks) / n
    if all('SQ_VALU_MFMA_BUSY_CYCLES' in kernels[k] and 'GRBM_GUI_ACTIVE' in kernels[k] for k in ks):
        busy = sum(kernels[k]['SQ_VALU_MFMA_BUSY_CYCLES'] for k in ks)
        act = sum(kernels[k]['GRBM_GUI_ACTIVE'] for k in ks)
        e['mfma_busy_fraction'] = busy / (act / 8.0 * 1024.0)
    out[key] = e
    fam_rows.append((key, e))
json.dump(out, open(os.path.join(prof, tag + '_pmc_bench.json'), 'w'), indent=1, sort_keys=True)
json.dump(line, open(os.path.join(prof, tag + '_bench_line.json'), 'w'), indent=1)

# ---- summary
with open(os.path.join(prof, tag + '_summary.md'), 'w') as f:
    f.write('# %s profile (tree at commit %s, csrc hash %s)\n\n' % (tag, head, out['_meta']['csrc_sha']))
    f.write('`rocprofv3 --kernel-trace --stats -- python3 bench.py --graph 0 --steps %d --warmup %d --preheat-s 0 --no-cpu-baseline --no-side-leg --no-other-configs` on MI355X (gfx950):\n'
            % (line['steps'], line['warmup']))
    f.write('FusionNet fp32 training, batch 8, 900x1600; %d steps in the trace.\n' % nstep)
    f.write('The traced and counted runs are SINGLE-STREAM (RCF_SINGLE_STREAM=1): a kernel\'s duration and counters are its own.  The default step runs the weight gradients and the encoder\'s depth branch on two side streams, so its wall clock is SHORTER than the sum of kernel times below.\n')
    f.write('Total kernel time %.1f ms = %.1f ms/step (bench wall clock without the profiler: see %s_bench_line.json).\n\n'
            % (tot / 1e3, tot / 1e3 / nstep, tag))
    rl = line.get('roofline', {})
    f.write('Arithmetic of the fp32 metric (bench.py `config.arithmetic`): %s\n\n' % line.get('config', {}).get('arithmetic', '-'))
    f.write('Dominant kernel family for `roofline`: `%s`: %.4f ms per launch from events inside bench.py, %.1f algorithmic fp32 TFLOP/s.\n\n'
            % (rl.get('kernel'), rl.get('avg_launch_ms', 0), rl.get('algorithmic_fp32_tflops', rl.get('achieved', 0))))
    f.write('PMC passes (`%s_pmc_bench.json`):\n\n| family | launches per step | MFMA-busy fraction | HBM GB per launch (FETCH x2 + WRITE) |\n|---|---|---|---|\n' % tag)
    for key, e in fam_rows:
        f.write('| %s | %.1f | %s | %s |\n' % (key, e['launches'], '%.3f' % e['mfma_busy_fraction'] if 'mfma_busy_fraction' in e else '-',
                                             '%.4f' % (e['hbm_bytes_per_launch'] / 1e9) if 'hbm_bytes_per_launch' in e else '-'))
    f.write('\n')
    f.write('| % | ms/step | calls/step | avg us | kernel |\n|---|---|---|---|---|\n')
    for n, (cnt, us) in sorted(agg.items(), key=lambda kv: -kv[1][1])[:48]:
        f.write('| %.2f | %.2f | %.1f | %.1f | `%s` |\n' % (100 * us / tot, us / 1e3 / nstep, cnt / nstep, us / cnt, n[:90]))
print('wrote', tag, 'files; total %.1f ms/step over %d steps' % (tot / 1e3 / nstep, nstep))
for key, e in fam_rows:
    print(key, e)
