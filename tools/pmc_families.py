'''rocprofv3 --pmc passes of one bench.py workload -> profiles/<tag>_pmc_<workload>.json: per kernel family (launches, HBM bytes per
launch, MFMA-busy fraction) and the whole step's HBM bytes.  Same corrections as tools/make_profile.py (MI355X_MICROARCH.md, HBM).

  python tools/pmc_families.py <pmc dir> <tag> <workload: bf16_train | bf16_infer | bf16_radarnet> [commit]

<pmc dir>: one sub-directory per pass (FETCH_SIZE, WRITE_SIZE, "SQ_VALU_MFMA_BUSY_CYCLES GRBM_GUI_ACTIVE") of
  rocprofv3 --pmc X --output-format csv -d DIR/X -o b -- python3 bench.py <workload flags> --graph 0 --steps 1 --warmup 1 --preheat-s 0 --no-cpu-baseline
Per-step figures: totals over the whole run divided by the number of forward passes in it (launches of the head's forward kernel).'''
import collections, csv, glob, hashlib, json, os, re, sys

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
pmc_dir, tag, workload = sys.argv[1:4]
head = sys.argv[4] if len(sys.argv) > 4 else 'unknown'
FAMILIES = [
    ('conv_b16_kernel 3x3 s1', r'conv_b16_kernel<DmaCfg<3, \d+, \d+, \d+, 1(, \d+)?>'),
    ('conv_b16_kernel 3x3 s2', r'conv_b16_kernel<DmaCfg<3, \d+, \d+, \d+, 2(, \d+)?>'),
    ('conv_b16_kernel 2x2 phases', r'conv_b16_kernel<DmaCfg<2,'),
    ('conv_b16_kernel 4x4 stem on the space-to-depth image', r'conv_b16_kernel<DmaCfg<4,'),
    ('conv1x1_b16_kernel', r'conv1x1_b16_kernel<'),
    ('conv_wgrad_split_kernel', r'conv_wgrad_split_kernel<'),
    ('conv_wgrad_tr_kernel', r'conv_wgrad_tr_kernel<'),
    ('conv_split_kernel (bf16 operands)', r'conv_split_kernel<'),
    ('BatchNorm / activation / fusion elementwise', r'(bn_act_|fuse_|head_bn_)'),
]


def clean(n):
    n = re.sub(r'\(anonymous namespace\)::|void ', '', n)
    return re.sub(r'\(ConvArgs\)|\(.*', '', n).strip()


# Every dispatch of the run is counted and divided by the number of STEPS the run made = launches of the output head's forward kernel,
# which runs exactly once per forward pass of every workload (a "second half of the dispatches" rule miscounts runs whose warm-up and
# measured parts differ in length).
pm = collections.defaultdict(lambda: collections.defaultdict(lambda: [0, 0.0]))
n_steps = 0
for f in glob.glob(os.path.join(pmc_dir, '**', '*counter_collection.csv'), recursive=True):
    rr = list(csv.DictReader(open(f)))
    first = rr[0]['Counter_Name'] if rr else None
    n_steps = max(n_steps, sum(1 for r in rr if r['Counter_Name'] == first and re.search(r'head_fwd', r['Kernel_Name'])))
    for r in rr:
        a = pm[clean(r['Kernel_Name'])][r['Counter_Name']]
        a[0] += 1
        a[1] += float(r['Counter_Value'])
n_steps = max(n_steps, 1)
kernels = {}
for n, cs in pm.items():
    e = {'launches': max(v[0] for v in cs.values())}
    for cn, (k, s) in cs.items():
        e[cn] = s
    kernels[n] = e


def csrc_sha():
    h = hashlib.sha256()
    d = os.path.join(ROOT, 'radar-camera-fusion-depth_amd', 'csrc')
    for name in sorted(os.listdir(d)):
        if not name.endswith(('.h', '.hip')):
            continue      # (a stray cache directory is not a kernel source)
        h.update(open(os.path.join(d, name), 'rb').read())
    return h.hexdigest()[:16]


def summarise(ks):
    n = sum(kernels[k]['launches'] for k in ks)
    e = {'launches': n / float(n_steps)}            # per step
    if all('FETCH_SIZE' in kernels[k] and 'WRITE_SIZE' in kernels[k] for k in ks):
        tot = sum((2.0 * kernels[k]['FETCH_SIZE'] + kernels[k]['WRITE_SIZE']) * 1024.0 for k in ks)
        e['hbm_bytes'] = tot / n_steps              # per step
        e['hbm_bytes_per_launch'] = tot / max(n, 1)
    if all('SQ_VALU_MFMA_BUSY_CYCLES' in kernels[k] and 'GRBM_GUI_ACTIVE' in kernels[k] for k in ks):
        act = sum(kernels[k]['GRBM_GUI_ACTIVE'] for k in ks)
        e['mfma_busy_fraction'] = sum(kernels[k]['SQ_VALU_MFMA_BUSY_CYCLES'] for k in ks) / (act / 8.0 * 1024.0) if act > 0 else None
    return e


out = {'_meta': {'head': head, 'csrc_sha': csrc_sha(), 'round': tag, 'workload': workload, 'steps_in_run': n_steps,
                 'corrections': 'FETCH_SIZE / WRITE_SIZE are KiB; FETCH_SIZE doubled (gfx950 tallies the 128-B requests of 16-B/lane reads as 64 B: '
                                'MI355X_MICROARCH.md, HBM); MFMA-busy = SQ_VALU_MFMA_BUSY_CYCLES / (GRBM_GUI_ACTIVE / 8 XCDs x 1024 SIMDs); separate '
                                '--pmc passes; per-step figures = totals over the run / launches of the head forward kernel (once per step)'},
       'whole_step': summarise(list(kernels))}
for key, rx in FAMILIES:
    ks = [k for k in kernels if re.match(rx, k)]
    if ks:
        out[key] = summarise(ks)
path = os.path.join(ROOT, 'profiles', '%s_pmc_%s.json' % (tag, workload))
json.dump(out, open(path, 'w'), indent=1, sort_keys=True)
print('wrote', path)
for k, v in out.items():
    if k != '_meta':
        print('%-52s launches/step %-7.1f HBM GB/step %-9s MFMA-busy %s' % (k, v['launches'], '%.3f' % (v['hbm_bytes'] / 1e9) if 'hbm_bytes' in v else '-',
                                                             '%.3f' % v['mfma_busy_fraction'] if v.get('mfma_busy_fraction') is not None else '-'))
