'''Summarise rocprofv3 --pmc counter_collection CSVs (one directory per pass) per kernel family.
usage: python tools/pmc_summary.py <dir with */*_counter_collection.csv> <out.json> [skip_first_n_dispatch_fraction]'''
import csv, glob, json, re, sys, collections

base, out = sys.argv[1], sys.argv[2]
agg = collections.defaultdict(lambda: collections.defaultdict(lambda: [0, 0.0]))
for f in glob.glob(base + '/*/*_counter_collection.csv'):
    rows = list(csv.DictReader(open(f)))
    if not rows:
        continue
    # keep only the second half of the dispatches (= the timed step after one warm-up step)
    ids = sorted(set(int(r['Dispatch_Id']) for r in rows))
    cut = ids[len(ids) // 2]
    for r in rows:
        if int(r['Dispatch_Id']) < cut:
            continue
        name = re.sub(r'\(anonymous namespace\)::|void ', '', r['Kernel_Name'])
        name = re.sub(r'\(ConvArgs\)|\(.*', '', name).strip()
        a = agg[name][r['Counter_Name']]
        a[0] += 1
        a[1] += float(r['Counter_Value'])
summary = {}
for name, cs in agg.items():
    e = {'launches': max(v[0] for v in cs.values())}
    for cn, (n, s) in cs.items():
        e[cn] = s
    if 'FETCH_SIZE' in e and 'WRITE_SIZE' in e:
        # units: KiB.  gfx950: FETCH_SIZE counts 128-B requests as 64 B for wide (16 B/lane) coalesced reads -> x2
        # (/opt/skills/guides/MI355X_MICROARCH.md, section HBM); WRITE_SIZE matched the output bytes exactly on a conv launch.
        e['hbm_bytes_per_launch'] = (2.0 * e['FETCH_SIZE'] + e['WRITE_SIZE']) * 1024.0 / e['launches']
    summary[name] = e
json.dump(summary, open(out, 'w'), indent=1, sort_keys=True)
fam = [k for k in summary if k.startswith('conv_fwd_kernel<FwdCfg<3, 3, 0, 1')]
n = sum(summary[k]['launches'] for k in fam)
if n and all('hbm_bytes_per_launch' in summary[k] for k in fam):
    tot = sum(summary[k]['hbm_bytes_per_launch'] * summary[k]['launches'] for k in fam)
    print('3x3 s1 conv_fwd family: %d launches, %.3f GB HBM per launch (FETCH x2 + WRITE)' % (n, tot / n / 1e9))
