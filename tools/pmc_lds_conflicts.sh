#!/bin/bash
# GPU box: LDS bank conflicts of the weight-gradient kernels (the transposing reads' layout check the microarchitecture guide asks for):
# SQ_LDS_BANK_CONFLICT = extra LDS cycles, SQ_LDS_IDX_ACTIVE = all LDS-array cycles, per kernel, from tools/wgrad_bench.py (both kernels).
cd /tmp && export TMPDIR=/tmp && cd "$GRAFT_REPO_ROOT"
out=${1:-gpurun_out/lds}
mkdir -p $out
RCF_WGRAD_BENCH_SWEEP=0 rocprofv3 --pmc SQ_LDS_BANK_CONFLICT SQ_LDS_IDX_ACTIVE --output-format csv -d $out/pmc -o l -- python3 tools/wgrad_bench.py 2 > $out/run.log 2>&1
python3 - "$out" <<'PY'
import csv, glob, collections, re, sys
out = sys.argv[1]
acc = collections.defaultdict(lambda: collections.defaultdict(float))
cnt = collections.Counter()
for f in glob.glob(out + '/pmc/**/*counter_collection.csv', recursive=True):
    for r in csv.DictReader(open(f)):
        n = re.sub(r'\(anonymous namespace\)::|void ', '', r['Kernel_Name'])
        n = re.sub(r'\(ConvArgs\)|\(.*', '', n).strip()
        if 'wgrad' not in n:
            continue
        acc[n][r['Counter_Name']] += float(r['Counter_Value'])
        if r['Counter_Name'] == 'SQ_LDS_IDX_ACTIVE':
            cnt[n] += 1
print('%-74s %8s %16s %16s %8s' % ('kernel', 'launches', 'LDS_IDX_ACTIVE', 'BANK_CONFLICT', 'ratio'))
for n, c in sorted(acc.items()):
    a, b = c.get('SQ_LDS_IDX_ACTIVE', 0.0), c.get('SQ_LDS_BANK_CONFLICT', 0.0)
    print('%-74s %8d %16.0f %16.0f %8.4f' % (n[:74], cnt[n], a, b, b / a if a else 0.0))
PY
