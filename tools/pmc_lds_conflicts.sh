#!/bin/bash
# GPU box: LDS bank conflicts per kernel (the layout check the microarchitecture guide asks for): SQ_LDS_BANK_CONFLICT = extra LDS cycles,
# SQ_LDS_IDX_ACTIVE = all LDS-array cycles.
#   bash tools/pmc_lds_conflicts.sh <out dir> wgrad          both weight-gradient kernels over tools/wgrad_bench.py's layer shapes
#   bash tools/pmc_lds_conflicts.sh <out dir> step [bf16]    every kernel of one single-stream training step of bench.py
cd /tmp && export TMPDIR=/tmp && cd "$GRAFT_REPO_ROOT"
out=${1:-gpurun_out/lds}; what=${2:-wgrad}; dt=${3:-f32}
mkdir -p $out
if [ "$what" = "wgrad" ]; then
  RCF_WGRAD_BENCH_SWEEP=0 rocprofv3 --pmc SQ_LDS_BANK_CONFLICT SQ_LDS_IDX_ACTIVE --output-format csv -d $out/pmc -o l -- python3 tools/wgrad_bench.py 2 > $out/run.log 2>&1
else
  RCF_SINGLE_STREAM=1 RCF_BATCH_PACK=0 rocprofv3 --pmc SQ_LDS_BANK_CONFLICT SQ_LDS_IDX_ACTIVE --output-format csv -d $out/pmc -o l -- python3 bench.py --dtype $dt --graph 0 --steps 1 --warmup 1 --preheat-s 0 --no-cpu-baseline --no-side-leg --no-other-configs > $out/run.log 2>&1
fi
python3 - "$out" "$what" <<'PY'
import csv, glob, collections, re, sys
out, what = sys.argv[1], sys.argv[2]
acc = collections.defaultdict(lambda: collections.defaultdict(float))
cnt = collections.Counter()
for f in glob.glob(out + '/pmc/**/*counter_collection.csv', recursive=True):
    for r in csv.DictReader(open(f)):
        n = re.sub(r'\(anonymous namespace\)::|void ', '', r['Kernel_Name'])
        n = re.sub(r'\(ConvArgs\)|\(.*', '', n).strip()
        if what == 'wgrad' and 'wgrad' not in n:
            continue
        acc[n][r['Counter_Name']] += float(r['Counter_Value'])
        if r['Counter_Name'] == 'SQ_LDS_IDX_ACTIVE':
            cnt[n] += 1
print('%-84s %8s %16s %16s %8s' % ('kernel', 'launches', 'LDS_IDX_ACTIVE', 'BANK_CONFLICT', 'ratio'))
rows = sorted(acc.items(), key=lambda kv: -kv[1].get('SQ_LDS_IDX_ACTIVE', 0.0))
for n, c in rows[:40]:
    a, b = c.get('SQ_LDS_IDX_ACTIVE', 0.0), c.get('SQ_LDS_BANK_CONFLICT', 0.0)
    if a <= 0:
        continue
    print('%-84s %8d %16.0f %16.0f %8.4f' % (n[:84], cnt[n], a, b, b / a if a else 0.0))
PY
