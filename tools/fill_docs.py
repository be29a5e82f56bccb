'''Fill the R3_* placeholders of DESIGN.md / README.md from the committed round-3 profile files (profiles/r03_*.json, *_kernels.txt).
    python tools/fill_docs.py "144-154"        # argument: the range of the metric seen across boxes this round'''
import json
import os
import re
import sys

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
P = os.path.join(ROOT, 'profiles')


def load(name):
    return json.load(open(os.path.join(P, name)))


main = load('r03_bench_default_line.json')
rl = main['roofline']
pmc = load('r03_pmc_bench.json')
fam = pmc.get('conv_split_kernel 3x3 s1', {})
busy = fam.get('mfma_busy_fraction')
traffic = fam.get('hbm_bytes_per_launch')
alg_gb = rl.get('algorithmic_gbytes_per_launch') or 0.2436


def total(fn):
    try:
        return float(re.search(r'total kernel time ([0-9.]+)', open(os.path.join(P, fn)).read()).group(1))
    except Exception:
        return float('nan')


vals = {
    'R3_F32_RANGE': sys.argv[1] if len(sys.argv) > 1 else '144-154',
    'R3_F32_MS': '%.1f' % main['ms_per_step'],
    'R3_F32X': '%.2f' % rl['useful_frac_of_f32_mfma_peak'],
    'R3_F32': '%.1f' % main['value'],
    'R3_3P_MS': '%.1f' % load('r03_bench_f32_3plane.json')['ms_per_step'],
    'R3_3P': '%.1f' % load('r03_bench_f32_3plane.json')['value'],
    'R3_ALG': '%.1f' % main['algorithmic_tflops'],
    'R3_DOM': '%.0f' % rl['algorithmic_fp32_tflops'],
    'R3_FRAC': '%.2f' % rl['frac'],
    'R3_BUSY': '%.2f' % busy if busy else 'n/a',
    'R3_TRAF': '%.2f' % (traffic / 1e9 / alg_gb) if traffic else 'n/a',
    'R3_USEFUL': '%.3f' % rl['useful_frac'],
    'R3_ENCFRAC': '%.2f' % rl['encoder_3x3']['frac_of_pipe_peak'],
    'R3_ENC': '%.0f' % rl['encoder_3x3']['tflops_algorithmic'],
    'R3_BF16_MS': '%.1f' % load('r03_bench_bf16.json')['ms_per_step'],
    'R3_BF16': '%.1f' % load('r03_bench_bf16.json')['value'],
    'R3_INF_MS': '%.1f' % load('r03_bench_infer.json')['ms_per_step'],
    'R3_INF32': '%.0f' % load('r03_bench_infer_f32.json')['value'],
    'R3_INF': '%.0f' % load('r03_bench_infer.json')['value'],
    'R3_RAD_MS': '%.1f' % load('r03_bench_radarnet.json')['ms_per_step'],
    'R3_RAD32': '%.0f' % load('r03_bench_radarnet_f32.json')['value'],
    'R3_RAD': '%.0f' % load('r03_bench_radarnet.json')['value'],
    'R3_CPU': '%.3f' % main['cpu_baseline']['value'],
    'R3_PCT': '%.0f' % (100.0 * main['value'] / 158.0),
    'R3_KTB': '%.1f' % total('r03_bf16_train_kernels.txt'),
    'R3_KT': '%.1f' % total('r03_fp32_train_kernels.txt'),
}
for doc in ('DESIGN.md', 'README.md'):
    path = os.path.join(ROOT, doc)
    s = open(path).read()
    for k in sorted(vals, key=len, reverse=True):      # longest keys first: R3_F32_MS before R3_F32
        s = s.replace(k, vals[k])
    left = sorted(set(re.findall(r'R3_[A-Z0-9_]+', s)))
    open(path, 'w').write(s)
    print(doc, 'unfilled:', left)
