'''Poll rocm-smi (sclk / power) while one kernel family runs in a loop.  usage: python tools/clock_probe.py [fwd|wgrad|bn] [seconds]'''
import sys, os, time, subprocess, threading
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
import rcf_amd
from rcf_amd import ops
what = sys.argv[1] if len(sys.argv) > 1 else 'fwd'
secs = float(sys.argv[2]) if len(sys.argv) > 2 else 4.0
dev = 'cuda'
ops.set_precision(os.environ.get('RCF_BENCH_PREC', 'fp32'))   # f16x2: the fp32 configuration's default arithmetic (unscaled call: scale one)
n, h, w, c = [int(v) for v in os.environ.get('RCF_PROBE_SHAPE', '8,225,400,64').split(',')]
d = ops.make_fwd_desc(n, h, w, c, 0, c, 3, 1, h, w, 0)
info = ops.conv_query(d)
SC = float(os.environ.get('RCF_BENCH_DATA_SCALE', '1'))   # 0: all-zero operands
ADT = ops.act_dtype()
x = (torch.randn(n, h, w, c, device=dev) * SC).to(ADT); wt = torch.randn(c, c, 3, 3, device=dev) * 0.05 * SC
packed = torch.empty(info.packed_weight_floats, device=dev); ops.conv_pack(d, wt, packed)
out = torch.empty(n, h, w, c, device=dev, dtype=ADT); dz = (torch.randn(n, h, w, c, device=dev) * SC).to(ADT); dw = torch.empty_like(wt)
ws = torch.empty(max(1, info.wgrad_workspace_floats), device=dev)
coef = torch.randn(4, c, device=dev)
def run():
    if what == 'fwd': ops.conv_fwd(d, x, None, packed, out, None)
    elif what == 'wgrad': ops.conv_wgrad(d, x, None, dz, dw, ws)
    else: ops.bn_act_fwd(x, coef, None, out, n * h * w, c, 1)
samples = []
stop = False
def poll():
    while not stop:
        try:
            o = subprocess.run(['rocm-smi', '--showclocks', '--showpower'], capture_output=True, text=True, timeout=5).stdout
            s = [l.strip() for l in o.splitlines() if 'sclk' in l or 'Power' in l or 'power' in l]
            samples.append(' | '.join(s))
        except Exception as e:
            samples.append('err %s' % e)
        time.sleep(0.3)
t = threading.Thread(target=poll); t.start()
run(); torch.cuda.synchronize()
t0 = time.time(); it = 0
while time.time() - t0 < secs:
    for _ in range(50): run()
    torch.cuda.synchronize(); it += 50
dt = time.time() - t0
stop = True; t.join()
print(what, '%.3f ms per launch' % (dt / it * 1e3))
import re
for s in samples[2:8]:
    m = re.findall(r'\((\d+)Mhz\)', s); w = re.findall(r'Power \(W\): ([0-9.]+)', s)
    print('   sclk %s MHz, socket power %s W' % (m[0] if m else '?', w[0] if w else '?'))
