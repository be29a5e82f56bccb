'''Micro-benchmark (GPU box): the hot FusionNet conv layer shapes at batch 8, 900x1600, one by one through the C ABI.
usage: [RCF_BENCH_PREC=fp32|bf16|bf16_operands] python tools/conv_bench.py [reps] [filter]'''
import sys, os
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
import rcf_amd
from rcf_amd import ops

N = 8
# name, ksize, stride, c1, c2, cout, h_in, w_in, up_from
LAYERS = [
    ('deconv0.deconv 64->32 up', 3, 1, 64, 0, 32, 900, 1600, (450, 800)),
    ('deconv0.conv   32->32',    3, 1, 32, 0, 32, 900, 1600, None),
    ('deconv1.deconv 64->64 up', 3, 1, 64, 0, 64, 450, 800, (225, 400)),
    ('deconv1.conv   64+32->64', 3, 1, 64, 32, 64, 450, 800, None),
    ('deconv2.conv   64+64->64', 3, 1, 64, 64, 64, 225, 400, None),
    ('deconv3.conv  128+128->128', 3, 1, 128, 128, 128, 113, 200, None),
    ('deconv4.conv  256+256->256', 3, 1, 256, 256, 256, 57, 100, None),
    ('blocks2_img    64->64',    3, 1, 64, 0, 64, 225, 400, None),
    ('blocks3_img   128->128',   3, 1, 128, 0, 128, 113, 200, None),
    ('blocks4_img   256->256',   3, 1, 256, 0, 256, 57, 100, None),
    ('blocks5_img   256->256',   3, 1, 256, 0, 256, 29, 50, None),
    ('blocks3_img.0 s2 64->128', 3, 2, 64, 0, 128, 225, 400, None),
    ('blocks4_img.0 s2 128->256', 3, 2, 128, 0, 256, 113, 200, None),
    ('blocks3_dep.0 s2 32->64', 3, 2, 32, 0, 64, 225, 400, None),
    ('fuse2 1x1 32->64',         1, 1, 32, 0, 64, 225, 400, None),
    ('stem 7x7 3->32',           7, 2, 3, 0, 32, 900, 1600, None),
]
reps = int(sys.argv[1]) if len(sys.argv) > 1 else 5
flt = sys.argv[2] if len(sys.argv) > 2 else ''
dev = 'cuda'
ops.set_precision(os.environ.get('RCF_BENCH_PREC', 'fp32'))
ADT = ops.act_dtype()
DATA_SCALE = float(os.environ.get('RCF_BENCH_DATA_SCALE', '1'))   # 0: all-zero operands (no toggling in the matrix pipe: power probe)
print('%-30s %9s | %8s %7s | %8s %7s | %8s %7s' % ('layer', 'GF', 'fwd ms', 'TF/s', 'dgrad ms', 'TF/s', 'wgrad ms', 'TF/s'))
tot = [0.0, 0.0, 0.0, 0.0]
for name, k, s, c1, c2, co, h, w, up in LAYERS:
    if flt and flt not in name:
        continue
    hs, ws = (h, w) if up is None else up
    d = ops.make_fwd_desc(N, h, w, c1, c2, co, k, s, hs, ws, 0 if up is None else 1)
    info = ops.conv_query(d)
    x1 = (torch.randn(N, hs, ws, c1, device=dev) * DATA_SCALE).to(torch.float32 if k == 7 else ADT)
    x2 = torch.randn(N, h, w, c2, device=dev).to(ADT) if c2 else None
    wt = torch.randn(co, c1 + c2, k, k, device=dev) * 0.05 * DATA_SCALE
    packed = torch.empty(info.packed_weight_floats, device=dev)
    ops.conv_pack(d, wt, packed)
    out = torch.empty(N, d.h_out, d.w_out, co, device=dev, dtype=ADT)
    part = torch.empty(info.n_partials, 2, co, device=dev, dtype=torch.float64)
    dz = (torch.randn(N, d.h_out, d.w_out, co, device=dev) * DATA_SCALE).to(ADT)
    dw = torch.empty_like(wt)
    wsb = torch.empty(max(1, info.wgrad_workspace_floats), device=dev)
    gf = ops.algorithmic_flops(d) / 1e9

    def timeit(fn):
        fn(); torch.cuda.synchronize()
        e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
        e0.record()
        for _ in range(reps): fn()
        e1.record(); torch.cuda.synchronize()
        return e0.elapsed_time(e1) / reps

    t_f = timeit(lambda: ops.conv_fwd(d, x1, x2, packed, out, part))
    t_d = float('nan')
    if k != 7:
        dd = ops.make_dgrad_desc(d, 0, c1, False)
        di = ops.conv_query(dd)
        pd = torch.empty(di.packed_weight_floats, device=dev)
        ops.conv_pack(dd, wt, pd)
        dx = torch.empty(N, h, w, c1, device=dev, dtype=ADT)
        t_d = timeit(lambda: ops.conv_fwd(dd, dz, None, pd, dx, None))
        gf_d = ops.algorithmic_flops(dd) / 1e9
    t_w = timeit(lambda: ops.conv_wgrad(d, x1, x2, dz, dw, wsb))
    print('%-30s %9.1f | %8.3f %7.1f | %8.3f %7.1f | %8.3f %7.1f' % (name, gf, t_f, gf / t_f, t_d, (gf_d / t_d) if k != 7 else float('nan'), t_w, gf / t_w))
    del x1, x2, out, dz, wsb
