'''Same-process A/B of the split weight-gradient kernels at the FusionNet layer shapes (batch 8, 900x1600 net): conv_wgrad_tr_kernel
(producer / consumer waves, transposing LDS reads; RCF_WGRAD_TR=1, the default) against conv_wgrad_split_kernel (RCF_WGRAD_TR=0),
through the C ABI, alternating, with the result compared bitwise.
usage (GPU box): python tools/wgrad_bench.py [reps] [filter]      # both tiers: fp32 tensors on two fp16 planes, bf16 tensors'''
import os
import sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
import rcf_amd   # noqa: F401
from rcf_amd import ops

N = 8
# name, ksize, c1, c2, cout, h, w
LAYERS = [
    ('deconv0.conv   32->32 @900x1600', 3, 32, 0, 32, 900, 1600),
    ('deconv1.conv   64+32->64 @450x800', 3, 64, 32, 64, 450, 800),
    ('deconv2.conv   64+64->64 @225x400', 3, 64, 64, 64, 225, 400),
    ('blocks2        64->64 @225x400', 3, 64, 0, 64, 225, 400),
    ('blocks2_dep    32->32 @225x400', 3, 32, 0, 32, 225, 400),
    ('blocks3        128->128 @113x200', 3, 128, 0, 128, 113, 200),
    ('deconv3.conv   128+128->128 @113x200', 3, 128, 128, 128, 113, 200),
    ('blocks4        256->256 @57x100', 3, 256, 0, 256, 57, 100),
    ('deconv4.conv   256+256->256 @57x100', 3, 256, 256, 256, 57, 100),
    ('blocks5        256->256 @29x50', 3, 256, 0, 256, 29, 50),
    ('blocks6        256->256 @15x25', 3, 256, 0, 256, 15, 25),
    ('2x2 phase      64->64 @225x400', 2, 64, 0, 64, 225, 400),
]
reps = int(sys.argv[1]) if len(sys.argv) > 1 else 10
flt = sys.argv[2] if len(sys.argv) > 2 else ''


def timeit(fn):
    fn()
    torch.cuda.synchronize()
    e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    e0.record()
    for _ in range(reps):
        fn()
    e1.record()
    torch.cuda.synchronize()
    return e0.elapsed_time(e1) * 1000.0 / reps


for prec in ('f16x2', 'bf16'):
    ops.set_precision(prec)
    adt = ops.act_dtype()
    print('--- %s: us per launch (wgrad kernel + its reduction), TF/s on the algorithmic FLOPs' % ('fp32 tensors, two fp16 planes' if prec == 'f16x2' else 'bf16 tensors'))
    print('%-40s %9s | %9s %7s | %9s %7s | %6s %s' % ('layer', 'GF', 'split us', 'TF/s', 'tr us', 'TF/s', 'ratio', 'bitwise'))
    tot = [0.0, 0.0]
    for name, k, c1, c2, co, h, w in LAYERS:
        if flt and flt not in name:
            continue
        d = ops.make_fwd_desc(N, h, w, c1, c2, co, k, 1)
        x1 = torch.randn(N, h, w, c1, device='cuda').to(adt)
        x2 = torch.randn(N, h, w, c2, device='cuda').to(adt) if c2 else None
        dz = torch.randn(N, d.h_out, d.w_out, co, device='cuda').to(adt)
        scales = ops.make_scales(ops.amax(x1), ops.amax(x2) if c2 else None, None, ops.amax(dz)) if prec == 'f16x2' else None
        gf = ops.algorithmic_flops(d) / 1e9
        res, t = {}, {}
        for rnd in range(2):       # alternate: old, new, old, new
            for tr in ('0', '1'):
                os.environ['RCF_WGRAD_TR'] = tr
                info = ops.conv_query(d)
                wsb = torch.empty(max(1, info.wgrad_workspace_floats), device='cuda')
                dw = torch.empty(co, c1 + c2, k, k, device='cuda')
                us = timeit(lambda: ops.conv_wgrad(d, x1, x2, dz, dw, wsb, scales=scales))
                t[tr] = min(t.get(tr, 1e30), us)
                res[tr] = (dw.clone(), info.wgrad_kernel_id)
        same = bool(torch.equal(res['0'][0], res['1'][0]))
        tot[0] += t['0']
        tot[1] += t['1']
        print('%-40s %9.1f | %9.1f %7.1f | %9.1f %7.1f | %6.3f %s  ids %d/%d'
              % (name, gf, t['0'], gf / t['0'] * 1e3, t['1'], gf / t['1'] * 1e3, t['1'] / t['0'], same, res['0'][1], res['1'][1]))
    print('%-40s %9s | %9.1f %7s | %9.1f %7s | %6.3f' % ('sum', '', tot[0], '', tot[1], '', tot[1] / max(tot[0], 1e-9)))
os.environ.pop('RCF_WGRAD_TR', None)

# fixed cost per launch (everything that does not scale with the pixels: launch, per-workgroup partial write, reduction kernel):
# T(n) = fixed + n * per_image  =>  fixed = 2 T(n) - T(2 n)
if os.environ.get('RCF_WGRAD_BENCH_SWEEP', '1') != '0':
    for prec in ('f16x2', 'bf16'):
        ops.set_precision(prec)
        adt = ops.act_dtype()
        for name, k, c1, co, h, w in (('64->64 @225x400', 3, 64, 64, 225, 400), ('256->256 @57x100', 3, 256, 256, 57, 100)):
            ts = {}
            for n in (4, 8, 16):
                d = ops.make_fwd_desc(n, h, w, c1, 0, co, k, 1)
                x1 = torch.randn(n, h, w, c1, device='cuda').to(adt)
                dz = torch.randn(n, h, w, co, device='cuda').to(adt)
                info = ops.conv_query(d)
                wsb = torch.empty(max(1, info.wgrad_workspace_floats), device='cuda')
                dw = torch.empty(co, c1, k, k, device='cuda')
                ts[n] = min(timeit(lambda: ops.conv_wgrad(d, x1, None, dz, dw, wsb)) for _ in range(3))
            print('%s %s: n=4 %.1f us, n=8 %.1f us, n=16 %.1f us -> fixed cost per launch %.1f us (from 4, 8) / %.1f us (from 8, 16)'
                  % (prec, name, ts[4], ts[8], ts[16], 2 * ts[4] - ts[8], 2 * ts[8] - ts[16]))
