'''GPU box: the 3x3 convolution on fp32 input (conversion inside the kernel: rcf_conv2d_fwd_scaled) against the same convolution on
operand planes (rcf_conv2d_fwd_planes), layer by layer at batch 8, 900x1600 -- plus what the one-off conversion pass costs.
usage: python tools/planes_bench.py [reps]'''
import os, sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
import rcf_amd
from rcf_amd import ops

N = 8
LAYERS = [('deconv0.conv   32->32', 32, 0, 32, 900, 1600), ('deconv1.conv   64+32->64', 64, 32, 64, 450, 800),
          ('deconv1.in     64->64', 64, 0, 64, 450, 800), ('blocks2_img    64->64', 64, 0, 64, 225, 400),
          ('deconv2.conv   64+64->64', 64, 64, 64, 225, 400), ('blocks3_img   128->128', 128, 0, 128, 113, 200),
          ('blocks4_img   256->256', 256, 0, 256, 57, 100)]
reps = int(sys.argv[1]) if len(sys.argv) > 1 else 10
ops.set_precision('f16x2')


def timeit(fn):
    fn(); torch.cuda.synchronize()
    e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    e0.record()
    for _ in range(reps): fn()
    e1.record(); torch.cuda.synchronize()
    return e0.elapsed_time(e1) / reps


print('%-28s %8s | %9s %7s | %9s %7s | %6s | %9s' % ('layer', 'GF', 'fp32 ms', 'TF/s', 'planes ms', 'TF/s', 'gain', 'to_planes'))
for name, c1, c2, co, h, w in LAYERS:
    d = ops.make_fwd_desc(N, h, w, c1, c2, co, 3, 1)
    info, qi = ops.conv_query(d), ops.conv_query_planes(d)
    x1 = torch.randn(N, h, w, c1, device='cuda')
    x2 = torch.randn(N, h, w, c2, device='cuda') if c2 else None
    wt = torch.randn(co, c1 + c2, 3, 3, device='cuda') * 0.05
    ax = ops.amax(x1)
    if x2 is not None:
        ops.amax(x2, ax, accumulate=True)
    aw = ops.amax(wt)
    packed = torch.empty(info.packed_weight_floats, device='cuda')
    ops.conv_pack(d, wt, packed, aw)
    out = torch.empty(N, h, w, co, device='cuda')
    part = torch.empty(max(info.n_partials, qi.n_partials), 2, co, device='cuda', dtype=torch.float64)
    sc = ops.make_scales(ax, ax if x2 is not None else None, aw)
    p1 = ops.to_planes(x1, ax)
    p2 = None if x2 is None else ops.to_planes(x2, ax)
    gf = ops.algorithmic_flops(d) / 1e9
    t0 = timeit(lambda: ops.conv_fwd(d, x1, x2, packed, out, part, scales=sc))
    ref = out.clone()
    t1 = timeit(lambda: ops.conv_fwd_planes(d, p1, p2, packed, out, part, sc))
    same = bool(torch.equal(out, ref))
    tp = timeit(lambda: ops.to_planes(x1, ax, p1))
    print('%-28s %8.1f | %9.3f %7.1f | %9.3f %7.1f | %5.1f%% | %9.3f  %s' % (name, gf, t0, gf / t0, t1, gf / t1, 100 * (t0 - t1) / t0, tp,
                                                                          '' if same else 'MISMATCH'))
