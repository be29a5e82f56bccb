'''Micro-benchmark (GPU box): the BatchNorm + LeakyReLU elementwise passes on FusionNet-sized tensors, as HBM TB/s (algorithmic bytes: tensors read + written).
usage: [RCF_BENCH_PREC=fp32|bf16] python tools/ew_bench.py'''
import sys, os
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
import rcf_amd
from rcf_amd import ops
from rcf_amd._lib import RCF_ACT_LEAKY_RELU

ops.set_precision(os.environ.get('RCF_BENCH_PREC', 'fp32'))
ADT = ops.act_dtype()
B = 2 if ADT == torch.bfloat16 else 4
SHAPES = [(8, 900, 1600, 32), (8, 450, 800, 64), (8, 225, 400, 64), (8, 113, 200, 128), (8, 57, 100, 256), (8, 29, 50, 256)]
print('%-22s %10s | %18s | %18s | %18s' % ('tensor', 'MB', 'bn_act_fwd', 'bn_act_bwd_reduce', 'bn_act_bwd_apply'))


def timeit(fn, reps=10):
    fn(); torch.cuda.synchronize()
    e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    e0.record()
    for _ in range(reps): fn()
    e1.record(); torch.cuda.synchronize()
    return e0.elapsed_time(e1) / reps


for n, h, w, c in SHAPES:
    npix = n * h * w
    z = torch.randn(n, h, w, c, device='cuda').to(ADT)
    dout = torch.randn(n, h, w, c, device='cuda').to(ADT)
    y = torch.empty_like(z)
    dz = torch.empty_like(z)
    coef = torch.rand(2, c, device='cuda') + 0.5
    bcoef = torch.rand(2, c, device='cuda') * 1e-3
    nb = ops.ew_blocks(npix, c)
    part = torch.empty(nb, 2, c, device='cuda', dtype=torch.float64)
    mb = npix * c * B / 1e6
    t_f = timeit(lambda: ops.bn_act_fwd(z, coef, None, y, npix, c, RCF_ACT_LEAKY_RELU))
    t_r = timeit(lambda: ops.bn_act_bwd_reduce(dout, z, coef, None, part, npix, c, RCF_ACT_LEAKY_RELU, False))
    t_a = timeit(lambda: ops.bn_act_bwd_apply(dout, z, coef, None, bcoef, dz, None, False, npix, c, RCF_ACT_LEAKY_RELU, False))
    f = lambda t, k: '%7.3f ms %5.2f TB/s' % (t, k * mb / t / 1e3)
    print('%-22s %10.1f | %18s | %18s | %18s' % ('%dx%dx%dx%d' % (n, h, w, c), mb, f(t_f, 2), f(t_r, 2), f(t_a, 3)))
