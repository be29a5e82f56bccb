'''GPU box: what the fused BatchNorm statistics cost the forward convolution kernels (fp64 per value in the epilogue): the same launch
with and without the partials buffer, fp32 tensors on two fp16 planes and bf16 tensors.'''
import os, sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__)))))
import torch
import rcf_amd  # noqa: F401
from rcf_amd import ops

def timeit(fn, reps=20):
    fn(); torch.cuda.synchronize()
    e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    e0.record()
    for _ in range(reps): fn()
    e1.record(); torch.cuda.synchronize()
    return e0.elapsed_time(e1) * 1000.0 / reps

for prec in ('f16x2', 'bf16'):
    ops.set_precision(prec)
    adt = ops.act_dtype()
    for name, c1, co, h, w in (('64->64 @225x400', 64, 64, 225, 400), ('128->128 @113x200', 128, 128, 113, 200), ('256->256 @57x100', 256, 256, 57, 100),
                               ('32->32 @900x1600', 32, 32, 900, 1600), ('64->64 @450x800', 64, 64, 450, 800)):
        d = ops.make_fwd_desc(8, h, w, c1, 0, co, 3, 1)
        info = ops.conv_query(d)
        x = torch.randn(8, h, w, c1, device='cuda').to(adt)
        wt = torch.randn(co, c1, 3, 3, device='cuda') * 0.05
        packed = torch.empty(info.packed_weight_floats, device='cuda')
        ops.conv_pack(d, wt, packed)
        z = torch.empty(8, h, w, co, device='cuda', dtype=adt)
        part = torch.empty(info.n_partials, 2, co, device='cuda', dtype=torch.float64)
        r = {}
        for k in ('stats', 'plain', 'stats', 'plain'):
            r[k] = min(r.get(k, 1e9), timeit(lambda: ops.conv_fwd(d, x, None, packed, z, part if k == 'stats' else None)))
        print('%s %s: %.1f us with the statistics, %.1f us without (%.1f %%)' % (prec, name, r['stats'], r['plain'], 100.0 * (r['stats'] / r['plain'] - 1.0)))
