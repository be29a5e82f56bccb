'''HBM stream probe (GPU box): torch copy / add vs the BN elementwise kernels on a 450x800x64 batch-8 tensor.'''
import sys, os
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
import rcf_amd
from rcf_amd import ops
dev = 'cuda'
n, h, w, c = 8, 450, 800, 64
x = torch.randn(n, h, w, c, device=dev)
y = torch.empty_like(x)
z = torch.randn_like(x)
def t(fn, reps=20):
    fn(); torch.cuda.synchronize()
    e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    e0.record()
    for _ in range(reps): fn()
    e1.record(); torch.cuda.synchronize()
    return e0.elapsed_time(e1) / reps
gb = x.numel() * 4 / 1e9
ms = t(lambda: y.copy_(x)); print('torch copy      %.3f ms  %.2f TB/s (r+w)' % (ms, 2 * gb / ms))
ms = t(lambda: torch.add(x, z, out=y)); print('torch add       %.3f ms  %.2f TB/s (2r+w)' % (ms, 3 * gb / ms))
ms = t(lambda: x.sum()); print('torch sum       %.3f ms  %.2f TB/s (r)' % (ms, gb / ms))
ms = t(lambda: y.fill_(1.0)); print('torch fill      %.3f ms  %.2f TB/s (w)' % (ms, gb / ms))
coef = torch.randn(4, c, device=dev)
npix = n * h * w
ms = t(lambda: ops.bn_act_fwd(x.view(-1), coef, None, y.view(-1), npix, c, 1) if hasattr(ops, 'bn_act_fwd') else None)
print('bn_act_fwd      %.3f ms  %.2f TB/s (r+w)' % (ms, 2 * gb / ms))
