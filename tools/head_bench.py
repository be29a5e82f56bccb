'''Micro-benchmark (GPU box): the 3x3 C->1 output head forward at the bench shapes, fp32 and bf16 tensors.  RCF_HEAD_MFMA=0: the tile kernel.'''
import sys, os
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
import rcf_amd
from rcf_amd import ops

def timeit(fn, reps=10):
    fn(); torch.cuda.synchronize()
    e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    e0.record()
    for _ in range(reps): fn()
    e1.record(); torch.cuda.synchronize()
    return e0.elapsed_time(e1) / reps

c, h, w = 32, 900, 1600
wt = torch.randn(1, c, 3, 3, device='cuda') * 0.1
coef = torch.stack([torch.rand(c) + 0.5, torch.randn(c) * 0.1, torch.zeros(c), torch.ones(c)]).cuda()
for n, dt in ((8, torch.float32), (8, torch.bfloat16), (32, torch.bfloat16)):
    x = torch.randn(n, h, w, c, device='cuda').to(dt)
    logit = torch.empty(n, h, w, device='cuda'); depth = torch.empty_like(logit)
    for cf in (None, coef):
        ms = timeit(lambda: ops.head_fwd(x, wt, logit, depth, 1.0, 100.0, coef=cf))
        gb = x.numel() * x.element_size() / 1e9
        print('head_fwd batch %d %s %s: %.3f ms, %.2f TB/s of input bytes' % (n, str(dt).split('.')[1], 'BN on load' if cf is not None else 'plain', ms, gb / ms))
    del x
