'''Accuracy of the 3x3 stride-2 forward kernels against an fp64 reference (error relative to sum |a b| per output).'''
import sys, os
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import numpy as np, torch, torch.nn.functional as F
import rcf_amd
from rcf_amd import ops
def rnd(*shape, seed=0, scale=1.0):
    g = torch.Generator().manual_seed(seed)
    return (torch.rand(*shape, generator=g) * 2 - 1) * scale
for (c1, co, n, h, w) in ((64, 128, 2, 57, 100), (128, 256, 2, 29, 50), (32, 64, 2, 113, 200), (256, 256, 2, 15, 25), (64, 128, 2, 225, 400)):
    x = rnd(n, c1, h, w, seed=1) + 0.3
    wt = rnd(co, c1, 3, 3, seed=2, scale=1.0 / np.sqrt(c1 * 9))
    ref = F.conv2d(x.double(), wt.double(), stride=2, padding=1)
    mag = F.conv2d(x.double().abs(), wt.double().abs(), stride=2, padding=1)
    d = ops.make_fwd_desc(n, h, w, c1, 0, co, 3, 2)
    info = ops.conv_query(d)
    packed = torch.empty(info.packed_weight_floats, device='cuda')
    ops.conv_pack(d, wt.cuda(), packed)
    out = torch.full((n, d.h_out, d.w_out, co), float('nan'), device='cuda')
    part = torch.empty((info.n_partials, 2, co), dtype=torch.float64, device='cuda')
    ops.conv_fwd(d, x.permute(0, 2, 3, 1).contiguous().cuda(), None, packed, out, part)
    torch.cuda.synchronize()
    got = out.cpu().permute(0, 3, 1, 2).double()
    err = ((got - ref).abs() / mag)
    s = part.sum(0).cpu()
    print('%3d->%3d %dx%dx%d kernel %d: max err/sum|ab| %.2e mean %.2e | argmax %s | stats err %.2e %.2e' % (
        c1, co, n, h, w, info.kernel_id, float(err.max()), float(err.mean()), np.unravel_index(int(err.argmax()), err.shape),
        float((s[0] - got.sum((0, 2, 3))).abs().max()), float(((s[1] - (got ** 2).sum((0, 2, 3))).abs() / (got ** 2).sum((0, 2, 3))).max())))
