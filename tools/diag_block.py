'''Diagnostic (GPU box): one DecoderBlock (UpConv + concat conv) through the engine vs fp64 torch, with intermediates.'''
import sys, os
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch, torch.nn.functional as F
import rcf_amd
from rcf_amd import net_utils, ops
from rcf_amd.engine import Engine, Act

torch.manual_seed(0)
rel = lambda a, b: float((a.double().cpu() - b.double().cpu()).abs().max() / (b.double().abs().max() + 1e-30))
nhwc = lambda t: t.permute(0, 2, 3, 1).contiguous().float().cuda()
nchw = lambda t: t.detach().cpu().permute(0, 3, 1, 2).contiguous()

for (cin, cskip, cout, hs, ws, h, w, n) in ((8, 8, 8, 18, 26, 35, 51, 2), (8, 0, 4, 35, 51, 70, 102, 2), (64, 32, 64, 12, 20, 24, 40, 1)):
    blk = net_utils.DecoderBlock(cin, cskip, cout, 'kaiming_uniform', 'leaky_relu', True, 'up').cuda()
    grads = {}
    eng = Engine(None, None, 1.0, 100.0)
    eng.grad_of = lambda p: grads.setdefault(id(p), torch.empty_like(p))
    x = torch.randn(n, cin, hs, ws, dtype=torch.float64)
    skip = torch.randn(n, cskip, h, w, dtype=torch.float64) if cskip else None
    dy = torch.randn(n, cout, h, w, dtype=torch.float64)
    # engine
    eng.training = True; eng.tape = []
    xa = Act(nhwc(x)); sa = Act(nhwc(skip)) if cskip else None
    out = eng.decoder_block(blk, xa, skip=sa, shape=(h, w))
    tape = eng.tape
    out.g = nhwc(dy)
    # run backward step by step, capturing the intermediate activation gradient
    tape.pop()()          # block.conv backward -> deconv.g
    # find the Act of deconv output: it is the `x` captured by the closure we just ran; recompute via second closure's out
    clos = tape[-1]
    deconv_act = [c.cell_contents for c in clos.__closure__ if isinstance(c.cell_contents, Act) and c.cell_contents is not xa][0]
    deconv_g = deconv_act.g.clone()
    tape.pop()()
    torch.cuda.synchronize()
    # fp64 reference
    W1 = blk.deconv.conv.conv.weight.detach().cpu().double().requires_grad_(True)
    g1 = blk.deconv.conv.batch_norm.weight.detach().cpu().double().requires_grad_(True)
    b1 = blk.deconv.conv.batch_norm.bias.detach().cpu().double().requires_grad_(True)
    W2 = blk.conv.conv.weight.detach().cpu().double().requires_grad_(True)
    g2 = blk.conv.batch_norm.weight.detach().cpu().double().requires_grad_(True)
    b2 = blk.conv.batch_norm.bias.detach().cpu().double().requires_grad_(True)
    xr = x.clone().requires_grad_(True); sr = skip.clone().requires_grad_(True) if cskip else None
    up = F.interpolate(xr, size=(h, w))
    z1 = F.conv2d(up, W1, padding=1)
    a1 = F.leaky_relu(F.batch_norm(z1, None, None, g1, b1, True, 0.1, 1e-5), 0.2); a1.retain_grad()
    cat = torch.cat([a1, sr], 1) if cskip else a1
    z2 = F.conv2d(cat, W2, padding=1)
    a2 = F.leaky_relu(F.batch_norm(z2, None, None, g2, b2, True, 0.1, 1e-5), 0.2)
    (a2 * dy).sum().backward()
    print('block cin=%d cskip=%d cout=%d %dx%d->%dx%d' % (cin, cskip, cout, hs, ws, h, w))
    print('  out          %.2e' % rel(nchw(out.t), a2))
    print('  d(deconv act)%.2e   (mean of ref %.3e, mean of diff %.3e)' % (rel(nchw(deconv_g), a1.grad), float(a1.grad.mean()), float((nchw(deconv_g).double() - a1.grad).mean())))
    print('  conv.W %.2e  conv.bn.w %.2e  conv.bn.b %.2e' % (rel(grads[id(blk.conv.conv.weight)], W2.grad), rel(grads[id(blk.conv.batch_norm.weight)], g2.grad), rel(grads[id(blk.conv.batch_norm.bias)], b2.grad)))
    print('  deconv.W %.2e  deconv.bn.w %.2e  deconv.bn.b %.2e' % (rel(grads[id(blk.deconv.conv.conv.weight)], W1.grad), rel(grads[id(blk.deconv.conv.batch_norm.weight)], g1.grad), rel(grads[id(blk.deconv.conv.batch_norm.bias)], b1.grad)))
    print('  dx %.2e' % rel(nchw(xa.g), xr.grad), (' dskip %.2e' % rel(nchw(sa.g), sr.grad)) if cskip else '')
    d = (nchw(deconv_g).double() - a1.grad).abs()
    print('  |diff| of d(deconv act): max %.3e at %s; border-mean %.3e interior-mean %.3e' % (float(d.max()), tuple(int(v) for v in (d == d.max()).nonzero()[0]), float(torch.cat([d[:, :, 0].flatten(), d[:, :, -1].flatten(), d[:, :, :, 0].flatten(), d[:, :, :, -1].flatten()]).mean()), float(d[:, :, 1:-1, 1:-1].mean())))
