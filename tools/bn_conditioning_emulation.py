'''
CPU emulation of the HIP BatchNorm coefficient forms on the hostile 'tiny_inputs' case (tests/test_hip_f16x2.py) -- which fp32
roundings of the coefficient rows cost the 3.6 % gradient error, and which form removes it.  Diagnostic: imports the oracle.

    python tools/bn_conditioning_emulation.py
'''
import os
import sys

import numpy as np
import torch

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import rcf_amd  # noqa: F401,E402
from rcf_amd import synth  # noqa: E402
from oracle import fusionnet_oracle as fo  # noqa: E402
from oracle.fusionnet_oracle import FusionNetOracle  # noqa: E402

MODE = 'stock'


class EmuBN(torch.autograd.Function):
    '''forward / backward of bn_finalize + bn_act_fwd + bn_act_bwd_* with the coefficient rows in fp32 (mode 'rows4'), with
    y = (z - mean) * scale + beta and mean as an unevaluated fp32 pair (mode 'pair')'''

    @staticmethod
    def forward(ctx, z, gamma, beta, mode):
        zd = z.double()
        cnt = z.numel() / z.shape[1]
        s1 = zd.sum((0, 2, 3))
        s2 = (zd * zd).sum((0, 2, 3))
        mean = s1 / cnt
        var = (s2 / cnt - mean * mean).clamp_min(0)
        invstd = 1.0 / torch.sqrt(var + fo.BN_EPS)
        scale = gamma.double() * invstd
        v = lambda t: t.view(1, -1, 1, 1)
        if mode == 'rows4':
            shift = (beta.double() - mean * scale).float()
            y = (zd * v(scale.float().double()) + v(shift.double())).float()     # one fma
            mean_f = mean.float().double()
            lo = torch.zeros_like(mean)
        else:
            mean_f = mean.float().double()
            lo = (mean - mean_f).float().double()
            t = (zd - v(mean_f)).float().double()          # exact-ish (Sterbenz when z ~ mean)
            t = (t - v(lo)).float().double()
            y = (t * v(scale.float().double()) + v(beta.double())).float()
        ctx.save_for_backward(z, scale.float(), mean_f, lo, invstd.float())
        ctx.cnt = cnt
        return y

    @staticmethod
    def backward(ctx, g):
        z, scale, mean_f, lo, invstd = ctx.saved_tensors
        v = lambda t: t.view(1, -1, 1, 1)
        t = (z.double() - v(mean_f)).float().double()
        t = (t - v(lo)).float().double()
        xh = (t * v(invstd.double())).float()
        s1 = g.double().sum((0, 2, 3))
        s2 = (g.double() * xh.double()).sum((0, 2, 3))
        b0 = (s1 / ctx.cnt).float()
        b1 = (s2 / ctx.cnt).float()
        dz = v(scale) * (g - v(b0) - xh * v(b1))
        return dz, s2.float(), s1.float(), None


def bn_forward(self, x):
    if MODE == 'stock' or x.dtype != torch.float32:
        return torch.nn.functional.batch_norm(x, None, None, self.weight, self.bias, True, 0.0, self.eps)
    return EmuBN.apply(x, self.weight, self.bias, MODE)


def rel(a, b):
    return float((a.double() - b.double()).abs().max() / (b.double().abs().max() + 1e-30))


def run(dtype, cb):
    o = FusionNetOracle(**synth.TINY)
    synth.fill_state_dict_([o.encoder, o.decoder], 17)
    for mod in (o.encoder, o.decoder):
        mod.to(dtype)
    o.train()
    r = o.forward(cb['image'].to(dtype), cb['input_depth'].to(dtype))
    l = o.compute_loss(r, cb['ground_truth'].to(dtype), cb['lidar_map'].to(dtype), 2.0)[0]
    l.backward()
    named = []
    for pre, mod in (('encoder.', o.encoder), ('decoder.', o.decoder)):
        named += [(pre + k, p) for k, p in mod.named_parameters()]
    return r.detach(), float(l), {k: p.grad.double() for k, p in named if p.grad is not None}


def main():
    global MODE
    torch.nn.BatchNorm2d.forward = bn_forward
    for kind in ('tiny_inputs', 'plain'):
        cb = synth.make_batch(2, 70, 102, 8, seed=321)
        if kind == 'tiny_inputs':
            cb['image'] *= 1e-6
            cb['input_depth'] *= 1e-6
        MODE = 'stock'
        _, _, g64 = run(torch.float64, cb)
        for mode in ('stock', 'rows4', 'pair'):
            MODE = mode
            out, loss, g = run(torch.float32, cb)
            errs = sorted(((rel(g[k], g64[k]), k) for k in g64), reverse=True)
            print('%-12s %-6s worst gradient tensors vs fp64: %s' % (kind, mode, ['%.2e %s' % e for e in errs[:3]]))


if __name__ == '__main__':
    main()
