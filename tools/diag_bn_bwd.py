'''Diagnostic (GPU box): every BatchNorm-backward call of one training step of the tiny net (hostile 'tiny_inputs' case by default),
each STAGE checked in fp64 from that stage's own inputs: the sums (reduce + finalize -> bcoef), the apply pass (dz), and what an
fp64 BatchNorm backward of the same (dout, z) would have given (conditioning of the fp32 coefficient rows).

    RCF_DIAG_CASE=tiny_inputs python tools/diag_bn_bwd.py
'''
import os
import sys

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import numpy as np
import torch
import rcf_amd  # noqa: F401
from rcf_amd import synth, train, ops, engine as eng_mod

cfg, wseed, shape, dseed = synth.TINY, 11, (2, 70, 102, 8), 101
if os.environ.get('RCF_DIAG_CASE', 'tiny_inputs') == 'tiny_inputs':
    wseed, dseed = 17, 321
b = synth.make_batch(*shape, seed=dseed)
if os.environ.get('RCF_DIAG_CASE', 'tiny_inputs') == 'tiny_inputs':
    b['image'] *= 1e-6
    b['input_depth'] *= 1e-6
rel = lambda a, r: float((a.double() - r.double()).abs().max() / (r.double().abs().max() + 1e-300))
SLOPE = 0.2

calls = []
orig_apply = eng_mod.ops.bn_act_bwd_apply


def apply_wrapped(dout, z, coef, out, bcoef, dz, dres, dres_accumulate, n_pix, c, act, has_res, amax=None):
    orig_apply(dout, z, coef, out, bcoef, dz, dres, dres_accumulate, n_pix, c, act, has_res, amax=amax)
    torch.cuda.synchronize()
    g = dout.double().reshape(-1, c)
    zz = z.double().reshape(-1, c)
    k = coef.double()
    if has_res:
        g = g * torch.where(out.double().reshape(-1, c) > 0, 1.0, SLOPE)
    y32 = (z.reshape(-1, c) * coef[0] + coef[1])                                   # the kernel's own fp32 sign test
    gp = g * torch.where(y32.double() > 0, 1.0, SLOPE)
    xh_k = (zz - k[2]) * k[3]                                                       # x-hat from the fp32 coefficient rows
    b0 = gp.mean(0)
    b1 = (gp * xh_k).mean(0)
    e_sums = max(rel(bcoef[0], b0), rel(bcoef[1], b1))
    dz_k = k[0] * (gp - bcoef[0].double() - xh_k * bcoef[1].double())               # apply pass from ITS inputs
    e_apply = rel(dz.reshape(-1, c), dz_k)
    # fp64 BatchNorm backward of the same (dout, z): mean / invstd / x-hat in fp64, gamma from scale / invstd
    mean = zz.mean(0)
    var = (zz * zz).mean(0) - mean * mean
    invstd = 1.0 / torch.sqrt(var.clamp_min(0) + 1e-5)
    gamma = k[0] / k[3]
    xh = (zz - mean) * invstd
    dz_64 = gamma * invstd * (gp - gp.mean(0) - xh * (gp * xh).mean(0))
    e_ideal = rel(dz.reshape(-1, c), dz_64)
    e_b1 = rel(bcoef[1], (gp * xh).mean(0))
    calls.append((tuple(z.shape), has_res, e_sums, e_apply, e_ideal, e_b1, float((mean.abs() * invstd).max()),
                  rel(coef[2], mean), rel(coef[3], invstd)))


eng_mod.ops.bn_act_bwd_apply = apply_wrapped

up_calls = []
orig_up = eng_mod.Engine._conv_up2x_backward


def up_wrapped(self, layer, info, x, dz, dz_amax=None):
    had = x.g is not None
    orig_up(self, layer, info, x, dz, dz_amax)
    torch.cuda.synchronize()
    if had or x.g is None:
        return
    import torch.nn.functional as F
    w = layer.conv.weight.detach().double().cpu()
    d = dz.double().cpu().permute(0, 3, 1, 2)
    full = F.conv_transpose2d(d, w, stride=1, padding=1)
    n, c, h2, w2 = full.shape
    ref = full.reshape(n, c, h2 // 2, 2, w2 // 2, 2).sum((3, 5))
    got = x.g.double().cpu().permute(0, 3, 1, 2)
    err = (got - ref).abs()
    idx = int(err.argmax())
    up_calls.append((tuple(got.shape), float(err.max() / ref.abs().max()), tuple(int(v) for v in np.unravel_index(idx, got.shape)),
                     float(d.abs().max()), float(d.pow(2).mean().sqrt()), float(ref.abs().max()), float(ref.pow(2).mean().sqrt()),
                     'ref there %.4e got %.4e' % (float(ref.reshape(-1)[idx]), float(got.reshape(-1)[idx]))))


eng_mod.Engine._conv_up2x_backward = up_wrapped

m = train.build_model(cfg, device='cuda')
synth.fill_state_dict_([m.encoder, m.decoder], wseed)
m.train()
g = {k: v.cuda() for k, v in b.items()}
oh = m.forward(g['image'], g['input_depth'])
lh, _ = m.compute_loss(g['image'], oh, g['ground_truth'], g['lidar_map'], 'l1', 0.0, -1, None, 2.0)
lh.backward()
torch.cuda.synchronize()
print('%-22s %4s %10s %10s %10s %10s %10s %10s %10s' % ('z shape (backward order)', 'res', 'sums', 'apply', 'vs fp64 BN', 'b1 vs 64',
                                                         '|mean|*istd', 'mean rel', 'invstd rel'))
for c_ in calls:
    print('%-22s %4d %10.2e %10.2e %10.2e %10.2e %10.2e %10.2e %10.2e' % c_)

# ---- the up-2x layers' input gradient (four summed 2x2 phases) against an fp64 transposed convolution + 2x2 block sum of ITS inputs
print('up-2x input gradients: shape of x, rel err (max-norm), at index, |dz|max, |dz|rms, |dx|max, |dx|rms')
for rec_ in up_calls:
    print(rec_)
