'''Diagnostic (GPU box): per-layer forward activations and incoming activation gradients of the tiny net,
HIP engine vs fp64 CPU oracle.'''
import sys, os
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
import rcf_amd
from rcf_amd import synth, train, engine as eng_mod
from oracle.fusionnet_oracle import FusionNetOracle, Conv2d as OConv

import sys as _s
if len(_s.argv) > 1 and _s.argv[1] == 'pub':
    cfg, wseed, shape, dseed = synth.PUBLISHED, 5, (2, 113, 200, 16), 9
else:
    cfg, wseed, shape, dseed = synth.TINY, 11, (2, 70, 102, 8), 101
b = synth.make_batch(*shape, seed=dseed)
rel = lambda a, b_: float((a.double() - b_.double()).abs().max() / (b_.double().abs().max() + 1e-30))
nchw = lambda t: t.detach().cpu().permute(0, 3, 1, 2).contiguous()

# ---- oracle fp64 with hooks
o = FusionNetOracle(**cfg); synth.fill_state_dict_([o.encoder, o.decoder], wseed)
for mod in (o.encoder, o.decoder): mod.double()
o.train()
ref = {}
def mk(name):
    def hook(mod, inp, out):
        out.retain_grad(); ref[name] = out
    return hook
for pre, mod in (('encoder.', o.encoder), ('decoder.', o.decoder)):
    for k, sub in mod.named_modules():
        if isinstance(sub, OConv): sub.register_forward_hook(mk(pre + k))
out = o.forward(b['image'].double(), b['input_depth'].double())
loss = o.compute_loss(out, b['ground_truth'].double(), b['lidar_map'].double(), 2.0)[0]
loss.backward()

# ---- oracle fp32 with hooks
o32 = FusionNetOracle(**cfg); synth.fill_state_dict_([o32.encoder, o32.decoder], wseed)
o32.train()
ref32 = {}
def mk32(name):
    def hook(mod, inp, out):
        out.retain_grad(); ref32[name] = out
    return hook
for pre, mod in (('encoder.', o32.encoder), ('decoder.', o32.decoder)):
    for k, sub in mod.named_modules():
        if isinstance(sub, OConv): sub.register_forward_hook(mk32(pre + k))
out32 = o32.forward(b['image'], b['input_depth'])
loss32 = o32.compute_loss(out32, b['ground_truth'], b['lidar_map'], 2.0)[0]
loss32.backward()

# ---- HIP with recording
m = train.build_model(cfg, device='cuda'); synth.fill_state_dict_([m.encoder, m.decoder], wseed)
names = {}
for pre, mod in (('encoder.', m.encoder), ('decoder.', m.decoder)):
    for k, sub in mod.named_modules(): names[id(sub)] = pre + k
rec = {}
E = m._engine
orig_cba = E.conv_bn_act
def cba(layer, x, x2=None, up_hw=None, res=None):
    out_ = orig_cba(layer, x, x2=x2, up_hw=up_hw, res=res)
    nm = names[id(layer)]
    rec[nm] = {'out': out_.t.clone(), 'res': res is not None}
    if E.tape is not None:
        inner = E.tape[-1]
        def wrapped():
            rec[nm]['g'] = out_.g.clone()
            inner()
        E.tape[-1] = wrapped
    return out_
E.conv_bn_act = cba
m.train()
g = {k: v.cuda() for k, v in b.items()}
oh = m.forward(g['image'], g['input_depth'])
lh, _ = m.compute_loss(g['image'], oh, g['ground_truth'], g['lidar_map'], 'l1', 0.0, -1, None, 2.0)
lh.backward(); torch.cuda.synchronize()
print('out %.2e loss %.9f vs %.9f' % (rel(oh.cpu(), out.detach()), float(lh), float(loss)))
order = list(rec.keys())
print('%-40s %10s %10s %10s %10s  %s' % ('layer (forward order)', 'fwd hip', 'fwd cpu32', 'grad hip', 'grad cpu32', 'shape'))
for nm in order:
    r = rec[nm]
    if r['res']:
        print('%-40s %10s %10s   (residual tail: oracle module output differs by design)' % (nm, '-', '-')); continue
    fe = rel(nchw(r['out']), ref[nm].detach())
    ge = rel(nchw(r['g']), ref[nm].grad) if 'g' in r else float('nan')
    print('%-40s %10.2e %10.2e %10.2e %10.2e  %s' % (nm, fe, rel(ref32[nm].detach(), ref[nm].detach()), ge, rel(ref32[nm].grad, ref[nm].grad), tuple(ref[nm].shape)))
