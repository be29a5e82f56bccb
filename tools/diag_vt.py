'''Diagnostic (GPU box): run the tiny net backward twice (virtual-tall tiling on / off) and report the first layer, in
backward order, whose incoming activation gradient is not bitwise identical, with the pixel pattern of the difference.'''
import sys, os
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
import rcf_amd
from rcf_amd import synth, train

cfg, wseed, shape, dseed = synth.TINY, 11, (2, 70, 102, 8), 101
b = synth.make_batch(*shape, seed=dseed)


def run(no_vt):
    if no_vt: os.environ['RCF_NO_VT'] = '1'
    else: os.environ.pop('RCF_NO_VT', None)
    m = train.build_model(cfg, device='cuda'); synth.fill_state_dict_([m.encoder, m.decoder], wseed)
    names = {}
    for pre, mod in (('encoder.', m.encoder), ('decoder.', m.decoder)):
        for k, sub in mod.named_modules(): names[id(sub)] = pre + k
    rec, order = {}, []
    E = m._engine
    orig = E.conv_bn_act
    def cba(layer, x, x2=None, up_hw=None, res=None):
        out_ = orig(layer, x, x2=x2, up_hw=up_hw, res=res)
        nm = names[id(layer)]
        rec[nm] = {'out': out_.t.clone()}
        order.append(nm)
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
    return rec, order

ra, order = run(False)
rb, _ = run(True)
for nm in order:
    if not torch.equal(ra[nm]['out'], rb[nm]['out']):
        d = (ra[nm]['out'] - rb[nm]['out']).abs()
        print('FWD differs at', nm, 'max', float(d.max()), 'count', int((d > 0).sum()), 'of', d.numel())
for nm in reversed(order):
    if not torch.equal(ra[nm]['g'], rb[nm]['g']):
        d = (ra[nm]['g'] - rb[nm]['g']).abs()      # (n, h, w, c)
        print('BWD incoming grad differs at', nm, 'shape', tuple(d.shape), 'max', float(d.max()), 'ref max', float(rb[nm]['g'].abs().max()),
              'count', int((d > 0).sum()), 'of', d.numel())
        rows = (d.amax(dim=(2, 3)) > 0)
        print('  rows with differences per image:', [[int(i) for i in rows[k].nonzero().flatten()][:40] for k in range(d.shape[0])])
        cols = (d.amax(dim=(0, 1, 3)) > 0).nonzero().flatten()
        print('  cols:', [int(i) for i in cols][:60])
        break
