'''Gradient parity against an fp64 oracle over several seeds (published net, 2x113x200): median / max per-tensor relative error of
the HIP path next to the CPU fp32 oracle's -- how much of a change in that figure is chaos and how much is the kernel.'''
import sys, os
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import numpy as np, torch
import rcf_amd
from rcf_amd import synth, train
from oracle.fusionnet_oracle import FusionNetOracle

def named(m):
    out = []
    for prefix, mod in (('encoder.', m.encoder), ('decoder.', m.decoder)):
        out += [(prefix + k, v) for k, v in mod.named_parameters()]
    return out

def rel(a, b):
    a, b = a.detach().cpu().double(), b.detach().cpu().double()
    return float((a - b).abs().max() / (b.abs().max() + 1e-30))

def oracle(wseed, cb, dtype):
    o = FusionNetOracle(**synth.PUBLISHED)
    synth.fill_state_dict_([o.encoder, o.decoder], wseed)
    for mod in (o.encoder, o.decoder): mod.to(dtype)
    o.train()
    out = o.forward(cb['image'].to(dtype), cb['input_depth'].to(dtype))
    loss = o.compute_loss(out, cb['ground_truth'].to(dtype), cb['lidar_map'].to(dtype), 2.0)[0]
    loss.backward()
    return {k: (None if p.grad is None else p.grad.detach().double()) for k, p in named(o)}

seeds = [int(s) for s in sys.argv[1:]] or [5, 6, 7]
for ws in seeds:
    cb = synth.make_batch(2, 113, 200, 16, seed=ws + 4)
    m = train.build_model(synth.PUBLISHED, device='cuda')
    synth.fill_state_dict_([m.encoder, m.decoder], ws)
    m.train()
    b = {k: v.cuda() for k, v in cb.items()}
    out = m.forward(b['image'], b['input_depth'])
    loss, _ = m.compute_loss(b['image'], out, b['ground_truth'], b['lidar_map'], 'l1', 0.0, -1, None, 2.0)
    loss.backward()
    g64, g32 = oracle(ws, cb, torch.float64), oracle(ws, cb, torch.float32)
    eh, ec = [], []
    dh = dc = nn = 0.0
    eh2 = ec2 = 0.0
    for k, p in named(m):
        if g64[k] is None: continue
        eh.append(rel(p.grad, g64[k])); ec.append(rel(g32[k], g64[k]))
        a, c, r = p.grad.detach().cpu().double().reshape(-1), g32[k].reshape(-1), g64[k].reshape(-1)
        dh += float(a @ r); dc += float(c @ r); nn += float(r @ r)
        eh2 += float(((a - r) ** 2).sum()); ec2 += float(((c - r) ** 2).sum())

    print('seed %d: HIP median %.2e max %.2e | CPU-fp32 median %.2e max %.2e | whole-gradient rel L2 err: HIP %.2e CPU-fp32 %.2e' % (ws, np.median(eh), max(eh), np.median(ec), max(ec), (eh2 / nn) ** 0.5, (ec2 / nn) ** 0.5), flush=True)
