'''Diagnostic (GPU box): per-parameter gradient error of the HIP path and of the fp32 CPU oracle, both against an
fp64 CPU oracle, so real errors can be told from fp32 chaos (sign / argmax flips).'''
import sys, os
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
import rcf_amd
from rcf_amd import synth, train
from oracle.fusionnet_oracle import FusionNetOracle


def named(m):
    out = []
    for pre, mod in (('encoder.', m.encoder), ('decoder.', m.decoder)):
        out += [(pre + k, p) for k, p in mod.named_parameters()]
    return out


def oracle_run(dtype, cfg, wseed, b):
    m = FusionNetOracle(**cfg); synth.fill_state_dict_([m.encoder, m.decoder], wseed)
    for mod in (m.encoder, m.decoder): mod.to(dtype)
    m.train()
    out = m.forward(b['image'].to(dtype), b['input_depth'].to(dtype))
    loss = m.compute_loss(out, b['ground_truth'].to(dtype), b['lidar_map'].to(dtype), 2.0)[0]
    loss.backward()
    return out.detach().double(), {k: p.grad.double() for k, p in named(m) if p.grad is not None}


def hip_run(cfg, wseed, b):
    m = train.build_model(cfg, device='cuda'); synth.fill_state_dict_([m.encoder, m.decoder], wseed)
    m.train()
    g = {k: v.cuda() for k, v in b.items()}
    out = m.forward(g['image'], g['input_depth'])
    loss, _ = m.compute_loss(g['image'], out, g['ground_truth'], g['lidar_map'], 'l1', 0.0, -1, None, 2.0)
    loss.backward()
    torch.cuda.synchronize()
    return out.detach().cpu().double(), {k: p.grad.detach().cpu().double() for k, p in named(m) if p.grad is not None}


rel = lambda a, b: float((a - b).abs().max() / (b.abs().max() + 1e-30))
for name, cfg, shape, wseed, dseed in (('tiny', synth.TINY, (2, 70, 102, 8), 11, 101), ('published', synth.PUBLISHED, (1, 224, 384, 32), 21, 301)):
    b = synth.make_batch(*shape, seed=dseed)
    o64, g64 = oracle_run(torch.float64, cfg, wseed, b)
    o32, g32 = oracle_run(torch.float32, cfg, wseed, b)
    oh, gh = hip_run(cfg, wseed, b)
    print('==== %s: out err cpu32 %.2e hip %.2e' % (name, rel(o32, o64), rel(oh, o64)))
    rows = sorted(((rel(gh[k], g64[k]), rel(g32[k], g64[k]), k) for k in g64), reverse=True)
    for eh, e32, k in rows[:25]:
        print('  hip %.2e  cpu32 %.2e  %s' % (eh, e32, k))
    print('  ... median hip %.2e cpu32 %.2e' % (sorted(r[0] for r in rows)[len(rows) // 2], sorted(r[1] for r in rows)[len(rows) // 2]))

# ---- tiny net, all parameters in backward-completion order
b = synth.make_batch(2, 70, 102, 8, seed=101)
o64, g64 = oracle_run(torch.float64, synth.TINY, 11, b)
oh, gh = hip_run(synth.TINY, 11, b)
m = train.build_model(synth.TINY, device='cpu')
names = {id(p): k for k, p in named(m)}
print('==== tiny, backward order')
for p in m._used_params:
    k = names[id(p)]
    print('  %.2e  %-55s |g|max %.3e' % (rel(gh[k], g64[k]), k, float(g64[k].abs().max())))
