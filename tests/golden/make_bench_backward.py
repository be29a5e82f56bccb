'''
Pins the BACKWARD and the first three optimizer steps of bench.py's headline step (BASELINE.json configs[1]: published FusionNet,
train-mode BatchNorm, batch 8, 900x1600, 64 radar points, weights seed 1234, data seed 1234, outlier removal (7, 1.5), masked L1 +
2.0 x lidar term, Adam lr 1e-3) to the CPU oracle -- the part of the metric's workload that bench_expected.json (forward / loss only)
does not hold.  The oracle is pinned to the real reference by make_golden.py (0.00e+00 on every gradient).

    python tests/golden/make_bench_backward.py            # fp32 oracle: 3 training steps at batch 8 (~45 GB, ~40 min on 8 cores)
    python tests/golden/make_bench_backward.py --fp64     # the fp64 yardstick of the step-1 gradients (T1b's rule); block-wise
                                                          # activation checkpointing keeps the fp64 run inside the container's memory

Writes tests/golden/bench_backward_b8.npz:
  meta                      n, h, w, points, data seed, weights seed
  losses                    loss of steps 1..3 (the same batch each step, like bench.py)
  grad_keys/grad_l2/grad_sum   every parameter gradient of step 1: L2 norm and sum (fp64 accumulation of the fp32 values)
  keys/idx/ref32            2048 seeded elements of each of the ten largest gradient tensors (the recipe of fixture T1b)
  fp64/fp64_absmax/ref32_rel_err   (--fp64) the same elements from the fp64 run, each tensor's max|g|, and the fp32 oracle's own
                            max-norm distance from fp64 on the sampled elements -- the yardstick a test holds the HIP gradients to
  param_abs_sum3            sum|p| over all parameters after the third Adam step
'''
import os
import sys
import time

ROOT = os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
sys.path.insert(0, ROOT)

import numpy as np   # noqa: E402
import torch   # noqa: E402

import rcf_amd   # noqa: E402,F401
from rcf_amd import synth   # noqa: E402
from oracle.fusionnet_oracle import FusionNetOracle, remove_outliers   # noqa: E402
from oracle import fusionnet_oracle as fo   # noqa: E402

PATH = os.path.join(ROOT, 'tests', 'golden', 'bench_backward_b8.npz')
CASE = (8, 900, 1600, 64, 1234, 1234)
if '--small' in sys.argv:   # plumbing check of this script
    CASE = (2, 70, 102, 8, 1234, 1234)
    PATH = '/tmp/bench_backward_small.npz'


def named(model):
    out = []
    for prefix, mod in (('encoder.', model.encoder), ('decoder.', model.decoder)):
        for k, p in mod.named_parameters():
            out.append((prefix + k, p))
    return out


def build(dtype):
    n, h, w, k, dseed, wseed = CASE
    model = FusionNetOracle(**synth.PUBLISHED)
    synth.fill_state_dict_([model.encoder, model.decoder], wseed)
    b = synth.make_batch(n, h, w, k, seed=dseed)
    if dtype == torch.float64:
        for mod in (model.encoder, model.decoder):
            mod.double()
        b = {kk: (v.double() if v.is_floating_point() else v) for kk, v in b.items()}
    model.train()
    gt = remove_outliers(b['ground_truth'], 7, 1.5)
    return model, b, gt


def checkpoint_blocks(model):
    '''fp64 only: recompute each ResNet / decoder block in backward instead of keeping its activations (the gradients are the same
    function of the same inputs; train-mode BatchNorm's running buffers get a second update, which this script does not record)'''
    from torch.utils.checkpoint import checkpoint
    for mod in list(model.encoder.modules()) + list(model.decoder.modules()):
        if isinstance(mod, (fo.ResNetBlock, fo.DecoderBlock, fo.Conv2d, fo.UpConv2d)):   # nested: a block's units are recomputed one at a time
            inner = mod.forward
            mod.forward = (lambda f: (lambda *a, **kw: checkpoint(f, *a, use_reentrant=False, **kw)))(inner)


def conv_per_sample():
    '''fp64 only: stock PyTorch has no fast fp64 convolution on the CPU; its fallback builds an im2col buffer of C_in x 9 x H x W
    elements PER SAMPLE, one per thread in parallel over the batch -- 8 x 6.6 GB for the 64-channel 900x1600 layer.  One sample per call
    keeps one buffer alive (the GEMM inside still uses every thread); the results are the same convolutions.'''
    orig = torch.nn.Conv2d.forward

    def forward(self, x):
        if x.dtype == torch.float64 and x.shape[0] > 1:
            return torch.cat([orig(self, x[i:i + 1]) for i in range(x.shape[0])], 0)
        return orig(self, x)
    torch.nn.Conv2d.forward = forward


def main():
    torch.manual_seed(0)
    torch.set_num_threads(8)
    t0 = time.time()
    if '--fp64' in sys.argv:
        rec = dict(np.load(PATH))
        model, b, gt = build(torch.float64)
        checkpoint_blocks(model)
        conv_per_sample()
        out = model.forward(b['image'], b['input_depth'])
        loss = model.compute_loss(out, gt, b['lidar_map'], 2.0)[0]
        loss.backward()
        print('fp64 loss %.9f  (fp32 %.9f)  %.0f s' % (float(loss.detach()), float(rec['losses'][0]), time.time() - t0), flush=True)
        g = dict(named(model))
        v64 = np.stack([g[k].grad.reshape(-1).numpy()[i] for k, i in zip(rec['keys'].tolist(), rec['idx'])]).astype(np.float64)
        amax = np.array([float(g[k].grad.abs().max()) for k in rec['keys'].tolist()])
        rec['fp64'] = v64
        rec['fp64_absmax'] = amax
        rec['fp64_loss'] = np.array(float(loss.detach()))
        rec['ref32_rel_err'] = np.array([float(np.abs(r.astype(np.float64) - v).max() / a) for r, v, a in zip(rec['ref32'], v64, amax)])
        # every tensor's norm in fp64 too: the yardstick for the norms
        rec['grad_l2_fp64'] = np.array([float(g[k].grad.norm()) for k in rec['grad_keys'].tolist()])
        print('fp32 oracle vs fp64, sampled elements (max-norm): ' + ' '.join('%.1e' % e for e in rec['ref32_rel_err']), flush=True)
        np.savez_compressed(PATH, **rec)
        return
    model, b, gt = build(torch.float32)
    opt = torch.optim.Adam([{'params': model.parameters(), 'weight_decay': 0.0}], lr=1e-3)
    losses = []
    rec = {'meta': np.array(CASE)}
    for step in range(3):
        out = model.forward(b['image'], b['input_depth'])
        loss = model.compute_loss(out, gt, b['lidar_map'], 2.0)[0]
        opt.zero_grad()
        loss.backward()
        losses.append(float(loss.detach()))
        print('step %d loss %.9f  %.0f s' % (step + 1, losses[-1], time.time() - t0), flush=True)
        if step == 0:
            g = {k: p.grad.detach() for k, p in named(model) if p.grad is not None}
            keys = list(g.keys())
            rec['grad_keys'] = np.array(keys)
            rec['grad_l2'] = np.array([float(g[k].double().norm()) for k in keys])
            rec['grad_sum'] = np.array([float(g[k].double().sum()) for k in keys])
            big = sorted(keys, key=lambda k: -g[k].numel())[:10]
            rs = np.random.RandomState(7)
            idx = np.stack([rs.choice(g[k].numel(), 2048, replace=False) for k in big]).astype(np.int64)
            rec['keys'] = np.array(big)
            rec['idx'] = idx
            rec['ref32'] = np.stack([g[k].reshape(-1).numpy()[i] for k, i in zip(big, idx)]).astype(np.float32)
            rec['output_mean'] = np.array(float(out.detach().double().mean()))
            rec['losses'] = np.array(losses, np.float64)
            np.savez_compressed(PATH, **rec)   # the step-1 record is safe even if the run is cut short
            del g
        opt.step()
    rec['losses'] = np.array(losses, np.float64)
    rec['param_abs_sum3'] = np.array(float(sum(p.detach().double().abs().sum() for _, p in named(model))))
    np.savez_compressed(PATH, **rec)
    print('losses', losses, 'param_abs_sum3', float(rec['param_abs_sum3']), flush=True)


if __name__ == '__main__':
    main()
