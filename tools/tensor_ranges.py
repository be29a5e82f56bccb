'''Dynamic range of every tensor the convolution kernels consume during one FusionNet training step (published net): would a
per-tensor power-of-two scale that puts max|x| at 2^13 keep it inside fp16's normal range (2^-14 ... 2^15)?  For each conv operand
(activations, incoming gradients, weights) prints max|x| and the fraction of the nonzero elements below max * 2^-27 (they would
land in fp16's subnormals and lose relative precision) and below max * 2^-37 (lost entirely, i.e. as absolute error <= max * 2^-38).
GPU box.  usage: python tools/tensor_ranges.py [batch height width]'''
import sys, os
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
import rcf_amd
from rcf_amd import ops, synth, train
from rcf_amd.net_utils import OutlierRemoval

B, H, W = (int(v) for v in sys.argv[1:4]) if len(sys.argv) > 3 else (2, 448, 768)
dev = torch.device('cuda')
m = train.build_model(synth.PUBLISHED, device=dev)
synth.fill_state_dict_([m.encoder, m.decoder], 1234)
m.batch_weight_packing = False
opt = train.make_optimizer(m, lr=1e-3)
m.train()
b = synth.make_batch(B, H, W, 64, seed=1)
args = [b[k].to(dev) for k in ('image', 'input_depth', 'ground_truth', 'lidar_map')]
o = OutlierRemoval(kernel_size=7, threshold=1.5)
for _ in range(3):   # a few steps in, so BatchNorm statistics and Adam have moved off the initial state
    train.train_step(m, opt, *args, outlier_removal=o)

rows = []


def stat(kind, t):
    if t is None:
        return
    a = t.detach().float().abs().flatten()
    nz = a[a > 0]
    if nz.numel() == 0:
        rows.append((kind, tuple(t.shape), 0.0, 0.0, 0.0, 0.0))
        return
    mx = float(nz.max())
    rows.append((kind, tuple(t.shape), mx, float(nz.min()), float((nz < mx * 2.0 ** -27).float().mean()), float((nz < mx * 2.0 ** -37).float().mean())))


f0, w0, p0 = ops.conv_fwd, ops.conv_wgrad, ops.conv_pack


def fwd(desc, in1, in2, packed, out, stat_partials=None, coef1=None, coef2=None):
    stat('dgrad dZ' if desc.w_mode != 0 else 'fwd x1', in1)
    stat('fwd x2', in2)
    return f0(desc, in1, in2, packed, out, stat_partials, coef1=coef1, coef2=coef2)


def wgrad(desc, in1, in2, dz, dw, workspace, coef1=None, coef2=None):
    stat('wgrad dZ', dz)
    return w0(desc, in1, in2, dz, dw, workspace, coef1=coef1, coef2=coef2)


def pack(desc, w, packed):
    stat('weight', w)
    return p0(desc, w, packed)


ops.conv_fwd, ops.conv_wgrad, ops.conv_pack = fwd, wgrad, pack
train.train_step(m, opt, *args, outlier_removal=o)
torch.cuda.synchronize()
print('%d conv operands in one training step at %dx%dx%d' % (len(rows), B, H, W))
for kind in ('fwd x1', 'fwd x2', 'dgrad dZ', 'wgrad dZ', 'weight'):
    rs = [r for r in rows if r[0] == kind]
    if not rs:
        continue
    worst27 = max(r[4] for r in rs)
    worst37 = max(r[5] for r in rs)
    print('%-9s %3d tensors: max|x| from %.1e to %.1e; smallest nonzero/max ratio 2^%.0f; worst tensor: %.3f %% of nonzeros below max*2^-27, %.4f %% below max*2^-37'
          % (kind, len(rs), min(r[2] for r in rs), max(r[2] for r in rs),
             min(torch.log2(torch.tensor(r[3] / r[2])).item() for r in rs if r[2] > 0), 100 * worst27, 100 * worst37))
