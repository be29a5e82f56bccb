'''Loss trajectories of the same training run (published FusionNet, same seeds, same batches) under the arithmetic tiers: fp32 on
three bf16 planes (the reference run), the same with another tiling (how far two fp32 runs drift apart on their own), fp32 on two
scaled fp16 planes (the default of compute_dtype='fp32') and bf16 tensors.  GPU box.
usage: python tools/trajectory.py [steps] [batch] [height] [width]'''
import sys, os
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
import rcf_amd
from rcf_amd import synth, train
from rcf_amd.net_utils import OutlierRemoval

steps = int(sys.argv[1]) if len(sys.argv) > 1 else 200
batch = int(sys.argv[2]) if len(sys.argv) > 2 else 4
h = int(sys.argv[3]) if len(sys.argv) > 3 else 224
w = int(sys.argv[4]) if len(sys.argv) > 4 else 384
dev = torch.device('cuda')
NB = 8   # distinct synthetic batches, cycled
batches = []
for i in range(NB):
    b = synth.make_batch(batch, h, w, 32, seed=100 + i)
    batches.append(tuple(b[k].to(dev) for k in ('image', 'input_depth', 'ground_truth', 'lidar_map')))
outlier = OutlierRemoval(kernel_size=7, threshold=1.5)
runs = {}
for mode in ('fp32_3plane', 'fp32_3plane other tiling', 'fp32', 'bf16'):
    # 'fp32 other tiling': the same exact-fp32 arithmetic with the virtual-tall tiling off (RCF_NO_VT=1) -- a different fp32
    # summation order only, to show how far two fp32 runs drift apart on their own
    if mode == 'fp32_3plane other tiling':
        os.environ['RCF_NO_VT'] = '1'
    else:
        os.environ.pop('RCF_NO_VT', None)
    model = train.build_model(synth.PUBLISHED, device=dev)
    synth.fill_state_dict_([model.encoder, model.decoder], 1234)
    model.compute_dtype = mode.split(' ')[0]
    model.train()
    opt = train.make_optimizer(model, lr=1e-4)
    losses = []
    for s in range(steps):
        losses.append(float(train.train_step(model, opt, *batches[s % NB], outlier_removal=outlier)[0].detach()))
    runs[mode] = losses
print('published FusionNet, batch %d, %dx%d, Adam lr 1e-4, %d distinct batches cycled; mean loss over each window of %d steps' % (batch, h, w, NB, NB))
print('%10s %12s %12s %12s %12s | %12s %12s %12s' % ('steps', '3 bf16 pl.', '3pl tiling2', '2 fp16 pl.', 'bf16', 'tiling2/3pl-1', '2pl/3pl-1', 'bf16/3pl-1'))
for s0 in range(0, steps, max(NB, steps // 10 // NB * NB)):
    win = lambda m: sum(runs[m][s0:s0 + NB]) / len(runs[m][s0:s0 + NB])
    a, a2, b, c = win('fp32_3plane'), win('fp32_3plane other tiling'), win('fp32'), win('bf16')
    print('%4d..%-4d %12.5f %12.5f %12.5f %12.5f | %+12.2e %+12.2e %+12.2e' % (s0, s0 + NB - 1, a, a2, b, c, a2 / a - 1, b / a - 1, c / a - 1))
print('first step: %.6f %.6f %.6f %.6f' % (runs['fp32_3plane'][0], runs['fp32_3plane other tiling'][0], runs['fp32'][0], runs['bf16'][0]))
