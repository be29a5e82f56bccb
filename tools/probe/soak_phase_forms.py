'''GPU box: 60 training steps (fp32 tier, published net, batch 8, 900x1600) -- loss trajectory and peak memory; run once with the defaults
and once with every one-launch phase form switched off (RCF_UP2X_ONE_LAUNCH=0 RCF_UP2X_WGRAD_ONE_LAUNCH=0 RCF_S2_DGRAD_ONE_LAUNCH=0
RCF_S2_WGRAD_ONE_LAUNCH=0): forward and input gradients are bitwise, weight gradients differ in fp32 summation order only.'''
import os
import sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__)))))
import torch
import rcf_amd  # noqa: F401
from rcf_amd import synth, train
from rcf_amd.net_utils import OutlierRemoval

dev = torch.device('cuda', 0)
model = train.build_model(synth.PUBLISHED, device=dev)
synth.fill_state_dict_([model.encoder, model.decoder], 1234)
model.compute_dtype = sys.argv[1] if len(sys.argv) > 1 else 'fp32'
opt = train.make_optimizer(model, lr=1e-4)
model.train()
b = synth.make_batch(8, 900, 1600, 64, seed=1234)
image, input_depth, gt, lidar = (b[k].to(dev) for k in ('image', 'input_depth', 'ground_truth', 'lidar_map'))
outlier = OutlierRemoval(kernel_size=7, threshold=1.5)
losses = []
for i in range(60):
    loss = train.train_step(model, opt, image, input_depth, gt, lidar, outlier_removal=outlier)[0]
    if i in (0, 1, 4, 9, 19, 39, 59):
        losses.append((i + 1, float(loss.detach())))
torch.cuda.synchronize()
print('losses', ' '.join('%d:%.6f' % l for l in losses))
print('peak reserved GB %.2f  allocated GB %.2f' % (torch.cuda.max_memory_reserved() / 1e9, torch.cuda.max_memory_allocated() / 1e9))
