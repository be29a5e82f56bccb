import os, sys
sys.path.insert(0, os.environ.get('GRAFT_REPO_ROOT', '/root/repo'))
import torch
from rcf_amd import synth, train
m = train.build_model(synth.PUBLISHED, device='cuda')
synth.fill_state_dict_([m.encoder, m.decoder], 7)
m.compute_dtype = os.environ.get('RCF_DTYPE', 'fp32')
b = {k: v.cuda() for k, v in synth.make_batch(8, 900, 1600, 64, seed=3).items()}
opt = train.make_optimizer(m, lr=1e-4)
m.train()
for i in range(80):
    loss = train.train_step(m, opt, b['image'], b['input_depth'], b['ground_truth'], b['lidar_map'])[0]
    if i in (5, 20, 40, 79):
        torch.cuda.synchronize()
        print(i, 'loss %.5f' % float(loss), 'allocated %.2f GB' % (torch.cuda.memory_allocated() / 1e9), 'reserved %.2f GB' % (torch.cuda.memory_reserved() / 1e9), 'peak %.2f GB' % (torch.cuda.max_memory_allocated() / 1e9))
