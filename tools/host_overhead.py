'''Host-side cost of one FusionNet training step: the published network on a tiny input (GPU time negligible), so the wall
clock per step is the Python + ctypes + allocator time needed to enqueue the ~1100 launches -- eagerly, and as one hipGraph replay
(FusionNetModel.capture_training_step).'''
import sys, os, time
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
import rcf_amd
from rcf_amd import synth, train
dev = torch.device('cuda')
model = train.build_model(synth.PUBLISHED, device=dev)
opt = train.make_optimizer(model, lr=1e-3)
b = synth.make_batch(1, 64, 96, 8, seed=1)
b = {k: v.to(dev) for k, v in b.items()}
args = (b["image"], b["input_depth"], b["ground_truth"], b["lidar_map"])
for _ in range(3): train.train_step(model, opt, *args)
torch.cuda.synchronize(); t0 = time.time()
n = 10
for _ in range(n): train.train_step(model, opt, *args)
torch.cuda.synchronize()
print('host-bound step (published net, 64x96, batch 1), eager: %.2f ms' % ((time.time() - t0) / n * 1e3))
step = model.capture_training_step(opt, *args)
for _ in range(3): step(*args)
torch.cuda.synchronize(); t0 = time.time()
n = 50
for _ in range(n): step(*args)
t_host = time.time() - t0          # time to ENQUEUE the replays (the host is free again after this)
torch.cuda.synchronize()
t_all = time.time() - t0
print('same step as one hipGraph replay: host %.3f ms per step to enqueue, %.3f ms per step including the (launch-bound) GPU execution'
      % (t_host / n * 1e3, t_all / n * 1e3))
