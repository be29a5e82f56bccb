'''GPU box: where the host spends its time enqueueing one eager training step (cProfile over N steps, top functions by own time).'''
import cProfile, pstats, io, os, sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__)))))
import torch
import rcf_amd  # noqa: F401
from rcf_amd import synth, train
from rcf_amd.net_utils import OutlierRemoval
dtype = sys.argv[1] if len(sys.argv) > 1 else 'bf16'
m = train.build_model(synth.PUBLISHED, device='cuda')
synth.fill_state_dict_([m.encoder, m.decoder], 1234)
m.compute_dtype = dtype
opt = train.make_optimizer(m, lr=1e-3)
m.train()
b = {k: v.cuda() for k, v in synth.make_batch(8, 900, 1600, 64, seed=1234).items()}
outl = OutlierRemoval(7, 1.5)
def step():
    return train.train_step(m, opt, b['image'], b['input_depth'], b['ground_truth'], b['lidar_map'], outlier_removal=outl)[0]
for _ in range(3):
    step()
torch.cuda.synchronize()
import time
N = 10
pr = cProfile.Profile()
t0 = time.time()
pr.enable()
for _ in range(N):
    step()
pr.disable()
t1 = time.time()
torch.cuda.synchronize()
print('%s: host enqueue %.2f ms per step (with the profiler on), step wall %.2f ms' % (dtype, (t1 - t0) / N * 1e3, (time.time() - t0) / N * 1e3))
s = io.StringIO()
ps = pstats.Stats(pr, stream=s).sort_stats('tottime')
ps.print_stats(28)
print(s.getvalue()[:6000])
