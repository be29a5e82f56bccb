'''Diagnostic (GPU box): host / device time of every graph segment and every collective of a segmented data-parallel step, one process
(world size 1 over gloo: the collectives are no-ops in value, the code path is the real one).'''
import os, sys, time
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
import torch.distributed as dist
import rcf_amd
from rcf_amd import synth, train
from rcf_amd.parallel import GradientBuckets

os.environ.setdefault('MASTER_ADDR', '127.0.0.1'); os.environ.setdefault('MASTER_PORT', '29533')
dist.init_process_group(os.environ.get('RCF_DIST_BACKEND', 'gloo'), rank=0, world_size=1)
cfg = synth.PUBLISHED if len(sys.argv) > 1 and sys.argv[1] == 'pub' else synth.TINY
shape = (8, 900, 1600, 64) if cfg is synth.PUBLISHED else (2, 64, 96, 6)
m = train.build_model(cfg, device='cuda')
m._is_data_parallel = True
m._dp = GradientBuckets(m)
opt = train.make_optimizer(m, lr=1e-3)
m.train()
b = {k: v.cuda() for k, v in synth.make_batch(*shape, seed=5).items()}
args = (b['image'], b['input_depth'], b['ground_truth'], b['lidar_map'])
for _ in range(2):
    train.train_step(m, opt, *args)
torch.cuda.synchronize()
t0 = time.time()
for _ in range(3):
    train.train_step(m, opt, *args)
torch.cuda.synchronize()
print('eager DP step: %.2f ms' % ((time.time() - t0) / 3 * 1e3))
step = m.capture_training_step(opt, *args)
segs = step.segments
print('%d segments' % len(segs))
for rep in range(3):
    torch.cuda.synchronize()
    line = []
    handles = []
    t_all = time.time()
    for g, ops_after in segs:
        t0 = time.time(); g.replay(); t1 = time.time(); torch.cuda.synchronize(); t2 = time.time()
        for op in ops_after:
            m._dp.run_exchange(op, handles)
        torch.cuda.synchronize(); t3 = time.time()
        line.append('launch %.2f run %.2f coll %.2f' % ((t1 - t0) * 1e3, (t2 - t1) * 1e3, (t3 - t2) * 1e3))
    print('replay %d: total %.2f ms | ' % (rep, (time.time() - t_all) * 1e3) + ' | '.join(line))
torch.cuda.synchronize()
t0 = time.time()
for _ in range(3):
    step()
torch.cuda.synchronize()
print('segmented step, no syncs in between: %.2f ms' % ((time.time() - t0) / 3 * 1e3))
