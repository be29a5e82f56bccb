'''GPU box: which stock aten operators one eager training step still calls (count, shapes), e.g. where the ~114 blit copies per step come from.'''
import os, sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__)))))
import torch
import rcf_amd  # noqa: F401
from rcf_amd import synth, train
from rcf_amd.net_utils import OutlierRemoval
dtype = sys.argv[1] if len(sys.argv) > 1 else 'fp32'
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
from torch.profiler import profile, ProfilerActivity
with profile(activities=[ProfilerActivity.CPU], record_shapes=True) as prof:
    step()
torch.cuda.synchronize()
rows = {}
for e in prof.events():
    if e.name.startswith('aten::'):
        k = (e.name, str(e.input_shapes)[:80])
        rows[k] = rows.get(k, 0) + 1
agg = {}
for (n, s), c in rows.items():
    agg[n] = agg.get(n, 0) + c
print(sorted(agg.items(), key=lambda kv: -kv[1])[:25])
for (n, s), c in sorted(rows.items(), key=lambda kv: -kv[1])[:30]:
    if n in ('aten::copy_', 'aten::clone', 'aten::to', 'aten::_to_copy', 'aten::zero_', 'aten::fill_', 'aten::zeros', 'aten::add_', 'aten::mul', 'aten::add', 'aten::sub', 'aten::contiguous', 'aten::select', 'aten::slice'):
        print(c, n, s)
