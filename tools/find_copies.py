'''Which host call sites enqueue device copies / fills during one eager FusionNet training step (torch profiler with stacks).'''
import sys, os
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
from torch.profiler import profile, ProfilerActivity
import rcf_amd
from rcf_amd import synth, train
dev = torch.device('cuda')
model = train.build_model(synth.PUBLISHED, device=dev)
opt = train.make_optimizer(model, lr=1e-3)
b = synth.make_batch(1, 64, 96, 8, seed=1)
b = {k: v.to(dev) for k, v in b.items()}
args = (b["image"], b["input_depth"], b["ground_truth"], b["lidar_map"])
for _ in range(3): train.train_step(model, opt, *args)
torch.cuda.synchronize()
with profile(activities=[ProfilerActivity.CPU, ProfilerActivity.CUDA], with_stack=True) as prof:
    train.train_step(model, opt, *args)
    torch.cuda.synchronize()
print(prof.key_averages().table(sort_by='count', row_limit=25, max_name_column_width=60))
cnt = {}
for e in prof.events():
    if e.name.startswith('aten::') and e.stack:
        site = next((s for s in e.stack if 'radar-camera' in s or 'rcf_amd' in s or 'tools/' in s), e.stack[0])
        cnt[(e.name, site)] = cnt.get((e.name, site), 0) + 1
for (n, s), c in sorted(cnt.items(), key=lambda kv: -kv[1])[:25]:
    print('%5d  %-28s %s' % (c, n, s))
