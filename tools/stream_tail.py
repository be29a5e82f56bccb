'''GPU box: when does each of the three streams finish its part of the backward?  (events recorded right before the join)'''
import os
import sys

import torch

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from rcf_amd import synth, train            # noqa: E402
from rcf_amd.engine import Engine           # noqa: E402

m = train.build_model(synth.PUBLISHED, device='cuda')
synth.fill_state_dict_([m.encoder, m.decoder], 7)
m.compute_dtype = os.environ.get('RCF_DTYPE', 'fp32')
b = {k: v.cuda() for k, v in synth.make_batch(8, 900, 1600, 64, seed=3).items()}
opt = train.make_optimizer(m, lr=1e-4)
m.train()
marks = {}
real_join = Engine.side_join


def join(self):
    if self.in_backward and 'main' not in marks:
        for name, st in (('main', torch.cuda.current_stream()), ('side', self._side), ('branch', self._branch)):
            if st is not None:
                e = torch.cuda.Event(enable_timing=True)
                e.record(st)
                marks[name] = e
    return real_join(self)


Engine.side_join = join
for i in range(6):
    marks.clear()
    torch.cuda.synchronize()
    t0 = torch.cuda.Event(enable_timing=True)
    t0.record()
    train.train_step(m, opt, b['image'], b['input_depth'], b['ground_truth'], b['lidar_map'])
    t1 = torch.cuda.Event(enable_timing=True)
    t1.record()
    torch.cuda.synchronize()
    if i >= 3:
        print('step %.2f ms; backward chain ends at: %s' % (t0.elapsed_time(t1), ', '.join('%s %.2f ms' % (k, t0.elapsed_time(v)) for k, v in marks.items())))
