import sys; sys.path.insert(0,'/root/repo')
import torch, rcf_amd
from rcf_amd import synth, train
from rcf_amd.net_utils import OutlierRemoval
dev=torch.device('cuda')
m=train.build_model(synth.PUBLISHED, device=dev); synth.fill_state_dict_([m.encoder,m.decoder],1234)
opt=train.make_optimizer(m, lr=1e-3); m.train()
import os
B,H,W=(int(v) for v in os.environ.get('DBG_SHAPE','2,224,384').split(','))
b=synth.make_batch(B,H,W,64,seed=1); args=[b[k].to(dev) for k in ('image','input_depth','ground_truth','lidar_map')]
o=OutlierRemoval(kernel_size=7, threshold=1.5)
from rcf_amd import ops
if os.environ.get('DBG_PROF'): m._engine.prof = ops.KernelTimer()
for i in range(5):
    train.train_step(m,opt,*args,outlier_removal=o)
    p=m._engine.plan
    print(i,p.state,p.dirty,p.pos,len(p.entries),p._n_pack,p._n_phase,p.last_mismatch)
