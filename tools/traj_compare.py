'''Diagnostic (GPU box): loss trajectory of N training steps at batch 2, 450x800 for the f32-MFMA and the split-bf16 conv paths
(same init, same data): they must start identical to ~1e-6 and drift apart only slowly (fp32 chaos).'''
import os, sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
import rcf_amd
from rcf_amd import synth, train

def run(split, steps=8):
    os.environ['RCF_CONV_SPLIT'] = '1' if split else '0'
    torch.manual_seed(1234)
    m = train.build_model(synth.PUBLISHED, device='cuda')
    opt = train.make_optimizer(m, lr=1e-3)
    m.train()
    b = {k: v.cuda() for k, v in synth.make_batch(2, 450, 800, 64, seed=1234).items()}
    out = []
    for _ in range(steps):
        loss, _, _ = train.train_step(m, opt, b['image'], b['input_depth'], b['ground_truth'], b['lidar_map'])
        out.append(float(loss))
    return out

a = run(False); b = run(True); a2 = run(False)
print('step   f32-mfma        split-bf16      rel diff     | f32 run-to-run rel diff')
for i, (x, y, z) in enumerate(zip(a, b, a2)):
    print('%3d  %12.6f  %12.6f   %.2e     | %.2e' % (i, x, y, abs(x - y) / abs(x), abs(x - z) / abs(x)))
