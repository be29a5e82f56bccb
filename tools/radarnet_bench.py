'''RadarNet stage-1 training step at the shipped size (bash/train_radarnet_nuscenes.sh: batch 6 images x 4 points, patch 900x288,
padded image 900x1888), fp32, synthetic data.  usage: python tools/radarnet_bench.py [n_images] [points_per_image] [steps]'''
import sys, os, time
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
import rcf_amd
from rcf_amd import synth, radarnet_model, optim

n = int(sys.argv[1]) if len(sys.argv) > 1 else 6
k = int(sys.argv[2]) if len(sys.argv) > 2 else 4
steps = int(sys.argv[3]) if len(sys.argv) > 3 else 5
dtype = sys.argv[4] if len(sys.argv) > 4 else 'fp32'
dev = torch.device('cuda')
m = radarnet_model.RadarNetModel(device=dev, **synth.RADARNET_PUBLISHED)
m.compute_dtype = dtype
synth.fill_state_dict_([m.encoder, m.decoder], 41)
b = synth.make_radarnet_batch(7, n=n, k=k, h=900, w=1888, patch_w=288)
b = {key: (v.to(dev) if isinstance(v, torch.Tensor) else [t.to(dev) for t in v]) for key, v in b.items()}
opt = torch.optim.Adam([{'params': m.parameters(), 'weight_decay': 0.0}], lr=2e-4)
m.train()
def step():
    logits = m.forward(b['image'], b['point'], b['bounding_boxes'])
    loss, _ = m.compute_loss(logits, b['ground_truth'], b['validity_map'], w_positive_class=2.0)
    opt.zero_grad(); loss.backward(); opt.step()
    return loss
for _ in range(2): loss = step()
torch.cuda.synchronize(); t0 = time.time()
for _ in range(steps): loss = step()
torch.cuda.synchronize(); dt = (time.time() - t0) / steps
print('RadarNet ' + dtype + ' train: %d images x %d points, %.1f ms/step, %.1f images/s, %.1f crops/s, loss %.5f' % (n, k, dt * 1e3, n / dt, n * k / dt, float(loss.detach())))
print('peak memory %.2f GB' % (torch.cuda.max_memory_allocated() / 1e9))
