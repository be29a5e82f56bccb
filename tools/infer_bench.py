'''FusionNet inference (eval-mode BatchNorm, no tape) at 900x1600: usage python tools/infer_bench.py [batch] [fp32|bf16] [reps] [graph]'''
import sys, os, time
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
import rcf_amd
from rcf_amd import synth, train
batch = int(sys.argv[1]) if len(sys.argv) > 1 else 8
dtype = sys.argv[2] if len(sys.argv) > 2 else 'fp32'
reps = int(sys.argv[3]) if len(sys.argv) > 3 else 5
use_graph = len(sys.argv) > 4 and sys.argv[4] == 'graph'
dev = torch.device('cuda')
m = train.build_model(synth.PUBLISHED, device=dev)
m.compute_dtype = dtype
m.eval()
b = synth.make_batch(batch, 900, 1600, 64, seed=3)
img, dep = b['image'].to(dev), b['input_depth'].to(dev)
with torch.no_grad():
    fwd = m.capture_inference(img, dep) if use_graph else m.forward
    for _ in range(2): out = fwd(img, dep)
    torch.cuda.synchronize(); t0 = time.time()
    for _ in range(reps): out = fwd(img, dep)
    torch.cuda.synchronize()
dt = (time.time() - t0) / reps
print('FusionNet inference %s%s batch %d: %.2f ms, %.1f samples/s, peak %.1f GB' % (dtype, ' hipGraph' if use_graph else '', batch, dt * 1e3, batch / dt, torch.cuda.max_memory_allocated() / 1e9))
