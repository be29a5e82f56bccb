'''Sample delivery, batch 8 at 900x1600: the reference's way (float32 CHW arrays built on the host, five
host->device copies) against the raw path (integer pixels up, crop + layout + float conversion on the GPU).  PNG inflation is the
same in both and is left out: the arrays start as what PIL returned.
usage: python tools/loader_bench.py [batch] [reps]'''
import sys, os, time
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import numpy as np
import torch
import rcf_amd
from rcf_amd import datasets

batch = int(sys.argv[1]) if len(sys.argv) > 1 else 8
reps = int(sys.argv[2]) if len(sys.argv) > 2 else 10
H, W, shape = 900, 1600, (768, 1408)
rs = np.random.RandomState(0)
img = rs.randint(0, 256, size=(batch, H, W, 3)).astype(np.uint8)
maps = [(rs.randint(0, 65536, size=(batch, H, W)) * (rs.rand(batch, H, W) < 0.3)).astype(np.uint16) for _ in range(4)]
crop = np.stack([rs.randint(0, H - shape[0] + 1, size=batch), rs.randint(0, W - shape[1] + 1, size=batch)], 1).astype(np.int32)
dev = torch.device('cuda')


def host_way():
    out = []
    for b in range(batch):   # what FusionNetTrainingDataset.__getitem__ does per sample (src/datasets.py:395-450)
        y0, x0 = crop[b]
        s = [np.transpose(img[b].astype(np.float32), (2, 0, 1))]
        for m in maps:
            z = m[b].astype(np.float32) / 256.0
            z[z <= 0] = 0.0
            s.append(z[None])
        out.append([t[:, y0:y0 + shape[0], x0:x0 + shape[1]].astype(np.float32) for t in s])
    tensors = [torch.from_numpy(np.stack([o[j] for o in out])) for j in range(5)]     # default_collate
    return [t.to(dev) for t in tensors]                                              # src/fusionnet_main.py:353-355


def raw_way():
    return datasets.to_device_batch([torch.from_numpy(img)] + [torch.from_numpy(m) for m in maps] + [torch.from_numpy(crop)], dev, shape)


a, b = host_way(), raw_way()
assert all(torch.equal(x, y) for x, y in zip(a, b))
for name, fn in (('host float32 arrays + 5 copies', host_way), ('raw integers + device decode', raw_way)):
    fn(); torch.cuda.synchronize(); t0 = time.time()
    for _ in range(reps): fn()
    torch.cuda.synchronize()
    print('%-34s %7.1f ms per batch of %d' % (name, (time.time() - t0) / reps * 1e3, batch))
d_img, d_maps = torch.from_numpy(img).to(dev), [torch.from_numpy(m).to(dev) for m in maps]
torch.cuda.synchronize(); t0 = time.time()
for _ in range(reps):
    datasets.ops.decode_images(d_img, crop, shape)
    for m in d_maps: datasets.ops.decode_maps(m, 256.0, crop, shape)
torch.cuda.synchronize()
print('%-34s %7.2f ms (inputs resident; %d + %d MB in, %d MB out)' % ('device decode kernels only', (time.time() - t0) / reps * 1e3,
      img.nbytes >> 20, sum(m.nbytes for m in maps) >> 20, batch * 7 * shape[0] * shape[1] * 4 >> 20))
