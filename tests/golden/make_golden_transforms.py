'''
Fixture T8: the REAL reference Transforms.transform (src/fusionnet_transforms.py) on seeded inputs, with
torchvision.transforms.functional.adjust_* supplied by oracle/transforms_oracle.py (parity unpinned at that boundary).  The
random decisions the reference drew are recovered by re-drawing them with rcf_amd.fusionnet_transforms.Transforms.draw from the
same CPU seed (same order of torch.rand calls) and stored, so the GPU test does not depend on any RNG stream.

    python tests/golden/make_golden_transforms.py
'''
import os
import sys
import types

sys.dont_write_bytecode = True
ROOT = os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
sys.path.insert(0, ROOT)
import numpy as np
import torch

from oracle import transforms_oracle


def main():
    tv = types.ModuleType('torchvision'); sys.modules['torchvision'] = tv
    tv.transforms = types.ModuleType('torchvision.transforms'); sys.modules['torchvision.transforms'] = tv.transforms
    fn = types.ModuleType('torchvision.transforms.functional')
    fn.adjust_brightness = transforms_oracle.adjust_brightness
    fn.adjust_contrast = transforms_oracle.adjust_contrast
    fn.adjust_saturation = transforms_oracle.adjust_saturation
    sys.modules['torchvision.transforms.functional'] = fn
    tv.transforms.functional = fn
    sys.path.insert(0, '/root/reference/src')
    import fusionnet_transforms as ref
    from rcf_amd.fusionnet_transforms import Transforms as Ours

    gold = os.path.dirname(os.path.abspath(__file__))
    out = {}
    cases = [
        dict(normalized_image_range=[0, 1], random_brightness=[0.8, 1.2], random_contrast=[0.8, 1.2], random_saturation=[0.8, 1.2],
             random_flip_type=['horizontal']),                                       # bash/train_fusionnet_nuscenes.sh:55-59
        dict(normalized_image_range=[-1, 1], random_brightness=[0.5, 1.5], random_contrast=[-1], random_saturation=[0.5, 1.5],
             random_flip_type=['horizontal', 'vertical']),
        dict(normalized_image_range=[0, 255], random_brightness=[-1], random_contrast=[0.6, 1.4], random_saturation=[-1],
             random_flip_type=['none']),
    ]
    rs = np.random.RandomState(5)
    for ci, kw in enumerate(cases):
        n, h, w = 6, 21, 34
        image = np.floor(rs.rand(n, 3, h, w) * 256.0).astype(np.float32)          # 0..255 values stored as float, like the dataset
        maps = [(rs.rand(n, 1, h, w) * (rs.rand(n, 1, h, w) < 0.3)).astype(np.float32) * 80.0 for _ in range(2)]
        t = ref.Transforms(**kw)
        torch.manual_seed(100 + ci)
        res = t.transform([torch.from_numpy(image.copy())], [torch.from_numpy(m.copy()) for m in maps], random_transform_probability=1.0)
        images_out, maps_out = (res[0], res[1])
        torch.manual_seed(100 + ci)
        dec = Ours(rng_device='cpu', **kw).draw(n, torch.device('cpu'), 1.0)
        out['image%d' % ci] = image
        for j, m in enumerate(maps):
            out['map%d_%d' % (ci, j)] = m
            out['map_out%d_%d' % (ci, j)] = maps_out[j].numpy()
        out['image_out%d' % ci] = images_out[0].numpy()
        for k, v in dec.items():
            out['dec%d_%s' % (ci, k)] = v.numpy()
        print('case %d: decisions' % ci, {k: v.tolist() for k, v in dec.items() if k.startswith('do_')})
    np.savez_compressed(os.path.join(gold, 'T8_transforms.npz'), n_cases=len(cases), **out)
    print('T8_transforms.npz %.1f KB' % (os.path.getsize(os.path.join(gold, 'T8_transforms.npz')) / 1024.0))


if __name__ == '__main__':
    main()
