'''
Golden vectors for the radar point -> grid scatter: runs the REAL reference function radarnet_main.forward
(src/radarnet_main.py:534-591) on CPU with a stand-in model whose forward() returns prescribed response crops,
so only the scatter logic of the reference is exercised.  Build-container only.

    python tests/golden/make_golden_scatter.py

Shims: torchvision is absent; the reference uses torchvision.transforms.functional.pad(image, (p,0,p,0), 'edge') once
(:540-543), provided here by torch.nn.functional.pad(mode='replicate'); tensorboard / PIL-free stubs for the imports.
'''
import os, sys, types
sys.dont_write_bytecode = True
ROOT = os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
sys.path.insert(0, ROOT)
import numpy as np
import torch


def _stub(name, **attrs):
    m = types.ModuleType(name); m.__dict__.update(attrs); sys.modules[name] = m; return m


def import_reference():
    def tv_pad(img, padding, padding_mode='constant'):
        l, t, r, b = padding
        mode = {'edge': 'replicate', 'constant': 'constant'}[padding_mode]
        x = img if img.dim() == 4 else img.unsqueeze(0)
        return torch.nn.functional.pad(x, (l, r, t, b), mode=mode).reshape(*img.shape[:-2], img.shape[-2] + t + b, img.shape[-1] + l + r)
    tv = _stub('torchvision')
    tv.ops = _stub('torchvision.ops'); tv.utils = _stub('torchvision.utils')
    tv.transforms = _stub('torchvision.transforms')
    tv.transforms.functional = _stub('torchvision.transforms.functional', pad=tv_pad)
    tb = _stub('torch.utils.tensorboard', SummaryWriter=object)
    if not hasattr(np, 'infty'): np.infty = np.inf
    sys.path.insert(0, '/root/reference/src')
    import radarnet_main
    return radarnet_main


class FakeModel(object):
    '''Stands in for RadarNetModel: forward() returns the prescribed crops (K x 1 x Hc x Wc).'''
    def __init__(self, crops, patch):
        self.crops = crops; self.input_patch_size_image = patch
    def forward(self, image, point, bounding_boxes, return_logits=False):
        return self.crops


def main():
    ref = import_reference()
    from oracle.radar_scatter_oracle import radar_scatter
    from rcf_amd import synth
    make_case = synth.make_scatter_case
    out = {}
    cases = [(8, 24, 40, 12, 1, True), (16, 30, 64, 16, 2, True), (64, 45, 100, 18, 3, False), (5, 9, 20, 6, 4, True)]
    for ci, (k, h, w, wc, seed, small_z) in enumerate(cases):
        crops, pts = make_case(k, h, w, wc, seed, small_z)
        model = FakeModel(torch.from_numpy(crops).unsqueeze(1), [h, wc])
        image = torch.zeros(1, 3, h, w)
        depth, resp = ref.forward(model, image, torch.from_numpy(pts), None, device=torch.device('cpu'))
        depth = depth.numpy().reshape(h, w).astype(np.float32); resp = resp.numpy().reshape(h, w)
        od, orr = radar_scatter(crops, pts, w, strict_reference=True)
        assert np.array_equal(od, depth), ('depth', ci)
        assert np.array_equal(orr, resp), ('response', ci)
        # the quirk must actually fire in the small-z cases: strict differs from z[argmax] somewhere
        nd, _ = radar_scatter(crops, pts, w, strict_reference=False)
        print('case %d: K=%d %dx%d wc=%d  oracle == reference; pixels where the in-place chain changes the answer: %d'
              % (ci, k, h, w, wc, int((np.trunc(nd) != od).sum())))
        out['meta%d' % ci] = np.array([k, h, w, wc, seed, int(small_z)])
        out['depth%d' % ci] = depth; out['resp%d' % ci] = resp
    np.savez_compressed(os.path.join(os.path.dirname(os.path.abspath(__file__)), 'T4_radar_scatter.npz'), n_cases=len(cases), **out)


if __name__ == '__main__':
    main()
