'''Numerics of Winograd F(2x2, 3x3) on two scaled fp16 planes (CPU emulation, no GPU): would the exact tier's bar hold?
Input transform V = B^T d B and weight transform U = G g G^T in fp32, both split into two fp16 planes of (value * power-of-two scale),
three products per multiply accumulated over the input channels (fp32 accumulate emulated in fp64 + one fp32 rounding per point sum --
an optimistic stand-in for the MFMA's fp32 accumulation order), output transform Y = A^T M A in fp32.  Compared with an fp64
convolution of the fp32 operands, as tests/test_hip_f16x2.py compares the direct kernels (EXACT_TOL 2e-6 of the output's max-abs).
    python tools/winograd_numerics.py'''
import numpy as np
import torch
import torch.nn.functional as F

BT = torch.tensor([[1, 0, -1, 0], [0, 1, 1, 0], [0, -1, 1, 0], [0, 1, 0, -1]], dtype=torch.float32)
G = torch.tensor([[1, 0, 0], [.5, .5, .5], [.5, -.5, .5], [0, 0, 1]], dtype=torch.float32)
AT = torch.tensor([[1, 1, 1, 0], [0, 1, -1, -1]], dtype=torch.float32)


def scale_of(amax):
    amax = float(amax)
    if amax == 0.0:
        return 1.0
    e = int(np.floor(np.log2(amax)))
    return 2.0 ** (14 - e)


def planes(t, s):
    ts = (t.float() * s)
    p0 = ts.to(torch.float16).to(torch.float32)
    p1 = (ts - p0).to(torch.float16).to(torch.float32)
    return p0.double(), p1.double()


def direct_two_plane(x, w):
    sx, sw = scale_of(x.abs().max()), scale_of(w.abs().max())
    x0, x1 = planes(x, sx)
    w0, w1 = planes(w, sw)
    c = lambda a, b: F.conv2d(a, b, padding=1)
    return ((c(x0, w0) + c(x0, w1) + c(x1, w0)) / (sx * sw)).float()


def winograd_two_plane(x, w):
    n, ci, h, wd = x.shape
    co = w.shape[0]
    assert h % 2 == 0 and wd % 2 == 0
    xp = F.pad(x, (1, 1, 1, 1))
    # 4x4 patches with stride 2: [n, ci, th, tw, 4, 4]
    d = xp.unfold(2, 4, 2).unfold(3, 4, 2)
    V = torch.einsum('ij,ncabjk,lk->ncabil', BT, d, BT)          # fp32 transform (each entry: +- of 4 inputs)
    U = torch.einsum('ij,ocjk,lk->ocil', G, w, G)                 # fp32
    sv, su = scale_of(V.abs().max()), scale_of(U.abs().max())
    V0, V1 = planes(V, sv)
    U0, U1 = planes(U, su)
    m = lambda a, b: torch.einsum('ncabil,ocil->noabil', a, b)
    M = ((m(V0, U0) + m(V0, U1) + m(V1, U0)) / (sv * su)).float()  # one fp32 rounding of the channel sum per point
    Y = torch.einsum('ij,noabjk,lk->noabil', AT, M, AT)           # fp32 output transform [n, co, th, tw, 2, 2]
    return Y.permute(0, 1, 2, 4, 3, 5).reshape(n, co, h, wd)


def rel(a, b):
    return float((a.double() - b.double()).abs().max() / b.double().abs().max())


if __name__ == '__main__':
    torch.manual_seed(0)
    print('%-44s %10s %10s %10s' % ('case', 'direct', 'winograd', 'fp32 conv'))
    for name, ci, co, h, w, kind in [('64->64 uniform', 64, 64, 32, 48, 'u'), ('256->256 uniform', 256, 256, 16, 16, 'u'),
                                     ('64->64 gaussian x, heavy-tailed w', 64, 64, 32, 48, 'g'), ('32->32 one outlier', 32, 32, 32, 48, 'o'),
                                     ('512 (concat) -> 256 uniform', 512, 256, 8, 12, 'u')]:
        x = torch.rand(2, ci, h, w) * 2 - 1 if kind != 'g' else torch.randn(2, ci, h, w)
        wt = (torch.rand(co, ci, 3, 3) * 2 - 1) / np.sqrt(ci * 9)
        if kind == 'g':
            wt = torch.distributions.StudentT(3.0).sample((co, ci, 3, 3)) / np.sqrt(ci * 9)
        if kind == 'o':
            x[0, 3, 5, 7] = 300.0
        ref = F.conv2d(x.double(), wt.double(), padding=1)
        print('%-44s %10.2e %10.2e %10.2e' % (name, rel(direct_two_plane(x, wt), ref), rel(winograd_two_plane(x, wt), ref),
                                             rel(F.conv2d(x, wt, padding=1), ref)))
