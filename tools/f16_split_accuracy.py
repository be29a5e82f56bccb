'''CPU experiment (no GPU): how accurate would a TWO-plane fp16 split with a per-tensor power-of-two scale be -- three MFMA products
per multiply like the bf16x3 tier, but 22 significant bits per operand instead of 16-17?  Convolution 64 -> 64, 3x3; the three /
four plane products are accumulated in fp64 here, so the figures isolate the operand representation (a kernel adds fp32 accumulation
error on top, as the fp32 reference column shows).  usage: python tools/f16_split_accuracy.py'''
import math
import torch
import torch.nn.functional as F

torch.manual_seed(0)


def planes16(t, scale):
    a = (t.double() * scale).float()
    h = a.to(torch.float16)
    l = (a - h.float()).to(torch.float16)
    return h.double() / scale, l.double() / scale


def planesb(t):
    t = t.float().contiguous()
    hi = (t.view(torch.int32) & -65536).view(torch.float32)
    lo = (t - hi).to(torch.bfloat16).float()
    return hi.double(), lo.double()


def pow2scale(t, target=2.0 ** 13):
    return 2.0 ** math.floor(math.log2(target / float(t.abs().max())))


conv = lambda a, b: F.conv2d(a, b, padding=1)
for name, gen in [('uniform [-1, 1)', lambda *s: torch.rand(*s) * 2 - 1),
                  ('heavy-tailed: randn x exp(2 randn)', lambda *s: torch.randn(*s) * torch.exp(2 * torch.randn(*s)))]:
    x, w = gen(2, 64, 40, 56), gen(64, 64, 3, 3) / 24
    ref = conv(x.double(), w.double())
    mag = conv(x.double().abs(), w.double().abs())
    e = lambda y: float(((y - ref).abs() / mag).max())
    a0, a1 = planes16(x, pow2scale(x))
    b0, b1 = planes16(w, pow2scale(w))
    c0, c1 = planesb(x)
    d0, d1 = planesb(w)
    print('%-36s max |err| / sum|a||b|:  fp32 conv (fp32 accumulate) %.1e | fp16 2 planes, 3 products %.1e (4 products %.1e) | '
          'bf16 2 planes, 3 products %.1e   [max |x| %.1e, min nonzero |x| %.1e]'
          % (name, e(conv(x, w).double()), e(conv(a0, b0) + conv(a0, b1) + conv(a1, b0)),
             e(conv(a0, b0) + conv(a0, b1) + conv(a1, b0) + conv(a1, b1)), e(conv(c0, d0) + conv(c0, d1) + conv(c1, d0)),
             float(x.abs().max()), float(x.abs()[x != 0].min())))
