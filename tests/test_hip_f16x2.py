'''
GPU parity tests of the two-plane fp16 arithmetic (rcf_conv_desc.precision = RCF_PREC_F16X2, include/rcf_hip.h): fp32 tensors, each
operand x of the split convolution kernels carried as two fp16 planes of x * s -- s the power of two that puts its tensor's max|x|
into [2^14, 2^15) -- multiplied as a0*b0 + a0*b1 + a1*b0 with fp32 accumulation and rescaled by 1 / (s_a s_b).  The per-tensor
maxima arrive as device scalars (ops.amax here; in the network, from the kernels that write the tensors).

Bars.  (1) Against an fp64 evaluation of exactly that formula the kernels differ only by fp32 summation order.  (2) Against an fp64
convolution of the fp32 operands the error must be that of the EXACT tier: within 2x of the f32-MFMA kernel's error (the bar
tests/test_hip_ops.py::test_split_bf16_conv_is_fp32_accurate holds the three-plane bf16 split to).  (3) Operand distributions that
stress a per-tensor scale: heavy tails, one huge outlier, all-zero and denormal tensors.  (4) The published net against the real
reference's fixtures at north_star's 1e-3 (measured ~1e-6, like the exact tier).
'''
import os

import numpy as np
import pytest
import torch
import torch.nn.functional as F

pytestmark = pytest.mark.gpu

NORTH_STAR = 1e-3     # BASELINE.json north_star: "within 1e-3 rel fp32"
EXACT_TOL = 2e-6      # against fp64, as a fraction of the reference tensor's max-abs: fp32-convolution class (K <= 4608 terms)
ORDER_TOL = 2e-6      # fp32 summation order only (against the emulated three-product formula)


@pytest.fixture(scope='module')
def ops():
    import rcf_amd  # noqa: F401
    from rcf_amd import _lib, ops as _ops
    assert torch.cuda.is_available(), 'GPU tests need a GPU'
    assert _lib.load().rcf_device_ok() == 1, 'librcf_hip.so: no gfx950 device'
    if os.environ.get('RCF_CONV_SPLIT') == '0':
        pytest.skip('the two-plane arithmetic lives in the split kernels, which RCF_CONV_SPLIT=0 turns off')
    return _ops


def rnd(*shape, seed=0, scale=1.0):
    g = torch.Generator().manual_seed(seed)
    return (torch.rand(*shape, generator=g) * 2 - 1) * scale


def nhwc(t):
    return t.permute(0, 2, 3, 1).contiguous().cuda()


def nchw(t):
    return t.detach().cpu().permute(0, 3, 1, 2).contiguous()


def rel(a, b):
    return float((a.double() - b.double()).abs().max() / (b.double().abs().max() + 1e-30))


def scale_of(amax):
    '''rcf_scale_of_amax (csrc/rcf_common.h): the power of two that puts amax into [2^14, 2^15); 1 for an all-zero tensor.'''
    amax = float(amax)
    if amax == 0.0:
        return 1.0
    e = int(np.floor(np.log2(amax))) if amax >= 2.0 ** -126 else -127
    return 2.0 ** min(max(14 - e, -126), 126)


def planes(t, s):
    '''The two fp16 planes of t * s as the kernels form them (both round to nearest even), returned in float64.'''
    ts = (t.float() * s).contiguous()
    p0 = ts.to(torch.float16).to(torch.float32)
    p1 = (ts - p0).to(torch.float16).to(torch.float32)
    return p0.double(), p1.double()


def x3(fn, a, b, amax_a=None, amax_b=None):
    '''fn bilinear in (a, b): the three products of the two-plane operands, in fp64, rescaled.'''
    sa = scale_of(a.abs().max() if amax_a is None else amax_a)
    sb = scale_of(b.abs().max() if amax_b is None else amax_b)
    a0, a1 = planes(a, sa)
    b0, b1 = planes(b, sb)
    return (fn(a0, b0) + fn(a0, b1) + fn(a1, b0)) / (sa * sb)


def dev_amax(ops, *ts):
    '''Device scalar holding the maximum of |.| over the given host tensors (what a producing kernel would have accumulated).'''
    out = torch.zeros(1, device='cuda')
    for t in ts:
        ops.amax(t.contiguous().cuda(), out, accumulate=True)
    return out


# (ksize, stride, c1, c2, cout, n, h, w, up_from)
CASES = [
    (3, 1, 64, 0, 64, 2, 33, 64, None),
    (3, 1, 64, 32, 64, 2, 17, 40, None),        # decoder concat
    (3, 1, 32, 0, 32, 1, 40, 100, None),        # 32-co tiles, 16-pixel rows
    (3, 1, 128, 128, 128, 1, 15, 26, None),
    (3, 1, 64, 0, 32, 1, 70, 102, (35, 51)),    # nearest-upsample gather on load
    (3, 1, 256, 0, 256, 2, 8, 13, None),        # small layer: 32-co half workgroups
    (3, 2, 32, 0, 64, 2, 45, 80, None),         # stride 2 on the split kernel
    (3, 2, 128, 0, 256, 1, 29, 50, None),
    (1, 1, 32, 0, 64, 2, 45, 80, None),         # 1x1 (fusion convs): the streaming kernel, operands straight from global memory
    (1, 1, 16, 0, 32, 2, 35, 51, None),
    (1, 1, 64, 0, 64, 1, 29, 50, None),         # 4 k-steps x 2 co tiles: the most weight pieces that fit the registers
    (1, 1, 32, 0, 128, 3, 20, 37, None),        # 2 k-steps x 4 co tiles
    (1, 2, 32, 0, 64, 2, 45, 80, None),         # projection: stride 2 through the source address
]


def _case(case, seed):
    k, s, c1, c2, co, n, h, w, up = case
    hs, ws = (h, w) if up is None else up
    x1 = rnd(n, c1, hs, ws, seed=seed)
    x2 = rnd(n, c2, h, w, seed=seed + 1) if c2 else None
    wt = rnd(co, c1 + c2, k, k, seed=seed + 2, scale=1.0 / np.sqrt((c1 + c2) * k * k))
    xin = x1 if up is None else F.interpolate(x1, size=(h, w))
    if x2 is not None:
        xin = torch.cat([xin, x2], 1)
    return x1, x2, wt, xin


def _run_fwd(ops, d, x1, x2, wt, amax_x=None, amax_w=None, stats=False):
    '''conv forward under descriptor d; with maxima (device scalars) the scaled entry points, else the plain ones.'''
    info = ops.conv_query(d)
    packed = torch.empty(info.packed_weight_floats, device='cuda')
    ops.conv_pack(d, wt.cuda(), packed, amax_w)
    out = torch.full((d.n, d.h_out, d.w_out, d.c_out), float('nan'), device='cuda')
    part = torch.full((info.n_partials, 2, d.c_out), float('nan'), device='cuda', dtype=torch.float64) if stats else None
    sc = ops.make_scales(amax_x, amax_x if x2 is not None else None, amax_w) if amax_w is not None else None
    ops.conv_fwd(d, nhwc(x1), None if x2 is None else nhwc(x2), packed, out, part, scales=sc)
    torch.cuda.synchronize()
    return nchw(out), part, info


@pytest.mark.parametrize('case', CASES, ids=[str(c) for c in CASES])
def test_conv_forward_input_gradient_weight_gradient(ops, case):
    k, s, c1, c2, co, n, h, w, up = case
    x1, x2, wt, xin = _case(case, 40)
    hs, ws = (h, w) if up is None else up
    ops.set_precision('f16x2')
    try:
        d = ops.make_fwd_desc(n, h, w, c1, c2, co, k, s, hs, ws, 0 if up is None else 1)
        ax, aw = dev_amax(ops, xin), dev_amax(ops, wt)
        got, part, info = _run_fwd(ops, d, x1, x2, wt, ax, aw, stats=True)
        assert 40000 <= info.kernel_id < 50000, info.kernel_id            # a two-plane split kernel was selected
        conv = lambda a, b: F.conv2d(a, b, stride=s, padding=k // 2)
        ref64 = conv(xin.double(), wt.double())
        assert rel(got, x3(conv, xin, wt)) < ORDER_TOL
        e = rel(got, ref64)
        assert e < EXACT_TOL, e
        # BatchNorm statistics of the written values (fp64 sums of the fp32 outputs)
        st = part.sum(0).cpu()
        np.testing.assert_allclose(st[0].numpy(), got.double().sum((0, 2, 3)).numpy(), rtol=1e-9, atol=1e-7)
        np.testing.assert_allclose(st[1].numpy(), (got.double() ** 2).sum((0, 2, 3)).numpy(), rtol=1e-9, atol=1e-7)

        # weight gradient
        dz = rnd(n, co, d.h_out, d.w_out, seed=77, scale=1e-3)    # gradients live decades below the activations
        adz = dev_amax(ops, dz)
        dw = torch.full(wt.shape, float('nan'), device='cuda')
        wsb = torch.empty(max(1, info.wgrad_workspace_floats), device='cuda')
        two_plane_wgrad = info.wgrad_kernel_id >= 50000
        ops.conv_wgrad(d, nhwc(x1), None if x2 is None else nhwc(x2), nhwc(dz), dw, wsb,
                       scales=ops.make_scales(ax, ax if x2 is not None else None, None, adz) if two_plane_wgrad else None)
        wg = lambda a, b: torch.nn.grad.conv2d_weight(a, wt.shape, b, stride=s, padding=k // 2)
        if s == 1 and k != 1:   # (1x1 weight gradients stay on the f32 MFMA)
            assert two_plane_wgrad, info.wgrad_kernel_id
            assert rel(dw.cpu(), x3(wg, xin, dz)) < ORDER_TOL
        assert rel(dw.cpu(), wg(xin.double(), dz.double())) < EXACT_TOL

        # input gradient of source 1 (stride 1: the same kernel on flipped weights; the stride-2 one is four phase convolutions)
        if s == 1 and up is None:
            dd = ops.make_dgrad_desc(d, 0, c1, False)
            di = ops.conv_query(dd)
            two_plane_dgrad = 40000 <= di.kernel_id < 50000
            assert two_plane_dgrad or (k == 1 and co > 64)    # (a 1x1 input gradient from > 64 channels stays on the f32 MFMA)
            pk = torch.empty(di.packed_weight_floats, device='cuda')
            ops.conv_pack(dd, wt.cuda(), pk, aw if two_plane_dgrad else None)
            dx = torch.full((n, h, w, c1), float('nan'), device='cuda')
            ops.conv_fwd(dd, nhwc(dz), None, pk, dx, None, scales=ops.make_scales(adz, None, aw) if two_plane_dgrad else None)
            dg = lambda a, b: torch.nn.grad.conv2d_input(xin.shape, b, a, stride=1, padding=k // 2)[:, :c1]
            if two_plane_dgrad:
                assert rel(nchw(dx), x3(dg, dz, wt)) < ORDER_TOL
            assert rel(nchw(dx), dg(dz.double(), wt.double())) < EXACT_TOL
            if two_plane_dgrad and k == 1:   # += into an existing gradient (the second consumer of a tensor)
                base = rnd(n, c1, h, w, seed=78, scale=1e-3)
                dx2 = nhwc(base).clone()
                dda = ops.make_dgrad_desc(d, 0, c1, True)
                ops.conv_fwd(dda, nhwc(dz), None, pk, dx2, None, scales=ops.make_scales(adz, None, aw))
                assert rel(nchw(dx2), base.double() + dg(dz.double(), wt.double())) < EXACT_TOL
    finally:
        ops.set_precision('fp32')


ACC_CASES = [(3, 1, 64, 0, 64, 2, 33, 64, None), (3, 1, 256, 256, 256, 1, 15, 26, None), (3, 1, 32, 0, 32, 1, 40, 100, None)]


@pytest.mark.parametrize('case', ACC_CASES, ids=[str(c) for c in ACC_CASES])
def test_two_fp16_planes_are_as_accurate_as_the_exact_tier(ops, case, monkeypatch):
    '''The admission bar of the exact tier (tests/test_hip_ops.py::test_split_bf16_conv_is_fp32_accurate): against an fp64 reference the
    kernel must be within 2x of the exact f32-MFMA kernel's error, on the forward pass and on the input gradient.  Printed next to it:
    the three-plane bf16 split.'''
    k, s, c1, c2, co, n, h, w, up = case
    x1, x2, wt, xin = _case(case, 10)
    ref64 = F.conv2d(xin.double(), wt.double(), padding=1)
    dz = rnd(*ref64.shape, seed=33, scale=1e-3)
    want_dx = torch.nn.grad.conv2d_input(xin.shape, wt.double(), dz.double(), padding=1)[:, :c1]
    errs = {}
    for mode in ('f32mfma', 'bf16x3planes', 'f16x2'):
        monkeypatch.setenv('RCF_CONV_SPLIT', '0' if mode == 'f32mfma' else '1')
        ops.set_precision('f16x2' if mode == 'f16x2' else 'fp32')
        try:
            d = ops.make_fwd_desc(n, h, w, c1, c2, co, k, s, h, w, 0)
            f16 = mode == 'f16x2'
            ax, aw, adz = (dev_amax(ops, xin), dev_amax(ops, wt), dev_amax(ops, dz)) if f16 else (None, None, None)
            got, _, info = _run_fwd(ops, d, x1, x2, wt, ax, aw)
            assert (40000 <= info.kernel_id < 50000) == f16 and (info.kernel_id >= 5000) == (mode != 'f32mfma')
            errs['fwd ' + mode] = rel(got, ref64)
            dd = ops.make_dgrad_desc(d, 0, c1, False)
            pk = torch.empty(ops.conv_query(dd).packed_weight_floats, device='cuda')
            ops.conv_pack(dd, wt.cuda(), pk, aw)
            dx = torch.full((n, h, w, c1), float('nan'), device='cuda')
            ops.conv_fwd(dd, nhwc(dz), None, pk, dx, None, scales=ops.make_scales(adz, None, aw) if f16 else None)
            errs['dx ' + mode] = rel(nchw(dx), want_dx)
        finally:
            ops.set_precision('fp32')
    print('error against fp64 (max-abs, relative to max|ref|):', {k_: '%.2e' % v for k_, v in errs.items()})
    assert errs['fwd f16x2'] < 2.0 * errs['fwd f32mfma'] + 1e-7 and errs['dx f16x2'] < 2.0 * errs['dx f32mfma'] + 1e-7
    assert errs['fwd f16x2'] < 1e-5 and errs['dx f16x2'] < 1e-5


def _dist(kind, shape, seed):
    g = torch.Generator().manual_seed(seed)
    u = torch.rand(*shape, generator=g) * 2 - 1
    if kind == 'uniform':
        return u
    if kind == 'heavy_tail':       # log-uniform magnitudes over eleven decades, random signs
        return torch.sign(u) * torch.pow(10.0, torch.rand(*shape, generator=g) * 11.0 - 8.0)
    if kind == 'outlier':          # one element 1e6 times everything else
        t = u.clone()
        t.view(-1)[t.numel() // 3] = 1.0e6
        return t
    if kind == 'tiny':             # a tensor living at 1e-30 (fp32 normal, far below fp16's range without the scale)
        return u * 1e-30
    if kind == 'huge':
        return u * 1e25
    if kind == 'denormal':         # fp32 denormals only
        return u * 1e-40
    if kind == 'zero':
        return torch.zeros(*shape)
    raise ValueError(kind)


@pytest.mark.parametrize('xkind,wkind', [('heavy_tail', 'uniform'), ('uniform', 'heavy_tail'), ('heavy_tail', 'heavy_tail'),
                                         ('outlier', 'uniform'), ('uniform', 'outlier'), ('tiny', 'uniform'), ('huge', 'tiny'),
                                         ('denormal', 'uniform'), ('zero', 'uniform'), ('uniform', 'zero')])
def test_per_tensor_scale_on_hostile_operand_distributions(ops, xkind, wkind, monkeypatch):
    '''What a per-tensor scale has to survive.  The error model (csrc/rcf_common.h): each operand element carries
    <= 2^-23 |x| + 2^-39 max|x| of error, so an output is off by at most  sum |a||b| * 2^-22  +  (2^-38 amax_a amax_b) * K  -- checked
    against fp64 per output element with that bound (x4 for the fp32 accumulation), and against the exact f32-MFMA kernel's own
    error where that is larger.  All-zero tensors give exact zeros; a tensor of fp32 denormals (1e-40) is representable after the
    scale and gives the f32 kernel's result up to its flush-to-zero of denormal products.'''
    n, c, co, h, w = 2, 64, 64, 20, 36
    x = _dist(xkind, (n, c, h, w), 1)
    wt = _dist(wkind, (co, c, 3, 3), 2) * (1.0 / 24.0 if wkind == 'uniform' else 1.0)
    ref = F.conv2d(x.double(), wt.double(), padding=1)
    bound = (F.conv2d(x.double().abs(), wt.double().abs(), padding=1) * 2.0 ** -22
             + 2.0 ** -38 * float(x.abs().max()) * float(wt.abs().max()) * 9 * c) * 4.0
    ops.set_precision('f16x2')
    try:
        d = ops.make_fwd_desc(n, h, w, c, 0, co, 3, 1, h, w, 0)
        got, _, info = _run_fwd(ops, d, x, None, wt, dev_amax(ops, x), dev_amax(ops, wt))
        assert 40000 <= info.kernel_id < 50000
    finally:
        ops.set_precision('fp32')
    monkeypatch.setenv('RCF_CONV_SPLIT', '0')
    got32, _, _ = _run_fwd(ops, ops.make_fwd_desc(n, h, w, c, 0, co, 3, 1, h, w, 0), x, None, wt)
    assert bool(torch.isfinite(got).all())
    err, err32 = (got.double() - ref).abs(), (got32.double() - ref).abs()
    if 'zero' in (xkind, wkind):
        assert float(got.abs().max()) == 0.0
        return
    if xkind == 'denormal':
        # products of 1e-40 and 0.04 are fp32 denormals: both kernels are at the mercy of the denormal floor; the scaled kernel must
        # not be worse than a few ulps of the smallest normal
        assert float(err.max()) <= max(4.0 * float(err32.max()), 2.0 ** -126)
        return
    worst = float((err / torch.maximum(bound, 2.0 * err32 + 1e-45)).max())
    print('%s x %s: max err / bound %.3f; max-abs rel %.2e (f32 MFMA %.2e)' % (xkind, wkind, worst, rel(got, ref), rel(got32, ref)))
    assert worst <= 1.0
    assert rel(got, ref) < 2.0 * rel(got32, ref) + 1e-7


@pytest.mark.parametrize('bad', [float('inf'), float('-inf'), float('nan')])
def test_non_finite_operands_propagate(ops, bad):
    '''One Inf / NaN element in the input (and, separately, in the weights) of a two-plane convolution.  The tensor's maximum is taken
    over its FINITE elements (rcf_abs_finite), so (1) every output outside the element's receptive field is what it is without the
    element -- held to the exact tier's bar against fp64 -- and (2) every output inside it is non-finite, as in an fp32 convolution
    (there: +-Inf or NaN; here always NaN -- the second plane of an Inf is Inf - Inf).  Nothing is silently flushed or dropped.'''
    n, c, co, h, w = 1, 64, 64, 20, 36
    x = rnd(n, c, h, w, seed=31)
    wt = rnd(co, c, 3, 3, seed=32, scale=1.0 / 24.0)
    iy, ix, ic = 7, 11, 5
    ops.set_precision('f16x2')
    try:
        d = ops.make_fwd_desc(n, h, w, c, 0, co, 3, 1, h, w, 0)
        # (a) the element in the input
        xb = x.clone()
        xb[0, ic, iy, ix] = bad
        amax = dev_amax(ops, xb)
        assert float(amax) == float(xb[torch.isfinite(xb)].abs().max())   # the maximum of the finite elements
        got, _, info = _run_fwd(ops, d, xb, None, wt, amax, dev_amax(ops, wt))
        assert 40000 <= info.kernel_id < 50000
        x0 = x.clone()
        x0[0, ic, iy, ix] = 0.0
        ref = F.conv2d(x0.double(), wt.double(), padding=1)
        rf = torch.zeros(1, 1, h, w, dtype=torch.bool)
        rf[0, 0, iy - 1:iy + 2, ix - 1:ix + 2] = True
        rf = rf.expand(n, co, h, w)
        assert not bool(torch.isfinite(got[rf]).any()), 'an output inside the receptive field stayed finite'
        assert bool(torch.isfinite(got[~rf]).all()), 'the element leaked outside its receptive field'
        assert float((got.double() - ref)[~rf].abs().max()) < EXACT_TOL * float(ref.abs().max())
        # (b) the element in the weights: every pixel of that output channel, nothing else
        wb = wt.clone()
        wb[9, 3, 1, 2] = bad
        got, _, _ = _run_fwd(ops, d, x, None, wb, dev_amax(ops, x), dev_amax(ops, wb))
        ref = F.conv2d(x.double(), wt.double(), padding=1)
        ch = torch.zeros(1, co, 1, 1, dtype=torch.bool)
        ch[0, 9] = True
        ch = ch.expand(n, co, h, w)
        inner = torch.zeros(n, co, h, w, dtype=torch.bool)
        inner[:, :, 1:-1, 1:-1] = True                     # (the border pixels of the channel miss the tap where padding sits under it)
        assert not bool(torch.isfinite(got[ch & inner]).any())
        assert bool(torch.isfinite(got[~ch]).all())
        assert float((got.double() - ref)[~ch].abs().max()) < EXACT_TOL * float(ref.abs().max())
    finally:
        ops.set_precision('fp32')


def test_unscaled_call_equals_scale_one_and_amax_entry_points(ops):
    '''rcf_amax accumulates a maximum (several tensors into one slot, misaligned and odd-sized inputs), rcf_amax_batch equals n single
    calls, the _amax variants of the elementwise kernels report the maximum of what they wrote -- and a conv called without maxima
    (null pointers = scale 1) equals the scaled call when the data already sits where the scale would put it.'''
    import ctypes
    from rcf_amd._lib import AmaxItem
    g = torch.Generator().manual_seed(9)
    ts = [torch.randn(n, generator=g).cuda() * sc for n, sc in ((1, 3.0), (7, 1e-3), (16384, 1.0), (16385, 2.0), (100003, 0.5), (5, 0.0))]
    for t in ts + [ts[4][1:]]:   # the last one is 4-byte aligned only
        assert float(ops.amax(t)) == float(t.abs().max())
    slot = torch.zeros(1, device='cuda')
    for t in ts:
        ops.amax(t, slot, accumulate=True)
    assert float(slot) == max(float(t.abs().max()) for t in ts)
    items = (AmaxItem * len(ts))()
    outs = torch.zeros(len(ts), device='cuda')
    for i, t in enumerate(ts):
        items[i].x, items[i].n, items[i].amax = t.data_ptr(), t.numel(), outs[i:i + 1].data_ptr()
    ops.amax_batch(items, len(ts))
    assert outs.tolist() == [float(t.abs().max()) for t in ts]
    # elementwise producers
    from rcf_amd._lib import RCF_ACT_LEAKY_RELU
    n_pix, c = 5 * 13 * 17, 32
    z = torch.randn(n_pix, c, generator=g).cuda() * 3
    res = torch.randn(n_pix, c, generator=g).cuda()
    coef = torch.stack([torch.rand(c, generator=g) + 0.5, torch.randn(c, generator=g), torch.randn(c, generator=g), torch.rand(c, generator=g) + 0.5]).cuda().contiguous()
    for r in (None, res):
        out, out2, am = torch.empty_like(z), torch.empty_like(z), torch.zeros(1, device='cuda')
        ops.bn_act_fwd(z, coef, r, out, n_pix, c, RCF_ACT_LEAKY_RELU, amax=am)
        ops.bn_act_fwd(z, coef, r, out2, n_pix, c, RCF_ACT_LEAKY_RELU)
        assert torch.equal(out, out2) and float(am) == float(out.abs().max())
    bcoef = torch.randn(2, c, generator=g).cuda() * 0.1
    dout = torch.randn(n_pix, c, generator=g).cuda() * 1e-3
    dz, dz2, am = torch.empty_like(z), torch.empty_like(z), torch.zeros(1, device='cuda')
    ops.bn_act_bwd_apply(dout, z, coef, None, bcoef, dz, None, False, n_pix, c, RCF_ACT_LEAKY_RELU, False, amax=am)
    ops.bn_act_bwd_apply(dout, z, coef, None, bcoef, dz2, None, False, n_pix, c, RCF_ACT_LEAKY_RELU, False)
    assert torch.equal(dz, dz2) and float(am) == float(dz.abs().max())
    zp, img = torch.randn(n_pix, c, generator=g).cuda(), torch.randn(n_pix, c, generator=g).cuda()
    fo, fo2, am = torch.empty_like(z), torch.empty_like(z), torch.zeros(1, device='cuda')
    ops.fuse_fwd(z, coef, zp, coef, img, fo, n_pix, c, amax=am)
    ops.fuse_fwd(z, coef, zp, coef, img, fo2, n_pix, c)
    assert torch.equal(fo, fo2) and float(am) == float(fo.abs().max())
    # scale 1: data with max in [2^14, 2^15) for x and w -> the scaled and the unscaled call are the same arithmetic
    n, cc, h, w = 1, 32, 12, 20
    x = rnd(n, cc, h, w, seed=3) * 2.0 ** 14.5
    wt = rnd(cc, cc, 3, 3, seed=4) * 2.0 ** 14.5
    ops.set_precision('f16x2')
    try:
        d = ops.make_fwd_desc(n, h, w, cc, 0, cc, 3, 1, h, w, 0)
        a, _, _ = _run_fwd(ops, d, x, None, wt, dev_amax(ops, x), dev_amax(ops, wt))
        b, _, _ = _run_fwd(ops, d, x, None, wt)
        assert torch.equal(a, b)
    finally:
        ops.set_precision('fp32')


def test_precision_levels_are_distinct_and_ordered(ops):
    '''One 64 -> 64 layer under the arithmetic levels of fp32 tensors: exact three-plane split ~ two fp16 planes (both fp32-class)
    << 'bf16_operands', i.e. each descriptor really selects its own kernels.'''
    case = (3, 1, 64, 0, 64, 2, 33, 64, None)
    k, s, c1, c2, co, n, h, w, up = case
    x1, _, wt, xin = _case(case, 5)
    ref = F.conv2d(xin.double(), wt.double(), padding=1)
    err, outs = {}, {}
    try:
        for mode in ('fp32', 'f16x2', 'bf16_operands'):
            ops.set_precision(mode)
            d = ops.make_fwd_desc(n, h, w, c1, c2, co, k, s, h, w, 0)
            f16 = mode == 'f16x2'
            outs[mode], _, _ = _run_fwd(ops, d, x1, None, wt, dev_amax(ops, xin) if f16 else None, dev_amax(ops, wt) if f16 else None)
            err[mode] = rel(outs[mode], ref)
    finally:
        ops.set_precision('fp32')
    print(err)
    assert err['fp32'] < 1e-6 and err['f16x2'] < 1e-6
    assert not torch.equal(outs['fp32'], outs['f16x2'])
    assert 100 * err['f16x2'] < err['bf16_operands'] < 1e-2


@pytest.mark.parametrize('cin,cout,n,hs,ws', [(64, 32, 1, 35, 51), (64, 64, 2, 12, 20), (128, 64, 2, 15, 25), (32, 64, 1, 9, 70), (256, 128, 1, 8, 13),
                                              (16, 32, 2, 40, 18), (64, 32, 2, 64, 96)])
def test_up2x_conv_as_four_phase_convs(ops, cin, cout, n, hs, ws):
    '''The exact-2x UpConv (nearest upsample + 3x3) as four 2x2 phase convolutions writing the strided output (RCF_PHASE_UP2X_FWD);
    the four phases' pre-summed weights share one maximum.  The one-launch form (conv_split_kernel<SplitCfg<2, ., 32, ., 2, 1, true>>:
    x staged and split once per channel chunk, four accumulator sets) is bitwise the four launches.'''
    x = rnd(n, cin, hs, ws, seed=3)
    wt = rnd(cout, cin, 3, 3, seed=4, scale=1.0 / np.sqrt(cin * 9))
    ref = F.conv2d(F.interpolate(x.double(), scale_factor=2), wt.double(), padding=1)
    ops.set_precision('f16x2')
    try:
        out = torch.full((n, 2 * hs, 2 * ws, cout), float('nan'), device='cuda')
        from rcf_amd._lib import RCF_PHASE_UP2X_FWD
        wph = ops.phase_weights(wt.cuda(), RCF_PHASE_UP2X_FWD)
        ax, aw = dev_amax(ops, x), ops.amax(wph)
        for a in range(2):
            for b in range(2):
                d = ops.make_up2x_fwd_desc(n, hs, ws, cin, cout, a, b)
                info = ops.conv_query(d)
                assert 40000 <= info.kernel_id < 50000
                packed = torch.empty(info.packed_weight_floats, device='cuda')
                ops.conv_pack(d, wph[a * 2 + b], packed, aw)
                ops.conv_fwd(d, nhwc(x), None, packed, out, None, scales=ops.make_scales(ax, None, aw))
        assert rel(nchw(out), ref) < EXACT_TOL
        # the same four phases in ONE launch (rcf_conv_desc.phase_sum == 2): bitwise the four launches' output, and the BatchNorm
        # statistics of all four phases in one set of partial rows
        dm = ops.make_up2x_fwd_desc(n, hs, ws, cin, cout, 0, 0, phase_out=True)
        im = ops.conv_query(dm)
        assert 40000 <= im.kernel_id < 50000
        pm = torch.empty(4 * im.packed_weight_floats, device='cuda')
        for ph in range(4):
            ops.conv_pack(dm, wph[ph], pm[ph * im.packed_weight_floats:(ph + 1) * im.packed_weight_floats], aw)
        out1 = torch.full((n, 2 * hs, 2 * ws, cout), float('nan'), device='cuda')
        part = torch.full((im.n_partials, 2, cout), float('nan'), device='cuda', dtype=torch.float64)
        ops.conv_fwd(dm, nhwc(x), None, pm, out1, part, scales=ops.make_scales(ax, None, aw))
        torch.cuda.synchronize()
        assert torch.equal(out1, out)
        st = part.sum(0).cpu()
        np.testing.assert_allclose(st[0].numpy(), nchw(out).double().sum((0, 2, 3)).numpy(), rtol=1e-9, atol=1e-6)
        np.testing.assert_allclose(st[1].numpy(), (nchw(out).double() ** 2).sum((0, 2, 3)).numpy(), rtol=1e-9, atol=1e-6)
    finally:
        ops.set_precision('fp32')


def _bn_coef(c, seed):
    '''[4][c]: scale, shift, mean, invstd of a BatchNorm layer (rcf_bn_finalize's coefficient rows), with both signs of scale.'''
    g = torch.Generator().manual_seed(seed)
    gamma = torch.rand(c, generator=g) * 2 - 1
    beta = torch.rand(c, generator=g) - 0.5
    mean = torch.rand(c, generator=g) - 0.5
    invstd = 0.5 + 2 * torch.rand(c, generator=g)
    return torch.stack([gamma * invstd, beta - mean * gamma * invstd, mean, invstd]).contiguous().cuda()


# (ksize-kind, c_in of the forward conv = channels of dx, c_out of the forward conv, n, h, w)
SUMS_CASES = [('3x3', 64, 64, 2, 33, 64), ('3x3', 32, 32, 1, 70, 102), ('3x3', 32, 64, 2, 17, 40), ('3x3', 256, 256, 2, 8, 13),
              ('3x3', 128, 64, 1, 29, 50), ('up2x', 64, 32, 1, 35, 51), ('up2x', 64, 64, 2, 12, 20)]


@pytest.mark.parametrize('case', SUMS_CASES, ids=[str(c) for c in SUMS_CASES])
def test_input_gradient_kernel_takes_the_batchnorm_backward_sums(ops, case):
    '''rcf_conv2d_dgrad_bn_sums: the input gradient dx of a convolution whose input was the activation of a BatchNorm + LeakyReLU
    block, written together with that block's backward sums -- against the two-launch form (rcf_conv2d_fwd_scaled, then
    rcf_bn_act_bwd_reduce over dx and z): dx bitwise the same, the sums (fp64, another order) and everything rcf_bn_bwd_finalize
    derives from them equal to fp64 rounding; plus an fp64 host evaluation of the sums.'''
    from rcf_amd._lib import RCF_ACT_LEAKY_RELU, RCF_PHASE_UP2X_DGRAD
    kind, ci, co, n, h, w = case
    ops.set_precision('f16x2')
    try:
        wt = rnd(co, ci, 3, 3, seed=11, scale=1.0 / np.sqrt(ci * 9)).cuda()
        if kind == '3x3':
            d = ops.make_fwd_desc(n, h, w, ci, 0, co, 3, 1)
            dd = ops.make_dgrad_desc(d, 0, ci, False)
            dz = nhwc(rnd(n, co, h, w, seed=12))
            wsrc = [wt]
        else:   # the four phases of the up-2x input gradient summed in one launch
            dd = ops.make_up2x_dgrad_desc(n, h, w, ci, co, 0, 0, False, phase_sum=True)
            dz = nhwc(rnd(n, co, 2 * h, 2 * w, seed=12))
            wph = ops.phase_weights(wt, RCF_PHASE_UP2X_DGRAD)
            wsrc = [wph[ph] for ph in range(4)]
        info = ops.conv_query(dd)
        assert info.bn_bwd_sums == 1 and 40000 <= info.kernel_id < 50000
        aw = ops.amax(wt if kind == '3x3' else wph)
        adz = ops.amax(dz)
        packed = torch.empty(len(wsrc) * info.packed_weight_floats, device='cuda')
        for i, wsl in enumerate(wsrc):
            ops.conv_pack(dd, wsl, packed[i * info.packed_weight_floats:(i + 1) * info.packed_weight_floats], aw)
        scales = ops.make_scales(adz, None, aw)
        z = nhwc(rnd(n, ci, h, w, seed=13))
        coef = _bn_coef(ci, 14)
        n_pix = n * h * w
        # two launches
        dx_ref = torch.full((n, h, w, ci), float('nan'), device='cuda')
        ops.conv_fwd(dd, dz, None, packed, dx_ref, None, scales=scales)
        nb = ops.ew_blocks(n_pix, ci)
        part_ref = torch.empty((nb, 2, ci), dtype=torch.float64, device='cuda')
        ops.bn_act_bwd_reduce(dx_ref, z, coef, None, part_ref, n_pix, ci, RCF_ACT_LEAKY_RELU, False)
        # one launch
        dx = torch.full((n, h, w, ci), float('nan'), device='cuda')
        part = torch.full((info.n_partials, 2, ci), float('nan'), dtype=torch.float64, device='cuda')
        ops.conv_dgrad_bn_sums(dd, dz, packed, dx, z, coef, part, scales)
        torch.cuda.synchronize()
        assert torch.equal(dx, dx_ref)
        s_ref, s_new = part_ref.sum(0), part.sum(0)
        # fp64 host evaluation of the two sums from dx
        g = dx.double() * torch.where(z * coef[0] + coef[1] > 0, 1.0, 0.2)   # src/net_utils.py:16 negative_slope 0.20; the sign test in fp32 like the kernels'
        xh = (z.double() - coef[2].double()) * coef[3].double()
        want = torch.stack([g.sum((0, 1, 2)), (g * xh).sum((0, 1, 2))])
        mag = torch.stack([g.abs().sum((0, 1, 2)), (g * xh).abs().sum((0, 1, 2))])
        assert float(((s_new - want).abs() / mag).max()) < 1e-6      # y and xhat are formed in fp32 by both kernels
        assert float(((s_new - s_ref).abs() / mag).max()) < 2e-7     # the reduce pass rounds xhat to fp32 per term, the epilogue does not
        outs = []
        for pt, rows in ((part_ref, nb), (part, info.n_partials)):
            bcoef = torch.empty((2, ci), device='cuda')
            dgam, dbet = torch.empty(ci, device='cuda'), torch.empty(ci, device='cuda')
            ops.bn_bwd_finalize(pt, rows, 2 * ci, ci, n_pix, bcoef, dgam, dbet)
            outs.append((bcoef, dgam, dbet))
        for a_, b_ in zip(outs[0], outs[1]):
            assert float((a_ - b_).abs().max()) <= 1e-6 * float(b_.abs().max())
    finally:
        ops.set_precision('fp32')


def test_batchnorm_backward_sums_entry_point_rejects_what_it_cannot_do(ops):
    '''bn_bwd_sums is 0 (and the call answers RCF_EUNSUPPORTED) for strided phase outputs, concat sources, other arithmetic.'''
    from rcf_amd import _lib
    ops.set_precision('f16x2')
    try:
        d = ops.make_fwd_desc(1, 16, 24, 32, 0, 64, 3, 2)
        s2 = ops.make_s2_dgrad_desc(d, 0, 1, False)                  # one strided phase of a stride-2 input gradient
        assert ops.conv_query(s2).bn_bwd_sums == 0
        cat = ops.make_fwd_desc(1, 16, 24, 32, 32, 64, 3, 1)         # two sources
        assert ops.conv_query(cat).bn_bwd_sums == 0
        pw = ops.make_dgrad_desc(ops.make_fwd_desc(1, 16, 24, 32, 0, 64, 1, 1), 0, 32, False)   # 1x1: not a split kernel
        assert ops.conv_query(pw).bn_bwd_sums == 0
    finally:
        ops.set_precision('fp32')
    ops.set_precision('fp32')   # ops level: the three-plane arithmetic
    try:
        d3 = ops.make_dgrad_desc(ops.make_fwd_desc(1, 16, 24, 32, 0, 64, 3, 1), 0, 32, False)
        assert ops.conv_query(d3).bn_bwd_sums == 0
        info = ops.conv_query(d3)
        t = torch.zeros(1, 16, 24, 64, device='cuda')
        dx = torch.zeros(1, 16, 24, 32, device='cuda')
        with pytest.raises(_lib.RcfError):
            ops.conv_dgrad_bn_sums(d3, t, torch.zeros(info.packed_weight_floats, device='cuda'), dx, dx, torch.zeros(4, 32, device='cuda'),
                                   torch.zeros(info.n_partials, 2, 32, dtype=torch.float64, device='cuda'), ops.make_scales(None, None, None))
    finally:
        ops.set_precision('fp32')


@pytest.mark.parametrize('c,co,n,h,w', [(3, 32, 2, 70, 102), (2, 16, 1, 45, 81), (3, 32, 1, 224, 384)])
def test_stem_7x7_stride2_as_4x4_on_the_fp32_space_to_depth_image(ops, c, co, n, h, w):
    '''The stems of the fp32 configuration (src/networks.py:332-345: 7x7, stride 2, pad 3, 3 / 2 input channels) as a 4x4 stride-1
    convolution on the fp32 space-to-depth image, two scaled fp16 planes: rcf_s2d_image_f32 (with the image's maximum),
    rcf_stem_weights_s2d, rcf_conv2d_fwd_scaled with ksize 4 -- against the fp64 7x7 convolution at the exact tier's bar, odd sizes
    included, and the BatchNorm statistics of the written values.'''
    x = rnd(n, c, h, w, seed=21) * 0.5 + 0.5            # images live in [0, 1]
    if c == 2:
        x = x * 80.0                                     # the radar depth / response channels reach tens of metres
    wt = rnd(co, c, 7, 7, seed=22, scale=1.0 / np.sqrt(c * 49))
    ref = F.conv2d(x.double(), wt.double(), stride=2, padding=3)
    s2d, amax = ops.s2d_image_f32(x.cuda())
    assert float(amax) == float(x.abs().max())
    assert tuple(s2d.shape) == (n, (h + 1) // 2, (w + 1) // 2, 16)
    d = ops.make_stem_s2d_desc(n, h, w, co, f32=True)
    info = ops.conv_query(d)
    assert 40000 <= info.kernel_id < 60000
    w4 = ops.stem_weights_s2d(wt.cuda())
    wmax = ops.amax(w4)
    packed = torch.empty(info.packed_weight_floats, device='cuda')
    ops.conv_pack(d, w4, packed, wmax)
    out = torch.full((n, d.h_out, d.w_out, co), float('nan'), device='cuda')
    part = torch.full((info.n_partials, 2, co), float('nan'), device='cuda', dtype=torch.float64)
    ops.conv_fwd(d, s2d, None, packed, out, part, scales=ops.make_scales(amax, None, wmax))
    torch.cuda.synchronize()
    got = nchw(out)
    assert tuple(got.shape) == tuple(ref.shape)
    e = rel(got, ref)
    print('stem %d -> %d at %dx%d on two fp16 planes: rel %.2e' % (c, co, h, w, e))
    assert e < EXACT_TOL
    st = part.sum(0).cpu()
    np.testing.assert_allclose(st[0].numpy(), got.double().sum((0, 2, 3)).numpy(), rtol=1e-9, atol=1e-6)
    np.testing.assert_allclose(st[1].numpy(), (got.double() ** 2).sum((0, 2, 3)).numpy(), rtol=1e-9, atol=1e-6)


def _named(model):
    out = []
    for prefix, mod in (('encoder.', model.encoder), ('decoder.', model.decoder)):
        out += [(prefix + k, v) for k, v in mod.named_parameters()]
    return out


@pytest.fixture(scope='module')
def env():
    import rcf_amd  # noqa: F401
    from rcf_amd import synth, train
    if os.environ.get('RCF_CONV_SPLIT') == '0':
        pytest.skip('the two-plane arithmetic lives in the split kernels, which RCF_CONV_SPLIT=0 turns off')
    return synth, train


def _step(env, cfg, g, mode, deconv_type='up'):
    synth, train = env
    n, h, w, k, dseed, wseed = [int(v) for v in g['meta']]
    m = train.build_model(cfg, device='cuda', deconv_type=deconv_type)
    synth.fill_state_dict_([m.encoder, m.decoder], wseed)
    m.compute_dtype = mode
    m.train()
    b = {kk: v.cuda() for kk, v in synth.make_batch(n, h, w, k, seed=dseed).items()}
    out = m.forward(image=b['image'], input_depth=b['input_depth'])
    loss, info = m.compute_loss(image=b['image'], output_depth=out, ground_truth=b['ground_truth'], lidar_map=b['lidar_map'],
                                loss_func='l1', w_smoothness=0.0, loss_smoothness_kernel_size=-1,
                                validity_map_loss_smoothness=None, w_lidar_loss=2.0)
    loss.backward()
    torch.cuda.synchronize()
    return m, out.detach(), float(loss.detach())


@pytest.mark.parametrize('fixture,deconv', [('T1_published_train.npz', 'up'), ('T10_transpose_published_train.npz', 'transpose')])
def test_published_net_train_step_against_the_reference_fixture(env, golden_dir, fixture, deconv):
    '''One training step of the published FusionNet under compute_dtype='f16x2' against the fixture generated by the REAL reference
    (fp32 PyTorch CPU): output and loss within north_star's 1e-3 (measured: ~1e-6), parameter-gradient norms within 1 %, and the
    mode is really in use (the output differs from the three-plane path's).'''
    synth, _ = env
    g = np.load(os.path.join(golden_dir, fixture))
    m, out, loss = _step(env, synth.PUBLISHED, g, 'f16x2', deconv)
    _, out32, _ = _step(env, synth.PUBLISHED, g, 'fp32_3plane', deconv)
    e = rel(out.cpu(), torch.as_tensor(g['output']))
    e32 = rel(out32.cpu(), torch.as_tensor(g['output']))
    print('output rel err vs the reference: f16x2 %.2e (three-plane path %.2e); loss %.6f ref %.6f' % (e, e32, loss, float(g['loss'][0])))
    assert e < NORTH_STAR and e < 2e-5
    assert not torch.equal(out, out32)
    assert abs(loss - float(g['loss'][0])) < 1e-4 * abs(float(g['loss'][0]))
    grads = dict(_named(m))
    worst = 0.0
    for key, l2 in zip(g['grad_keys'].tolist(), g['grad_l2'].tolist()):
        got = float(grads[key].grad.double().norm())
        worst = max(worst, abs(got - l2) / max(l2, 1e-6))
    print('worst parameter-gradient norm deviation: %.2e' % worst)
    assert worst < 1e-2


@pytest.mark.parametrize('tier', ['fp32', 'fp32_3plane'])
def test_published_net_gradients_element_by_element_against_the_reference_samples(env, golden_dir, tier):
    '''Fixture T1b (tests/golden/make_golden.py): 2048 seeded elements of each of the ten largest gradient tensors of fixture T1's
    step, from the REAL reference in fp32, next to the same elements of the fp64 run.  The reference's own fp32 values are 1.5e-3 ..
    1.9e-2 (max-norm) from fp64 on these tensors -- LeakyReLU / max-pool decisions that differ between any two fp32 evaluations --
    so the HIP elements are held to fp64 with the reference's own distance as the yardstick: per tensor within 5x, median over the
    ten within 3x (the rule of tests/test_hip_model.py::_check_gradients_against_fp64), and to the reference's fp32 values themselves
    within the sum of both distances.'''
    synth, train = env
    s = np.load(os.path.join(golden_dir, 'T1b_published_grad_samples.npz'))
    n, h, w, k, dseed, wseed = [int(v) for v in s['meta']]
    m = train.build_model(synth.PUBLISHED, device='cuda')
    synth.fill_state_dict_([m.encoder, m.decoder], wseed)
    m.compute_dtype = tier
    m.train()
    b = {kk: v.cuda() for kk, v in synth.make_batch(n, h, w, k, seed=dseed).items()}
    out = m.forward(image=b['image'], input_depth=b['input_depth'])
    loss, _ = m.compute_loss(image=b['image'], output_depth=out, ground_truth=b['ground_truth'], lidar_map=b['lidar_map'],
                             loss_func='l1', w_smoothness=0.0, loss_smoothness_kernel_size=-1,
                             validity_map_loss_smoothness=None, w_lidar_loss=2.0)
    loss.backward()
    torch.cuda.synchronize()
    grads = {kk: p.grad for kk, p in _named(m) if p.grad is not None}
    e_hip, e_ref = [], []
    for key, idx, ref32, v64, amax in zip(s['keys'].tolist(), s['idx'], s['ref32'], s['fp64'], s['fp64_absmax']):
        got = grads[key].detach().reshape(-1).cpu().double().numpy()[idx]
        eh, er = float(np.abs(got - v64).max() / amax), float(np.abs(ref32.astype(np.float64) - v64).max() / amax)
        e_hip.append(eh)
        e_ref.append(er)
        assert eh <= 5.0 * er + 2e-4, (key, eh, er)
        assert float(np.abs(got - ref32).max() / amax) <= eh + er + 1e-6
    print('%s: ten largest gradient tensors, sampled elements vs fp64 (max-norm): HIP %s | the reference in fp32 %s'
          % (tier, ' '.join('%.1e' % e for e in e_hip), ' '.join('%.1e' % e for e in e_ref)))
    assert np.median(e_hip) <= 3.0 * np.median(e_ref) + 2e-5


@pytest.mark.parametrize('deconv', ['up', 'transpose'])
def test_batchnorm_sums_from_the_input_gradient_kernels_in_the_published_net(env, golden_dir, deconv):
    '''The published net's training step with the BatchNorm-backward sums taken by the input-gradient kernels (default) against the
    same step with the separate reduction pass (Engine.bn_sums_in_dgrad = False): the path is really in use (one launch per eligible
    block), outputs and loss bitwise the same (the forward does not change), every parameter gradient equal to fp64 rounding of the
    sums -- and both hold the reference fixture's gradient norms.'''
    synth, train = env
    g = np.load(os.path.join(golden_dir, 'T1_published_train.npz' if deconv == 'up' else 'T10_transpose_published_train.npz'))
    n, h, w, k, dseed, wseed = [int(v) for v in g['meta']]
    runs = {}
    for fused in (True, False):
        m = train.build_model(synth.PUBLISHED, device='cuda', deconv_type=deconv)
        synth.fill_state_dict_([m.encoder, m.decoder], wseed)
        m.train()
        m._engine.bn_sums_in_dgrad = fused
        b = {kk: v.cuda() for kk, v in synth.make_batch(n, h, w, k, seed=dseed).items()}
        out = m.forward(image=b['image'], input_depth=b['input_depth'])
        loss, _ = m.compute_loss(image=b['image'], output_depth=out, ground_truth=b['ground_truth'], lidar_map=b['lidar_map'],
                                 loss_func='l1', w_smoothness=0.0, loss_smoothness_kernel_size=-1,
                                 validity_map_loss_smoothness=None, w_lidar_loss=2.0)
        loss.backward()
        torch.cuda.synchronize()
        runs[fused] = (out.detach().clone(), float(loss.detach()), {kk: p.grad.detach().clone() for kk, p in _named(m) if p.grad is not None},
                       m._engine.bn_sums_taken)
    print('launches that took the sums: %d (published net, deconv_type=%s)' % (runs[True][3], deconv))
    assert runs[False][3] == 0
    assert runs[True][3] >= (16 if deconv == 'up' else 8)   # conv1 of 16 ResNet blocks (+ the up-conv decoder's deconv / conv layers)
    assert torch.equal(runs[True][0], runs[False][0]) and runs[True][1] == runs[False][1]
    worst = max(rel(runs[True][2][kk], runs[False][2][kk]) for kk in runs[False][2])
    print('worst parameter-gradient difference between the two forms: %.2e' % worst)
    assert worst < 2e-5
    for key, l2 in zip(g['grad_keys'].tolist(), g['grad_l2'].tolist()):
        got = float(runs[True][2][key].double().norm())
        assert abs(got - l2) < 1e-2 * max(l2, 1e-6)


HOSTILE_SEEDS = {'outlier_pixels': (321, 322, 323, 324), 'tiny_inputs': (321, 322, 323, 324, 325, 326, 327, 328), 'huge_inputs': (321, 322, 323, 324)}
FLIPPED = 1e-3   # a tensor this far from fp64 carries a LeakyReLU / max-pool decision that differs from the fp64 run's (bimodal: see below)


def _hostile_batch(synth, kind, seed):
    cb = synth.make_batch(2, 70, 102, 8, seed=seed)
    if kind == 'outlier_pixels':
        cb['image'][0, 1, 10, 17] = 1.0e4
        cb['image'][1, 0, 40, 3] = -3.0e3
        cb['input_depth'][0, 0, 22, 50] = 5.0e5
    elif kind == 'tiny_inputs':
        cb['image'] *= 1e-6
        cb['input_depth'] *= 1e-6
    else:
        cb['image'] *= 1e4
        cb['input_depth'] *= 1e4
    return cb


def _hostile_oracle(synth, cb, dtype):
    from oracle.fusionnet_oracle import FusionNetOracle
    o = FusionNetOracle(**synth.TINY)
    synth.fill_state_dict_([o.encoder, o.decoder], 17)
    for mod in (o.encoder, o.decoder):
        mod.to(dtype)
    o.train()
    r = o.forward(cb['image'].to(dtype), cb['input_depth'].to(dtype))
    l = o.compute_loss(r, cb['ground_truth'].to(dtype), cb['lidar_map'].to(dtype), 2.0)[0]
    l.backward()
    return r.detach(), float(l.detach()), {k: p.grad.double() for k, p in _named(o) if p.grad is not None}


def _hostile_hip(synth, train, cb, tier):
    m = train.build_model(synth.TINY, device='cuda')
    synth.fill_state_dict_([m.encoder, m.decoder], 17)
    m.compute_dtype = tier
    m.train()
    b = {kk: v.cuda() for kk, v in cb.items()}
    out = m.forward(image=b['image'], input_depth=b['input_depth'])
    loss, _ = m.compute_loss(image=b['image'], output_depth=out, ground_truth=b['ground_truth'], lidar_map=b['lidar_map'],
                             loss_func='l1', w_smoothness=0.0, loss_smoothness_kernel_size=-1,
                             validity_map_loss_smoothness=None, w_lidar_loss=2.0)
    loss.backward()
    torch.cuda.synchronize()
    return out.detach().cpu(), float(loss.detach()), {k: p.grad.detach().cpu().clone() for k, p in _named(m) if p.grad is not None}


@pytest.mark.parametrize('kind', ['outlier_pixels', 'tiny_inputs', 'huge_inputs'])
def test_training_step_with_hostile_inputs_against_the_oracle(env, kind):
    '''The whole step (tiny net, train mode) on inputs that stress the per-tensor scales end to end: a few image pixels and radar
    depths 10^4 times the rest, all inputs scaled by 1e-6, all inputs scaled by 1e+4.  Output and loss against the fp32 CPU oracle at
    north_star's bar on every seed, for both operand arithmetics.  Parameter gradients against the fp64 oracle with the fp32 CPU
    oracle's own distance from fp64 as the yardstick -- for EVERY kind, no tier-against-tier stand-in -- judged over a seed sweep:

    round 3 exempted 'tiny_inputs' at its one seed (321: HIP 3.6e-2 from fp64, CPU fp32 4e-4) and blamed the BatchNorm coefficient
    rows.  Round 4 isolated it (tools/diag_net.py RCF_DIAG_BLOCK, tools/diag_bn_bwd.py, profiles/r04_tiny_inputs_diagnosis.txt): every
    BatchNorm-backward stage and every input-gradient kernel of that step equals an fp64 evaluation of ITS OWN inputs to ~1e-7
    (test_batchnorm_backward_stages_on_the_hostile_step below), |mean| * invstd is O(1) in all 53 blocks -- and ONE element of
    decoder.deconv0.deconv's BatchNorm output (of 57,120) lies within the forward pass's round-off of zero and takes the other
    LeakyReLU branch than in the fp64 run.  The case is chaotic for ANY fp32 implementation: the stock-PyTorch fp32 oracle is 1e-3 ..
    8.7e-2 from its own fp64 run on 14 of the 20 data seeds 321..340 (8.7e-2 at 322, 8.3e-2 at 324; 4 of 20 even with unscaled inputs).
    A seed whose worst tensor is > 1e-3 from fp64 carries such a flipped decision; the distribution is bimodal (<= 4e-4 | >= 1e-3).'''
    synth, train = env
    seeds = HOSTILE_SEEDS[kind]
    rows = []
    for si, seed in enumerate(seeds):
        cb = _hostile_batch(synth, kind, seed)
        ref, rl, g32 = _hostile_oracle(synth, cb, torch.float32)
        _, _, g64 = _hostile_oracle(synth, cb, torch.float64)
        e_cpu = {k: rel(g32[k], g64[k]) for k in g64}
        res = {tier: _hostile_hip(synth, train, cb, tier) for tier in (('fp32', 'fp32_3plane') if si == 0 else ('fp32',))}
        for tier, (out, loss, grads) in res.items():
            assert bool(torch.isfinite(out).all())
            assert rel(out, ref) < NORTH_STAR and abs(loss - rl) < NORTH_STAR * rl, (kind, seed, tier)
        e_hip = {k: rel(res['fp32'][2][k], g64[k]) for k in g64}
        row = dict(seed=seed, hip_worst=max(e_hip.values()), cpu_worst=max(e_cpu.values()),
                   hip_med=float(np.median(list(e_hip.values()))), cpu_med=float(np.median(list(e_cpu.values()))), e_hip=e_hip, e_cpu=e_cpu)
        if si == 0:
            e3 = {k: rel(res['fp32_3plane'][2][k], g64[k]) for k in g64}
            row['hip3_worst'] = max(e3.values())
            # the two-plane tier is not further from the oracle than the three-plane tier by more than fp32 round-off allows
            assert rel(res['fp32'][0], ref) < 2.0 * rel(res['fp32_3plane'][0], ref) + 2e-5
        rows.append(row)
        print('%s seed %d: gradients vs fp64, worst tensor HIP %.2e (CPU fp32 oracle %.2e), median tensor HIP %.2e (CPU %.2e)%s'
              % (kind, seed, row['hip_worst'], row['cpu_worst'], row['hip_med'], row['cpu_med'],
                 '' if si else ', three-plane tier worst %.2e' % row['hip3_worst']))
    # (1) seeds on which neither run carries a flipped decision: every tensor at the CPU oracle's own distance from fp64
    calm = [r for r in rows if r['hip_worst'] < FLIPPED and r['cpu_worst'] < FLIPPED]
    for r in calm:
        for k, err in r['e_hip'].items():
            assert err < 3.0 * r['e_cpu'][k] + 1e-5, (kind, r['seed'], k, err, r['e_cpu'][k])
    # (2) the HIP path takes a differing decision no more often than the fp32 CPU oracle does (both are draws from the same chaos: the
    # counts are two small binomial samples), and a seed that carries one is moved by what the CPU oracle's own flipped seeds are moved
    # by (<= 8.7e-2 over seeds 321..340) -- not by more
    n_hip = sum(r['hip_worst'] >= FLIPPED for r in rows)
    n_cpu = sum(r['cpu_worst'] >= FLIPPED for r in rows)
    gm = lambda key: float(np.exp(np.mean([np.log(r[key] + 1e-12) for r in rows])))
    print('%s: seeds with a flipped decision HIP %d / CPU %d of %d; geometric mean of the worst tensor HIP %.2e / CPU %.2e, of the median '
          'tensor HIP %.2e / CPU %.2e; %d calm seeds checked tensor by tensor' % (kind, n_hip, n_cpu, len(rows), gm('hip_worst'), gm('cpu_worst'),
                                                                                 gm('hip_med'), gm('cpu_med'), len(calm)))
    assert n_hip <= n_cpu + max(2, len(rows) // 3)
    assert max(r['hip_worst'] for r in rows) < 0.2
    if kind != 'tiny_inputs':
        assert calm, 'outlier / huge inputs are not chaotic: at least one seed must be checked tensor by tensor'


def test_batchnorm_backward_stages_on_the_hostile_step(env):
    '''Every BatchNorm-backward launch of the 'tiny_inputs' step (seed 321: the step round 3 exempted), each STAGE against an fp64
    evaluation of the stage's own inputs: the per-channel sums (reduce + finalize -> mean g, mean g * xhat), the apply pass (dz), and
    the complete fp64 BatchNorm backward of the same (dout, z) with mean / invstd / xhat formed in fp64 -- i.e. the fp32 coefficient
    rows (scale, shift, mean, invstd) cost nothing measurable: conditioning |mean| * invstd stays O(1) even here.'''
    synth, train = env
    from rcf_amd import engine as eng_mod
    cb = _hostile_batch(synth, 'tiny_inputs', 321)
    calls = []
    orig = eng_mod.ops.bn_act_bwd_apply

    def wrapped(dout, z, coef, out, bcoef, dz, dres, dres_accumulate, n_pix, c, act, has_res, amax=None):
        orig(dout, z, coef, out, bcoef, dz, dres, dres_accumulate, n_pix, c, act, has_res, amax=amax)
        torch.cuda.synchronize()
        g = dout.double().reshape(-1, c)
        zz = z.double().reshape(-1, c)
        k = coef.double()
        if has_res:
            g = g * torch.where(out.double().reshape(-1, c) > 0, 1.0, 0.2)
        y32 = z.reshape(-1, c) * coef[0] + coef[1]                       # the kernels' own LeakyReLU decision
        gp = g * torch.where(y32.double() > 0, 1.0, 0.2)
        xh_k = (zz - k[2]) * k[3]
        e_sums = max(rel(bcoef[0], gp.mean(0)), rel(bcoef[1], (gp * xh_k).mean(0)))
        e_apply = rel(dz.reshape(-1, c), k[0] * (gp - bcoef[0].double() - xh_k * bcoef[1].double()))
        mean = zz.mean(0)
        invstd = 1.0 / torch.sqrt(((zz * zz).mean(0) - mean * mean).clamp_min(0) + 1e-5)
        xh = (zz - mean) * invstd
        e_ideal = rel(dz.reshape(-1, c), (k[0] / k[3]) * invstd * (gp - gp.mean(0) - xh * (gp * xh).mean(0)))
        calls.append((e_sums, e_apply, e_ideal, float((mean.abs() * invstd).max())))

    eng_mod.ops.bn_act_bwd_apply = wrapped
    try:
        _hostile_hip(synth, train, cb, 'fp32')
    finally:
        eng_mod.ops.bn_act_bwd_apply = orig
    assert len(calls) >= 50
    worst = [max(c[i] for c in calls) for i in range(4)]
    print('%d BatchNorm-backward launches: worst sums %.2e, apply %.2e, against the all-fp64 BatchNorm backward %.2e; max |mean| * invstd %.2f'
          % (len(calls), worst[0], worst[1], worst[2], worst[3]))
    assert worst[0] < 2e-6 and worst[1] < 2e-6 and worst[2] < 2e-6


def test_sparse_radar_channel_and_tiny_beta_against_the_oracle(env):
    '''The reference's real input statistics where a per-TENSOR scale is weakest: the radar depth / response channels are <= 1 % non-zero
    (a handful of returns of 1..80 m, zeros elsewhere: every tensor of the depth branch is mostly tiny values under one large
    maximum), and every BatchNorm beta is shrunk to 1e-4 of its draw, so nothing re-centres the activations away from zero.
    Published net, train mode, 2 x 96 x 160: output and loss against the fp32 CPU oracle at north_star's bar; parameter gradients
    against the fp64 oracle, per tensor, with the fp32 CPU oracle's own distance as the yardstick (median within 3x, whole-gradient
    relative L2 error within 3x) -- on both operand arithmetics.'''
    from oracle.fusionnet_oracle import FusionNetOracle
    synth, train = env
    cb = synth.make_batch(2, 96, 160, 8, seed=77)
    rs = np.random.RandomState(5)
    keep = torch.from_numpy(rs.rand(2, 1, 96, 160) < 0.008)
    depth = torch.from_numpy(rs.uniform(1.0, 80.0, size=(2, 1, 96, 160)).astype(np.float32)) * keep
    resp = torch.from_numpy(rs.uniform(32.0, 64.0, size=(2, 1, 96, 160)).astype(np.float32)) * keep
    cb['input_depth'] = torch.cat([depth, resp], 1)
    assert float((cb['input_depth'] != 0).float().mean()) < 0.01

    def shrink_beta(mods):
        with torch.no_grad():
            for mod in mods:
                for k, p in mod.named_parameters():
                    if k.endswith('batch_norm.bias'):
                        p.mul_(1e-4)

    def oracle(dtype):
        o = FusionNetOracle(**synth.PUBLISHED)
        synth.fill_state_dict_([o.encoder, o.decoder], 23)
        shrink_beta([o.encoder, o.decoder])
        for mod in (o.encoder, o.decoder):
            mod.to(dtype)
        o.train()
        r = o.forward(cb['image'].to(dtype), cb['input_depth'].to(dtype))
        l = o.compute_loss(r, cb['ground_truth'].to(dtype), cb['lidar_map'].to(dtype), 2.0)[0]
        l.backward()
        return r.detach(), float(l.detach()), {k: p.grad.double() for k, p in _named(o) if p.grad is not None}
    ref, rl, g32 = oracle(torch.float32)
    _, _, g64 = oracle(torch.float64)

    def l2(g):
        num = sum(float(((g[k].double().cpu() - g64[k]) ** 2).sum()) for k in g64)
        return (num / sum(float((g64[k] ** 2).sum()) for k in g64)) ** 0.5
    e_cpu = np.array([rel(g32[k], g64[k]) for k in g64])
    for tier in ('fp32', 'fp32_3plane'):
        m = train.build_model(synth.PUBLISHED, device='cuda')
        synth.fill_state_dict_([m.encoder, m.decoder], 23)
        shrink_beta([m.encoder, m.decoder])
        m.compute_dtype = tier
        m.train()
        b = {kk: v.cuda() for kk, v in cb.items()}
        out = m.forward(image=b['image'], input_depth=b['input_depth'])
        loss, _ = m.compute_loss(image=b['image'], output_depth=out, ground_truth=b['ground_truth'], lidar_map=b['lidar_map'],
                                 loss_func='l1', w_smoothness=0.0, loss_smoothness_kernel_size=-1,
                                 validity_map_loss_smoothness=None, w_lidar_loss=2.0)
        loss.backward()
        torch.cuda.synchronize()
        grads = {k: p.grad.detach().cpu() for k, p in _named(m) if p.grad is not None}
        e_hip = np.array([rel(grads[k], g64[k]) for k in g64])
        print('sparse radar channel, tiny beta, %s: output rel %.2e, loss rel %.2e; gradients vs fp64: median tensor HIP %.2e (CPU fp32 %.2e), '
              'worst HIP %.2e (CPU %.2e), whole-gradient L2 HIP %.2e (CPU %.2e)'
              % (tier, rel(out.detach().cpu(), ref), abs(float(loss.detach()) - rl) / rl, np.median(e_hip), np.median(e_cpu), e_hip.max(), e_cpu.max(),
                 l2(grads), l2(g32)))
        assert rel(out.detach().cpu(), ref) < NORTH_STAR and abs(float(loss.detach()) - rl) < NORTH_STAR * rl
        assert np.median(e_hip) <= 3.0 * np.median(e_cpu) + 1e-5
        assert l2(grads) <= 3.0 * l2(g32) + 1e-5
        assert e_hip.max() <= 5.0 * e_cpu.max() + 2e-4
        del m


def test_three_adam_steps_follow_the_reference_trajectory(env, golden_dir):
    '''Fixture T2 (three Adam steps of the real reference on the tiny net): the losses under f16x2 stay within north_star's bar.'''
    synth, train = env
    g = np.load(os.path.join(golden_dir, 'T2_tiny_adam3.npz'))
    n, h, w, k, dseed, wseed = [int(v) for v in g['meta']]
    m = train.build_model(synth.TINY, device='cuda')
    synth.fill_state_dict_([m.encoder, m.decoder], wseed)
    m.compute_dtype = 'f16x2'
    m.train()
    opt = train.make_optimizer(m, lr=1e-3)
    losses = []
    for step in range(3):
        b = {kk: v.cuda() for kk, v in synth.make_batch(n, h, w, k, seed=dseed + step).items()}
        losses.append(float(train.train_step(m, opt, b['image'], b['input_depth'], b['ground_truth'], b['lidar_map'])[0]))
    np.testing.assert_allclose(losses, g['losses'][:3], rtol=NORTH_STAR)
    psum = float(sum(p.detach().double().abs().sum() for p in m.parameters()))
    assert abs(psum - float(g['param_abs_sum'])) < 1e-4 * float(g['param_abs_sum'])


def test_captured_training_step_is_bitwise_the_eager_step(env, golden_dir):
    synth, train = env
    g = np.load(os.path.join(golden_dir, 'T1_published_train.npz'))
    n, h, w, k, dseed, wseed = [int(v) for v in g['meta']]
    b = {kk: v.cuda() for kk, v in synth.make_batch(n, h, w, k, seed=dseed).items()}
    args = (b['image'], b['input_depth'], b['ground_truth'], b['lidar_map'])
    res = []
    for captured in (False, True):
        m = train.build_model(synth.PUBLISHED, device='cuda')
        synth.fill_state_dict_([m.encoder, m.decoder], wseed)
        m.compute_dtype = 'f16x2'
        m.train()
        opt = train.make_optimizer(m, lr=1e-3)
        if captured:
            step = m.capture_training_step(opt, *args)
            ls = [float(step(*args).detach()) for _ in range(2)]
        else:
            ls = [float(train.train_step(m, opt, *args)[0]) for _ in range(2)]
        torch.cuda.synchronize()
        res.append((ls, [p.detach().clone() for _, p in _named(m)]))
    assert res[0][0] == res[1][0]
    for a, c in zip(res[0][1], res[1][1]):
        assert torch.equal(a, c)


@pytest.mark.parametrize('prec', ['f16x2', 'fp32'])
@pytest.mark.parametrize('cin,cout,n,hs,ws', [(64, 32, 2, 35, 51), (64, 64, 2, 24, 40), (128, 64, 1, 15, 25), (256, 256, 2, 8, 13), (32, 64, 1, 40, 70), (256, 256, 1, 7, 12),
                                              (64, 32, 1, 7, 12)])
def test_up2x_weight_gradient_four_phases_in_one_launch(ops, prec, cin, cout, n, hs, ws):
    '''rcf_conv2d_wgrad on the merged up-2x descriptor (phase_sum == 2): the four phase weight gradients [4][co][ci][2][2] from ONE
    launch of the split kernel (workgroup = (slot, phase); the four phases of a slot share an XCD).  Same products as the four
    per-phase calls, summed over fewer tile slots per phase: equal to them within fp32 summation order, and the folded 3x3 gradient
    within the tier's bar of the fp64 reference.'''
    x = rnd(n, cin, hs, ws, seed=21)
    dz = rnd(n, cout, 2 * hs, 2 * ws, seed=22, scale=1e-3)
    wshape = (cout, cin, 3, 3)
    ref = torch.nn.grad.conv2d_weight(F.interpolate(x.double(), scale_factor=2), wshape, dz.double(), padding=1)
    ops.set_precision(prec)
    try:
        ax, adz = dev_amax(ops, x), dev_amax(ops, dz)
        xg, dzg = nhwc(x), nhwc(dz)
        dwp = torch.full((4, cout, cin, 2, 2), float('nan'), device='cuda')
        for ph in range(4):
            d = ops.make_up2x_fwd_desc(n, hs, ws, cin, cout, ph >> 1, ph & 1)
            qi = ops.conv_query(d)
            wsb = torch.empty(max(1, qi.wgrad_workspace_floats), device='cuda')
            two = qi.wgrad_kernel_id >= 50000
            ops.conv_wgrad(d, xg, None, dzg, dwp[ph], wsb, scales=ops.make_scales(ax, None, None, adz) if (two and prec == 'f16x2') else None)
        dm = ops.make_up2x_fwd_desc(n, hs, ws, cin, cout, 0, 0, phase_out=True)
        qm = ops.conv_query(dm)
        nws = max(1, qm.wgrad_workspace_floats)
        guard = torch.full((nws + 65536,), 123.0, device='cuda')     # the launch must stay inside the workspace the query asked for
        wsm = guard[:nws]
        dwm = torch.full((4, cout, cin, 2, 2), float('nan'), device='cuda')
        ops.conv_wgrad(dm, xg, None, dzg, dwm, wsm, scales=ops.make_scales(ax, None, None, adz) if (qm.wgrad_kernel_id >= 50000 and prec == 'f16x2') else None)
        torch.cuda.synchronize()
        assert bool((guard[nws:] == 123.0).all())
        assert not torch.isnan(dwm).any()
        assert rel(dwm.cpu(), dwp.cpu().double()) < 2e-6
        dw = torch.empty(wshape, device='cuda')
        ops.phase_wgrad_fold(dwm, dw)
        torch.cuda.synchronize()
        assert rel(dw.cpu(), ref) < EXACT_TOL
    finally:
        ops.set_precision('fp32')


@pytest.mark.parametrize('cin,cout,n,h,w', [(32, 64, 2, 45, 80), (64, 128, 2, 35, 51), (128, 256, 1, 29, 50), (256, 256, 1, 8, 6), (64, 64, 1, 64, 96)])
def test_stride2_input_gradient_four_phases_from_one_staged_tile(ops, cin, cout, n, h, w):
    '''rcf_conv_desc.phase_sum == 3 on fp32 tensors / two fp16 planes (conv_split_kernel<SplitCfg<2, ., 32, ., 2, 1, 2>>): dz staged and
    split once per chunk, the nine (phase, tap) products that exist.  Bitwise the four per-phase launches (which multiply the other
    seven taps by zero weights), with and without accumulation, odd and even extents; against fp64 within the tier's bar.'''
    from rcf_amd._lib import RCF_PHASE_S2_DGRAD
    x = rnd(n, cin, h, w, seed=1).double().requires_grad_(True)
    wt = rnd(cout, cin, 3, 3, seed=2, scale=1.0 / np.sqrt(cin * 9))
    ref = F.conv2d(x, wt.double(), stride=2, padding=1)
    dz = rnd(*ref.shape, seed=3, scale=1e-3)
    (ref * dz.double()).sum().backward()
    ops.set_precision('f16x2')
    try:
        fwd = ops.make_fwd_desc(n, h, w, cin, 0, cout, 3, 2)
        wd = ops.phase_weights(wt.cuda(), RCF_PHASE_S2_DGRAD)
        adz, aw = dev_amax(ops, dz), ops.amax(wd)
        sc = ops.make_scales(adz, None, aw)
        for accumulate in (False, True):
            base = rnd(n, cin, h, w, seed=7, scale=1e-3) if accumulate else torch.zeros(n, cin, h, w)
            dx4 = nhwc(base) if accumulate else torch.full((n, h, w, cin), float('nan'), device='cuda')
            dx1 = dx4.clone()
            for ph in range(4):
                dd = ops.make_s2_dgrad_desc(fwd, ph >> 1, ph & 1, accumulate)
                info = ops.conv_query(dd)
                assert 40000 <= info.kernel_id < 50000
                packed = torch.empty(info.packed_weight_floats, device='cuda')
                ops.conv_pack(dd, wd[ph], packed, aw)
                ops.conv_fwd(dd, nhwc(dz), None, packed, dx4, None, scales=sc)
            dm = ops.make_s2_dgrad_desc(fwd, 0, 0, accumulate, phase_out=True)
            im = ops.conv_query(dm)
            assert 40000 <= im.kernel_id < 50000
            pm = torch.empty(4 * im.packed_weight_floats, device='cuda')
            for ph in range(4):
                ops.conv_pack(dm, wd[ph], pm[ph * im.packed_weight_floats:(ph + 1) * im.packed_weight_floats], aw)
            ops.conv_fwd(dm, nhwc(dz), None, pm, dx1, None, scales=sc)
            torch.cuda.synchronize()
            assert not torch.isnan(dx1).any()
            assert torch.equal(dx1, dx4), accumulate
            assert rel(nchw(dx1), x.grad + base.double()) < EXACT_TOL, accumulate
    finally:
        ops.set_precision('fp32')


def test_stride2_input_gradient_merged_form_is_refused_on_the_exact_tier(ops):
    '''phase_sum == 3 exists as the merged kernels only (two planes / bf16 tensors): the exact three-plane tier answers RCF_EUNSUPPORTED
    and the engine keeps the four launches there.'''
    ops.set_precision('fp32')
    fwd = ops.make_fwd_desc(1, 32, 48, 64, 0, 128, 3, 2)
    dm = ops.make_s2_dgrad_desc(fwd, 0, 0, False, phase_out=True)
    with pytest.raises(ops._lib.RcfError):
        ops.conv_query(dm)


@pytest.mark.parametrize('prec', ['f16x2', 'fp32'])
@pytest.mark.parametrize('cin,cout,n,h,w', [(32, 64, 2, 45, 80), (64, 128, 2, 35, 51), (128, 256, 1, 29, 50), (256, 256, 2, 15, 25), (256, 256, 1, 14, 24)])
def test_stride2_weight_gradient_four_phases_in_one_launch(ops, prec, cin, cout, n, h, w):
    '''rcf_conv2d_wgrad on the phase_sum == 1 descriptor: the four phase weight gradients of a 3x3 stride-2 convolution from ONE launch
    (workgroup = (slot, phase), the phases of a slot on one XCD sharing the dz tile); equal to the four per-phase calls within fp32
    summation order, and the gathered 3x3 gradient within the tier's bar of fp64.'''
    x = rnd(n, cin, h, w, seed=41)
    wshape = (cout, cin, 3, 3)
    ho, wo = (h - 1) // 2 + 1, (w - 1) // 2 + 1
    dz = rnd(n, cout, ho, wo, seed=42, scale=1e-3)
    ref = torch.nn.grad.conv2d_weight(x.double(), wshape, dz.double(), stride=2, padding=1)
    ops.set_precision(prec)
    try:
        fwd = ops.make_fwd_desc(n, h, w, cin, 0, cout, 3, 2)
        ax, adz = dev_amax(ops, x), dev_amax(ops, dz)
        xg, dzg = nhwc(x), nhwc(dz)
        dwp = torch.full((4, cout, cin, 2, 2), float('nan'), device='cuda')
        for ph in range(4):
            d = ops.make_s2_wgrad_desc(fwd, ph >> 1, ph & 1)
            qi = ops.conv_query(d)
            wsb = torch.empty(max(1, qi.wgrad_workspace_floats), device='cuda')
            two = qi.wgrad_kernel_id >= 50000 and prec == 'f16x2'
            ops.conv_wgrad(d, xg, None, dzg, dwp[ph], wsb, scales=ops.make_scales(ax, None, None, adz) if two else None)
        dm = ops.make_s2_wgrad_desc(fwd, 0, 0, all_phases=True)
        qm = ops.conv_query(dm)
        nws = max(1, qm.wgrad_workspace_floats)
        guard = torch.full((nws + 65536,), 123.0, device='cuda')     # the launch must stay inside the workspace the query asked for
        wsm = guard[:nws]
        dwm = torch.full((4, cout, cin, 2, 2), float('nan'), device='cuda')
        ops.conv_wgrad(dm, xg, None, dzg, dwm, wsm, scales=ops.make_scales(ax, None, None, adz) if (qm.wgrad_kernel_id >= 50000 and prec == 'f16x2') else None)
        torch.cuda.synchronize()
        assert bool((guard[nws:] == 123.0).all())
        assert not torch.isnan(dwm).any()
        assert rel(dwm.cpu(), dwp.cpu().double()) < 2e-6
        dw = torch.empty(wshape, device='cuda')
        ops.phase_wgrad_gather_s2(dwm, dw)
        torch.cuda.synchronize()
        assert rel(dw.cpu(), ref) < EXACT_TOL
    finally:
        ops.set_precision('fp32')
