'''
GPU tests of the bf16-STORAGE configuration (BASELINE.json configs 2-4: bf16 NHWC tensors in HBM, bf16 MFMA operands, fp32
accumulation; fp32 weights / BatchNorm statistics / loss / optimizer), kernel by kernel, through the C ABI.

Oracle for a kernel with bf16 tensors: the SAME computation in fp32 (stock PyTorch CPU ops) on the bf16-valued inputs, rounded to
bf16 once at the end -- what `x.bfloat16()` tensors mean.  Elementwise kernels do fp32 arithmetic on the loaded values and round on
store, so they must agree with the fp32 twin run on the same (bf16-valued) inputs after one rounding, to one bf16 ulp; the
convolutions accumulate in fp32, so they must agree with torch's fp32 convolution of the bf16-valued operands to accumulation-order
noise (1e-5) plus the final rounding (2^-9 relative).
'''

import numpy as np
import pytest
import torch
import torch.nn.functional as F

pytestmark = pytest.mark.gpu

BF16_EPS = 2.0 ** -8   # one bf16 ulp relative to the value (8 significant bits)


@pytest.fixture(scope='module')
def ops():
    import rcf_amd  # noqa: F401
    from rcf_amd import _lib, ops as _ops
    assert torch.cuda.is_available()
    _lib.load()
    return _ops


@pytest.fixture(autouse=True)
def _restore_precision(ops):
    yield
    ops.set_precision('fp32')


def rnd(*shape, seed=0, scale=1.0):
    g = torch.Generator().manual_seed(seed)
    return (torch.rand(*shape, generator=g) * 2 - 1) * scale


def b16(t):
    '''the fp32 tensor holding the bf16-rounded values of t'''
    return t.bfloat16().float()


def nhwc_b(t):
    return t.permute(0, 2, 3, 1).contiguous().cuda().bfloat16()


def nhwc_f(t):
    return t.permute(0, 2, 3, 1).contiguous().cuda()


def nchw(t):
    return t.detach().float().cpu().permute(0, 3, 1, 2).contiguous()


def close_bf16(got, want, slack=1.0):
    '''got (bf16-valued) equals round_bf16(want) up to `slack` ulps of the larger of the two'''
    got, want = got.double(), want.double()
    tol = slack * BF16_EPS * torch.maximum(got.abs(), want.abs()) + 1e-30
    bad = (got - want).abs() > tol
    return int(bad.sum()), float(((got - want).abs() / (want.abs().max() + 1e-30)).max())


# (ksize, stride, c1, c2, cout, n, h, w, up_from)
CASES = [
    (3, 1, 16, 0, 32, 2, 20, 37, None),
    (3, 1, 64, 32, 64, 2, 17, 40, None),        # decoder concat
    (3, 1, 32, 0, 32, 1, 40, 100, None),
    (3, 1, 8, 8, 8, 2, 35, 51, None),           # tiny net, f32-MFMA kernel with bf16 tensors
    (3, 1, 256, 0, 128, 1, 29, 50, (15, 25)),   # non-2x nearest gather
    (3, 1, 64, 0, 64, 3, 33, 40, None),         # batch 3: virtual-tall tiling
    (3, 2, 32, 0, 64, 2, 45, 80, None),         # stride 2
    (1, 1, 16, 0, 32, 2, 35, 51, None),         # fusion 1x1
    (1, 1, 32, 0, 64, 3, 33, 40, None),         # fusion / projection 1x1, batch 3 (virtual-tall tiling in the weight gradient)
    (1, 1, 64, 0, 128, 2, 17, 23, None),
    (1, 1, 128, 0, 256, 1, 15, 25, None),
    (1, 1, 128, 0, 200, 1, 9, 11, None),        # the last n-tile of the pointwise kernel is partial
    (1, 1, 48, 0, 40, 2, 13, 17, None),         # channel counts off the 32-channel blocks
    (1, 2, 32, 0, 64, 2, 45, 80, None),         # projection
    (1, 2, 128, 0, 256, 2, 29, 50, None),       # deep projection: 128 / 256 channels through the pointwise kernel's n-tiles
    (7, 2, 3, 0, 32, 2, 70, 102, None),         # stem: fp32 input, bf16 output
    (7, 2, 2, 0, 16, 2, 70, 102, None),
]


def _case(case, seed):
    k, s, c1, c2, co, n, h, w, up = case
    hs, ws = (h, w) if up is None else up
    x1 = rnd(n, c1, hs, ws, seed=seed)
    if k != 7:
        x1 = b16(x1)            # the stems read the fp32 network input as it is
    x2 = b16(rnd(n, c2, h, w, seed=seed + 1)) if c2 else None
    wt = rnd(co, c1 + c2, k, k, seed=seed + 2, scale=1.0 / np.sqrt((c1 + c2) * k * k))
    return x1, x2, wt


def _ref(case, x1, x2, wt, operands_bf16):
    k, s, c1, c2, co, n, h, w, up = case
    xin = x1 if up is None else F.interpolate(x1, size=(h, w))
    if x2 is not None:
        xin = torch.cat([xin, x2], 1)
    if operands_bf16:
        return F.conv2d(b16(xin).double(), b16(wt).double(), stride=s, padding=k // 2).float()
    return F.conv2d(xin.double(), wt.double(), stride=s, padding=k // 2).float()


def _desc(ops, case):
    k, s, c1, c2, co, n, h, w, up = case
    hs, ws = (h, w) if up is None else up
    return ops.make_fwd_desc(n, h, w, c1, c2, co, k, s, hs, ws, 0 if up is None else 1)


@pytest.mark.parametrize('case', CASES, ids=[str(c) for c in CASES])
def test_conv_forward_bf16_tensors(ops, case):
    ops.set_precision('bf16')
    k = case[0]
    x1, x2, wt = _case(case, 10)
    d = _desc(ops, case)
    assert d.storage == 1
    info = ops.conv_query(d)
    split = (info.kernel_id % 20000) // 1000 in (5, 6, 7, 9)
    ref = _ref(case, x1, x2, wt, operands_bf16=split)     # f32-MFMA kernels (1x1, stride 2, stems, tiny) keep fp32 operands
    packed = torch.empty(info.packed_weight_floats, device='cuda')
    ops.conv_pack(d, wt.cuda(), packed)
    out = torch.full((d.n, d.h_out, d.w_out, d.c_out), float('nan'), device='cuda').bfloat16()
    partials = torch.full((info.n_partials, 2, d.c_out), float('nan'), device='cuda', dtype=torch.float64)
    in1 = nhwc_f(x1) if k == 7 else nhwc_b(x1)
    ops.conv_fwd(d, in1, None if x2 is None else nhwc_b(x2), packed, out, partials)
    torch.cuda.synchronize()
    got = nchw(out)
    nbad, e = close_bf16(got, ref, slack=1.01)
    assert nbad <= max(2, got.numel() // 2000), (nbad, e)     # accumulation-order noise can flip a rounding at a tie
    assert e < 2 * BF16_EPS
    # BatchNorm statistics are those of the STORED (rounded) values
    s = partials.sum(0).cpu()
    gd = got.double()
    assert float((s[0] - gd.sum((0, 2, 3))).abs().max()) < 1e-6 * max(1.0, float(gd.abs().sum((0, 2, 3)).max()))
    assert float(((s[1] - (gd ** 2).sum((0, 2, 3))).abs() / (gd ** 2).sum((0, 2, 3))).max()) < 1e-6   # fp32 partial sums per accumulator, fp64 across tiles


@pytest.mark.parametrize('case', [c for c in CASES if c[0] != 7], ids=[str(c) for c in CASES if c[0] != 7])
def test_conv_input_gradient_bf16_tensors(ops, case):
    ops.set_precision('bf16')
    k, s, c1, c2, co, n, h, w, up = case
    x1, x2, wt = _case(case, 20)
    d = _desc(ops, case)
    dz = b16(rnd(n, co, d.h_out, d.w_out, seed=33))
    for src, off, cnt in ((x1, 0, c1), (x2, c1, c2)):
        if src is None:
            continue
        for accumulate in (False, True):
            dd = ops.make_dgrad_desc(d, off, cnt, accumulate and not (src is x1 and up is not None))
            info = ops.conv_query(dd)
            split = (info.kernel_id % 20000) // 1000 in (5, 6, 7, 9)
            wq = b16(wt) if split else wt
            xs = torch.zeros(n, c1 + c2, h, w, dtype=torch.double, requires_grad=True)
            (F.conv2d(xs, wq.double(), stride=s, padding=k // 2) * dz.double()).sum().backward()
            want_full = xs.grad[:, off:off + cnt].float()
            packed = torch.empty(info.packed_weight_floats, device='cuda')
            ops.conv_pack(dd, wt.cuda(), packed)
            if src is x1 and up is not None:
                tmp = torch.full((n, h, w, cnt), float('nan'), device='cuda').bfloat16()
                ops.conv_fwd(dd, nhwc_b(dz), None, packed, tmp, None)
                base = b16(rnd(n, cnt, up[0], up[1], seed=5)) if accumulate else torch.zeros(n, cnt, up[0], up[1])
                dst = nhwc_b(base) if accumulate else torch.full((n, up[0], up[1], cnt), float('nan'), device='cuda').bfloat16()
                ops.upsample_nearest_bwd(tmp, dst, accumulate)
                torch.cuda.synchronize()
                # the fan-out sum of the ROUNDED full-resolution gradient, in fp32, rounded once more
                full = b16(want_full)
                xsrc = torch.zeros(n, cnt, up[0], up[1], requires_grad=True)
                (F.interpolate(xsrc, size=(h, w)) * full).sum().backward()
                want = xsrc.grad + base
                nbad, e = close_bf16(nchw(dst), want, slack=2.0)
            else:
                base = b16(rnd(n, cnt, h, w, seed=5)) if accumulate else torch.zeros(n, cnt, h, w)
                dst = nhwc_b(base) if accumulate else torch.full((n, h, w, cnt), float('nan'), device='cuda').bfloat16()
                ops.conv_fwd(dd, nhwc_b(dz), None, packed, dst, None)
                torch.cuda.synchronize()
                want = want_full + base
                nbad, e = close_bf16(nchw(dst), want, slack=1.01)
            assert nbad <= max(2, dst.numel() // 1000), (off, accumulate, nbad, e)
            assert e < 3 * BF16_EPS


@pytest.mark.parametrize('case', CASES, ids=[str(c) for c in CASES])
def test_conv_weight_gradient_bf16_tensors(ops, case):
    ops.set_precision('bf16')
    k, s, c1, c2, co, n, h, w, up = case
    x1, x2, wt = _case(case, 30)
    d = _desc(ops, case)
    info = ops.conv_query(d)
    dz = b16(rnd(n, co, d.h_out, d.w_out, seed=44))
    wd = wt.clone().double().requires_grad_(True)
    xin = x1 if up is None else F.interpolate(x1, size=(h, w))
    if x2 is not None:
        xin = torch.cat([xin, x2], 1)
    # split kernels round x and dz to bf16 (they already are, the stem input excepted -- and stems run on the f32 MFMA)
    (F.conv2d(xin.double(), wd, stride=s, padding=k // 2) * dz.double()).sum().backward()
    ws = torch.empty(max(1, info.wgrad_workspace_floats), device='cuda')
    dw = torch.full(tuple(wt.shape), float('nan'), device='cuda')
    in1 = nhwc_f(x1) if k == 7 else nhwc_b(x1)
    ops.conv_wgrad(d, in1, None if x2 is None else nhwc_b(x2), nhwc_b(dz), dw, ws)
    torch.cuda.synchronize()
    e = float((dw.cpu().double() - wd.grad).abs().max() / wd.grad.abs().max())
    assert e < 2e-4, e      # operands are exact: only fp32 accumulation order differs (the gradient itself stays fp32)


SUMS_CASES_B16 = [('3x3', 64, 64, 2, 33, 64), ('3x3', 32, 32, 1, 70, 102), ('3x3', 256, 256, 2, 8, 13), ('3x3', 16, 32, 3, 20, 37),
                  ('up2x', 64, 32, 1, 35, 51), ('up2x', 64, 64, 2, 12, 20)]


@pytest.mark.parametrize('case', SUMS_CASES_B16, ids=[str(c) for c in SUMS_CASES_B16])
def test_input_gradient_kernel_takes_the_batchnorm_backward_sums_bf16_tensors(ops, case):
    '''rcf_conv2d_dgrad_bn_sums on bf16 tensors (conv_b16_kernel<C, false, true>): the input gradient dx written together with the
    backward sums of the BatchNorm + LeakyReLU block whose output it is the gradient of -- dx bitwise what rcf_conv2d_fwd writes,
    the sums equal to rcf_bn_act_bwd_reduce_b16's over dx and z (same terms; the epilogue adds 8 of them in fp32 before the fp64
    sum) and to an fp64 host evaluation.'''
    from rcf_amd._lib import RCF_ACT_LEAKY_RELU, RCF_PHASE_UP2X_DGRAD
    kind, ci, co, n, h, w = case
    ops.set_precision('bf16')
    wt = rnd(co, ci, 3, 3, seed=11, scale=1.0 / np.sqrt(ci * 9)).cuda()
    if kind == '3x3':
        d = ops.make_fwd_desc(n, h, w, ci, 0, co, 3, 1)
        dd = ops.make_dgrad_desc(d, 0, ci, False)
        dz = nhwc_b(rnd(n, co, h, w, seed=12))
        wsrc = [wt]
    else:
        dd = ops.make_up2x_dgrad_desc(n, h, w, ci, co, 0, 0, False, phase_sum=True)
        dz = nhwc_b(rnd(n, co, 2 * h, 2 * w, seed=12))
        wph = ops.phase_weights(wt, RCF_PHASE_UP2X_DGRAD)
        wsrc = [wph[ph] for ph in range(4)]
    info = ops.conv_query(dd)
    assert info.bn_bwd_sums == 1
    packed = torch.empty(len(wsrc) * info.packed_weight_floats, device='cuda')
    for i, wsl in enumerate(wsrc):
        ops.conv_pack(dd, wsl, packed[i * info.packed_weight_floats:(i + 1) * info.packed_weight_floats])
    z = nhwc_b(rnd(n, ci, h, w, seed=13))
    g_ = torch.Generator().manual_seed(14)
    gamma, beta = torch.rand(ci, generator=g_) * 2 - 1, torch.rand(ci, generator=g_) - 0.5
    mean, invstd = torch.rand(ci, generator=g_) - 0.5, 0.5 + 2 * torch.rand(ci, generator=g_)
    coef = torch.stack([gamma * invstd, beta - mean * gamma * invstd, mean, invstd]).contiguous().cuda()
    n_pix = n * h * w
    dx_ref = torch.full((n, h, w, ci), float('nan'), device='cuda', dtype=torch.bfloat16)
    ops.conv_fwd(dd, dz, None, packed, dx_ref, None)
    nb = ops.ew_blocks(n_pix, ci)
    part_ref = torch.empty((nb, 2, ci), dtype=torch.float64, device='cuda')
    ops.bn_act_bwd_reduce(dx_ref, z, coef, None, part_ref, n_pix, ci, RCF_ACT_LEAKY_RELU, False)
    dx = torch.full((n, h, w, ci), float('nan'), device='cuda', dtype=torch.bfloat16)
    part = torch.full((info.n_partials, 2, ci), float('nan'), dtype=torch.float64, device='cuda')
    ops.conv_dgrad_bn_sums(dd, dz, packed, dx, z, coef, part, None)
    torch.cuda.synchronize()
    assert torch.equal(dx, dx_ref)
    s_ref, s_new = part_ref.sum(0), part.sum(0)
    zf = z.float()
    gd = dx.double() * torch.where(zf * coef[0] + coef[1] > 0, 1.0, 0.2)
    xh = (zf.double() - coef[2].double()) * coef[3].double()
    want = torch.stack([gd.sum((0, 1, 2)), (gd * xh).sum((0, 1, 2))])
    mag = torch.stack([gd.abs().sum((0, 1, 2)), (gd * xh).abs().sum((0, 1, 2))])
    assert float(((s_new - want).abs() / mag).max()) < 1e-6
    assert float(((s_new - s_ref).abs() / mag).max()) < 1e-6


@pytest.mark.parametrize('c1,co,h,w', [(64, 32, 35, 51), (64, 64, 29, 50), (128, 64, 15, 25), (16, 32, 40, 18), (32, 64, 9, 70), (256, 128, 8, 13)])
def test_up2x_forward_four_phases_from_one_staged_tile(ops, c1, co, h, w):
    '''rcf_conv_desc.phase_sum == 2 on bf16 tensors (conv_b16_kernel<DmaCfg<2, ., ., ., 1, true>>): x staged once per channel chunk,
    the 16 (phase, tap) products into four accumulator sets.  Bitwise the four per-phase launches (same products in the same order
    per phase), statistics of the whole output; 32- and 64-channel n-tiles, 32- and 16-pixel tile rows, ragged edges.'''
    from rcf_amd._lib import RCF_PHASE_UP2X_FWD
    ops.set_precision('bf16')
    n = 2
    x = b16(rnd(n, c1, h, w, seed=11))
    wt = rnd(co, c1, 3, 3, seed=12, scale=1.0 / np.sqrt(c1 * 9))
    ref = F.conv2d(F.interpolate(x.double(), scale_factor=2), wt.double(), padding=1)
    wp = ops.phase_weights(wt.cuda(), RCF_PHASE_UP2X_FWD)
    xg = nhwc_b(x)
    z = torch.full((n, 2 * h, 2 * w, co), float('nan'), device='cuda').bfloat16()
    for ph in range(4):
        d = ops.make_up2x_fwd_desc(n, h, w, c1, co, ph >> 1, ph & 1)
        info = ops.conv_query(d)
        packed = torch.empty(info.packed_weight_floats, device='cuda')
        ops.conv_pack(d, wp[ph], packed)
        ops.conv_fwd(d, xg, None, packed, z, None)
    dm = ops.make_up2x_fwd_desc(n, h, w, c1, co, 0, 0, phase_out=True)
    im = ops.conv_query(dm)
    pm = torch.empty(4 * im.packed_weight_floats, device='cuda')
    for ph in range(4):
        ops.conv_pack(dm, wp[ph], pm[ph * im.packed_weight_floats:(ph + 1) * im.packed_weight_floats])
    z1 = torch.full((n, 2 * h, 2 * w, co), float('nan'), device='cuda').bfloat16()
    part = torch.full((im.n_partials, 2, co), float('nan'), device='cuda', dtype=torch.float64)
    ops.conv_fwd(dm, xg, None, pm, z1, part)
    z2 = torch.full((n, 2 * h, 2 * w, co), float('nan'), device='cuda').bfloat16()
    ops.conv_fwd(dm, xg, None, pm, z2, None)      # without statistics
    torch.cuda.synchronize()
    assert not torch.isnan(z1.float()).any()
    assert torch.equal(z1, z) and torch.equal(z2, z)
    e = float((nchw(z1).double() - ref).abs().max() / ref.abs().max())
    assert e < 3 * BF16_EPS, e
    st = part.sum(0).cpu()
    np.testing.assert_allclose(st[0].numpy(), nchw(z).double().sum((0, 2, 3)).numpy(), rtol=1e-6, atol=1e-3)
    np.testing.assert_allclose(st[1].numpy(), (nchw(z).double() ** 2).sum((0, 2, 3)).numpy(), rtol=1e-6, atol=1e-3)


def test_phase_convolutions_bf16_tensors(ops):
    '''Exact-2x UpConv as four 2x2 phase convolutions, its merged-phase input gradient and its weight gradient on bf16 tensors,
    against the 9-tap reference (phase weights are pre-summed in fp32 and then rounded once: 1 extra bf16 ulp of slack).'''
    from rcf_amd._lib import RCF_PHASE_UP2X_DGRAD, RCF_PHASE_UP2X_FWD
    ops.set_precision('bf16')
    n, c1, co, h, w = 2, 64, 32, 35, 51
    x = b16(rnd(n, c1, h, w, seed=1))
    wt = rnd(co, c1, 3, 3, seed=2, scale=1.0 / np.sqrt(c1 * 9))
    xr = x.clone().double().requires_grad_(True)
    wr = wt.clone().double().requires_grad_(True)
    ref = F.conv2d(F.interpolate(xr, scale_factor=2), wr, padding=1)
    dz = b16(rnd(n, co, 2 * h, 2 * w, seed=3))
    (ref * dz.double()).sum().backward()
    wp = ops.phase_weights(wt.cuda(), RCF_PHASE_UP2X_FWD)
    z = torch.full((n, 2 * h, 2 * w, co), float('nan'), device='cuda').bfloat16()
    xg = nhwc_b(x)
    descs = []
    for ph in range(4):
        d = ops.make_up2x_fwd_desc(n, h, w, c1, co, ph >> 1, ph & 1)
        info = ops.conv_query(d)
        packed = torch.empty(info.packed_weight_floats, device='cuda')
        ops.conv_pack(d, wp[ph], packed)
        ops.conv_fwd(d, xg, None, packed, z, None)
        descs.append(d)
    torch.cuda.synchronize()
    e = float((nchw(z).double() - ref.detach()).abs().max() / ref.detach().abs().max())
    assert e < 3 * BF16_EPS, e
    # the same four phases in ONE launch (rcf_conv_desc.phase_sum == 2): bitwise the four launches' output, statistics of all of it
    dm = ops.make_up2x_fwd_desc(n, h, w, c1, co, 0, 0, phase_out=True)
    im = ops.conv_query(dm)
    pm = torch.empty(4 * im.packed_weight_floats, device='cuda')
    for ph in range(4):
        ops.conv_pack(dm, wp[ph], pm[ph * im.packed_weight_floats:(ph + 1) * im.packed_weight_floats])
    z1 = torch.full((n, 2 * h, 2 * w, co), float('nan'), device='cuda').bfloat16()
    part = torch.full((im.n_partials, 2, co), float('nan'), device='cuda', dtype=torch.float64)
    ops.conv_fwd(dm, xg, None, pm, z1, part)
    torch.cuda.synchronize()
    assert torch.equal(z1, z)
    st = part.sum(0).cpu()
    np.testing.assert_allclose(st[0].numpy(), nchw(z).double().sum((0, 2, 3)).numpy(), rtol=1e-6, atol=1e-3)
    np.testing.assert_allclose(st[1].numpy(), (nchw(z).double() ** 2).sum((0, 2, 3)).numpy(), rtol=1e-6, atol=1e-3)
    # weight gradient: four 2x2 wgrads folded back to 3x3
    dwp = torch.empty((4, co, c1, 2, 2), device='cuda')
    dzg = nhwc_b(dz)
    for ph, d in enumerate(descs):
        qi = ops.conv_query(d)
        ws = torch.empty(max(1, qi.wgrad_workspace_floats), device='cuda')
        ops.conv_wgrad(d, xg, None, dzg, dwp[ph], ws)
    dw = torch.empty((co, c1, 3, 3), device='cuda')
    ops.phase_wgrad_fold(dwp, dw)
    torch.cuda.synchronize()
    e = float((dw.cpu().double() - wr.grad).abs().max() / wr.grad.abs().max())
    assert e < 2e-4, e
    # input gradient: the four phases of dZ summed inside one launch
    wd = ops.phase_weights(wt.cuda(), RCF_PHASE_UP2X_DGRAD)
    dd = ops.make_up2x_dgrad_desc(n, h, w, c1, co, 0, 0, False, phase_sum=True)
    qi = ops.conv_query(dd)
    packed = torch.empty(4 * qi.packed_weight_floats, device='cuda')
    for ph in range(4):
        ops.conv_pack(dd, wd[ph], packed[ph * qi.packed_weight_floats:(ph + 1) * qi.packed_weight_floats])
    dx = torch.full((n, h, w, c1), float('nan'), device='cuda').bfloat16()
    ops.conv_fwd(dd, dzg, None, packed, dx, None)
    torch.cuda.synchronize()
    e = float((nchw(dx).double() - xr.grad).abs().max() / xr.grad.abs().max())
    assert e < 3 * BF16_EPS, e


@pytest.mark.parametrize('c,n,h,w,has_res', [(32, 2, 19, 23, False), (64, 1, 30, 17, True), (4, 2, 35, 51, True), (256, 2, 8, 13, False)])
def test_elementwise_twins_equal_fp32_kernels_on_bf16_values(ops, c, n, h, w, has_res):
    '''bn_act fwd / bwd_reduce / bwd_apply, fuse fwd / bwd, max pool, nearest-upsample backward: NAME_b16 on bf16 tensors ==
    NAME on the same values held in fp32, rounded to bf16 once (bit-exact for the elementwise results, exact for the fp64 sums).'''
    from rcf_amd._lib import RCF_ACT_LEAKY_RELU
    npix = n * h * w
    z = b16(rnd(n, h, w, c, seed=1, scale=2.0) + 0.3).cuda()
    res = b16(rnd(n, h, w, c, seed=2)).cuda()
    dout = b16(rnd(n, h, w, c, seed=3)).cuda()
    coef = torch.stack([rnd(c, seed=4) * 0.5 + 1.0, rnd(c, seed=5) * 0.1, rnd(c, seed=6) * 0.1 + 0.3, rnd(c, seed=7) * 0.2 + 1.0]).cuda()
    bcoef = torch.stack([rnd(c, seed=8) * 0.01, rnd(c, seed=9) * 0.01]).cuda()

    def both(fn):
        return fn(lambda t: t), fn(lambda t: None if t is None else t.bfloat16())

    # forward
    def fwd(cast):
        out = torch.empty_like(cast(z))
        ops.bn_act_fwd(cast(z), coef, cast(res) if has_res else None, out, npix, c, RCF_ACT_LEAKY_RELU)
        return out
    of, ob = both(fwd)
    assert torch.equal(ob.float(), of.bfloat16().float())
    out_act = of.bfloat16()     # what the network keeps
    # backward reduce: identical fp64 partial sums (same values, same order)
    nb = ops.ew_blocks(npix, c)

    def red(cast):
        part = torch.empty((nb, 2, c), dtype=torch.float64, device='cuda')
        ops.bn_act_bwd_reduce(cast(dout), cast(z), coef, cast(out_act.float()), part, npix, c, RCF_ACT_LEAKY_RELU, has_res)
        return part
    pf, pb = both(red)
    assert torch.equal(pf, pb)

    def app(cast):
        dz = torch.empty_like(cast(z))
        dres = torch.empty_like(cast(z)) if has_res else None
        ops.bn_act_bwd_apply(cast(dout), cast(z), coef, cast(out_act.float()), bcoef, dz, dres, False, npix, c, RCF_ACT_LEAKY_RELU, has_res)
        return dz if dres is None else torch.cat([dz, dres])
    af, ab = both(app)
    assert torch.equal(ab.float(), af.bfloat16().float())
    # max pool + its backward (indices identical, values pass through unrounded)
    if c >= 4:
        def pool(cast):
            ho, wo = (h - 1) // 2 + 1, (w - 1) // 2 + 1
            out = torch.empty((n, ho, wo, c), device='cuda', dtype=cast(z).dtype)
            idx = torch.empty((n, ho, wo, c), dtype=torch.uint8, device='cuda')
            ops.maxpool_fwd(cast(z), out, idx)
            din = torch.empty_like(cast(z))
            ops.maxpool_bwd(cast(dout)[:, :ho, :wo].contiguous(), idx, din, False)
            return out, idx, din
        (o1, i1, d1), (o2, i2, d2) = both(pool)
        assert torch.equal(i1, i2) and torch.equal(o2.float(), o1) and torch.equal(d2.float(), d1.bfloat16().float())


def test_fusion_twins_equal_fp32_kernels_on_bf16_values(ops):
    n, h, w, c = 2, 17, 23, 64
    npix = n * h * w
    zw = b16(rnd(n, h, w, c, seed=1, scale=2.0)).cuda()
    zp = b16(rnd(n, h, w, c, seed=2, scale=2.0)).cuda()
    img = b16(rnd(n, h, w, c, seed=3)).cuda()
    dout = b16(rnd(n, h, w, c, seed=4)).cuda()
    cw = torch.stack([rnd(c, seed=5) * 0.5 + 1.0, rnd(c, seed=6) * 0.1, rnd(c, seed=7) * 0.1, rnd(c, seed=8) * 0.2 + 1.0]).cuda()
    cp = torch.stack([rnd(c, seed=9) * 0.5 + 1.0, rnd(c, seed=10) * 0.1, rnd(c, seed=11) * 0.1, rnd(c, seed=12) * 0.2 + 1.0]).cuda()
    bw = torch.stack([rnd(c, seed=13) * 0.01, rnd(c, seed=14) * 0.01]).cuda()
    bp = torch.stack([rnd(c, seed=15) * 0.01, rnd(c, seed=16) * 0.01]).cuda()
    res = []
    for cast in (lambda t: t, lambda t: t.bfloat16()):
        out = torch.empty_like(cast(zw))
        ops.fuse_fwd(cast(zw), cw, cast(zp), cp, cast(img), out, npix, c)
        nb = ops.ew_blocks(npix, c)
        part = torch.empty((nb, 4, c), dtype=torch.float64, device='cuda')
        ops.fuse_bwd_reduce(cast(dout), cast(zw), cw, cast(zp), cp, part, npix, c)
        dzw, dzp, dimg = torch.empty_like(out), torch.empty_like(out), torch.empty_like(out)
        ops.fuse_bwd_apply(cast(dout), cast(zw), cw, cast(zp), cp, bw, bp, dzw, dzp, dimg, False, npix, c)
        res.append((out, part, dzw, dzp, dimg))
    f, b = res
    assert torch.equal(f[1], b[1])
    for i in (0, 2, 3, 4):
        assert torch.equal(b[i].float(), f[i].bfloat16().float()), i


@pytest.mark.parametrize('c_d,c_i,n,h,w', [(16, 32, 2, 37, 53), (32, 64, 2, 19, 23), (64, 128, 1, 31, 17), (128, 256, 2, 8, 13), (16, 24, 1, 9, 11),
                                           (64, 70, 1, 5, 7)])
def test_inference_fusion_in_one_pass(ops, c_d, c_i, n, h, w):
    '''rcf_fuse_wp_infer_b16 = sigmoid(BN_w(W1 d)) * BN_p(W2 d) + img with eval-mode BatchNorm (src/networks.py:863-866), against (a) fp32
    torch on the same bf16-rounded operands (scaled weights rounded to bf16 as the kernel's B operand is): one bf16 ulp of the result
    plus the accumulation order; (b) the three-kernel arrangement it replaces (two 1x1 convolutions that round zw, zp to bf16, then
    rcf_fuse_fwd_b16): within the bf16 roundings that arrangement adds.  Ragged pixel counts and channel counts off the 32-channel tile.'''
    d = b16(rnd(n, h, w, c_d, seed=1, scale=2.0))
    img = b16(rnd(n, h, w, c_i, seed=2))
    w1 = rnd(c_i, c_d, 1, 1, seed=3, scale=(1.0 / c_d) ** 0.5)
    w2 = rnd(c_i, c_d, 1, 1, seed=4, scale=(1.0 / c_d) ** 0.5)
    cw = torch.stack([rnd(c_i, seed=5) * 0.5 + 1.0, rnd(c_i, seed=6) * 0.3, rnd(c_i, seed=7) * 0.1, rnd(c_i, seed=8) * 0.2 + 1.0])
    cp = torch.stack([rnd(c_i, seed=9) * 0.5 + 1.0, rnd(c_i, seed=10) * 0.3, rnd(c_i, seed=11) * 0.1, rnd(c_i, seed=12) * 0.2 + 1.0])
    assert ops.fuse_wp_infer_supported(c_d, c_i)
    out = torch.full((n, h, w, c_i), float('nan'), device='cuda').bfloat16()
    ops.fuse_wp_infer(d.cuda().bfloat16(), w1.cuda(), cw.cuda(), w2.cuda(), cp.cuda(), img.cuda().bfloat16(), out)
    torch.cuda.synchronize()
    got = out.float().cpu()
    assert torch.isfinite(got).all()
    ws1 = b16(w1.view(c_i, c_d) * cw[0][:, None]).double()
    ws2 = b16(w2.view(c_i, c_d) * cp[0][:, None]).double()
    yw = d.double().view(-1, c_d) @ ws1.t() + cw[1].double()
    yp = d.double().view(-1, c_d) @ ws2.t() + cp[1].double()
    want = (torch.sigmoid(yw) * yp + img.double().view(-1, c_i)).view(n, h, w, c_i)
    err = (got.double() - want).abs()
    bound = BF16_EPS * want.abs() + 2e-5 * (1.0 + yp.abs().view(n, h, w, c_i))   # one ulp of the result + fp32 accumulation of c_d terms
    assert (err <= bound).all(), float((err - bound).max())
    if c_i % 32 == 0:   # (b) the arrangement it replaces (its kernels take whole channel tiles)
        ops.set_precision('bf16')
        desc = ops.make_fwd_desc(n, h, w, c_d, 0, c_i, 1, 1, h, w, 0)
        info = ops.conv_query(desc)
        zs = []
        for wt in (w1, w2):
            packed = torch.empty(info.packed_weight_floats, device='cuda')
            ops.conv_pack(desc, wt.cuda(), packed)
            z = torch.empty((n, h, w, c_i), device='cuda').bfloat16()
            ops.conv_fwd(desc, d.cuda().bfloat16(), None, packed, z, None)
            zs.append(z)
        old = torch.empty_like(out)
        ops.fuse_fwd(zs[0], cw.cuda(), zs[1], cp.cuda(), img.cuda().bfloat16(), old, n * h * w, c_i)
        torch.cuda.synchronize()
        rel = float((old.float() - out.float()).norm() / out.float().norm())
        assert rel < 3 * BF16_EPS, rel


def test_inference_fusion_refuses_what_it_does_not_cover(ops):
    from rcf_amd import _lib
    assert not ops.fuse_wp_infer_supported(48, 96) and not ops.fuse_wp_infer_supported(16, 31)
    d = torch.zeros((1, 4, 4, 48), device='cuda').bfloat16()
    img = torch.zeros((1, 4, 4, 96), device='cuda').bfloat16()
    coef = torch.zeros((4, 96), device='cuda')
    wt = torch.zeros((96, 48, 1, 1), device='cuda')
    with pytest.raises(_lib.RcfError):
        ops.fuse_wp_infer(d, wt, coef, wt, coef, img, torch.empty_like(img))
    with pytest.raises(ValueError):
        ops.fuse_wp_infer(d.float(), wt, coef, wt, coef, img, torch.empty_like(img))


def test_head_kernels_bf16_input(ops):
    '''3x3 C->1 head on a bf16 activation (and on a deferred bf16 z + coefficients): logits / depth stay fp32 and equal the fp32
    kernel's on the same values; the input gradient is the fp32 one rounded once.'''
    n, h, w, c = 2, 37, 50, 32
    x = b16(rnd(n, h, w, c, seed=1)).cuda()
    wt = (rnd(1, c, 3, 3, seed=2) * 0.1).cuda()
    coef = torch.stack([rnd(c, seed=4) * 0.5 + 1.0, rnd(c, seed=5) * 0.1, rnd(c, seed=6) * 0.1, rnd(c, seed=7) * 0.2 + 1.0]).cuda()
    dl = rnd(n, h, w, seed=9).cuda()
    for cf in (None, coef):
        outs = []
        for cast in (lambda t: t, lambda t: t.bfloat16()):
            logit = torch.empty((n, h, w), device='cuda')
            depth = torch.empty((n, h, w), device='cuda')
            ops.head_fwd(cast(x), wt, logit, depth, 1.0, 100.0, coef=cf)
            dw = torch.empty_like(wt)
            ops.head_bwd_wgrad(cast(x), dl, dw, coef=cf)
            outs.append((logit, depth, dw))
        if cf is None:
            # stored bf16 activations run on the bf16 MFMA against the three exact bf16 planes of the fp32 weights
            # (head_fwd_mfma_b16_kernel), fp32 tensors on the tile kernel: the same exact products in another summation order
            assert float((outs[0][0] - outs[1][0]).abs().max()) <= 2e-6 * float(outs[0][0].abs().max())
            assert float((outs[0][1] - outs[1][1]).abs().max()) <= 2e-6 * float(outs[0][1].abs().max())
        else:   # BatchNorm + LeakyReLU on load: both storages feed the same fp32 values to the same f32-MFMA steps
            assert torch.equal(outs[0][0], outs[1][0]) and torch.equal(outs[0][1], outs[1][1])
        assert float((outs[0][2] - outs[1][2]).abs().max()) <= 1e-6 * float(outs[0][2].abs().max())
    dx_f = torch.empty((n, h, w, c), device='cuda')
    dx_b = torch.empty((n, h, w, c), device='cuda').bfloat16()
    ops.head_bwd_dgrad(dl, wt, dx_f)
    ops.head_bwd_dgrad(dl, wt, dx_b)
    assert torch.equal(dx_b.float(), dx_f.bfloat16().float())


def test_convert_round_trip(ops):
    x = rnd(1000003, seed=3).cuda()
    y = torch.empty_like(x).bfloat16()
    ops.convert(x, y)
    assert torch.equal(y, x.bfloat16())          # round to nearest even, like torch
    z = torch.ones_like(x)
    ops.convert(y, z, accumulate=True)
    assert torch.equal(z, 1.0 + y.float())


def test_tiny_network_train_step_bf16_storage_against_oracle(ops):
    '''End to end on the tiny topology (odd sizes at every level): one training step with bf16 tensors against the fp32 CPU oracle.'''
    from oracle.fusionnet_oracle import FusionNetOracle
    from rcf_amd import synth, train
    m = train.build_model(synth.TINY, device='cuda')
    synth.fill_state_dict_([m.encoder, m.decoder], 7)
    m.compute_dtype = 'bf16'
    o = FusionNetOracle(**synth.TINY)
    synth.fill_state_dict_([o.encoder, o.decoder], 7)
    cb = synth.make_batch(2, 70, 102, 8, seed=77)
    b = {k: v.cuda() for k, v in cb.items()}
    opt = train.make_optimizer(m, lr=1e-3)
    m.train()
    loss, _, out = train.train_step(m, opt, b['image'], b['input_depth'], b['ground_truth'], b['lidar_map'])
    torch.cuda.synchronize()
    o.train()
    ref = o.forward(cb['image'], cb['input_depth'])
    ref_loss = o.compute_loss(ref, cb['ground_truth'], cb['lidar_map'], 2.0)[0]
    ref_loss.backward()
    e = float((out.detach().cpu() - ref.detach()).abs().max() / ref.detach().abs().max())
    print('tiny bf16-storage step: out rel %.2e, loss %.5f vs %.5f' % (e, float(loss), float(ref_loss)))
    assert e < 6e-2 and abs(float(loss) - float(ref_loss)) < 2e-2 * abs(float(ref_loss))
    # parameter order differs (arena order vs module order): compare per named tensor
    names_m = dict(list(m.encoder.named_parameters()) + [('d.' + k, v) for k, v in m.decoder.named_parameters()])
    names_o = dict(list(o.encoder.named_parameters()) + [('d.' + k, v) for k, v in o.decoder.named_parameters()])
    num = den_a = den_b = 0.0
    for k, p in names_o.items():
        if p.grad is None:
            continue
        a, bb = names_m[k].grad.double().cpu().reshape(-1), p.grad.double().reshape(-1)
        num += float(torch.dot(a, bb)); den_a += float(torch.dot(a, a)); den_b += float(torch.dot(bb, bb))
    cos = num / np.sqrt(den_a * den_b)
    print('gradient cosine vs fp32 oracle %.4f' % cos)
    assert cos > 0.95


@pytest.mark.parametrize('cfg', ['tiny', 'published'])
def test_inference_with_the_one_pass_fusion_against_the_oracle_and_the_three_kernel_form(ops, cfg):
    '''bf16 inference (eval-mode BatchNorm folded) with the fusion levels the one-pass kernel covers (tiny: the 16-channel depth levels;
    published: all six) against the fp32 CPU oracle in eval mode and against the same forward with RCF_FUSE_WP_ONE_PASS off.'''
    from oracle.fusionnet_oracle import FusionNetOracle
    from rcf_amd import synth, train
    conf = synth.TINY if cfg == 'tiny' else synth.PUBLISHED
    m = train.build_model(conf, device='cuda')
    synth.fill_state_dict_([m.encoder, m.decoder], 7)
    m.compute_dtype = 'bf16'
    m.eval()
    o = FusionNetOracle(**conf)
    synth.fill_state_dict_([o.encoder, o.decoder], 7)
    o.eval()
    cb = synth.make_batch(2, 70, 102, 8, seed=78)
    calls = []
    real = ops.fuse_wp_infer
    ops.fuse_wp_infer = lambda *a: (calls.append(a[0].shape[-1]), real(*a))[1]
    try:
        with torch.no_grad():
            one = m.forward(cb['image'].cuda(), cb['input_depth'].cuda()).float().cpu()
            n_one = len(calls)
            m._engine.fuse_wp_one_pass = False
            three = m.forward(cb['image'].cuda(), cb['input_depth'].cuda()).float().cpu()
            ref = o.forward(cb['image'], cb['input_depth'])
    finally:
        ops.fuse_wp_infer = real
    assert n_one == (4 if cfg == 'tiny' else 6) and len(calls) == n_one, calls   # tiny: depth filters 4 8 16 16 16 16
    e_one = float((one - ref).abs().max() / ref.abs().max())
    e_three = float((three - ref).abs().max() / ref.abs().max())
    print('%s bf16 inference vs oracle: one-pass fusion %.2e, three kernels %.2e' % (cfg, e_one, e_three))
    assert e_one < 6e-2 and e_one < 1.5 * e_three + 5e-3


@pytest.mark.parametrize('c,co,n,h,w', [(3, 32, 2, 70, 102), (2, 16, 2, 71, 101), (3, 8, 1, 64, 96), (2, 4, 3, 33, 35)])
def test_stem_as_4x4_convolution_on_the_space_to_depth_image(ops, c, co, n, h, w):
    '''The 7x7 stride-2 stems with bf16 tensors: rcf_s2d_image_b16 + rcf_stem_weights_s2d + the ksize-4 convolution == torch's
    conv2d(k 7, stride 2, pad 3) of the bf16-rounded image and weights (fp32 accumulate), rounded once; odd and even extents.'''
    ops.set_precision('bf16')
    x = rnd(n, c, h, w, seed=1)
    wt = rnd(co, c, 7, 7, seed=2, scale=1.0 / np.sqrt(c * 49))
    ref = F.conv2d(b16(x).double(), b16(wt).double(), stride=2, padding=3).float()
    s2d = ops.s2d_image(x.cuda())
    # the space-to-depth image itself
    hs, ws = (h + 1) // 2, (w + 1) // 2
    want = torch.zeros(n, hs, ws, 16)
    for a in range(2):
        for b in range(2):
            sub = b16(x)[:, :, a::2, b::2]
            want[:, :sub.shape[2], :sub.shape[3], a * 8 + b * 4:a * 8 + b * 4 + c] = sub.permute(0, 2, 3, 1)
    assert torch.equal(s2d.float().cpu(), want)
    d = ops.make_stem_s2d_desc(n, h, w, co)
    info = ops.conv_query(d)
    packed = torch.empty(info.packed_weight_floats, device='cuda')
    ops.conv_pack(d, ops.stem_weights_s2d(wt.cuda()), packed)
    out = torch.full((n, d.h_out, d.w_out, co), float('nan'), device='cuda').bfloat16()
    part = torch.empty((info.n_partials, 2, co), dtype=torch.float64, device='cuda')
    ops.conv_fwd(d, s2d, None, packed, out, part)
    torch.cuda.synchronize()
    assert tuple(nchw(out).shape) == tuple(ref.shape)
    nbad, e = close_bf16(nchw(out), ref, slack=1.01)
    assert nbad <= max(2, out.numel() // 2000) and e < 2 * BF16_EPS, (nbad, e)


def test_bf16_training_trajectory_tracks_fp32(ops):
    '''Ten Adam steps on one batch (published net, 128x192): the loss goes down in the bf16 configuration as it does in fp32 --
    bf16 tensors + fp32 master weights / BatchNorm statistics / optimizer keep the optimisation on track (every step within 5 % of
    the fp32 run's loss, the same overall decrease).'''
    from rcf_amd import synth, train
    cb = synth.make_batch(2, 128, 192, 16, seed=5)
    b = {k: v.cuda() for k, v in cb.items()}
    traj = {}
    for mode in ('fp32', 'bf16'):
        m = train.build_model(synth.PUBLISHED, device='cuda')
        synth.fill_state_dict_([m.encoder, m.decoder], 3)
        m.compute_dtype = mode
        opt = train.make_optimizer(m, lr=1e-3)
        m.train()
        traj[mode] = [float(train.train_step(m, opt, b['image'], b['input_depth'], b['ground_truth'], b['lidar_map'])[0].detach())
                      for _ in range(10)]
    f, h = np.array(traj['fp32']), np.array(traj['bf16'])
    print('loss fp32 %s\nloss bf16 %s' % (np.round(f, 3), np.round(h, 3)))
    assert f[-1] < 0.8 * f[0] and h[-1] < 0.8 * h[0]
    assert np.max(np.abs(h - f) / f) < 0.05


@pytest.mark.parametrize('c1,co,h,w', [(64, 32, 35, 51), (64, 64, 24, 40), (128, 64, 15, 25), (256, 256, 8, 13)])
def test_up2x_weight_gradient_four_phases_in_one_launch_bf16(ops, c1, co, h, w):
    '''rcf_conv2d_wgrad on the merged up-2x descriptor (phase_sum == 2), bf16 tensors: the four phase weight gradients from one launch,
    equal to the four per-phase calls within fp32 summation order (the products are the same bf16 x bf16 ones).'''
    ops.set_precision('bf16')
    n = 2
    x = b16(rnd(n, c1, h, w, seed=31))
    dz = b16(rnd(n, co, 2 * h, 2 * w, seed=32, scale=1e-2))
    xg, dzg = nhwc_b(x), nhwc_b(dz)
    dwp = torch.full((4, co, c1, 2, 2), float('nan'), device='cuda')
    for ph in range(4):
        d = ops.make_up2x_fwd_desc(n, h, w, c1, co, ph >> 1, ph & 1)
        qi = ops.conv_query(d)
        ws = torch.empty(max(1, qi.wgrad_workspace_floats), device='cuda')
        ops.conv_wgrad(d, xg, None, dzg, dwp[ph], ws)
    dm = ops.make_up2x_fwd_desc(n, h, w, c1, co, 0, 0, phase_out=True)
    qm = ops.conv_query(dm)
    wsm = torch.full((max(1, qm.wgrad_workspace_floats),), float('nan'), device='cuda')
    dwm = torch.full((4, co, c1, 2, 2), float('nan'), device='cuda')
    ops.conv_wgrad(dm, xg, None, dzg, dwm, wsm)
    torch.cuda.synchronize()
    assert not torch.isnan(dwm).any()
    e = float((dwm.double() - dwp.double()).abs().max() / dwp.double().abs().max())
    assert e < 2e-6, e


@pytest.mark.parametrize('cin,cout,h,w', [(32, 64, 45, 80), (64, 128, 35, 51), (128, 256, 29, 50), (256, 256, 8, 6)])
def test_stride2_input_gradient_four_phases_from_one_staged_tile_bf16(ops, cin, cout, h, w):
    '''rcf_conv_desc.phase_sum == 3 on bf16 tensors (conv_b16_kernel<DmaCfg<2, ., 32, ., 1, 2>>): dz staged once per chunk, the nine
    (phase, tap) products that exist; bitwise the four per-phase launches, with and without accumulation, odd and even extents.'''
    from rcf_amd._lib import RCF_PHASE_S2_DGRAD
    ops.set_precision('bf16')
    n = 2
    wt = rnd(cout, cin, 3, 3, seed=2, scale=1.0 / np.sqrt(cin * 9))
    ho, wo = (h + 2 - 3) // 2 + 1, (w + 2 - 3) // 2 + 1
    dz = b16(rnd(n, cout, ho, wo, seed=3))
    ref = torch.nn.grad.conv2d_input((n, cin, h, w), wt.double(), dz.double(), stride=2, padding=1)
    fwd = ops.make_fwd_desc(n, h, w, cin, 0, cout, 3, 2)
    wd = ops.phase_weights(wt.cuda(), RCF_PHASE_S2_DGRAD)
    dzg = nhwc_b(dz)
    for accumulate in (False, True):
        base = b16(rnd(n, cin, h, w, seed=7)) if accumulate else None
        dx4 = nhwc_b(base) if accumulate else torch.full((n, h, w, cin), float('nan'), device='cuda').bfloat16()
        dx1 = dx4.clone()
        for ph in range(4):
            dd = ops.make_s2_dgrad_desc(fwd, ph >> 1, ph & 1, accumulate)
            info = ops.conv_query(dd)
            packed = torch.empty(info.packed_weight_floats, device='cuda')
            ops.conv_pack(dd, wd[ph], packed)
            ops.conv_fwd(dd, dzg, None, packed, dx4, None)
        dm = ops.make_s2_dgrad_desc(fwd, 0, 0, accumulate, phase_out=True)
        im = ops.conv_query(dm)
        pm = torch.empty(4 * im.packed_weight_floats, device='cuda')
        for ph in range(4):
            ops.conv_pack(dm, wd[ph], pm[ph * im.packed_weight_floats:(ph + 1) * im.packed_weight_floats])
        ops.conv_fwd(dm, dzg, None, pm, dx1, None)
        torch.cuda.synchronize()
        assert not torch.isnan(dx1.float()).any()
        assert torch.equal(dx1, dx4), accumulate
        if not accumulate:
            e = float((nchw(dx1).double() - ref).abs().max() / ref.abs().max())
            assert e < 3 * BF16_EPS, e


@pytest.mark.parametrize('cin,cout,h,w', [(32, 64, 45, 80), (64, 128, 35, 51), (256, 256, 15, 25)])
def test_stride2_weight_gradient_four_phases_in_one_launch_bf16(ops, cin, cout, h, w):
    '''rcf_conv2d_wgrad on the phase_sum == 1 descriptor, bf16 tensors: one launch, equal to the four per-phase calls within fp32
    summation order.'''
    ops.set_precision('bf16')
    n = 2
    ho, wo = (h - 1) // 2 + 1, (w - 1) // 2 + 1
    x = b16(rnd(n, cin, h, w, seed=51))
    dz = b16(rnd(n, cout, ho, wo, seed=52, scale=1e-2))
    fwd = ops.make_fwd_desc(n, h, w, cin, 0, cout, 3, 2)
    xg, dzg = nhwc_b(x), nhwc_b(dz)
    dwp = torch.full((4, cout, cin, 2, 2), float('nan'), device='cuda')
    for ph in range(4):
        d = ops.make_s2_wgrad_desc(fwd, ph >> 1, ph & 1)
        qi = ops.conv_query(d)
        ws = torch.empty(max(1, qi.wgrad_workspace_floats), device='cuda')
        ops.conv_wgrad(d, xg, None, dzg, dwp[ph], ws)
    dm = ops.make_s2_wgrad_desc(fwd, 0, 0, all_phases=True)
    qm = ops.conv_query(dm)
    wsm = torch.full((max(1, qm.wgrad_workspace_floats),), float('nan'), device='cuda')
    dwm = torch.full((4, cout, cin, 2, 2), float('nan'), device='cuda')
    ops.conv_wgrad(dm, xg, None, dzg, dwm, wsm)
    torch.cuda.synchronize()
    assert not torch.isnan(dwm).any()
    e = float((dwm.double() - dwp.double()).abs().max() / dwp.double().abs().max())
    assert e < 2e-6, e
