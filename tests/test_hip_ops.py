'''
GPU parity tests, kernel by kernel, THROUGH THE C ABI (rcf_amd.ops -> librcf_hip.so) against stock fp32
PyTorch CPU ops on the same seeded inputs.  Tolerance: 2e-4 of the reference tensor's max-abs (fp32 summation
order is the only difference: the MFMA is an exact fmaf chain); north_star's bar is 1e-3.
'''

import numpy as np
import pytest
import torch
import torch.nn.functional as F

pytestmark = pytest.mark.gpu

TOL = 2e-4


@pytest.fixture(scope='module')
def ops():
    import rcf_amd  # noqa: F401
    from rcf_amd import _lib, ops as _ops
    assert torch.cuda.is_available(), 'GPU tests need a GPU'
    _lib.load()
    assert _lib.load().rcf_device_ok() == 1, 'librcf_hip.so: no gfx950 device'
    return _ops


def dev(t):
    return t.cuda()


def nhwc(t):
    return t.permute(0, 2, 3, 1).contiguous().cuda()


def nchw(t):
    return t.detach().cpu().permute(0, 3, 1, 2).contiguous()


def rel(a, b):
    return float((a - b).abs().max() / (b.abs().max() + 1e-30))


def rnd(*shape, seed=0, scale=1.0):
    g = torch.Generator().manual_seed(seed)
    return (torch.rand(*shape, generator=g) * 2 - 1) * scale


# (ksize, stride, c1, c2, cout, n, h, w, up_from)
CONV_CASES = [
    (3, 1, 16, 0, 32, 2, 20, 37, None),
    (3, 1, 64, 0, 64, 1, 33, 64, None),
    (3, 1, 64, 32, 64, 2, 17, 40, None),        # decoder concat 64+32 -> 64 (deconv1.conv)
    (3, 1, 32, 0, 32, 1, 40, 100, None),        # w=100 picks the 16x16 tile
    (3, 1, 8, 8, 8, 2, 35, 51, None),           # tiny net concat, CK=8 path
    (3, 1, 4, 0, 4, 2, 70, 102, None),          # tiny net deconv0
    (3, 1, 256, 0, 128, 1, 29, 50, (15, 25)),   # UpConv 15x25 -> 29x50 (non-2x nearest)
    (3, 1, 64, 0, 32, 1, 70, 102, (35, 51)),    # UpConv exact 2x
    (3, 1, 128, 128, 128, 1, 15, 26, None),     # 256 -> 128 concat
    (3, 2, 32, 0, 64, 2, 45, 80, None),         # stride 2 (blocks3.0.conv1)
    (3, 2, 8, 0, 16, 2, 35, 51, None),
    (3, 2, 128, 0, 256, 1, 29, 50, None),
    (1, 1, 16, 0, 32, 2, 35, 51, None),         # fusion 1x1
    (1, 1, 128, 0, 256, 1, 15, 25, None),
    (1, 2, 32, 0, 64, 2, 45, 80, None),         # projection 1x1 stride 2
    (1, 2, 8, 0, 16, 1, 35, 51, None),
    (7, 2, 3, 0, 32, 2, 70, 102, None),         # image stem
    (7, 2, 2, 0, 16, 2, 70, 102, None),         # depth stem
    (7, 2, 2, 0, 4, 1, 64, 96, None),           # tiny depth stem
]


def _conv_case(case, seed):
    k, s, c1, c2, co, n, h, w, up = case
    hs, ws = (h, w) if up is None else up
    x1 = rnd(n, c1, hs, ws, seed=seed)
    x2 = rnd(n, c2, h, w, seed=seed + 1) if c2 else None
    wt = rnd(co, c1 + c2, k, k, seed=seed + 2, scale=1.0 / np.sqrt((c1 + c2) * k * k))
    return x1, x2, wt


def _ref_conv(case, x1, x2, wt):
    k, s, c1, c2, co, n, h, w, up = case
    xin = x1 if up is None else F.interpolate(x1, size=(h, w))
    if x2 is not None:
        xin = torch.cat([xin, x2], 1)
    return F.conv2d(xin, wt, stride=s, padding=k // 2)


def _desc(ops, case):
    k, s, c1, c2, co, n, h, w, up = case
    hs, ws = (h, w) if up is None else up
    gather = 0 if up is None else 1
    return ops.make_fwd_desc(n, h, w, c1, c2, co, k, s, hs, ws, gather)


@pytest.mark.parametrize('case', CONV_CASES, ids=[str(c) for c in CONV_CASES])
def test_conv_forward_and_bn_statistics(ops, case):
    x1, x2, wt = _conv_case(case, 10)
    ref = _ref_conv(case, x1, x2, wt)
    d = _desc(ops, case)
    info = ops.conv_query(d)
    packed = torch.empty(info.packed_weight_floats, device='cuda')
    ops.conv_pack(d, dev(wt), packed)
    out = torch.full((d.n, d.h_out, d.w_out, d.c_out), float('nan'), device='cuda')
    partials = torch.full((info.n_partials, 2, d.c_out), float('nan'), device='cuda', dtype=torch.float64)
    ops.conv_fwd(d, nhwc(x1), None if x2 is None else nhwc(x2), packed, out, partials)
    torch.cuda.synchronize()
    got = nchw(out)
    assert got.shape == ref.shape
    assert rel(got, ref) < TOL
    s = partials.sum(0).cpu()
    assert rel(s[0], ref.double().sum((0, 2, 3))) < 1e-3 or float((s[0] - ref.double().sum((0, 2, 3))).abs().max()) < 1e-2
    assert rel(s[1], (ref.double() ** 2).sum((0, 2, 3))) < TOL


@pytest.mark.parametrize('case', [c for c in CONV_CASES if c[0] != 7], ids=[str(c) for c in CONV_CASES if c[0] != 7])
def test_conv_input_gradient(ops, case):
    k, s, c1, c2, co, n, h, w, up = case
    x1, x2, wt = _conv_case(case, 20)
    x1.requires_grad_(True)
    if x2 is not None:
        x2.requires_grad_(True)
    ref = _ref_conv(case, x1, x2, wt)
    dz = rnd(*ref.shape, seed=33)
    (ref * dz).sum().backward()
    d = _desc(ops, case)
    dzg = nhwc(dz)
    for src, off, cnt in ((x1, 0, c1), (x2, c1, c2)):
        if src is None:
            continue
        for accumulate in (False, True):
            dd = ops.make_dgrad_desc(d, off, cnt, accumulate and not (src is x1 and up is not None))
            info = ops.conv_query(dd)
            packed = torch.empty(info.packed_weight_floats, device='cuda')
            ops.conv_pack(dd, dev(wt), packed)
            base = rnd(n, cnt, h, w, seed=5) if accumulate else torch.zeros(n, cnt, h, w)
            if src is x1 and up is not None:
                tmp = torch.full((n, h, w, cnt), float('nan'), device='cuda')
                ops.conv_fwd(dd, dzg, None, packed, tmp, None)
                base = rnd(n, cnt, up[0], up[1], seed=5) if accumulate else torch.zeros(n, cnt, up[0], up[1])
                dst = nhwc(base) if accumulate else torch.full((n, up[0], up[1], cnt), float('nan'), device='cuda')
                ops.upsample_nearest_bwd(tmp, dst, accumulate)
            else:
                dst = nhwc(base) if accumulate else torch.full((n, h, w, cnt), float('nan'), device='cuda')
                ops.conv_fwd(dd, dzg, None, packed, dst, None)
            torch.cuda.synchronize()
            want = src.grad + base
            assert rel(nchw(dst), want) < TOL, (off, accumulate)


@pytest.mark.parametrize('case', CONV_CASES, ids=[str(c) for c in CONV_CASES])
def test_conv_weight_gradient(ops, case):
    x1, x2, wt = _conv_case(case, 30)
    wt.requires_grad_(True)
    ref = _ref_conv(case, x1, x2, wt)
    dz = rnd(*ref.shape, seed=44)
    (ref * dz).sum().backward()
    d = _desc(ops, case)
    info = ops.conv_query(d)
    ws = torch.empty(max(1, info.wgrad_workspace_floats), device='cuda')
    dw = torch.full(tuple(wt.shape), float('nan'), device='cuda')
    ops.conv_wgrad(d, nhwc(x1), None if x2 is None else nhwc(x2), nhwc(dz), dw, ws)
    torch.cuda.synchronize()
    assert rel(dw.cpu(), wt.grad) < TOL


@pytest.mark.parametrize('c,n,h,w,has_res', [(32, 2, 19, 23, False), (64, 1, 30, 17, True), (4, 2, 35, 51, True),
                                             (256, 2, 8, 13, False), (16, 3, 9, 9, True)])
def test_batchnorm_leakyrelu_residual_forward_backward(ops, c, n, h, w, has_res):
    z = rnd(n, c, h, w, seed=1, scale=2.0) + 0.3
    res = rnd(n, c, h, w, seed=2) if has_res else None
    gamma = rnd(c, seed=3) * 0.5 + 1.0
    beta = rnd(c, seed=4) * 0.1
    rm0, rv0 = rnd(c, seed=5) * 0.1, rnd(c, seed=6) * 0.25 + 1.0
    dout = rnd(n, c, h, w, seed=7)

    zz = z.clone().requires_grad_(True)
    g, b = gamma.clone().requires_grad_(True), beta.clone().requires_grad_(True)
    rr = res.clone().requires_grad_(True) if has_res else None
    rm, rv = rm0.clone(), rv0.clone()
    y = F.leaky_relu(F.batch_norm(zz, rm, rv, g, b, True, 0.1, 1e-5), 0.2)
    if has_res:
        y = F.leaky_relu(y + rr, 0.2)
    (y * dout).sum().backward()

    n_pix = n * h * w
    zg = nhwc(z)
    # statistics partials as the conv epilogue would produce them (one partial row)
    part = torch.stack([zg.view(-1, c).double().sum(0), (zg.view(-1, c).double() ** 2).sum(0)]).view(1, 2, c).contiguous()
    coef = torch.empty(4, c, device='cuda')
    rmg, rvg = dev(rm0.clone()), dev(rv0.clone())
    ops.bn_finalize(part, 1, c, n_pix, dev(gamma), dev(beta), rmg, rvg, 0.1, 1e-5, True, coef)
    out = torch.empty_like(zg)
    resg = nhwc(res) if has_res else None
    ops.bn_act_fwd(zg, coef, resg, out, n_pix, c, 1)
    torch.cuda.synchronize()
    assert rel(nchw(out), y.detach()) < TOL
    assert rel(rmg.cpu(), rm) < 1e-5 and rel(rvg.cpu(), rv) < 1e-5

    nb = ops.ew_blocks(n_pix, c)
    bpart = torch.empty(nb, 2, c, device='cuda', dtype=torch.float64)
    doutg = nhwc(dout)
    ops.bn_act_bwd_reduce(doutg, zg, coef, out, bpart, n_pix, c, 1, has_res)
    bcoef = torch.empty(2, c, device='cuda')
    dgamma, dbeta = torch.empty(c, device='cuda'), torch.empty(c, device='cuda')
    ops.bn_bwd_finalize(bpart, nb, 2 * c, c, n_pix, bcoef, dgamma, dbeta)
    dz = torch.empty_like(zg)
    dres = nhwc(torch.ones(n, c, h, w)) if has_res else None
    ops.bn_act_bwd_apply(doutg, zg, coef, out, bcoef, dz, dres, True, n_pix, c, 1, has_res)
    torch.cuda.synchronize()
    assert rel(nchw(dz), zz.grad) < TOL
    assert rel(dgamma.cpu(), g.grad) < TOL and rel(dbeta.cpu(), b.grad) < TOL
    if has_res:
        assert rel(nchw(dres), rr.grad + 1.0) < TOL

    # eval mode: coefficients from the running statistics
    coef_e = torch.empty(4, c, device='cuda')
    ops.bn_finalize(None, 0, c, n_pix, dev(gamma), dev(beta), dev(rm0.clone()), dev(rv0.clone()), 0.1, 1e-5, False, coef_e)
    out_e = torch.empty_like(zg)
    ops.bn_act_fwd(zg, coef_e, None, out_e, n_pix, c, 0)
    want = F.batch_norm(z, rm0.clone(), rv0.clone(), gamma, beta, False, 0.1, 1e-5)
    assert rel(nchw(out_e), want) < TOL


@pytest.mark.parametrize('c,n,h,w', [(32, 2, 17, 21), (8, 2, 35, 51), (256, 1, 15, 25)])
def test_weight_and_project_fusion_forward_backward(ops, c, n, h, w):
    zw, zp, img, dout = (rnd(n, c, h, w, seed=s, scale=sc) for s, sc in ((1, 2.0), (2, 1.5), (3, 1.0), (4, 1.0)))
    gw, bw, gp, bp = rnd(c, seed=5) * 0.5 + 1, rnd(c, seed=6) * 0.1, rnd(c, seed=7) * 0.5 + 1, rnd(c, seed=8) * 0.1
    leaves = [t.clone().requires_grad_(True) for t in (zw, zp, img, gw, bw, gp, bp)]
    a, b, i_, g1, b1, g2, b2 = leaves
    run = lambda z, g, bb: F.batch_norm(z, torch.zeros(c), torch.ones(c), g, bb, True, 0.1, 1e-5)
    y = torch.sigmoid(run(a, g1, b1)) * run(b, g2, b2) + i_
    (y * dout).sum().backward()

    n_pix = n * h * w
    zwg, zpg, imgg, doutg = nhwc(zw), nhwc(zp), nhwc(img), nhwc(dout)

    def coef_of(zg, g, bb):
        part = torch.stack([zg.view(-1, c).double().sum(0), (zg.view(-1, c).double() ** 2).sum(0)]).view(1, 2, c).contiguous()
        coef = torch.empty(4, c, device='cuda')
        ops.bn_finalize(part, 1, c, n_pix, dev(g), dev(bb), torch.zeros(c, device='cuda'), torch.ones(c, device='cuda'),
                        0.1, 1e-5, True, coef)
        return coef
    cw, cp = coef_of(zwg, gw, bw), coef_of(zpg, gp, bp)
    out = torch.empty_like(zwg)
    ops.fuse_fwd(zwg, cw, zpg, cp, imgg, out, n_pix, c)
    torch.cuda.synchronize()
    assert rel(nchw(out), y.detach()) < TOL

    nb = ops.ew_blocks(n_pix, c)
    bpart = torch.empty(nb, 4, c, device='cuda', dtype=torch.float64)
    ops.fuse_bwd_reduce(doutg, zwg, cw, zpg, cp, bpart, n_pix, c)
    bcw, bcp = torch.empty(2, c, device='cuda'), torch.empty(2, c, device='cuda')
    dgw, dbw, dgp, dbp = (torch.empty(c, device='cuda') for _ in range(4))
    ops.bn_bwd_finalize(bpart, nb, 4 * c, c, n_pix, bcw, dgw, dbw)
    ops.bn_bwd_finalize(bpart.view(-1)[2 * c:], nb, 4 * c, c, n_pix, bcp, dgp, dbp)
    dzw, dzp = torch.empty_like(zwg), torch.empty_like(zpg)
    dimg = torch.empty_like(imgg)
    ops.fuse_bwd_apply(doutg, zwg, cw, zpg, cp, bcw, bcp, dzw, dzp, dimg, False, n_pix, c)
    torch.cuda.synchronize()
    assert rel(nchw(dzw), a.grad) < 5e-4 and rel(nchw(dzp), b.grad) < TOL and rel(nchw(dimg), i_.grad) < 1e-6
    assert rel(dgw.cpu(), g1.grad) < 5e-4 and rel(dbw.cpu(), b1.grad) < 5e-4
    assert rel(dgp.cpu(), g2.grad) < TOL and rel(dbp.cpu(), b2.grad) < TOL


@pytest.mark.parametrize('c,n,h,w', [(32, 2, 35, 51), (16, 1, 450 // 10, 80), (4, 2, 7, 9), (8, 1, 1, 5)])
def test_maxpool_forward_backward(ops, c, n, h, w):
    x = rnd(n, c, h, w, seed=1).requires_grad_(True)
    y = F.max_pool2d(x, 3, 2, 1)
    dy = rnd(*y.shape, seed=2)
    (y * dy).sum().backward()
    out = torch.empty(n, y.shape[2], y.shape[3], c, device='cuda')
    idx = torch.empty(n, y.shape[2], y.shape[3], c, dtype=torch.uint8, device='cuda')
    ops.maxpool_fwd(nhwc(x.detach()), out, idx)
    din = nhwc(torch.ones(n, c, h, w))
    ops.maxpool_bwd(nhwc(dy), idx, din, True)
    torch.cuda.synchronize()
    assert torch.equal(nchw(out), y.detach())
    assert rel(nchw(din), x.grad + 1.0) < 1e-6


@pytest.mark.parametrize('hs,ws,hu,wu', [(15, 25, 29, 50), (35, 51, 70, 102), (57, 100, 113, 200), (3, 4, 3, 4), (2, 3, 7, 11)])
def test_nearest_upsample_backward(ops, hs, ws, hu, wu):
    n, c = 2, 8
    x = rnd(n, c, hs, ws, seed=1).requires_grad_(True)
    y = F.interpolate(x, size=(hu, wu))
    dy = rnd(n, c, hu, wu, seed=2)
    (y * dy).sum().backward()
    dsrc = torch.full((n, hs, ws, c), float('nan'), device='cuda')
    ops.upsample_nearest_bwd(nhwc(dy), dsrc, False)
    torch.cuda.synchronize()
    assert rel(nchw(dsrc), x.grad) < 1e-6


@pytest.mark.parametrize('c,n,h,w', [(32, 2, 33, 47), (4, 2, 70, 102), (8, 1, 5, 3)])
def test_output_head_forward_backward(ops, c, n, h, w):
    x = rnd(n, c, h, w, seed=1).requires_grad_(True)
    wt = (rnd(1, c, 3, 3, seed=2) * 0.3).requires_grad_(True)
    o = F.conv2d(x, wt, padding=1)
    d = 1.0 / (torch.sigmoid(o) + 1.0 / 100.0)
    dd = rnd(n, 1, h, w, seed=3)
    (d * dd).sum().backward()
    xg = nhwc(x.detach())
    logit, depth = torch.empty(n, h, w, device='cuda'), torch.empty(n, h, w, device='cuda')
    ops.head_fwd(xg, dev(wt.detach()), logit, depth, 1.0, 100.0)
    dlogit = torch.empty_like(logit)
    ops.head_bwd_logit(dev(dd.view(n, h, w).contiguous()), logit, dlogit, 1.0, 100.0)
    dx = torch.empty_like(xg)
    ops.head_bwd_dgrad(dlogit, dev(wt.detach()), dx)
    dw = torch.empty(1, c, 3, 3, device='cuda')
    ops.head_bwd_wgrad(xg, dlogit, dw)
    torch.cuda.synchronize()
    assert rel(logit.cpu(), o.detach().view(n, h, w)) < TOL
    assert rel(depth.cpu(), d.detach().view(n, h, w)) < TOL
    assert rel(nchw(dx), x.grad) < TOL
    assert rel(dw.cpu(), wt.grad) < TOL


def test_masked_l1_loss_forward_backward(ops):
    from oracle.fusionnet_oracle import FusionNetOracle
    n, h, w = 2, 37, 53
    g = torch.Generator().manual_seed(3)
    d = (torch.rand(n, 1, h, w, generator=g) * 60 + 1).requires_grad_(True)
    gt = torch.rand(n, 1, h, w, generator=g) * 80 * (torch.rand(n, 1, h, w, generator=g) < 0.3)
    lidar = torch.rand(n, 1, h, w, generator=g) * 80 * (torch.rand(n, 1, h, w, generator=g) < 0.05)
    o = FusionNetOracle.__new__(FusionNetOracle)
    loss, ls, ll = FusionNetOracle.compute_loss(o, d, gt, lidar, 2.0)
    loss.backward()
    sums = torch.empty(4, dtype=torch.float64, device='cuda')
    dg, gtg, lg = dev(d.detach()), dev(gt), dev(lidar)
    ops.l1_loss_fwd(dg, gtg, lg, sums)
    val = torch.empty(3, device='cuda')
    ops.l1_loss_value(sums, 2.0, val)
    dd = torch.empty_like(dg)
    ops.l1_loss_bwd(dg, gtg, lg, sums, None, 2.0, dd)
    torch.cuda.synchronize()
    np.testing.assert_allclose(val.cpu().numpy(), [float(loss), float(ls), float(ll)], rtol=1e-5)
    assert rel(dd.cpu(), d.grad) < 1e-5


def test_adam_matches_torch_optim(ops):
    p0, steps = rnd(10007, seed=1), 4
    ref = torch.nn.Parameter(p0.clone())
    opt = torch.optim.Adam([{'params': [ref], 'weight_decay': 0.01}], lr=1e-2)
    p, m, v = dev(p0.clone()), torch.zeros(10007, device='cuda'), torch.zeros(10007, device='cuda')
    for s in range(1, steps + 1):
        g = rnd(10007, seed=10 + s)
        ref.grad = g.clone()
        opt.step()
        ops.adam_step(p, dev(g), m, v, 1e-2, 0.9, 0.999, 1e-8, 0.01, s)
    torch.cuda.synchronize()
    assert rel(p.cpu(), ref.detach()) < 1e-5


def test_layout_round_trip(ops):
    x = rnd(2, 3, 11, 7, seed=1)
    y = ops.nchw_to_nhwc(dev(x))
    assert torch.equal(y.cpu(), x.permute(0, 2, 3, 1).contiguous())
    assert torch.equal(ops.nhwc_to_nchw(y).cpu(), x)


def test_unsupported_shapes_return_errors(ops):
    from rcf_amd import _lib
    with pytest.raises(_lib.RcfError):
        ops.conv_query(ops.make_fwd_desc(1, 8, 8, 16, 0, 1, 3, 1))       # c_out == 1 belongs to the head kernel
    with pytest.raises(_lib.RcfError):
        ops.conv_query(ops.make_fwd_desc(1, 8, 8, 16, 0, 16, 5, 1))      # 5x5 does not exist in FusionNet
    with pytest.raises(_lib.RcfError):
        ops.bn_act_fwd(torch.zeros(4, device='cuda'), torch.zeros(16, device='cuda'), None, torch.zeros(4, device='cuda'), 1, 6, 1)
    with pytest.raises(_lib.RcfError):
        ops.conv_fwd(ops.make_fwd_desc(1, 8, 8, 16, 0, 16, 3, 1), torch.zeros(4), None, torch.zeros(4), torch.zeros(4))  # CPU tensors


@pytest.mark.parametrize('cin,cout,n,hs,ws', [(64, 32, 1, 35, 51), (64, 64, 2, 12, 20), (8, 4, 2, 35, 51), (32, 32, 1, 9, 33)])
def test_up2x_conv_as_four_phase_convs(ops, cin, cout, n, hs, ws):
    '''conv3x3(F.interpolate(x, exactly 2x)) == four 2x2 phase convs on x: forward (+BN statistics), dX, dW.'''
    from rcf_amd._lib import RCF_PHASE_UP2X_DGRAD, RCF_PHASE_UP2X_FWD
    x = rnd(n, cin, hs, ws, seed=1).requires_grad_(True)
    wt = rnd(cout, cin, 3, 3, seed=2, scale=1.0 / np.sqrt(cin * 9)).requires_grad_(True)
    ref = F.conv2d(F.interpolate(x, size=(2 * hs, 2 * ws)), wt, padding=1)
    dz = rnd(*ref.shape, seed=3)
    (ref * dz).sum().backward()
    xg, dzg, wg = nhwc(x.detach()), nhwc(dz), dev(wt.detach())

    wp = ops.phase_weights(wg, RCF_PHASE_UP2X_FWD)
    z = torch.full((n, 2 * hs, 2 * ws, cout), float('nan'), device='cuda')
    descs, parts = [], []
    for ph in range(4):
        d = ops.make_up2x_fwd_desc(n, hs, ws, cin, cout, ph >> 1, ph & 1)
        info = ops.conv_query(d)
        packed = torch.empty(info.packed_weight_floats, device='cuda')
        ops.conv_pack(d, wp[ph], packed)
        part = torch.full((info.n_partials, 2, cout), float('nan'), device='cuda', dtype=torch.float64)
        ops.conv_fwd(d, xg, None, packed, z, part)
        descs.append(d); parts.append(part)
    torch.cuda.synchronize()
    assert rel(nchw(z), ref.detach()) < TOL
    s = torch.cat(parts).sum(0).cpu()
    assert rel(s[1], (ref.detach().double() ** 2).sum((0, 2, 3))) < TOL

    dwp = torch.full((4, cout, cin, 2, 2), float('nan'), device='cuda')
    for ph, d in enumerate(descs):
        info = ops.conv_query(d)
        ws_ = torch.empty(max(1, info.wgrad_workspace_floats), device='cuda')
        ops.conv_wgrad(d, xg, None, dzg, dwp[ph], ws_)
    dw = torch.full((cout, cin, 3, 3), float('nan'), device='cuda')
    ops.phase_wgrad_fold(dwp, dw)
    torch.cuda.synchronize()
    assert rel(dw.cpu(), wt.grad) < TOL

    wd = ops.phase_weights(wg, RCF_PHASE_UP2X_DGRAD)
    base = rnd(n, cin, hs, ws, seed=9)
    dx = nhwc(base)
    for ph in range(4):
        dd = ops.make_up2x_dgrad_desc(n, hs, ws, cin, cout, ph >> 1, ph & 1, True)
        info = ops.conv_query(dd)
        packed = torch.empty(info.packed_weight_floats, device='cuda')
        ops.conv_pack(dd, wd[ph], packed)
        ops.conv_fwd(dd, dzg, None, packed, dx, None)
    torch.cuda.synchronize()
    assert rel(nchw(dx), x.grad + base) < TOL

    # the same input gradient with the four phases summed inside ONE launch
    for accumulate in (False, True):
        dd = ops.make_up2x_dgrad_desc(n, hs, ws, cin, cout, 0, 0, accumulate, phase_sum=True)
        info = ops.conv_query(dd)
        packed = torch.empty(4 * info.packed_weight_floats, device='cuda')
        for ph in range(4):
            ops.conv_pack(dd, wd[ph], packed[ph * info.packed_weight_floats:(ph + 1) * info.packed_weight_floats])
        dx1 = nhwc(base) if accumulate else torch.full((n, hs, ws, cin), float('nan'), device='cuda')
        ops.conv_fwd(dd, dzg, None, packed, dx1, None)
        torch.cuda.synchronize()
        assert rel(nchw(dx1), x.grad + (base if accumulate else 0)) < TOL, accumulate


@pytest.mark.parametrize('cin,cout,n,h,w', [(32, 64, 2, 45, 80), (8, 16, 2, 35, 51), (128, 256, 1, 29, 50), (64, 128, 1, 8, 6)])
def test_stride2_input_gradient_as_four_phase_convs(ops, cin, cout, n, h, w):
    '''The transposed convolution (dX of a 3x3 stride-2 conv) in 4 phases, odd and even extents.'''
    from rcf_amd._lib import RCF_PHASE_S2_DGRAD
    x = rnd(n, cin, h, w, seed=1).requires_grad_(True)
    wt = rnd(cout, cin, 3, 3, seed=2, scale=1.0 / np.sqrt(cin * 9))
    ref = F.conv2d(x, wt, stride=2, padding=1)
    dz = rnd(*ref.shape, seed=3)
    (ref * dz).sum().backward()
    fwd = ops.make_fwd_desc(n, h, w, cin, 0, cout, 3, 2)
    wd = ops.phase_weights(dev(wt), RCF_PHASE_S2_DGRAD)
    for accumulate in (False, True):
        base = rnd(n, cin, h, w, seed=7) if accumulate else torch.zeros(n, cin, h, w)
        dx = nhwc(base) if accumulate else torch.full((n, h, w, cin), float('nan'), device='cuda')
        for ph in range(4):
            dd = ops.make_s2_dgrad_desc(fwd, ph >> 1, ph & 1, accumulate)
            info = ops.conv_query(dd)
            packed = torch.empty(info.packed_weight_floats, device='cuda')
            ops.conv_pack(dd, wd[ph], packed)
            ops.conv_fwd(dd, nhwc(dz), None, packed, dx, None)
        torch.cuda.synchronize()
        assert rel(nchw(dx), x.grad + base) < TOL, accumulate


@pytest.mark.parametrize('case', [(64, 128, 2, 45, 63), (128, 256, 2, 29, 50), (32, 64, 3, 36, 52), (256, 256, 1, 15, 25)], ids=lambda c: str(c))
def test_1x1_stride2_input_gradient_at_dz_resolution_is_bitwise_the_dense_form(ops, case):
    '''The ResNet projections' input gradient (1x1, stride 2): dX(2y, 2x) = W^T dZ(y, x), zero elsewhere.  ops.make_pw_s2_dgrad_desc runs
    it as a 1x1 convolution of dZ with out_stride 2 into a zero-filled dX (a quarter of the pixels of the zero-dilated dense form the
    engine used before round 6); same kernel, same dot products: bitwise, also when accumulating.  Odd input sizes: the last even row /
    column exists, the one after it does not.'''
    c1, co, n, h, w = case
    x = rnd(n, c1, h, w, seed=50)
    x.requires_grad_(True)
    wt = rnd(co, c1, 1, 1, seed=51, scale=0.2)
    ref = F.conv2d(x, wt, stride=2)
    dz = rnd(*ref.shape, seed=52)
    (ref * dz).sum().backward()
    fwd = ops.make_fwd_desc(n, h, w, c1, 0, co, 1, 2)
    dzg = nhwc(dz)
    for acc in (False, True):
        base = rnd(n, c1, h, w, seed=53) if acc else torch.zeros(n, c1, h, w)
        dd = ops.make_dgrad_desc(fwd, 0, c1, acc)
        info = ops.conv_query(dd)
        packed = torch.empty(info.packed_weight_floats, device='cuda')
        ops.conv_pack(dd, dev(wt), packed)
        dense = nhwc(base) if acc else torch.full((n, h, w, c1), float('nan'), device='cuda')
        ops.conv_fwd(dd, dzg, None, packed, dense, None)
        dl = ops.make_pw_s2_dgrad_desc(fwd, acc)
        il = ops.conv_query(dl)
        packed_l = torch.empty(il.packed_weight_floats, device='cuda')
        ops.conv_pack(dl, dev(wt), packed_l)
        low = nhwc(base) if acc else torch.zeros((n, h, w, c1), device='cuda')
        ops.conv_fwd(dl, dzg, None, packed_l, low, None)
        torch.cuda.synchronize()
        assert rel(nchw(low), x.grad + base) < TOL, acc
        assert torch.equal(low, dense), acc


def test_radar_scatter_matches_reference_golden_and_oracle(ops, golden_dir):
    '''Bit-exact (integer/index work): golden vectors from the real radarnet_main.forward, then larger seeded cases vs the oracle.'''
    import os
    from rcf_amd import synth
    from oracle.radar_scatter_oracle import radar_scatter
    g = np.load(os.path.join(golden_dir, 'T4_radar_scatter.npz'))
    for ci in range(int(g['n_cases'])):
        k, h, w, wc, seed, small_z = [int(v) for v in g['meta%d' % ci]]
        crops, pts = synth.make_scatter_case(k, h, w, wc, seed, bool(small_z))
        depth, resp = ops.radar_scatter(torch.from_numpy(crops).cuda(), torch.from_numpy(pts).cuda(), w, True)
        torch.cuda.synchronize()
        assert np.array_equal(depth.cpu().numpy(), g['depth%d' % ci]), ci
        assert np.array_equal(resp.cpu().numpy(), g['resp%d' % ci]), ci
    for (k, h, w, wc, seed, small_z) in ((64, 225, 400, 72, 11, True), (96, 90, 1600, 288, 12, False), (1, 7, 9, 4, 13, True)):
        crops, pts = synth.make_scatter_case(k, h, w, wc, seed, small_z)
        for strict in (True, False):
            depth, resp = ops.radar_scatter(torch.from_numpy(crops).cuda(), torch.from_numpy(pts).cuda(), w, strict)
            od, orr = radar_scatter(crops, pts, w, strict_reference=strict)
            assert np.array_equal(depth.cpu().numpy(), od) and np.array_equal(resp.cpu().numpy(), orr), (k, h, w, strict)


def test_radar_scatter_from_logits_thresholds_on_the_sign(ops):
    '''rcf_radar_scatter_logits (what pipeline.radarnet_forward calls): sigmoid + the 0.5 threshold of src/radarnet_main.py:563-567 taken
    inside the kernel on the SIGN of the logit.  Against the oracle run on the CPU's torch.sigmoid of the same logits: the integer depth
    map and the kept / dropped pattern must be identical, the responses equal to an ulp of expf; logits placed at 0, +-1e-3 and beyond
    sigmoid's saturation (ties between points: the first one wins, like torch.max).'''
    from oracle.radar_scatter_oracle import radar_scatter
    rs = np.random.RandomState(5)
    for (k, h, w, wc) in ((24, 40, 200, 32), (64, 96, 400, 72)):
        logits = (rs.randn(k, h, wc) * 3.0).astype(np.float32)
        logits[0, 0, :8] = [0.0, 1e-3, -1e-3, 20.0, 25.0, 88.0, -0.0, -30.0]
        logits[1, 0, 3:6] = [20.0, 25.0, 88.0]            # saturated ties with point 0 where the crops overlap
        pts = np.stack([rs.randint(wc // 2, wc // 2 + w, size=k).astype(np.float32), np.zeros(k, np.float32),
                        rs.uniform(1.0, 80.0, size=k).astype(np.float32)], 1)   # x in the PADDED canvas, as the reference passes it
        pts[1, 0] = pts[0, 0]                                # the two points' crops coincide
        depth, resp = ops.radar_scatter(torch.from_numpy(logits).cuda(), torch.from_numpy(pts).cuda(), w, True, logits=True)
        torch.cuda.synchronize()
        cpu_resp = torch.sigmoid(torch.from_numpy(logits)).numpy()
        assert not np.any((logits < 0) & (cpu_resp >= 0.5)), 'test logits must stay out of (-6e-8, 0)'
        od, orr = radar_scatter(cpu_resp, pts, w, strict_reference=True)
        assert np.array_equal(depth.cpu().numpy(), od)
        got = resp.cpu().numpy()
        assert np.array_equal(got > 0, orr > 0)
        assert float(np.abs(got - orr).max()) <= 2.4e-7


@pytest.mark.parametrize('n,h,w,ks,thr', [(2, 37, 53, 7, 1.5), (1, 900 // 4, 1600 // 4, 7, 1.5), (3, 20, 31, 5, 0.5), (1, 9, 70, 3, 2.0)])
def test_outlier_removal_bit_exact(ops, n, h, w, ks, thr):
    '''Comparisons and copies only: bit-exact against the oracle (itself pinned to the reference class by make_golden.py).'''
    from rcf_amd import synth
    from rcf_amd.net_utils import OutlierRemoval
    from oracle.fusionnet_oracle import remove_outliers
    gt = synth.make_batch(n, h, w, 4, seed=5)['ground_truth']
    want = remove_outliers(gt, ks, thr)
    got = OutlierRemoval(ks, thr).remove_outliers(gt.cuda())
    torch.cuda.synchronize()
    assert got.shape == gt.shape and torch.equal(got.cpu(), want)
    assert int((want != gt).sum()) > 0


SPLIT_CASES = [c for c in CONV_CASES if c[0] == 3 and c[1] == 1 and c[8] is None and max(c[2], c[3]) > 8]


@pytest.mark.parametrize('case', SPLIT_CASES, ids=[str(c) for c in SPLIT_CASES])
def test_split_bf16_conv_is_fp32_accurate(ops, case, monkeypatch):
    '''
    The default 3x3 stride-1 forward / input-gradient convolution runs on the bf16 matrix pipe (exact 3-way operand split,
    6 partial products, fp32 accumulate); RCF_CONV_SPLIT=0 selects the exact-f32-MFMA kernel.  The split kernel must be as
    close to an fp64 reference as the f32-MFMA kernel is.
    '''
    k, s, c1, c2, co, n, h, w, up = case
    x1, x2, wt = _conv_case(case, 10)
    xin = x1 if x2 is None else torch.cat([x1, x2], 1)
    ref64 = F.conv2d(xin.double(), wt.double(), padding=1)
    dz = rnd(*ref64.shape, seed=33)
    errs = {}
    for mode in ('0', '1'):
        monkeypatch.setenv('RCF_CONV_SPLIT', mode)
        d = _desc(ops, case)
        info = ops.conv_query(d)
        assert (info.kernel_id >= 5000) == (mode == '1')
        packed = torch.empty(info.packed_weight_floats, device='cuda')
        ops.conv_pack(d, dev(wt), packed)
        out = torch.full((d.n, d.h_out, d.w_out, d.c_out), float('nan'), device='cuda')
        partials = torch.full((info.n_partials, 2, d.c_out), float('nan'), device='cuda', dtype=torch.float64)
        ops.conv_fwd(d, nhwc(x1), None if x2 is None else nhwc(x2), packed, out, partials)
        torch.cuda.synchronize()
        errs['fwd' + mode] = float((nchw(out).double() - ref64).abs().max() / ref64.abs().max())
        s2 = partials.sum(0).cpu()
        assert rel(s2[1], (ref64 ** 2).sum((0, 2, 3))) < 1e-5
        # input gradient of source 1 through the same kernel
        dd = ops.make_dgrad_desc(d, 0, c1, False)
        di = ops.conv_query(dd)
        pd = torch.empty(di.packed_weight_floats, device='cuda')
        ops.conv_pack(dd, dev(wt), pd)
        dx = torch.full((n, h, w, c1), float('nan'), device='cuda')
        ops.conv_fwd(dd, nhwc(dz), None, pd, dx, None)
        torch.cuda.synchronize()
        want = torch.nn.grad.conv2d_input(xin.shape, wt.double(), dz.double(), padding=1)[:, :c1]
        errs['dx' + mode] = float((nchw(dx).double() - want).abs().max() / want.abs().max())
    print('split vs f32-mfma error against fp64:', {k_: '%.2e' % v for k_, v in errs.items()})
    assert errs['fwd1'] < 2.0 * errs['fwd0'] + 1e-7 and errs['dx1'] < 2.0 * errs['dx0'] + 1e-7
    assert errs['fwd1'] < 1e-5 and errs['dx1'] < 1e-5


@pytest.mark.gpu
@pytest.mark.parametrize('c', [4, 32])
def test_head_dgrad_fused_into_bn_backward(ops, c):
    '''rcf_head_bn_bwd_reduce / _apply == rcf_head_bwd_dgrad followed by rcf_bn_act_bwd_reduce / _apply (same sums, same dz).'''
    import torch
    torch.manual_seed(7)
    dev = 'cuda'
    n, h, w = 2, 21, 45
    z = torch.randn(n, h, w, c, device=dev)
    dlogit = torch.randn(n, h, w, device=dev)
    w_head = torch.randn(1, c, 3, 3, device=dev) * 0.2
    coef = torch.stack([torch.rand(c, device=dev) + 0.5, torch.randn(c, device=dev) * 0.1, torch.randn(c, device=dev) * 0.1,
                        torch.rand(c, device=dev) + 0.5]).contiguous()
    n_pix = n * h * w
    # unfused reference path (both HIP): materialise dout, then the generic BN backward
    dout = torch.empty_like(z)
    ops.head_bwd_dgrad(dlogit, w_head, dout)
    nb = ops.ew_blocks(n_pix, c)
    part = torch.empty(nb, 2, c, dtype=torch.float64, device=dev)
    out = torch.empty_like(z)
    ops.bn_act_fwd(z, coef, None, out, n_pix, c, 1)
    ops.bn_act_bwd_reduce(dout, z, coef, out, part, n_pix, c, 1, False)
    bcoef = torch.empty(2, c, device=dev)
    dg, db = torch.empty(c, device=dev), torch.empty(c, device=dev)
    ops.bn_bwd_finalize(part, nb, 2 * c, c, n_pix, bcoef, dg, db)
    dz = torch.empty_like(z)
    ops.bn_act_bwd_apply(dout, z, coef, out, bcoef, dz, None, False, n_pix, c, 1, False)
    # fused path
    nb2 = ops.head_bn_blocks(n, h, w, c)
    assert nb2 > 0
    part2 = torch.empty(nb2, 2, c, dtype=torch.float64, device=dev)
    ops.head_bn_bwd_reduce(dlogit, w_head, z, coef, part2)
    bcoef2 = torch.empty(2, c, device=dev)
    dg2, db2 = torch.empty(c, device=dev), torch.empty(c, device=dev)
    ops.bn_bwd_finalize(part2, nb2, 2 * c, c, n_pix, bcoef2, dg2, db2)
    dz2 = torch.empty_like(z)
    ops.head_bn_bwd_apply(dlogit, w_head, z, coef, bcoef2, dz2)
    np.testing.assert_allclose(dg2.cpu().numpy(), dg.cpu().numpy(), rtol=2e-5, atol=1e-5)
    np.testing.assert_allclose(db2.cpu().numpy(), db.cpu().numpy(), rtol=2e-5, atol=1e-5)
    np.testing.assert_allclose(dz2.cpu().numpy(), dz.cpu().numpy(), rtol=2e-5, atol=2e-6)


@pytest.mark.gpu
@pytest.mark.parametrize('c', [4, 32])
def test_head_applies_previous_bn_on_load(ops, c):
    '''rcf_head_fwd_bn / rcf_head_bwd_wgrad_bn on the raw conv output == rcf_bn_act_fwd followed by the plain head kernels.'''
    import torch
    torch.manual_seed(11)
    dev = 'cuda'
    n, h, w = 2, 19, 37
    z = torch.randn(n, h, w, c, device=dev)
    coef = torch.stack([torch.rand(c, device=dev) + 0.5, torch.randn(c, device=dev) * 0.3, torch.zeros(c, device=dev),
                        torch.ones(c, device=dev)]).contiguous()
    w_head = torch.randn(1, c, 3, 3, device=dev) * 0.2
    y = torch.empty_like(z)
    ops.bn_act_fwd(z, coef, None, y, n * h * w, c, 1)
    logit, depth = torch.empty(n, h, w, device=dev), torch.empty(n, h, w, device=dev)
    logit2, depth2 = torch.empty_like(logit), torch.empty_like(depth)
    ops.head_fwd(y, w_head, logit, depth, 1.0, 100.0)
    ops.head_fwd(z, w_head, logit2, depth2, 1.0, 100.0, coef=coef)
    np.testing.assert_allclose(logit2.cpu().numpy(), logit.cpu().numpy(), rtol=1e-5, atol=1e-5)
    np.testing.assert_allclose(depth2.cpu().numpy(), depth.cpu().numpy(), rtol=1e-5, atol=1e-5)
    dlogit = torch.randn(n, h, w, device=dev)
    dw, dw2 = torch.empty_like(w_head), torch.empty_like(w_head)
    ops.head_bwd_wgrad(y, dlogit, dw)
    ops.head_bwd_wgrad(z, dlogit, dw2, coef=coef)
    np.testing.assert_allclose(dw2.cpu().numpy(), dw.cpu().numpy(), rtol=1e-4, atol=1e-4)


@pytest.mark.gpu
@pytest.mark.parametrize('two_src', [False, True])
def test_conv_applies_producer_bn_on_load(ops, two_src):
    '''rcf_conv2d_fwd_bn / rcf_conv2d_wgrad_bn on raw conv outputs + coefficients == the plain kernels on the materialised
    activations (split kernels; bit-identical operands, so only the summation order inside the kernel could differ: none does).'''
    import os
    import torch
    if os.environ.get('RCF_CONV_SPLIT') == '0':
        pytest.skip('BN-on-load lives in the split kernels, which RCF_CONV_SPLIT=0 turns off')
    torch.manual_seed(5)
    dev = 'cuda'
    n, h, w, c1, c2, co = 2, 37, 53, 32, (16 if two_src else 0), 64
    z1 = torch.randn(n, h, w, c1, device=dev)
    z2 = torch.randn(n, h, w, c2, device=dev) if two_src else None

    def coef(c):
        return torch.stack([torch.rand(c, device=dev) + 0.5, torch.randn(c, device=dev) * 0.3, torch.zeros(c, device=dev),
                            torch.ones(c, device=dev)]).contiguous()
    k1 = coef(c1)
    y1 = torch.empty_like(z1)
    ops.bn_act_fwd(z1, k1, None, y1, n * h * w, c1, 1)
    wt = torch.randn(co, c1 + c2, 3, 3, device=dev) * 0.1
    d = ops.make_fwd_desc(n, h, w, c1, c2, co, 3, 1, h, w, 0)
    info = ops.conv_query(d)
    assert info.bn_on_load == 1 and info.wgrad_bn_on_load == 1
    packed = torch.empty(info.packed_weight_floats, device=dev)
    ops.conv_pack(d, wt, packed)
    out_a, out_b = torch.empty(n, h, w, co, device=dev), torch.empty(n, h, w, co, device=dev)
    # source 2 stays a plain (already materialised) tensor: per-source coefficients
    ops.conv_fwd(d, y1, z2, packed, out_a, None)
    ops.conv_fwd(d, z1, z2, packed, out_b, None, coef1=k1)
    assert torch.equal(out_a, out_b)
    dz = torch.randn(n, h, w, co, device=dev)
    dw_a, dw_b = torch.empty_like(wt), torch.empty_like(wt)
    ws = torch.empty(max(1, info.wgrad_workspace_floats), device=dev)
    ops.conv_wgrad(d, y1, z2, dz, dw_a, ws)
    ops.conv_wgrad(d, z1, z2, dz, dw_b, ws, coef1=k1)
    assert torch.equal(dw_a, dw_b)


@pytest.mark.gpu
@pytest.mark.parametrize('with_res', [False, True], ids=['conv+bn+lrelu', 'resnet tail'])
@pytest.mark.parametrize('shape', [(2, 37, 53, 32, 16, 64), (1, 20, 70, 64, 0, 32), (2, 33, 40, 128, 0, 128)], ids=str)
def test_inference_epilogue_matches_eval_batchnorm_reference(ops, shape, with_res):
    '''rcf_scale_channels + rcf_conv2d_fwd_act == lrelu(BN_eval(conv(x))) (and lrelu(. + res)) of torch in fp32: BatchNorm folded into
    the weights / a bias, activation and residual in the split kernel's epilogue.  Bar 1e-5 relative (folding moves roundings).'''
    import os
    import torch
    import torch.nn.functional as F
    if os.environ.get('RCF_CONV_SPLIT') == '0':
        pytest.skip('the inference epilogue lives in the split kernels, which RCF_CONV_SPLIT=0 turns off')
    torch.manual_seed(11)
    dev = 'cuda'
    n, h, w, c1, c2, co = shape
    x1 = torch.randn(n, c1, h, w)
    x2 = torch.randn(n, c2, h, w) if c2 else None
    wt = torch.randn(co, c1 + c2, 3, 3) / np.sqrt(9 * (c1 + c2))
    gamma, beta = torch.rand(co) + 0.5, torch.randn(co) * 0.2
    mean, var = torch.randn(co) * 0.3, torch.rand(co) + 0.3
    res = torch.randn(n, co, h, w) if with_res else None
    xin = x1 if x2 is None else torch.cat([x1, x2], 1)
    ref = F.leaky_relu(F.batch_norm(F.conv2d(xin.double(), wt.double(), padding=1), mean.double(), var.double(), gamma.double(),
                                    beta.double(), False, 0.1, 1e-5), 0.2)
    if with_res:
        ref = F.leaky_relu(ref + res.double(), 0.2)
    scale = (gamma / torch.sqrt(var + 1e-5)).to(dev)
    bias = (beta - mean * gamma / torch.sqrt(var + 1e-5)).to(dev)
    d = ops.make_fwd_desc(n, h, w, c1, c2, co, 3, 1, h, w, 0)
    info = ops.conv_query(d)
    assert info.fwd_act == 1
    nhwc = lambda t: None if t is None else t.permute(0, 2, 3, 1).contiguous().to(dev)
    folded = ops.scale_channels(wt.to(dev), scale)
    assert torch.equal(folded, wt.to(dev) * scale.view(-1, 1, 1, 1))
    packed = torch.empty(info.packed_weight_floats, device=dev)
    ops.conv_pack(d, folded, packed)
    out = torch.empty(n, h, w, co, device=dev)
    ops.conv_fwd_act(d, nhwc(x1), nhwc(x2), packed, bias, nhwc(res), out)
    assert rel(out.cpu().permute(0, 3, 1, 2).double(), ref) < 1e-5
    # descriptors without a split kernel refuse instead of computing something else
    d1 = ops.make_fwd_desc(n, h, w, c1, 0, co, 1, 1, h, w, 0)
    assert ops.conv_query(d1).fwd_act == 0
    with pytest.raises(Exception):
        ops.conv_fwd_act(d1, nhwc(x1), None, packed, bias, None, out)


# ---------------------------------------------------------------- RadarNet stage-1 ops
@pytest.mark.gpu
def test_roi_pool_forward_backward_matches_restated_torchvision(ops):
    import torch
    from oracle.roi_pool_oracle import roi_pool as roi_pool_ref
    torch.manual_seed(3)
    dev = 'cuda'
    n, c, h, w = 2, 8, 23, 40
    x = torch.randn(n, c, h, w)
    boxes = [torch.tensor([[3.0, 0.0, 19.0, 22.0], [30.0, 0.0, 46.0, 22.0]]), torch.tensor([[-4.0, 0.0, 12.0, 22.0], [10.5, 2.0, 26.5, 20.0]])]
    # (the last two pool sizes are LARGER than the roi at that scale: bins narrower than a pixel, many bins per input pixel)
    for scale, out_hw in ((1.0, (23, 16)), (0.5, (11, 8)), (0.25, (5, 4)), (0.5, (23, 16)), (0.25, (9, 11))):
        hs, ws_ = max(1, int(h * scale)), max(1, int(w * scale))
        xs = torch.nn.functional.interpolate(x, size=(hs, ws_)).clone().requires_grad_(True)
        ref = roi_pool_ref(xs, boxes, out_hw, scale)
        gref = torch.randn_like(ref)
        ref.backward(gref)
        rois = torch.cat([torch.cat([torch.full((b.shape[0], 1), float(i)), b], 1) for i, b in enumerate(boxes)], 0).to(dev)
        x_nhwc = xs.detach().permute(0, 2, 3, 1).contiguous().to(dev)
        ctot, coff = c + 4, 4
        out = torch.zeros(rois.shape[0], out_hw[0], out_hw[1], ctot, device=dev)
        am = torch.empty(rois.shape[0], out_hw[0], out_hw[1], c, dtype=torch.int32, device=dev)
        ops.roi_pool_fwd(x_nhwc, rois, out, am, out_hw, scale, out_coff=coff)
        got = out[..., coff:].permute(0, 3, 1, 2).cpu()
        assert torch.equal(got, ref.detach()), scale        # a max: bit-exact
        assert float(out[..., :coff].abs().max()) == 0.0    # the other channels of the shared buffer are untouched
        dout = torch.zeros_like(out)
        dout[..., coff:] = gref.permute(0, 2, 3, 1).to(dev)
        din = torch.zeros_like(x_nhwc)
        ops.roi_pool_bwd(dout, am, rois, din, out_hw, dout_coff=coff)
        np.testing.assert_allclose(din.permute(0, 3, 1, 2).cpu().numpy(), xs.grad.numpy(), rtol=1e-5, atol=1e-5)
        # the gather form (rcf_roi_pool_bwd_gather): overwrite, accumulate on top of an earlier gradient, bf16 tensors
        ding = torch.full_like(x_nhwc, float('nan'))
        ops.roi_pool_bwd_gather(dout, am, rois, ding, False, out_hw, scale, dout_coff=coff)
        np.testing.assert_allclose(ding.permute(0, 3, 1, 2).cpu().numpy(), xs.grad.numpy(), rtol=1e-5, atol=1e-5)
        prev = torch.randn_like(x_nhwc)
        dacc = prev.clone()
        ops.roi_pool_bwd_gather(dout, am, rois, dacc, True, out_hw, scale, dout_coff=coff)
        np.testing.assert_allclose((dacc - prev).permute(0, 3, 1, 2).cpu().numpy(), xs.grad.numpy(), rtol=1e-5, atol=2e-5)
        db = torch.empty_like(x_nhwc).bfloat16()
        ops.roi_pool_bwd_gather(dout.bfloat16(), am, rois, db, False, out_hw, scale, dout_coff=coff)
        want = torch.zeros_like(x_nhwc)
        ops.roi_pool_bwd(dout.bfloat16().float(), am, rois, want, out_hw, dout_coff=coff)
        assert float((db.float() - want).abs().max()) <= 2.0 ** -8 * float(want.abs().max()) + 1e-6


@pytest.mark.gpu
@pytest.mark.parametrize('m,n_in,n_out,hw', [(4, 3, 32, 1), (24, 128, 2016, 63), (48, 32, 64, 1)])
def test_fully_connected_forward_backward(ops, m, n_in, n_out, hw):
    import torch
    torch.manual_seed(9)
    dev = 'cuda'
    x = torch.randn(m, n_in, requires_grad=True)
    lin = torch.nn.Linear(n_in, n_out)
    y_ref = torch.nn.functional.leaky_relu(lin(x), 0.2)
    g = torch.randn_like(y_ref)
    y_ref.backward(g)
    c = n_out // hw
    ctot, coff = (c + 8, 8) if hw > 1 else (0, 0)
    xd, wd, bd = x.detach().to(dev), lin.weight.detach().to(dev), lin.bias.detach().to(dev)
    if hw > 1:   # feature f = ch * hw + p  ->  NHWC (m, p, coff + ch)
        y = torch.zeros(m, hw, ctot, device=dev)
        dy = torch.zeros(m, hw, ctot, device=dev)
        dy[..., coff:] = g.view(m, c, hw).permute(0, 2, 1).to(dev)
    else:
        y = torch.empty(m, n_out, device=dev)
        dy = g.to(dev)
    ops.fc_fwd(xd, wd, bd, y, True, hw, ctot, coff)
    got = y[..., coff:].permute(0, 2, 1).reshape(m, n_out) if hw > 1 else y
    np.testing.assert_allclose(got.cpu().numpy(), y_ref.detach().numpy(), rtol=1e-5, atol=1e-5)
    dw, db, dx = torch.empty_like(wd), torch.empty_like(bd), torch.empty_like(xd)
    ops.fc_bwd(xd, wd, y, dy, dw, db, dx, True, hw, ctot, coff)
    np.testing.assert_allclose(dw.cpu().numpy(), lin.weight.grad.numpy(), rtol=1e-4, atol=1e-5)
    np.testing.assert_allclose(db.cpu().numpy(), lin.bias.grad.numpy(), rtol=1e-4, atol=1e-5)
    np.testing.assert_allclose(dx.cpu().numpy(), x.grad.numpy(), rtol=1e-4, atol=1e-5)


@pytest.mark.gpu
def test_masked_bce_with_logits_loss(ops):
    '''RadarNetModel.compute_loss (src/radarnet_model.py:131-171).'''
    import torch
    torch.manual_seed(4)
    dev = 'cuda'
    logit = (torch.randn(3, 1, 37, 29) * 4).requires_grad_(True)
    target = (torch.rand(3, 1, 37, 29) < 0.3).float()
    valid = (torch.rand(3, 1, 37, 29) < 0.7).float()
    pw = 2.0
    ref = torch.nn.functional.binary_cross_entropy_with_logits(logit, target, reduction='none', pos_weight=torch.tensor(pw))
    ref = torch.sum(valid * ref) / torch.sum(valid)
    ref.backward()
    sums = torch.empty(2, dtype=torch.float64, device=dev)
    loss = torch.empty(1, device=dev)
    ld, td, vd = logit.detach().to(dev), target.to(dev), valid.to(dev)
    ops.bce_loss_fwd(ld, td, vd, sums, loss, pw)
    np.testing.assert_allclose(float(loss), float(ref), rtol=1e-5)
    dl = torch.empty_like(ld)
    ops.bce_loss_bwd(ld, td, vd, sums, torch.ones(1, device=dev), dl, pw)
    np.testing.assert_allclose(dl.cpu().numpy(), logit.grad.numpy(), rtol=1e-4, atol=1e-8)


# ---------------------------------------------------------------- bf16-operand mode (BASELINE.json "bf16" configurations)
@pytest.mark.gpu
@pytest.mark.parametrize('ksize,c1,c2,co,up', [(3, 32, 0, 64, False), (3, 64, 32, 64, False), (3, 64, 0, 32, False), (2, 32, 0, 64, True)])
def test_bf16_operand_mode_matches_bf16_rounded_reference(ops, ksize, c1, c2, co, up):
    '''rcf_conv_desc.precision = RCF_PREC_BF16: operands rounded to bf16 (nearest even), fp32 accumulate.  Against torch convs on
    bf16-rounded fp32 tensors (fp64 accumulate) the only difference is the fp32 accumulation order: tight tolerance.'''
    import os
    import torch
    if os.environ.get('RCF_CONV_SPLIT') == '0':
        pytest.skip('the bf16-operand mode lives in the split kernels, which RCF_CONV_SPLIT=0 turns off')
    torch.manual_seed(21)
    dev = 'cuda'
    n, h, w = 2, 29, 45
    rnd = lambda t: t.to(torch.bfloat16).to(torch.float32)
    x1 = torch.randn(n, h, w, c1, device=dev)
    x2 = torch.randn(n, h, w, c2, device=dev) if c2 else None
    ops.set_precision('bf16_operands')   # fp32 tensors, bf16 MFMA operands (the bf16-TENSOR configuration: tests/test_hip_bf16.py)
    try:
        if up:   # phase (1, 0) of the exact-2x UpConv: 2x2 conv, strided output
            wt = torch.randn(co, c1, 2, 2, device=dev) * 0.2
            d = ops.make_up2x_fwd_desc(n, h, w, c1, co, 1, 0)
        else:
            wt = torch.randn(co, c1 + c2, ksize, ksize, device=dev) * 0.1
            d = ops.make_fwd_desc(n, h, w, c1, c2, co, ksize, 1, h, w, 0)
        info = ops.conv_query(d)
        assert info.kernel_id >= 20000, info.kernel_id     # a bf16-operand kernel was selected
        packed = torch.empty(info.packed_weight_floats, device=dev)
        ops.conv_pack(d, wt, packed)
        out = torch.zeros(n, d.out_h_phys, d.out_w_phys, co, device=dev)
        ops.conv_fwd(d, x1, x2, packed, out, None)
        xin = torch.cat([x1, x2], 3) if c2 else x1
        xr, wr = rnd(xin).double().permute(0, 3, 1, 2), rnd(wt).double()
        if not up:
            ref = torch.nn.functional.conv2d(xr, wr, padding=ksize // 2)
            np.testing.assert_allclose(out.permute(0, 3, 1, 2).cpu().numpy(), ref.cpu().numpy(), rtol=2e-5, atol=2e-5)
            # weight gradient: dz and x rounded to bf16
            dz = torch.randn(n, h, w, co, device=dev)
            dw = torch.empty_like(wt)
            ws = torch.empty(max(1, info.wgrad_workspace_floats), device=dev)
            assert info.wgrad_kernel_id >= 30000
            ops.conv_wgrad(d, x1, x2, dz, dw, ws)
            wref = torch.nn.grad.conv2d_weight(xr, wr.shape, rnd(dz).double().permute(0, 3, 1, 2), padding=ksize // 2)
            np.testing.assert_allclose(dw.cpu().numpy(), wref.cpu().numpy(), rtol=2e-5, atol=2e-4)
            # input gradient of source 1: dz and W rounded
            dd = ops.make_dgrad_desc(d, 0, c1, False)
            di = ops.conv_query(dd)
            pk = torch.empty(di.packed_weight_floats, device=dev)
            ops.conv_pack(dd, wt, pk)
            dx = torch.empty(n, h, w, c1, device=dev)
            ops.conv_fwd(dd, dz, None, pk, dx, None)
            xref = torch.nn.grad.conv2d_input(xr.shape, wr, rnd(dz).double().permute(0, 3, 1, 2), padding=ksize // 2)[:, :c1]
            np.testing.assert_allclose(dx.permute(0, 3, 1, 2).cpu().numpy(), xref.cpu().numpy(), rtol=2e-5, atol=2e-4)
        else:
            # phase (a, b) = (1, 0): rows use source rows (y, y+1) (pad 0 on top), columns (x-1, x) (pad 1 on the left)
            xp = torch.nn.functional.pad(xr, (1, 0, 0, 1))
            ref = torch.nn.functional.conv2d(xp, wr)
            got = out[:, 1::2, 0::2, :].permute(0, 3, 1, 2)
            np.testing.assert_allclose(got.cpu().numpy(), ref.cpu().numpy(), rtol=2e-5, atol=2e-5)
    finally:
        ops.set_precision('fp32')


# ---------------------------------------------------------------- input augmentation
@pytest.mark.gpu
def test_transforms_match_reference_fixture_t8(ops):
    '''rcf_amd.fusionnet_transforms.Transforms against fixture T8: the real reference Transforms.transform with the decisions it
    drew (torchvision's adjust_* restated, parity unpinned there).  Flips are exact; the photometric chain may differ by one
    intensity level on the rare pixel whose blend lands within fp32 round-off of an integer.'''
    import os
    import torch
    from rcf_amd.fusionnet_transforms import Transforms
    g = np.load(os.path.join(os.path.dirname(__file__), 'golden', 'T8_transforms.npz'))
    cases = [
        dict(normalized_image_range=[0, 1], random_brightness=[0.8, 1.2], random_contrast=[0.8, 1.2], random_saturation=[0.8, 1.2],
             random_flip_type=['horizontal']),
        dict(normalized_image_range=[-1, 1], random_brightness=[0.5, 1.5], random_contrast=[-1], random_saturation=[0.5, 1.5],
             random_flip_type=['horizontal', 'vertical']),
        dict(normalized_image_range=[0, 255], random_brightness=[-1], random_contrast=[0.6, 1.4], random_saturation=[-1],
             random_flip_type=['none']),
    ]
    assert int(g['n_cases']) == len(cases)
    for ci, kw in enumerate(cases):
        t = Transforms(**kw)
        dec = {k[len('dec%d_' % ci):]: torch.from_numpy(g[k]).cuda() for k in g.files if k.startswith('dec%d_' % ci)}
        image = torch.from_numpy(g['image%d' % ci]).cuda()
        maps = [torch.from_numpy(g['map%d_%d' % (ci, j)]).cuda() for j in range(2)]
        images_out, maps_out = t.apply([image], maps, dec)
        for j in range(2):
            assert torch.equal(maps_out[j].cpu(), torch.from_numpy(g['map_out%d_%d' % (ci, j)])), (ci, j)
        ref = g['image_out%d' % ci]
        got = images_out[0].cpu().numpy()
        level = {1: 1.0 / 255.0, 2: 2.0 / 255.0, 0: 1.0}[t._norm_mode]
        diff = np.abs(got - ref)
        assert float(diff.max()) <= level * 1.001 + 1e-6, (ci, float(diff.max()))
        assert int((diff > 1e-5).sum()) <= max(3, int(0.002 * diff.size)), (ci, int((diff > 1e-5).sum()))
    # the random path: same API as the reference, decisions drawn on the device
    t = Transforms(**cases[0])
    img = torch.from_numpy(g['image0']).cuda()
    [out], [m0, m1] = t.transform([img], [torch.from_numpy(g['map0_0']).cuda(), torch.from_numpy(g['map0_1']).cuda()], 1.0)
    assert out.shape == img.shape and float(out.max()) <= 1.0 and float(out.min()) >= 0.0
    [out2] = Transforms(normalized_image_range=[0, 1]).transform([img])
    np.testing.assert_allclose(out2.cpu().numpy(), np.floor(g['image0']) / 255.0, rtol=0, atol=1e-7)


@pytest.mark.parametrize('c1,co,n,h,w', [(32, 64, 2, 45, 80), (64, 128, 1, 57, 101), (128, 256, 2, 29, 50), (16, 32, 3, 35, 51)])
@pytest.mark.parametrize('prec', ['fp32', 'bf16'])
def test_stride2_weight_gradient_as_four_phase_weight_gradients(ops, c1, co, n, h, w, prec):
    '''dW of a 3x3 stride-2 convolution = the nine real taps of four 2x2 weight gradients on the phase images of its input
    (rcf_phase_wgrad_gather_s2), odd and even extents, against torch's weight gradient; with bf16 tensors the operands are exact
    (bf16-valued), so the bar is the same.'''
    x = rnd(n, c1, h, w, seed=1)
    wt = rnd(co, c1, 3, 3, seed=2, scale=1.0 / np.sqrt(c1 * 9))
    ho, wo = (h - 1) // 2 + 1, (w - 1) // 2 + 1
    dz = rnd(n, co, ho, wo, seed=3)
    if prec == 'bf16':
        x, dz = x.bfloat16().float(), dz.bfloat16().float()
    wd = wt.clone().double().requires_grad_(True)
    (F.conv2d(x.double(), wd, stride=2, padding=1) * dz.double()).sum().backward()
    ops.set_precision(prec)
    try:
        cast = (lambda t: t.bfloat16()) if prec == 'bf16' else (lambda t: t)
        d = ops.make_fwd_desc(n, h, w, c1, 0, co, 3, 2)
        xg, dzg = cast(nhwc(x)), cast(nhwc(dz))
        dwp = torch.full((4, co, c1, 2, 2), float('nan'), device='cuda')
        for ph in range(4):
            dp = ops.make_s2_wgrad_desc(d, ph >> 1, ph & 1)
            qi = ops.conv_query(dp)
            ws = torch.empty(max(1, qi.wgrad_workspace_floats), device='cuda')
            ops.conv_wgrad(dp, xg, None, dzg, dwp[ph], ws)
        dw = torch.full((co, c1, 3, 3), float('nan'), device='cuda')
        ops.phase_wgrad_gather_s2(dwp, dw)
        torch.cuda.synchronize()
    finally:
        ops.set_precision('fp32')
    assert rel(dw.cpu().double(), wd.grad) < TOL


@pytest.mark.parametrize('c1,co,n,h,w', [(32, 64, 2, 45, 80), (64, 128, 2, 57, 100), (128, 256, 1, 29, 51)])
def test_stride2_forward_on_the_three_plane_split_kernel(ops, c1, co, n, h, w, monkeypatch):
    '''The LSTEP = 2 split kernel with fp32 precision (opt-in, RCF_S2_SPLIT=1; the default for bf16 operands): fp32-accurate --
    its error against an fp64 reference is within 1.5x of the exact f32-MFMA kernel's on the same data.'''
    import os
    if os.environ.get('RCF_CONV_SPLIT') == '0':
        pytest.skip('the split kernels are switched off (RCF_CONV_SPLIT=0)')
    x = rnd(n, c1, h, w, seed=1) + 0.3
    wt = rnd(co, c1, 3, 3, seed=2, scale=1.0 / np.sqrt(c1 * 9))
    ref = F.conv2d(x.double(), wt.double(), stride=2, padding=1)
    mag = F.conv2d(x.double().abs(), wt.double().abs(), stride=2, padding=1)
    errs = {}
    for flag in ('0', '1'):
        monkeypatch.setenv('RCF_S2_SPLIT', flag)
        d = ops.make_fwd_desc(n, h, w, c1, 0, co, 3, 2)
        info = ops.conv_query(d)
        assert (info.kernel_id // 1000 == 6) == (flag == '1'), info.kernel_id
        packed = torch.empty(info.packed_weight_floats, device='cuda')
        ops.conv_pack(d, dev(wt), packed)
        out = torch.full((n, d.h_out, d.w_out, co), float('nan'), device='cuda')
        ops.conv_fwd(d, nhwc(x), None, packed, out, None)
        torch.cuda.synchronize()
        errs[flag] = float(((nchw(out).double() - ref).abs() / mag).max())
    assert errs['1'] < 1.5 * errs['0'] + 1e-8 and errs['1'] < 1e-6, errs


@pytest.mark.gpu
@pytest.mark.parametrize('n,h,w,c', [(8, 900, 1600, 32), (8, 450, 800, 64)])
def test_batchnorm_reductions_at_batch8_full_resolution(ops, n, h, w, c):
    '''The cross-image reductions train-mode BatchNorm needs, at the benchmark's own extent (8 x 900 x 1600 x 32: 369 M values, 64-bit
    offsets): the forward statistics a convolution's epilogue accumulates (sum z, sum z^2 per channel, fp64) and the backward sums
    (sum g, sum g * xhat) of rcf_bn_act_bwd_reduce, against fp64 torch reductions of the very tensors the kernels read -- the
    full-batch counterpart of the small-size checks above (the batch-8 eval-mode test in test_configs_gpu.py covers everything that
    does not reduce over the batch).'''
    import torch
    from rcf_amd._lib import RCF_ACT_LEAKY_RELU
    g = torch.Generator(device='cuda').manual_seed(11)
    x = torch.rand(n, h, w, c, device='cuda', generator=g) * 2 - 1
    wt = (torch.rand(c, c, 3, 3, device='cuda', generator=g) * 2 - 1) / (3.0 * c ** 0.5)
    d = ops.make_fwd_desc(n, h, w, c, 0, c, 3, 1, h, w, 0)
    info = ops.conv_query(d)
    packed = torch.empty(info.packed_weight_floats, device='cuda')
    ops.conv_pack(d, wt, packed)
    z = torch.empty(n, h, w, c, device='cuda')
    part = torch.empty(info.n_partials, 2, c, device='cuda', dtype=torch.float64)
    ops.conv_fwd(d, x, None, packed, z, part)
    st = part.sum(0)
    want1 = z.view(-1, c).double().sum(0)
    want2 = (z.view(-1, c).double() ** 2).sum(0)
    assert float((st[0] - want1).abs().max()) <= 1e-9 * float(want2.sqrt().max() * (n * h * w) ** 0.5) + 1e-6
    assert float(((st[1] - want2) / want2).abs().max()) < 1e-12
    # a few output values of the last image against the fp64 convolution of its neighbourhood (the 64-bit addressing of image 7)
    ref = torch.nn.functional.conv2d(x[n - 1:, h - 8:, :64].permute(0, 3, 1, 2).double(), wt.double(), padding=1)[:, :, 1:-1, 1:-1]
    got = z[n - 1:, h - 7:h - 1, 1:63].permute(0, 3, 1, 2).double()
    assert float((got - ref[:, :, :6]).abs().max()) < 1e-5
    # BatchNorm coefficients of these statistics, then the backward sums
    n_pix = n * h * w
    mean = want1 / n_pix
    var = want2 / n_pix - mean ** 2
    invstd = (var + 1e-5).rsqrt()
    gamma = torch.rand(c, device='cuda', generator=g, dtype=torch.float64) + 0.5
    beta = torch.rand(c, device='cuda', generator=g, dtype=torch.float64) - 0.5
    coef = torch.stack([gamma * invstd, beta - mean * gamma * invstd, mean, invstd]).float().contiguous()
    del x
    dout = (torch.rand(n, h, w, c, device='cuda', generator=g) * 2 - 1) * 1e-3
    nb = ops.ew_blocks(n_pix, c)
    bpart = torch.empty(nb, 2, c, device='cuda', dtype=torch.float64)
    ops.bn_act_bwd_reduce(dout, z, coef, None, bpart, n_pix, c, RCF_ACT_LEAKY_RELU, False)
    got_b = bpart.sum(0)
    s_g = torch.zeros(c, device='cuda', dtype=torch.float64)
    s_gx = torch.zeros(c, device='cuda', dtype=torch.float64)
    cf = coef.double()
    for i in range(n):          # image by image: the fp64 temporaries of one image are 1.5 GB
        zz, dd = z[i].view(-1, c), dout[i].view(-1, c)
        y = zz.double() * cf[0] + cf[1]                   # the kernel's v_fma rounds once: its sign is the exact sign
        gg = (dd * torch.where(y > 0, 1.0, 0.2).float()).double()     # g is an fp32 product in the kernel, summed in fp64
        xh = ((zz - coef[2]) * coef[3]).double()         # xhat as the kernel forms it (fp32), summed in fp64
        s_g += gg.sum(0)
        s_gx += (gg * xh).sum(0)
    scale = float(dout.double().abs().sum() / c)
    assert float((got_b[0] - s_g).abs().max()) < 1e-9 * scale
    assert float((got_b[1] - s_gx).abs().max()) < 1e-9 * scale * 4


def test_images_beyond_the_buffer_range_are_refused_loudly(ops):
    '''The split / DMA convolution kernels address their tensors through buffer descriptors based at a tile's first image (32-bit byte
    offsets checked against a 2 GB range, a tile touching at most two consecutive images): an image of 1 GB or more would put valid
    pixels out of range, where the hardware returns zeros / drops stores SILENTLY -- so such a shape must be refused by the query and
    by every launch (RCF_EUNSUPPORTED), never computed.  A 900 x 1600 image with 64 channels is 0.37 GB.'''
    from rcf_amd import _lib
    ok = ops.make_fwd_desc(1, 900, 1600, 64, 0, 64, 3, 1)
    assert ops.conv_query(ok).packed_weight_floats > 0
    big = ops.make_fwd_desc(1, 4096, 4096, 64, 0, 64, 3, 1)          # 4.3 GB per fp32 image
    with pytest.raises(_lib.RcfError):
        ops.conv_query(big)
    # virtual tall image (batch > 1, 3x3 stride 1): a tile higher than the image spans ceil(34 / (h + 1)) + 1 images from ONE descriptor
    # -- 8 rows: 5 images, so 0.43 GB per image is the limit.  0.54 GB images (8 x 131072 x 128 fp32) would overflow the 2 GB range
    wide = ops.make_fwd_desc(2, 8, 131072, 128, 0, 128, 3, 1)
    try:
        info = ops.conv_query(wide)
        assert (info.kernel_id // 100) % 10 < 4, 'a virtual-tall tiling was chosen for 0.54 GB images of 8 rows: %d' % info.kernel_id
    except _lib.RcfUnsupported:
        pass
