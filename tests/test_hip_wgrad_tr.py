'''
GPU tests of conv_wgrad_tr_kernel (csrc/rcf_conv_wgrad_tr.h): the split weight gradient with producer / consumer waves, NHWC tiles
in LDS and gfx950's transposing LDS read (ds_read_b64_tr_b16) -- the backward of the reference's Conv2d w.r.t. its weight
(src/net_utils.py:29-91 under loss.backward(), src/fusionnet_main.py:398).

Three statements per shape, through the C ABI (rcf_conv2d_wgrad / rcf_conv2d_wgrad_scaled):
  1. parity: against stock PyTorch CPU fp32 autograd of the same convolution (fp32 tensors on two fp16 planes: 1e-4 of max|dW|;
     bf16 tensors: fp32 convolution of the bf16-valued operands, accumulation-order noise);
  2. the kernel really ran: rcf_conv2d_query reports the tr kernel's id (hundreds digit + 2) and RCF_WGRAD_TR=0 takes it away;
  3. BITWISE equal to conv_wgrad_split_kernel (RCF_WGRAD_TR=0) on the same inputs: same tiles, same operand planes, same lane ->
     K-slot assignment, same MFMA order => the same partial sums, so everything the old kernel's tests established carries over.
Shapes cover every workgroup configuration (64x64, 32x64, 64x32, 32x32 channels), image borders in both directions, tiles that
straddle images of the virtual tall image (n > 1, small h), ragged channel chunks (96 = 64 + 32, concat 64 + 32, 48), the 2x2 phase
descriptors with strided dz (up-2x) and strided x (stride-2 phases), and 1x1 with bf16 tensors.
'''
import os

import numpy as np
import pytest
import torch
import torch.nn.functional as F

pytestmark = pytest.mark.gpu


@pytest.fixture(scope='module')
def ops():
    import rcf_amd  # noqa: F401
    from rcf_amd import _lib, ops as _ops
    assert torch.cuda.is_available()
    _lib.load()
    return _ops


@pytest.fixture(autouse=True)
def _restore(ops):
    yield
    ops.set_precision('fp32')
    os.environ.pop('RCF_WGRAD_TR', None)


def rnd(*shape, seed=0, scale=1.0):
    g = torch.Generator().manual_seed(seed)
    return (torch.rand(*shape, generator=g) * 2 - 1) * scale


def rel(a, b):
    return float((a.double() - b.double()).abs().max() / (b.double().abs().max() + 1e-30))


# (ksize, c1, c2, c_out, n, h, w)
CASES = [
    (3, 64, 0, 64, 2, 37, 53),      # 64 x 64: ring of halo rows, odd sizes
    (3, 64, 0, 64, 1, 64, 96),      # interior tiles only in the middle
    (3, 128, 0, 128, 3, 15, 25),    # 2 x 2 chunks, virtual tall image
    (3, 32, 0, 64, 2, 33, 47),      # 32 x 64 (two row slices)
    (3, 64, 0, 32, 2, 33, 47),      # 64 x 32
    (3, 32, 0, 32, 2, 40, 70),      # 32 x 32: 16-row tiles, four row slices
    (3, 16, 0, 32, 1, 35, 51),      # half-empty 32-channel block
    (3, 64, 32, 64, 2, 29, 50),     # concat: the skip connection's chunk is half empty
    (3, 96, 0, 64, 1, 30, 44),      # ragged second chunk
    (3, 256, 0, 256, 2, 8, 13),     # image smaller than a tile
    (2, 64, 0, 64, 2, 22, 31),      # 2x2 (pad 1): the phase convolutions' kernel size
    (2, 32, 0, 32, 1, 33, 49),
]


def _run(ops, prec, k, c1, c2, co, n, h, w, tr, desc_fn=None, seed=30):
    '''-> (dw tensor on the CPU, reference dw or None, kernel id)'''
    ops.set_precision(prec)
    os.environ['RCF_WGRAD_TR'] = '1' if tr else '0'
    b16 = prec == 'bf16'
    x1 = rnd(n, c1, h, w, seed=seed)
    x2 = rnd(n, c2, h, w, seed=seed + 1) if c2 else None
    d = ops.make_fwd_desc(n, h, w, c1, c2, co, k, 1) if desc_fn is None else desc_fn()
    dz = rnd(n, co, d.h_out, d.w_out, seed=seed + 14)
    if b16:
        x1, dz = x1.bfloat16().float(), dz.bfloat16().float()
        x2 = None if x2 is None else x2.bfloat16().float()
    dt = torch.bfloat16 if b16 else torch.float32
    to = lambda t: t.permute(0, 2, 3, 1).contiguous().cuda().to(dt)
    g1, g2, gz = to(x1), (None if x2 is None else to(x2)), to(dz)
    info = ops.conv_query(d)
    ws = torch.empty(max(1, info.wgrad_workspace_floats), device='cuda')
    dw = torch.full((co, c1 + c2, k, k), float('nan'), device='cuda')
    scales = None
    if prec == 'f16x2':
        a1 = ops.amax(g1)
        a2 = ops.amax(g2) if g2 is not None else None
        az = ops.amax(gz)   # (all three kept alive until the launch: rcf_conv_scales holds raw device pointers)
        scales = ops.make_scales(a1, a2, None, az)
    ops.conv_wgrad(d, g1, g2, gz, dw, ws, scales=scales)
    torch.cuda.synchronize()
    return dw.cpu(), (x1, x2, dz), info.wgrad_kernel_id


def _ref(k, x1, x2, dz, co):
    xin = x1 if x2 is None else torch.cat([x1, x2], 1)
    wt = torch.zeros(co, xin.shape[1], k, k, requires_grad=True)
    pad = 1
    out = F.conv2d(xin, wt, stride=1, padding=pad)
    assert out.shape == dz.shape, (out.shape, dz.shape)
    (out * dz).sum().backward()
    return wt.grad


@pytest.mark.parametrize('prec', ['f16x2', 'bf16'])
@pytest.mark.parametrize('case', CASES, ids=[str(c) for c in CASES])
def test_tr_weight_gradient_parity_and_bitwise_the_previous_kernel(ops, case, prec):
    k, c1, c2, co, n, h, w = case
    new, (x1, x2, dz), kid = _run(ops, prec, *case, tr=True)
    old, _, kid_old = _run(ops, prec, *case, tr=False)
    assert (kid // 100) % 10 in (3, 7), kid            # px 16 (+100) + tr (+200) (+400 virtual tall image)
    assert (kid_old // 100) % 10 in (1, 5), kid_old
    ref = _ref(k, x1, x2, dz, co)
    e = rel(new, ref)
    print('%s %s: id %d (previous %d), vs CPU fp32 autograd %.2e, bitwise the previous kernel: %s'
          % (prec, case, kid, kid_old, e, bool(torch.equal(new, old))))
    assert torch.isfinite(new).all()
    assert e < (1e-4 if prec == 'f16x2' else 3e-5)
    assert torch.equal(new, old)


@pytest.mark.parametrize('prec', ['f16x2', 'bf16'])
@pytest.mark.parametrize('a,b', [(0, 0), (0, 1), (1, 0), (1, 1)])
def test_tr_weight_gradient_of_an_up2x_phase_reads_dz_strided(ops, prec, a, b):
    '''One phase of the up-2x convolution (ops.make_up2x_fwd_desc: pad (1 - a, 1 - b), dz read at (2y + a, 2x + b)) -- the per-phase
    launches the engine falls back to when the merged phase-pair form is switched off.'''
    n, hs, ws, ci, co = 2, 21, 30, 64, 32
    ops.set_precision(prec)
    b16 = prec == 'bf16'
    dt = torch.bfloat16 if b16 else torch.float32
    x = rnd(n, ci, hs, ws, seed=3)
    dz = rnd(n, co, 2 * hs, 2 * ws, seed=4)
    if b16:
        x, dz = x.bfloat16().float(), dz.bfloat16().float()
    to = lambda t: t.permute(0, 2, 3, 1).contiguous().cuda().to(dt)
    gx, gz = to(x), to(dz)
    outs = {}
    for tr in (True, False):
        os.environ['RCF_WGRAD_TR'] = '1' if tr else '0'
        d = ops.make_up2x_fwd_desc(n, hs, ws, ci, co, a, b)
        info = ops.conv_query(d)
        wsb = torch.empty(max(1, info.wgrad_workspace_floats), device='cuda')
        dw = torch.full((co, ci, 2, 2), float('nan'), device='cuda')
        amx, amz = (ops.amax(gx), ops.amax(gz)) if prec == 'f16x2' else (None, None)   # (kept alive: rcf_conv_scales holds raw device pointers)
        scales = ops.make_scales(amx, None, None, amz) if prec == 'f16x2' else None
        ops.conv_wgrad(d, gx, None, gz, dw, wsb, scales=scales)
        torch.cuda.synchronize()
        outs[tr] = (dw.cpu(), info.wgrad_kernel_id)
    # reference: 2x2 convolution of x padded by (1 - a, a) x (1 - b, b) against the phase image of dz
    wt = torch.zeros(co, ci, 2, 2, requires_grad=True)
    xp = F.pad(x, (1 - b, b, 1 - a, a))
    out = F.conv2d(xp, wt)
    (out * dz[:, :, a::2, b::2]).sum().backward()
    e = rel(outs[True][0], wt.grad)
    print('%s phase (%d, %d): ids %d / %d, vs CPU %.2e' % (prec, a, b, outs[True][1], outs[False][1], e))
    assert (outs[True][1] // 100) % 10 in (3, 7)
    assert e < (1e-4 if prec == 'f16x2' else 3e-5)
    assert torch.equal(outs[True][0], outs[False][0])


@pytest.mark.parametrize('prec', ['f16x2', 'bf16'])
@pytest.mark.parametrize('a,b', [(0, 0), (1, 1)])
def test_tr_weight_gradient_of_a_stride2_phase_reads_x_strided(ops, prec, a, b):
    '''One phase of a 3x3 stride-2 convolution's weight gradient (ops.make_s2_wgrad_desc: x gathered at (2y + a, 2x + b), odd input
    sizes so that the last phase row / column does not exist).'''
    n, h, w, ci, co = 2, 45, 63, 64, 64
    ops.set_precision(prec)
    b16 = prec == 'bf16'
    dt = torch.bfloat16 if b16 else torch.float32
    fwd = ops.make_fwd_desc(n, h, w, ci, 0, co, 3, 2)
    x = rnd(n, ci, h, w, seed=5)
    dz = rnd(n, co, fwd.h_out, fwd.w_out, seed=6)
    if b16:
        x, dz = x.bfloat16().float(), dz.bfloat16().float()
    to = lambda t: t.permute(0, 2, 3, 1).contiguous().cuda().to(dt)
    gx, gz = to(x), to(dz)
    outs = {}
    for tr in (True, False):
        os.environ['RCF_WGRAD_TR'] = '1' if tr else '0'
        d = ops.make_s2_wgrad_desc(fwd, a, b)
        info = ops.conv_query(d)
        wsb = torch.empty(max(1, info.wgrad_workspace_floats), device='cuda')
        dw = torch.full((co, ci, 2, 2), float('nan'), device='cuda')
        amx, amz = (ops.amax(gx), ops.amax(gz)) if prec == 'f16x2' else (None, None)   # (kept alive: rcf_conv_scales holds raw device pointers)
        scales = ops.make_scales(amx, None, None, amz) if prec == 'f16x2' else None
        ops.conv_wgrad(d, gx, None, gz, dw, wsb, scales=scales)
        torch.cuda.synchronize()
        outs[tr] = (dw.cpu(), info.wgrad_kernel_id)
    # reference: the phase image of x (zero beyond the source), padded by one on top / left, against dz
    xph = x[:, :, a::2, b::2]
    xph = F.pad(xph, (1, fwd.w_out - xph.shape[3], 1, fwd.h_out - xph.shape[2]))
    wt = torch.zeros(co, ci, 2, 2, requires_grad=True)
    out = F.conv2d(xph, wt)
    assert out.shape == dz.shape
    (out * dz).sum().backward()
    e = rel(outs[True][0], wt.grad)
    print('%s stride-2 phase (%d, %d): ids %d / %d, vs CPU %.2e' % (prec, a, b, outs[True][1], outs[False][1], e))
    assert (outs[True][1] // 100) % 10 in (3, 7)
    assert e < (1e-4 if prec == 'f16x2' else 3e-5)
    assert torch.equal(outs[True][0], outs[False][0])


@pytest.mark.parametrize('case', [(64, 128, 2, 29, 50), (32, 64, 1, 45, 80), (128, 256, 3, 15, 25), (16, 32, 2, 35, 51)],
                         ids=lambda c: str(c))
def test_tr_weight_gradient_1x1_with_bf16_tensors(ops, case):
    c1, co, n, h, w = case
    ops.set_precision('bf16')
    x = rnd(n, c1, h, w, seed=7).bfloat16().float()
    dz = rnd(n, co, h, w, seed=8).bfloat16().float()
    to = lambda t: t.permute(0, 2, 3, 1).contiguous().cuda().bfloat16()
    gx, gz = to(x), to(dz)
    outs = {}
    for tr in (True, False):
        os.environ['RCF_WGRAD_TR'] = '1' if tr else '0'
        d = ops.make_fwd_desc(n, h, w, c1, 0, co, 1, 1)
        info = ops.conv_query(d)
        wsb = torch.empty(max(1, info.wgrad_workspace_floats), device='cuda')
        dw = torch.full((co, c1, 1, 1), float('nan'), device='cuda')
        ops.conv_wgrad(d, gx, None, gz, dw, wsb)
        torch.cuda.synchronize()
        outs[tr] = (dw.cpu(), info.wgrad_kernel_id)
    ref = torch.einsum('nchw,nohw->oc', x.double(), dz.double()).float().reshape(co, c1, 1, 1)
    e = rel(outs[True][0], ref)
    print('1x1 %s: ids %d / %d, vs fp64 einsum %.2e' % (case, outs[True][1], outs[False][1], e))
    assert (outs[True][1] // 100) % 10 in (3, 7)
    assert e < 3e-5
    assert torch.equal(outs[True][0], outs[False][0])


@pytest.mark.parametrize('prec', ['f16x2', 'bf16'])
@pytest.mark.parametrize('case', [(64, 64, 2, 29, 50, 15, 25), (128, 64, 1, 57, 100, 29, 50), (32, 32, 2, 35, 70, 15, 29)], ids=lambda c: str(c))
def test_tr_weight_gradient_with_the_nearest_upsample_gather(ops, prec, case):
    '''UpConv2d's convolution (src/net_utils.py:193-198): x is F.interpolate(source, size=(h, w)) (nearest), gathered while staging.'''
    c1, co, n, h, w, hs, ws = case
    ops.set_precision(prec)
    b16 = prec == 'bf16'
    dt = torch.bfloat16 if b16 else torch.float32
    x = rnd(n, c1, hs, ws, seed=9)
    dz = rnd(n, co, h, w, seed=10)
    if b16:
        x, dz = x.bfloat16().float(), dz.bfloat16().float()
    to = lambda t: t.permute(0, 2, 3, 1).contiguous().cuda().to(dt)
    gx, gz = to(x), to(dz)
    outs = {}
    for tr in (True, False):
        os.environ['RCF_WGRAD_TR'] = '1' if tr else '0'
        d = ops.make_fwd_desc(n, h, w, c1, 0, co, 3, 1, hs, ws, 1)
        info = ops.conv_query(d)
        wsb = torch.empty(max(1, info.wgrad_workspace_floats), device='cuda')
        dw = torch.full((co, c1, 3, 3), float('nan'), device='cuda')
        amx, amz = (ops.amax(gx), ops.amax(gz)) if prec == 'f16x2' else (None, None)   # (kept alive: rcf_conv_scales holds raw device pointers)
        scales = ops.make_scales(amx, None, None, amz) if prec == 'f16x2' else None
        ops.conv_wgrad(d, gx, None, gz, dw, wsb, scales=scales)
        torch.cuda.synchronize()
        outs[tr] = (dw.cpu(), info.wgrad_kernel_id)
    wt = torch.zeros(co, c1, 3, 3, requires_grad=True)
    out = F.conv2d(F.interpolate(x, size=(h, w)), wt, padding=1)
    (out * dz).sum().backward()
    e = rel(outs[True][0], wt.grad)
    print('%s nearest %s: ids %d / %d, vs CPU %.2e' % (prec, case, outs[True][1], outs[False][1], e))
    assert (outs[True][1] // 100) % 10 in (3, 7)
    assert e < (1e-4 if prec == 'f16x2' else 3e-5)
    assert torch.equal(outs[True][0], outs[False][0])


@pytest.mark.parametrize('prec', ['f16x2', 'bf16'])
@pytest.mark.parametrize('case', [(64, 128, 2, 45, 63), (32, 64, 8, 57, 100), (128, 256, 2, 29, 50)], ids=lambda c: str(c))
def test_tr_four_phase_stride2_weight_gradient_in_one_launch(ops, prec, case):
    '''phase_sum == 1 (ops.make_s2_wgrad_desc(all_phases=True)): (slot, phase) workgroups, dw = [4][co][ci][2][2] -- bitwise the
    previous kernel's one-launch form, and each phase equal to its single-phase launch.'''
    ci, co, n, h, w = case
    ops.set_precision(prec)
    b16 = prec == 'bf16'
    dt = torch.bfloat16 if b16 else torch.float32
    fwd = ops.make_fwd_desc(n, h, w, ci, 0, co, 3, 2)
    x = rnd(n, ci, h, w, seed=11)
    dz = rnd(n, co, fwd.h_out, fwd.w_out, seed=12)
    if b16:
        x, dz = x.bfloat16().float(), dz.bfloat16().float()
    to = lambda t: t.permute(0, 2, 3, 1).contiguous().cuda().to(dt)
    gx, gz = to(x), to(dz)
    amx, amz = (ops.amax(gx), ops.amax(gz)) if prec == 'f16x2' else (None, None)   # (kept alive: rcf_conv_scales holds raw device pointers)
    scales = ops.make_scales(amx, None, None, amz) if prec == 'f16x2' else None
    outs = {}
    for tr in (True, False):
        os.environ['RCF_WGRAD_TR'] = '1' if tr else '0'
        d = ops.make_s2_wgrad_desc(fwd, 0, 0, all_phases=True)
        info = ops.conv_query(d)
        wsb = torch.empty(max(1, info.wgrad_workspace_floats), device='cuda')
        dw = torch.full((4, co, ci, 2, 2), float('nan'), device='cuda')
        ops.conv_wgrad(d, gx, None, gz, dw, wsb, scales=scales)
        torch.cuda.synchronize()
        outs[tr] = (dw.cpu(), info.wgrad_kernel_id)
    assert (outs[True][1] // 100) % 10 in (3, 7) and (outs[False][1] // 100) % 10 in (1, 5)
    assert torch.isfinite(outs[True][0]).all()
    assert torch.equal(outs[True][0], outs[False][0])
    os.environ['RCF_WGRAD_TR'] = '1'
    worst = 0.0
    for a in (0, 1):
        for b in (0, 1):
            xph = x[:, :, a::2, b::2]
            xph = F.pad(xph, (1, fwd.w_out - xph.shape[3], 1, fwd.h_out - xph.shape[2]))
            wt = torch.zeros(co, ci, 2, 2, requires_grad=True)
            (F.conv2d(xph, wt) * dz).sum().backward()
            worst = max(worst, rel(outs[True][0][2 * a + b], wt.grad))
    print('%s four-phase stride-2 weight gradient %s: ids %d / %d, worst phase vs CPU %.2e' % (prec, case, outs[True][1], outs[False][1], worst))
    assert worst < (1e-4 if prec == 'f16x2' else 3e-5)


def test_tr_kernel_is_not_taken_where_it_does_not_apply(ops):
    '''The merged up-2x phase pairs and the exact three-plane tier keep conv_wgrad_split_kernel.'''
    os.environ['RCF_WGRAD_TR'] = '1'
    ops.set_precision('f16x2')
    d = ops.make_up2x_fwd_desc(2, 20, 30, 64, 32, 0, 0, phase_out=True)
    assert (ops.conv_query(d).wgrad_kernel_id // 100) % 10 in (1, 5)
    ops.set_precision('fp32')
    d = ops.make_fwd_desc(2, 29, 50, 64, 0, 64, 3, 1)
    assert (ops.conv_query(d).wgrad_kernel_id // 100) % 10 in (1, 5)
