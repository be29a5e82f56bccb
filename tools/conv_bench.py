'''Micro-benchmark (GPU box): the hot FusionNet conv layer shapes at batch 8, 900x1600, one by one through the C ABI.
usage: [RCF_BENCH_PREC=fp32|f16x2|bf16|bf16_operands] python tools/conv_bench.py [reps] [filter]
(the first table: the direct kernels of every layer shape, RCF_BENCH_PREC=fp32|bf16|bf16_operands; the second: the phase forms the engine runs
for the up-2x and stride-2 layers, also under f16x2 = the default fp32 tier)'''
import sys, os
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
import rcf_amd
from rcf_amd import ops

N = 8
# name, ksize, stride, c1, c2, cout, h_in, w_in, up_from
LAYERS = [
    ('deconv0.deconv 64->32 up', 3, 1, 64, 0, 32, 900, 1600, (450, 800)),
    ('deconv0.conv   32->32',    3, 1, 32, 0, 32, 900, 1600, None),
    ('deconv1.deconv 64->64 up', 3, 1, 64, 0, 64, 450, 800, (225, 400)),
    ('deconv1.conv   64+32->64', 3, 1, 64, 32, 64, 450, 800, None),
    ('deconv2.conv   64+64->64', 3, 1, 64, 64, 64, 225, 400, None),
    ('deconv3.conv  128+128->128', 3, 1, 128, 128, 128, 113, 200, None),
    ('deconv4.conv  256+256->256', 3, 1, 256, 256, 256, 57, 100, None),
    ('blocks2_img    64->64',    3, 1, 64, 0, 64, 225, 400, None),
    ('blocks3_img   128->128',   3, 1, 128, 0, 128, 113, 200, None),
    ('blocks4_img   256->256',   3, 1, 256, 0, 256, 57, 100, None),
    ('blocks5_img   256->256',   3, 1, 256, 0, 256, 29, 50, None),
    ('blocks3_img.0 s2 64->128', 3, 2, 64, 0, 128, 225, 400, None),
    ('blocks4_img.0 s2 128->256', 3, 2, 128, 0, 256, 113, 200, None),
    ('blocks3_dep.0 s2 32->64', 3, 2, 32, 0, 64, 225, 400, None),
    ('fuse2 1x1 32->64',         1, 1, 32, 0, 64, 225, 400, None),
    ('stem 7x7 3->32',           7, 2, 3, 0, 32, 900, 1600, None),
]
reps = int(sys.argv[1]) if len(sys.argv) > 1 else 5
flt = sys.argv[2] if len(sys.argv) > 2 else ''
dev = 'cuda'
ops.set_precision(os.environ.get('RCF_BENCH_PREC', 'fp32'))
ADT = ops.act_dtype()
DATA_SCALE = float(os.environ.get('RCF_BENCH_DATA_SCALE', '1'))   # 0: all-zero operands (no toggling in the matrix pipe: power probe)
print('%-30s %9s | %8s %7s | %8s %7s | %8s %7s' % ('layer', 'GF', 'fwd ms', 'TF/s', 'dgrad ms', 'TF/s', 'wgrad ms', 'TF/s'))
tot = [0.0, 0.0, 0.0, 0.0]
for name, k, s, c1, c2, co, h, w, up in ([] if os.environ.get('RCF_BENCH_PREC') == 'f16x2' else LAYERS):   # (f16x2 needs the operands' maxima: second table)
    if flt and flt not in name:
        continue
    hs, ws = (h, w) if up is None else up
    d = ops.make_fwd_desc(N, h, w, c1, c2, co, k, s, hs, ws, 0 if up is None else 1)
    info = ops.conv_query(d)
    x1 = (torch.randn(N, hs, ws, c1, device=dev) * DATA_SCALE).to(torch.float32 if k == 7 else ADT)
    x2 = torch.randn(N, h, w, c2, device=dev).to(ADT) if c2 else None
    wt = torch.randn(co, c1 + c2, k, k, device=dev) * 0.05 * DATA_SCALE
    packed = torch.empty(info.packed_weight_floats, device=dev)
    ops.conv_pack(d, wt, packed)
    out = torch.empty(N, d.h_out, d.w_out, co, device=dev, dtype=ADT)
    part = torch.empty(info.n_partials, 2, co, device=dev, dtype=torch.float64)
    dz = (torch.randn(N, d.h_out, d.w_out, co, device=dev) * DATA_SCALE).to(ADT)
    dw = torch.empty_like(wt)
    wsb = torch.empty(max(1, info.wgrad_workspace_floats), device=dev)
    gf = ops.algorithmic_flops(d) / 1e9

    def timeit(fn):
        fn(); torch.cuda.synchronize()
        e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
        e0.record()
        for _ in range(reps): fn()
        e1.record(); torch.cuda.synchronize()
        return e0.elapsed_time(e1) / reps

    t_f = timeit(lambda: ops.conv_fwd(d, x1, x2, packed, out, part))
    t_d = float('nan')
    if k != 7:
        dd = ops.make_dgrad_desc(d, 0, c1, False)
        di = ops.conv_query(dd)
        pd = torch.empty(di.packed_weight_floats, device=dev)
        ops.conv_pack(dd, wt, pd)
        dx = torch.empty(N, h, w, c1, device=dev, dtype=ADT)
        t_d = timeit(lambda: ops.conv_fwd(dd, dz, None, pd, dx, None))
        gf_d = ops.algorithmic_flops(dd) / 1e9
    t_w = timeit(lambda: ops.conv_wgrad(d, x1, x2, dz, dw, wsb))
    print('%-30s %9.1f | %8.3f %7.1f | %8.3f %7.1f | %8.3f %7.1f' % (name, gf, t_f, gf / t_f, t_d, (gf_d / t_d) if k != 7 else float('nan'), t_w, gf / t_w))
    del x1, x2, out, dz, wsb

# ---- the phase forms the engine runs for the exact-2x up-convolutions and the stride-2 layers' backward (one launch each where the library
# has it: rcf_conv_desc.phase_sum; otherwise the four per-phase launches).  TF/s on the 3x3 convolution's algorithmic FLOPs (what the layer
# computes: 2 N Ho Wo Co 9 Ci), so a column compares with the direct kernels' above; the phase forms execute 4/9 (up-2x) resp. 9/9 of them.
from rcf_amd._lib import RCF_PHASE_S2_DGRAD, RCF_PHASE_UP2X_DGRAD, RCF_PHASE_UP2X_FWD, RCF_PREC_F16X2
F16 = ops.get_precision() == RCF_PREC_F16X2


def _amax(t):
    return t.float().abs().max().reshape(1) if F16 else None


def _timeit(fn):
    fn(); torch.cuda.synchronize()
    e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    e0.record()
    for _ in range(reps): fn()
    e1.record(); torch.cuda.synchronize()
    return e0.elapsed_time(e1) / reps


def _pack4(desc, w4, aw):
    q = ops.conv_query(desc)
    buf = torch.empty(4 * q.packed_weight_floats, device=dev)
    for ph in range(4):
        dst = buf[ph * q.packed_weight_floats:(ph + 1) * q.packed_weight_floats]
        ops.conv_pack(desc, w4[ph], dst, aw) if F16 else ops.conv_pack(desc, w4[ph], dst)
    return q, buf


print()
print('%-34s %9s | %8s %7s | %8s %7s | %8s %7s   (phase forms, one launch each)' % ('layer', 'GF(3x3)', 'fwd ms', 'TF/s', 'dgrad ms', 'TF/s', 'wgrad ms', 'TF/s'))
UP = [('deconv0.deconv 64->32 up', 64, 32, 450, 800), ('deconv1.deconv 64->64 up', 64, 64, 225, 400), ('deconv2.deconv 128->64 up', 128, 64, 113, 200)]
S2 = [('blocks3_img.0 s2 64->128', 64, 128, 225, 400), ('blocks4_img.0 s2 128->256', 128, 256, 113, 200), ('blocks3_dep.0 s2 32->64', 32, 64, 225, 400)]
for name, c1, co, h, w in UP:
    if flt and flt not in name:
        continue
    x = (torch.randn(N, h, w, c1, device=dev) * DATA_SCALE).to(ADT)
    dz = (torch.randn(N, 2 * h, 2 * w, co, device=dev) * DATA_SCALE).to(ADT)
    wt = torch.randn(co, c1, 3, 3, device=dev) * 0.05 * DATA_SCALE
    ax, adz = _amax(x), _amax(dz)
    gf3 = 2.0 * N * (2 * h) * (2 * w) * co * 9 * c1 / 1e9
    wp, wd = ops.phase_weights(wt, RCF_PHASE_UP2X_FWD), ops.phase_weights(wt, RCF_PHASE_UP2X_DGRAD)
    awp, awd = _amax(wp), _amax(wd)
    dm = ops.make_up2x_fwd_desc(N, h, w, c1, co, 0, 0, phase_out=True)
    qm, pm = _pack4(dm, wp, awp)
    z = torch.empty(N, 2 * h, 2 * w, co, device=dev, dtype=ADT)
    part = torch.empty(qm.n_partials, 2, co, device=dev, dtype=torch.float64)
    t_f = _timeit(lambda: ops.conv_fwd(dm, x, None, pm, z, part, scales=ops.make_scales(ax, None, awp) if F16 else None))
    dd = ops.make_up2x_dgrad_desc(N, h, w, c1, co, 0, 0, False, phase_sum=True)
    qd, pd = _pack4(dd, wd, awd)
    dx = torch.empty(N, h, w, c1, device=dev, dtype=ADT)
    t_d = _timeit(lambda: ops.conv_fwd(dd, dz, None, pd, dx, None, scales=ops.make_scales(adz, None, awd) if F16 else None))
    dwp = torch.empty(4, co, c1, 2, 2, device=dev)
    dw = torch.empty(co, c1, 3, 3, device=dev)
    wsm = torch.empty(max(1, qm.wgrad_workspace_floats), device=dev)

    def wg():
        ops.conv_wgrad(dm, x, None, dz, dwp, wsm, scales=ops.make_scales(ax, None, None, adz) if (F16 and qm.wgrad_kernel_id >= 50000) else None)
        ops.phase_wgrad_fold(dwp, dw)
    t_w = _timeit(wg)
    print('%-34s %9.1f | %8.3f %7.1f | %8.3f %7.1f | %8.3f %7.1f' % (name, gf3, t_f, gf3 / t_f, t_d, gf3 / t_d, t_w, gf3 / t_w))
for name, c1, co, h, w in S2:
    if flt and flt not in name:
        continue
    fwd = ops.make_fwd_desc(N, h, w, c1, 0, co, 3, 2)
    x = (torch.randn(N, h, w, c1, device=dev) * DATA_SCALE).to(ADT)
    dz = (torch.randn(N, fwd.h_out, fwd.w_out, co, device=dev) * DATA_SCALE).to(ADT)
    wt = torch.randn(co, c1, 3, 3, device=dev) * 0.05 * DATA_SCALE
    ax, adz = _amax(x), _amax(dz)
    gf3 = ops.algorithmic_flops(fwd) / 1e9
    wd = ops.phase_weights(wt, RCF_PHASE_S2_DGRAD)
    awd = _amax(wd)
    dx = torch.empty(N, h, w, c1, device=dev, dtype=ADT)
    try:
        dm = ops.make_s2_dgrad_desc(fwd, 0, 0, False, phase_out=True)
        qd, pd = _pack4(dm, wd, awd)
        t_d = _timeit(lambda: ops.conv_fwd(dm, dz, None, pd, dx, None, scales=ops.make_scales(adz, None, awd) if F16 else None))
    except ops._lib.RcfError:   # the exact three-plane tier: four launches
        ds, ps = [], []
        for ph in range(4):
            d1 = ops.make_s2_dgrad_desc(fwd, ph >> 1, ph & 1, False)
            q1 = ops.conv_query(d1)
            p1 = torch.empty(q1.packed_weight_floats, device=dev)
            ops.conv_pack(d1, wd[ph], p1)
            ds.append(d1); ps.append(p1)
        t_d = _timeit(lambda: [ops.conv_fwd(ds[ph], dz, None, ps[ph], dx, None) for ph in range(4)])
    wm = ops.make_s2_wgrad_desc(fwd, 0, 0, all_phases=True)
    qw = ops.conv_query(wm)
    dwp = torch.empty(4, co, c1, 2, 2, device=dev)
    dw = torch.empty(co, c1, 3, 3, device=dev)
    wsm = torch.empty(max(1, qw.wgrad_workspace_floats), device=dev)

    def wg():
        ops.conv_wgrad(wm, x, None, dz, dwp, wsm, scales=ops.make_scales(ax, None, None, adz) if (F16 and qw.wgrad_kernel_id >= 50000) else None)
        ops.phase_wgrad_gather_s2(dwp, dw)
    t_w = _timeit(wg)
    print('%-34s %9.1f | %8s %7s | %8.3f %7.1f | %8.3f %7.1f' % (name, gf3, '-', '-', t_d, gf3 / t_d, t_w, gf3 / t_w))
