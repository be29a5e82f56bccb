'''Micro-benchmark (GPU box): the exact-2x up-convolutions of the decoder (conv3x3 of a nearest-upsampled x as four 2x2 phase
convolutions), forward, through the C ABI -- the four per-phase launches against the one-launch form (rcf_conv_desc.phase_sum == 2).
usage: [RCF_BENCH_PREC=f16x2|fp32|bf16] [RCF_UP2X_MERGED=0|1] python tools/up2x_bench.py [batch] [reps]'''
import os
import sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
import rcf_amd  # noqa: F401
from rcf_amd import ops
from rcf_amd._lib import RCF_PHASE_UP2X_FWD

n = int(sys.argv[1]) if len(sys.argv) > 1 else 8
reps = int(sys.argv[2]) if len(sys.argv) > 2 else 10
ops.set_precision(os.environ.get('RCF_BENCH_PREC', 'bf16'))
ADT = ops.act_dtype()
eb = 2 if ADT == torch.bfloat16 else 4
LAYERS = [(64, 32, 450, 800), (64, 64, 225, 400), (128, 64, 113, 200), (256, 128, 57, 100), (256, 256, 29, 50)]


def timeit(fn):
    fn(); torch.cuda.synchronize()
    e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    e0.record()
    for _ in range(reps):
        fn()
    e1.record(); torch.cuda.synchronize()
    return e0.elapsed_time(e1) / reps


print('%-26s | %10s %8s | %10s %8s | bitwise' % ('up-2x forward, batch %d' % n, '4 launches', 'TB/s', '1 launch', 'TB/s'))
for c1, co, h, w in LAYERS:
    x = torch.randn(n, h, w, c1, device='cuda').to(ADT)
    wt = torch.randn(co, c1, 3, 3, device='cuda') * 0.05
    wp = ops.phase_weights(wt, RCF_PHASE_UP2X_FWD)
    amax = x.float().abs().max().reshape(1) if ops.get_precision() == ops._lib.RCF_PREC_F16X2 else None
    z4 = torch.empty(n, 2 * h, 2 * w, co, device='cuda', dtype=ADT)
    z1 = torch.empty_like(z4)
    descs, packs = [], []
    for ph in range(4):
        d = ops.make_up2x_fwd_desc(n, h, w, c1, co, ph >> 1, ph & 1)
        info = ops.conv_query(d)
        p = torch.empty(info.packed_weight_floats, device='cuda')
        ops.conv_pack(d, wp[ph], p)
        descs.append(d); packs.append(p)
    part4 = torch.empty(4 * info.n_partials, 2, co, device='cuda', dtype=torch.float64)
    npart = info.n_partials
    wmax = wp.abs().max().reshape(1) if amax is not None else None
    kw = {} if amax is None else {'scales': ops.make_scales(amax, None, wmax)}

    def four():
        for ph in range(4):
            ops.conv_fwd(descs[ph], x, None, packs[ph], z4, part4[ph * npart:(ph + 1) * npart], **kw)
    dm = ops.make_up2x_fwd_desc(n, h, w, c1, co, 0, 0, phase_out=True)
    try:
        im = ops.conv_query(dm)
    except Exception as e:
        print('%3d->%3d @ %3dx%3d: one-launch form refused (%s)' % (c1, co, h, w, str(e)[:60]))
        continue
    pm = torch.empty(4 * im.packed_weight_floats, device='cuda')
    for ph in range(4):
        ops.conv_pack(dm, wp[ph], pm[ph * im.packed_weight_floats:(ph + 1) * im.packed_weight_floats])
    part1 = torch.empty(im.n_partials, 2, co, device='cuda', dtype=torch.float64)
    one = lambda: ops.conv_fwd(dm, x, None, pm, z1, part1, **kw)
    t4, t1 = timeit(four), timeit(one)
    gb = (x.numel() + z4.numel()) * eb / 1e9
    print('%3d->%3d @ %3dx%3d %8.2f GB | %8.1f us %8.2f | %8.1f us %8.2f | %s' % (c1, co, h, w, gb, 1000 * t4, gb / t4, 1000 * t1, gb / t1,
                                                                                bool(torch.equal(z1, z4))))
    if 'timing' in os.environ.get('RCF_HIP_LIB', ''):   # the -DRCF_PHASE_TIMING build (tools/phase_timing.py): where the waves' cycles go
        import ctypes
        lib = ops._lib.load()
        fn = lib.rcf_debug_phase_cycles_b16impl if ADT == torch.bfloat16 else lib.rcf_debug_phase_cycles
        fn.argtypes = [ctypes.c_void_p, ctypes.c_int]
        buf = (ctypes.c_ulonglong * 8)()
        fn(None, 1)
        for _ in range(reps):
            one()
        fn(buf, 1)
        print('      one launch, %% of wave cycles: ' + '  '.join('[%d] %.1f' % (i, 100.0 * buf[i] / buf[7]) for i in range(7)))

print()
print('%-26s | %10s %8s | %10s %8s | max rel diff' % ('up-2x weight grad, batch %d' % n, '4 launches', 'TB/s', '1 launch', 'TB/s'))
for c1, co, h, w in LAYERS:
    x = torch.randn(n, h, w, c1, device='cuda').to(ADT)
    dz = (torch.randn(n, 2 * h, 2 * w, co, device='cuda') * 1e-2).to(ADT)
    f16 = ops.get_precision() == ops._lib.RCF_PREC_F16X2
    ax = x.float().abs().max().reshape(1) if f16 else None
    adz = dz.float().abs().max().reshape(1) if f16 else None
    dwp = torch.empty(4, co, c1, 2, 2, device='cuda')
    dwm = torch.empty_like(dwp)
    descs, wss = [], []
    for ph in range(4):
        d = ops.make_up2x_fwd_desc(n, h, w, c1, co, ph >> 1, ph & 1)
        qi = ops.conv_query(d)
        descs.append(d); wss.append(torch.empty(max(1, qi.wgrad_workspace_floats), device='cuda'))
    kw = {'scales': ops.make_scales(ax, None, None, adz)} if (f16 and qi.wgrad_kernel_id >= 50000) else {}

    def four():
        for ph in range(4):
            ops.conv_wgrad(descs[ph], x, None, dz, dwp[ph], wss[ph], **kw)
    dm = ops.make_up2x_fwd_desc(n, h, w, c1, co, 0, 0, phase_out=True)
    qm = ops.conv_query(dm)
    wsm = torch.empty(max(1, qm.wgrad_workspace_floats), device='cuda')
    one = lambda: ops.conv_wgrad(dm, x, None, dz, dwm, wsm, **kw)
    try:
        t4, t1 = timeit(four), timeit(one)
    except Exception as e:
        print('%3d->%3d @ %3dx%3d: %s' % (c1, co, h, w, str(e)[:80]))
        continue
    gb = (x.numel() + dz.numel()) * eb / 1e9
    diff = float((dwm.double() - dwp.double()).abs().max() / dwp.double().abs().max())
    print('%3d->%3d @ %3dx%3d %8.2f GB | %8.1f us %8.2f | %8.1f us %8.2f | %.2e' % (c1, co, h, w, gb, 1000 * t4, gb / t4, 1000 * t1, gb / t1, diff))

print()
print('%-30s | %10s %8s | %10s %8s | bitwise' % ('stride-2 input grad, batch %d' % n, '4 launches', 'TB/s', '1 launch', 'TB/s'))
from rcf_amd._lib import RCF_PHASE_S2_DGRAD
for cin, cout, h, w in [(32, 64, 450, 800), (64, 128, 225, 400), (128, 256, 113, 200), (256, 256, 57, 100), (256, 256, 29, 50)]:
    fwd = ops.make_fwd_desc(n, h, w, cin, 0, cout, 3, 2)
    dz = (torch.randn(n, fwd.h_out, fwd.w_out, cout, device='cuda') * 1e-2).to(ADT)
    wt = torch.randn(cout, cin, 3, 3, device='cuda') * 0.05
    wd = ops.phase_weights(wt, RCF_PHASE_S2_DGRAD)
    f16 = ops.get_precision() == ops._lib.RCF_PREC_F16X2
    adz = dz.float().abs().max().reshape(1) if f16 else None
    aw = wd.abs().max().reshape(1) if f16 else None
    kw = {'scales': ops.make_scales(adz, None, aw)} if f16 else {}
    dx4 = torch.empty(n, h, w, cin, device='cuda', dtype=ADT)
    dx1 = torch.empty_like(dx4)
    descs, packs = [], []
    for ph in range(4):
        d = ops.make_s2_dgrad_desc(fwd, ph >> 1, ph & 1, False)
        info = ops.conv_query(d)
        p = torch.empty(info.packed_weight_floats, device='cuda')
        ops.conv_pack(d, wd[ph], p, aw) if f16 else ops.conv_pack(d, wd[ph], p)
        descs.append(d); packs.append(p)

    def four():
        for ph in range(4):
            ops.conv_fwd(descs[ph], dz, None, packs[ph], dx4, None, **kw)
    dm = ops.make_s2_dgrad_desc(fwd, 0, 0, False, phase_out=True)
    try:
        im = ops.conv_query(dm)
    except Exception as e:
        print('%3d->%3d @ %3dx%3d: merged form refused (%s)' % (cin, cout, h, w, str(e)[:60]))
        continue
    pm = torch.empty(4 * im.packed_weight_floats, device='cuda')
    for ph in range(4):
        dst = pm[ph * im.packed_weight_floats:(ph + 1) * im.packed_weight_floats]
        ops.conv_pack(dm, wd[ph], dst, aw) if f16 else ops.conv_pack(dm, wd[ph], dst)
    one = lambda: ops.conv_fwd(dm, dz, None, pm, dx1, None, **kw)
    t4, t1 = timeit(four), timeit(one)
    gb = (dz.numel() + dx4.numel()) * eb / 1e9
    print('%3d<-%3d @ %3dx%3d %8.2f GB | %8.1f us %8.2f | %8.1f us %8.2f | %s' % (cin, cout, h, w, gb, 1000 * t4, gb / t4, 1000 * t1, gb / t1,
                                                                                bool(torch.equal(dx1, dx4))))
