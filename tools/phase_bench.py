'''Micro-benchmark (GPU box) of the exact-2x UpConv layers as the engine runs them: four 2x2 phase convolutions forward, the merged
phase_sum input gradient, four phase weight gradients.  usage: python tools/phase_bench.py [reps]'''
import sys, os
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
import rcf_amd
from rcf_amd import ops
from rcf_amd._lib import RCF_PHASE_UP2X_FWD, RCF_PHASE_UP2X_DGRAD

N = 8
LAYERS = [('deconv0.deconv 64->32 450x800 -> 900x1600', 64, 32, 450, 800), ('deconv1.deconv 64->64 225x400 -> 450x800', 64, 64, 225, 400),
          ('deconv2.deconv 128->64 113x200 -> 226x400', 128, 64, 113, 200), ('deconv3.deconv 256->128 57x100', 256, 128, 57, 100)]
reps = int(sys.argv[1]) if len(sys.argv) > 1 else 10
dev = 'cuda'
ops.set_precision(os.environ.get('RCF_BENCH_PREC', 'fp32'))   # 'f16x2': the two-plane kernels (unscaled: randn data sits in fp16's range)


def timeit(fn):
    fn(); torch.cuda.synchronize()
    e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    e0.record()
    for _ in range(reps): fn()
    e1.record(); torch.cuda.synchronize()
    return e0.elapsed_time(e1) / reps


print('%-46s %8s | %8s %7s | %8s %7s | %8s %7s' % ('layer', 'GF(4/9)', 'fwd ms', 'TF/s', 'dgrad ms', 'TF/s', 'wgrad ms', 'TF/s'))
for name, c1, co, h, w in LAYERS:
    x = torch.randn(N, h, w, c1, device=dev)
    wt = torch.randn(co, c1, 3, 3, device=dev) * 0.05
    z = torch.empty(N, 2 * h, 2 * w, co, device=dev)
    dz = torch.randn_like(z)
    wp = ops.phase_weights(wt, RCF_PHASE_UP2X_FWD)
    descs = [ops.make_up2x_fwd_desc(N, h, w, c1, co, ph >> 1, ph & 1) for ph in range(4)]
    infos = [ops.conv_query(d) for d in descs]
    packs = []
    for ph in range(4):
        p = torch.empty(infos[ph].packed_weight_floats, device=dev)
        ops.conv_pack(descs[ph], wp[ph], p)
        packs.append(p)
    gf = 2.0 * N * (2 * h) * (2 * w) * co * c1 * 4 / 1e9    # 4 of the 9 taps' MACs

    def fwd():
        for ph in range(4): ops.conv_fwd(descs[ph], x, None, packs[ph], z, None)
    wd = ops.phase_weights(wt, RCF_PHASE_UP2X_DGRAD)
    dd = ops.make_up2x_dgrad_desc(N, h, w, c1, co, 0, 0, False, phase_sum=True)
    qi = ops.conv_query(dd)
    pd = torch.empty(4 * qi.packed_weight_floats, device=dev)
    for ph in range(4): ops.conv_pack(dd, wd[ph], pd[ph * qi.packed_weight_floats:(ph + 1) * qi.packed_weight_floats])
    dx = torch.empty_like(x)
    dwp = torch.empty(4, co, c1, 2, 2, device=dev)
    wss = [torch.empty(max(1, i.wgrad_workspace_floats), device=dev) for i in infos]

    def wgrad():
        for ph in range(4): ops.conv_wgrad(descs[ph], x, None, dz, dwp[ph], wss[ph])
    t_f, t_d, t_w = timeit(fwd), timeit(lambda: ops.conv_fwd(dd, dz, None, pd, dx, None)), timeit(wgrad)
    print('%-46s %8.1f | %8.3f %7.1f | %8.3f %7.1f | %8.3f %7.1f   (kernel ids %d / %d / %d)' % (name, gf, t_f, gf / t_f, t_d, gf / t_d, t_w, gf / t_w,
          infos[0].kernel_id, qi.kernel_id, infos[0].wgrad_kernel_id))
