'''GPU probe of the two-plane fp16 split ("the lead for the exact tier"): the library built with -DRCF_X3_F16
carries fp16 planes (11 + 11 significant bits, three products, NO scaling) in its two-plane kernels, so on data inside fp16's range
this measures what that arithmetic delivers on the hardware -- error against fp64 and TF/s -- next to the exact fp32 tier and the
bf16 two-plane tier of the shipped library.
Build (container):  hipcc --offload-arch=gfx950 -O3 -std=c++17 -fPIC -DRCF_X3_F16 -I include -c radar-camera-fusion-depth_amd/csrc/rcf_conv.hip \
                        -o build/objf/rcf_conv.o && hipcc --offload-arch=gfx950 -shared -fPIC build/objf/rcf_conv.o <the other build/obj/*.o> \
                        -o tools/probe/librcf_hip_f16probe.so
Run (GPU box):      python tools/f16_plane_probe.py            (starts itself once per library)'''
import os, subprocess, sys
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)

if len(sys.argv) == 1:
    for tag, lib in (('shipped library', None), ('-DRCF_X3_F16 build', os.path.join(ROOT, 'tools', 'probe', 'librcf_hip_f16probe.so'))):
        env = dict(os.environ)
        if lib:
            env['RCF_HIP_LIB'] = lib
        print('==== %s' % tag, flush=True)
        subprocess.run([sys.executable, os.path.abspath(__file__), 'child'] + (['f16'] if lib else []), env=env, check=True)
    raise SystemExit(0)

import torch
import torch.nn.functional as F
import rcf_amd  # noqa: F401
from rcf_amd import ops

f16 = len(sys.argv) > 2
modes = ['bf16x3'] if f16 else ['fp32', 'bf16x3']
g = torch.Generator().manual_seed(1)
rnd = lambda *s: (torch.rand(*s, generator=g) * 2 - 1)
nhwc = lambda t: t.permute(0, 2, 3, 1).contiguous().cuda()


def timeit(fn, reps=10):
    fn(); torch.cuda.synchronize()
    e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    e0.record()
    for _ in range(reps): fn()
    e1.record(); torch.cuda.synchronize()
    return e0.elapsed_time(e1) / reps


for mode in modes:
    ops.set_precision(mode)
    label = 'fp16 two planes, 3 products' if f16 else {'fp32': 'bf16 three planes, 6 products (exact tier)', 'bf16x3': 'bf16 two planes, 3 products'}[mode]
    # accuracy on a 64 -> 64 layer, uniform data (inside fp16's range), against fp64
    n, c, h, w = 2, 64, 40, 56
    x, wt, dz = rnd(n, c, h, w), rnd(c, c, 3, 3) / 24, rnd(n, c, h, w)
    d = ops.make_fwd_desc(n, h, w, c, 0, c, 3, 1)
    info = ops.conv_query(d)
    packed = torch.empty(info.packed_weight_floats, device='cuda')
    ops.conv_pack(d, wt.cuda(), packed)
    out = torch.empty(n, h, w, c, device='cuda')
    ops.conv_fwd(d, nhwc(x), None, packed, out, None)
    ref = F.conv2d(x.double(), wt.double(), padding=1)
    mag = F.conv2d(x.double().abs(), wt.double().abs(), padding=1)
    e_f = float(((out.cpu().permute(0, 3, 1, 2).double() - ref).abs() / mag).max())
    dw = torch.empty_like(wt).cuda()
    ws = torch.empty(max(1, info.wgrad_workspace_floats), device='cuda')
    ops.conv_wgrad(d, nhwc(x), None, nhwc(dz), dw, ws)
    wref = torch.nn.grad.conv2d_weight(x.double(), wt.shape, dz.double(), padding=1)
    wmag = torch.nn.grad.conv2d_weight(x.double().abs(), wt.shape, dz.double().abs(), padding=1)
    e_w = float(((dw.cpu().double() - wref).abs() / wmag).max())
    # speed on two FusionNet layers at batch 8
    sp = []
    for (c1, c2, co, hh, ww) in [(64, 32, 64, 450, 800), (64, 0, 64, 225, 400)]:
        dd = ops.make_fwd_desc(8, hh, ww, c1, c2, co, 3, 1)
        qi = ops.conv_query(dd)
        x1 = torch.randn(8, hh, ww, c1, device='cuda')
        x2 = torch.randn(8, hh, ww, c2, device='cuda') if c2 else None
        wt2 = torch.randn(co, c1 + c2, 3, 3, device='cuda') * 0.05
        pk = torch.empty(qi.packed_weight_floats, device='cuda')
        ops.conv_pack(dd, wt2, pk)
        o = torch.empty(8, hh, ww, co, device='cuda')
        dzz = torch.randn(8, hh, ww, co, device='cuda')
        dww = torch.empty_like(wt2)
        wsb = torch.empty(max(1, qi.wgrad_workspace_floats), device='cuda')
        gf = ops.algorithmic_flops(dd) / 1e9
        sp.append((gf / timeit(lambda: ops.conv_fwd(dd, x1, x2, pk, o, None)), gf / timeit(lambda: ops.conv_wgrad(dd, x1, x2, dzz, dww, wsb))))
    print('%-44s max|err|/sum|a||b|: fwd %.1e  wgrad %.1e | 64+32->64 @450x800: fwd %4.0f wgrad %4.0f TF/s | 64->64 @225x400: fwd %4.0f wgrad %4.0f TF/s'
          % (label, e_f, e_w, sp[0][0], sp[0][1], sp[1][0], sp[1][1]), flush=True)
ops.set_precision('fp32')
