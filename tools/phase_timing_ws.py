'''Where the consumer and the producer waves of conv_split_ws_kernel spend their cycles (s_memtime stamps, diagnostics build).
Build (container):  hipcc --offload-arch=gfx950 -O3 -std=c++17 -fPIC -shared -DRCF_PHASE_TIMING -I include \
                        radar-camera-fusion-depth_amd/csrc/*.hip -o tools/probe/librcf_hip_timing.so
Run (GPU box):      RCF_HIP_LIB=tools/probe/librcf_hip_timing.so python tools/phase_timing_ws.py'''
import ctypes, os, sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
import rcf_amd
from rcf_amd import ops, _lib

LAYERS = [('blocks2 64->64 @225x400', 64, 0, 64, 225, 400), ('blocks3 128->128 @113x200', 128, 0, 128, 113, 200),
          ('deconv0.conv 32->32 @900x1600', 32, 0, 32, 900, 1600), ('deconv1.conv 64+32->64 @450x800', 64, 32, 64, 450, 800)]
lib = _lib.load()
ops.set_precision('f16x2')
fn = lib.rcf_debug_phase_cycles
fn.argtypes = [ctypes.c_void_p, ctypes.c_int]
buf = (ctypes.c_ulonglong * 8)()
N = 8
for name, c1, c2, co, h, w in LAYERS:
    x1 = torch.randn(N, h, w, c1, device='cuda')
    x2 = torch.randn(N, h, w, c2, device='cuda') if c2 else None
    wt = torch.randn(co, c1 + c2, 3, 3, device='cuda') * 0.05
    desc = ops.make_fwd_desc(N, h, w, c1, c2, co, 3, 1)
    info = ops.conv_query(desc)
    packed = torch.empty(info.packed_weight_floats, device='cuda')
    ops.conv_pack(desc, wt, packed)
    z = torch.empty(N, h, w, co, device='cuda')
    part = torch.empty(info.n_partials, 2, co, device='cuda', dtype=torch.float64)
    for _ in range(3): ops.conv_fwd(desc, x1, x2, packed, z, part)
    fn(None, 1)
    reps = 5
    for _ in range(reps): ops.conv_fwd(desc, x1, x2, packed, z, part)
    fn(buf, 1)
    print('%s (kernel id %d, %d workgroups)' % (name, info.kernel_id, info.n_partials))
    ct, pt = buf[3], buf[6]
    for i, nm in ((0, 'consumer: row = B reads + DMA issue + MFMAs'), (1, 'consumer: epilogue / next A reads'), (2, 'consumer: DMA wait + barrier')):
        print('   %-46s %5.1f %%' % (nm, 100.0 * buf[i] / max(ct, 1)))
    print('   %-46s %5.1f %%   (%.0f cycles per consumer wave per launch)' % ('consumer: other', 100.0 * (ct - sum(buf[:3])) / max(ct, 1), ct / reps / (info.n_partials * 4)))
    for i, nm in ((4, 'producer: convert + ds_write + reload'), (5, 'producer: barrier')):
        print('   %-46s %5.1f %%' % (nm, 100.0 * buf[i] / max(pt, 1)))
    print('   %-46s %5.1f %%   (%.0f cycles per producer wave per launch)' % ('producer: other (prologue, tile setup)', 100.0 * (pt - buf[4] - buf[5]) / max(pt, 1), pt / reps / (info.n_partials * 2)))
