'''Where the waves of conv_split_kernel spend their cycles (s_memtime stamps, diagnostics build of the library).
Build (container):  hipcc --offload-arch=gfx950 -O3 -std=c++17 -fPIC -shared -DRCF_PHASE_TIMING -I include \
                        radar-camera-fusion-depth_amd/csrc/*.hip -o tools/probe/librcf_hip_timing.so
Run (GPU box):      RCF_HIP_LIB=tools/probe/librcf_hip_timing.so [RCF_BENCH_PREC=bf16] python tools/phase_timing.py'''
import ctypes, os, sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
import rcf_amd
from rcf_amd import ops, _lib

LAYERS = [('blocks2 64->64 @225x400', 64, 0, 64, 225, 400, None), ('blocks3 128->128 @113x200', 128, 0, 128, 113, 200, None),
          ('deconv0.conv 32->32 @900x1600', 32, 0, 32, 900, 1600, None), ('deconv1.conv 64+32->64 @450x800', 64, 32, 64, 450, 800, None),
          ('blocks4 256->256 @57x100', 256, 0, 256, 57, 100, None)]
DMA_NAMES = ['DMA wait + barrier', 'address arithmetic + DMA issue (next item)', 'acc init + MFMAs + LDS reads', 'epilogue', '-', '-', '-', 'whole wave']
NAMES = ['row prologue (B reads, DMA/load issue)', 'row MFMAs + interleaved LDS reads', 'output epilogue', 'barrier: A tile free',
         'store_a (load wait, split, ds_write)', 'DMA wait + barrier after store_a', 'DMA wait + barrier between rows', 'whole wave']
lib = _lib.load()
ops.set_precision(os.environ.get('RCF_BENCH_PREC', 'fp32'))
ADT = ops.act_dtype()
fn = lib.rcf_debug_phase_cycles_b16impl if ADT == torch.bfloat16 else lib.rcf_debug_phase_cycles
fn.argtypes = [ctypes.c_void_p, ctypes.c_int]
buf = (ctypes.c_ulonglong * 8)()
N = 8
for name, c1, c2, co, h, w, _ in LAYERS:
    x1 = torch.randn(N, h, w, c1, device='cuda').to(ADT)
    x2 = torch.randn(N, h, w, c2, device='cuda').to(ADT) if c2 else None
    wt = torch.randn(co, c1 + c2, 3, 3, device='cuda') * 0.05
    desc = ops.make_fwd_desc(N, h, w, c1, c2, co, 3, 1)
    info = ops.conv_query(desc)
    packed = torch.empty(info.packed_weight_floats, device='cuda')
    ops.conv_pack(desc, wt, packed)
    z = torch.empty(N, h, w, co, device='cuda', dtype=ADT)
    part = torch.empty(info.n_partials, 2, co, device='cuda', dtype=torch.float64)
    for _ in range(3): ops.conv_fwd(desc, x1, x2, packed, z, part)
    fn(None, 1)
    reps = 5
    for _ in range(reps): ops.conv_fwd(desc, x1, x2, packed, z, part)
    fn(buf, 1)
    tot = buf[7]
    print('%s (kernel id %d)' % (name, info.kernel_id))
    names = DMA_NAMES if (ADT == torch.bfloat16 and os.environ.get('RCF_B16_DMA', '1') != '0') else NAMES
    for i in range(7):
        print('   %-42s %5.1f %%' % (names[i], 100.0 * buf[i] / tot))
    print('   %-42s %5.1f %%   (%.0f cycles per wave per launch)' % ('other (tile bookkeeping, acc init)', 100.0 * (tot - sum(buf[:7])) / tot, tot / reps / (512 * 4)))
