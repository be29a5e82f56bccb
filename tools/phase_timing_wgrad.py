'''Where the waves of conv_wgrad_split_kernel spend their cycles (diagnostics build, see tools/phase_timing.py).
Run (GPU box): RCF_HIP_LIB=tools/probe/librcf_hip_timing.so [RCF_BENCH_PREC=bf16] python tools/phase_timing_wgrad.py'''
import ctypes, os, sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
import rcf_amd
from rcf_amd import ops, _lib

LAYERS = [('blocks2 64->64 @225x400', 64, 0, 64, 225, 400), ('blocks3 128->128 @113x200', 128, 0, 128, 113, 200),
          ('deconv0.conv 32->32 @900x1600', 32, 0, 32, 900, 1600), ('deconv1.conv 64+32->64 @450x800', 64, 32, 64, 450, 800)]
NAMES = ['barrier: LDS free', 'wait global loads + transpose to LDS', 'publishing barrier', 'address arithmetic + load issue (next tile)',
         'MFMA steps', 'slice reduction + partial write', '-', 'whole wave']
lib = _lib.load()
ops.set_precision(os.environ.get('RCF_BENCH_PREC', 'fp32'))
ADT = ops.act_dtype()
fn = lib.rcf_debug_phase_cycles_b16impl if ADT == torch.bfloat16 else lib.rcf_debug_phase_cycles
fn.argtypes = [ctypes.c_void_p, ctypes.c_int]
buf = (ctypes.c_ulonglong * 8)()
N = 8
for name, c1, c2, co, h, w in LAYERS:
    x1 = torch.randn(N, h, w, c1, device='cuda').to(ADT)
    x2 = torch.randn(N, h, w, c2, device='cuda').to(ADT) if c2 else None
    dz = torch.randn(N, h, w, co, device='cuda').to(ADT)
    desc = ops.make_fwd_desc(N, h, w, c1, c2, co, 3, 1)
    info = ops.conv_query(desc)
    ws = torch.empty(max(1, info.wgrad_workspace_floats), device='cuda')
    dw = torch.empty(co, c1 + c2, 3, 3, device='cuda')
    for _ in range(2): ops.conv_wgrad(desc, x1, x2, dz, dw, ws)
    fn(None, 1)
    reps = 5
    for _ in range(reps): ops.conv_wgrad(desc, x1, x2, dz, dw, ws)
    fn(buf, 1)
    tot = buf[7]
    print('%s (wgrad kernel id %d)' % (name, info.wgrad_kernel_id))
    for i in range(6):
        print('   %-46s %5.1f %%' % (NAMES[i], 100.0 * buf[i] / tot))
    print('   %-46s %5.1f %%' % ('other', 100.0 * (tot - sum(buf[:6])) / tot))
