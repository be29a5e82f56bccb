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
# conv_wgrad_tr_kernel (round 6; wgrad kernel id with hundreds digit 3 / 7): consumer waves fill slots 0, 4, 5, 7, producer waves 1, 2, 6
TR_CONSUMER = [(4, 'MFMA steps (+ interleaved tr reads, operand shifts)'), (0, 'barrier: waiting for the producers / the other consumers'),
               (5, 'slice reduction + partial write')]
TR_PRODUCER = [(1, 'stage the next tile (bf16: DMA issue; fp32: loads, plane split, LDS writes)'), (2, 'idle at the barrier')]
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
    if (info.wgrad_kernel_id // 100) % 10 in (3, 7):
        print('   consumer waves (4 of 8):')
        for i, nm in TR_CONSUMER:
            print('      %-72s %5.1f %%' % (nm, 100.0 * buf[i] / buf[7]))
        print('      %-72s %5.1f %%' % ('other', 100.0 * (buf[7] - sum(buf[i] for i, _ in TR_CONSUMER)) / buf[7]))
        print('   producer waves (4 of 8):')
        for i, nm in TR_PRODUCER:
            print('      %-72s %5.1f %%' % (nm, 100.0 * buf[i] / buf[6]))
        print('      %-72s %5.1f %%' % ('other (per-thread constants, first tile)', 100.0 * (buf[6] - buf[1] - buf[2]) / buf[6]))
        continue
    for i in range(6):
        print('   %-46s %5.1f %%' % (NAMES[i], 100.0 * buf[i] / tot))
    print('   %-46s %5.1f %%' % ('other', 100.0 * (tot - sum(buf[:6])) / tot))
