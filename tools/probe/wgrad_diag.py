'''Timing experiment (GPU box) on a DIAGNOSTICS build of the library (hipcc ... -DRCF_WGRAD_DIAG -shared -o tools/probe/librcf_hip_wdiag.so, then
RCF_HIP_LIB=tools/probe/librcf_hip_wdiag.so): conv_wgrad_tr_kernel without its partial write / with the consumers alone / with the
producers alone (RCF_WGRAD_DIAG=1/2/3: wrong results, the same schedule otherwise).  The shipped library ignores the variable.'''
import os, sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__)))))
import torch
import rcf_amd  # noqa: F401
from rcf_amd import ops

def timeit(fn, reps=20):
    fn(); torch.cuda.synchronize()
    e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    e0.record()
    for _ in range(reps): fn()
    e1.record(); torch.cuda.synchronize()
    return e0.elapsed_time(e1) * 1000.0 / reps

for prec in ('f16x2', 'bf16'):
    ops.set_precision(prec)
    adt = ops.act_dtype()
    for name, c1, co, h, w in (('64->64 @225x400', 64, 64, 225, 400), ('128->128 @113x200', 128, 128, 113, 200), ('32->32 @900x1600', 32, 32, 900, 1600)):
        d = ops.make_fwd_desc(8, h, w, c1, 0, co, 3, 1)
        x1 = torch.randn(8, h, w, c1, device='cuda').to(adt)
        dz = torch.randn(8, h, w, co, device='cuda').to(adt)
        info = ops.conv_query(d)
        wsb = torch.empty(max(1, info.wgrad_workspace_floats), device='cuda')
        dw = torch.empty(co, c1, 3, 3, device='cuda')
        r = {}
        for diag in ('0', '1', '2', '3', '0', '1', '2', '3'):
            os.environ['RCF_WGRAD_DIAG'] = diag
            r[diag] = min(r.get(diag, 1e9), timeit(lambda: ops.conv_wgrad(d, x1, None, dz, dw, wsb)))
        print('%s %s: %.1f us | no partial write %.1f | consumers alone (no staging after the first tile) %.1f | producers alone (no MFMA steps) %.1f'
              % (prec, name, r['0'], r['1'], r['2'], r['3']))
os.environ.pop('RCF_WGRAD_DIAG', None)
