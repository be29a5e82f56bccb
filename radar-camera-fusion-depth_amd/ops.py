'''
Tensor-level wrappers over the C ABI (include/rcf_hip.h).  torch is used only to own device memory and the
stream; every computation below is a hand-written HIP kernel in librcf_hip.so.  All activations are fp32 NHWC
contiguous torch tensors of shape (N, H, W, C); weights are OIHW like torch.nn.Conv2d.weight.
'''

import ctypes
import os

import torch

from . import _lib
from ._lib import (ConvDesc, ConvInfo, RCF_ACT_LEAKY_RELU, RCF_ACT_NONE, RCF_GATHER_DIRECT, RCF_GATHER_NEAREST,
                   RCF_GATHER_STRIDED2, RCF_GATHER_ZERO_INSERT, RCF_PHASE_S2_DGRAD, RCF_PHASE_UP2X_DGRAD, RCF_PHASE_UP2X_FWD,
                   RCF_W_DGRAD, RCF_W_FORWARD, check)


LAUNCHES = [0]   # C-ABI launches enqueued by this module so far (FusionNetModel's segmented capture: "did anything run since the last cut?")


_RAW_STREAM = getattr(torch._C, '_cuda_getCurrentRawStream', None)
_GET_DEVICE = getattr(torch._C, '_cuda_getDevice', None)


def _stream():
    LAUNCHES[0] += 1
    # the current stream of the CURRENT device: the model entry points (FusionNetModel / RadarNetModel forward, backward, loss) make
    # the tensors' device current for the whole call (torch.cuda.device guard); direct users of this module do the same.
    # ~1,100 launches per training step pass through here: the raw accessors (what torch.cuda.current_stream() wraps in a Stream
    # object after resolving the device through torch.cuda.is_available() and os.environ) cost ~0.3 us instead of ~2.5 us
    if _RAW_STREAM is not None and _GET_DEVICE is not None:
        return _RAW_STREAM(_GET_DEVICE())
    return torch.cuda.current_stream().cuda_stream


def _p(t):
    if t is None:
        return None
    if not t.is_cuda:
        raise _lib.RcfError('rcf ops need CUDA(HIP) tensors; got a %s tensor -- there is no CPU path' % t.device)
    if not t.is_contiguous():
        raise _lib.RcfError('rcf ops need contiguous tensors')
    return t.data_ptr()


def _f64(t):
    if t is not None and t.dtype != torch.float64:
        raise _lib.RcfError('this rcf buffer is fp64; got %s' % t.dtype)
    return _p(t)


def _f32(t):
    if t is not None and t.dtype != torch.float32:
        raise _lib.RcfError('this rcf buffer is fp32; got %s' % t.dtype)
    return _p(t)


def _a(t):
    """Pointer of an NHWC activation / gradient tensor: fp32, or bf16 in the bf16-storage configuration."""
    if t is not None and t.dtype != torch.float32 and t.dtype != torch.bfloat16:
        raise _lib.RcfError('rcf activations are fp32 or bf16; got %s' % t.dtype)
    return _p(t)


def _fn(name, *acts):
    """The entry point for the storage of the given activation tensors: NAME (fp32) or NAME_b16 (bf16); they must agree."""
    kinds = set(t.dtype for t in acts if t is not None)
    if len(kinds) > 1:
        raise _lib.RcfError('%s: activation tensors disagree in dtype: %s' % (name, sorted(str(k) for k in kinds)))
    b16 = bool(kinds) and next(iter(kinds)) == torch.bfloat16
    return getattr(_lib.load(), name + ('_b16' if b16 else ''))


def conv_out_hw(h, w, ksize, stride, pad):
    return (h + 2 * pad - ksize) // stride + 1, (w + 2 * pad - ksize) // stride + 1


_PRECISION = [0]   # rcf_conv_desc.precision of every descriptor built below (RCF_PREC_FP32 / RCF_PREC_BF16)
_STORAGE = [0]     # rcf_conv_desc.storage (RCF_STORE_FP32 / RCF_STORE_BF16)


def set_precision(p):
    '''0 / 'fp32': fp32 arithmetic and fp32 tensors (default).  'bf16': bf16 tensors in HBM + bf16 MFMA operands, fp32 accumulate
    (BASELINE.json configs 2-4).  'bf16_operands': fp32 tensors, operands rounded to bf16 in the split conv kernels.
    'f16x2': fp32 tensors, each operand of the split conv kernels as two fp16 planes of x * (its tensor's power-of-two scale) and
    three products (22-23 significant bits; RCF_PREC_F16X2 in include/rcf_hip.h) -- half the matrix work of the three-plane bf16
    split at its accuracy class; the per-tensor maxima travel as device scalars (`scales=` of conv_fwd / conv_wgrad).'''
    if p not in (0, 1, 'fp32', 'bf16', 'bf16_operands', 'f16x2'):
        raise ValueError('unknown precision %r' % (p,))
    _PRECISION[0] = 1 if p in (1, 'bf16', 'bf16_operands') else (_lib.RCF_PREC_F16X2 if p == 'f16x2' else 0)
    _STORAGE[0] = 1 if p == 'bf16' else 0


def get_precision():
    return _PRECISION[0]


def precision_of(compute_dtype):
    '''The ops.set_precision name a model's compute_dtype stands for.  'fp32' -- the reference's arithmetic, the configuration the
    benchmark metric is quoted on -- runs the split convolution kernels on two scaled fp16 planes ('f16x2': fp32-convolution-class
    errors against fp64, tests/test_hip_f16x2.py) unless RCF_FP32_TIER=3plane asks for the three-plane bf16 split ('fp32_3plane' as a
    compute_dtype does the same per model).  Everything else passes through.'''
    import os
    if compute_dtype == 'fp32':
        tier = 'fp32' if os.environ.get('RCF_FP32_TIER', FP32_TIER_DEFAULT) == '3plane' else 'f16x2'
        if not _TIER_LOGGED[0]:   # once per process: 'fp32' names an accuracy class, this line says which arithmetic delivers it
            _TIER_LOGGED[0] = True
            import logging
            logging.getLogger('rcf_amd').info(
                "compute_dtype='fp32': fp32 tensors; split convolution kernels on %s (RCF_FP32_TIER=3plane / compute_dtype='fp32_3plane' "
                "select the exact three-plane bf16 split)", 'two scaled fp16 planes, three products per multiply (RCF_PREC_F16X2)'
                if tier == 'f16x2' else 'three bf16 planes, six products per multiply (exact)')
        return tier
    if compute_dtype == 'fp32_3plane':
        return 'fp32'
    return compute_dtype


FP32_TIER_DEFAULT = 'f16x2'
_TIER_LOGGED = [False]


def act_dtype():
    '''torch dtype of the NHWC activation / gradient tensors under the current setting.'''
    return torch.bfloat16 if _STORAGE[0] else torch.float32


def _check_storage(desc, *acts):
    want = torch.bfloat16 if desc.storage else torch.float32
    for t in acts:
        if t is not None and t.dtype != want:
            raise _lib.RcfError('descriptor storage is %s but a tensor is %s' % (want, t.dtype))


def make_fwd_desc(n, h_in, w_in, c1, c2, c_out, ksize, stride, h_src1=None, w_src1=None, gather=RCF_GATHER_DIRECT):
    '''Descriptor of the reference's Conv2d (padding = ksize // 2, src/net_utils.py:61) on [src1 | src2].'''
    pad = ksize // 2
    h_out, w_out = conv_out_hw(h_in, w_in, ksize, stride, pad)
    return ConvDesc(n=n, h_in=h_in, w_in=w_in, c1=c1, c2=c2,
                    h_src1=h_in if h_src1 is None else h_src1, w_src1=w_in if w_src1 is None else w_src1,
                    gather1=gather, h_out=h_out, w_out=w_out, c_out=c_out, ksize=ksize, stride=stride, pad=pad, pad_x=pad,
                    w_mode=RCF_W_FORWARD, w_o=c_out, w_i=c1 + c2, w_i_off=0, accumulate=0,
                    out_stride=1, out_off_y=0, out_off_x=0, out_h_phys=h_out, out_w_phys=w_out, in_off_y=0, in_off_x=0, phase_sum=0, precision=_PRECISION[0], storage=_STORAGE[0])


def make_dgrad_desc(fwd, i_off, i_cnt, accumulate):
    '''
    The input gradient of the conv `fwd` (on its LOGICAL input, i.e. at the upsampled resolution when the
    forward gathered), for the input-channel slice [i_off, i_off + i_cnt): itself a stride-1 convolution of dZ
    with flipped taps and pad ksize-1-pad; a stride-2 forward makes dZ zero-dilated (a transposed convolution).
    '''
    k = fwd.ksize
    return ConvDesc(n=fwd.n, h_in=fwd.h_in, w_in=fwd.w_in, c1=fwd.c_out, c2=0,
                    h_src1=fwd.h_out, w_src1=fwd.w_out,
                    gather1=RCF_GATHER_ZERO_INSERT if fwd.stride == 2 else RCF_GATHER_DIRECT,
                    h_out=fwd.h_in, w_out=fwd.w_in, c_out=i_cnt, ksize=k, stride=1, pad=k - 1 - fwd.pad, pad_x=k - 1 - fwd.pad,
                    w_mode=RCF_W_DGRAD, w_o=fwd.w_o, w_i=fwd.w_i, w_i_off=i_off, accumulate=1 if accumulate else 0,
                    out_stride=1, out_off_y=0, out_off_x=0, out_h_phys=fwd.h_in, out_w_phys=fwd.w_in, in_off_y=0, in_off_x=0, phase_sum=0, precision=_PRECISION[0], storage=_STORAGE[0])


def make_pw_s2_dgrad_desc(fwd, accumulate):
    '''The input gradient of a 1x1 STRIDE-2 convolution `fwd` (the ResNet projections, src/net_utils.py:300-307) at the resolution of dZ:
    dX(2y, 2x) = W^T dZ(y, x) and dX is zero everywhere else, so instead of a dense 1x1 convolution over the zero-dilated dZ at the input
    resolution (make_dgrad_desc: four times the pixels, three quarters of them zeros) this is a plain 1x1 convolution of dZ whose outputs
    land at the even positions of dX (out_stride 2).  The caller zero-fills dX first unless it accumulates.  Same kernel, same dot
    products: bitwise the dense form.'''
    if fwd.ksize != 1 or fwd.stride != 2 or fwd.c2 != 0:
        raise ValueError('a 1x1 stride-2 forward descriptor is required')
    return ConvDesc(n=fwd.n, h_in=fwd.h_out, w_in=fwd.w_out, c1=fwd.c_out, c2=0, h_src1=fwd.h_out, w_src1=fwd.w_out,
                    gather1=RCF_GATHER_DIRECT, h_out=fwd.h_out, w_out=fwd.w_out, c_out=fwd.c1, ksize=1, stride=1, pad=0, pad_x=0,
                    w_mode=RCF_W_DGRAD, w_o=fwd.w_o, w_i=fwd.w_i, w_i_off=0, accumulate=1 if accumulate else 0,
                    out_stride=2, out_off_y=0, out_off_x=0, out_h_phys=fwd.h_in, out_w_phys=fwd.w_in, in_off_y=0, in_off_x=0, phase_sum=0,
                    precision=_PRECISION[0], storage=_STORAGE[0])


# ---- 2x2 phase convolutions (include/rcf_hip.h, RCF_PHASE_*) -------------------------------------------------------
def make_up2x_fwd_desc(n, hs, ws, c_in, c_out, a, b, phase_out=False):
    """Phase (a,b) of conv3x3(nearest-upsample-2x(x)): a 2x2 conv on x writing output pixels (2y+a, 2x+b).
    phase_out=True: all four phases in one launch (a, b ignored; weights of the 4 phases packed back to back; phase_sum == 2)."""
    if phase_out:
        return ConvDesc(n=n, h_in=hs, w_in=ws, c1=c_in, c2=0, h_src1=hs, w_src1=ws, gather1=RCF_GATHER_DIRECT,
                        h_out=hs, w_out=ws, c_out=c_out, ksize=2, stride=1, pad=1, pad_x=1,
                        w_mode=RCF_W_FORWARD, w_o=c_out, w_i=c_in, w_i_off=0, accumulate=0,
                        out_stride=2, out_off_y=0, out_off_x=0, out_h_phys=2 * hs, out_w_phys=2 * ws, in_off_y=0, in_off_x=0, phase_sum=2,
                        precision=_PRECISION[0], storage=_STORAGE[0])
    return ConvDesc(n=n, h_in=hs, w_in=ws, c1=c_in, c2=0, h_src1=hs, w_src1=ws, gather1=RCF_GATHER_DIRECT,
                    h_out=hs, w_out=ws, c_out=c_out, ksize=2, stride=1, pad=1 - a, pad_x=1 - b,
                    w_mode=RCF_W_FORWARD, w_o=c_out, w_i=c_in, w_i_off=0, accumulate=0,
                    out_stride=2, out_off_y=a, out_off_x=b, out_h_phys=2 * hs, out_w_phys=2 * ws, in_off_y=0, in_off_x=0, phase_sum=0, precision=_PRECISION[0], storage=_STORAGE[0])


def make_up2x_dgrad_desc(n, hs, ws, c_in, c_out, a, b, accumulate, phase_sum=False):
    """dX += 2x2 conv of phase (a,b) of dZ (read strided) with the transposed, flipped phase weights.
    phase_sum=True: all four phases in one launch (a, b ignored; weights of the 4 phases packed back to back)."""
    return ConvDesc(n=n, h_in=hs, w_in=ws, c1=c_out, c2=0, h_src1=2 * hs, w_src1=2 * ws, gather1=RCF_GATHER_STRIDED2,
                    h_out=hs, w_out=ws, c_out=c_in, ksize=2, stride=1, pad=a, pad_x=b,
                    w_mode=RCF_W_FORWARD, w_o=c_in, w_i=c_out, w_i_off=0, accumulate=1 if accumulate else 0,
                    out_stride=1, out_off_y=0, out_off_x=0, out_h_phys=hs, out_w_phys=ws, in_off_y=a, in_off_x=b,
                    phase_sum=1 if phase_sum else 0, precision=_PRECISION[0], storage=_STORAGE[0])


def make_s2_dgrad_desc(fwd, a, b, accumulate, phase_out=False):
    """Input gradient of a 3x3 stride-2 conv for input pixels (2y+a, 2x+b): a 2x2 conv on dZ (a transposed conv, 4 phases).
    phase_out=True: all four phases in one launch (a, b ignored; the output grid is phase (0, 0)'s, the largest; weights of the 4
    phases packed back to back; phase_sum == 3: dZ staged once, only the nine taps that exist)."""
    if phase_out:
        return ConvDesc(n=fwd.n, h_in=fwd.h_out, w_in=fwd.w_out, c1=fwd.c_out, c2=0, h_src1=fwd.h_out, w_src1=fwd.w_out,
                        gather1=RCF_GATHER_DIRECT, h_out=(fwd.h_in + 1) // 2, w_out=(fwd.w_in + 1) // 2, c_out=fwd.c1, ksize=2, stride=1, pad=0, pad_x=0,
                        w_mode=RCF_W_FORWARD, w_o=fwd.c1, w_i=fwd.c_out, w_i_off=0, accumulate=1 if accumulate else 0,
                        out_stride=2, out_off_y=0, out_off_x=0, out_h_phys=fwd.h_in, out_w_phys=fwd.w_in, in_off_y=0, in_off_x=0, phase_sum=3,
                        precision=_PRECISION[0], storage=_STORAGE[0])
    hy = (fwd.h_in - a + 1) // 2
    wx = (fwd.w_in - b + 1) // 2
    return ConvDesc(n=fwd.n, h_in=fwd.h_out, w_in=fwd.w_out, c1=fwd.c_out, c2=0, h_src1=fwd.h_out, w_src1=fwd.w_out,
                    gather1=RCF_GATHER_DIRECT, h_out=hy, w_out=wx, c_out=fwd.c1, ksize=2, stride=1, pad=0, pad_x=0,
                    w_mode=RCF_W_FORWARD, w_o=fwd.c1, w_i=fwd.c_out, w_i_off=0, accumulate=1 if accumulate else 0,
                    out_stride=2, out_off_y=a, out_off_x=b, out_h_phys=fwd.h_in, out_w_phys=fwd.w_in, in_off_y=0, in_off_x=0, phase_sum=0, precision=_PRECISION[0], storage=_STORAGE[0])


def make_s2_wgrad_desc(fwd, a, b, all_phases=False):
    """Phase (a,b) of the weight gradient of the 3x3 stride-2 conv `fwd`: a 2x2 weight gradient on the phase image
    x[2y+a, 2x+b] of its input against the same dZ (rcf_phase_wgrad_gather_s2 picks the nine real taps).
    all_phases=True: the four phases in one conv_wgrad launch (a, b ignored; phase_sum == 1; dw = [4][co][ci][2][2])."""
    if all_phases:
        d = make_s2_wgrad_desc(fwd, 0, 0)
        d.phase_sum = 1
        return d
    return ConvDesc(n=fwd.n, h_in=fwd.h_out, w_in=fwd.w_out, c1=fwd.c1, c2=0, h_src1=fwd.h_in, w_src1=fwd.w_in,
                    gather1=RCF_GATHER_STRIDED2, h_out=fwd.h_out, w_out=fwd.w_out, c_out=fwd.c_out, ksize=2, stride=1, pad=1, pad_x=1,
                    w_mode=RCF_W_FORWARD, w_o=fwd.c_out, w_i=fwd.c1, w_i_off=0, accumulate=0,
                    out_stride=1, out_off_y=0, out_off_x=0, out_h_phys=fwd.h_out, out_w_phys=fwd.w_out, in_off_y=a, in_off_x=b,
                    phase_sum=0, precision=_PRECISION[0], storage=_STORAGE[0])


def phase_wgrad_gather_s2(dwp, dw_oihw):
    o, i = dw_oihw.shape[0], dw_oihw.shape[1]
    check(_lib.load().rcf_phase_wgrad_gather_s2(_f32(dwp), _f32(dw_oihw), o, i, _stream()), 'rcf_phase_wgrad_gather_s2')


def phase_weights(w_oihw, mode):
    """[4][O'][I'][2][2] phase weights of a 3x3 OIHW weight (RCF_PHASE_*)."""
    o, i = w_oihw.shape[0], w_oihw.shape[1]
    out = torch.empty((4, o, i, 2, 2) if mode == RCF_PHASE_UP2X_FWD else (4, i, o, 2, 2), dtype=torch.float32, device=w_oihw.device)
    check(_lib.load().rcf_phase_weights(_f32(w_oihw), _f32(out), o, i, mode, _stream()), 'rcf_phase_weights')
    return out


def phase_wgrad_fold(dwp, dw_oihw):
    o, i = dw_oihw.shape[0], dw_oihw.shape[1]
    check(_lib.load().rcf_phase_wgrad_fold(_f32(dwp), _f32(dw_oihw), o, i, _stream()), 'rcf_phase_wgrad_fold')


def algorithmic_flops(desc):
    """2*MACs of the convolution a descriptor stands for, counted on REAL channels and on the forward conv's
    output grid (a stride-2 input gradient counts the forward conv's MACs, not the zero-dilated ones)."""
    k2 = desc.ksize * desc.ksize
    if desc.ksize == 2:     # a phase conv: count what it executes (its 4 phases together do 4/9 resp. 16/36 of the 3x3 MACs)
        # (phase_sum == 3, the merged stride-2 input gradient: the nine taps that exist, on phase (0, 0)'s grid)
        return 2.0 * desc.n * desc.h_out * desc.w_out * desc.c_out * desc.c1 * (9 if desc.phase_sum == 3 else k2 * (4 if desc.phase_sum else 1))
    if desc.w_mode == RCF_W_DGRAD:
        return 2.0 * desc.n * desc.h_src1 * desc.w_src1 * desc.c1 * k2 * desc.c_out
    return 2.0 * desc.n * desc.h_out * desc.w_out * desc.c_out * k2 * (desc.c1 + desc.c2)


def algorithmic_bytes(desc):
    """HBM bytes a conv launch must move at minimum: every input element read once, every output written once
    (read-modify-write when it accumulates); weights are negligible."""
    n = desc.n
    if desc.gather1 == RCF_GATHER_STRIDED2 and not desc.phase_sum:
        in1 = desc.h_in * desc.w_in * desc.c1            # one phase of the source
    else:
        in1 = desc.h_src1 * desc.w_src1 * desc.c1
    in2 = desc.h_in * desc.w_in * desc.c2
    out = desc.h_out * desc.w_out * desc.c_out * (4 if desc.phase_sum >= 2 else 1)    # (the four output phases of an up-2x forward)
    return (2.0 if desc.storage else 4.0) * n * (in1 + in2 + out * (2 if desc.accumulate else 1))


class KernelTimer(object):
    """Brackets kernel launches with events on the stream they are launched on (torch's current stream is the
    stream handed to the C ABI) and accumulates (launches, algorithmic flops, ms) per kernel id."""

    def __init__(self, only_ids=None):
        self.only = None if only_ids is None else set(only_ids)
        self.pending = []
        self.cur = None
        self.scope = None   # set by the engine: 'encoder' / 'decoder' (whose layer the bracketed launch belongs to)

    def begin(self, kid, flops, desc=None):
        if self.only is not None and kid not in self.only:
            self.cur = None
            return
        ev = torch.cuda.Event(enable_timing=True)
        ev.record()
        tag = None
        self.bytes = getattr(self, 'bytes', {})
        if desc is not None:
            self.bytes[kid] = self.bytes.get(kid, 0.0) + (algorithmic_bytes(desc) if (kid % 20000) < 10000 else 0.0)
            tag = '%s %s k%d s%d %d+%d->%d @%dx%d g%d' % (self.scope or '-', 'dgrad' if desc.w_mode == RCF_W_DGRAD else ('wgrad' if (kid % 20000) >= 10000 else 'fwd'),
                                                         desc.ksize, desc.stride, desc.c1, desc.c2, desc.c_out, desc.h_out, desc.w_out,
                                                         desc.gather1)
        self.cur = (kid, flops, ev, tag)

    def end(self):
        if self.cur is None:
            return
        ev = torch.cuda.Event(enable_timing=True)
        ev.record()
        self.pending.append(self.cur + (ev,))
        self.cur = None

    def collect(self):
        """Call after a synchronize.  Returns {kernel id: [launches, flops, ms]}."""
        out = {}
        self.layers = {}
        for kid, flops, e0, tag, e1 in self.pending:
            ms = e0.elapsed_time(e1)
            r = out.setdefault(kid, [0, 0.0, 0.0])
            r[0] += 1
            r[1] += flops
            r[2] += ms
            if tag is not None:
                q = self.layers.setdefault((kid, tag), [0, 0.0, 0.0])
                q[0] += 1
                q[1] += flops
                q[2] += ms
        self.pending = []
        return out


def conv_query(desc):
    info = ConvInfo()
    check(_lib.load().rcf_conv2d_query(ctypes.byref(desc), ctypes.byref(info)), 'rcf_conv2d_query')
    return info


def conv_pack(desc, w_oihw, packed, amax_w=None):
    """amax_w (RCF_PREC_F16X2 descriptors): device scalar holding max|w| -- the fp16 planes hold w * its power-of-two scale."""
    if amax_w is not None:
        check(_lib.load().rcf_conv2d_pack_weights_scaled(ctypes.byref(desc), _f32(w_oihw), _f32(packed), _f32(amax_w), _stream()),
              'rcf_conv2d_pack_weights_scaled')
        return
    check(_lib.load().rcf_conv2d_pack_weights(ctypes.byref(desc), _f32(w_oihw), _f32(packed), _stream()),
          'rcf_conv2d_pack_weights')


def make_scales(amax_in1=None, amax_in2=None, amax_w=None, amax_dz=None):
    """rcf_conv_scales from device scalars (None = unscaled operand)."""
    return _lib.ConvScales(_f32(amax_in1), _f32(amax_in2), _f32(amax_w), _f32(amax_dz))


def amax(x, out=None, accumulate=False):
    """max|x| over an fp32 tensor as a device scalar (rcf_amax); out given: written there (after zeroing unless accumulate)."""
    if out is None:
        out = torch.zeros(1, dtype=torch.float32, device=x.device)
    elif not accumulate:
        out.zero_()
    check(_lib.load().rcf_amax(_f32(x), x.numel(), _f32(out), _stream()), 'rcf_amax')
    return out


def amax_batch(items, n):
    """items: ctypes array of _lib.AmaxItem; the output slots must have been zeroed (rcf_amax_batch accumulates)."""
    check(_lib.load().rcf_amax_batch(items, n, _stream()), 'rcf_amax_batch')


def conv_pack_batch(items, n):
    """items: ctypes array of _lib.PackItem (built once, reused every step); rcf_conv2d_pack_weights_batch."""
    check(_lib.load().rcf_conv2d_pack_weights_batch(items, n, _stream()), 'rcf_conv2d_pack_weights_batch')


def phase_weights_batch(items, n):
    check(_lib.load().rcf_phase_weights_batch(items, n, _stream()), 'rcf_phase_weights_batch')


def conv_fwd(desc, in1, in2, packed, out, stat_partials=None, coef1=None, coef2=None, scales=None):
    """coef1 / coef2: in1 / in2 are raw conv outputs whose BatchNorm + lrelu is applied on load (rcf_conv_info.bn_on_load).
    scales (make_scales; RCF_PREC_F16X2 descriptors on the split kernels): the operands' per-tensor maxima."""
    # the 7x7 stems read the fp32 network input whatever the storage is (include/rcf_hip.h, rcf_conv_desc.storage)
    _check_storage(desc, None if desc.ksize == 7 else in1, in2, out)
    if scales is not None:
        check(_lib.load().rcf_conv2d_fwd_scaled(ctypes.byref(desc), _a(in1), _a(in2), _f32(packed), _a(out), _f64(stat_partials),
                                                ctypes.byref(scales), _stream()), 'rcf_conv2d_fwd_scaled')
    elif coef1 is None and coef2 is None:
        check(_lib.load().rcf_conv2d_fwd(ctypes.byref(desc), _a(in1), _a(in2), _f32(packed), _a(out),
                                         _f64(stat_partials), _stream()), 'rcf_conv2d_fwd')
    else:
        check(_lib.load().rcf_conv2d_fwd_bn(ctypes.byref(desc), _a(in1), _f32(coef1), _a(in2), _f32(coef2), _f32(packed),
                                            _a(out), _f64(stat_partials), _stream()), 'rcf_conv2d_fwd_bn')


def conv_dgrad_bn_sums(desc, dz, packed, dx, bn_z, bn_coef, sum_partials, scales):
    """Input gradient dx that is dY of a BatchNorm + LeakyReLU block (raw output bn_z, coefficients bn_coef) and its only writer: also
    leaves that block's backward sums (sum g, sum g * xhat) in sum_partials [n_partials][2][c]; rcf_conv_info.bn_bwd_sums.
    scales: make_scales(...) for two-plane fp16 descriptors, None for bf16 tensors."""
    _check_storage(desc, dz, dx, bn_z)
    check(_lib.load().rcf_conv2d_dgrad_bn_sums(ctypes.byref(desc), _a(dz), _f32(packed), _a(dx), _a(bn_z), _f32(bn_coef),
                                               _f64(sum_partials), None if scales is None else ctypes.byref(scales), _stream()),
          'rcf_conv2d_dgrad_bn_sums')


def conv_fwd_act(desc, in1, in2, packed, bias, res, out):
    """Inference epilogue in the matrix kernel: out = lrelu(conv + bias) (then lrelu(. + res)); rcf_conv_info.fwd_act."""
    _check_storage(desc, in1, in2, res, out)
    check(_lib.load().rcf_conv2d_fwd_act(ctypes.byref(desc), _a(in1), _a(in2), _f32(packed), _f32(bias), _a(res), _a(out),
                                         _stream()), 'rcf_conv2d_fwd_act')


def scale_channels(w_oihw, scale):
    """w[o] * scale[o]: BatchNorm scale folded into a conv weight (inference)."""
    w = w_oihw.contiguous()
    out = torch.empty_like(w)
    check(_lib.load().rcf_scale_channels(_f32(w), _f32(scale), _f32(out), w.shape[0], w.numel() // w.shape[0], _stream()),
          'rcf_scale_channels')
    return out


def conv_wgrad(desc, in1, in2, dz, dw, workspace, coef1=None, coef2=None, scales=None):
    _check_storage(desc, None if desc.ksize == 7 else in1, in2, dz)
    if scales is not None:
        check(_lib.load().rcf_conv2d_wgrad_scaled(ctypes.byref(desc), _a(in1), _a(in2), _a(dz), _f32(dw), _f32(workspace),
                                                  ctypes.byref(scales), _stream()), 'rcf_conv2d_wgrad_scaled')
    elif coef1 is None and coef2 is None:
        check(_lib.load().rcf_conv2d_wgrad(ctypes.byref(desc), _a(in1), _a(in2), _a(dz), _f32(dw), _f32(workspace),
                                           _stream()), 'rcf_conv2d_wgrad')
    else:
        check(_lib.load().rcf_conv2d_wgrad_bn(ctypes.byref(desc), _a(in1), _f32(coef1), _a(in2), _f32(coef2), _a(dz),
                                              _f32(dw), _f32(workspace), _stream()), 'rcf_conv2d_wgrad_bn')


def bn_finalize(partials, n_partials, c, count, gamma, beta, running_mean, running_var, momentum, eps, training, coef):
    check(_lib.load().rcf_bn_finalize(_f64(partials), n_partials, c, float(count), _f32(gamma), _f32(beta),
                                      _f32(running_mean), _f32(running_var), momentum, eps, 1 if training else 0,
                                      _f32(coef), _stream()), 'rcf_bn_finalize')


def bn_act_fwd(z, coef, res, out, n_pix, c, act, amax=None):
    """amax (fp32 tensors): zeroed device scalar that also receives max|out| (rcf_bn_act_fwd_amax)."""
    if amax is not None:
        check(_lib.load().rcf_bn_act_fwd_amax(_f32(z), _f32(coef), _f32(res), _f32(out), n_pix, c, act, _f32(amax), _stream()),
              'rcf_bn_act_fwd_amax')
        return
    check(_fn('rcf_bn_act_fwd', z, res, out)(_a(z), _f32(coef), _a(res), _a(out), n_pix, c, act, _stream()), 'rcf_bn_act_fwd')


def fuse_fwd(zw, coef_w, zp, coef_p, img, out, n_pix, c, amax=None):
    if amax is not None:
        check(_lib.load().rcf_fuse_fwd_amax(_f32(zw), _f32(coef_w), _f32(zp), _f32(coef_p), _f32(img), _f32(out), n_pix, c,
                                            _f32(amax), _stream()), 'rcf_fuse_fwd_amax')
        return
    check(_fn('rcf_fuse_fwd', zw, zp, img, out)(_a(zw), _f32(coef_w), _a(zp), _f32(coef_p), _a(img), _a(out), n_pix, c,
                                                _stream()), 'rcf_fuse_fwd')


def fuse_wp_infer_supported(c_d, c_i):
    return bool(_lib.load().rcf_fuse_wp_infer_supported(int(c_d), int(c_i)))


def fuse_wp_infer(d, w1, coef_w, w2, coef_p, img, out):
    """Inference 'weight_and_project' fusion in one pass (bf16 NHWC tensors): sigmoid(BN_w(W1 d)) * BN_p(W2 d) + img, eval-mode BN."""
    if d.dtype != torch.bfloat16 or img.dtype != torch.bfloat16 or out.dtype != torch.bfloat16:
        raise ValueError('fuse_wp_infer takes bf16 activation tensors')
    c_d, c_i = d.shape[-1], img.shape[-1]
    if tuple(w1.shape[:2]) != (c_i, c_d) or tuple(w2.shape[:2]) != (c_i, c_d) or tuple(out.shape) != tuple(img.shape):
        raise ValueError('fuse_wp_infer: shapes disagree')
    n_pix = img.numel() // c_i
    check(_lib.load().rcf_fuse_wp_infer_b16(_a(d), _f32(w1), _f32(coef_w), _f32(w2), _f32(coef_p), _a(img), _a(out), n_pix, c_d, c_i,
                                            _stream()), 'rcf_fuse_wp_infer_b16')


def ew_blocks(n_pix, c):
    nb = _lib.load().rcf_ew_blocks(n_pix, c)
    if nb <= 0:
        raise _lib.RcfError('rcf_ew_blocks: unsupported channel count %d' % c)
    return nb


def bn_act_bwd_reduce(dout, z, coef, out, partials, n_pix, c, act, has_res):
    check(_fn('rcf_bn_act_bwd_reduce', dout, z, out)(_a(dout), _a(z), _f32(coef), _a(out), _f64(partials), n_pix, c, act,
                                                     1 if has_res else 0, _stream()), 'rcf_bn_act_bwd_reduce')


def bn_bwd_finalize(partials, n_blocks, stride, c, count, bcoef, dgamma, dbeta):
    check(_lib.load().rcf_bn_bwd_finalize(_f64(partials), n_blocks, stride, c, float(count), _f32(bcoef), _f32(dgamma),
                                          _f32(dbeta), _stream()), 'rcf_bn_bwd_finalize')


def head_bn_blocks(n, h, w, c):
    """Partial rows of head_bn_bwd_reduce, or <= 0 when the fused path does not cover the shape."""
    return _lib.load().rcf_head_bn_blocks(n, h, w, c)


def head_bn_bwd_reduce(dlogit, w_head, z, coef, partials):
    n, h, w, c = z.shape
    check(_fn('rcf_head_bn_bwd_reduce', z)(_f32(dlogit), _f32(w_head), _a(z), _f32(coef), _f64(partials), n, h, w, c, _stream()),
          'rcf_head_bn_bwd_reduce')


def head_bn_bwd_apply(dlogit, w_head, z, coef, bcoef, dz, amax=None):
    n, h, w, c = z.shape
    if amax is not None:
        check(_lib.load().rcf_head_bn_bwd_apply_amax(_f32(dlogit), _f32(w_head), _f32(z), _f32(coef), _f32(bcoef), _f32(dz), n, h, w, c,
                                                     _f32(amax), _stream()), 'rcf_head_bn_bwd_apply_amax')
        return
    check(_fn('rcf_head_bn_bwd_apply', z, dz)(_f32(dlogit), _f32(w_head), _a(z), _f32(coef), _f32(bcoef), _a(dz), n, h, w, c,
                                              _stream()), 'rcf_head_bn_bwd_apply')


def bn_act_bwd_apply(dout, z, coef, out, bcoef, dz, dres, dres_accumulate, n_pix, c, act, has_res, amax=None):
    if amax is not None:
        check(_lib.load().rcf_bn_act_bwd_apply_amax(_f32(dout), _f32(z), _f32(coef), _f32(out), _f32(bcoef), _f32(dz), _f32(dres),
                                                    1 if dres_accumulate else 0, n_pix, c, act, 1 if has_res else 0, _f32(amax),
                                                    _stream()), 'rcf_bn_act_bwd_apply_amax')
        return
    check(_fn('rcf_bn_act_bwd_apply', dout, z, out, dz, dres)(_a(dout), _a(z), _f32(coef), _a(out), _f32(bcoef), _a(dz), _a(dres),
                                                              1 if dres_accumulate else 0, n_pix, c, act, 1 if has_res else 0, _stream()),
          'rcf_bn_act_bwd_apply')


def fuse_bwd_reduce(dout, zw, coef_w, zp, coef_p, partials, n_pix, c):
    check(_fn('rcf_fuse_bwd_reduce', dout, zw, zp)(_a(dout), _a(zw), _f32(coef_w), _a(zp), _f32(coef_p), _f64(partials),
                                                   n_pix, c, _stream()), 'rcf_fuse_bwd_reduce')


def fuse_bwd_apply(dout, zw, coef_w, zp, coef_p, bcoef_w, bcoef_p, dzw, dzp, dimg, dimg_accumulate, n_pix, c):
    check(_fn('rcf_fuse_bwd_apply', dout, zw, zp, dzw, dzp, dimg)(_a(dout), _a(zw), _f32(coef_w), _a(zp), _f32(coef_p), _f32(bcoef_w),
                                                                  _f32(bcoef_p), _a(dzw), _a(dzp), _a(dimg), 1 if dimg_accumulate else 0,
                                                                  n_pix, c, _stream()), 'rcf_fuse_bwd_apply')


def maxpool_fwd(x, out, idx):
    n, h, w, c = x.shape
    check(_fn('rcf_maxpool3x3s2_fwd', x, out)(_a(x), _a(out), _p(idx), n, h, w, c, _stream()), 'rcf_maxpool3x3s2_fwd')


def maxpool_bwd(dout, idx, din, accumulate):
    n, h, w, c = din.shape
    check(_fn('rcf_maxpool3x3s2_bwd', dout, din)(_a(dout), _p(idx), _a(din), 1 if accumulate else 0, n, h, w, c, _stream()),
          'rcf_maxpool3x3s2_bwd')


def upsample_nearest_bwd(dup, dsrc, accumulate):
    n, hu, wu, c = dup.shape
    _, hs, ws, _ = dsrc.shape
    check(_fn('rcf_upsample_nearest_bwd', dup, dsrc)(_a(dup), _a(dsrc), 1 if accumulate else 0, n, hu, wu, hs, ws, c,
                                                     _stream()), 'rcf_upsample_nearest_bwd')


def head_fwd(x, w, logit, depth, dmin, dmax, coef=None):
    """coef given: x is the previous block's raw conv output z; its BatchNorm + lrelu is applied on load."""
    n, h, ww, c = x.shape
    if coef is None:
        check(_fn('rcf_head_fwd', x)(_a(x), _f32(w), _f32(logit), _f32(depth), n, h, ww, c, dmin, dmax, _stream()),
              'rcf_head_fwd')
    else:
        check(_fn('rcf_head_fwd_bn', x)(_a(x), _f32(coef), _f32(w), _f32(logit), _f32(depth), n, h, ww, c, dmin, dmax,
                                        _stream()), 'rcf_head_fwd_bn')


def head_bwd_logit(ddepth, logit, dlogit, dmin, dmax):
    check(_lib.load().rcf_head_bwd_logit(_f32(ddepth), _f32(logit), _f32(dlogit), logit.numel(), dmin, dmax, _stream()),
          'rcf_head_bwd_logit')


def head_bwd_dgrad(dlogit, w, dx):
    n, h, ww, c = dx.shape
    check(_fn('rcf_head_bwd_dgrad', dx)(_f32(dlogit), _f32(w), _a(dx), n, h, ww, c, _stream()), 'rcf_head_bwd_dgrad')


def head_bwd_wgrad(x, dlogit, dw, coef=None):
    n, h, ww, c = x.shape
    nws = _lib.load().rcf_head_wgrad_workspace_floats(n, h, ww, c)
    ws = torch.empty(nws, dtype=torch.float32, device=x.device)
    if coef is None:
        check(_fn('rcf_head_bwd_wgrad', x)(_a(x), _f32(dlogit), _f32(dw), _f32(ws), n, h, ww, c, _stream()),
              'rcf_head_bwd_wgrad')
    else:
        check(_fn('rcf_head_bwd_wgrad_bn', x)(_a(x), _f32(coef), _f32(dlogit), _f32(dw), _f32(ws), n, h, ww, c, _stream()),
              'rcf_head_bwd_wgrad_bn')


LOSS_KINDS = {'l1': 0, 'l2': 1, 'smoothl1': 2}   # RCF_LOSS_* (include/rcf_hip.h): FusionNetModel.compute_loss's loss_func values


def l1_loss_fwd(depth, gt, lidar, sums, kind='l1'):
    """sums[4] (fp64) = (sum of the per-pixel term over gt > 0, count, the same over lidar > 0, count) of the masked loss `kind`."""
    n = depth.numel()
    ws = torch.empty(_lib.load().rcf_loss_workspace_floats(n), dtype=torch.float32, device=depth.device)
    if sums.dtype != torch.float64:
        raise _lib.RcfError('loss sums must be float64[4]')
    check(_lib.load().rcf_masked_loss_fwd(_f32(depth), _f32(gt), _f32(lidar), _f32(ws), _p(sums), n, LOSS_KINDS[kind], _stream()),
          'rcf_masked_loss_fwd')


def smoothness_loss_fwd(image_nchw, depth, sums):
    """sums[4] (fp64) = (sum_x, count_x, sum_y, count_y) of the local smoothness term (rcf_smoothness_loss_fwd)."""
    n, c, h, w = image_nchw.shape
    ws = torch.empty(_lib.load().rcf_loss_workspace_floats(depth.numel()), dtype=torch.float32, device=depth.device)
    check(_lib.load().rcf_smoothness_loss_fwd(_f32(image_nchw), _f32(depth), _f32(ws), _f64(sums), n, c, h, w, _stream()), 'rcf_smoothness_loss_fwd')


def smoothness_loss_bwd(image_nchw, depth, sums, upstream, w_smoothness, ddepth):
    """ddepth += upstream * w_smoothness * d(smoothness)/d(depth)."""
    n, c, h, w = image_nchw.shape
    check(_lib.load().rcf_smoothness_loss_bwd(_f32(image_nchw), _f32(depth), _f64(sums), _f32(upstream), float(w_smoothness), _f32(ddepth),
                                              n, c, h, w, _stream()), 'rcf_smoothness_loss_bwd')


def l1_loss_value(sums, w_lidar, loss):
    check(_lib.load().rcf_l1_loss_value(_p(sums), w_lidar, _f32(loss), _stream()), 'rcf_l1_loss_value')


def l1_loss_bwd(depth, gt, lidar, sums, upstream, w_lidar, ddepth, kind='l1'):
    check(_lib.load().rcf_masked_loss_bwd(_f32(depth), _f32(gt), _f32(lidar), _p(sums), _f32(upstream), w_lidar, _f32(ddepth),
                                          depth.numel(), LOSS_KINDS[kind], _stream()), 'rcf_masked_loss_bwd')


def outlier_removal(depth, kernel_size=7, threshold=1.5):
    """OutlierRemoval.remove_outliers (src/net_utils.py:591-638) on an N x 1 x H x W (or N x H x W) sparse depth map."""
    shape = depth.shape
    n, h, w = shape[0], shape[-2], shape[-1]
    d = depth.contiguous()
    out = torch.empty_like(d)
    scratch = torch.empty(1, dtype=torch.float32, device=d.device)
    check(_lib.load().rcf_outlier_removal(_f32(d), _f32(out), _f32(scratch), n, h, w, kernel_size, threshold, _stream()),
          'rcf_outlier_removal')
    return out


def adam_step(p, g, m, v, lr, beta1, beta2, eps, weight_decay, step):
    check(_lib.load().rcf_adam_step(_f32(p), _f32(g), _f32(m), _f32(v), p.numel(), lr, beta1, beta2, eps, weight_decay,
                                    step, _stream()), 'rcf_adam_step')


def adam_step_dev(p, g, m, v, state):
    """Adam with step count and hyper-parameters in the device buffer `state` (float32[8]): capturable in a hipGraph."""
    check(_lib.load().rcf_adam_step_dev(_f32(p), _f32(g), _f32(m), _f32(v), p.numel(), _f32(state), _stream()), 'rcf_adam_step_dev')


def nchw_to_nhwc(x):
    n, c, h, w = x.shape
    out = torch.empty((n, h, w, c), dtype=torch.float32, device=x.device)
    check(_lib.load().rcf_nchw_to_nhwc(_f32(x), _f32(out), n, c, h, w, _stream()), 'rcf_nchw_to_nhwc')
    return out


def nhwc_to_nchw(x):
    n, h, w, c = x.shape
    out = torch.empty((n, c, h, w), dtype=torch.float32, device=x.device)
    check(_lib.load().rcf_nhwc_to_nchw(_f32(x), _f32(out), n, c, h, w, _stream()), 'rcf_nhwc_to_nchw')
    return out


def radar_scatter(crops, points, width, strict_reference=True, logits=False):
    """crops: sigmoid responses (K, H, Wc); logits=True: the correspondence logits instead (threshold on their sign, rcf_radar_scatter_logits)."""
    k, h, wc = crops.shape
    depth = torch.empty((h, width), dtype=torch.float32, device=crops.device)
    resp = torch.empty((h, width), dtype=torch.float32, device=crops.device)
    fn = _lib.load().rcf_radar_scatter_logits if logits else _lib.load().rcf_radar_scatter
    check(fn(_f32(crops), _f32(points), k, h, width, wc, 1 if strict_reference else 0, _f32(depth), _f32(resp), _stream()),
          'rcf_radar_scatter_logits' if logits else 'rcf_radar_scatter')
    return depth, resp


# ---------------------------------------------------------------- RadarNet stage 1
def roi_pool_fwd(x, rois, out, argmax, pooled_hw, spatial_scale, out_coff=0):
    """x (N,H,W,C); rois (R,5); out (R,PH,PW,Ctot) written at channel offset out_coff; argmax (R,PH,PW,C) int32."""
    n, h, w, c = x.shape
    if argmax.dtype != torch.int32:
        raise _lib.RcfError('roi_pool argmax must be int32')
    check(_fn('rcf_roi_pool_fwd', x, out)(_a(x), _f32(rois), _a(out), _p(argmax), rois.shape[0], n, h, w, c, pooled_hw[0],
                                          pooled_hw[1], float(spatial_scale), out.shape[-1], out_coff, _stream()), 'rcf_roi_pool_fwd')


def roi_pool_bwd(dout, argmax, rois, din, pooled_hw, dout_coff=0):
    n, h, w, c = din.shape
    # din is fp32 for either storage of dout (the scatter uses fp32 atomics)
    check(_fn('rcf_roi_pool_bwd', dout)(_a(dout), _p(argmax), _f32(rois), _f32(din), rois.shape[0], n, h, w, c, pooled_hw[0],
                                        pooled_hw[1], dout.shape[-1], dout_coff, _stream()), 'rcf_roi_pool_bwd')


def roi_pool_bwd_gather(dout, argmax, rois, din, accumulate, pooled_hw, scale, dout_coff=0):
    """din (same storage as dout) = / += the pooled gradient, gathered per input pixel: no atomics, no zero fill."""
    n, h, w, c = din.shape
    check(_fn('rcf_roi_pool_bwd_gather', dout, din)(_a(dout), _p(argmax), _f32(rois), _a(din), 1 if accumulate else 0, rois.shape[0], n, h, w,
                                                    c, pooled_hw[0], pooled_hw[1], float(scale), dout.shape[-1], dout_coff, _stream()),
          'rcf_roi_pool_bwd_gather')


def fc_fwd(x, w, b, y, act, hw=1, cstride=0, coff=0):
    m, n_in = x.shape
    check(_fn('rcf_fc_fwd', y)(_f32(x), _f32(w), _f32(b), _a(y), m, n_in, w.shape[0], 1 if act else 0, hw, cstride, coff,
                               _stream()), 'rcf_fc_fwd')


def fc_bwd(x, w, y, dy, dw, db, dx, act, hw=1, cstride=0, coff=0):
    m, n_in = x.shape
    ws = None
    if dx is not None:
        ws = torch.empty(_lib.load().rcf_fc_bwd_workspace_floats(m, n_in, w.shape[0]), dtype=torch.float32, device=x.device)
    check(_fn('rcf_fc_bwd', y, dy)(_f32(x), _f32(w), _a(y), _a(dy), _f32(dw), _f32(db), _f32(dx), _f32(ws), m, n_in, w.shape[0],
                                   1 if act else 0, hw, cstride, coff, _stream()), 'rcf_fc_bwd')


def bce_loss_fwd(logit, target, valid, sums, loss, pos_weight):
    ws = torch.empty(_lib.load().rcf_bce_workspace_doubles(), dtype=torch.float64, device=logit.device)
    check(_lib.load().rcf_bce_loss_fwd(_f32(logit), _f32(target), _f32(valid), _p(ws), _p(sums), _f32(loss), logit.numel(),
                                       float(pos_weight), _stream()), 'rcf_bce_loss_fwd')


def bce_loss_bwd(logit, target, valid, sums, upstream, dlogit, pos_weight):
    check(_lib.load().rcf_bce_loss_bwd(_f32(logit), _f32(target), _f32(valid), _p(sums), _f32(upstream), _f32(dlogit),
                                       logit.numel(), float(pos_weight), _stream()), 'rcf_bce_loss_bwd')


# ---------------------------------------------------------------- input augmentation
def transform_images(images, do_b, f_b, do_c, f_c, do_s, f_s, do_hf, do_vf, norm_mode):
    """images (N,3,H,W) fp32; do_* uint8[N] or None; f_* float32[N]; returns the transformed float images."""
    n, c, h, w = images.shape
    if c != 3:
        raise _lib.RcfError('transform_images expects RGB images')
    x = images.contiguous()
    out = torch.empty_like(x)
    ws = torch.empty(_lib.load().rcf_transform_workspace_bytes(n), dtype=torch.uint8, device=x.device)
    for t in (do_b, do_c, do_s, do_hf, do_vf):
        if t is not None and t.dtype != torch.uint8:
            raise _lib.RcfError('transform decisions must be uint8')
    check(_lib.load().rcf_transform_images(_f32(x), _f32(out), n, h, w, _p(do_b), _f32(f_b), _p(do_c), _f32(f_c), _p(do_s), _f32(f_s),
                                           _p(do_hf), _p(do_vf), norm_mode, _p(ws), _stream()), 'rcf_transform_images')
    return out


def transform_flip(maps, do_hf, do_vf):
    n, c, h, w = maps.shape
    x = maps.contiguous()
    out = torch.empty_like(x)
    check(_lib.load().rcf_transform_flip(_f32(x), _f32(out), n, c, h, w, _p(do_hf), _p(do_vf), _stream()), 'rcf_transform_flip')
    return out


# ---------------------------------------------------------------- on-disk formats finished on the device
def _crop_arg(crop_yx, n, src_h, src_w, h, w, device):
    """crop offsets (n, 2) int32 on the device, bounds-checked on the host BEFORE they are uploaded (numpy / list / CPU tensor)."""
    if crop_yx is None:
        if (h, w) != (src_h, src_w):
            raise _lib.RcfError('a crop needs its (y0, x0) offsets')
        return None
    c = torch.as_tensor(crop_yx, dtype=torch.int32)
    if c.is_cuda:
        raise _lib.RcfError('crop offsets are drawn on the host (datasets.random_crop): pass a list / numpy / CPU tensor')
    if tuple(c.shape) != (n, 2):
        raise _lib.RcfError('crop offsets must have shape (%d, 2)' % n)
    if n and (int(c.min()) < 0 or int(c[:, 0].max()) + h > src_h or int(c[:, 1].max()) + w > src_w):
        raise _lib.RcfError('crop window leaves the source image')
    return c.contiguous().to(device, non_blocking=True)


def decode_images(raw, crop_yx=None, shape=None, normalize=False):
    """raw (N, H, W, 3) uint8 on the device -> (N, 3, h, w) float32: data_utils.load_image(..., data_format='CHW') + random_crop."""
    if raw.dtype != torch.uint8 or raw.dim() != 4 or raw.shape[3] != 3:
        raise _lib.RcfError('decode_images expects (N, H, W, 3) uint8')
    n, src_h, src_w = int(raw.shape[0]), int(raw.shape[1]), int(raw.shape[2])
    h, w = (src_h, src_w) if shape is None else (int(shape[0]), int(shape[1]))
    crop = _crop_arg(crop_yx, n, src_h, src_w, h, w, raw.device)
    out = torch.empty((n, 3, h, w), dtype=torch.float32, device=raw.device)
    check(_lib.load().rcf_decode_image_u8(_p(raw), _f32(out), n, src_h, src_w, h, w, _p(crop), 1 if normalize else 0, _stream()),
          'rcf_decode_image_u8')
    return out


_PIXEL_TYPES = {torch.uint8: _lib.RCF_PIXEL_U8, torch.uint16: _lib.RCF_PIXEL_U16, torch.int32: _lib.RCF_PIXEL_I32}


def decode_maps(raw, multiplier=256.0, crop_yx=None, shape=None, clamp_nonpositive=True, with_validity=False):
    """raw (N, H, W) uint8 / uint16 / int32 on the device -> (N, 1, h, w) float32 (data_utils.load_depth / load_response + crop);
    with_validity also returns load_depth_with_validity_map's second output."""
    if raw.dtype not in _PIXEL_TYPES or raw.dim() != 3:
        raise _lib.RcfError('decode_maps expects (N, H, W) uint8 / uint16 / int32, got %s %s' % (raw.dtype, tuple(raw.shape)))
    n, src_h, src_w = int(raw.shape[0]), int(raw.shape[1]), int(raw.shape[2])
    h, w = (src_h, src_w) if shape is None else (int(shape[0]), int(shape[1]))
    crop = _crop_arg(crop_yx, n, src_h, src_w, h, w, raw.device)
    out = torch.empty((n, 1, h, w), dtype=torch.float32, device=raw.device)
    valid = torch.empty_like(out) if with_validity else None
    check(_lib.load().rcf_decode_map(_p(raw), _PIXEL_TYPES[raw.dtype], _f32(out), _f32(valid), n, src_h, src_w, h, w, _p(crop),
                                     float(multiplier), 1 if clamp_nonpositive else 0, _stream()), 'rcf_decode_map')
    return (out, valid) if with_validity else out


def encode_maps(z, multiplier=256.0):
    """np.uint32(z * multiplier) of data_utils.save_depth / save_response, as an int32 tensor holding the uint32 bit patterns."""
    x = z.contiguous()
    out = torch.empty(x.shape, dtype=torch.int32, device=x.device)
    check(_lib.load().rcf_encode_map_u32(_f32(x), _p(out), x.numel(), float(multiplier), _stream()), 'rcf_encode_map_u32')
    return out


def points_to_depth_map(points, depth, height, width):
    """points (2, N) = (x, y), depth (N,) on the device -> (H, W) float32 map, the last point of a pixel wins
    (setup/setup_dataset_nuscenes_with_denseGT.py:814-840).  Raises IndexError for points outside the image, like numpy."""
    if points.dim() != 2 or points.shape[0] != 2 or depth.dim() != 1 or depth.shape[0] != points.shape[1]:
        raise _lib.RcfError('points_to_depth_map expects points (2, N) and depth (N,)')
    # np.round of the reference in the points' own precision (half to even, like torch.round); integers are exact in float32
    pts = torch.round(points).to(torch.float32).contiguous()
    d = depth.to(torch.float32).contiguous()
    n = int(pts.shape[1])
    out = torch.empty((int(height), int(width)), dtype=torch.float32, device=pts.device)
    ws = torch.empty(_lib.load().rcf_points_to_depth_map_workspace_bytes(int(height), int(width)), dtype=torch.uint8, device=pts.device)
    check(_lib.load().rcf_points_to_depth_map(_f32(pts[0]), _f32(pts[1]), _f32(d), n, _f32(out), int(height), int(width), _p(ws),
                                              _stream()), 'rcf_points_to_depth_map')
    n_bad = int(ws[-4:].view(torch.int32).item())
    if n_bad:
        raise IndexError('%d point(s) fall outside the %d x %d image' % (n_bad, height, width))
    return out


def convert(src, dst, accumulate=False):
    """dst (+)= src between fp32 and bf16 tensors of the same element count (rcf_convert)."""
    kinds = {torch.float32: _lib.RCF_STORE_FP32, torch.bfloat16: _lib.RCF_STORE_BF16}
    if src.dtype not in kinds or dst.dtype not in kinds or src.numel() != dst.numel():
        raise _lib.RcfError('convert: fp32 / bf16 tensors of equal size expected')
    check(_lib.load().rcf_convert(_p(src), kinds[src.dtype], _p(dst), kinds[dst.dtype], src.numel(), 1 if accumulate else 0, _stream()),
          'rcf_convert')
    return dst


# ---------------------------------------------------------------- the stems on the space-to-depth image (bf16 tensors)
def s2d_image(image_nchw):
    """(N, C <= 4, H, W) fp32 -> (N, ceil(H/2), ceil(W/2), 16) bf16: channel a*8 + b*4 + c = pixel (2y + a, 2x + b), channel c."""
    n, c, h, w = image_nchw.shape
    out = torch.empty((n, (h + 1) // 2, (w + 1) // 2, 16), dtype=torch.bfloat16, device=image_nchw.device)
    check(_lib.load().rcf_s2d_image_b16(_f32(image_nchw), _p(out), n, c, h, w, _stream()), 'rcf_s2d_image_b16')
    return out


def s2d_image_f32(image_nchw):
    """(N, C <= 4, H, W) fp32 -> ((N, ceil(H/2), ceil(W/2), 16) fp32, device scalar max|pixel|): the stems' input of the fp32
    configuration on the two-plane fp16 arithmetic."""
    n, c, h, w = image_nchw.shape
    out = torch.empty((n, (h + 1) // 2, (w + 1) // 2, 16), dtype=torch.float32, device=image_nchw.device)
    amax = torch.zeros(1, dtype=torch.float32, device=image_nchw.device)
    check(_lib.load().rcf_s2d_image_f32(_f32(image_nchw), _f32(out), n, c, h, w, _f32(amax), _stream()), 'rcf_s2d_image_f32')
    return out, amax


def stem_weights_s2d(w7):
    """OIHW 7x7 weight (C <= 4 input channels) -> OIHW 4x4 weight over the 16 space-to-depth channels."""
    co, c = w7.shape[0], w7.shape[1]
    out = torch.empty((co, 16, 4, 4), dtype=torch.float32, device=w7.device)
    check(_lib.load().rcf_stem_weights_s2d(_f32(w7.contiguous()), _f32(out), co, c, _stream()), 'rcf_stem_weights_s2d')
    return out


def make_stem_s2d_desc(n, h, w, c_out, f32=False):
    """The 7x7 stride-2 pad-3 stem on an (h, w) image as a 4x4 stride-1 conv on its space-to-depth image (bf16 tensors, or with
    f32=True fp32 tensors on the two-plane fp16 arithmetic)."""
    hs, ws = (h + 1) // 2, (w + 1) // 2
    ho, wo = conv_out_hw(h, w, 7, 2, 3)
    return ConvDesc(n=n, h_in=hs, w_in=ws, c1=16, c2=0, h_src1=hs, w_src1=ws, gather1=RCF_GATHER_DIRECT, h_out=ho, w_out=wo,
                    c_out=c_out, ksize=4, stride=1, pad=2, pad_x=2, w_mode=RCF_W_FORWARD, w_o=c_out, w_i=16, w_i_off=0, accumulate=0,
                    out_stride=1, out_off_y=0, out_off_x=0, out_h_phys=ho, out_w_phys=wo, in_off_y=0, in_off_x=0, phase_sum=0,
                    precision=_lib.RCF_PREC_F16X2 if f32 else _lib.RCF_PREC_BF16, storage=_lib.RCF_STORE_FP32 if f32 else _lib.RCF_STORE_BF16)
