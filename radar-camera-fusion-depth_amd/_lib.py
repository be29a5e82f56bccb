'''
ctypes binding of librcf_hip.so (include/rcf_hip.h).  There is NO fallback: if the library is missing
or a call returns non-zero the product path raises.  Build with ``python -c "import __graft_entry__ as g; g.build()"``.
'''

import ctypes
import os
from ctypes import POINTER, Structure, c_char_p, c_double, c_float, c_int, c_longlong, c_size_t, c_void_p

_HERE = os.path.dirname(os.path.abspath(__file__))
LIB_PATH = os.environ.get('RCF_HIP_LIB') or os.path.join(_HERE, 'librcf_hip.so')   # RCF_HIP_LIB: an explicitly chosen build of the same library

RCF_GATHER_DIRECT, RCF_GATHER_NEAREST, RCF_GATHER_ZERO_INSERT, RCF_GATHER_STRIDED2 = 0, 1, 2, 3
RCF_PHASE_UP2X_FWD, RCF_PHASE_UP2X_DGRAD, RCF_PHASE_S2_DGRAD = 0, 1, 2
RCF_W_FORWARD, RCF_W_DGRAD = 0, 1
RCF_ACT_NONE, RCF_ACT_LEAKY_RELU = 0, 1
RCF_PIXEL_U8, RCF_PIXEL_U16, RCF_PIXEL_I32 = 0, 1, 2


class ConvDesc(Structure):
    _fields_ = [(n, c_int) for n in (
        'n', 'h_in', 'w_in', 'c1', 'c2', 'h_src1', 'w_src1', 'gather1', 'h_out', 'w_out', 'c_out',
        'ksize', 'stride', 'pad', 'pad_x', 'w_mode', 'w_o', 'w_i', 'w_i_off', 'accumulate',
        'out_stride', 'out_off_y', 'out_off_x', 'out_h_phys', 'out_w_phys', 'in_off_y', 'in_off_x', 'phase_sum', 'precision', 'storage')]


class ConvInfo(Structure):
    _fields_ = [('packed_weight_floats', c_size_t), ('n_partials', c_int),
                ('wgrad_workspace_floats', c_size_t), ('kernel_id', c_int), ('wgrad_kernel_id', c_int),
                ('bn_on_load', c_int), ('wgrad_bn_on_load', c_int), ('fwd_act', c_int), ('bn_bwd_sums', c_int)]


class PackItem(Structure):     # rcf_pack_item
    _fields_ = [('desc', POINTER(ConvDesc)), ('w_oihw', c_void_p), ('packed', c_void_p), ('amax_w', c_void_p)]


class ConvScales(Structure):   # rcf_conv_scales: device pointers to per-tensor maxima (RCF_PREC_F16X2), nullable
    _fields_ = [('amax_in1', c_void_p), ('amax_in2', c_void_p), ('amax_w', c_void_p), ('amax_dz', c_void_p)]


class AmaxItem(Structure):     # rcf_amax_item
    _fields_ = [('x', c_void_p), ('n', c_longlong), ('amax', c_void_p)]


class PhaseItem(Structure):    # rcf_phase_item
    _fields_ = [('w_oihw', c_void_p), ('out', c_void_p), ('o', c_int), ('i', c_int), ('mode', c_int)]


RCF_PREC_FP32, RCF_PREC_BF16, RCF_PREC_F16X2 = 0, 1, 2
RCF_STORE_FP32, RCF_STORE_BF16 = 0, 1

_P = c_void_p
_SIGNATURES = {
    'rcf_version': (c_char_p, []),
    'rcf_device_ok': (c_int, []),
    'rcf_conv2d_query': (c_int, [POINTER(ConvDesc), POINTER(ConvInfo)]),
    'rcf_conv2d_pack_weights': (c_int, [POINTER(ConvDesc), _P, _P, _P]),
    'rcf_conv2d_pack_weights_batch': (c_int, [POINTER(PackItem), c_int, _P]),
    'rcf_phase_weights_batch': (c_int, [POINTER(PhaseItem), c_int, _P]),
    'rcf_conv2d_fwd': (c_int, [POINTER(ConvDesc), _P, _P, _P, _P, _P, _P]),
    'rcf_conv2d_fwd_bn': (c_int, [POINTER(ConvDesc), _P, _P, _P, _P, _P, _P, _P, _P]),
    'rcf_conv2d_fwd_act': (c_int, [POINTER(ConvDesc), _P, _P, _P, _P, _P, _P, _P]),
    'rcf_scale_channels': (c_int, [_P, _P, _P, c_int, c_int, _P]),
    'rcf_conv2d_wgrad': (c_int, [POINTER(ConvDesc), _P, _P, _P, _P, _P, _P]),
    'rcf_conv2d_wgrad_bn': (c_int, [POINTER(ConvDesc), _P, _P, _P, _P, _P, _P, _P, _P]),
    'rcf_phase_weights': (c_int, [_P, _P, c_int, c_int, c_int, _P]),
    'rcf_phase_wgrad_fold': (c_int, [_P, _P, c_int, c_int, _P]),
    'rcf_phase_wgrad_gather_s2': (c_int, [_P, _P, c_int, c_int, _P]),
    'rcf_bn_finalize': (c_int, [_P, c_int, c_int, c_double, _P, _P, _P, _P, c_float, c_float, c_int, _P, _P]),
    'rcf_bn_act_fwd': (c_int, [_P, _P, _P, _P, c_longlong, c_int, c_int, _P]),
    'rcf_fuse_fwd': (c_int, [_P, _P, _P, _P, _P, _P, c_longlong, c_int, _P]),
    'rcf_ew_blocks': (c_int, [c_longlong, c_int]),
    'rcf_bn_act_bwd_reduce': (c_int, [_P, _P, _P, _P, _P, c_longlong, c_int, c_int, c_int, _P]),
    'rcf_bn_bwd_finalize': (c_int, [_P, c_int, c_int, c_int, c_double, _P, _P, _P, _P]),
    'rcf_bn_act_bwd_apply': (c_int, [_P, _P, _P, _P, _P, _P, _P, c_int, c_longlong, c_int, c_int, c_int, _P]),
    'rcf_roi_pool_fwd': (c_int, [_P, _P, _P, _P, c_int, c_int, c_int, c_int, c_int, c_int, c_int, c_float, c_int, c_int, _P]),
    'rcf_roi_pool_bwd': (c_int, [_P, _P, _P, _P, c_int, c_int, c_int, c_int, c_int, c_int, c_int, c_int, c_int, _P]),
    'rcf_roi_pool_bwd_gather': (c_int, [_P, _P, _P, _P, c_int, c_int, c_int, c_int, c_int, c_int, c_int, c_int, c_float, c_int, c_int, _P]),
    'rcf_fc_fwd': (c_int, [_P, _P, _P, _P, c_int, c_int, c_int, c_int, c_int, c_int, c_int, _P]),
    'rcf_fc_bwd_workspace_floats': (c_size_t, [c_int, c_int, c_int]),
    'rcf_fc_bwd': (c_int, [_P, _P, _P, _P, _P, _P, _P, _P, c_int, c_int, c_int, c_int, c_int, c_int, c_int, _P]),
    'rcf_bce_workspace_doubles': (c_size_t, []),
    'rcf_bce_loss_fwd': (c_int, [_P, _P, _P, _P, _P, _P, c_longlong, c_float, _P]),
    'rcf_bce_loss_bwd': (c_int, [_P, _P, _P, _P, _P, _P, c_longlong, c_float, _P]),
    'rcf_transform_workspace_bytes': (c_size_t, [c_int]),
    'rcf_transform_images': (c_int, [_P, _P, c_int, c_int, c_int, _P, _P, _P, _P, _P, _P, _P, _P, c_int, _P, _P]),
    'rcf_transform_flip': (c_int, [_P, _P, c_int, c_int, c_int, c_int, _P, _P, _P]),
    'rcf_decode_image_u8': (c_int, [_P, _P, c_int, c_int, c_int, c_int, c_int, _P, c_int, _P]),
    'rcf_decode_map': (c_int, [_P, c_int, _P, _P, c_int, c_int, c_int, c_int, c_int, _P, c_float, c_int, _P]),
    'rcf_encode_map_u32': (c_int, [_P, _P, c_longlong, c_float, _P]),
    'rcf_points_to_depth_map_workspace_bytes': (c_size_t, [c_int, c_int]),
    'rcf_points_to_depth_map': (c_int, [_P, _P, _P, c_int, _P, c_int, c_int, _P, _P]),
    'rcf_head_bn_blocks': (c_int, [c_int, c_int, c_int, c_int]),
    'rcf_head_bn_bwd_reduce': (c_int, [_P, _P, _P, _P, _P, c_int, c_int, c_int, c_int, _P]),
    'rcf_head_bn_bwd_apply': (c_int, [_P, _P, _P, _P, _P, _P, c_int, c_int, c_int, c_int, _P]),
    'rcf_fuse_bwd_reduce': (c_int, [_P, _P, _P, _P, _P, _P, c_longlong, c_int, _P]),
    'rcf_fuse_bwd_apply': (c_int, [_P, _P, _P, _P, _P, _P, _P, _P, _P, _P, c_int, c_longlong, c_int, _P]),
    'rcf_maxpool3x3s2_fwd': (c_int, [_P, _P, _P, c_int, c_int, c_int, c_int, _P]),
    'rcf_maxpool3x3s2_bwd': (c_int, [_P, _P, _P, c_int, c_int, c_int, c_int, c_int, _P]),
    'rcf_upsample_nearest_bwd': (c_int, [_P, _P, c_int, c_int, c_int, c_int, c_int, c_int, c_int, _P]),
    'rcf_head_fwd': (c_int, [_P, _P, _P, _P, c_int, c_int, c_int, c_int, c_float, c_float, _P]),
    'rcf_head_fwd_bn': (c_int, [_P, _P, _P, _P, _P, c_int, c_int, c_int, c_int, c_float, c_float, _P]),
    'rcf_head_bwd_logit': (c_int, [_P, _P, _P, c_longlong, c_float, c_float, _P]),
    'rcf_head_bwd_dgrad': (c_int, [_P, _P, _P, c_int, c_int, c_int, c_int, _P]),
    'rcf_head_wgrad_workspace_floats': (c_size_t, [c_int, c_int, c_int, c_int]),
    'rcf_head_bwd_wgrad': (c_int, [_P, _P, _P, _P, c_int, c_int, c_int, c_int, _P]),
    'rcf_head_bwd_wgrad_bn': (c_int, [_P, _P, _P, _P, _P, c_int, c_int, c_int, c_int, _P]),
    'rcf_loss_workspace_floats': (c_size_t, [c_longlong]),
    'rcf_l1_loss_fwd': (c_int, [_P, _P, _P, _P, _P, c_longlong, _P]),
    'rcf_masked_loss_fwd': (c_int, [_P, _P, _P, _P, _P, c_longlong, c_int, _P]),
    'rcf_masked_loss_bwd': (c_int, [_P, _P, _P, _P, _P, c_float, _P, c_longlong, c_int, _P]),
    'rcf_smoothness_loss_fwd': (c_int, [_P, _P, _P, _P, c_int, c_int, c_int, c_int, _P]),
    'rcf_smoothness_loss_bwd': (c_int, [_P, _P, _P, _P, c_float, _P, c_int, c_int, c_int, c_int, _P]),
    'rcf_l1_loss_value': (c_int, [_P, c_float, _P, _P]),
    'rcf_l1_loss_bwd': (c_int, [_P, _P, _P, _P, _P, c_float, _P, c_longlong, _P]),
    'rcf_outlier_removal': (c_int, [_P, _P, _P, c_int, c_int, c_int, c_int, c_float, _P]),
    'rcf_adam_step': (c_int, [_P, _P, _P, _P, c_longlong, c_float, c_float, c_float, c_float, c_float, c_int, _P]),
    'rcf_adam_step_dev': (c_int, [_P, _P, _P, _P, c_longlong, _P, _P]),
    'rcf_nchw_to_nhwc': (c_int, [_P, _P, c_int, c_int, c_int, c_int, _P]),
    'rcf_nhwc_to_nchw': (c_int, [_P, _P, c_int, c_int, c_int, c_int, _P]),
    'rcf_radar_scatter': (c_int, [_P, _P, c_int, c_int, c_int, c_int, c_int, _P, _P, _P]),
    'rcf_radar_scatter_logits': (c_int, [_P, _P, c_int, c_int, c_int, c_int, c_int, _P, _P, _P]),
}
B16_TWINS = ('rcf_bn_act_fwd', 'rcf_fuse_fwd', 'rcf_bn_act_bwd_reduce', 'rcf_bn_act_bwd_apply', 'rcf_fuse_bwd_reduce', 'rcf_fuse_bwd_apply',
             'rcf_head_bn_bwd_reduce', 'rcf_head_bn_bwd_apply', 'rcf_maxpool3x3s2_fwd', 'rcf_maxpool3x3s2_bwd', 'rcf_upsample_nearest_bwd',
             'rcf_head_fwd', 'rcf_head_fwd_bn', 'rcf_head_bwd_dgrad', 'rcf_head_bwd_wgrad', 'rcf_head_bwd_wgrad_bn', 'rcf_roi_pool_fwd',
             'rcf_roi_pool_bwd', 'rcf_roi_pool_bwd_gather', 'rcf_fc_fwd', 'rcf_fc_bwd')
for _name in B16_TWINS:   # NAME_b16: same argument list, NHWC activation tensors hold bf16 (include/rcf_hip.h)
    _SIGNATURES[_name + '_b16'] = _SIGNATURES[_name]
_SIGNATURES.update({
    # two fp16 operand planes with per-tensor scales (RCF_PREC_F16X2)
    'rcf_conv2d_pack_weights_scaled': (c_int, [POINTER(ConvDesc), _P, _P, _P, _P]),
    'rcf_conv2d_fwd_scaled': (c_int, [POINTER(ConvDesc), _P, _P, _P, _P, _P, POINTER(ConvScales), _P]),
    'rcf_conv2d_wgrad_scaled': (c_int, [POINTER(ConvDesc), _P, _P, _P, _P, _P, POINTER(ConvScales), _P]),
    'rcf_conv2d_dgrad_bn_sums': (c_int, [POINTER(ConvDesc), _P, _P, _P, _P, _P, _P, POINTER(ConvScales), _P]),
    'rcf_amax': (c_int, [_P, c_longlong, _P, _P]),
    'rcf_amax_batch': (c_int, [POINTER(AmaxItem), c_int, _P]),
    'rcf_bn_act_fwd_amax': (c_int, [_P, _P, _P, _P, c_longlong, c_int, c_int, _P, _P]),
    'rcf_fuse_fwd_amax': (c_int, [_P, _P, _P, _P, _P, _P, c_longlong, c_int, _P, _P]),
    'rcf_bn_act_bwd_apply_amax': (c_int, [_P, _P, _P, _P, _P, _P, _P, c_int, c_longlong, c_int, c_int, c_int, _P, _P]),
    'rcf_head_bn_bwd_apply_amax': (c_int, [_P, _P, _P, _P, _P, _P, c_int, c_int, c_int, c_int, _P, _P]),
})
_SIGNATURES['rcf_convert'] = (c_int, [_P, c_int, _P, c_int, c_longlong, c_int, _P])
_SIGNATURES['rcf_fuse_wp_infer_supported'] = (c_int, [c_int, c_int])
_SIGNATURES['rcf_fuse_wp_infer_b16'] = (c_int, [_P, _P, _P, _P, _P, _P, _P, c_longlong, c_int, c_int, _P])
_SIGNATURES['rcf_s2d_image_b16'] = (c_int, [_P, _P, c_int, c_int, c_int, c_int, _P])
_SIGNATURES['rcf_s2d_image_f32'] = (c_int, [_P, _P, c_int, c_int, c_int, c_int, _P, _P])
_SIGNATURES['rcf_stem_weights_s2d'] = (c_int, [_P, _P, c_int, c_int, _P])
'''Every symbol include/rcf_hip.h declares, with its ctypes signature.'''

_lib = None


class RcfError(RuntimeError):
    """code: the library's return value (-1 RCF_EINVAL, -2 RCF_EUNSUPPORTED, > 0 a hipError_t); None for host-side failures."""
    code = None


class RcfUnsupported(RcfError):
    """RCF_EUNSUPPORTED: the shape / mode has no kernel -- the only error a caller may answer with another form of the same op."""
    code = -2


def load():
    '''Load librcf_hip.so once.  Raises RcfError (never falls back) when it is absent or incomplete.'''
    global _lib
    if _lib is not None:
        return _lib
    if not os.path.exists(LIB_PATH):
        raise RcfError('librcf_hip.so not found at %s -- the HIP extension is required; run '
                       '__graft_entry__.build() (hipcc --offload-arch=gfx950). There is no CPU fallback.' % LIB_PATH)
    lib = ctypes.CDLL(LIB_PATH)
    for name, (restype, argtypes) in _SIGNATURES.items():
        try:
            fn = getattr(lib, name)
        except AttributeError:
            raise RcfError('librcf_hip.so does not export %s (stale build?)' % name)
        fn.restype = restype
        fn.argtypes = argtypes
    _lib = lib
    return lib


def check(rc, what):
    if rc != 0:
        kind = {-1: 'invalid argument', -2: 'unsupported shape'}.get(rc, 'hipError_t %d' % rc)
        e = (RcfUnsupported if rc == -2 else RcfError)('%s failed: %s' % (what, kind))
        e.code = rc
        raise e
