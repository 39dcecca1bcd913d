"""ctypes binding of libsisr_hip.so (C ABI: include/sisr_hip.h).

The library is the product: there is NO CPU or eager-PyTorch fallback.  If it is missing or does
not match this mirror of the header, importing the hot path raises.
"""
import ctypes as C
import os
import subprocess

_HERE = os.path.dirname(os.path.abspath(__file__))
CSRC = os.path.join(_HERE, 'csrc')
LIB_PATH = os.environ.get('SISR_LIB') or os.path.join(CSRC, 'libsisr_hip.so')      # SISR_LIB: developer builds

# enums (sisr_hip.h)
PRO_NONE, PRO_ACT, PRO_AFFINE_ACT, PRO_BNBWD, PRO_BNACT_BWD, PRO_ACT_BWD, PRO_TANH_BWD, PRO_RES_AFFINE = range(8)
X_NHWC, X_NCHW, X_UNSHUFFLE2 = range(3)
Y_NHWC, Y_NCHW, Y_SHUFFLE2 = range(3)
EPI_NONE, EPI_TANH = range(2)

_f = C.c_void_p      # device pointers travel as void*
_i32 = C.c_int32
_i64 = C.c_int64
_f32 = C.c_float


class ConvPlan(C.Structure):
    _fields_ = [(n, _i32) for n in (
        'TH', 'TW', 'TN', 'tiles_y', 'tiles_x', 'n_groups', 'n_tiles', 'CK', 'PS', 'KROWP', 'n_chunk',
        'CoutPad', 'msub', 'nsub', 'lds_bytes', 'wpk_elems', 'variant')] + [
        (n, C.c_uint32) for n in ('m_tiles_x', 'm_thw', 'm_tw', 'm_iw', 'm_wrow')]


class DeepPlan(C.Structure):
    _fields_ = ([(n, _i32) for n in ('enabled', 'TH', 'TW', 'tiles_x', 'tiles_q', 'BN', 'n_ntiles', 'n_chunk', 'split', 'cps',
                                     'PR', 'IW', 'IH_max', 'NIT', 'wimg_elems', 'classes')] +
                [('ws_bytes', _i64)] +
                [(n, C.c_uint32) for n in ('m_tiles_x', 'm_tw', 'm_ho', 'm_pr', 'm_iw', 'rsv2')])


class ConvDesc(C.Structure):
    _fields_ = ([(n, _f) for n in ('x1', 'x2', 'pa', 'pb', 'pd', 'ps', 'pt', 'wpk', 'bias', 'res', 'y',
                                   'stat_part', 'cnt_part', 'bnb_x', 'bnb_scale', 'bnb_shift', 'bnb_mean',
                                   'bnb_invstd', 'bnb_slope_p', 'bnb_part', 'x_out',
                                   'fin_stat', 'fin_cnt', 'fin_gamma', 'fin_beta', 'fin_rm', 'fin_rv', 'fin_k')] +
                [(n, _i32) for n in ('N', 'H', 'W', 'Cin', 'Ho', 'Wo', 'Cout', 'KH', 'KW', 'stride',
                                     'pad_y', 'pad_x', 'x_mode', 'pro_mode')] +
                [('pro_slope_p', _f), ('pro_slope', _f32)] +
                [('y_mode', _i32), ('epi_act', _i32), ('bnb_act', _i32), ('bnb_slope', _f32)] +
                [(n, _i32) for n in ('y_sy', 'y_oy', 'y_sx', 'y_ox', 'y_H', 'y_W')] +
                [(n, _i32) for n in ('x_bf16', 'y_bf16', 'res_bf16', 'bnbx_bf16')] +
                [('fin_rows', _i32), ('fin_momentum', _f32), ('fin_eps', _f32), ('mfma_split', _i32)] +
                [('plan', ConvPlan)] +
                [('wdeep', _f), ('deep_ws', _f), ('epi_scale_p', _f), ('wdeep_c', _f * 4), ('deep_ckh', _i32 * 4), ('deep', DeepPlan)])


class WgradDeepPlan(C.Structure):
    _fields_ = ([(n, _i32) for n in ('enabled', 'TH', 'TW', 'tiles_x', 'tiles_q', 'n_tiles', 'PR', 'IW', 'IH_max', 'IWd', 'IHd_max',
                                     'NPOS_max', 'XP_max', 'NITX', 'NITD', 'n_pb', 'tiles_per_pb', 'n_cib', 'n_cob', 'lds_bytes', 'slab_bf16', 'batch_first_wg')] +
                [(n, C.c_uint32) for n in ('m_tiles_x', 'm_tw', 'm_ho', 'm_pr', 'm_iw', 'm_iwd')])


class WgradDesc(C.Structure):
    _fields_ = ([(n, _f) for n in ('x1', 'x2', 'pa', 'pb', 'pd', 'ps', 'pt',
                                   'g1', 'g2', 'qa', 'qb', 'qd', 'qs', 'qt', 'slab', 'bias_slab')] +
                [(n, _i32) for n in ('N', 'H', 'W', 'Cin', 'Ho', 'Wo', 'Cout', 'KH', 'KW', 'stride',
                                     'pad_y', 'pad_x', 'x_mode', 'pro_mode')] +
                [('pro_slope_p', _f), ('pro_slope', _f32), ('g_mode', _i32), ('gpro_mode', _i32),
                 ('gpro_slope_p', _f), ('gpro_slope', _f32)] +
                [(n, _i32) for n in ('TH', 'TW', 'TN', 'tiles_y', 'tiles_x', 'n_groups', 'n_tiles',
                                     'CK', 'PS', 'KROWP', 'n_chunk', 'CoutPad',
                                     'NJ', 'NP', 'NT', 'TSTEP', 'TVALID',
                                     'grid_x', 'n_slabs', 'slab_elems', 'lds_bytes')] +
                [(n, C.c_uint32) for n in ('m_tiles_x', 'm_tiles_y', 'm_iw', 'm_twp', 'm_kw')] +
                [('x_bf16', _i32), ('g_bf16', _i32), ('mfma_split', _i32)] +
                [('slab_stride', _i64), ('deep', WgradDeepPlan)])


class WeightDesc(C.Structure):
    _fields_ = ([(n, _f) for n in ('w_orig', 'u', 'v', 'u_used', 'v_used', 'sigma', 'sn_work', 'wpk_fwd', 'wpk_dgrad')] +
                [(n, _i32) for n in ('Cout', 'Cin', 'KH', 'KW', 'training', 'shuffle2',
                                     'f_CK', 'f_PS', 'f_KROWP', 'f_n_chunk', 'f_CoutPad',
                                     'd_CK', 'd_PS', 'd_KROWP', 'd_n_chunk', 'd_CoutPad')] +
                [('wpk_dcls', _f * 4)] +
                [(n, _i32 * 4) for n in ('c_KH', 'c_KW', 'c_R0y', 'c_R0x', 'c_CK', 'c_PS', 'c_KROWP',
                                         'c_n_chunk', 'c_CoutPad')] +
                [('wbf_fwd', _f), ('wbf_dgrad', _f), ('bf_f_CoutPad', _i32), ('bf_d_CoutPad', _i32),
                 ('bf_f_CK', _i32), ('bf_d_CK', _i32), ('wbf_dcls', _f * 4), ('bf_c_CoutPad', _i32 * 4),
                 ('bf_f_lanes', _i32), ('bf_d_lanes', _i32), ('f_ldsimg', _i32), ('d_ldsimg', _i32),
                 ('wdp_fwd', _f), ('wdp_dgrad', _f), ('wdp_dcls', _f * 4), ('wdp_scaled', _i32), ('wdp_cls_kw', _i32)])
WLDS_WORDS = 2 * 2 * 9 * 32 * 36


class WeightGradDesc(C.Structure):
    _fields_ = ([(n, _f) for n in ('dwpk', 'w_orig', 'u_used', 'v_used', 'sigma', 'grad', 'dbias_pk',
                                   'grad_bias')] +
                [(n, _i32) for n in ('Cout', 'Cin', 'KH', 'KW', 'shuffle2', 'CK', 'PS', 'KROWP', 'n_chunk',
                                     'CoutPad', 'layout')])


class AdamDesc(C.Structure):
    _fields_ = [('p', _f), ('m', _f), ('v', _f), ('g', _f), ('numel', _i64), ('block_start', _i64)]


class BnBwdDesc(C.Structure):
    _fields_ = ([(n, _f) for n in ('dy', 'x', 'scale', 'shift', 'mean', 'invstd', 'gamma', 'work',
                                   'qa', 'qb', 'qd', 'dgamma', 'dbeta', 'dslope')] +
                [('P', _i64), ('C', _i32), ('act_mode', _i32), ('slope_p', _f), ('slope', _f32),
                 ('grid', _i32), ('dy_bf16', _i32), ('x_bf16', _i32)])


_SIGS = {
    'sisr_conv2d_plan': [C.POINTER(ConvDesc)],
    'sisr_conv2d_f32': [C.POINTER(ConvDesc), _f],
    'sisr_wgrad_plan': [C.POINTER(WgradDesc), _i32],
    'sisr_conv2d_wgrad_f32': [C.POINTER(WgradDesc), _f],
    'sisr_conv2d_plan_bf16': [C.POINTER(ConvDesc)],
    'sisr_conv2d_bf16': [C.POINTER(ConvDesc), _f],
    'sisr_conv2d_deep_plan': [C.POINTER(ConvDesc), _i32, _i32, _i32],
    'sisr_conv2d_deep_eligible': [C.POINTER(ConvDesc)],
    'sisr_conv2d_trunk_eligible': [C.POINTER(ConvDesc)],
    'sisr_conv2d_bf16_parts': [C.POINTER(ConvDesc)],
    'sisr_conv2d_trunk_f32_eligible': [C.POINTER(ConvDesc)],
    'sisr_conv2d_f32_parts': [C.POINTER(ConvDesc)],
    'sisr_conv2d_f32_bnb_parts': [C.POINTER(ConvDesc)],
    'sisr_conv2d_thin_eligible': [C.POINTER(ConvDesc)],
    'sisr_conv2d_toimage_eligible': [C.POINTER(ConvDesc)],
    'sisr_conv2d_toimage_f32_eligible': [C.POINTER(ConvDesc)],
    'sisr_wgrad_trunk_eligible': [C.POINTER(WgradDesc)],
    'sisr_wgrad_bf16_slabs': [C.POINTER(WgradDesc)],
    'sisr_wgrad_trunk_f32_eligible': [C.POINTER(WgradDesc)],
    'sisr_wgrad_f32_slabs': [C.POINTER(WgradDesc)],
    'sisr_wgrad_thin_eligible': [C.POINTER(WgradDesc)],
    'sisr_wgrad_toimage_eligible': [C.POINTER(WgradDesc)],
    'sisr_wgrad_toimage_f32_eligible': [C.POINTER(WgradDesc)],
    'sisr_wgrad_plan_bf16': [C.POINTER(WgradDesc), _i32],
    'sisr_wgrad_deep_plan': [C.POINTER(WgradDesc), _i32],
    'sisr_wgrad_deep_eligible': [C.POINTER(WgradDesc)],
    'sisr_wgrad_deep_batch': [C.POINTER(WgradDesc), _f, _i32, _f],
    'sisr_wgrad_trunk_batch_arg_bytes': [],
    'sisr_wgrad_trunk_batch_args': [C.POINTER(WgradDesc), _i32, _f],
    'sisr_wgrad_trunk_batch': [C.POINTER(WgradDesc), _f, _i32, _i32, _f],
    'sisr_wgrad_trunk_f32_batch_arg_bytes': [],
    'sisr_wgrad_trunk_f32_batch_args': [C.POINTER(WgradDesc), _i32, _f],
    'sisr_wgrad_trunk_f32_batch': [C.POINTER(WgradDesc), _f, _i32, _i32, _f],
    'sisr_conv2d_wgrad_bf16': [C.POINTER(WgradDesc), _f],
    'sisr_tr16_selftest': [_f, _f],
    'sisr_slab_reduce_f32': [_f, _f, _i32, _i64, _i64, _f],
    'sisr_slab_reduce_multi': [_f, _f, _f, _f, _f, _i32, _f],
    'sisr_wgrad_bf16_slab_lead': [C.POINTER(WgradDesc)],
    'sisr_weights_prepare': [_f, _i32, _i32, _i32, _f],
    'sisr_weights_sn': [_f, _i32, _i32, _i32, _f],
    'sisr_weights_pack': [_f, _i32, _i32, _i32, _f],
    'sisr_weights_pack_deep': [_f, _i32, _i32, _i32, _f],
    'sisr_weights_grad': [_f, _i32, _f, _i32, _f],
    'sisr_weights_grad_fast': [_f, _i32, _f, _i32, _i32, _f],
    'sisr_weights_grad_tiles': [C.POINTER(WeightGradDesc)],
    'sisr_bn_finalize': [_f, _f, _i32, _i32, _f, _f, _f, _f, _f32, _f32, _f, _f, _f, _f, _f],
    'sisr_bn_eval_consts': [_f, _f, _f, _f, _f32, _i32, _f, _f, _f],
    'sisr_bn_bwd_plan': [C.POINTER(BnBwdDesc)],
    'sisr_bn_bwd': [C.POINTER(BnBwdDesc), _f],
    'sisr_bn_bwd_finalize': [C.POINTER(BnBwdDesc), _f],
    'sisr_bn_bwd_finalize_slab': [C.POINTER(BnBwdDesc), _f, _f, _i32, _i64, _i64, _f],
    'sisr_eltwise_res_affine': [_f, _f, _f32, _f, _f, _f, _f, _i64, _i32, _i32, _f],
    'sisr_prelu_slope_grad': [_f, _f, _i64, _f, _f, _i32, _f],
    'sisr_add': [_f, _f, _f, _i64, _i32, _f],
    'sisr_adam_blocks': [_i64],
    'sisr_adam_step': [_f, _i32, _i64] + [C.c_double] * 7 + [_f],
    'sisr_nhwc_to_nchw': [_f, _f, _f, _f, _f32, _f, _i64, _i32, _i32, _i32, _i32, _i32, _f],
    'sisr_nchw_to_nhwc': [_f, _i64, _f, _i32, _i32, _i32, _i32, _i32, _f],
    'sisr_nchw_grad_to_nhwc4': [_f, _f, _f, _i32, _i32, _i32, _i32, _i32, _f],
    'sisr_maxpool2_fwd': [_f, _f, _i32, _i32, _i32, _i32, _i32, _f],
    'sisr_maxpool2_relu_bwd': [_f, _f, _f, _i32, _i32, _i32, _i32, _i32, _f],
    'sisr_add_relu_masked': [_f, _f, _f, _f, _i64, _i32, _f],
    'sisr_fc_forward': [_f, _f32, _f, _f, _f, _i32, _i32, _i32, _i32, _f],
    'sisr_fc_dgrad_splits': [_i32, _i32],
    'sisr_fc_dgrad': [_f, _f, _f, _f, _i32, _i32, _i32, _f],
    'sisr_fc_wgrad': [_f, _f, _f32, _f, _f, _i32, _i32, _i32, _f],
    'sisr_act_bwd': [_f, _f, _f, _i64, _i32, _f32, _f],
    'sisr_fc_head_ws_floats': [_i32],
    'sisr_fc_head_forward': [_f, _f, _f, _f, _f, _f32, _f, _f, _f, _i32, _i32, _i32, _f],
    'sisr_fc_head_backward': [_f, _f, _f, _f, _f32, _f, _f, _f, _f, _i32, _i32, _f],
    'sisr_fc1_dgrad': [_f, _f, _f, _i32, _i32, _i32, _f],
    'sisr_fc_wgrad_rows': [_f, _f, _f32, _f, _i32, _i32, _i32, _f],
    'sisr_resize_coeffs': [_i32, _i32, _f, _f],
    'sisr_resize_u8_normalize': [_f, _f, _i32, _i32, _i32, _i32, _i32, _i32, _f, _f, _i32, _f, _f, _i32, _f32, _f32, _f],
    'sisr_bicubic_fwd': [_f, _f, _i32, _i32, _i32, _i32, _i32, _i32, _f],
    'sisr_bicubic_bwd': [_f, _f, _f, _i32, _i32, _i32, _i32, _i32, _f],
    'sisr_struct_sizes': [C.POINTER(_i32), _i32],
    'sisr_device_info': [C.POINTER(_i32), C.POINTER(_i32), C.c_char_p, _i32],
    'sisr_mfma_selftest': [_f, _f],
    'sisr_clear_last_error': [],
}
EXPORTS = sorted(list(_SIGS) + ['sisr_version'])

_lib = None


def build(verbose=False):
    """Compile csrc/*.hip for gfx950 with hipcc (cross-compiles without a GPU)."""
    r = subprocess.run(['make', '-C', CSRC, '-j8'], capture_output=True, text=True)
    if verbose or r.returncode != 0:
        print(r.stdout[-4000:], r.stderr[-4000:])
    if r.returncode != 0 or not os.path.exists(LIB_PATH):
        raise RuntimeError('building libsisr_hip.so failed (hipcc --offload-arch=gfx950)')
    return LIB_PATH


def lib():
    """The loaded library.  Raises RuntimeError if it is absent or mismatched -- no fallback."""
    global _lib
    if _lib is not None:
        return _lib
    if not os.path.exists(LIB_PATH):
        raise RuntimeError(
            'libsisr_hip.so not found at %s: the MI355X HIP library IS the implementation of this '
            'path (no CPU/eager fallback). Build it with `python __graft_entry__.py build` or '
            '`make -C %s`.' % (LIB_PATH, CSRC))
    L = C.CDLL(LIB_PATH)
    for name, args in _SIGS.items():
        fn = getattr(L, name)
        fn.argtypes = args
        fn.restype = C.c_int
    L.sisr_adam_blocks.restype = C.c_int64
    L.sisr_wgrad_bf16_slab_lead.restype = C.c_int64
    L.sisr_version.restype = C.c_char_p
    L.sisr_version.argtypes = []
    sizes = (_i32 * 8)()
    n = L.sisr_struct_sizes(sizes, 8)
    mine = [C.sizeof(t) for t in (ConvDesc, WgradDesc, WeightDesc, WeightGradDesc, BnBwdDesc, ConvPlan, DeepPlan, WgradDeepPlan)]
    if n != 8 or list(sizes[:8]) != mine:
        raise RuntimeError('libsisr_hip.so does not match the Python mirror of sisr_hip.h: %s vs %s'
                           % (list(sizes[:8]), mine))
    _lib = L
    return L


def check(status, what):
    if status != 0:
        raise RuntimeError('%s failed with status %d' % (what, status))


def check_count(value, what):
    """entry points that answer a sizing question return a count >= 0 or a negative status"""
    if value < 0:
        raise RuntimeError('%s failed with status %d' % (what, value))
    return value
