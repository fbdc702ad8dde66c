"""ctypes binding of libsaicv_hip.so (include/saicv_hip.h).

The product path has no CPU fallback: if the library is missing, or a tensor is not on a
HIP device, the call raises.  PyTorch is used only for device memory and streams.
"""
import ctypes
import os
from ctypes import POINTER, Structure, c_char_p, c_double, c_float, c_int, c_long, c_size_t, c_void_p

import torch

HERE = os.path.dirname(os.path.abspath(__file__))
LIB_PATH = os.path.join(HERE, 'libsaicv_hip.so')

BF16, F32 = 0, 1


class ConvDesc(Structure):
    _fields_ = [('N', c_int), ('H', c_int), ('W', c_int), ('C', c_int), ('K', c_int), ('R', c_int),
                ('S', c_int), ('stride', c_int), ('pad', c_int), ('OH', c_int), ('OW', c_int),
                ('dtype', c_int)]


_P = c_void_p
_PD = POINTER(ConvDesc)


class AttnDesc(Structure):
    """saicv_attn_desc (include/saicv_hip.h): streaming attention problem descriptor."""
    _fields_ = [('q', _P), ('k', _P), ('v', _P),
                ('q_rs', c_long), ('k_rs', c_long), ('v_rs', c_long),
                ('q_bs', c_long), ('k_bs', c_long), ('v_bs', c_long),
                ('out', _P), ('o_rs', c_long), ('o_bs', c_long),
                ('dout', _P), ('dq', _P), ('dk', _P), ('dv', _P),
                ('lse', _P), ('dsum', _P), ('key_bias', _P), ('rel_h', _P), ('rel_w', _P),
                ('d_rel_h', _P), ('d_rel_w', _P),
                ('Sh', c_int), ('Sw', c_int), ('B', c_int), ('H', c_int), ('Nq', c_int), ('Nk', c_int),
                ('scale', c_float), ('dropout_p', c_float), ('seed', ctypes.c_uint32), ('seed_device', _P)]


_PA = POINTER(AttnDesc)


class DgradFuse(Structure):
    """saicv_dgrad_fuse (include/saicv_hip.h)"""
    _fields_ = [('addend', c_void_p), ('addend_gate', c_void_p), ('bn_y', c_void_p), ('bn_mask', c_void_p),
                ('bn_mean', c_void_p), ('bn_invstd', c_void_p), ('part_g', c_void_p), ('part_gx', c_void_p),
                ('part_rows', c_int), ('reserved', c_int)]


_PF = POINTER(DgradFuse)


class PackDesc(Structure):
    """saicv_pack_desc (include/saicv_hip.h)"""
    _fields_ = [('w', c_void_p), ('sO', c_long), ('sI', c_long), ('sR', c_long), ('sS', c_long),
                ('O', c_int), ('I', c_int), ('R', c_int), ('S', c_int), ('Ip', c_int), ('Op', c_int),
                ('wf', c_void_p), ('wd', c_void_p), ('tile_begin', c_int), ('tiles_i', c_int), ('tiles_o', c_int),
                ('tile', c_int)]

class MixPlan(Structure):
    """saicv_mix_plan (include/saicv_hip.h)"""
    _fields_ = [('mode', c_int), ('yl', c_int), ('yh', c_int), ('xl', c_int), ('xh', c_int), ('lam', ctypes.c_float),
                ('one_minus_lam', ctypes.c_float), ('label_lam', ctypes.c_float), ('label_one_minus_lam', ctypes.c_float)]


class EraseBox(Structure):
    """saicv_erase_box (include/saicv_hip.h)"""
    _fields_ = [('b', c_int), ('top', c_int), ('left', c_int), ('h', c_int), ('w', c_int), ('mode', c_int), ('color', ctypes.c_float * 4)]


# name -> (restype, argtypes); mirrors include/saicv_hip.h one to one
SIGNATURES = {
    'saicv_version': (c_int, []),
    'saicv_last_error_string': (c_char_p, []),
    'saicv_set_deterministic': (c_int, [c_int]),
    'saicv_get_deterministic': (c_int, []),
    'saicv_deterministic_prepare': (c_int, [_P]),
    'saicv_pack_input': (c_int, [c_int, _P, c_long, c_long, c_long, c_long, _P, c_int, c_int, c_int, c_int, c_int, _P]),
    'saicv_pack_weight': (c_int, [c_int, _P, c_long, c_long, c_long, c_long, c_int, c_int, c_int, c_int, c_int, c_int, _P, _P, _P]),
    'saicv_unpack_wgrad': (c_int, [_P, c_int, c_int, c_int, c_int, c_int, _P, c_long, c_long, c_long, c_long, c_int, _P]),
    'saicv_pack_weight_batched': (c_int, [c_int, _P, c_int, c_int, _P]),
    'saicv_pack_input_s2d': (c_int, [c_int, _P, c_long, c_long, c_long, c_long, _P, c_int, c_int, c_int, c_int, c_int, c_int, _P]),
    'saicv_pack_weight_s2d': (c_int, [c_int, _P, c_long, c_long, c_long, c_long, c_int, c_int, c_int, c_int, c_int, _P, _P]),
    'saicv_unpack_wgrad_s2d': (c_int, [_P, c_int, c_int, c_int, c_int, c_int, _P, c_long, c_long, c_long, c_long, c_int, _P]),
    'saicv_conv2d_stat_rows': (c_int, [_PD]),
    'saicv_conv2d_fwd': (c_int, [_PD, _P, _P, _P, _P, c_int, _P, _P, _P]),
    'saicv_conv2d_dgrad': (c_int, [_PD, _P, _P, _P, _P]),
    'saicv_conv2d_wgrad': (c_int, [_PD, _P, _P, _P, _P]),
    'saicv_conv2d_wgrad_bias': (c_int, [_PD, _P, _P, _P, _P, _P]),
    'saicv_colsum': (c_int, [c_int, _P, c_int, c_int, _P, _P]),
    'saicv_linear_fwd': (c_int, [c_int, _P, _P, _P, _P, c_int, c_int, c_int, c_int, _P, _P, c_int, _P]),
    'saicv_linear_dgrad': (c_int, [c_int, _P, _P, _P, c_int, c_int, c_int, _P, _P]),
    'saicv_linear_wgrad': (c_int, [c_int, _P, _P, _P, _P, c_int, c_int, c_int, _P]),
    'saicv_conv2d_dgrad_add': (c_int, [_PD, _P, _P, _P, _P, _P]),
    'saicv_conv2d_dgrad_stat_rows': (c_int, [_PD]),
    'saicv_conv2d_dgrad_fused': (c_int, [_PD, _P, _P, _PF, _P, _P]),
    'saicv_conv2d_fwd_stats': (c_int, [_PD, _P, _P, _P, _P, _P, c_int, _P]),
    'saicv_bn_act_fwd_stats': (c_int, [c_int, _P, _P, _P, _P, _P, c_int, c_double, _P, _P, _P, _P, c_double, c_double, _P, _P, _P,
                                       c_size_t, c_int, c_int, _P, _P]),
    'saicv_bn_act_fwd_join': (c_int, [c_int, _P, _P, _P, _P, _P, _P, _P, _P, _P, c_int, c_double, _P, _P, _P, _P, c_double, c_double,
                                      _P, _P, _P, c_size_t, c_int, c_int, _P, _P]),
    'saicv_bn_act_bwd_inline': (c_int, [c_int, _P, _P, _P, _P, _P, _P, _P, _P, c_int, _P, _P, _P, _P, c_size_t, c_int, c_int,
                                        c_int, _P]),
    'saicv_bn_act_bwd_from_partials': (c_int, [c_int, _P, _P, _P, _P, _P, _P, _P, _P, c_int, _P, _P, _P, _P, c_size_t, c_int,
                                               c_int, c_int, _P, _P]),
    'saicv_row_scale': (c_int, [c_int, _P, _P, _P, c_size_t, c_int, c_int, _P]),
    'saicv_bn_ws_floats': (c_size_t, [c_int]),
    'saicv_bn_finalize_fwd': (c_int, [_P, _P, c_int, c_int, c_double, _P, _P, _P, _P, c_double, c_double, _P, _P, _P, _P, _P, _P, _P]),
    'saicv_bn_eval_coeffs': (c_int, [c_int, _P, _P, _P, _P, c_double, _P, _P, _P]),
    'saicv_bn_act_fwd': (c_int, [c_int, _P, _P, _P, _P, _P, c_size_t, c_int, c_int, _P, _P]),
    'saicv_bn_bwd_ws_floats': (c_size_t, [c_size_t, c_int, c_int]),
    'saicv_bn_act_bwd': (c_int, [c_int, _P, _P, _P, _P, _P, _P, _P, _P, _P, _P, _P, c_size_t, c_int, c_int, c_int, _P, _P]),
    'saicv_maxpool_fwd': (c_int, [c_int, _P, _P, _P, c_int, c_int, c_int, c_int, c_int, c_int, c_int, c_int, c_int, _P]),
    'saicv_maxpool_bwd': (c_int, [c_int, _P, _P, _P, c_int, c_int, c_int, c_int, c_int, c_int, c_int, c_int, c_int, _P]),
    'saicv_bn_relu_maxpool_fwd': (c_int, [c_int, _P, _P, _P, _P, _P] + [c_int] * 9 + [_P]),
    'saicv_bn_relu_maxpool_bwd_ws_floats': (c_size_t, [c_int]),
    'saicv_bn_relu_maxpool_bwd': (c_int, [c_int, _P, _P, _P, _P, _P, _P, _P, _P, _P, _P, _P, c_int, _P] + [c_int] * 9 + [_P]),
    'saicv_avgpool_fwd': (c_int, [c_int, _P, _P, c_int, c_int, c_int, _P]),
    'saicv_avgpool_bwd': (c_int, [c_int, _P, _P, c_int, c_int, c_int, _P]),
    'saicv_softmax_ce_fwd': (c_int, [_P, _P, c_int, c_int, c_int, _P, _P, _P, _P]),
    'saicv_scale_by_scalar': (c_int, [c_int, _P, _P, _P, c_size_t, _P]),
    'saicv_sgd_flat': (c_int, [_P, _P, _P, _P, _P, _P, _P, _P, c_size_t, _P]),
    'saicv_adamw_flat': (c_int, [_P, _P, _P, _P, _P, _P, _P, _P, _P, _P, c_size_t, _P]),
    'saicv_grad_stats': (c_int, [_P, c_size_t, _P, _P, _P]),
    'saicv_grad_clip_scale': (c_int, [_P, c_size_t, _P, _P, c_double, _P]),
    'saicv_grad_clip_value': (c_int, [_P, c_size_t, _P, c_double, _P]),
    'saicv_scaler_update': (c_int, [_P, _P, c_double, c_double, c_int, _P]),
    # transformer kernels (tfm.hip)
    'saicv_layernorm_fwd': (c_int, [c_int, _P, _P, _P, _P, _P, _P, c_int, c_int, c_double, _P]),
    'saicv_layernorm_bwd': (c_int, [c_int, _P, _P, _P, _P, _P, _P, _P, _P, _P, _P, c_int, c_int, c_int, _P]),
    'saicv_relu_dropout_fwd': (c_int, [c_int, _P, _P, c_size_t, c_double, ctypes.c_uint, _P, _P]),
    'saicv_relu_dropout_bwd': (c_int, [c_int, _P, _P, _P, c_size_t, c_double, _P]),
    'saicv_dropout_add_layernorm_fwd': (c_int, [c_int, _P, _P, c_double, ctypes.c_uint, _P, _P, _P, _P, _P, _P, _P, c_int, c_int, c_double, _P]),
    'saicv_dropout_add_layernorm_bwd': (c_int, [c_int, _P, _P, _P, _P, _P, c_double, ctypes.c_uint, _P, _P, _P, _P, _P, _P, c_int, c_int, c_int, _P]),
    'saicv_layernorm_bwd_scaled': (c_int, [c_int, _P, _P, _P, _P, _P, _P, _P, _P, _P, _P, c_int, c_int, c_int, _P, c_int, _P, _P]),
    'saicv_layernorm_bwd_ws_floats': (c_size_t, [c_int, c_int]),
    'saicv_gelu_fwd': (c_int, [c_int, _P, _P, c_size_t, _P]),
    'saicv_gelu_bwd': (c_int, [c_int, _P, _P, _P, c_size_t, _P]),
    'saicv_attention_fwd': (c_int, [c_int, _P, _P, _P, c_int, c_int, c_int, c_int, c_double, _P]),
    'saicv_attention_bwd': (c_int, [c_int, _P, _P, _P, _P, _P, c_int, c_int, c_int, c_int, c_double, _P]),
    'saicv_linear_gelu_fwd': (c_int, [c_int, _P, _P, _P, _P, _P, c_int, c_int, c_int, _P]),
    'saicv_linear_dgrad_gelu': (c_int, [c_int, _P, _P, _P, _P, c_int, c_int, c_int, _P]),
    'saicv_linear_gelu_fwd_aux': (c_int, [c_int, _P, _P, _P, _P, _P, c_int, c_int, c_int, _P]),
    'saicv_linear_dgrad_mul': (c_int, [c_int, _P, _P, _P, _P, c_int, c_int, c_int, _P]),
    'saicv_window_partition': (c_int, [c_int, _P, _P, c_int, c_int, c_int, c_int, c_int, _P]),
    'saicv_window_unpartition': (c_int, [c_int, _P, _P, _P, c_int, c_int, c_int, c_int, c_int, _P]),
    'saicv_relpos_fwd': (c_int, [c_int, _P, c_long, c_long, _P, _P, _P, _P, c_int, c_int, c_int, c_int, _P]),
    'saicv_relpos_bwd': (c_int, [c_int, _P, _P, c_long, c_long, _P, _P, _P, _P, _P, _P, _P, c_int, c_int, c_int, c_int, _P]),
    'saicv_relpos_bwd_ws_floats': (c_size_t, [c_int, c_int]),
    'saicv_mask_loss_stats': (c_int, [c_int, _P, _P, _P, c_int, c_int, c_size_t, c_double, c_double, c_double, _P]),
    'saicv_mask_loss_grad': (c_int, [c_int, _P, _P, _P, _P, c_int, c_int, c_size_t, c_double, c_double, _P]),
    'saicv_hyper_product_fwd': (c_int, [c_int, _P, _P, _P, c_int, c_int, c_int, c_int, _P]),
    'saicv_hyper_product_bwd': (c_int, [c_int, _P, _P, _P, _P, _P, c_int, c_int, c_int, c_int, _P]),
    'saicv_upsample4_fwd': (c_int, [c_int, _P, _P, c_int, c_int, c_int, _P]),
    'saicv_upsample4_bwd': (c_int, [c_int, _P, _P, c_int, c_int, c_int, _P]),
    'saicv_mask_loss_stats_up4': (c_int, [c_int, _P, _P, _P, c_int, c_int, c_int, c_int, c_double, c_double, c_double, _P]),
    'saicv_mask_loss_grad_up4': (c_int, [c_int, _P, _P, _P, _P, c_int, c_int, c_int, c_int, c_double, c_double, _P]),
    'saicv_rope_apply': (c_int, [c_int, _P, _P, _P, _P, c_int, c_int, c_int, c_int, c_int, c_int, _P]),
    'saicv_swiglu_fwd': (c_int, [c_int, _P, _P, _P, c_size_t, _P]),
    'saicv_swiglu_bwd': (c_int, [c_int, _P, _P, _P, _P, _P, c_size_t, _P]),
    'saicv_u8_normalize': (c_int, [_P, _P, _P, _P, c_size_t, c_int, _P]),
    'saicv_random_erase': (c_int, [_P, _P, c_int, c_int, c_int, c_int, c_int, ctypes.c_uint, _P]),
    'saicv_mixup_cutmix': (c_int, [c_int, _P, _P, _P, _P, _P, c_int, c_int, c_int, c_int, _P]),
    'saicv_soft_labels': (c_int, [_P, _P, ctypes.c_float, ctypes.c_float, _P, c_int, c_int, _P]),
    'saicv_detr_sine_pe': (c_int, [_P, _P, c_int, c_int, c_int, c_int, ctypes.c_float, ctypes.c_float, _P]),
    'saicv_detr_box_loss_fwd': (c_int, [_P, _P, _P, _P, _P, c_int, c_int, c_int, c_int, c_double, c_double, _P, _P]),
    'saicv_detr_box_loss_bwd': (c_int, [_P, _P, _P, _P, _P, _P, _P, _P, c_int, c_int, c_int, c_int, c_double, c_double, _P, _P]),
    'saicv_detr_assign': (c_int, [_P, _P, c_int, c_int, c_int, _P, _P, _P, _P]),
    'saicv_sam_prompt_tokens': (c_int, [_P, c_int, c_int, _P, _P, c_int, _P, ctypes.c_float, _P, _P, c_int, _P]),
    'saicv_sam_prompt_tokens_bwd': (c_int, [_P, _P, _P, c_int, c_int, _P]),
    'saicv_sam_grid_pe': (c_int, [_P, c_int, c_int, _P, _P]),
    'saicv_dwconv2d_fwd': (c_int, [c_int, _P, _P, _P, _P] + [c_int] * 10 + [_P]),
    'saicv_dwconv2d_dgrad': (c_int, [c_int, _P, _P, _P] + [c_int] * 10 + [_P]),
    'saicv_dwconv2d_wgrad': (c_int, [c_int, _P, _P, _P, _P] + [c_int] * 10 + [_P]),
    'saicv_act_fwd': (c_int, [c_int, c_int, c_double, _P, _P, c_size_t, _P]),
    'saicv_act_bwd': (c_int, [c_int, c_int, c_double, _P, _P, _P, c_size_t, _P]),
    'saicv_mul_fwd': (c_int, [c_int, _P, _P, _P, c_size_t, _P]),
    'saicv_mul_bwd': (c_int, [c_int, _P, _P, _P, _P, _P, c_size_t, _P]),
    'saicv_channel_scale_add_fwd': (c_int, [c_int, _P, _P, _P, _P, c_size_t, c_int, _P]),
    'saicv_channel_scale_add_bwd': (c_int, [c_int, _P, _P, _P, _P, _P, c_size_t, c_int, _P]),
    'saicv_bn_stats': (c_int, [c_int, _P, c_size_t, c_int, _P, _P, _P]),
    'saicv_resize_bilinear_add_fwd': (c_int, [c_int, c_int, _P, _P, _P, c_int, c_int, c_int, c_int, c_int, c_int, _P]),
    'saicv_resize_bilinear_bwd': (c_int, [c_int, _P, _P, c_int, c_int, c_int, c_int, c_int, c_int, _P]),
    'saicv_retina_assign': (c_int, [_P, _P, _P, _P, c_int, c_int, c_int, c_int, _P]),
    'saicv_focal_loss_level': (c_int, [_P, _P, _P, _P, c_int, c_int, c_int, c_int, c_int, c_double, c_double, _P]),
    'saicv_smoothl1_level': (c_int, [_P, _P, _P, _P, c_int, c_int, c_int, c_int, c_double, _P]),
    'saicv_fcos_assign': (c_int, [_P, _P, _P, _P, _P, c_int, c_int, c_int, c_double, c_int, _P]),
    'saicv_det_best_class': (c_int, [_P, _P, _P, _P, c_int, c_int, c_int, c_int, c_int, _P]),
    'saicv_groupnorm_ws_floats': (c_size_t, [c_int, c_int]),
    'saicv_groupnorm_fwd': (c_int, [c_int, _P, _P, _P, _P, _P, _P, _P, c_int, c_int, c_int, c_int, c_double, c_int, _P]),
    'saicv_groupnorm_bwd': (c_int, [c_int, _P, _P, _P, _P, _P, _P, _P, _P, _P, c_int, c_int, c_int, c_int, c_int, _P]),
    'saicv_sam_sample_point': (c_int, [c_int, _P, _P, c_long, _P, c_int, ctypes.c_float, ctypes.c_float, ctypes.c_uint, _P, _P,
                                       c_int, c_int, c_int, _P]),
    'saicv_sam_sample_point_dseed': (c_int, [c_int, _P, _P, c_long, _P, c_int, ctypes.c_float, ctypes.c_float, ctypes.c_uint, _P, _P, _P,
                                             c_int, c_int, c_int, _P]),
    'saicv_comm_available': (c_int, []),
    'saicv_comm_unique_id': (c_int, [_P]),
    'saicv_comm_reduce_scatter': (c_int, [_P, _P, _P, c_size_t, c_int, _P]),
    'saicv_comm_all_gather': (c_int, [_P, _P, _P, c_size_t, _P]),
    'saicv_comm_create': (c_int, [_P, c_int, c_int, POINTER(c_void_p)]),
    'saicv_comm_allreduce_bucket': (c_int, [_P, _P, c_size_t, c_int, _P]),
    'saicv_comm_broadcast': (c_int, [_P, _P, c_size_t, c_int, _P]),
    'saicv_comm_join': (c_int, [_P, _P]),
    'saicv_comm_stats': (c_int, [_P, POINTER(c_int), POINTER(c_int), POINTER(ctypes.c_ulonglong), POINTER(ctypes.c_ulonglong)]),
    'saicv_comm_destroy': (c_int, [_P]),
    'saicv_attention_stream_fwd': (c_int, [c_int, c_int, _PA, _P]),
    'saicv_attention_stream_bwd': (c_int, [c_int, c_int, _PA, _P]),
}

_lib = None


def available():
    return os.path.exists(LIB_PATH)


def lib():
    """Loads the library (once).  Raises if it has not been built -- there is no fallback."""
    global _lib
    if _lib is None:
        if not os.path.exists(LIB_PATH):
            raise RuntimeError(
                f'{LIB_PATH} is missing: run `python __graft_entry__.py` (build()) first. '
                'The MI355X HIP extension is required; there is no CPU/eager fallback.')
        handle = ctypes.CDLL(LIB_PATH)
        for name, (res, args) in SIGNATURES.items():
            try:
                fn = getattr(handle, name)
            except AttributeError:
                continue
            fn.restype = res
            fn.argtypes = args
        _lib = handle
    return _lib


def check(rc, what=''):
    if rc != 0:
        msg = lib().saicv_last_error_string()
        raise RuntimeError(f'libsaicv_hip {what} failed ({rc}): {msg.decode() if msg else "?"}')


def dtype_code(dt):
    if dt == torch.bfloat16:
        return BF16
    if dt == torch.float32:
        return F32
    raise TypeError(f'saicv kernels support bfloat16 and float32, got {dt}')


def epc(dt):
    return 8 if dt == torch.bfloat16 else 4


def stream():
    return torch.cuda.current_stream().cuda_stream


def ptr(t):
    return 0 if t is None else t.data_ptr()


def require_gpu(*tensors):
    for t in tensors:
        if t is not None and not t.is_cuda:
            raise RuntimeError('saicv HIP kernels need tensors on an MI355X device (got a CPU tensor); '
                               'there is no CPU fallback in the product path')
