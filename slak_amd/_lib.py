"""ctypes binding of libslak_hip.so (C ABI in include/slak_hip.h).

The product path has NO fallback: if the HIP library is missing or a call fails, this raises.
(The reference binds its native module with ``import _depthwise_conv2d_implicit_gemm_C``:
depthwise_conv2d_implicit_gemm.py:8.)
"""
import ctypes
import os

_HERE = os.path.dirname(os.path.abspath(__file__))
LIB_PATH = os.path.join(_HERE, "lib", "libslak_hip.so")

SLAK_F32, SLAK_F16, SLAK_BF16 = 0, 1, 2
ALGO_AUTO, ALGO_DIRECT, ALGO_MFMA = 0, 1, 2
OP_FWD, OP_BWD_DATA, OP_BWD_FILTER = 0, 1, 2
OK, ERR_INVALID_ARG, ERR_UNSUPPORTED = 0, 1, 2

_lib = None


class SlakHipError(RuntimeError):
    pass


class MaskSegment(ctypes.Structure):
    _fields_ = [("weight", ctypes.c_void_p), ("mask", ctypes.c_void_p), ("grad", ctypes.c_void_p),
                ("momentum", ctypes.c_void_p), ("numel", ctypes.c_longlong)]


class AdamwSegment(ctypes.Structure):
    _fields_ = [("param", ctypes.c_void_p), ("exp_avg", ctypes.c_void_p), ("exp_avg_sq", ctypes.c_void_p), ("mask", ctypes.c_void_p),
                ("param_bf16", ctypes.c_void_p), ("step", ctypes.c_void_p), ("numel", ctypes.c_longlong), ("group", ctypes.c_int)]


class AdamwGroup(ctypes.Structure):
    _fields_ = [("lr", ctypes.c_double), ("beta1", ctypes.c_double), ("beta2", ctypes.c_double), ("eps", ctypes.c_double),
                ("weight_decay", ctypes.c_double)]


ADAMW_MAX_GROUPS = 64


class EmaSegment(ctypes.Structure):
    _fields_ = [("ema", ctypes.c_void_p), ("model", ctypes.c_void_p), ("mask", ctypes.c_void_p), ("numel", ctypes.c_longlong),
                ("dtype", ctypes.c_int)]


# name -> (restype, argtypes): every symbol include/slak_hip.h declares (tests/test_boundary.py checks this)
_vp, _i, _sz, _d = ctypes.c_void_p, ctypes.c_int, ctypes.c_size_t, ctypes.c_double
_CONV = [_vp, _i, _vp, _i, _vp, _i, _i, _i, _i, _i, _i, _i, _vp, _sz, _vp]
SIGNATURES = {
    "slak_status_string": (ctypes.c_char_p, [_i]),
    "slak_last_hip_error": (ctypes.c_char_p, []),
    "slak_version": (_i, []),
    "slak_device_info": (_i, [ctypes.POINTER(_i), ctypes.POINTER(_i), ctypes.c_char_p, _sz]),
    "slak_set_conv_algo": (_i, [_i]),
    "slak_set_fp32_matrix_cores": (_i, [_i]),
    "slak_get_fp32_matrix_cores": (_i, []),
    "slak_get_fp32_matrix_cores_effective": (_i, []),
    "slak_set_fp32_matrix_cores_thread": (_i, [_i, ctypes.POINTER(ctypes.c_int)]),
    "slak_dwconv2d_workspace_bytes": (_sz, [_i, _i, _i, _i, _i, _i, _i, _i]),
    "slak_dwconv2d_forward": (_i, _CONV),
    "slak_dwconv2d_backward_data": (_i, _CONV),
    "slak_dwconv2d_backward_data_accumulate": (_i, _CONV),
    "slak_dwconv2d_backward_filter": (_i, [_vp, _i, _vp, _i, _vp, _i, _i, _i, _i, _i, _i, _vp, _sz, _vp]),
    "slak_mask_plan_create": (_i, [ctypes.POINTER(MaskSegment), _i, ctypes.POINTER(_vp)]),
    "slak_mask_plan_set_grads": (_i, [_vp, ctypes.POINTER(_vp), _vp]),
    "slak_mask_plan_set_momentum": (_i, [_vp, ctypes.POINTER(_vp), _vp]),
    "slak_mask_plan_destroy": (_i, [_vp]),
    "slak_mask_apply": (_i, [_vp, _vp]),
    "slak_mask_prune_and_grow": (_i, [_vp, _d, _vp]),
    "slak_mask_prune": (_i, [_vp, _d, _vp]),
    "slak_mask_read_stats": (_i, [_vp, ctypes.POINTER(_d), _vp]),
    "slak_mask_checksum": (_i, [_vp, ctypes.POINTER(ctypes.c_ulonglong), _vp]),
    "slak_adamw_plan_create": (_i, [ctypes.POINTER(AdamwSegment), _i, ctypes.POINTER(_vp)]),
    "slak_adamw_step": (_i, [_vp, _vp, ctypes.POINTER(AdamwGroup), _i, _vp]),
    "slak_adamw_plan_destroy": (_i, [_vp]),
    "slak_ema_plan_create": (_i, [ctypes.POINTER(EmaSegment), _i, ctypes.POINTER(_vp)]),
    "slak_ema_update": (_i, [_vp, _d, _vp]),
    "slak_ema_plan_destroy": (_i, [_vp]),
    "slak_bn3_workspace_bytes": (_sz, [_i, _i]),
    "slak_bn3_forward_sums": (_i, [_vp, _vp, _vp, _vp, _i, _i, _i, _vp, _sz, _vp, ctypes.POINTER(_vp), ctypes.POINTER(_i), _i]),
    "slak_bn3_forward_apply": (_i, [_vp, _vp, _vp, _vp, _d, _vp, ctypes.POINTER(_vp), ctypes.POINTER(_vp), ctypes.POINTER(_vp), ctypes.POINTER(_vp),
                                    ctypes.c_float, ctypes.c_float, _i, _i, _vp, _vp, _vp, _i, _i, _i, _vp]),
    "slak_bn3_backward_sums": (_i, [_vp, _vp, _vp, _vp, _vp, _vp, _i, _i, _i, _vp, _sz, _vp]),
    "slak_bn3_backward_apply": (_i, [_vp, _vp, _vp, _vp, _vp, _vp, _d, _vp, _vp, ctypes.POINTER(_vp), _vp, _vp, _vp, _vp, _vp, _vp, _i, _i, _i, _vp]),
    "slak_bn3_forward_local": (_i, [_vp, _vp, _vp, ctypes.POINTER(_vp), ctypes.POINTER(_vp), ctypes.POINTER(_vp), ctypes.POINTER(_vp),
                                    ctypes.c_float, ctypes.c_float, _i, _vp, _vp, _vp, _i, _i, _i, _vp, _sz, _vp, ctypes.POINTER(_vp), ctypes.POINTER(_i), _i]),
    "slak_dwconv2d_forward_stats": (_i, [_vp, _i, _vp, _i, _vp, _i, _vp, _i, ctypes.POINTER(_i), _i, _i, _i, _i, _i, _i, _vp]),
    "slak_dwconv2d_tri_stats_rows": (_i, [_i, _i, _i, _i, _i, _i]),
    "slak_dwconv2d_tri_forward_stats": (_i, [_vp, _vp, _vp, _vp, _vp, _vp, _vp, _vp, _i, _i, _i, _i, _i, _i, _vp]),
    "slak_bn3_backward_local": (_i, [_vp, _vp, _vp, _vp, _vp, ctypes.POINTER(_vp), _vp, _vp, _vp, _vp, _vp, _vp, _i, _i, _i, _vp, _sz, _vp]),
    "slak_bn3_forward_sums_counted": (_i, [_vp, _vp, _vp, _vp, _i, _i, _i, _vp, _sz, _vp, ctypes.POINTER(_vp), ctypes.POINTER(_i), _i]),
    "slak_bn3_backward_sums_dup": (_i, [_vp, _vp, _vp, _vp, _vp, _vp, _vp, _i, _i, _i, _vp, _sz, _vp]),
    "slak_pack_w1t_fragments": (_i, [_vp, _vp, _i, _i, _vp]),
    "slak_bn3_backward_local_to": (_i, [_vp, _vp, _vp, _vp, _vp, ctypes.POINTER(_vp), _vp, ctypes.POINTER(_vp), ctypes.POINTER(_vp), _vp, _vp, _vp, _i, _i, _i, _vp, _sz, _vp]),
    "slak_bn3_backward_apply_to": (_i, [_vp, _vp, _vp, _vp, _vp, _vp, _d, _vp, _vp, ctypes.POINTER(_vp), _vp, ctypes.POINTER(_vp), ctypes.POINTER(_vp), _vp, _vp, _vp, _i, _i, _i, _vp]),
    "slak_block_tail_workspace_bytes": (_sz, [_i, _i, _i]),
    "slak_gelu_bwd_workspace_bytes": (_sz, [_i, _i]),
    "slak_gelu_backward_bias": (_i, [_vp, _vp, _vp, _vp, _i, _i, _vp, _sz, _vp]),
    "slak_ln_cf_workspace_bytes": (_sz, [_i, _i, _i]),
    "slak_ln_channels_first_forward": (_i, [_vp, _i, _vp, _vp, _vp, _i, _vp, _vp, _i, _i, _i, ctypes.c_float, _vp]),
    "slak_ln_channels_first_backward": (_i, [_vp, _i, _vp, _i, _vp, _vp, _vp, _vp, _vp, _vp, _i, _i, _i, _vp, _sz, _vp]),
    "slak_ln_channels_first_forward_pair": (_i, [_vp, _i, _vp, _vp, _vp, _i, _vp, _vp, _vp, _i, _i, _i, ctypes.c_float, _vp]),
    "slak_ln_channels_first_backward_pair_supported": (_i, [_i, _i, _i, _i, _i]),
    "slak_ln_channels_first_backward_pair": (_i, [_vp, _i, _vp, _vp, _i, _vp, _vp, _vp, _vp, _vp, _vp, _i, _i, _i, _vp, _sz, _vp]),
    "slak_ln_nchw_to_nhwc_forward": (_i, [_vp, _vp, _vp, _vp, _vp, _vp, _i, _i, _i, ctypes.c_float, _vp]),
    "slak_ln_nchw_to_nhwc_backward": (_i, [_vp, _vp, _vp, _vp, _vp, _vp, _vp, _vp, _i, _i, _i, _vp, _sz, _vp]),
    "slak_dwconv2d_tri_supported": (_i, [_i, _i, _i, _i, _i, _i]),
    "slak_dwconv2d_tri_supported_op": (_i, [_i, _i, _i, _i, _i, _i, _i]),
    "slak_debug_last_kernel": (ctypes.c_char_p, []),
    "slak_debug_marker": (_i, [_i, _vp]),
    "slak_dwconv2d_tri_forward": (_i, [_vp, _vp, _vp, _vp, _vp, _vp, _vp, _i, _i, _i, _i, _i, _i, _vp]),
    "slak_dwconv2d_tri_backward_data": (_i, [_vp, _vp, _vp, _vp, _vp, _vp, _vp, _i, _i, _i, _i, _i, _i, _vp]),
    "slak_dwconv2d_tri_filter_workspace_bytes": (_sz, [_i, _i, _i, _i, _i, _i]),
    "slak_dwconv2d_tri_backward_supported": (_i, [_i, _i, _i, _i, _i, _i]),
    "slak_dwconv2d_tri_backward": (_i, [_vp] * 11 + [_i] * 6 + [_vp, _sz, _vp]),
    "slak_dwconv2d_pair_filter_workspace_bytes": (_sz, [_i, _i, _i, _i, _i, _i]),
    "slak_dwconv2d_pair_backward_filter": (_i, [_vp, _vp, _vp, _vp, _vp, _i, _i, _i, _i, _i, _i, _vp, _sz, _vp]),
    "slak_dwconv2d_tri_backward_filter": (_i, [_vp, _vp, _vp, _vp, _vp, _vp, _vp, _i, _i, _i, _i, _i, _i, _vp, _sz, _vp]),
    "slak_stem_patchify": (_i, [_vp, _vp, _i, _i, _i, _i, _vp]),
    "slak_stem_conv_forward_supported": (_i, [_i, _i, _i, _i, _i]),
    "slak_stem_conv_forward": (_i, [_vp, _vp, _vp, _vp, _vp, _i, _i, _i, _i, _i, _vp]),
    "slak_stem_wgrad_supported": (_i, [_i, _i, _i, _i]),
    "slak_stem_wgrad_workspace_bytes": (_sz, [_i, _i, _i, _i]),
    "slak_stem_wgrad": (_i, [_vp, _vp, _vp, _vp, _i, _i, _i, _i, _vp, _sz, _vp]),
    "slak_fill_channel_bias_bf16": (_i, [_vp, _vp, _i, _i, _i, _vp]),
    "slak_nchw_to_pixel_major_bf16": (_i, [_vp, _vp, _i, _i, _i, _vp]),
    "slak_channel_sums_workspace_bytes": (_sz, [_i]),
    "slak_channel_sums_bf16": (_i, [_vp, _vp, _i, _i, _i, _vp, _sz, _vp]),
    "slak_ln_patch_supported": (_i, [_i, _i, _i, _i]),
    "slak_ln_patch_forward": (_i, [_vp, _vp, _vp, _vp, _vp, _vp, _i, _i, _i, _i, ctypes.c_float, _vp]),
    "slak_ln_patch_backward": (_i, [_vp, _vp, _vp, _vp, _vp, _vp, _vp, _vp, _i, _i, _i, _i, _vp, _sz, _vp]),
    "slak_linear_nt_supported": (_i, [_i, _i, _i, _i]),
    "slak_linear_nt": (_i, [_vp, _vp, _vp, _vp, _vp, _i, _i, _i, _vp]),
    "slak_linear_mlp_fwd_supported": (_i, [_i, _i, _i]),
    "slak_linear_mlp_fwd": (_i, [_vp] * 8 + [_i, _i, _i, _vp]),
    "slak_defer_reductions_begin": (_i, []),
    "slak_defer_reductions_end": (_i, []),
    "slak_linear_nt_gelu_bwd_supported": (_i, [_i, _i, _i]),
    "slak_linear_nt_gelu_bwd_workspace_bytes": (_sz, [_i, _i, _i]),
    "slak_linear_nt_gelu_bwd": (_i, [_vp, _vp, _vp, _vp, _vp, _i, _i, _i, _vp, _sz, _vp]),
    "slak_linear_nt_gelu_bwd_dt_supported": (_i, [_i, _i, _i]),
    "slak_linear_nt_gelu_bwd_dt": (_i, [_vp, _vp, _vp, _vp, _vp, _vp, _vp, _i, _i, _i, _vp, _sz, _vp]),
    "slak_linear_gemm_supported": (_i, [_i, _i, _i, _i]),
    "slak_linear_gemm_workspace_bytes": (_sz, [_i, _i, _i, _i]),
    "slak_linear_gemm": (_i, [_vp] * 7 + [_i, _i, _i, _i, _vp, _sz, _vp]),
    "slak_transpose_bf16_batch": (_i, [_vp, _vp, _vp, _vp, _i, _vp]),
    "slak_linear_wgrad_supported": (_i, [_i, _i, _i]),
    "slak_linear_wgrad_workspace_bytes": (_sz, [_i, _i, _i]),
    "slak_linear_wgrad": (_i, [_vp, _vp, _vp, _i, _i, _i, _vp, _sz, _vp]),
    "slak_scale_residual_forward": (_i, [_vp, _i, _vp, _vp, _vp, _vp, _vp, _i, _i, _i, _vp]),
    "slak_scale_residual_backward": (_i, [_vp, _vp, _vp, _vp, _vp, _vp, _vp, _vp, _vp, _i, _i, _i, _vp, _sz, _vp]),
}


def lib():
    """Load the library once; raise (never fall back) if it is not built."""
    global _lib
    if _lib is None:
        if not os.path.exists(LIB_PATH):
            raise SlakHipError(
                "libslak_hip.so is not built (%s). Run `python -m slak_amd.build` "
                "(or __graft_entry__.build()); there is no CPU fallback." % LIB_PATH)
        L = ctypes.CDLL(LIB_PATH)
        for name, (res, args) in SIGNATURES.items():
            fn = getattr(L, name)
            fn.restype, fn.argtypes = res, args
        _lib = L
    return _lib


def check(status, what=""):
    if status != 0:
        L = lib()
        msg = L.slak_status_string(status).decode()
        if status == 4:
            msg += ": " + L.slak_last_hip_error().decode()
        raise SlakHipError("%s failed: %s (status %d)" % (what or "libslak_hip call", msg, status))
