"""ctypes binding of ``libcvx_engine.so`` (C ABI declared in ``include/cvx_engine.h``).

The product path has no fallback: if the HIP library is missing or a call fails, ``CvxError`` is raised.
"""
from __future__ import annotations

import ctypes as C
import os

_HERE = os.path.dirname(os.path.abspath(__file__))
LIB_PATH = os.environ.get("CVX_LIB") or os.path.join(_HERE, "lib", "libcvx_engine.so")   # CVX_LIB: A/B runs of two builds on one box
ABI_VERSION = 4


class CvxError(RuntimeError):
    pass


class BufDesc(C.Structure):
    _fields_ = [("h", C.c_int32), ("w", C.c_int32), ("c", C.c_int32), ("kind", C.c_int32)]


class View(C.Structure):
    _fields_ = [("buf", C.c_int32), ("coff", C.c_int32), ("c", C.c_int32), ("pix_off", C.c_int32)]


class OpDesc(C.Structure):
    _fields_ = [
        ("type", C.c_int32),
        ("in_", View), ("out", View), ("res", View),
        ("ih", C.c_int32), ("iw", C.c_int32), ("oh", C.c_int32), ("ow", C.c_int32),
        ("k", C.c_int32), ("stride", C.c_int32), ("pad", C.c_int32), ("dil", C.c_int32),
        ("act", C.c_int32), ("needs_dgrad", C.c_int32), ("w_cin", C.c_int32),
        ("w_off", C.c_int64), ("gamma_off", C.c_int64), ("beta_off", C.c_int64), ("bias_off", C.c_int64),
        ("rmean_off", C.c_int64), ("rvar_off", C.c_int64),
        ("lane", C.c_int32), ("flags", C.c_int32),
    ]


BUF_ACT_F16, BUF_PRED_F32 = 0, 1
OP_CONV, OP_MAXPOOL5, OP_UPSAMPLE2, OP_MAXPOOL2, OP_DWCONVT, OP_COPY, OP_MAXPOOL3S2, OP_AVGPOOL, OP_RESIZE, OP_MAXPOOL3S1, OP_L2NORM = 1, 2, 3, 4, 5, 6, 7, 8, 9, 10, 11
OP_DROPOUT = 12
ACT_BN_SILU, ACT_BIAS, ACT_BN_RELU, ACT_BN_LINEAR, ACT_BIAS_RELU, ACT_BIAS_LINEAR = 1, 2, 3, 4, 5, 6
OPF_RES_PRE_ACT = 1
OPF_CONV_BIAS = 2
OPF_RAW_F16 = 4

_P, _I32, _I64, _F, _U64 = C.c_void_p, C.c_int32, C.c_int64, C.c_float, C.c_uint64

# entry points of the tuning build only (include/cvx_engine_experimental.h): bound when the loaded library exports them
EXPERIMENTAL_PROTOTYPES = {
    "cvx_chain_pair_unit": (_I32, [_P, _I32, _I32, _I32, _I32, _P, _P, _P, _P, _P, _P, _I32, _P, _I32, _I32, _I32, C.POINTER(_F), _P]),
    "cvx_chain_conv_unit": (_I32, [_P, _I32, _I32, _I32, _I32, _P, _I32, _I32, _I32, _I32, _P, _P, _I32, _P, _I32, _I32, _I32, C.POINTER(_F), _P]),
    "cvx_chain_detect_unit": (_I32, [_P, _I32, _I32, _I32, _I32, _I32, _I32, _I32, _P, _P, _P, _P, _P, _P, _P, _P, _P, _P, _P, _I32, _I32, _I32,
                                     _I32, _I32, C.POINTER(_F), _P]),
    "cvx_wgrad_time_unit": (_I32, [_P, _P, _I32, _I32, _I32, _I32, _I32, _I32, _I32, _I32, _I32, _I32, _P, _I64, C.POINTER(_F), C.POINTER(_I32), _P]),
}

# name -> (restype, argtypes); every symbol include/cvx_engine.h declares
PROTOTYPES = {
    "cvx_last_error": (C.c_char_p, []),
    "cvx_abi_version": (_I32, []),
    "cvx_engine_create": (_I32, [C.POINTER(_P), C.POINTER(BufDesc), _I32, C.POINTER(OpDesc), _I32, _I32, _I32, _I32, _P]),
    "cvx_engine_destroy": (_I32, [_P]),
    "cvx_engine_bind": (_I32, [_P, _P, _P, _I64, _P, _I64]),
    "cvx_engine_set_bn": (_I32, [_P, _F, _F]),
    "cvx_engine_forward": (_I32, [_P, _P, _I32, _I32, _P]),
    "cvx_engine_backward": (_I32, [_P, _P, _F]),
    "cvx_engine_backward_begin": (_I32, [_P, _P, _F]),
    "cvx_engine_backward_range": (_I32, [_P, _I32, _I32]),
    "cvx_engine_grads_ready": (_I32, [_P, _I32, _I32, _P]),
    "cvx_engine_backward_end": (_I32, [_P]),
    "cvx_engine_workspace_bytes": (_I64, [_P]),
    "cvx_engine_plan_generation": (_I64, [_P]),
    "cvx_engine_debug_copy": (_I32, [_P, _I32, _I32, _P, _I64]),
    "cvx_engine_profile": (_I32, [_P, _I32]),
    "cvx_engine_profile_read": (_I32, [_P, _I32, C.POINTER(C.c_double), C.POINTER(C.c_double), C.POINTER(C.c_double),
                                       C.POINTER(_I64)]),
    "cvx_debug_clock_buffer": (_I32, [_P]),
    "cvx_debug_conv_tile_plan": (_I32, [_I32, _I32, _I32, _I32, _I32, _P]),
    "cvx_engine_profile_dump": (_I32, [_P, C.c_char_p]),
    "cvx_pred_level_to_nchw": (_I32, [_P, _I32, _I32, _I32, _I32, _I32, _I32, _P, _P]),
    "cvx_nchw_grad_to_dpred": (_I32, [_P, _I32, _I32, _I32, _I32, _I32, _I32, _F, _P, _P]),
    "cvx_loss_v8_workspace_bytes": (_I64, [_I32, _I32, _I32, _I32]),
    "cvx_loss_v8": (_I32, [_P, _I32, _I32, _I32, _P, _I32, _I32, C.POINTER(_I32), C.POINTER(_F), _I32, _F, _F, _F, _F, _P, _P, _P,
                           _I64, _P]),
    "cvx_loss_v8_strided": (_I32, [_P, _I32, _I32, _I32, _I32, _P, _I32, _I32, C.POINTER(_I32), C.POINTER(_F), _I32, _F, _F, _F, _F, _P, _P, _P,
                                   _I64, _P]),
    "cvx_loss_v8_assignment": (_I32, [_P, _I32, _I32, _I32, _P, _P, _P]),
    "cvx_pack_targets": (_I32, [_P, _P, _P, _I32, _P, _P]),
    "cvx_decode_strided": (_I32, [_P, _I32, _I32, _I32, _I32, C.POINTER(_I32), C.POINTER(_F), _I32, _P, _P]),
    "cvx_adam_step": (_I32, [_P, _P, _P, _P, _I64, _F, _F, _F, _F, _I32, _P, _I32, _P]),
    "cvx_check_finite": (_I32, [_P, _I64, _P, _P]),
    "cvx_adam_step_dev": (_I32, [_P, _P, _P, _P, _I64, _F, _F, _F, _P, _P, _I32, _F, _P]),
    "cvx_engine_set_stream": (_I32, [_P, _P]),
    "cvx_engine_exchange_stream": (_P, [_P]),
    "cvx_decode": (_I32, [_P, _I32, _I32, _I32, C.POINTER(_I32), C.POINTER(_F), _I32, _P, _P]),
    "cvx_yolo7_decode": (_I32, [_P, _I32, _I32, _I32, _P, _P, _I32, _I32, _I32, _P, _P, _P]),
    "cvx_ssd_decode": (_I32, [_P, _P, _P, _I32, _I32, _I32, _F, _F, _P, _P, _P]),
    "cvx_ssd_decode_max": (_I32, [_P, _P, _P, _I32, _I32, _I32, _F, _F, _P, _P, _P, _P]),
    "cvx_pred_cols_to_nchw": (_I32, [_P, _I32, _I32, _I32, _I32, _I32, _I32, _I32, _P, _I64, _I64, _P]),
    "cvx_nms_workspace_bytes": (_I64, [_I32, _I32]),
    "cvx_nms": (_I32, [_P, _I32, _I32, _I32, _F, _F, _I32, _P, _P, _P, _P, _I64, _P]),
    "cvx_nms_variant": (_I32, [_P, _I32, _I32, _I32, _F, _F, _I32, _I32, _P, _P, _P, _P, _I64, _P]),
    "cvx_centernet_decode_workspace_bytes": (_I64, [_I32, _I32, _I32, _I32]),
    "cvx_centernet_decode": (_I32, [_P, _I32, _I32, _I32, _I32, _I32, _I32, _I32, _I32, _F, _F, _I32, _P, _P, _P, _P, _P, _P, _P, _I64, _P]),
    "cvx_conv2d_nhwc": (_I32, [_P, _I32, _I32, _I32, _I32, _P, _I32, _I32, _I32, _I32, _I32, _I32, _P, _P, _P, _P]),
    "cvx_conv2d_dgrad_nhwc": (_I32, [_P, _I32, _I32, _I32, _I32, _P, _I32, _I32, _I32, _I32, _I32, _P, _P]),
    "cvx_conv2d_wgrad_workspace_bytes": (_I64, [_I32, _I32, _I32, _I32, _I32, _I32]),
    "cvx_conv2d_wgrad_nhwc": (_I32, [_P, _P, _I32, _I32, _I32, _I32, _I32, _I32, _I32, _I32, _I32, _P, _P, _I64, _P]),
    "cvx_bn_silu_train_nhwc": (_I32, [_P, _I32, _I32, _I32, _P, _P, _F, _F, _P, _P, _P, _P, _P, _P, _P, _P]),
    "cvx_bn_silu_bwd_nhwc": (_I32, [_P, _P, _I32, _I32, _I32, _P, _P, _P, _F, _P, _P, _P, _P, _I32, _P]),
    "cvx_letterbox_geometry": (_I32, [_I32, _I32, _I32, _I32, _P, _P, _P, _P, _P]),
    "cvx_letterbox_u8_to_nchw": (_I32, [_P, _I32, _I32, _I32, _I32, _P, _I32, _I32, _P]),
    "cvx_engine_set_seed": (_I32, [_P, _U64]),
    "cvx_engine_keep_shadows": (_I32, [_P]),
    "cvx_engine_set_fusion": (_I32, [_P, _I32]),
    "cvx_engine_fused_groups": (_I32, [_P]),
    "cvx_centernet_loss_workspace_bytes": (_I64, [_I32, _I32]),
    "cvx_centernet_loss": (_I32, [_P, _I32, _I32, _I32, _I32, _I32, _I32, _P, _P, _P, _P, _P, _I32, _F, _F, _F, _F, _P, _P, _P, _P, _P]),
    "cvx_yolo7_loss_workspace_bytes": (_I64, [_I32, _I32, _I32, _I32]),
    "cvx_yolo7_loss": (_I32, [_P, _I32, _I32, _I32, _P, _P, _P, _P, _I32, _F, _F, _F, _F, _F, _F, _P, _P, _P, _P, _P]),
    "cvx_centernet_draw_targets": (_I32, [_P, _P, _I32, _I32, _I32, _I32, _I32, _P, _P, _P, _P, _P, _P]),
    "cvx_ssd_encode_targets": (_I32, [_P, _P, _I32, _I32, _P, _I32, _I32, _F, _F, _F, _P, _P, _P]),
    "cvx_multibox_loss_workspace_bytes": (_I64, [_I32, _I32]),
    "cvx_multibox_loss": (_I32, [_P, _P, _P, _I32, _I32, _I32, _F, _F, _F, _P, _P, _P, _P, _P]),
    "cvx_seg_loss_workspace_bytes": (_I64, [_I32, _I32, _I32, _I32]),
    "cvx_seg_loss": (_I32, [_P, _I32, _I32, _I32, _I32, _I32, _I32, _I32, _P, _I32, _F, _F, _I64, _F, _P, _P, _P, _P, _P]),
    "cvx_resize_bilinear_nchw_grad_to_rows": (_I32, [_P, _I32, _I32, _I32, _I32, _I32, _I32, _F, _P, _I32, _P]),
    "cvx_maxpool3_train_nhwc": (_I32, [_P, _I32, _I32, _I32, _I32, _I32, _P, _P, _P]),
    "cvx_maxpool3_bwd_nhwc": (_I32, [_P, _P, _I32, _I32, _I32, _I32, _I32, _P, _I32, _P]),
    "cvx_l2norm_bwd_nhwc": (_I32, [_P, _P, _P, _I32, _I32, _I32, _F, _P, _P, _I32, _P]),
    "cvx_nchw_cols_grad_to_pred": (_I32, [_P, _I64, _I64, _I32, _I32, _I32, _I32, _I32, _F, _P, _I32, _I32, _P]),
    "cvx_dwconvt_nhwc": (_I32, [_P, _I32, _I32, _I32, _I32, _I32, _P, _P, _P]),
    "cvx_dwconvt_bwd_nhwc": (_I32, [_P, _P, _I32, _I32, _I32, _I32, _I32, _P, _P, _I32, _P, _F, _P]),
    "cvx_maxpool2_bwd_nhwc": (_I32, [_P, _P, _I32, _I32, _I32, _I32, _I32, _P, _I32, _P]),
    "cvx_avgpool_global_bwd_nhwc": (_I32, [_P, _I32, _I32, _I32, _P, _I32, _P]),
    "cvx_resize_bilinear_bwd_nhwc": (_I32, [_P, _I32, _I32, _I32, _I32, _I32, _I32, _P, _I32, _P]),
    "cvx_dropout_nhwc": (_I32, [_P, _I32, _I32, _I32, _F, _U64, _P, _I32, _P]),
    "cvx_bn_act_train_nhwc": (_I32, [_P, _I32, _I32, _I32, _P, _P, _F, _F, _P, _P, _P, _I32, _I32, _P, _P, _P, _P, _P]),
    "cvx_bn_act_bwd_nhwc": (_I32, [_P, _P, _P, _I32, _I32, _I32, _P, _P, _P, _I32, _I32, _F, _P, _P, _P, _P, _I32, _P]),
    "cvx_maxpool5_nhwc": (_I32, [_P, _I32, _I32, _I32, _I32, _P, _P, _P]),
    "cvx_maxpool5_bwd_nhwc": (_I32, [_P, _P, _I32, _I32, _I32, _I32, _P, _I32, _P]),
    "cvx_resize_bilinear_rows_to_nchw": (_I32, [_P, _I32, _I32, _I32, _I32, _I32, _I32, _I32, _P, _P]),
    "cvx_maxpool_nhwc": (_I32, [_P, _I32, _I32, _I32, _I32, _I32, _I32, _I32, _P, _P]),
    "cvx_avgpool_global_nhwc": (_I32, [_P, _I32, _I32, _I32, _P, _P]),
    "cvx_resize_bilinear_nhwc": (_I32, [_P, _I32, _I32, _I32, _I32, _I32, _I32, _P, _P]),
    "cvx_l2norm_nhwc": (_I32, [_P, _P, _I32, _I32, _I32, _P, _P]),
    "cvx_upsample2_nhwc": (_I32, [_P, _I32, _I32, _I32, _I32, _P, _P]),
    "cvx_upsample2_bwd_nhwc": (_I32, [_P, _I32, _I32, _I32, _I32, _P, _I32, _P]),
    "cvx_stem_train_nchw": (_I32, [_P, _I32, _I32, _I32, _P, _I32, _P, _P, _F, _F, _P, _P, _P, _P, _P, _P, _P]),
    "cvx_stem_eval_nchw": (_I32, [_P, _I32, _I32, _I32, _P, _I32, _P, _P, _P, _P]),
    "cvx_stem_wgrad_nchw": (_I32, [_P, _I32, _I32, _I32, _P, _I32, _P, _P]),
    "cvx_comm_unique_id": (_I32, [_P]),
    "cvx_comm_create": (_I32, [C.POINTER(_P), _P, _I32, _I32, _I32]),
    "cvx_comm_destroy": (_I32, [_P]),
    "cvx_allreduce_f32": (_I32, [_P, _I64, _P, _P]),
    "cvx_allreduce_grads": (_I32, [_P, _P, _P]),
    "cvx_engine_backward_exchange": (_I32, [_P, _P, _F, _P, _P, _I32, _P]),
    "cvx_stem_backward_nchw": (_I32, [_P, _I32, _I32, _I32, _P, _P, _I32, _P, _P, _P, _F, _P, _P, _P, _P]),
    "cvx_stem_backward_recompute_nchw": (_I32, [_P, _I32, _I32, _I32, _P, _P, _I32, _P, _P, _P, _P, _F, _P, _P, _P, _P]),
}

_lib = None


def load():
    """Loads the shared library once; raises CvxError (never falls back) if it is absent."""
    global _lib
    if _lib is not None:
        return _lib
    if not os.path.exists(LIB_PATH):
        raise CvxError(f"{LIB_PATH} not found: build it with `python -c 'import __graft_entry__ as g; g.build()'` "
                       "(hipcc --offload-arch=gfx950). There is no CPU fallback.")
    lib = C.CDLL(LIB_PATH)
    for name, (res, args) in PROTOTYPES.items():
        fn = getattr(lib, name)          # AttributeError here = header/library mismatch
        fn.restype = res
        fn.argtypes = args
    for name, (res, args) in EXPERIMENTAL_PROTOTYPES.items():
        fn = getattr(lib, name, None)
        if fn is not None:
            fn.restype = res
            fn.argtypes = args
    if lib.cvx_abi_version() != ABI_VERSION:
        raise CvxError(f"ABI mismatch: library {lib.cvx_abi_version()} != binding {ABI_VERSION}")
    _lib = lib
    return lib


def has_chain() -> bool:
    """True when the loaded library is a tuning build that carries the chain kernel (tools/build_tuning.sh, CVX_LIB=...)."""
    return hasattr(load(), "cvx_chain_pair_unit")


def check(rc: int, what: str = ""):
    if rc != 0:
        msg = load().cvx_last_error()
        raise CvxError(f"{what}: {msg.decode() if msg else 'unknown error'}")


def ptr(t):
    """Raw device/host pointer of a torch tensor (0 -> NULL for None)."""
    return C.c_void_p(0 if t is None else t.data_ptr())


def stream_ptr(device=None):
    import torch
    return C.c_void_p(torch.cuda.current_stream(device).cuda_stream)
