"""ctypes binding of include/rsrgan.h (librsrgan_hip.so).  Fails loudly when the library is missing."""
from __future__ import annotations

import ctypes as C
import os

_HERE = os.path.dirname(os.path.abspath(__file__))
LIB_PATH = os.path.join(_HERE, "lib", "librsrgan_hip.so")

G_TYPES = {"lstm": 0, "res_lstm_l": 1, "res_lstm_base": 2, "dnn": 3, "rced": 4}
FLAG_BATCH_NORM = 32          # include/rsrgan.h RSRGAN_FLAG_BATCH_NORM
D_TYPES = {"lstm": 0, "dnn": 1}
NET_G, NET_D = 0, 1
SCALARS = {"g_learning_rate": 0, "d_learning_rate": 1, "mse_lambda": 2, "d_real": 3, "d_fake": 4,
           "l2_scale": 5, "clip_norm": 6, "adam_step": 7, "adam_step_d": 8}
WHAT = {"variables": 0, "adam_m": 1, "adam_v": 2, "ema": 3}

# every symbol include/rsrgan.h declares (checked by tests/test_capi_symbols.py)
SYMBOLS = ["rsrgan_default_cfg", "rsrgan_create", "rsrgan_destroy", "rsrgan_last_error", "rsrgan_set_scalar",
           "rsrgan_get_scalar", "rsrgan_num_tensors", "rsrgan_tensor_info", "rsrgan_param_count",
           "rsrgan_get_params", "rsrgan_set_params", "rsrgan_get_grads", "rsrgan_forward_g", "rsrgan_d_step",
           "rsrgan_g_step", "rsrgan_d_backward", "rsrgan_g_backward", "rsrgan_apply", "rsrgan_grad_buffer",
           "rsrgan_grad_bucket_count", "rsrgan_grad_bucket_info", "rsrgan_grad_bucket_wait",
           "rsrgan_profile_begin", "rsrgan_profile_read", "rsrgan_profile_read_kind", "rsrgan_profile_launches", "rsrgan_op_launch_floor", "rsrgan_device_status", "rsrgan_set_dropout",
           "rsrgan_op_gemm", "rsrgan_version",
           "rsrgan_segan_default_cfg", "rsrgan_segan_create", "rsrgan_segan_destroy", "rsrgan_segan_set_scalar",
           "rsrgan_segan_num_tensors", "rsrgan_segan_tensor_info", "rsrgan_segan_param_count", "rsrgan_segan_get_params",
           "rsrgan_segan_set_params", "rsrgan_segan_forward_g", "rsrgan_segan_d_backward", "rsrgan_segan_g_backward",
           "rsrgan_segan_grad_buffer", "rsrgan_segan_apply"]


class RsrganCfg(C.Structure):
    _fields_ = [("batch_size", C.c_int32), ("max_frames", C.c_int32), ("input_dim", C.c_int32),
                ("output_dim", C.c_int32), ("g_type", C.c_int32), ("g_layers", C.c_int32), ("g_cells", C.c_int32),
                ("g_proj", C.c_int32), ("d_type", C.c_int32), ("d_layers", C.c_int32), ("d_cells", C.c_int32),
                ("d_proj", C.c_int32), ("l2_scale", C.c_float), ("clip_norm", C.c_float), ("adam_beta1", C.c_float),
                ("adam_beta2", C.c_float), ("adam_eps", C.c_float), ("ema_decay", C.c_float),
                ("lrelu_alpha", C.c_float), ("forget_bias", C.c_float), ("cross_validation", C.c_int32),
                ("flags", C.c_int32), ("d_joint_off", C.c_int32), ("d_joint_dim", C.c_int32), ("g_splice", C.c_int32)]


class SeganCfg(C.Structure):
    """include/rsrgan.h rsrgan_segan_cfg"""
    _fields_ = [("batch_size", C.c_int32), ("input_len", C.c_int32), ("output_dim", C.c_int32), ("n_layers", C.c_int32),
                ("g_depths", C.c_int32 * 16), ("d_depths", C.c_int32 * 16), ("g_kwidth", C.c_int32), ("d_kwidth", C.c_int32),
                ("g_prelu", C.c_int32), ("lrelu_alpha", C.c_float), ("vbn_eps", C.c_float), ("rms_decay", C.c_float),
                ("rms_eps", C.c_float)]


class RsrganError(RuntimeError):
    pass


_lib = None


def load():
    """dlopen librsrgan_hip.so; raise (never fall back) if it is not built."""
    global _lib
    if _lib is not None:
        return _lib
    if not os.path.exists(LIB_PATH):
        raise ImportError("%s is missing: build it with `python -c 'import __graft_entry__ as g; g.build()'` "
                          "(hipcc --offload-arch=gfx950); rsrgan_amd has no CPU fallback" % LIB_PATH)
    lib = C.CDLL(LIB_PATH)
    p, i32, i64, f32, vp = C.c_void_p, C.c_int32, C.c_int64, C.c_float, C.c_void_p
    lib.rsrgan_last_error.restype = C.c_char_p
    lib.rsrgan_default_cfg.argtypes = [i32, C.POINTER(RsrganCfg)]
    lib.rsrgan_create.argtypes = [C.POINTER(RsrganCfg), C.c_uint64, C.POINTER(vp)]
    lib.rsrgan_destroy.argtypes = [vp]
    lib.rsrgan_set_scalar.argtypes = [vp, i32, C.c_double]
    lib.rsrgan_get_scalar.argtypes = [vp, i32, C.POINTER(C.c_double)]
    lib.rsrgan_num_tensors.argtypes = [vp, i32]
    lib.rsrgan_tensor_info.argtypes = [vp, i32, i32, C.c_char_p, i32, C.POINTER(i32), C.POINTER(i32), C.POINTER(i64)]
    lib.rsrgan_param_count.argtypes = [vp, i32]
    lib.rsrgan_param_count.restype = i64
    lib.rsrgan_get_params.argtypes = [vp, i32, i32, p, vp]
    lib.rsrgan_set_params.argtypes = [vp, i32, i32, p, vp]
    lib.rsrgan_get_grads.argtypes = [vp, i32, p, vp]
    lib.rsrgan_forward_g.argtypes = [vp, p, p, i32, p, vp]
    lib.rsrgan_d_step.argtypes = [vp, p, p, p, i32, p, p, p, i32, vp]
    lib.rsrgan_g_step.argtypes = [vp, p, p, p, i32, p, p, i32, i32, vp]
    lib.rsrgan_d_backward.argtypes = [vp, p, p, p, i32, p, p, p, vp]
    lib.rsrgan_g_backward.argtypes = [vp, p, p, p, i32, p, p, i32, vp]
    lib.rsrgan_apply.argtypes = [vp, i32, vp]
    lib.rsrgan_grad_buffer.argtypes = [vp, i32, C.POINTER(p), C.POINTER(i64)]
    lib.rsrgan_grad_bucket_count.argtypes = [vp, i32]
    lib.rsrgan_grad_bucket_info.argtypes = [vp, i32, i32, C.POINTER(i64), C.POINTER(i64)]
    lib.rsrgan_grad_bucket_wait.argtypes = [vp, i32, i32, vp]
    lib.rsrgan_profile_begin.argtypes = [vp]
    lib.rsrgan_profile_read.argtypes = [vp, C.POINTER(i32), C.POINTER(C.c_double), C.POINTER(C.c_double)]
    lib.rsrgan_profile_read_kind.argtypes = [vp, i32, C.POINTER(i32), C.POINTER(C.c_double), C.POINTER(C.c_double)]
    lib.rsrgan_device_status.argtypes = [vp, C.POINTER(i32)]
    lib.rsrgan_set_dropout.argtypes = [vp, f32, C.c_uint64]
    lib.rsrgan_profile_launches.argtypes = [vp, C.POINTER(i64)]
    lib.rsrgan_op_launch_floor.argtypes = [i32, i32, C.POINTER(C.c_double), vp]
    lib.rsrgan_op_gemm.argtypes = [p, i32, i32, p, i32, i32, p, i32, i32, i32, i32, p, i32, f32, i32, vp]
    lib.rsrgan_segan_default_cfg.argtypes = [C.POINTER(SeganCfg)]
    lib.rsrgan_segan_create.argtypes = [C.POINTER(SeganCfg), C.c_uint64, C.POINTER(vp)]
    lib.rsrgan_segan_destroy.argtypes = [vp]
    lib.rsrgan_segan_set_scalar.argtypes = [vp, i32, C.c_double]
    lib.rsrgan_segan_num_tensors.argtypes = [vp, i32]
    lib.rsrgan_segan_tensor_info.argtypes = [vp, i32, i32, C.c_char_p, i32, C.POINTER(i32), C.POINTER(i32), C.POINTER(i64)]
    lib.rsrgan_segan_param_count.argtypes = [vp, i32]
    lib.rsrgan_segan_param_count.restype = i64
    lib.rsrgan_segan_get_params.argtypes = [vp, i32, i32, p, vp]
    lib.rsrgan_segan_set_params.argtypes = [vp, i32, i32, p, vp]
    lib.rsrgan_segan_forward_g.argtypes = [vp, p, p, p, vp]
    lib.rsrgan_segan_d_backward.argtypes = [vp, p, p, p, p, p, p, p, i32, vp]
    lib.rsrgan_segan_g_backward.argtypes = [vp, p, p, p, p, p, p, i32, vp]
    lib.rsrgan_segan_grad_buffer.argtypes = [vp, i32, C.POINTER(p), C.POINTER(i64)]
    lib.rsrgan_segan_apply.argtypes = [vp, i32, vp]
    _lib = lib
    return lib


def check(rc):
    if rc != 0:
        raise RsrganError("librsrgan_hip error %d: %s" % (rc, load().rsrgan_last_error().decode()))
