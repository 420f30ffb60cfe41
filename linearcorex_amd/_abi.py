"""ctypes binding of include/lcx.h (liblcx_hip.so).

There is deliberately no CPU fallback: if the HIP library is missing or no MI355X is visible the
import of the library / creation of a handle raises.  (The NumPy restatement under oracle/ is test
infrastructure and is never imported from here.)
"""
from __future__ import annotations

import ctypes as C
import os

import numpy as np

_HERE = os.path.dirname(os.path.abspath(__file__))
LIB_PATH = os.path.join(_HERE, "liblcx_hip.so")

LCX_F32, LCX_F64 = 0, 1
S_TC, S_MAX_UJ, S_INVALID, S_TANGENT, S_SUM_LOG_RJ, S_COUNT = 0, 1, 2, 3, 4, 8
SB_H = 8                   # offset of H in the scalar exchange buffer (include/lcx.h, lcx_read_sbuf)

# lcx_moment_key
M_UJ, M_RHO, M_RY, M_INVRHO, M_RHOINVRHO, M_QIJ, M_SI, M_QISI2, M_MI, M_XIZJ, M_XI2_GIVEN_Y, \
    M_GRAD, M_UPDATE, M_SIG_GRAD, M_H, M_Y, M_SYN_XIZJ, M_SYN_X2Y, M_SYN_XIYJ, M_CY, M_YJ2 = range(21)

_i64, _i32, _dbl, _vp = C.c_int64, C.c_int, C.c_double, C.c_void_p
COMM_ID_BYTES = 128
# lcx_allreduce_fn: int (*)(void* user, void* dev_buf, int64_t count, int dtype, void* hip_stream)
ALLREDUCE_FN = C.CFUNCTYPE(C.c_int, _vp, _vp, _i64, _i32, _vp)

# name -> (argtypes); every function returns int status except the two noted below
SIGNATURES = {
    "lcx_abi_version": [],
    "lcx_last_error": [],
    "lcx_device_count": [C.POINTER(_i32)],
    "lcx_create": [C.POINTER(_vp), _i64, _i64, _i32, _i32, _i32],
    "lcx_destroy": [_vp],
    "lcx_set_stream": [_vp, _vp],
    "lcx_synchronize": [_vp],
    "lcx_set_world": [_vp, _i32],
    "lcx_set_linear_mode": [_vp, _i32],
    "lcx_set_trial_reuse": [_vp, _i32],
    "lcx_set_sample_divisor": [_vp, C.c_double],
    "lcx_comm_selftest": [_vp, _i32, C.POINTER(_i32), C.POINTER(_dbl)],
    "lcx_comm_probe": [],
    "lcx_x_layout": [_vp, C.POINTER(_i32)],
    "lcx_set_f32_gemm": [_vp, _i32],
    "lcx_f32_gemm": [_vp, C.POINTER(_i32)],
    "lcx_set_exchange": [_vp, _i32],
    "lcx_exchange_layout": [_vp, C.POINTER(_i64), C.POINTER(_i64), C.POINTER(_vp), C.POINTER(_vp)],
    "lcx_bind_exchange": [_vp, _vp, _vp],
    "lcx_comm_unique_id": [_vp],
    "lcx_comm_init": [_vp, _i32, _i32, _vp],
    "lcx_set_exchange_hook": [_vp, ALLREDUCE_FN, _vp],
    "lcx_exchange_info": [_vp, C.POINTER(_i32), C.POINTER(_i32), C.POINTER(_i64)],
    "lcx_upload_x": [_vp, _vp, _i64],
    "lcx_upload_preprocess": [_vp, _vp, _i64, _i32, _i32, _dbl, _i32, _vp, _vp, C.POINTER(_i64), C.POINTER(_dbl)],
    "lcx_project_raw": [_vp, _vp, _i64, _i64, _i32, _vp, _vp, _vp],
    "lcx_predict": [_vp, _vp, _i64, _i32, _vp, _i32, _vp, _vp, _vp, _i64, C.POINTER(_dbl)],
    "lcx_invert": [_vp, _vp, _i64, _i64, _i32, _vp, _vp, _vp, _i64],
    "lcx_generate_x": [_vp, C.c_uint64, _i32, _i32, _i64],
    "lcx_download_x": [_vp, _vp, _i64],
    "lcx_set_ws": [_vp, _vp],
    "lcx_get_ws": [_vp, _i32, _vp],
    "lcx_permute_factors": [_vp, C.POINTER(C.c_int32)],
    "lcx_moments_a": [_vp, _i32],
    "lcx_moments_b": [_vp, _i32, _dbl, _i32],
    "lcx_moments_c": [_vp, _i32],
    "lcx_moments_detail": [_vp, _i32],
    "lcx_update_a": [_vp],
    "lcx_update_b": [_vp, _dbl],
    "lcx_update_c": [_vp, _dbl],
    "lcx_update_d": [_vp],
    "lcx_make_trial": [_vp, _dbl],
    "lcx_trial_linear_a": [_vp, _dbl],
    "lcx_trial_linear_b": [_vp, _dbl, _dbl],
    "lcx_accept_trial": [_vp],
    "lcx_iterate": [_vp, _dbl, _dbl, _dbl, _i32, C.POINTER(_dbl)],
    "lcx_syn_moments_b": [_vp, _i32, _dbl],
    "lcx_syn_moments_c": [_vp, _i32],
    "lcx_syn_update_a": [_vp],
    "lcx_syn_update_b": [_vp, _dbl],
    "lcx_covariance_rows_syn": [_vp, _vp, _i64, _i64, _vp],
    "lcx_rescale_ws": [_vp, _dbl, _dbl],
    "lcx_init_scale_ws": [_vp],
    "lcx_read_state": [_vp, _i32, C.POINTER(_dbl)],
    "lcx_get_moment": [_vp, _i32, _i32, _dbl, _vp],
    "lcx_set_moment": [_vp, _i32, _i32, _vp],
    "lcx_read_sbuf": [_vp, _i64, _i64, C.POINTER(_dbl)],
    "lcx_covariance_rows": [_vp, _dbl, _vp, _i64, _i64, _vp],
    "lcx_covariance": [_vp, _i32, _dbl, _vp, _vp, _i64, C.POINTER(_dbl)],
    "lcx_bytes_resident": [_vp, C.POINTER(_i64), C.POINTER(_i64)],
    "lcx_project": [_vp, _vp, _i64, _i64, _vp],
    "lcx_timing_enable": [_vp, _i32],
    "lcx_timing_sample": [_vp, _i32],
    "lcx_timing_read": [_vp, _i32, C.POINTER(_i64), C.POINTER(_dbl)],
    "lcx_timing_passes": [_vp, _i32, C.POINTER(_i64)],
    "lcx_timing_reset": [_vp],
    "lcx_geometry": [_vp, C.POINTER(_i64), C.POINTER(_i64), C.POINTER(_i32), C.POINTER(_i64)],
    "lcx_kernel_name": [_vp, _i32, C.c_char_p, _i64],
}

_lib = None


class LcxError(RuntimeError):
    pass


def load():
    """Load liblcx_hip.so (built by __graft_entry__.build()).  Raises if it is missing."""
    global _lib
    if _lib is not None:
        return _lib
    if not os.path.exists(LIB_PATH):
        raise LcxError(
            "linearcorex_amd: HIP library %s is missing - build it with "
            "`python -c 'import __graft_entry__ as g; g.build()'` (there is no CPU fallback)" % LIB_PATH)
    lib = C.CDLL(LIB_PATH)
    for name, args in SIGNATURES.items():
        fn = getattr(lib, name)           # AttributeError here == header/library mismatch
        fn.argtypes = args
        fn.restype = C.c_char_p if name == "lcx_last_error" else C.c_int
    _lib = lib
    return lib


def check(status):
    if status != 0:
        msg = load().lcx_last_error()
        raise LcxError("liblcx_hip status %d: %s" % (status, (msg or b"").decode("utf-8", "replace")))


def np_ptr(a):
    return a.ctypes.data_as(C.c_void_p)


def dtype_code(dt):
    dt = np.dtype(dt)
    if dt == np.float32:
        return LCX_F32
    if dt == np.float64:
        return LCX_F64
    raise ValueError("working dtype must be float32 or float64")
