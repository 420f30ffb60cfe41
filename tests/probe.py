"""ctypes binding of tools/liblcx_probe.so (tools/lcx_probe.h): the engine compiled together with the kernel unit-test
hooks and micro-benchmarks that the product library (include/lcx.h) no longer exports.  Test / lab infrastructure."""
import ctypes as C
import os

import numpy as np

from linearcorex_amd import _abi
from linearcorex_amd.backend import HipBackend

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
LIB_PATH = os.path.join(ROOT, "tools", "liblcx_probe.so")
_i64, _i32, _dbl, _vp = C.c_int64, C.c_int, C.c_double, C.c_void_p
PROBE_SIGNATURES = {
    "lcx_bench_gemm": [_vp, _i32, _i32, C.POINTER(_dbl)],
    "lcx_bench_graph": [_vp, _dbl, _i32, C.POINTER(_dbl), C.POINTER(_dbl)],
    "lcx_test_gemm_nt": [_i32, _i32, _vp, _i64, _i64, _i64, _vp, _i32, _vp, _i32, _i32],
    "lcx_test_gemm_tn": [_i32, _i32, _vp, _i64, _i64, _i64, _vp, _i32, _vp, _vp, _i32, _i32],
}
_lib = None


def load():
    global _lib
    if _lib is None:
        if not os.path.exists(LIB_PATH):
            raise OSError("%s not found: run `python __graft_entry__.py` (build()) first" % LIB_PATH)
        lib = C.CDLL(LIB_PATH)
        for name, args in list(_abi.SIGNATURES.items()) + list(PROBE_SIGNATURES.items()):
            fn = getattr(lib, name)
            fn.argtypes = args
            fn.restype = C.c_char_p if name == "lcx_last_error" else C.c_int
        _lib = lib
    return _lib


def check(status):
    if status != 0:
        raise _abi.LcxError("liblcx_probe status %d: %s" % (status, (load().lcx_last_error() or b"").decode("utf-8", "replace")))


class ProbeBackend(HipBackend):
    """A shard handle created through the probe library (same ABI), with the micro-benchmarks on top."""

    def __init__(self, n_samples, nv_local, n_hidden, dtype=np.float32, device=0):
        HipBackend.__init__(self, n_samples, nv_local, n_hidden, dtype, device, lib=load())

    def bench_graph(self, eps=0.1, iters=50):
        """(direct_ms, graph_ms) per moment evaluation: plain launches vs a replayed hipGraph (experiment)."""
        d, g = C.c_double(), C.c_double()
        check(self.lib.lcx_bench_graph(self.h, float(eps), int(iters), C.byref(d), C.byref(g)))
        return d.value, g.value

    def bench_gemm(self, kind, iters=20):
        ms = C.c_double()
        check(self.lib.lcx_bench_gemm(self.h, int(kind), int(iters), C.byref(ms)))
        return ms.value


def gemm_nt_check(a, b_km, m_pad, dtype, device=0, split=1, waves=4):
    """Isolated run of the X.B^T kernel: a (n x k), b_km (k x m_pad) -> (n x m_pad)."""
    lib = load()
    a = np.ascontiguousarray(a, dtype=dtype)
    b = np.ascontiguousarray(b_km, dtype=dtype)
    out = np.empty((a.shape[0], m_pad), dtype=dtype)
    check(lib.lcx_test_gemm_nt(_abi.dtype_code(dtype), device, _abi.np_ptr(a), a.shape[0], a.shape[1],
                               a.shape[1], _abi.np_ptr(b), m_pad, _abi.np_ptr(out), split, waves))
    return out


def gemm_tn_check(a, b_km, m_pad, dtype, device=0, rowscale=None, split=1, waves=4):
    """Isolated run of the A^T.B kernel: a (k x v), b_km (k x m_pad) -> (v x m_pad)."""
    lib = load()
    a = np.ascontiguousarray(a, dtype=dtype)
    b = np.ascontiguousarray(b_km, dtype=dtype)
    rs = None if rowscale is None else np.ascontiguousarray(rowscale, dtype=dtype)
    out = np.empty((a.shape[1], m_pad), dtype=dtype)
    check(lib.lcx_test_gemm_tn(_abi.dtype_code(dtype), device, _abi.np_ptr(a), a.shape[0], a.shape[1],
                               a.shape[1], _abi.np_ptr(b), m_pad,
                               None if rs is None else _abi.np_ptr(rs), _abi.np_ptr(out), split, waves))
    return out
