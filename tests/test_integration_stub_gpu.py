"""The ctypes stub INTEGRATION.md shows a reference maintainer (section B) must actually work: this test carries
the same class, drives one `_update_ns` iteration of the reference's control flow with it (:290-334) and compares
with the oracle."""
import ctypes as C
import os

import numpy as np
import pytest

from oracle import corex_oracle as O

pytestmark = pytest.mark.gpu
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))


def _stub():
    _lcx = C.CDLL(os.path.join(ROOT, "linearcorex_amd", "liblcx_hip.so"))
    _lcx.lcx_last_error.restype = C.c_char_p

    def _ck(rc):
        if rc:
            raise RuntimeError(_lcx.lcx_last_error().decode())
    _p = lambda a: a.ctypes.data_as(C.c_void_p)  # noqa: E731

    class LcxDevice(object):                                     # verbatim from INTEGRATION.md, section B
        def __init__(self, x32, n_hidden):
            self.h = C.c_void_p()
            ns, nv = x32.shape
            _ck(_lcx.lcx_create(C.byref(self.h), C.c_int64(ns), C.c_int64(nv), n_hidden, 0, 0))
            _ck(_lcx.lcx_upload_x(self.h, _p(np.ascontiguousarray(x32)), C.c_int64(nv)))

        def set_ws(self, ws):
            _ck(_lcx.lcx_set_ws(self.h, _p(np.ascontiguousarray(ws))))

        def moments(self, which, eps, quick):
            _ck(_lcx.lcx_moments_a(self.h, which))
            _ck(_lcx.lcx_moments_b(self.h, which, C.c_double(eps), int(quick)))
            _ck(_lcx.lcx_moments_c(self.h, which))
            s = (C.c_double * 8)()
            _ck(_lcx.lcx_read_state(self.h, which, s))
            return None if s[2] else s[0]

        def direction(self, eps):
            _ck(_lcx.lcx_update_b(self.h, C.c_double(eps)))
            _ck(_lcx.lcx_update_c(self.h, C.c_double(eps)))
            _ck(_lcx.lcx_update_d(self.h))
            s = (C.c_double * 8)()
            _ck(_lcx.lcx_read_state(self.h, 0, s))
            return s[3]

        def trial(self, eta):
            _ck(_lcx.lcx_make_trial(self.h, C.c_double(eta)))

        def accept(self):
            _ck(_lcx.lcx_accept_trial(self.h))

        def update_ns(self, eps, tol, tc_cur, more=True):
            o = (C.c_double * 8)()
            _ck(_lcx.lcx_iterate(self.h, C.c_double(eps), C.c_double(tol), C.c_double(tc_cur), int(more), o))
            return int(o[0]), o[1], int(o[3])

        def covariance(self, eps, std32):
            nv = std32.shape[0]
            out = np.empty((nv, nv), np.float32)
            _ck(_lcx.lcx_covariance(self.h, 0, C.c_double(eps), _p(std32), _p(out), C.c_int64(nv), None))
            return out

        def get_ws(self, m, nv):
            out = np.empty((m, nv), np.float32)
            _ck(_lcx.lcx_get_ws(self.h, 0, _p(out)))
            return out

        def close(self):
            _lcx.lcx_destroy(self.h)
    return LcxDevice


def test_integration_stub_runs_update_ns():
    LcxDevice = _stub()
    n, v, m, eps = 500, 333, 5, 0.36
    x, _ = O.gen_planted(n, v, m, seed=8)
    x32 = O.preprocess(x.astype(np.float32))[0]
    w = np.random.RandomState(0).randn(m, v).astype(np.float32)
    w /= (10.0 * O.norm(x32, w, 0))[:, np.newaxis]
    w *= 3.0
    dev = LcxDevice(x32, m)
    dev.set_ws(w)
    tc = dev.moments(0, eps, False)
    mo = O.moments_ns(x32, w, eps, quick=False)
    assert abs(tc - float(mo["TC"])) < 2e-3 * max(1.0, abs(float(mo["TC"])))
    # the reference's _update_ns control flow (:305-334) on the stub
    w_ref, m_ref, info = O.update_ns(x32, w, mo, eps)
    tangent = dev.direction(eps)
    assert tangent < 0 and abs(tangent - info["tangent"]) < 2e-3 * abs(info["tangent"])
    eta, n_trials = 1.0, 0
    while True:
        dev.trial(eta)
        tc_new = dev.moments(1, eps, True)
        n_trials += 1
        if tc_new is None or -tc_new > -tc + 0.1 * eta * tangent:
            eta *= 0.5
            continue
        break
    dev.accept()
    assert n_trials == info["n_trials"] and eta == info["eta"]
    assert abs(tc_new - float(m_ref["TC"])) < 2e-3 * max(1.0, abs(float(m_ref["TC"])))
    assert np.max(np.abs(dev.get_ws(m, v) - w_ref)) < 1e-4
    dev.close()


def test_integration_stub_update_ns_in_one_call():
    """The shortest integration of INTEGRATION.md: `_update_ns` (:290-334) as one lcx_iterate call per iteration, and
    get_covariance (:443-451) as one lcx_covariance call."""
    LcxDevice = _stub()
    n, v, m, eps = 500, 333, 5, 0.36
    x, _ = O.gen_planted(n, v, m, seed=8)
    x32 = O.preprocess(x.astype(np.float32))[0]
    w = np.random.RandomState(0).randn(m, v).astype(np.float32)
    w /= (10.0 * O.norm(x32, w, 0))[:, np.newaxis]
    dev = LcxDevice(x32, m)
    dev.set_ws(w)
    tc = dev.moments(0, eps, False)
    mo = O.moments_ns(x32, w, eps, quick=False)
    w_ref = w
    for it in range(4):
        status, tc, trials = dev.update_ns(eps, 1e-5, tc, more=it < 3)
        w_ref, mo, info = O.update_ns(x32, w_ref, mo, eps)
        assert status == 0 and trials == info["n_trials"]
        assert abs(tc - float(mo["TC"])) < 2e-3 * max(1.0, abs(float(mo["TC"])))
    assert np.max(np.abs(dev.get_ws(m, v) - w_ref)) < 2e-4
    cov = dev.covariance(eps, np.ones(v, np.float32))
    cov_ref = O.covariance_ns(mo, eps, (np.zeros(v, np.float32), np.ones(v, np.float32)))
    assert np.max(np.abs(cov - cov_ref)) < 5e-3 * np.max(np.abs(cov_ref))
    dev.close()


def test_round4_entry_points_through_raw_ctypes(monkeypatch):
    """The entry points round 4 added, bound the way a reference maintainer would bind them (no package code in between):
    lcx_x_layout (what lcx_create chose for the shard), lcx_set_sample_divisor (the reference's `self.n_samples` when a handle holds
    another batch than the fitted one, :249 / :260 / :392-394), lcx_comm_probe (local: can librccl be bound?), and lcx_comm_selftest
    refusing a handle without a transport."""
    lib = C.CDLL(os.path.join(ROOT, "linearcorex_amd", "liblcx_hip.so"))
    lib.lcx_last_error.restype = C.c_char_p
    _p = lambda a: a.ctypes.data_as(C.c_void_p)  # noqa: E731
    rng = np.random.RandomState(4)
    n, v, m = 640, 900, 6
    x = O.preprocess(rng.randn(n, v))[0]
    w = rng.randn(m, v)
    w /= (10.0 * O.norm(x, w, 0))[:, np.newaxis]
    for lay, want in (("panel", 2), ("rows", 0)):
        monkeypatch.setenv("LCX_X_LAYOUT", lay)
        h = C.c_void_p()
        assert lib.lcx_create(C.byref(h), C.c_int64(n), C.c_int64(v), m, 1, 0) == 0, lib.lcx_last_error()     # 1 = LCX_F64
        layout = C.c_int(-1)
        assert lib.lcx_x_layout(h, C.byref(layout)) == 0 and layout.value == want
        assert lib.lcx_x_layout(h, None) != 0
        assert lib.lcx_upload_x(h, _p(np.ascontiguousarray(x)), C.c_int64(v)) == 0
        assert lib.lcx_set_ws(h, _p(np.ascontiguousarray(w))) == 0
        tcs = []
        for n_div in (n, 2 * n):
            assert lib.lcx_set_sample_divisor(h, C.c_double(n_div)) == 0
            assert lib.lcx_moments_a(h, 0) == 0 and lib.lcx_moments_b(h, 0, C.c_double(0.0), 0) == 0 and lib.lcx_moments_c(h, 0) == 0
            s = (C.c_double * 8)()
            assert lib.lcx_read_state(h, 0, s) == 0
            ref = O.moments_ns(x, w, 0.0, quick=False, n_samples=n_div)
            assert abs(s[0] - float(ref["TC"])) < 1e-9 * max(1.0, abs(float(ref["TC"])))
            assert abs(s[1] - float(ref["uj"].max())) < 1e-12
            tcs.append(s[0])
        assert abs(tcs[0] - tcs[1]) > 1e-3                       # the divisor matters
        assert lib.lcx_set_sample_divisor(h, C.c_double(0.0)) != 0
        ok, sec = C.c_int(7), C.c_double()
        assert lib.lcx_comm_selftest(h, 0, C.byref(ok), C.byref(sec)) != 0 and ok.value == 0
        assert b"no transport" in lib.lcx_last_error()
        assert lib.lcx_destroy(h) == 0
    assert lib.lcx_comm_probe() == 0, lib.lcx_last_error()       # librccl ships with the ROCm image


def test_f32_gemm_entry_points_through_raw_ctypes(monkeypatch):
    """lcx_set_f32_gemm / lcx_f32_gemm bound the way a reference maintainer would bind them: a float32 panel shard switches between the
    float32 MFMA and the bf16-split X passes between two launches (same moments to float32 rounding, both at the float32 bar against
    the oracle); a float64 handle and a row-major shard accept the call and stay in mode 0; a bad mode is refused."""
    lib = C.CDLL(os.path.join(ROOT, "linearcorex_amd", "liblcx_hip.so"))
    lib.lcx_last_error.restype = C.c_char_p
    _p = lambda a: a.ctypes.data_as(C.c_void_p)  # noqa: E731
    rng = np.random.RandomState(4)
    n, v, m = 1280, 1900, 40
    x = O.preprocess(rng.randn(n, v))[0].astype(np.float32)
    w = rng.randn(m, v).astype(np.float32)
    w /= (10.0 * O.norm(x, w, 0))[:, np.newaxis]
    ref = O.moments_ns(x.astype(np.float64), w.astype(np.float64), 0.0, quick=False)
    for lay, dt, can in (("panel", 0, True), ("rows", 0, False), ("panel", 1, False)):       # 0 = LCX_F32, 1 = LCX_F64
        monkeypatch.setenv("LCX_X_LAYOUT", lay)
        monkeypatch.setenv("LCX_GEMM", "ct")
        h = C.c_void_p()
        assert lib.lcx_create(C.byref(h), C.c_int64(n), C.c_int64(v), m, dt, 0) == 0, lib.lcx_last_error()
        mode = C.c_int(-1)
        dflt = 1 if (os.environ.get("LCX_F32_GEMM") == "split" and can) else 0     # the default: float32 MFMA unless the environment says split
        assert lib.lcx_f32_gemm(h, C.byref(mode)) == 0 and mode.value == dflt
        assert lib.lcx_f32_gemm(h, None) != 0 and lib.lcx_set_f32_gemm(h, 2) != 0
        xx, ww = (x, w) if dt == 0 else (x.astype(np.float64), w.astype(np.float64))
        assert lib.lcx_upload_x(h, _p(np.ascontiguousarray(xx)), C.c_int64(v)) == 0
        assert lib.lcx_set_ws(h, _p(np.ascontiguousarray(ww))) == 0
        tcs = []
        for want in (1, 0):
            assert lib.lcx_set_f32_gemm(h, want) == 0
            assert lib.lcx_f32_gemm(h, C.byref(mode)) == 0 and mode.value == (want if can else 0)
            name = C.create_string_buffer(256)
            assert lib.lcx_kernel_name(h, 0, name, C.c_int64(256)) == 0
            assert (b"gemm_split_kernel" in name.value) == (mode.value == 1)
            assert lib.lcx_moments_a(h, 0) == 0 and lib.lcx_moments_b(h, 0, C.c_double(0.0), 0) == 0 and lib.lcx_moments_c(h, 0) == 0
            s = (C.c_double * 8)()
            assert lib.lcx_read_state(h, 0, s) == 0
            assert abs(s[0] - float(ref["TC"])) < (2e-4 if dt == 0 else 1e-9) * max(1.0, abs(float(ref["TC"])))
            tcs.append(s[0])
        assert abs(tcs[0] - tcs[1]) < 1e-5 * max(1.0, abs(tcs[1]))
        assert lib.lcx_destroy(h) == 0
