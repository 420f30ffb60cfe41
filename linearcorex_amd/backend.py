"""HipBackend: one n_variables shard of a fit, resident on one MI355X, driven through the C ABI.

The method set below is the *backend interface* the host driver (`corex.Corex`) talks to; it
mirrors the dependency levels of the reference's `_calculate_moments_ns` / `_update_ns`
(linearcorex.py:236-334), cut where a sum over all variables is needed.  Tests exercise the
multi-rank host logic with a NumPy test double of the same interface (tests/shard_double.py).
"""
from __future__ import annotations

import ctypes as C

import numpy as np

from . import _abi

# reference moment name -> (ABI key, per-variable?)   (linearcorex.py:249-287)
MOMENT_KEYS = {
    "uj": _abi.M_UJ, "rho": _abi.M_RHO, "ry": _abi.M_RY, "invrho": _abi.M_INVRHO,
    "rhoinvrho": _abi.M_RHOINVRHO, "Qij": _abi.M_QIJ, "Si": _abi.M_SI, "Qi-Si^2": _abi.M_QISI2,
    "MI": _abi.M_MI, "X_i Z_j": _abi.M_XIZJ, "X_i^2 | Y": _abi.M_XI2_GIVEN_Y,
    "grad": _abi.M_GRAD, "update": _abi.M_UPDATE, "sig_grad": _abi.M_SIG_GRAD, "H": _abi.M_H,
    "Y": _abi.M_Y,
    # synergistic branch (:336-373)
    "syn X_i Z_j": _abi.M_SYN_XIZJ, "syn X_i^2 | Y": _abi.M_SYN_X2Y, "syn X_i Y_j": _abi.M_SYN_XIYJ,
    "cy": _abi.M_CY, "Y_j^2": _abi.M_YJ2,
}


class HipBackend:
    def __init__(self, n_samples, nv_local, n_hidden, dtype=np.float32, device=0, lib=None):
        # lib: another build of the same ABI (tools/liblcx_probe.so, the lab: the engine plus kernel test hooks)
        self.lib = lib if lib is not None else _abi.load()
        self.dtype = np.dtype(dtype)
        self.n_samples, self.nv, self.m = int(n_samples), int(nv_local), int(n_hidden)
        self.device = int(device)
        h = C.c_void_p()
        _abi.check(self.lib.lcx_create(C.byref(h), self.n_samples, self.nv, self.m,
                                       _abi.dtype_code(self.dtype), self.device))
        self.h = h
        self._ex = None            # torch exchange tensors when sharded
        self.torch_stream = None
        self.generation = 0        # bumps whenever moment set 0 changes (guards lazy readback)

    # ---- lifetime ---------------------------------------------------------------------------
    def close(self):
        if getattr(self, "h", None):
            self.lib.lcx_destroy(self.h)
            self.h = None

    def __del__(self):
        try:
            self.close()
        except Exception:
            pass

    def _a(self, arr):
        arr = np.ascontiguousarray(arr, dtype=self.dtype)
        return arr, _abi.np_ptr(arr)

    # ---- geometry / timing -------------------------------------------------------------------
    def geometry(self):
        n_pad, ldx, mp = C.c_int64(), C.c_int64(), C.c_int()
        info = (C.c_int64 * 8)()
        _abi.check(self.lib.lcx_geometry(self.h, C.byref(n_pad), C.byref(ldx), C.byref(mp), info))
        names = ["nt_split", "nt_waves", "tn_split", "tn_waves", "nt_blocks_per_cu", "tn_blocks_per_cu",
                 "pv_grid", "n_cus"]
        g = dict(zip(names, [int(v) for v in info]))
        g.update(n_pad=n_pad.value, ldx=ldx.value, m_pad=mp.value)
        return g

    def timing_enable(self, on=True):
        _abi.check(self.lib.lcx_timing_enable(self.h, 1 if on else 0))

    def timing_sample(self, every):
        _abi.check(self.lib.lcx_timing_sample(self.h, int(every)))

    def timing_reset(self):
        _abi.check(self.lib.lcx_timing_reset(self.h))

    def timing_read(self):
        out = {}
        # gemm_nt2: the merged pass X.[grad | ws + update]^T (2 x n_hidden columns, one read of X; float32 large shards)
        for kind, name in ((0, "gemm_nt"), (1, "gemm_tn"), (2, "gemm_nt2")):
            n, ms = C.c_int64(), C.c_double()
            _abi.check(self.lib.lcx_timing_read(self.h, kind, C.byref(n), C.byref(ms)))
            out[name] = (n.value, ms.value)
        return out

    def timing_passes(self):
        """X passes issued while timing was on (every pass, whatever `timing_sample` says)."""
        n = C.c_int64()
        tot = 0
        for kind in (0, 1, 2):
            _abi.check(self.lib.lcx_timing_passes(self.h, kind, C.byref(n)))
            tot += n.value
        return tot

    def timing_passes_by_kind(self):
        out = {}
        n = C.c_int64()
        for kind, name in ((0, "gemm_nt"), (1, "gemm_tn"), (2, "gemm_nt2")):
            _abi.check(self.lib.lcx_timing_passes(self.h, kind, C.byref(n)))
            out[name] = n.value
        return out

    EXCHANGE_SITES = ((3, "y"), (4, "direction"), (5, "scalars"), (6, "small"))

    def timing_exchange_read(self):
        """The exchange steps issued inside the library while timing was on (include/lcx.h, timing kinds 3-6):
        site -> (issued, timed, total ms of the timed ones)."""
        out = {}
        issued, n, ms = C.c_int64(), C.c_int64(), C.c_double()
        for kind, name in self.EXCHANGE_SITES:
            _abi.check(self.lib.lcx_timing_passes(self.h, kind, C.byref(issued)))
            _abi.check(self.lib.lcx_timing_read(self.h, kind, C.byref(n), C.byref(ms)))
            out[name] = (issued.value, n.value, ms.value)
        return out

    def kernel_name(self, kind):
        buf = C.create_string_buffer(256)
        _abi.check(self.lib.lcx_kernel_name(self.h, int(kind), buf, 256))
        return buf.value.decode()

    def set_world(self, world):
        _abi.check(self.lib.lcx_set_world(self.h, int(world)))

    def set_exchange(self, enable):
        _abi.check(self.lib.lcx_set_exchange(self.h, 1 if enable else 0))

    def set_linear_mode(self, enable):
        _abi.check(self.lib.lcx_set_linear_mode(self.h, 1 if enable else 0))

    def set_trial_reuse(self, enable):
        _abi.check(self.lib.lcx_set_trial_reuse(self.h, 1 if enable else 0))

    def set_sample_divisor(self, n_samples):
        """the reference's `self.n_samples` when this handle holds another batch than the fitted one (transform(details=True))"""
        _abi.check(self.lib.lcx_set_sample_divisor(self.h, float(n_samples)))

    def synchronize(self):
        _abi.check(self.lib.lcx_synchronize(self.h))

    # ---- exchange inside the library (include/lcx.h: lcx_comm_init / lcx_set_exchange_hook) -------
    def comm_unique_id(self):
        buf = C.create_string_buffer(_abi.COMM_ID_BYTES)
        _abi.check(self.lib.lcx_comm_unique_id(buf))
        return buf.raw

    def comm_probe(self):
        """Can THIS process load librccl (lcx_comm_probe)?  Local, no collective: the ranks compare notes before any of them
        enters ncclCommInitRank."""
        _abi.check(self.lib.lcx_comm_probe())

    def comm_init(self, world, rank, unique_id):
        """RCCL communicator owned by the handle (collective: every rank of the group calls it with rank 0's id)."""
        assert len(unique_id) == _abi.COMM_ID_BYTES
        _abi.check(self.lib.lcx_comm_init(self.h, int(world), int(rank), C.c_char_p(unique_id)))

    def comm_selftest(self, rank=0):
        """First contact with the bound transport (collective): the Y exchange buffer is all-reduced at its real size with
        patterns whose outcome is known / must be rank-identical (include/lcx.h, lcx_comm_selftest).  Raises LcxError with the
        diagnosis on EVERY rank if any rank saw a wrong sum or other bits; returns the seconds one such all-reduce took."""
        ok, sec = C.c_int(0), C.c_double(0.0)
        _abi.check(self.lib.lcx_comm_selftest(self.h, int(rank), C.byref(ok), C.byref(sec)))
        assert ok.value == 1
        return sec.value

    def set_exchange_hook(self, allreduce):
        """allreduce(dev_ptr, count, dtype_code, hip_stream) -> None sums the buffer over the ranks in place (any transport).
        The ctypes thunk is kept alive on the backend for as long as the library may call it."""
        if allreduce is None:
            _abi.check(self.lib.lcx_set_exchange_hook(self.h, _abi.ALLREDUCE_FN(0), None))
            self._hook = None
            return

        def thunk(user, ptr, count, dtype, stream):
            try:
                allreduce(int(ptr), int(count), int(dtype), stream)
                return 0
            except BaseException:          # must not propagate into C; the library turns the code into LCX_ERR_COMM
                import traceback
                traceback.print_exc()
                return 1
        self._hook = _abi.ALLREDUCE_FN(thunk)
        _abi.check(self.lib.lcx_set_exchange_hook(self.h, self._hook, None))

    def exchange_info(self):
        kind, world, n = C.c_int(), C.c_int(), C.c_int64()
        _abi.check(self.lib.lcx_exchange_info(self.h, C.byref(kind), C.byref(world), C.byref(n)))
        return {"kind": {-1: "none", 0: "caller", 1: "rccl", 2: "hook"}[kind.value], "world": world.value,
                "allreduces_issued": n.value}

    # ---- exchange buffers ---------------------------------------------------------------------
    def exchange_tensors(self):
        """(ybuf, sbuf) as torch tensors on this GPU, bound as the handle's exchange buffers and
        with the handle's work moved to torch's current stream (so that torch.distributed
        collectives on those tensors are stream-ordered with the kernels)."""
        if self._ex is None:
            import torch
            ye, se = C.c_int64(), C.c_int64()
            _abi.check(self.lib.lcx_exchange_layout(self.h, C.byref(ye), C.byref(se), None, None))
            dev = torch.device("cuda", self.device)
            tdt = torch.float32 if self.dtype == np.float32 else torch.float64
            y = torch.zeros(ye.value, dtype=tdt, device=dev)
            s = torch.zeros(se.value, dtype=torch.float64, device=dev)
            torch.cuda.synchronize(dev)
            _abi.check(self.lib.lcx_bind_exchange(self.h, C.c_void_p(y.data_ptr()), C.c_void_p(s.data_ptr())))
            # A stream of our own that torch knows about: torch's default stream is the null stream (handle
            # 0), which the engine's non-blocking stream is NOT ordered with.  Every collective on the
            # exchange tensors must be issued under `with torch.cuda.stream(self.torch_stream)`.
            self.torch_stream = torch.cuda.Stream(device=dev)
            assert self.torch_stream.cuda_stream != 0
            _abi.check(self.lib.lcx_set_stream(self.h, C.c_void_p(self.torch_stream.cuda_stream)))
            self._ex = (y, s)
        return self._ex

    def stream_context(self):
        """Context under which torch work (collectives) is stream-ordered with the engine's kernels."""
        import torch
        return torch.cuda.stream(self.torch_stream)

    def sbuf_ranges(self):
        """(scalars + H, detail) as (offset, count) in the scalar exchange buffer (include/lcx.h, lcx_read_sbuf)."""
        mp = self.geometry()["m_pad"] if not hasattr(self, "m_pad") else self.m_pad
        self.m_pad = mp
        return (0, _abi.SB_H + mp * mp), (_abi.SB_H + mp * mp, self.m + 3)

    def read_sbuf(self, count):
        """the first `count` detail sums (lcx_moments_detail / lcx_syn_moments_b)"""
        out = np.empty(int(count), dtype=np.float64)
        off = self.sbuf_ranges()[1][0]
        _abi.check(self.lib.lcx_read_sbuf(self.h, off, int(count), out.ctypes.data_as(C.POINTER(C.c_double))))
        return out

    # ---- data ---------------------------------------------------------------------------------
    def upload_x(self, x):
        x, p = self._a(x)
        assert x.shape == (self.n_samples, self.nv), x.shape
        _abi.check(self.lib.lcx_upload_x(self.h, p, x.shape[1]))

    PP_KINDS = {"standard": 1, "outliers": 2, "empirical": 3}     # anything else passes through (reference :404-405)

    def upload_preprocess(self, x_raw, gaussianize, missing_values=None, theta=None):
        """preprocess(x, fit=theta is None) of the reference (:397-429) on the device for this shard.
        Returns (theta, n_obs, max_abs): theta = (mean, std) in the working dtype, n_obs per column
        (int64 array) when missing values are enabled, else n_samples."""
        x, p = self._a(x_raw)
        assert x.shape == (self.n_samples, self.nv), x.shape
        kind = self.PP_KINDS.get(gaussianize, 0)
        fit = theta is None
        mean = np.zeros(self.nv, self.dtype) if fit else np.ascontiguousarray(theta[0], self.dtype)
        std = np.ones(self.nv, self.dtype) if fit else np.ascontiguousarray(theta[1], self.dtype)
        n_obs = np.zeros(self.nv, np.int64)
        mx = C.c_double()
        has_missing = missing_values is not None
        _abi.check(self.lib.lcx_upload_preprocess(self.h, p, x.shape[1], kind, 1 if has_missing else 0,
                                                  float(missing_values) if has_missing else 0.0, 1 if fit else 0,
                                                  _abi.np_ptr(mean), _abi.np_ptr(std),
                                                  n_obs.ctypes.data_as(C.POINTER(C.c_int64)), C.byref(mx)))
        return (mean, std), (n_obs if has_missing else self.n_samples), mx.value

    def project_resident(self):
        """x~ . ws^T of the resident shard itself: (n_samples, m), this shard's partial."""
        self.moments_a(0)
        return self.get_moment(0, "Y")

    def project_raw(self, x_raw, gaussianize, theta):
        x, p = self._a(x_raw)
        assert x.ndim == 2 and x.shape[1] == self.nv
        kind = self.PP_KINDS.get(gaussianize, 0)
        mean = np.ascontiguousarray(theta[0], self.dtype)
        std = np.ascontiguousarray(theta[1], self.dtype)
        out = np.empty((x.shape[0], self.m), dtype=self.dtype)
        _abi.check(self.lib.lcx_project_raw(self.h, p, x.shape[0], x.shape[1], kind, _abi.np_ptr(mean), _abi.np_ptr(std),
                                            _abi.np_ptr(out)))
        return out

    def generate_x(self, seed, kind=0, n_groups=1, col_offset=0):
        _abi.check(self.lib.lcx_generate_x(self.h, int(seed), int(kind), int(n_groups), int(col_offset)))

    def download_x(self):
        out = np.empty((self.n_samples, self.nv), dtype=self.dtype)
        _abi.check(self.lib.lcx_download_x(self.h, _abi.np_ptr(out), self.nv))
        return out

    def set_ws(self, w):
        w, p = self._a(w)
        assert w.shape == (self.m, self.nv), w.shape
        _abi.check(self.lib.lcx_set_ws(self.h, p))
        self.generation += 1

    def get_ws(self, which=0):
        out = np.empty((self.m, self.nv), dtype=self.dtype)
        _abi.check(self.lib.lcx_get_ws(self.h, which, _abi.np_ptr(out)))
        return out

    def permute_factors(self, order):
        order = np.ascontiguousarray(order, dtype=np.int32)
        _abi.check(self.lib.lcx_permute_factors(self.h, order.ctypes.data_as(C.POINTER(C.c_int32))))
        self.generation += 1

    # ---- moments (linearcorex.py:236-288) -------------------------------------------------------
    def moments_a(self, which):
        _abi.check(self.lib.lcx_moments_a(self.h, which))

    def moments_b(self, which, eps, quick):
        _abi.check(self.lib.lcx_moments_b(self.h, which, float(eps), 1 if quick else 0))
        if which == 0:
            self.generation += 1

    def moments_c(self, which):
        _abi.check(self.lib.lcx_moments_c(self.h, which))

    def moments_detail(self, which):
        _abi.check(self.lib.lcx_moments_detail(self.h, which))

    # ---- update (linearcorex.py:290-334) ---------------------------------------------------------
    def update_a(self):
        _abi.check(self.lib.lcx_update_a(self.h))

    def update_b(self, eps):
        _abi.check(self.lib.lcx_update_b(self.h, float(eps)))

    def update_c(self, eps):
        _abi.check(self.lib.lcx_update_c(self.h, float(eps)))

    def update_d(self):
        _abi.check(self.lib.lcx_update_d(self.h))

    def make_trial(self, eta):
        _abi.check(self.lib.lcx_make_trial(self.h, float(eta)))

    def trial_linear_a(self, eta):
        _abi.check(self.lib.lcx_trial_linear_a(self.h, float(eta)))

    def trial_linear_b(self, eps, eta):
        _abi.check(self.lib.lcx_trial_linear_b(self.h, float(eps), float(eta)))

    def accept_trial(self):
        _abi.check(self.lib.lcx_accept_trial(self.h))
        self.generation += 1

    def iterate(self, eps, tol, tc_cur, more):
        """One whole `_update_ns` (linearcorex.py:290-334) inside the library (one GPU): returns the 8 scalars of
        lcx_iterate (status, TC, tangent, trials, invalid trials, step-too-small, evaluations, next iteration started)."""
        out = np.empty(8, dtype=np.float64)
        _abi.check(self.lib.lcx_iterate(self.h, float(eps), float(tol), float(tc_cur), 1 if more else 0,
                                        out.ctypes.data_as(C.POINTER(C.c_double))))
        if out[0] != 1:
            self.generation += 1
        return out

    # ---- synergistic branch (linearcorex.py:336-384) ---------------------------------------------
    def syn_moments_b(self, which, yscale):
        _abi.check(self.lib.lcx_syn_moments_b(self.h, which, float(yscale)))
        if which == 0:
            self.generation += 1

    def syn_moments_c(self, which):
        _abi.check(self.lib.lcx_syn_moments_c(self.h, which))

    def syn_update_a(self):
        _abi.check(self.lib.lcx_syn_update_a(self.h))

    def syn_update_b(self, eta):
        _abi.check(self.lib.lcx_syn_update_b(self.h, float(eta)))

    def covariance_syn(self, std):
        """get_covariance of the synergistic branch (linearcorex.py:452-455) for this shard."""
        return self._covariance(1, 0.0, std)

    def rescale_ws(self, eps_old, eps_new):
        _abi.check(self.lib.lcx_rescale_ws(self.h, float(eps_old), float(eps_new)))
        self.generation += 1

    def init_scale_ws(self):
        _abi.check(self.lib.lcx_init_scale_ws(self.h))
        self.generation += 1

    # ---- readback ---------------------------------------------------------------------------------
    def read_state(self, which):
        out = np.empty(_abi.S_COUNT, dtype=np.float64)
        _abi.check(self.lib.lcx_read_state(self.h, which, out.ctypes.data_as(C.POINTER(C.c_double))))
        return out

    def moment_shape(self, name):
        m, nv, n = self.m, self.nv, self.n_samples
        return {"uj": (m,), "ry": (m, m), "H": (m, m), "Si": (nv,), "Qi-Si^2": (nv,),
                "X_i^2 | Y": (nv,), "X_i Z_j": (nv, m), "Y": (n, m), "syn X_i Z_j": (nv, m), "syn X_i^2 | Y": (nv,),
                "syn X_i Y_j": (nv, m), "cy": (m, m), "Y_j^2": (m,)}.get(name, (m, nv))

    def get_moment(self, which, name, eps=0.0):
        out = np.empty(self.moment_shape(name), dtype=self.dtype)
        _abi.check(self.lib.lcx_get_moment(self.h, which, MOMENT_KEYS[name], float(eps), _abi.np_ptr(out)))
        return out

    def set_moment(self, which, name, value):
        value, p = self._a(value)
        assert value.shape == self.moment_shape(name)
        _abi.check(self.lib.lcx_set_moment(self.h, which, MOMENT_KEYS[name], p))

    # ---- outputs ------------------------------------------------------------------------------------
    def covariance(self, eps, std):
        """get_covariance (linearcorex.py:443-451) for this shard."""
        return self._covariance(0, eps, std)

    def _covariance(self, syn, eps, std):
        std, ps = self._a(std)
        out = np.empty((self.nv, self.nv), dtype=self.dtype)
        ksec = C.c_double()
        _abi.check(self.lib.lcx_covariance(self.h, int(syn), float(eps), ps, _abi.np_ptr(out), self.nv, C.byref(ksec)))
        self._cov_kernel_seconds = ksec.value
        return out

    def covariance_rows(self, eps, std, row0, nrows, syn=False):
        """rows [row0, row0 + nrows) of get_covariance (linearcorex.py:443-455) for this shard, (nrows, nv)."""
        std, ps = self._a(std)
        out = np.empty((int(nrows), self.nv), dtype=self.dtype)
        if syn:
            _abi.check(self.lib.lcx_covariance_rows_syn(self.h, ps, int(row0), int(nrows), _abi.np_ptr(out)))
        else:
            _abi.check(self.lib.lcx_covariance_rows(self.h, float(eps), ps, int(row0), int(nrows), _abi.np_ptr(out)))
        return out

    def last_covariance_device_seconds(self):
        """device time of the product kernels of the last get_covariance call (HIP events)"""
        return getattr(self, "_cov_kernel_seconds", None)

    F32_GEMM = {"mfma": 0, "split": 1}

    def set_f32_gemm(self, mode):
        """Arithmetic of the two X passes of a float32 panel-layout shard (include/lcx.h, lcx_set_f32_gemm): "mfma" = float32
        MFMA (default), "split" = exact three-way bf16 split of every operand, 6 partial products on the bf16 pipe.  Returns
        the mode in force ("split" is only taken where the shard supports it)."""
        _abi.check(self.lib.lcx_set_f32_gemm(self.h, self.F32_GEMM[mode]))
        return self.f32_gemm()

    def f32_gemm(self):
        m = C.c_int()
        _abi.check(self.lib.lcx_f32_gemm(self.h, C.byref(m)))
        return "split" if m.value == 1 else "mfma"

    def bytes_resident(self):
        tot, xb = C.c_int64(), C.c_int64()
        _abi.check(self.lib.lcx_bytes_resident(self.h, C.byref(tot), C.byref(xb)))
        lay = C.c_int()
        _abi.check(self.lib.lcx_x_layout(self.h, C.byref(lay)))
        return {"total": tot.value, "x": xb.value,
                "x_layout": {0: "row-major + transposed copy", 1: "row-major", 2: "panel-major (one copy)"}[lay.value]}

    def predict(self, y, xz=None, syn=False, gaussianize=None, theta=None):
        """predict (linearcorex.py:440-441) for this shard: invert(y . X_i Z_j^T), (n_rows, nv_local).  xz: X_i Z_j
        (nv_local, m) of a restored model, None = the resident moments."""
        y, py = self._a(y)
        assert y.ndim == 2 and y.shape[1] == self.m, y.shape
        kind = self.PP_KINDS.get(gaussianize, 0) % 3               # 'empirical' has no inverse: passes through (:437-438)
        mean = std = None
        if kind:
            mean, std = np.ascontiguousarray(theta[0], self.dtype), np.ascontiguousarray(theta[1], self.dtype)
            assert mean.shape == std.shape == (self.nv,)
        pxz = None
        if xz is not None:
            xz, pxz = self._a(xz)
            assert xz.shape == (self.nv, self.m), xz.shape
        out = np.empty((y.shape[0], self.nv), dtype=self.dtype)
        ksec = C.c_double()
        _abi.check(self.lib.lcx_predict(self.h, py, y.shape[0], 1 if syn else 0, pxz, kind,
                                        None if mean is None else _abi.np_ptr(mean), None if std is None else _abi.np_ptr(std),
                                        _abi.np_ptr(out), self.nv, C.byref(ksec)))
        self._predict_kernel_seconds = ksec.value
        return out

    def invert(self, x, gaussianize, theta):
        """invert (linearcorex.py:431-438) of host rows (n_rows, nv_local)."""
        x, px = self._a(x)
        assert x.ndim == 2 and x.shape[1] == self.nv, x.shape
        kind = self.PP_KINDS.get(gaussianize, 0) % 3
        mean = std = None
        if kind:
            mean, std = np.ascontiguousarray(theta[0], self.dtype), np.ascontiguousarray(theta[1], self.dtype)
        out = np.empty_like(x)
        _abi.check(self.lib.lcx_invert(self.h, px, x.shape[0], x.shape[1], kind,
                                       None if mean is None else _abi.np_ptr(mean), None if std is None else _abi.np_ptr(std),
                                       _abi.np_ptr(out), x.shape[1]))
        return out

    def project(self, x):
        x, p = self._a(x)
        assert x.ndim == 2 and x.shape[1] == self.nv
        out = np.empty((x.shape[0], self.m), dtype=self.dtype)
        _abi.check(self.lib.lcx_project(self.h, p, x.shape[0], x.shape[1], _abi.np_ptr(out)))
        return out
