"""Drop-in `Corex` for the non-synergistic Linear CorEx fit path, executed on MI355X.

Public surface = the reference's (linearcorex/linearcorex.py:72-74 constructor, :103-107 fit /
fit_transform, :386 transform, :440 predict, :431 invert, :397 preprocess, :443 get_covariance,
:193 clusters, properties tc/tcs/mis, attributes ws/moments/theta/history ...).  What differs is
where the work happens: X, W and every moment live in HBM for the whole fit and this class only
sequences the dependency levels of `_calculate_moments_ns` / `_update_ns` (:236-334) through the
C ABI (include/lcx.h), reading back a few scalars per line-search trial for control flow.

There is no NumPy fallback: without liblcx_hip.so and a GPU `fit` raises.
"""
from __future__ import annotations

import numpy as np

from .comm import SingleComm
from .preprocess import g, g_inv, mean_impute   # noqa: F401  (re-exported like the reference module)

# The line search a model runs when the caller does not name one (LCX_LINE_SEARCH overrides it: what the parity matrix of
# tests/ uses to run every end-to-end fixture under both re-associations).
DEFAULT_LINE_SEARCH = "exact"

_DETAIL_KEYS = ("MI", "X_i Y_j", "X_i Z_j", "X_i^2 | Y", "I(Y_j ; X)", "I(X_i ; Y)", "TCs",
                "TC_no_overlap", "TC_direct", "additivity")
_DEVICE_KEYS = ("uj", "rho", "ry", "invrho", "rhoinvrho", "Qij", "Si", "Qi-Si^2", "MI", "X_i Z_j",
                "X_i^2 | Y")


class ShardedMomentError(RuntimeError):
    """A per-variable moment of a fit over several ranks was touched before `gather_moments` put it together (dict access never
    issues a collective).  The only error `DeviceMoments.get` / `materialize` treat as "absent": a failing readback (LcxError,
    also a RuntimeError) propagates."""


class _MomentKeys(object):
    """keys() of a `DeviceMoments`: the reference's key list (set-like enough for `sorted`, `in`, `len`, `set(...)`)"""

    def __init__(self, m):
        self._m = m

    def __iter__(self):
        return iter(self._m._listed())

    def __len__(self):
        return len(self._m._listed())

    def __contains__(self, k):
        return k in self._m._listed()

    def __repr__(self):
        return "dict_keys(%r)" % (self._m._listed(),)


class _MomentItems(_MomentKeys):
    """items(): the value of a key still on the device is copied out when the iteration reaches it"""

    def __iter__(self):
        return ((k, self._m[k]) for k in self._m._listed())

    def __contains__(self, kv):
        k, v = kv
        return k in self._m._listed() and self._m[k] is v

    def __repr__(self):
        return "dict_items(%r)" % (self._m._listed(),)


class _MomentValues(_MomentKeys):
    def __iter__(self):
        return (self._m[k] for k in self._m._listed())

    def __contains__(self, v):
        return any(u is v for u in self)

    def __repr__(self):
        return "dict_values(of %r)" % (self._m._listed(),)


class DeviceMoments(dict):
    """`self.moments`: same keys as the reference dict (linearcorex.py:249-287).  Scalars and
    per-factor vectors are stored eagerly; the m x nv arrays stay on the GPU and are copied out on
    first access (they are only valid while the fit state they describe is still resident).

    Several ranks: a per-variable array is sharded over the ranks, and putting it together is a collective.  Dict access
    and pickling never issue one: `Corex._finish` gathers the lazy keys on ALL ranks while the model is small
    (LCX_EAGER_GATHER_ELEMS, default 2**24 elements of n_hidden x n_variables); above that they stay on their shards
    until every rank calls `Corex.gather_moments(keys)`, and touching one before that raises instead of hanging the
    ranks that did not ask."""

    # the reference's dict in its insertion order (:249-273 quick, :277-287 detail): what keys() / items() / iteration / len() list
    _ORDER_QUICK = ("uj", "rho", "ry", "Y_j^2", "invrho", "rhoinvrho", "Qij", "Si", "Qi-Si^2", "TC")
    _ORDER_DETAIL = _ORDER_QUICK + ("MI", "X_i Y_j", "X_i Z_j", "X_i^2 | Y", "I(Y_j ; X)", "I(X_i ; Y)", "TCs", "TC_no_overlap",
                                    "TC_direct", "additivity")

    def __init__(self, owner, generation, eps, eager, details=None):
        super().__init__(eager)
        self._owner, self._gen, self._eps = owner, generation, eps
        self._lazy = set(_DEVICE_KEYS) | {"Y_j^2", "X_i Y_j", "I(X_i ; Y)"}
        # details: the evaluation behind this dict was the reference's quick=False one (None: told by the eager keys)
        self._details = ("TCs" in eager) if details is None else bool(details)

    _REPLICATED = ("uj", "ry")

    # ---- enumeration: every key the reference's dict holds after the same call, whether or not its array has left the device.
    # Listing never copies and never issues a collective; VALUE access materialises (several ranks: ShardedMomentError until
    # `gather_moments`, as with []).  Once the state the dict describes is gone from the device, what was read stays listed.
    def _order(self):
        return self._ORDER_DETAIL if self._details else self._ORDER_QUICK

    def _listed(self):
        order = self._order()
        live = self._resident()
        keys = [k for k in order if dict.__contains__(self, k) or (live and k in self._lazy)]
        seen = set(keys)
        return keys + [k for k in dict.keys(self) if k not in seen]

    def __iter__(self):
        return iter(self._listed())

    def __len__(self):
        return len(self._listed())

    def keys(self):
        return _MomentKeys(self)

    def items(self):
        return _MomentItems(self)

    def values(self):
        return _MomentValues(self)

    def copy(self):
        return dict(self.items())

    def _stored(self):
        """what has been read so far, as a plain dict in the reference's order (no copy from the device, no collective)"""
        return {k: dict.__getitem__(self, k) for k in self._listed() if dict.__contains__(self, k)}

    def __repr__(self):
        return "%s(%s)" % (type(self).__name__, ", ".join(
            "%r: %s" % (k, repr(dict.__getitem__(self, k)) if dict.__contains__(self, k) else "<on device>") for k in self._listed()))

    def _resident(self):
        be = self._owner._backend if self._owner is not None else None
        return be is not None and be.generation == self._gen

    def _fetch(self, key, collective=False, for_key=None):
        if not self._resident():
            raise KeyError("%r: these moments are no longer resident on the device" % key)
        o, be = self._owner, self._owner._backend
        if key == "Y_j^2":
            return o.yscale ** 2 / (1.0 - self["uj"])                      # :262
        if key == "X_i Y_j":
            return self._get(key, "rho", collective).T * np.sqrt(self["Y_j^2"])             # :279
        if key == "I(X_i ; Y)":
            return -0.5 * np.log(self._get(key, "X_i^2 | Y", collective))                   # :283
        if key not in self._REPLICATED and o._comm.world > 1 and not collective:
            raise self._sharded_error(for_key or key)
        val = be.get_moment(0, key, self._eps)
        if key not in self._REPLICATED:
            val = o._gather(val, key)
        return val

    def _sharded_error(self, key):
        return ShardedMomentError("moments[%r] is sharded over %d ranks: call model.gather_moments([%r]) on every rank first "
                            "(dict access never issues a collective)" % (key, self._owner._comm.world, key))

    _DERIVED = {"X_i Y_j": "rho", "I(X_i ; Y)": "X_i^2 | Y"}       # derived key -> the device array it is computed from

    def _get(self, for_key, key, collective):
        if dict.__contains__(self, key):
            return dict.__getitem__(self, key)
        val = self._fetch(key, collective, for_key=for_key)
        dict.__setitem__(self, key, val)
        return val

    def gather(self, keys=None):
        """Collective (every rank must call it with the same keys): materialise sharded keys as full arrays."""
        for k in (keys if keys is not None else sorted(self._lazy)):
            if k in self._lazy and not dict.__contains__(self, k) and self._resident():
                dict.__setitem__(self, k, self._fetch(k, collective=True))
        return self

    def __missing__(self, key):
        if key in self._lazy:
            val = self._fetch(key)
            dict.__setitem__(self, key, val)
            return val
        raise KeyError(key)

    def __contains__(self, key):
        if dict.__contains__(self, key):
            return True
        if key not in self._lazy or not self._resident():
            return False
        single = self._owner._comm.world == 1
        if single or key in self._REPLICATED or key == "Y_j^2":
            return True
        # several ranks: a derived key is there without a collective once the array it is computed from was gathered
        return key in self._DERIVED and dict.__contains__(self, self._DERIVED[key])

    def get(self, key, default=None):
        try:
            return self[key]
        except (KeyError, ShardedMomentError):       # sharded and not gathered: absent as far as dict.get goes
            return default

    def materialize(self, keys=None):
        """Copy out what can be copied out WITHOUT a collective (one rank: every lazy key; several ranks: the replicated
        ones - sharded keys that were not gathered are left out)."""
        for k in (keys or sorted(self._lazy)):
            if k in self:
                try:
                    self[k]
                except (KeyError, ShardedMomentError):
                    pass
        return self._stored()

    def __reduce__(self):
        return (dict, (self.materialize(),))


class SynMoments(DeviceMoments):
    """`self.moments` of the synergistic branch (keys of `_calculate_moments_syn`, linearcorex.py:336-373)."""
    _SYN_DEVICE = {"X_i Y_j": "syn X_i Y_j", "X_i Z_j": "syn X_i Z_j", "X_i^2 | Y": "syn X_i^2 | Y", "rho": "rho",
                   "cy": "cy", "Y_j^2": "Y_j^2", "ry": "ry"}

    # insertion order of `_calculate_moments_syn` (:348-373), which has no quick form
    _ORDER_SYN = ("X_i Y_j", "cy", "Y_j^2", "ry", "rho", "invrho", "rhoinvrho", "Qij", "Qi", "Si", "MI", "X_i Z_j", "X_i^2 | Y", "TCs",
                  "additivity", "TC")

    def __init__(self, owner, generation, eager):
        DeviceMoments.__init__(self, owner, generation, 0, eager, details=True)
        self._lazy = set(self._SYN_DEVICE) | {"invrho", "rhoinvrho", "Qij", "Qi", "Si", "MI"}

    def _order(self):
        return self._ORDER_SYN

    _REPLICATED = ("cy", "Y_j^2", "ry")

    _DERIVED = {"invrho": "rho", "MI": "rho", "Si": "rho", "rhoinvrho": "rho", "Qij": "rho", "Qi": "rho"}

    def _fetch(self, key, collective=False, for_key=None):
        if not self._resident():
            raise KeyError("%r: these moments are no longer resident on the device" % key)
        o, be = self._owner, self._owner._backend
        if key in self._SYN_DEVICE:
            if key not in self._REPLICATED and o._comm.world > 1 and not collective:
                raise self._sharded_error(for_key or key)
            val = be.get_moment(0, self._SYN_DEVICE[key])
            if key in ("X_i Y_j", "X_i Z_j"):                # (nv_local, m): shard axis first
                return o._gather(np.ascontiguousarray(val.T)).T
            return val if key in self._REPLICATED else o._gather(val, key)
        rho = self._get(key, "rho", collective)
        if key == "invrho":
            return 1.0 / (1.0 - rho ** 2)                                   # :360
        if key == "rhoinvrho":
            return rho * self._get(key, "invrho", collective)               # :361
        if key == "Qij":
            return np.dot(self["ry"], self._get(key, "rhoinvrho", collective))              # :362
        if key == "Qi":
            return np.einsum('ki,ki->i', self._get(key, "rhoinvrho", collective), self._get(key, "Qij", collective))    # :363
        if key == "Si":
            return np.sum(rho * self._get(key, "rhoinvrho", collective), axis=0)            # :364
        if key == "MI":
            return -0.5 * np.log1p(-rho ** 2)                               # :366
        raise KeyError(key)

    def __contains__(self, key):
        if dict.__contains__(self, key):
            return True
        if key not in self._lazy or not self._resident():
            return False
        if self._owner._comm.world == 1 or key in self._REPLICATED:
            return True
        return key in self._DERIVED and dict.__contains__(self, self._DERIVED[key])


class Corex(object):
    """Linear Total Correlation Explanation on MI355X (reference docstring: linearcorex.py:22-70).

    Parameters are the reference's.  `gpu` is accepted for signature compatibility and IGNORED: the reference's `gpu=False` selects its
    NumPy path and `gpu=True` its cudamat path (:84-86); this class has exactly one path, the HIP one, and raises `LcxError` where there
    is no MI355X or no liblcx_hip.so - it never computes on the host (BASELINE configs[0], "NumPy CPU reference path", is the reference
    itself / the oracle of tests/, not something this package ships).  Keyword-only extras:
      dtype   working precision on the device: np.float32 (what the reference computes, :108) or
              np.float64 (the precision the 1e-6 get_covariance tolerance is defined in);
      device  HIP device index (default: LOCAL_RANK or 0);
      comm    a `linearcorex_amd.comm.Comm` to shard n_variables over ranks;
      eliminate_synergy  alias of discourage_overlap (the name used by the reference docstring);
      line_search  None (default) = DEFAULT_LINE_SEARCH ("exact"; LCX_LINE_SEARCH overrides).  "exact": every back-tracking trial re-evaluates the moments with two passes over
              X, as the reference does (:321); "linear": X^T.(X.u^T) is linear in u, so the trial
              moments follow from the direction already computed for `_sig` (:301) and cost no pass
              over X - same mathematics, different rounding; the solution is re-evaluated exactly
              every `refresh_every` iterations and at every annealing stage.
              "exact-y": as "exact", except that the trials AFTER the first one of an iteration take X.w_update^T by
              linearity from this iteration's own exact products (one pass over X per such trial instead of two; no drift,
              no re-anchoring - include/lcx.h, lcx_set_trial_reuse).  Needs the line search inside the library.
      f32_gemm  arithmetic of the two passes over X of a float32 fit (include/lcx.h, lcx_set_f32_gemm): "mfma" = float32 MFMA,
              exact float32 products (what None gives unless LCX_F32_GEMM=split is set); "split" = every operand split exactly
              into three bf16 numbers, 6 of the 9 partial products on the bf16 matrix pipe with float32 accumulation: results
              within float32 rounding of "mfma", passes 1.3-1.5 x faster on large shards.  Taken only where the shard supports
              it (panel-major layout, 32 / 64 / 128 padded factors); `self.f32_gemm` tells which one the fit ran.
    """

    # defaults for models pickled by an earlier build
    _ex = None
    _engine_exchange = None
    _f32_gemm_asked = None
    f32_gemm = "mfma"

    def __init__(self, n_hidden=10, max_iter=10000, tol=1e-5, anneal=True, missing_values=None,
                 discourage_overlap=True, gaussianize='standard', gpu=False,
                 verbose=False, seed=None, *, dtype=np.float32, device=None, comm=None,
                 eliminate_synergy=None, line_search=None, refresh_every=16, f32_gemm=None, _backend_factory=None):
        if eliminate_synergy is not None:
            discourage_overlap = bool(eliminate_synergy)
        self.m = n_hidden
        self.max_iter = max_iter
        self.tol = tol
        self.anneal = anneal
        self.eps = 0
        self.missing_values = missing_values
        self.discourage_overlap = discourage_overlap
        self.gaussianize = gaussianize
        self.gpu = gpu                      # kept for signature compatibility; the device path is the only path
        self.yscale = 1.
        np.random.seed(seed)                # same global-RNG side effect as the reference (:89)
        self.verbose = verbose
        if verbose:
            np.set_printoptions(precision=3, suppress=True, linewidth=160)
            print('Linear CorEx with {:d} latent factors'.format(n_hidden))
        self.n_samples, self.nv = 0, 0
        self.ws = np.zeros((0, 0))
        self.moments = {}
        self.theta = None
        self.history = {}
        self.last_update = 0
        self.n_obs = 0
        # device side
        self.dtype = np.dtype(dtype)
        self.device = device
        self._comm = comm if comm is not None else SingleComm()
        self._backend_factory = _backend_factory
        self._backend = None
        self._ex = None
        self._engine_exchange = None
        self._cols = (0, 0)
        self._tc_cur = np.nan
        import os
        # None = the package default (DEFAULT_LINE_SEARCH / LCX_LINE_SEARCH): where "exact-y" cannot run (see _make_backend)
        # a defaulted model takes "exact" instead, a model that was ASKED for "exact-y" refuses
        self._line_search_asked = line_search
        if line_search is None:
            line_search = os.environ.get("LCX_LINE_SEARCH") or DEFAULT_LINE_SEARCH
        if line_search not in ("exact", "linear", "exact-y"):
            raise ValueError("line_search must be 'exact', 'linear' or 'exact-y'")
        self.line_search = self._line_search_wanted = line_search
        if f32_gemm not in (None, "mfma", "split"):
            raise ValueError("f32_gemm must be None, 'mfma' or 'split'")
        self._f32_gemm_asked = f32_gemm         # None: the library's default (float32 MFMA unless LCX_F32_GEMM=split)
        self.f32_gemm = "mfma"                  # what the fit ran, settled in _make_backend
        self.refresh_every = int(refresh_every)
        self._since_exact = 0
        self.stats = {"iterations": 0, "moment_evals": 0, "trials": 0, "invalid_trials": 0}
        self._check_ranks = os.environ.get("LCX_CHECK_RANKS", "0") not in ("", "0")
        # LCX_HOST_LOOP=1: sequence the levels of `_update_ns` from this class (what a caller-owned exchange needs) instead of
        # handing the whole iteration to the engine (lcx_iterate) - same results, kept selectable for tests
        self._in_library = os.environ.get("LCX_HOST_LOOP", "0") in ("", "0")

    def _assert_same_on_all_ranks(self, values, what):
        """Debug aid (LCX_CHECK_RANKS=1): the host decisions of the line search are taken from all-reduced scalars and
        must be bit-identical on every rank - a divergence would leave the ranks in mismatched collectives (a hang).
        Costs two small all-reduces per call."""
        if not self._check_ranks or self._ex is None or self._comm.world == 1:
            return
        import torch
        v = np.asarray(values, dtype=np.float64)
        v = np.where(np.isnan(v), -1.2345e300, v)
        with self._backend.stream_context():
            hi = torch.tensor(v, dtype=torch.float64, device=self._ex[1].device)
            lo = -hi.clone()
            self._comm.allreduce_max(hi)
            self._comm.allreduce_max(lo)
            hi, lo = hi.cpu().numpy(), -lo.cpu().numpy()
        if not (np.array_equal(hi, lo) and np.array_equal(hi, v)):
            raise RuntimeError("rank %d: %s differs across ranks: mine %r, max %r, min %r"
                               % (self._comm.rank, what, v.tolist(), hi.tolist(), lo.tolist()))

    # ------------------------------------------------------------------------------------------
    # backend plumbing
    # ------------------------------------------------------------------------------------------
    def _make_backend(self, n_samples, nv_local, temporary=False):
        """temporary: a short-lived handle beside the fit's own (transform(details=True) of a new batch): same transport, whose
        first-contact self-test the fit's handle has already passed"""
        if self._backend is not None:
            self._backend.close()
        if self._backend_factory is not None:
            be = self._backend_factory(n_samples, nv_local, self.m, self.dtype)
        else:
            from .backend import HipBackend
            dev = self.device
            if dev is None:
                import os
                dev = int(os.environ.get("LOCAL_RANK", "0"))
            be = HipBackend(n_samples, nv_local, self.m, self.dtype, dev)
        self._backend = be
        exchange = getattr(self._comm, "exchange", self._comm.world > 1)
        be.set_world(self._comm.world)
        if exchange and self._comm.world == 1 and hasattr(be, "set_exchange"):
            be.set_exchange(True)
        self._ex = be.exchange_tensors() if exchange else None
        # the exchange steps inside the engine (RCCL communicator of the handle, or a hook for other transports): the levels
        # then all-reduce what they produce themselves and this class issues no collective on the hot path
        if not exchange:
            self._engine_exchange = None
        elif temporary:
            self._engine_exchange = self._comm.bind_engine(be, first_contact=False)
        else:
            self._engine_exchange = self._comm.bind_engine(be)
        # which line search runs - decided here, before any data moves.  "exact-y" lives inside lcx_iterate: it needs the
        # line search in the library (not LCX_HOST_LOOP=1, not the per-trial prints of verbose > 1) and, with several ranks,
        # the exchange inside the engine
        ls = getattr(self, "_line_search_wanted", self.line_search)
        if ls == "exact-y":
            able = (self._in_library and self.verbose <= 1 and hasattr(be, "iterate") and hasattr(be, "set_trial_reuse")
                    and (self._ex is None or bool(self._engine_exchange)))
            if not able:
                if getattr(self, "_line_search_asked", ls) == "exact-y":
                    be.close()
                    self._backend = None
                    raise RuntimeError("line_search='exact-y' runs inside lcx_iterate only (not with LCX_HOST_LOOP=1, verbose > 1, "
                                       "a backend without lcx_iterate or a caller-owned exchange)")
                ls = "exact"
        self.line_search = ls
        if hasattr(be, "set_linear_mode"):
            be.set_linear_mode(ls == "linear")
        if hasattr(be, "set_trial_reuse"):
            be.set_trial_reuse(ls == "exact-y")
        # arithmetic of the X passes of a float32 shard: asked for explicitly, or the library's default (LCX_F32_GEMM)
        if hasattr(be, "set_f32_gemm"):
            asked = getattr(self, "_f32_gemm_asked", None)
            self.f32_gemm = be.set_f32_gemm(asked) if asked is not None else be.f32_gemm()
        return be

    def _allreduce(self, tensor):
        """Exchange step: all-reduce on the stream the engine's kernels run on - unless the engine does it itself."""
        if self._engine_exchange:
            return
        with self._backend.stream_context():
            self._comm.allreduce(tensor)

    def _y_main(self):
        """elements of the Y exchange buffer a level exchange covers: [Y (n_pad x m_pad) | tail (m_pad^2)]; shards that can run
        the merged pass keep Y_g behind it (include/lcx.h, lcx_exchange_layout), which only the engine's own exchange touches"""
        be = self._backend
        n = getattr(be, "_ybuf_main", None)
        if n is None:
            g = be.geometry()
            n = g["n_pad"] * g["m_pad"] + g["m_pad"] ** 2 if "n_pad" in g else len(self._ex[0])
            be._ybuf_main = n
        return n

    def _xy(self):
        if self._ex is not None and not self._engine_exchange:
            self._allreduce(self._ex[0][:self._y_main()])

    def _x_scalars(self):
        """Scalar exchange of a moment evaluation: the two TC sums, the tangent partial of the current direction
        and the H partial of the evaluated set (include/lcx.h, lcx_read_sbuf) - one all-reduce."""
        if self._ex is not None:
            off, cnt = self._backend.sbuf_ranges()[0]
            self._allreduce(self._ex[1][off:off + cnt])

    def _x_h(self):
        if self._ex is not None:
            mp2 = be_mp2(self._backend)
            self._allreduce(self._ex[1][8:8 + mp2])

    def _x_detail(self):
        if self._ex is not None:
            off, cnt = self._backend.sbuf_ranges()[1]
            self._allreduce(self._ex[1][off:off + cnt])

    def _xtail(self):
        """all-reduce only the m_pad^2 tail of the Y exchange buffer (W.W^T partials)."""
        if self._ex is not None and not self._engine_exchange:
            n = self._y_main()
            self._allreduce(self._ex[0][n - be_mp2(self._backend):n])

    def _gather(self, local, key=None):
        """Per-variable arrays are sharded on the last (m x nv, nv) or first (nv x m) axis."""
        if self._comm.world == 1:
            return local
        if key == "X_i Z_j":
            return self._comm.gather_columns(np.ascontiguousarray(local.T), self.nv, self._ex[1]).T
        return self._comm.gather_columns(local, self.nv, self._ex[1])

    # ------------------------------------------------------------------------------------------
    # public API
    # ------------------------------------------------------------------------------------------
    def fit_transform(self, x):
        """`fit(x)` then `transform(x)` (:103-105).  The preprocessed x is still resident after the fit, so the latent factors are
        one pass X~ . ws^T over it - the same numbers `transform(x)` produces from a second upload of x (same theta; imputation and
        the rank transform are functions of this very batch) without moving the matrix over PCIe again."""
        self.fit(x)
        y = self.transform_fitted()
        return self.transform(x) if y is None else y

    def transform_fitted(self, details=False):
        """Latent factors of the data this model was fitted on, taken from the shard that is still resident on the device
        (what `transform(x_fit)` returns; the stacking recipe vis_corex.py:542 asks for exactly this).  None when the data
        are not resident any more (a model restored from a pickle): call `transform(x)` then.
        details=True: `(y, moments)` as `transform(x_fit, details=True)` (:392-394) - the full moments of the fitted data under the
        final weights are what `fit` left in `self.moments` (:163), so nothing is evaluated and no second handle is built (at the size
        of BASELINE configs[3] on one GPU a second resident copy of X would not fit)."""
        be = self._backend
        if not getattr(self, "_x_resident", False) or be is None or not hasattr(be, "project_resident") or self.ws.size == 0:
            return None
        if self.gaussianize == 'empirical':       # the reference's second preprocess of x says these again (:416-417, :425)
            print("Warning: correct inversion/transform of empirical gauss transform not implemented.")
        elif self.gaussianize == 'standard' and self.verbose and getattr(self, "_fit_max_abs", 0.0) > 6:
            print("Warning: outliers more than 6 stds away from mean. Consider using gaussianize='outliers'")
        y = be.project_resident()                 # lcx_moments_a(0): with the exchange in the engine already summed over the ranks
        if self._ex is not None and not self._engine_exchange:
            import torch
            with be.stream_context():
                t = torch.from_numpy(y).to(self._ex[1].device)
                self._comm.allreduce(t)
                y = t.cpu().numpy()
        return (y, self.moments) if details else y

    def fit(self, x):
        if self.m is None:
            raise NotImplementedError("n_hidden=None (pick_n_hidden) is broken in the reference (SURVEY.md §2 #14)")
        x = np.asarray(x, dtype=self.dtype)                    # reference casts to float32 (:108)
        self.n_samples, self.nv = x.shape
        c0, c1 = self._comm.shard(self.nv)
        self._cols = (c0, c1)
        be = self._make_backend(self.n_samples, c1 - c0)
        self._x_resident = True
        # preprocess(x, fit=True) (:109, :397-429) happens on the device, per shard, while uploading - 'empirical'
        # (:424-426: per-column rank -> normal quantile) included: a segmented sort of the transposed copy
        if self.gaussianize == 'empirical':
            print("Warning: correct inversion/transform of empirical gauss transform not implemented.")     # :425
        theta, n_obs, max_abs = be.upload_preprocess(np.ascontiguousarray(x[:, c0:c1]), self.gaussianize,
                                                     self.missing_values, None)
        if self.gaussianize in ('standard', 'outliers'):
            self.theta = (self._gather(theta[0]), self._gather(theta[1]))
        self.n_obs = self._gather(n_obs) if self.missing_values is not None else len(x)
        if self.gaussianize == 'standard' and self.verbose:
            if self._ex is not None:
                import torch
                with be.stream_context():
                    t = torch.tensor([max_abs], dtype=torch.float64, device=self._ex[1].device)
                    self._comm.allreduce_max(t)
                    max_abs = float(t.item())
            self._fit_max_abs = max_abs
            if max_abs > 6:
                print("Warning: outliers more than 6 stds away from mean. Consider using gaussianize='outliers'")
        del x
        return self._fit_resident()

    def fit_generated(self, n_samples, n_variables, seed=1, kind=0, n_groups=1):
        """Fit on synthetic data generated and standardised on the device (sizes that cannot be
        staged on the host: BASELINE.json configs 3-4).  kind 0 = iid N(0,1), 1 = planted groups."""
        self.n_samples, self.nv = int(n_samples), int(n_variables)
        c0, c1 = self._comm.shard(self.nv)
        self._cols = (c0, c1)
        be = self._make_backend(self.n_samples, c1 - c0)
        self._x_resident = True
        be.generate_x(seed, kind, n_groups, c0)
        self.theta = (np.zeros(self.nv, self.dtype), np.ones(self.nv, self.dtype))
        self.n_obs = self.n_samples
        return self._fit_resident()

    def _attach_shard(self, x_local, nv_total):
        """Make an already preprocessed column block resident as this rank's shard (bench / tests):
        x_local is n_samples x (c1-c0) for (c0, c1) = comm.shard(nv_total)."""
        self.n_samples, self.nv = int(x_local.shape[0]), int(nv_total)
        self._cols = self._comm.shard(self.nv)
        assert x_local.shape[1] == self._cols[1] - self._cols[0]
        be = self._make_backend(self.n_samples, x_local.shape[1])
        self._x_resident = True
        be.upload_x(np.ascontiguousarray(x_local, dtype=self.dtype))
        if self.theta is None:
            self.theta = (np.zeros(self.nv, self.dtype), np.ones(self.nv, self.dtype))
        return be

    def _init_weights(self):
        """Random start, normalised so that uj = 0.01 (:113-122).  Returns the annealing schedule."""
        if self.m is None:
            raise NotImplementedError("n_hidden=None (pick_n_hidden) is broken in the reference (SURVEY.md §2 #14)")
        be, (c0, c1) = self._backend, self._cols
        anneal_schedule = [0.]
        if self.ws.size == 0:
            w = np.random.randn(self.m, self.nv).astype(self.dtype)          # :116
            be.set_ws(np.ascontiguousarray(w[:, c0:c1]))
            del w
            be.moments_a(0)
            self._xy()
            be.init_scale_ws()                                                # ws /= 10*_norm (:117)
            if self.anneal:
                anneal_schedule = [0.6 ** k for k in range(1, 7)] + [0]
        else:
            be.set_ws(np.ascontiguousarray(np.asarray(self.ws, dtype=self.dtype)[:, c0:c1]))
        self.eps = 0
        self.moments = self._calculate_moments(quick=True)                    # :122
        return anneal_schedule

    def _begin_stage(self, i_eps, eps):
        """Change the annealing parameter and rescale so that uj < 1 still holds (:127-134)."""
        eps0 = self.eps
        self.eps = eps
        if i_eps > 0:
            self._backend.rescale_ws(eps0, eps)                               # :129-133
        self.moments = self._calculate_moments(quick=False, details=False)    # :134

    def _iterate(self, more=False):
        """One pass of the loop body (:137-151).  Returns delta, or None if the solution went invalid.
        more: the caller will iterate again unless this iteration converges (lets the engine start the next one early)."""
        last_tc = self.tc
        self.moments = self._update_ns(more=more)                             # :139
        if not self.moments or not np.isfinite(self.tc):                      # :144-149
            try:
                print("Error: TC is no longer finite: {}".format(self.tc))
            except Exception:
                print("Error... updates giving invalid solutions?")
                return None
        delta = np.abs(self.tc - last_tc)
        self.update_records(self.moments, delta)
        self.stats["iterations"] += 1
        return delta

    def _finish(self):
        """Detail moments, sort factors by TC, recompute (:160-163)."""
        be = self._backend
        self.moments = self._calculate_moments(quick=False, details=True)     # :160
        order = np.argsort(-self.moments["TCs"])                              # :161
        be.permute_factors(order)                                             # :162
        self.moments = self._calculate_moments(quick=False, details=True)     # :163
        self.ws = self._gather(be.get_ws(0))
        self._gather_if_small()
        return self

    def _gather_if_small(self):
        """Several ranks, end of `fit` (every rank is here): put the sharded moments together now if that is cheap, so
        that later dict access / pickling on a single rank needs no collective (see DeviceMoments)."""
        if self._comm.world > 1 and isinstance(self.moments, DeviceMoments):
            import os
            limit = int(os.environ.get("LCX_EAGER_GATHER_ELEMS", str(1 << 24)))
            if self.m * self.nv <= limit:
                self.moments.gather()

    def gather_moments(self, keys=None):
        """Collective: every rank calls it with the same `keys` (default: all); afterwards those keys of `self.moments`
        are full (n_hidden x n_variables) arrays on every rank, as in the reference."""
        if isinstance(self.moments, DeviceMoments):
            self.moments.gather(keys)
        return self.moments

    # ------------------------------------------------------------------------------------------
    # synergistic branch: discourage_overlap=False (linearcorex.py:119-121, :141, :336-384)
    # ------------------------------------------------------------------------------------------
    def _calculate_moments_syn(self, which=0, details=True):
        """`_calculate_moments_syn` (:336-373) of the weights of set `which`; returns the moments dict when
        which == 0, else only the state scalars (the caller accepts the set afterwards)."""
        be = self._backend
        be.moments_a(which)                          # Y_partial = X.W^T (:347)
        self._xy()
        be.syn_moments_b(which, self.yscale)         # cy, ry, X^T.Y, rho, X_i Z_j, X_i^2|Y, partial sums
        self._x_detail()
        be.syn_moments_c(which)                      # TC (:373)
        self.stats["moment_evals"] += 1
        st = be.read_state(which)
        if which != 0:
            return st
        return self._syn_moments_dict(st, details)

    def _syn_moments_dict(self, st, details):
        be = self._backend
        eager = {"TC": np.float64(st[0])}
        if details:
            sums = be.read_sbuf(self.m + 3)
            yj2 = be.get_moment(0, "Y_j^2").astype(np.float64)
            iyx = 0.5 * np.log(yj2) - 0.5 * np.log(self.yscale ** 2)           # :369
            eager["TCs"] = sums[:self.m] - iyx                                 # :371
            eager["additivity"] = sums[self.m + 2] - sums[self.m + 1]          # :372
        self._tc_cur = st[0]
        return SynMoments(self, be.generation, eager)

    def _update_syn(self, x=None, eta=0.5):
        """One damped fixed-point step (:375-384); the new weights stay on the device."""
        be = self._backend
        be.syn_update_a()                            # H partial (:378)
        self._x_h()
        be.syn_update_b(eta)                         # ws' (:380-382) -> set 1
        st = self._calculate_moments_syn(which=1)    # :383
        be.accept_trial()
        return self._syn_moments_dict(st, details=False)

    def _fit_resident_syn(self):
        if self.m is None:
            raise NotImplementedError("n_hidden=None (pick_n_hidden) is broken in the reference (SURVEY.md §2 #14)")
        be, (c0, c1) = self._backend, self._cols
        if self.ws.size == 0:
            w = np.random.randn(self.m, self.nv) * self.yscale ** 2 / np.sqrt(self.nv)      # :121 (float64 draws)
        else:
            w = np.asarray(self.ws)
        be.set_ws(np.ascontiguousarray(w[:, c0:c1], dtype=self.dtype))
        del w
        self.eps = 0
        self.moments = self._calculate_moments_syn(details=False)              # :122 (and :134: same weights)
        delta = 0.
        for i_loop in range(self.max_iter):                                    # :136-155
            last_tc = self.tc
            self.moments = self._update_syn(eta=0.1)                           # :141
            if not np.isfinite(self.tc):
                print("Error: TC is no longer finite: {}".format(self.tc))
            delta = np.abs(self.tc - last_tc)
            self.update_records(self.moments, delta)
            self.stats["iterations"] += 1
            if delta < self.tol:
                if self.verbose:
                    print('{:d} iterations to tol: {:f}, TC={:f}'.format(i_loop, self.tol, self.tc))
                break
        else:
            if self.verbose:
                print("Warning: Convergence not achieved in {:d} iterations. "
                      "Final delta: {:f}".format(self.max_iter, float(delta)))
        self.moments = self._calculate_moments_syn()                           # :160
        order = np.argsort(-self.moments["TCs"])                               # :161
        be.permute_factors(order)                                              # :162
        self.moments = self._calculate_moments_syn()                           # :163
        self.ws = self._gather(be.get_ws(0))
        self._gather_if_small()
        return self

    def _fit_resident(self):
        if not self.discourage_overlap:
            return self._fit_resident_syn()
        self.stage_iterations = []             # iterations spent in each annealing stage (not in the reference; for reports)
        for i_eps, eps in enumerate(self._init_weights()):
            self._begin_stage(i_eps, eps)
            delta = 0.
            self.stage_iterations.append(0)
            for i_loop in range(self.max_iter):
                delta = self._iterate(more=i_loop + 1 < self.max_iter)
                if delta is None:
                    self.ws = self._gather(self._backend.get_ws(0))
                    return self
                self.stage_iterations[-1] = i_loop + 1
                if delta < self.tol:
                    if self.verbose:
                        print('{:d} iterations to tol: {:f}, TC={:f}'.format(i_loop, self.tol, self.tc))
                    break
            else:
                if self.verbose:
                    print("Warning: Convergence not achieved in {:d} iterations. "
                          "Final delta: {:f}".format(self.max_iter, float(delta)))
        return self._finish()

    def update_records(self, moments, delta):
        """History book-keeping (linearcorex.py:166-175)."""
        self.history.setdefault("TC", []).append(moments["TC"])
        if self.verbose > 1:
            print("TC={:.3f}\tadd={:.3f}\tdelta={:.6f}".format(moments["TC"], moments.get("additivity", 0), delta))
        if self.verbose:
            self.history.setdefault("additivity", []).append(moments.get("additivity", 0))
            self.history.setdefault("TCs", []).append(moments.get("TCs", np.zeros(self.m)))

    @property
    def tc(self):
        return self.moments["TC"]

    @property
    def tcs(self):
        return self.moments["TCs"]

    @property
    def mis(self):
        return - 0.5 * np.log1p(-self.moments["rho"] ** 2)

    def clusters(self):
        return np.argmax(np.abs(self.ws), axis=0)

    # ------------------------------------------------------------------------------------------
    # hot path: the reference's private methods, re-expressed as device levels + exchanges
    # ------------------------------------------------------------------------------------------
    def _moments_levels(self, which, quick):
        be = self._backend
        be.moments_a(which)                 # Y_partial = X.W^T (:247), W.W^T partial
        self._xy()
        be.moments_b(which, self.eps, quick)  # uj, early exit, X^T.Y, rho ... Qi-Si^2, TC partial sums, H partial
        self._x_scalars()
        be.moments_c(which)                 # TC
        self.stats["moment_evals"] += 1

    def _scalar(self, x):
        return self.dtype.type(x)

    def _calculate_moments(self, quick=False, details=False):
        """`_calculate_moments_ns(x, self.ws, quick)` (:236-288) on the current weights."""
        be = self._backend
        self._moments_levels(0, quick)
        self._since_exact = 0
        st = be.read_state(0)
        if st[2] != 0:
            return False                                                       # :250-251
        self._tc_cur = st[0]
        eager = {"TC": self._scalar(st[0])}
        if details:
            be.moments_detail(0)
            self._x_detail()
            sums = be.read_sbuf(self.m + 3)
            uj = be.get_moment(0, "uj")
            eager["uj"] = uj
            ysq = self.yscale ** 2 / (1.0 - uj)
            # note: under NumPy >= 2 the reference gets float64 here (np.log of a Python float, :282)
            iyx = 0.5 * np.log(ysq) - 0.5 * np.log(self.yscale ** 2)
            eager["I(Y_j ; X)"] = iyx
            eager["TCs"] = sums[:self.m] - iyx                                 # :284
            eager["TC_no_overlap"] = sums[self.m] - iyx.sum()                  # :285
            eager["TC_direct"] = sums[self.m + 1] - iyx                        # :286
            eager["additivity"] = self._scalar(sums[self.m + 2] - sums[self.m + 1])   # :287
        return DeviceMoments(self, be.generation, self.eps, eager)

    def _calculate_moments_ns(self, x=None, ws=None, quick=False):
        if x is not None or ws is not None:
            raise NotImplementedError("moments are evaluated on the resident X / ws only")
        return self._calculate_moments(quick=quick, details=not quick)

    _SINGULAR_WARNING = ('Warning: covariance is nearly singular and this causes a loss of numerical precision.'
                         'For this reason, we can no longer find an update that increases the objective. '
                         'Hopefully this is a good solution. If not, this is caused by having many variables that are '
                         'near duplicates. You could try again with the duplicates removed to look for other structure.')

    def _update_ns_in_library(self, more):
        """`_update_ns` (:290-334) as ONE call into the engine (lcx_iterate): line-search decisions are taken in native code
        right where the trial's scalars arrive, and the next iteration's first launches are already queued when this returns."""
        be, m = self._backend, self.moments
        r = be.iterate(self.eps, self.tol, self._tc_cur, more)
        self._assert_same_on_all_ranks(r[:6], "lcx_iterate (status, TC, tangent, trials, invalid trials, too small)")
        status = int(r[0])
        self.stats["trials"] += int(r[3])
        self.stats["invalid_trials"] += int(r[4])
        self.stats["moment_evals"] += int(r[6])
        if status == 1:                                                        # :306-311
            print(self._SINGULAR_WARNING)
            return m
        if r[5] and self.verbose:                                              # :316-319
            print('Warning: step size becoming too small')
        if status == 2:                                                        # (w_update, False), reported by fit (:144-149)
            return False
        self._tc_cur = r[1]
        return DeviceMoments(self, be.generation, self.eps, {"TC": self._scalar(r[1])})

    def _update_ns(self, x=None, more=False):
        """One fixed-point iteration with back-tracking (:290-334).  Returns the new moments (the
        new weights stay on the device)."""
        be = self._backend
        m = self.moments
        if self._in_library and (self._ex is None or self._engine_exchange) and self.line_search in ("exact", "exact-y") \
                and self.verbose <= 1 and hasattr(be, "iterate"):
            self._iterated_in_library = True
            return self._update_ns_in_library(more)
        assert self.line_search != "exact-y"         # (refused or replaced in _make_backend, before any data moved)
        # H (:294) is already global: it came with the scalar exchange of the evaluation that produced set 0
        be.update_b(self.eps)                # grad (:296-300), Bj partial, Y_g partial
        self._xy()
        be.update_c(self.eps)                # X^T.Y_g, sig_grad, update, tangent partial (:301-305)
        be.update_d()                        # the tangent becomes global with the first trial's scalar exchange
        tc_cur = self._tc_cur
        linear = self.line_search == "linear"
        update_tangent = None
        eta = 1.
        last = None                          # (invalid?, tc) of the last evaluated trial
        while True:
            if eta < min(self.tol, 1e-10):                                     # :316-319
                if self.verbose:
                    print('Warning: step size becoming too small')
                break
            if linear:
                be.trial_linear_a(eta)                                         # :320, Y' = Y + eta*Y(update)
                self._xtail()
                be.trial_linear_b(self.eps, eta)                               # D' = D + eta*D(update) ... TC sums, H
                self._x_scalars()
                be.moments_c(1)
                self.stats["moment_evals"] += 1
            else:
                be.make_trial(eta)                                             # :320
                self._moments_levels(1, True)                                  # :321
            self.stats["trials"] += 1
            st = be.read_state(1)
            self._assert_same_on_all_ranks([st[0], st[1], st[2], st[3], eta], "trial state (TC, max uj, invalid, tangent, eta)")
            if update_tangent is None:
                update_tangent = st[3]             # the trial's scalars carry the tangent of its direction
                if update_tangent >= 0:                                        # :306-311
                    self.stats["trials"] -= 1      # the speculative trial is discarded
                    be.update_a()                  # ... and so is its H: restore the H of the kept solution
                    self._x_h()
                    print(self._SINGULAR_WARNING)
                    return m
            invalid, tc_new = st[2] != 0, st[0]
            last = (invalid, tc_new)
            if invalid:                                                        # :322-326
                self.stats["invalid_trials"] += 1
                eta *= 0.5
                if self.verbose > 1:
                    print('back:{:.7f}'.format(eta))
                continue
            wolfe1 = -tc_new <= -tc_cur + 0.1 * eta * update_tangent           # :327
            if not wolfe1:
                eta *= 0.5
                if self.verbose > 1:
                    print('wolfe1:{:.7f}'.format(eta))
                continue
            break
        if last is None or last[0]:
            # step size underflow right after an invalid trial: the reference returns
            # (w_update, False) here and `fit` reports it (:144-149)
            be.accept_trial()
            return False
        be.accept_trial()                                                      # :334 / :139
        self._tc_cur = last[1]
        if linear:
            self._since_exact += 1
            if self._since_exact >= self.refresh_every:
                # re-anchor the recurrences Y += eta*Y(update), D += eta*D(update) on a fresh evaluation
                fresh = self._calculate_moments(quick=False, details=False)
                self.stats["refreshes"] = self.stats.get("refreshes", 0) + 1
                return fresh
        return DeviceMoments(self, be.generation, self.eps, {"TC": self._scalar(last[1])})

    # ------------------------------------------------------------------------------------------
    # outputs
    # ------------------------------------------------------------------------------------------
    def transform(self, x, details=False):
        """x -> latent factors Y = x~ . ws^T (:386-395).
        details=True (:392-394): also the full moments of THIS batch under the fitted weights.  The batch becomes the resident shard
        of a temporary handle of its own (device memory for a second matrix of the batch's size while the call lasts; with several
        ranks the handle joins the exchange like the fit's did, without repeating the transport's self-test).  For the data the
        model was fitted on, `transform_fitted(details=True)` returns the same pair from what is already resident."""
        x = np.asarray(x, dtype=self.dtype)
        ns, nv = x.shape
        assert self.nv == nv, "Incorrect number of variables in input, %d instead of %d" % (nv, self.nv)
        be = self._resident_backend()
        c0, c1 = self._cols
        if self.missing_values is None and self.gaussianize != 'empirical':
            theta = None
            if self.gaussianize in ('standard', 'outliers'):
                theta = (self.theta[0][c0:c1], self.theta[1][c0:c1])
            else:
                theta = (np.zeros(c1 - c0, self.dtype), np.ones(c1 - c0, self.dtype))
            y = be.project_raw(np.ascontiguousarray(x[:, c0:c1]), self.gaussianize, theta)   # preprocess on the device
            engine_summed = True
        else:
            # imputation needs the column means of the whole new batch (:403), the rank transform its whole columns (:424-426):
            # the batch becomes the resident shard of a handle of its own and goes through the same device preprocess as a fit
            y = self._project_new_batch(np.ascontiguousarray(x[:, c0:c1]))
            engine_summed = False
        if self._ex is not None and not (self._engine_exchange and engine_summed):   # (else the engine summed the partials itself)
            import torch
            with be.stream_context():
                t = torch.from_numpy(y).to(self._ex[1].device)
                self._comm.allreduce(t)
                y = t.cpu().numpy()
        if details:
            # :392-394: the FULL moments of the batch that was handed in (not of the fitted data), with the fitted weights
            return y, self._moments_of_batch(np.ascontiguousarray(x[:, c0:c1]))
        return y

    def _moments_of_batch(self, x_local):
        """`self._calculate_moments(x, self.ws)` of `transform(x, details=True)` (:392-394): x is preprocessed with the fitted
        theta (:389), becomes the resident shard of a handle of its own, and the levels of a full evaluation run on it with the
        fitted W - divided, as the reference divides, by the sample count of the FIT (`self.n_samples`, :249 / :260 / :355:
        lcx_set_sample_divisor), whatever the batch's own row count is.  Returns a plain dict (the handle is gone afterwards);
        several ranks: a collective, the per-variable arrays come back gathered."""
        w_local = self._backend.get_ws(0)
        sh = Corex.__new__(Corex)                  # same hyper-parameters, theta, eps, comm; its own handle, stats and moments
        sh.__dict__.update(self.__dict__)          # (not copy.copy: that goes through __getstate__ and drops comm / factory)
        sh._backend, sh._ex, sh._engine_exchange = None, None, None
        sh.moments, sh.history = {}, {}
        sh.stats = dict.fromkeys(self.stats, 0)
        be = sh._make_backend(x_local.shape[0], x_local.shape[1], temporary=True)
        try:
            if self.gaussianize == 'empirical':
                print("Warning: correct inversion/transform of empirical gauss transform not implemented.")     # :425
            be.upload_preprocess(x_local, self.gaussianize, self.missing_values, self._theta_local())
            be.set_ws(w_local)
            be.set_sample_divisor(self.n_samples)
            mo = sh._calculate_moments_syn() if not self.discourage_overlap else sh._calculate_moments(quick=False, details=True)
            if self._comm.world > 1:
                mo.gather()
            out = mo.materialize()
            # derived keys the reference's dict holds as well
            for k in ("Y_j^2", "X_i Y_j", "I(X_i ; Y)", "invrho", "Qi"):
                if k not in out and k in mo:
                    out[k] = mo[k]
            return out
        finally:
            be.close()
            sh._backend = None

    def _project_new_batch(self, x_local):
        """x~ . ws^T of a new batch whose preprocessing needs whole columns; per-shard partial like `project_raw`."""
        be = self._backend
        if self._backend_factory is not None:
            tmp = self._backend_factory(x_local.shape[0], x_local.shape[1], self.m, self.dtype)
        else:
            from .backend import HipBackend
            tmp = HipBackend(x_local.shape[0], x_local.shape[1], self.m, self.dtype, be.device)
        try:
            if self.gaussianize == 'empirical':
                print("Warning: correct inversion/transform of empirical gauss transform not implemented.")     # :425
            tmp.upload_preprocess(x_local, self.gaussianize, self.missing_values, self._theta_local())
            tmp.set_ws(be.get_ws(0))
            return tmp.project_resident()
        finally:
            tmp.close()

    def preprocess(self, x, fit=False):
        """Per-marginal standardisation (:397-429) of a host array, on the host: the reference's public helper.  `fit` /
        `transform` do not call it - they preprocess on the device (lcx_upload_preprocess / lcx_project_raw)."""
        from .preprocess import preprocess as _pp
        x, self.theta, self.n_obs = _pp(x, self.theta if not fit else None, self.gaussianize,
                                        self.missing_values, verbose=self.verbose)
        return x

    def _theta_local(self):
        c0, c1 = self._cols
        if self.gaussianize in ('standard', 'outliers'):
            return (np.asarray(self.theta[0])[c0:c1], np.asarray(self.theta[1])[c0:c1])
        return None

    def invert(self, x):
        """Undo the preprocessing (:431-438), on the device (rows staged through the shard's GPU).  Several ranks: a
        collective, like every call that returns a full (.., n_variables) array."""
        if self.gaussianize not in ('standard', 'outliers'):
            return x
        x = np.asarray(x, dtype=self.dtype)
        if x.ndim == 1:
            return self.invert(x[np.newaxis])[0]
        be = self._resident_backend()
        c0, c1 = self._cols
        return self._gather(be.invert(np.ascontiguousarray(x[:, c0:c1]), self.gaussianize, self._theta_local()))

    def predict(self, y):
        """:440-441: invert(y . X_i Z_j^T) - the rank-n_hidden product and the inverse marginal map run on the device
        (lcx_predict), the (n_rows, n_variables) result is staged back in row blocks."""
        y = np.asarray(y, dtype=self.dtype)
        if y.ndim == 1:
            return self.predict(y[np.newaxis])[0]
        be = self._resident_backend()
        c0, c1 = self._cols
        live = isinstance(self.moments, DeviceMoments) and self.moments._resident() and getattr(self, "_x_resident", True)
        xz = None if live else np.ascontiguousarray(self._host_moment("X_i Z_j")[c0:c1])
        out = be.predict(y, xz, syn=not self.discourage_overlap, gaussianize=self.gaussianize, theta=self._theta_local())
        return self._gather(out)

    def _host_moment(self, key):
        """A moment a restored (unpickled) model needs on the host; absent if the fit was sharded and the key never gathered."""
        m = self.moments
        if isinstance(m, DeviceMoments) or key in m:
            return m[key]
        raise RuntimeError("moments[%r] is not part of this model: it was fitted on several ranks and pickled before "
                           "model.gather_moments([%r]) was called on every rank (sharded moments are not gathered "
                           "implicitly above LCX_EAGER_GATHER_ELEMS elements)" % (key, key))

    def get_covariance(self, rows=None):
        """Covariance estimate (:443-455), nv x nv: a rank-n_hidden product on the device for both branches.

        rows (not in the reference): None = the whole matrix; (start, stop) / slice / range = that block of rows,
        (stop - start, nv) - what a model of 10^5 variables can still hand out.  Several ranks: every rank holds the columns of
        its own variables; the call is a collective and every rank gets the same (rows, nv) array."""
        be = self._resident_backend(need_moments=True)
        std = np.asarray(self.theta[1], dtype=self.dtype)
        syn = not self.discourage_overlap
        if rows is None and self._comm.world == 1:
            return be.covariance_syn(std) if syn else be.covariance(self.eps, std)     # :452-455 / :446-451
        if rows is None:
            r0, r1 = 0, self.nv
        elif isinstance(rows, (slice, range)):
            r0, r1, step = rows.indices(self.nv) if isinstance(rows, slice) else (rows.start, rows.stop, rows.step)
            if step != 1:
                raise ValueError("get_covariance(rows=...): a contiguous block of rows")
        else:
            r0, r1 = (int(t) for t in rows)
        if not 0 <= r0 <= r1 <= self.nv:
            raise ValueError("get_covariance(rows=(%d, %d)): outside [0, %d]" % (r0, r1, self.nv))
        if r0 == r1:                       # an empty block of rows: the same (0, nv) array from one rank and from several
            return np.empty((0, self.nv), dtype=self.dtype)
        if self._comm.world == 1:
            return be.covariance_rows(self.eps, std, r0, r1 - r0, syn)
        return self._covariance_rows_sharded(r0, r1, std, syn)

    def _covariance_rows_sharded(self, r0, r1, std, syn):
        """Rows [r0, r1) of the covariance over several ranks.  cov_ik = sum_j a_ij b_kj (a = b = z = rhoinvrho / (1 + Si),
        divided by 1 - eps^2, :447-448; or a = X_i Z_j, b = X_i Y_j, :453), diagonal 1, times std_i std_k.  The row operand of
        the requested variables is put together from its owners by one all-reduce of (rows x n_hidden) numbers per block;
        every rank multiplies it with the operand of its own columns on the device (the rank-n_hidden product of lcx_predict,
        kind 0) and the column blocks are gathered."""
        import torch
        be, (c0, c1) = self._backend, self._cols
        if syn:                                                                # :453
            a_loc = np.asarray(be.get_moment(0, "syn X_i Z_j"), self.dtype)
            b_loc = np.asarray(be.get_moment(0, "syn X_i Y_j"), self.dtype)
            denom = 1.0
        else:                                                                  # :447-448
            z = be.get_moment(0, "rhoinvrho") / (1 + be.get_moment(0, "Si"))
            a_loc = b_loc = np.ascontiguousarray(z.T, self.dtype)
            denom = 1.0 - self.eps ** 2
        b_scaled = np.ascontiguousarray(b_loc * std[c0:c1, np.newaxis], self.dtype)
        out = np.empty((r1 - r0, self.nv), dtype=self.dtype)
        block = 4096
        for s in range(r0, r1, block):
            e = min(r1, s + block)
            a = np.zeros((e - s, self.m), dtype=self.dtype)
            lo, hi = max(s, c0), min(e, c1)
            if lo < hi:                                                        # the rows this rank owns
                a[lo - s:hi - s] = a_loc[lo - c0:hi - c0] * (std[lo:hi, np.newaxis] / self.dtype.type(denom))
            with be.stream_context():
                t = torch.from_numpy(a).to(self._ex[1].device)
                self._comm.allreduce(t)
                a = t.cpu().numpy()
            blk = be.predict(a, xz=b_scaled)                                   # (e - s, nv_local) = a . b_scaled^T
            if lo < hi:
                blk[np.arange(lo - s, hi - s), np.arange(lo - c0, hi - c0)] = std[lo:hi] ** 2      # fill_diagonal (:449 / :454)
            out[s - r0:e - r0] = self._gather(blk)
        return out

    # ------------------------------------------------------------------------------------------
    # persistence (vis_corex.py:549 pickles the model)
    # ------------------------------------------------------------------------------------------
    def _resident_backend(self, need_moments=False):
        if self._backend is None:
            if self.ws.size == 0:
                raise RuntimeError("model is not fitted")
            # restored from a pickle: bring W (and what get_covariance needs) back to the device.  The data are NOT
            # resident any more, so the handle is created for a single (empty) sample: transform / get_covariance only
            # need W and the per-variable moments, not an n_samples x n_variables shard and its transposed copy
            self._cols = self._comm.shard(self.nv)
            be = self._make_backend(1, self.nv)
            be.set_ws(np.asarray(self.ws, dtype=self.dtype))
            self._moments_restored = False
            self._x_resident = False
        if need_moments and getattr(self, "_moments_restored", True) is False:
            if self.discourage_overlap:                                        # what :447 reads
                self._backend.set_moment(0, "rhoinvrho", self._host_moment("rhoinvrho"))
                self._backend.set_moment(0, "Si", self._host_moment("Si"))
            else:                                                              # what :453 reads
                self._backend.set_moment(0, "syn X_i Z_j", self._host_moment("X_i Z_j"))
                self._backend.set_moment(0, "syn X_i Y_j", self._host_moment("X_i Y_j"))
            self._moments_restored = True
        return self._backend

    def __getstate__(self):
        d = dict(self.__dict__)
        if isinstance(d.get("moments"), DeviceMoments):
            d["moments"] = d["moments"].materialize()
        for k in ("_backend", "_ex", "_backend_factory", "_engine_exchange"):
            d[k] = None
        if not isinstance(d.get("_comm"), SingleComm):
            d["_comm"] = SingleComm()
        return d

    def __setstate__(self, d):
        self.__dict__.update(d)


def pick_n_hidden(data, repeat=1, verbose=False, **kwargs):
    """The reference's helper of the same name (linearcorex.py:458-480), a caller of the fit path: fit models with 1, 2, ... factors
    and record `TC_no_overlap` until the score drops below 0.95 of the best one; returns the list of (score, n).  kwargs go to `Corex`
    (seed, dtype, device, ...).  (`Corex(n_hidden=None)`, which the reference routes here (:111-112), is broken upstream - the list this
    returns is assigned to `self.m` - and refused by this package; the helper itself works there and here.)"""
    def one_score(n_factors):
        out = Corex(n_hidden=n_factors, **kwargs).fit(data)
        score = out.moments["TC_no_overlap"]
        if out._backend is not None:                    # the scan grows one factor per round: do not keep every model's shard resident
            out._backend.close()
            out._backend = None
        return score

    results, best, n_factors = [], -np.inf, 1
    while True:
        score = max(one_score(n_factors) for _ in range(repeat))
        if verbose:
            print(("n: {}, score: {}".format(n_factors, score)))
        results.append((score, n_factors))
        if score < 0.95 * best:                         # (:474) the first count whose score falls 5 % below the best so far ends the scan
            return results
        best = max(best, score)
        n_factors += 1


def be_mp2(be):
    """number of doubles of H in the scalar exchange buffer (m_pad^2)."""
    mp = getattr(be, "m_pad", None)
    if mp is None:
        mp = be.geometry()["m_pad"]
        be.m_pad = mp
    return mp * mp
