"""Eight ranks of the engine in ONE process, exchanging through an ASYNCHRONOUS on-device transport.

The multi-process tests (tests/test_distributed_gpu.py) reach the library's exchange through gloo, whose all_reduce blocks the
host: a missing stream dependency between a level's kernels and its collective could hide there.  RCCL refuses two ranks per
device, so its asynchronous launches are only exercised in a one-rank group.  Here every rank is a host thread with its own
engine handle on GPU 0, and the exchange hook (include/lcx.h, lcx_set_exchange_hook) is a stream-ordered sum like RCCL's: the
rank's buffer is ready behind an event on the stream the library hands over, every rank's stream waits for all ready events, forms
the sum of the N buffers in rank order into a scratch tensor (rank-identical bits), and copies it back once every rank has read
every input - all enqueued, nothing waits on the host but the hand-over of the event handles.  The ranks run the reference's loop
(:124-159) with the exchange steps and the line search inside lcx_iterate, LCX_CHECK_RANKS=1, on uneven shards; the trajectory must
equal the one-rank run (same line-search trial count).  No process launch, no host-staged transport: seconds per case."""
import os
import threading

import numpy as np
import pytest

pytestmark = pytest.mark.gpu

TIMEOUT = 120.0


class _Shared:
    def __init__(self, world):
        self.world = world
        self.barrier = threading.Barrier(world, timeout=TIMEOUT)
        self.bufs = [None] * world
        self.ready = [None] * world
        self.summed = [None] * world
        self.slots = [None] * world


class ThreadComm:
    """The `comm=` object of `Corex` for ranks that are threads of one process (what linearcorex_amd.comm.Comm is for processes)."""
    exchange = True

    def __init__(self, shared, rank, bounds):
        self.s, self.rank, self.world, self.bounds = shared, rank, shared.world, list(bounds)
        self.selftest_seconds = None
        self.calls = 0

    def shard(self, nv, rank=None):
        r = self.rank if rank is None else rank
        assert self.bounds[-1] == nv
        return self.bounds[r], self.bounds[r + 1]

    def barrier(self):
        self.s.barrier.wait()

    # host-synchronous helpers (LCX_CHECK_RANKS, end-of-fit gathers): not on the hot path
    def _exchange_host(self, value):
        self.s.slots[self.rank] = value
        self.s.barrier.wait()
        vals = list(self.s.slots)
        self.s.barrier.wait()
        return vals

    def allreduce(self, tensor):
        import torch
        vals = self._exchange_host(tensor.detach().clone())
        torch.cuda.synchronize()
        tensor.copy_(sum(vals[1:], vals[0].clone()))

    def allreduce_max(self, tensor):
        import torch
        vals = self._exchange_host(tensor.detach().clone())
        torch.cuda.synchronize()
        out = vals[0].clone()
        for v in vals[1:]:
            out = torch.maximum(out, v)
        tensor.copy_(out)

    def gather_columns(self, local, nv, like):
        parts = self._exchange_host(np.ascontiguousarray(local))
        return np.concatenate(parts, axis=-1)

    def bind_engine(self, backend, first_contact=True):
        import torch
        dev = torch.device("cuda", backend.device)
        sh, me, world = self.s, self.rank, self.world
        streams = {}

        class _View:
            def __init__(self, ptr, count, dtype):
                self.__cuda_array_interface__ = {"shape": (count,), "typestr": "<f4" if dtype == 0 else "<f8",
                                                 "data": (ptr, False), "version": 2}

        def allreduce(ptr, count, dtype, stream):
            sid = int(stream or 0)
            if sid not in streams:
                streams[sid] = torch.cuda.ExternalStream(sid, device=dev)
            s = streams[sid]
            t = torch.as_tensor(_View(ptr, count, dtype), device=dev)
            self.calls += 1
            with torch.cuda.stream(s):
                ready = torch.cuda.Event()
                ready.record(s)                           # behind the kernels that produced the buffer (stream order)
            sh.bufs[me], sh.ready[me] = t, ready
            sh.barrier.wait()                             # host: every rank's tensor and event handle are posted
            assert all(b.numel() == count for b in sh.bufs), [b.numel() for b in sh.bufs]
            with torch.cuda.stream(s):
                for r in range(world):
                    s.wait_event(sh.ready[r])
                tmp = sh.bufs[0].clone()
                for r in range(1, world):
                    tmp += sh.bufs[r]                     # rank order on every rank: identical bits
                done = torch.cuda.Event()
                done.record(s)
            sh.summed[me] = done
            sh.barrier.wait()
            with torch.cuda.stream(s):
                for r in range(world):
                    s.wait_event(sh.summed[r])            # nobody overwrites an input before everybody has read it
                t.copy_(tmp)
            sh.barrier.wait()                             # the slots may be reused by the next call

        backend.set_exchange_hook(allreduce)
        if first_contact:
            self.selftest_seconds = backend.comm_selftest(self.rank)
        return "hook"


def _planted(n, v, m, dtype, seed):
    from oracle import corex_oracle as O
    from linearcorex_amd.preprocess import preprocess as pp
    x, _ = O.gen_planted(n, v, min(m, 24), seed=seed)
    return pp(x.astype(dtype), None, "standard", None)[0]


def _run_loop(model, iters):
    for i_eps, eps in enumerate(model._init_weights()):
        model._begin_stage(i_eps, eps)
        for k in range(iters):
            model._iterate(more=k + 1 < iters)
    return np.asarray(model.history["TC"], np.float64)


def _fit(xt_local, v, m, dtype, w0, iters, comm=None, line_search="exact"):
    from linearcorex_amd import Corex
    model = Corex(n_hidden=m, seed=None, dtype=dtype, tol=0.0, device=0, comm=comm, line_search=line_search)
    be = model._attach_shard(xt_local, v)
    model.ws = w0                  # the same start on every rank and in the one-rank run (the global NumPy RNG is not per thread)
    h = _run_loop(model, iters)
    ws = model._gather(be.get_ws(0))
    out = {"history": h, "ws": ws, "trials": model.stats["trials"], "in_library": bool(getattr(model, "_iterated_in_library", False)),
           "transport": model._engine_exchange, "allreduces": be.exchange_info()["allreduces_issued"],
           "kernels": (be.kernel_name(0), be.kernel_name(1)), "merged_form": bool(be.kernel_name(2))}
    be.close()
    return out


WIDTHS = [700, 64, 1333, 7, 513, 1, 900, 300]        # eight uneven shards: one variable, less than a panel, less than a block's columns


@pytest.mark.parametrize("tag,m,gemm,pipeline,line_search", [("f64", 32, None, None, "exact"), ("f32", 128, "ct", None, "exact"),
                                                             ("f32", 64, "ct", None, "exact"), ("f32", 64, "ct", "chunks:3", "exact"),
                                                             ("f64", 24, None, "chunks:3:pass", "exact-y"),
                                                             ("f64", 24, None, "signal:3", "exact"), ("f32", 24, None, "signal:4:poll", "exact-y")])
def test_eight_thread_ranks_async_transport(tag, m, gemm, pipeline, line_search, monkeypatch):
    import torch          # noqa: F401 - before the library: liblcx_hip.so must resolve against the HIP runtime torch's wheel carries
    monkeypatch.setenv("LCX_CHECK_RANKS", "1")
    if gemm:
        monkeypatch.setenv("LCX_GEMM", gemm)
    dt = np.float32 if tag == "f32" else np.float64
    world, n, iters = 8, 4096, 12          # (a given start means ONE stage, eps = 0: linearcorex.py:113-119 anneals from a random start only)
    bounds = np.concatenate([[0], np.cumsum(WIDTHS)]).tolist()
    v = bounds[-1]
    xt = _planted(n, v, m, dt, seed=91)
    w0 = (np.random.RandomState(3).randn(m, v) * 0.004).astype(dt)          # uj well below 1
    monkeypatch.delenv("LCX_Y_PIPELINE", raising=False)
    one = _fit(xt, v, m, dt, w0, iters, line_search=line_search)
    if pipeline:
        monkeypatch.setenv("LCX_Y_PIPELINE", pipeline)
    shared = _Shared(world)
    results, errors = [None] * world, [None] * world

    def rank_main(r):
        try:
            import torch
            torch.cuda.set_device(0)
            comm = ThreadComm(shared, r, bounds)
            c0, c1 = comm.shard(v)
            results[r] = _fit(np.ascontiguousarray(xt[:, c0:c1]), v, m, dt, w0, iters, comm=comm, line_search=line_search)
            results[r]["selftest"] = comm.selftest_seconds
            results[r]["hook_calls"] = comm.calls
        except BaseException as e:          # noqa: BLE001 - reported by the test; the others fail on the broken barrier
            errors[r] = e
            shared.barrier.abort()

    threads = [threading.Thread(target=rank_main, args=(r,), daemon=True) for r in range(world)]
    for t in threads:
        t.start()
    for t in threads:
        t.join(TIMEOUT * 3)
    first = next((e for e in errors if e is not None and not isinstance(e, threading.BrokenBarrierError)), None)
    assert first is None, repr(first)
    assert all(e is None for e in errors), errors
    assert all(r is not None for r in results)
    r0 = results[0]
    assert r0["transport"] == "hook" and r0["in_library"] and r0["selftest"] > 0
    # every rank walked the same trajectory, bit for bit (decisions from all-reduced scalars), and issued the same collectives
    for r in results[1:]:
        assert np.array_equal(r["history"], r0["history"]) and np.array_equal(r["ws"], r0["ws"])
        assert r["trials"] == r0["trials"] and r["hook_calls"] == r0["hook_calls"] and r["allreduces"] == r0["allreduces"]
    assert r0["hook_calls"] > iters * 3
    # ... which is the one-rank run's to rounding, with the same number of line-search trials
    tol = 1e-10 if tag == "f64" else 5e-5
    h, h1 = r0["history"], one["history"]
    assert len(h) == len(h1) == iters and np.all(np.isfinite(h)) and h[-1] > h[0]
    assert np.max(np.abs(h - h1) / np.maximum(1.0, np.abs(h1))) < tol
    assert r0["trials"] == one["trials"]
    assert np.max(np.abs(r0["ws"] - one["ws"])) < 20 * tol * float(np.max(np.abs(one["ws"])))
    if gemm == "ct":
        assert "gemm_c" in r0["kernels"][0] or "gemm_split" in r0["kernels"][0]
    if tag == "f32" and m == 64:
        # the case that found a bug in round 5: with these widths SOME shards have a merged form (X.[grad | ws+update]^T as one
        # 128-column pass) and some do not; taking it changes the sequence of collectives, so the ranks must agree - all or none
        # (Impl::agree_on_merged).  Before that, ranks met in all-reduces of different sizes: a hang under RCCL.
        # What the ranks agreed on is what lcx_kernel_name(2) reports (none takes it: no name on any rank); that SOME shards would have
        # had one is seen on the same shards alone, where nothing needs agreeing
        assert not any(r["merged_form"] for r in results), [r["merged_form"] for r in results]
        alone = []
        for c0, c1 in ((bounds[2], bounds[3]), (bounds[3], bounds[4])):          # 1333 variables, 7 variables
            one_shard = _fit(np.ascontiguousarray(xt[:, c0:c1]), c1 - c0, m, dt, np.ascontiguousarray(w0[:, c0:c1]), 1)
            alone.append(one_shard["merged_form"])
        assert len(set(alone)) == 2, alone


def test_signalled_exchange_with_many_slots_is_bit_identical(monkeypatch):
    """LCX_Y_PIPELINE=signal on a shard whose pass writes MANY partial slots (few samples, many variables: 16 slots, summed by the wide
    reduction kernel): the chunk's reduction behind the signal is the unpipelined kernel on the chunk's range, whatever the slot count -
    two thread ranks, float64 on the wave-split v_mfma_f64_4x4x4 kernel, both wait forms, against the unpipelined run bit for bit."""
    import torch          # noqa: F401
    monkeypatch.setenv("LCX_CHECK_RANKS", "1")
    world, n, m, dt, iters = 2, 192, 24, np.float64, 6
    bounds = [0, 5000, 12000]
    v = bounds[-1]
    xt = _planted(n, v, m, dt, seed=93)
    w0 = (np.random.RandomState(4).randn(m, v) * 0.002).astype(dt)
    runs = {}
    for mode in ("off", "signal:3", "signal:2:poll"):
        if mode == "off":
            monkeypatch.delenv("LCX_Y_PIPELINE", raising=False)
        else:
            monkeypatch.setenv("LCX_Y_PIPELINE", mode)
        shared = _Shared(world)
        results, errors = [None] * world, [None] * world

        def rank_main(r):
            try:
                import torch
                torch.cuda.set_device(0)
                comm = ThreadComm(shared, r, bounds)
                c0, c1 = comm.shard(v)
                from linearcorex_amd import Corex
                model = Corex(n_hidden=m, seed=None, dtype=dt, tol=0.0, device=0, comm=comm)
                be = model._attach_shard(np.ascontiguousarray(xt[:, c0:c1]), v)
                model.ws = w0
                h = _run_loop(model, iters)
                results[r] = {"history": h, "ws": model._gather(be.get_ws(0)), "slots": be.geometry()["nt_split"], "kernel": be.kernel_name(0),
                              "allreduces": be.exchange_info()["allreduces_issued"], "trials": model.stats["trials"]}
                be.close()
            except BaseException as e:          # noqa: BLE001
                errors[r] = e
                shared.barrier.abort()

        threads = [threading.Thread(target=rank_main, args=(r,), daemon=True) for r in range(world)]
        for t in threads:
            t.start()
        for t in threads:
            t.join(TIMEOUT * 3)
        first = next((e for e in errors if e is not None and not isinstance(e, threading.BrokenBarrierError)), None)
        assert first is None, repr(first)
        assert all(r is not None for r in results)
        runs[mode] = results[0]
        assert np.array_equal(results[0]["history"], results[1]["history"]) and np.array_equal(results[0]["ws"], results[1]["ws"])
    off = runs["off"]
    assert off["slots"] >= 12 and "gemm_tn4_kernel" in off["kernel"], (off["slots"], off["kernel"])          # the wide slot reduction
    assert len(off["history"]) == iters and np.all(np.isfinite(off["history"]))
    for mode in ("signal:3", "signal:2:poll"):
        r = runs[mode]
        assert np.array_equal(r["history"], off["history"]) and np.array_equal(r["ws"], off["ws"]) and r["trials"] == off["trials"], mode
        assert r["allreduces"] > off["allreduces"], mode


def test_eight_thread_ranks_whole_fit(monkeypatch):
    """The whole `fit` path over the same transport: random start and its normalisation (:113-117), seven annealing stages with their
    rescaling (:127-134), four iterations each, the final detail moments, factor sort and gathers (:160-163) - eight uneven thread
    ranks against the one-rank run.  (`np.random.randn` is made a pure function of its shape for the test: the reference draws the
    start from NumPy's global RNG, which threads of one process would share.)"""
    import torch          # noqa: F401
    from linearcorex_amd import Corex
    monkeypatch.setenv("LCX_CHECK_RANKS", "1")
    monkeypatch.delenv("LCX_Y_PIPELINE", raising=False)
    monkeypatch.setattr(np.random, "randn", lambda *shape: np.random.RandomState(0).randn(*shape))
    world, n, m, dt = 8, 2048, 24, np.float64
    bounds = np.concatenate([[0], np.cumsum(WIDTHS)]).tolist()
    v = bounds[-1]
    xt = _planted(n, v, m, dt, seed=92)

    def fit(x_local, comm):
        model = Corex(n_hidden=m, seed=0, dtype=dt, max_iter=4, device=0, comm=comm)
        be = model._attach_shard(x_local, v)
        model._fit_resident()
        out = {"history": np.asarray(model.history["TC"], np.float64), "ws": model.ws.copy(), "clusters": model.clusters(),
               "tcs": np.asarray(model.tcs, np.float64), "rho": np.asarray(model.moments["rho"]), "trials": model.stats["trials"]}
        be.close()
        return out

    one = fit(xt, None)
    shared = _Shared(world)
    results, errors = [None] * world, [None] * world

    def rank_main(r):
        try:
            torch.cuda.set_device(0)
            comm = ThreadComm(shared, r, bounds)
            c0, c1 = comm.shard(v)
            results[r] = fit(np.ascontiguousarray(xt[:, c0:c1]), comm)
        except BaseException as e:          # noqa: BLE001
            errors[r] = e
            shared.barrier.abort()

    threads = [threading.Thread(target=rank_main, args=(r,), daemon=True) for r in range(world)]
    for t in threads:
        t.start()
    for t in threads:
        t.join(TIMEOUT * 3)
    first = next((e for e in errors if e is not None and not isinstance(e, threading.BrokenBarrierError)), None)
    assert first is None, repr(first)
    assert all(e is None for e in errors) and all(r is not None for r in results)
    r0 = results[0]
    for r in results[1:]:
        assert np.array_equal(r["history"], r0["history"]) and np.array_equal(r["ws"], r0["ws"]) and r["trials"] == r0["trials"]
    assert len(r0["history"]) == len(one["history"]) == 28 and r0["trials"] == one["trials"]
    assert np.max(np.abs(r0["history"] - one["history"]) / np.maximum(1.0, np.abs(one["history"]))) < 1e-10
    assert r0["ws"].shape == (m, v) and np.max(np.abs(r0["ws"] - one["ws"])) < 1e-9 * float(np.max(np.abs(one["ws"])))
    assert np.array_equal(r0["clusters"], one["clusters"])
    assert np.max(np.abs(r0["tcs"] - one["tcs"])) < 1e-9 * max(1.0, float(np.max(np.abs(one["tcs"]))))
    assert r0["rho"].shape == (m, v) and np.max(np.abs(r0["rho"] - one["rho"])) < 1e-9


@pytest.mark.parametrize("branch", ["ns", "syn", "ns-missing", "ns-empirical"])
def test_eight_thread_ranks_public_api_vs_oracle(branch, monkeypatch):
    """The public surface over eight uneven thread ranks against the ORACLE: `fit(x)` on raw data (gaussianize='outliers': the device
    preprocess per shard, theta gathered), `transform`, `predict`, `get_covariance(rows=...)` across shard boundaries, `clusters`, the
    moments dict - both branches (discourage_overlap True / False).  What tests/_dist_worker.py checks with two gloo processes."""
    import torch          # noqa: F401
    from linearcorex_amd import Corex
    from oracle import corex_oracle as O
    monkeypatch.setenv("LCX_CHECK_RANKS", "1")
    monkeypatch.delenv("LCX_Y_PIPELINE", raising=False)
    monkeypatch.setattr(np.random, "randn", lambda *shape: np.random.RandomState(0).randn(*shape))
    world, n, m, max_iter = 8, 500, 6, 12
    widths = [40, 3, 120, 7, 64, 1, 90, 75]
    bounds = np.concatenate([[0], np.cumsum(widths)]).tolist()
    v = bounds[-1]
    x, _ = O.gen_planted(n, v, m, seed=4)
    x[:, ::17] = np.sign(x[:, ::17]) * np.abs(x[:, ::17]) ** 1.5
    syn = branch == "syn"
    gz, missing = "outliers", None
    if branch == "ns-missing":            # missing cells (-1e6 sentinel): imputation by the batch's column means, per-column n_obs
        gz, missing = "standard", -1e6
        x[np.random.RandomState(8).rand(*x.shape) < 0.04] = missing
    elif branch == "ns-empirical":        # per-column rank -> normal quantile (the segmented sort of every rank's own columns)
        gz = "empirical"
    # (the oracle draws its start the way the reference does: np.random.seed(seed) + the global randn - unpatch for it)
    monkeypatch.undo()
    ref = (O.fit_syn if syn else O.fit_ns)(x, m, seed=0, dtype=np.float64, gaussianize=gz, max_iter=max_iter, keep_x=True,
                                           **({"missing_values": missing} if missing is not None else {}))
    monkeypatch.setenv("LCX_CHECK_RANKS", "1")
    monkeypatch.setattr(np.random, "randn", lambda *shape: np.random.RandomState(0).randn(*shape))
    shared = _Shared(world)
    results, errors = [None] * world, [None] * world

    def rank_main(r):
        try:
            torch.cuda.set_device(0)
            comm = ThreadComm(shared, r, bounds)
            model = Corex(n_hidden=m, seed=0, dtype=np.float64, max_iter=max_iter, device=0, comm=comm, gaussianize=gz,
                          missing_values=missing, discourage_overlap=not syn)
            y_fit = model.fit_transform(x)
            y = model.transform(x)
            xr = model.predict(y[:40])
            b = bounds[3]
            # ('empirical' leaves no theta: get_covariance needs theta[1] in the reference as here, :449)
            rows = model.get_covariance(rows=(b - 60, b + 90)) if gz != "empirical" else np.zeros((150, v))
            results[r] = {"history": np.asarray(model.history["TC"], np.float64), "ws": model.ws.copy(), "clusters": model.clusters(),
                          "y_fit": y_fit, "y": y, "predict": xr, "cov_rows": rows, "row0": b - 60, "tcs": np.asarray(model.tcs),
                          "rho": np.asarray(model.moments["rho"]), "theta": model.theta}
            model._backend.close()
        except BaseException as e:          # noqa: BLE001
            errors[r] = e
            shared.barrier.abort()

    threads = [threading.Thread(target=rank_main, args=(r,), daemon=True) for r in range(world)]
    for t in threads:
        t.start()
    for t in threads:
        t.join(TIMEOUT * 3)
    first = next((e for e in errors if e is not None and not isinstance(e, threading.BrokenBarrierError)), None)
    assert first is None, repr(first)
    assert all(e is None for e in errors) and all(r is not None for r in results)
    r0 = results[0]
    for r in results[1:]:
        for k in ("history", "ws", "y", "predict", "cov_rows", "clusters"):
            assert np.array_equal(r[k], r0[k]), k
    h, h_ref = r0["history"], np.asarray(ref.history_tc)
    assert len(h) == len(h_ref) and np.max(np.abs(h - h_ref) / np.maximum(1, np.abs(h_ref))) < 1e-8
    assert np.array_equal(r0["clusters"], ref.clusters())
    assert np.max(np.abs(r0["ws"] - ref.ws)) < 1e-7
    y_ref = ref.transform(ref.x_tilde)
    assert np.max(np.abs(r0["y"] - y_ref)) < 1e-7 and np.max(np.abs(r0["y_fit"] - y_ref)) < 1e-7
    if gz != "empirical":                 # (the reference cannot invert the empirical transform either: :425)
        assert np.max(np.abs(r0["predict"] - O.predict(ref.moments["X_i Z_j"], y_ref[:40], ref.theta, gz))) < 1e-6
    assert np.max(np.abs(r0["rho"] - ref.moments["rho"])) < 1e-7 and np.max(np.abs(r0["tcs"] - ref.moments["TCs"])) < 1e-7
    if gz != "empirical":
        cov_ref = ref.get_covariance()
        rows, a = r0["cov_rows"], r0["row0"]
        assert rows.shape == (150, v) and np.max(np.abs(rows - cov_ref[a:a + 150])) < 1e-6 * float(np.max(np.abs(cov_ref)))
