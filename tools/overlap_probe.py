#!/usr/bin/env python3
"""Does the pipelined Y exchange (LCX_Y_PIPELINE=signal:n) hide an exposed all-reduce behind the pass?  One GPU.

There is one GPU per box here, so the all-reduce is EMULATED: the exchange hook (include/lcx.h, lcx_set_exchange_hook) puts a kernel on
the stream the library hands over that does nothing for `latency + bytes / bandwidth` microseconds - what the collective would occupy
that stream for - and sums nothing (a group of one rank: the sum is the buffer itself).  Everything else is the product path: the
world > 1 kernels, the engine's streams, events and signal words, lcx_iterate.  The shard is one whose pass runs SEVERAL rounds of
blocks (many samples, few variables per rank: float64, 32 factors on the wave-split kernel), because a one-round pass - config 2 -
completes all its row chunks together and has nothing to overlap with.

    python tools/overlap_probe.py [n_samples n_variables n_hidden [latency_us [GBps [iters]]]]
"""
import os
import sys
import time

import numpy as np

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)


class DelayComm:
    """`comm=` of a group of ONE rank whose transport takes time: every all-reduce occupies its stream for latency + bytes / bandwidth."""
    rank, world, exchange = 0, 1, True

    def __init__(self, latency_us, gbps, ticks_per_us):
        self.latency_us, self.gbps, self.tpu = latency_us, gbps, ticks_per_us
        self.calls, self.busy_us, self.selftest_seconds = 0, 0.0, None

    def shard(self, nv, rank=None):
        return 0, nv

    def barrier(self):
        pass

    def allreduce(self, tensor):
        pass

    def allreduce_max(self, tensor):
        pass

    def gather_columns(self, local, nv, like=None):
        return local

    def bind_engine(self, backend, first_contact=True):
        import torch
        dev = torch.device("cuda", backend.device)
        streams = {}

        def allreduce(ptr, count, dtype, stream):
            sid = int(stream or 0)
            if sid not in streams:
                streams[sid] = torch.cuda.ExternalStream(sid, device=dev)
            us = self.latency_us + count * (4 if dtype == 0 else 8) / (self.gbps * 1e3) if self.gbps > 0 else 0.0
            self.calls += 1
            self.busy_us += us
            if us > 0:
                with torch.cuda.stream(streams[sid]):
                    torch.cuda._sleep(int(us * self.tpu))
        backend.set_exchange_hook(allreduce)
        return "hook"


def main():
    import torch
    import __graft_entry__ as ge
    ge.build(probe=False)
    from linearcorex_amd import Corex
    a = sys.argv[1:]
    n, v, m = (int(a[0]), int(a[1]), int(a[2])) if len(a) >= 3 else (80000, 2560, 32)
    latency_us = float(a[3]) if len(a) >= 4 else 25.0
    gbps = float(a[4]) if len(a) >= 5 else 100.0
    iters = int(a[5]) if len(a) >= 6 else 6
    torch.cuda.set_device(0)
    # the wave-split kernels (the in-launch signalling exists for them; by itself the engine gives this shape the stream-K kernels,
    # whose pass is one launch with the slot reductions behind it - nothing of it can overlap)
    os.environ.setdefault("LCX_GEMM", "tn")
    # ticks of torch.cuda._sleep per microsecond, measured
    e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    torch.cuda._sleep(1000000)
    torch.cuda.synchronize()
    e0.record()
    torch.cuda._sleep(20000000)
    e1.record()
    torch.cuda.synchronize()
    tpu = 20000000 / (e0.elapsed_time(e1) * 1e3)
    x = np.random.RandomState(1).randn(n, v)
    x = (x - x.mean(0)) / x.std(0)
    np.random.randn = lambda *shape: np.random.RandomState(0).randn(*shape)          # the same start for every run
    print("# shard %d x %d x %d float64, one rank, emulated all-reduce = %.0f us + bytes / %.0f GB/s (N x m buffer: %.1f MB -> %.0f us), %d iterations per stage"
          % (n, v, m, latency_us, gbps, n * m * 8 / 1e6, latency_us + n * m * 8 / (gbps * 1e3), iters), flush=True)

    def run(mode, lat, bw):
        a_, c_ = run_once(mode, lat, bw)
        b_, _ = run_once(mode, lat, bw)
        return (a_ if a_["ms"] <= b_["ms"] else b_), c_          # the better of two (a stall of the shared host shows up as an outlier)

    def run_once(mode, lat, bw):
        if mode == "off":
            os.environ.pop("LCX_Y_PIPELINE", None)
        else:
            os.environ["LCX_Y_PIPELINE"] = mode
        comm = DelayComm(lat, bw, tpu)
        model = Corex(n_hidden=m, seed=0, dtype=np.float64, tol=0.0, max_iter=10 ** 9, device=0, comm=comm)
        be = model._attach_shard(x, v)
        t_iter, n_it = 0.0, 0
        be.timing_reset()
        be.timing_sample(1)
        for walk in range(2):                      # the first walk warms up
            model.ws, model.history = np.zeros((0, 0)), {}
            t_iter, n_it = 0.0, 0
            comm.calls, comm.busy_us = 0, 0.0
            for i_eps, eps in enumerate(model._init_weights()):
                model._begin_stage(i_eps, eps)
                be.synchronize()
                torch.cuda.synchronize()
                be.timing_enable(walk == 1)
                t0 = time.perf_counter()
                for k in range(iters):
                    model._iterate(more=k + 1 < iters)
                be.synchronize()
                torch.cuda.synchronize()
                t_iter += time.perf_counter() - t0
                n_it += iters
                be.timing_enable(False)
        tr = be.timing_read()
        out = {"ms": t_iter / n_it * 1e3, "tc": float(model.tc), "trials": model.stats["trials"], "geo": be.geometry(),
               "kernel": be.kernel_name(0), "xbt_us": 1e3 * tr["gemm_nt"][1] / max(1, tr["gemm_nt"][0]),
               "xty_us": 1e3 * tr["gemm_tn"][1] / max(1, tr["gemm_tn"][0])}
        be.close()
        return out, comm

    base, _ = run("off", 0.0, 0.0)
    print("# pass kernel %s, %d row tiles x %d slots on %d CUs" % (base["kernel"], base["geo"]["n_pad"] // 64, base["geo"]["nt_split"], base["geo"]["n_cus"]))
    print("%-16s %-22s %10s %14s %10s %10s %s" % ("LCX_Y_PIPELINE", "emulated transport", "ms/iter", "exposed ms/iter", "X.B^T us", "X^T.Y us", "final TC"))
    print("%-16s %-22s %10.4f %14s %10.1f %10.1f %.12f" % ("off", "none (free sums)", base["ms"], "-", base["xbt_us"], base["xty_us"], base["tc"]))
    rows = {}
    modes = tuple(os.environ["LCX_PROBE_MODES"].split(",")) if os.environ.get("LCX_PROBE_MODES") else ("off", "signal:2", "signal:4", "signal:8", "chunks:4")
    for mode in modes:
        for lat, bw, label in ((0.0, 0.0, "free"), (latency_us, gbps, "%.0f us + B/%.0f GB/s" % (latency_us, gbps))):
            if mode == "off" and label == "free":
                continue
            r, comm = run(mode, lat, bw)
            assert r["tc"] == base["tc"] and r["trials"] == base["trials"], (mode, r["tc"], base["tc"])
            rows[(mode, label)] = r["ms"]
            print("%-16s %-22s %10.4f %14.4f %10.1f %10.1f %.12f" % (mode, label, r["ms"], r["ms"] - base["ms"], r["xbt_us"], r["xty_us"], r["tc"]), flush=True)
    slow = "%.0f us + B/%.0f GB/s" % (latency_us, gbps)
    if ("off", slow) not in rows or any((mo, slow) not in rows for mo in ("signal:2", "signal:4", "signal:8", "chunks:4")):
        return
    serial = rows[("off", slow)] - base["ms"]
    print("# exposed exchange per iteration: unpipelined %.4f ms; " % serial
          + "; ".join("%s %.4f (%.0f %% hidden)" % (mo, rows[(mo, slow)] - base["ms"], 100 * (1 - (rows[(mo, slow)] - base["ms"]) / serial))
                      for mo in ("signal:2", "signal:4", "signal:8", "chunks:4")))


if __name__ == "__main__":
    main()
