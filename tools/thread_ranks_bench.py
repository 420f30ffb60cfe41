#!/usr/bin/env python3
"""What does the library's multi-rank path cost when the transport is not the bottleneck?

`bench.py --gpus 8` rehearsed on one GPU goes through gloo, which stages every all-reduce through the host (82 ms for 25.7 MB at 8
ranks): its 0.20 "efficiency" measures gloo.  Here the 8 ranks are threads of one process, each with its own engine handle on GPU 0,
exchanging through the asynchronous on-device transport of tests/test_thread_ranks_gpu.py (stream-ordered sums behind events).  The
GPU is time-sliced by 8 shards, so the ideal time of an iteration of all ranks is 8 x the one-rank time of one shard; what is above
that is the engine's exchange path (extra kernels, events, the hook) plus the sums themselves (8 x N x m elements read per
all-reduce, on this one GPU's HBM) and the Python threads' GIL.

    python tools/thread_ranks_bench.py [n_samples n_variables_per_rank n_hidden f32|f64 [world [iters_per_stage]]]
"""
import json
import os
import sys
import threading
import time

import numpy as np

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)


def main():
    import torch
    from linearcorex_amd import Corex
    from tests.test_thread_ranks_gpu import ThreadComm, _Shared
    a = sys.argv[1:]
    n, v_per, m = (int(a[0]), int(a[1]), int(a[2])) if len(a) >= 3 else (50000, 16000, 128)
    dt = np.float64 if (len(a) >= 4 and a[3] == "f64") else np.float32
    world = int(a[4]) if len(a) >= 5 else 8
    iters = int(a[5]) if len(a) >= 6 else 5
    np.random.randn = lambda *shape: np.random.RandomState(0).randn(*shape)       # the start is drawn from the global RNG: one per process

    def run(comm, v_total, c0, out, sites_too=True):
        model = Corex(n_hidden=m, seed=0, dtype=dt, tol=0.0, max_iter=10 ** 9, device=0, comm=comm)
        model.n_samples, model.nv = n, v_total
        model._cols = (c0, c0 + v_per)
        be = model._make_backend(n, v_per)
        be.generate_x(1, 0, 1, c0)
        model.theta = (np.zeros(1), np.ones(1))
        model._x_resident = True

        def sync():
            be.synchronize()
            torch.cuda.synchronize()
            if comm is not None:
                comm.barrier()

        t_iter = 0.0
        be.timing_reset()
        be.timing_sample(4)
        for walk in range(2):                      # the first walk warms up
            model.ws = np.zeros((0, 0))
            model.history = {}
            t_iter = 0.0
            for i_eps, eps in enumerate(model._init_weights()):
                model._begin_stage(i_eps, eps)
                sync()
                be.timing_enable(walk == 1 and sites_too)
                t0 = time.perf_counter()
                for k in range(iters):
                    model._iterate(more=k + 1 < iters)
                sync()
                t_iter += time.perf_counter() - t0
                be.timing_enable(False)
        # the exchange steps by site (include/lcx.h, timing kinds 3-6: HIP events on the stream that carries each all-reduce) and the X passes
        sites = {k: (issued / (7 * iters)) * (ms / cnt) for k, (issued, cnt, ms) in be.timing_exchange_read().items() if cnt}
        passes = {k: (be.timing_passes_by_kind()[k] / (7 * iters)) * (ms / cnt) for k, (cnt, ms) in be.timing_read().items() if cnt}
        out.update(ms_per_iteration=t_iter / (7 * iters) * 1e3, final_tc=float(model.tc), trials=model.stats["trials"],
                   allreduces=be.exchange_info()["allreduces_issued"], exchange_ms=sites, pass_ms=passes)
        be.close()

    one = {}
    run(None, v_per, 0, one)
    shared = _Shared(world)
    outs, errs = [dict() for _ in range(world)], [None] * world
    bounds = [v_per * r for r in range(world + 1)]

    def rank_main(r, sites_too):
        try:
            torch.cuda.set_device(0)
            run(ThreadComm(shared, r, bounds), v_per * world, bounds[r], outs[r], sites_too)
        except BaseException as e:              # noqa: BLE001
            errs[r] = e
            shared.barrier.abort()

    # twice: untimed (the headline: event pairs cost host time under the one GIL the 8 "ranks" share), then with the timing sites on
    tn = None
    for sites_too in (False, True):
        threads = [threading.Thread(target=rank_main, args=(r, sites_too), daemon=True) for r in range(world)]
        for t in threads:
            t.start()
        for t in threads:
            t.join(3600)
        bad = [e for e in errs if e is not None]
        if bad:
            raise bad[0]
        if tn is None:
            tn = max(o["ms_per_iteration"] for o in outs)
    t1 = one["ms_per_iteration"]
    rec = {"shard": "%d x %d x %d %s" % (n, v_per, m, np.dtype(dt).name), "ranks_on_one_gpu": world, "iterations_per_stage": iters,
           "one_rank_ms_per_iteration": t1, "all_ranks_ms_per_iteration": tn, "ideal_time_sliced_ms": world * t1,
           "fraction_of_ideal": world * t1 / tn, "allreduces_issued_per_rank": outs[0]["allreduces"],
           # per iteration of ONE rank, by site; the all-reduce durations include the wait for the other (time-sliced) ranks' partial sums
           "exchange_ms_per_iteration_rank0": {k: round(v, 4) for k, v in outs[0]["exchange_ms"].items()},
           "exchange_ms_per_iteration_rank_min_max": [round(min(sum(o["exchange_ms"].values()) for o in outs), 4),
                                                      round(max(sum(o["exchange_ms"].values()) for o in outs), 4)],
           "x_pass_ms_per_iteration_rank0": {k: round(v, 4) for k, v in outs[0]["pass_ms"].items()},
           "x_pass_ms_per_iteration_one_rank_alone": {k: round(v, 4) for k, v in one["pass_ms"].items()},
           "all_ranks_ms_per_iteration_with_the_timing_sites_on": max(o["ms_per_iteration"] for o in outs),
           "same_decisions_on_every_rank": len({o["trials"] for o in outs}) == 1 and len({o["final_tc"] for o in outs}) == 1,
           "transport": "asynchronous on-device sums behind events (tests/test_thread_ranks_gpu.py), ranks = threads of one process"}
    print(json.dumps(rec))


if __name__ == "__main__":
    main()
