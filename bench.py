#!/usr/bin/env python3
"""bench.py - fit iterations/sec of the Linear CorEx hot path on MI355X (BASELINE.json metric).

A "step" is one fixed-point iteration of the fit loop (reference `_update_ns` + its book-keeping,
linearcorex.py:137-151).  The run walks the reference's 7-stage annealing schedule (:119-134); in
every stage a window of EXACTLY --steps iterations is timed (barrier + synchronize on both sides, MAX
over ranks), stage changes (:127-134: rescale + a non-quick moment evaluation) are timed separately.
When one walk of the schedule is shorter than ~0.6 s of GPU work it is repeated on the same resident
data (same start, same trajectory) and the per-stage MEDIAN window is used, so the figure does not move
with --steps / --warmup:

    ms_per_step = sum_over_stages( median_over_repeats( window ) ) / (7 * steps)

Workloads (BASELINE.json `configs`):
  N=1  headline  c3       50k x 100k, n_hidden 64, float32, X generated on the device  (MFMA roofline run; >= 3 walks)
       nested    config.c2: 10k x 5k, n_hidden 32, float64                            (HBM-bound; own protocol)
                 config.c4shard: 50k x 125k, n_hidden 128, float32                    (the one-GPU point of the --gpus N series)
                 config.c4_unsharded_one_gpu: 50k x 1M, n_hidden 128, float32         (configs[3] whole, one resident copy of X)
                 config.linear_trial_mode / later_trials_by_linearity                 (opt-in line searches, reported beside)
                 config.fit_to_convergence[_planted]                                  (wall clock of whole fits)
  N>1  headline  c4shard  50k x 125k per GPU, n_hidden 128, float32, n_variables sharded (weak scaling), all-reduces issued by
                          the library (RCCL communicator per handle), CPU baseline on rank 0
       nested    config.single_gpu_same_shard, config.c2_weak: 10k x 5k per GPU, float64

    python bench.py                                   # N=1, c3 headline + c2 block + CPU baselines
    python bench.py --gpus 8 --steps 20 --warmup 5    # spawns 8 ranks itself (torch.distributed.run as a child)
    python -m torch.distributed.run --nproc-per-node N ... bench.py --gpus N ...   # or launched as ranks

Prints ONE compact JSON line (a few KB: the contract keys, `roofline`, `cpu_baseline`, scalar riders of the nested
measurements) on rank 0's stdout.  Every nested block (c2, c4shard, the opt-in line searches, convergence runs, use-site
split, launch geometry ...) goes to the side file named in `config.detail` (default gpurun_out/bench_detail.json) and as one
`BENCH_DETAIL {...}` line to stderr - the driver keeps only a few KB of stdout, and a 21 KB line could not be parsed.
"""
import argparse
import json
import math
import os
import subprocess
import sys
import time

ROOT = os.path.dirname(os.path.abspath(__file__))
sys.path.insert(0, ROOT)

# the workload table + hardware peaks, the rank launcher, the profile quotes and the stdout record live in benchkit/ (plain modules, no
# GPU, no oracle); this file keeps the measurement and the CPU-baseline leg.  Re-exported: tests and tools address them as bench.<name>
from benchkit.workloads import (BF16_MFMA_PEAK_TFLOPS, DESCRIPTION, FP32_MFMA_PEAK_TFLOPS, FP64_MFMA_PEAK_TFLOPS, HBM_PEAK_GBS,  # noqa: E402,F401
                                MEASURED_CEILINGS, MIN_TIMED_SECONDS, SPLIT_PRODUCTS, WORKLOADS, _adhoc)
from benchkit.launch import spawn_ranks, supervise_rank                                              # noqa: E402,F401
from benchkit.profiles import _lib_src_hash, load_pmc_traffic, load_rocprof_avg                      # noqa: E402,F401
from benchkit.record import ROOFLINE_LINE_KEYS, _pick, _r, compact_line, emit, series_of             # noqa: E402,F401

def parse(argv=None):
    ap = argparse.ArgumentParser()
    ap.add_argument("--gpus", type=int, default=1)
    ap.add_argument("--steps", type=int, default=20, help="iterations per timed window (one window per annealing stage)")
    ap.add_argument("--warmup", type=int, default=5, help="untimed iterations before the first window")
    ap.add_argument("--workload", default="auto",
                    help="auto (c3 at 1 GPU, c4shard per GPU above, each with its nested c2 block), one of %s, or an ad-hoc "
                         "shard shape NxVxM:f32|f64 (not a BASELINE line)" % ", ".join(sorted(WORKLOADS)))
    ap.add_argument("--no-extras", dest="extras", action="store_false",
                    help="headline measurement only: no nested c2 block, no linear-mode run, no fit to convergence, no "
                         "get_covariance timing, no CPU baseline (what the profiling scripts run)")
    ap.add_argument("--cpu-seconds", type=float, default=20.0,
                    help="time budget of the CPU-baseline sample of the headline workload (0 = skip)")
    ap.add_argument("--cpu-iters-per-stage", type=int, default=10,
                    help="CPU baseline of the c2 block: oracle iterations per annealing stage (0 = skip)")
    ap.add_argument("--convergence-max-iter", type=int, default=60,
                    help="iterations per annealing stage allowed to the whole-fit wall-clock measurement of a generated headline "
                         "workload (0 = skip)")
    ap.add_argument("--convergence-planted-max-iter", type=int, default=2000,
                    help="iterations per annealing stage allowed to the whole-fit wall-clock measurement on planted data")
    ap.add_argument("--c4full-steps", type=int, default=0,
                    help="iterations per window of a config-4-unsharded-on-one-GPU block (209 GB resident) added to the N=1 "
                         "detail record (0 = skip, the default: `--workload c4full` measures it on its own)")
    ap.add_argument("--detail-out", default=None,
                    help="where the full record goes (default gpurun_out/bench_detail.json, bench_detail_gpusN.json for N > 1)")
    ap.add_argument("--max-line-bytes", type=int, default=4096, help="budget of the stdout line")
    ap.add_argument("--no-kernel-timing", action="store_true")
    ap.add_argument("--timing-sample", type=int, default=0,
                    help="HIP-event pairs around every n-th X pass of the timed windows (a pair costs ~5 us of stream time; "
                         "0 = 1 for passes above 1 ms, 16 below)")
    ap.add_argument("--force-exchange", action="store_true",
                    help="1 GPU only: run the multi-rank device path (world>1 kernels + RCCL all-reduces in a group of "
                         "one rank) to measure the fixed cost of the exchange steps")
    ap.add_argument("--line-search", default="exact", choices=["linear", "exact", "exact-y"],
                    help="exact (default, the headline): reference-shaped, every trial makes 2 passes over X "
                         "(linearcorex.py:321); linear: trials cost no pass over X")
    ap.add_argument("--repeats", type=int, default=0, help="walks of the schedule (0 = as many as MIN_TIMED_SECONDS needs)")
    ap.add_argument("--f32-gemm", default="mfma", choices=["mfma", "split"],
                    help="arithmetic of the X passes of the float32 workloads behind `value` (include/lcx.h lcx_set_f32_gemm): float32 MFMA "
                         "(default) or the exact three-way bf16 split on the bf16 matrix pipe; the other one is measured beside it")
    args = ap.parse_args(argv)
    _adhoc(args.workload)
    return args


# ------------------------------------------------------------------------------------------------------
# CPU baselines: the NumPy oracle (a port of the reference path, pinned to it bit for bit) on the host cores
# ------------------------------------------------------------------------------------------------------
def _usable_cores():
    """Cores this process may actually use: the affinity mask, capped by the cgroup CPU quota (the GPU box shows 256
    logical CPUs under a 16-CPU quota: a BLAS pool sized by the former spends its time being throttled)."""
    n = len(os.sched_getaffinity(0)) if hasattr(os, "sched_getaffinity") else (os.cpu_count() or 1)
    try:
        with open("/sys/fs/cgroup/cpu.max") as f:
            quota, period = f.read().split()[:2]
        if quota != "max":
            n = min(n, max(1, int(float(quota) / float(period))))
    except Exception:
        pass
    return max(1, n)


class _BlasPool:
    """Context: BLAS threads = usable cores; reports (threads, vendor/version)."""

    def __enter__(self):
        self.threads, self.vendor, self._ctx = _usable_cores(), "unknown", None
        try:
            from threadpoolctl import threadpool_info, threadpool_limits
            self._ctx = threadpool_limits(limits=self.threads, user_api="blas")
            self._ctx.__enter__()
            infos = [p for p in threadpool_info() if p.get("user_api") == "blas"]
            self.threads = max([p.get("num_threads", 1) for p in infos] or [self.threads])
            self.vendor = ", ".join(sorted({"%s %s" % (p.get("internal_api", "?"), p.get("version", "")) for p in infos}))
        except Exception:
            pass
        return self

    def __exit__(self, *a):
        if self._ctx is not None:
            self._ctx.__exit__(*a)


def cpu_baseline_resident(x, m, dtype, iters_per_stage):
    """Small workloads (c2): same X as the GPU run, the whole 7-stage schedule bounded to iters_per_stage
    iterations per stage."""
    import numpy as np
    from oracle import corex_oracle as O
    xt = O.preprocess(np.asarray(x, dtype=dtype))[0]
    with _BlasPool() as pool:
        threads, vendor = pool.threads, pool.vendor
        t0 = time.perf_counter()
        res = O.fit_ns_preprocessed(xt, m, seed=0, dtype=dtype, max_iter=iters_per_stage, tol=0.0, finish=False)
        t1 = time.perf_counter()
    n_it = len(res.history_tc)
    return {"value": n_it / (t1 - t0), "unit": "iterations/s", "cores": threads, "kind": "port", "host": _host_facts(),
            "sample": "%d iterations (%d per annealing stage x 7, stage changes included) of the same workload on the same X, "
                      "NumPy %s / %s, BLAS threads=%d, %.1f s" % (n_it, iters_per_stage, np.__version__, vendor, threads, t1 - t0),
            "trials_per_iteration": res.n_trials / max(1, n_it)}


def _host_facts():
    """what else competes for the cores the CPU figures are taken on"""
    try:
        la = [round(v, 2) for v in os.getloadavg()]
    except OSError:
        la = None
    return {"loadavg_1_5_15": la, "usable_cores": _usable_cores(), "logical_cpus": os.cpu_count()}


def cpu_fit_to_convergence(x, m, dtype):
    """BASELINE.md section 3's CPU leg: wall-clock of the oracle's whole fit (reference defaults: tol 1e-5 per annealing stage,
    linearcorex.py:136-155; preprocessing and the final detail moments included) on the same host matrix the device fit ran on."""
    import numpy as np
    from oracle import corex_oracle as O
    before = _host_facts()
    with _BlasPool() as pool:
        t0 = time.perf_counter()
        res = O.fit_ns(np.asarray(x), m, seed=0, dtype=dtype)
        t1 = time.perf_counter()
        threads, vendor = pool.threads, pool.vendor
    n_it = len(res.history_tc)
    return {"seconds": t1 - t0, "iterations": n_it, "trials": int(res.n_trials), "trials_per_iteration": res.n_trials / max(1, n_it),
            "TC": float(res.history_tc[-1]), "tol": 1e-5, "cores": threads, "kind": "port",
            "iterations_per_sec_incl_setup": n_it / (t1 - t0), "host": before,
            "sample": "the whole fit of the oracle (NumPy %s / %s, BLAS threads=%d) on the same X" % (np.__version__, vendor, threads)}


def c1_block(device):
    """BASELINE.json configs[0]: tests/data/test_big5.csv (2000 x 50), n_hidden=5 - the latency-bound end of the path.  The matrix is
    the one the reference's CLI parses from that file (vis_corex.py:496-512: header row and label column dropped, float), carried
    as `x_raw` in tests/golden/g1_big5.npz (the reference's tree does not exist on the GPU box).  Whole fits with the reference's
    defaults, device (second fit of a process: library loaded, allocator warm) and oracle, in both precisions."""
    import numpy as np
    from linearcorex_amd import Corex
    from oracle import corex_oracle as O
    x = np.load(os.path.join(ROOT, "tests", "golden", "g1_big5.npz"))["x_raw"].astype(np.float64)
    blk = {"workload": "c1: tests/data/test_big5.csv %d x %d, n_hidden=5 (BASELINE.json configs[0]); whole fit, tol 1e-5" % x.shape,
           "host": _host_facts()}
    for tag, dt in (("f32", np.float32), ("f64", np.float64)):
        Corex(n_hidden=5, seed=0, dtype=dt, device=device).fit(x)._backend.close()
        t0 = time.perf_counter()
        mdl = Corex(n_hidden=5, seed=0, dtype=dt, device=device).fit(x)
        t1 = time.perf_counter()
        n_it = len(mdl.history["TC"])
        with _BlasPool() as pool:
            t2 = time.perf_counter()
            ref = O.fit_ns(x, 5, seed=0, dtype=dt)
            t3 = time.perf_counter()
            threads = pool.threads
        n_ref = len(ref.history_tc)
        blk[tag] = {"fit_seconds": t1 - t0, "iterations": n_it, "trials": int(mdl.stats["trials"]), "TC": float(mdl.tc),
                    "us_per_iteration": (t1 - t0) / n_it * 1e6,
                    "cpu_fit_seconds": t3 - t2, "cpu_iterations": n_ref, "cpu_trials": int(ref.n_trials), "cpu_TC": float(ref.history_tc[-1]),
                    "cpu_us_per_iteration": (t3 - t2) / n_ref * 1e6, "cpu_cores": threads,
                    "device_over_cpu_wall_clock": (t1 - t0) / (t3 - t2),
                    "same_clusters": bool(np.array_equal(mdl.clusters(), ref.clusters()))}
        mdl._backend.close()
    return blk


def _host_gaussian(n, v, dtype, seed):
    """iid N(0,1) of shape (n, v) drawn by all cores (one Generator per row block), then standardised per column
    like preprocess 'standard' (reference :409-415)."""
    import numpy as np
    from concurrent.futures import ThreadPoolExecutor
    x = np.empty((n, v), dtype=dtype)
    workers = max(1, min(32, _usable_cores()))
    bounds = [(n * k // workers, n * (k + 1) // workers) for k in range(workers)]

    def fill(k):
        r0, r1 = bounds[k]
        g = np.random.Generator(np.random.PCG64([seed, k]))
        step = max(1, (1 << 24) // v)
        for r in range(r0, r1, step):
            e = min(r1, r + step)
            x[r:e] = g.standard_normal((e - r, v), dtype=dtype)
    with ThreadPoolExecutor(workers) as ex:
        list(ex.map(fill, range(workers)))
    cb = [(v * k // workers, v * (k + 1) // workers) for k in range(workers)]

    def standardise(k):
        c0, c1 = cb[k]
        for c in range(c0, c1, 2048):
            e = min(c1, c + 2048)
            blk = x[:, c:e]
            mu = blk.mean(axis=0, dtype=np.float64)
            blk -= mu.astype(dtype)
            sd = np.sqrt(np.einsum("ij,ij->j", blk, blk, dtype=np.float64) / n)
            blk /= np.maximum(sd, 1e-10).astype(dtype)
    with ThreadPoolExecutor(workers) as ex:
        list(ex.map(standardise, range(workers)))
    return x


def _host_memory_available():
    avail = None
    try:
        with open("/proc/meminfo") as f:
            for ln in f:
                if ln.startswith("MemAvailable:"):
                    avail = int(ln.split()[1]) * 1024
    except Exception:
        pass
    try:
        with open("/sys/fs/cgroup/memory.max") as f:
            lim = f.read().strip()
            if lim.isdigit():
                used = 0
                try:
                    with open("/sys/fs/cgroup/memory.current") as g:
                        used = int(g.read().strip())
                except Exception:
                    pass
                avail = min(avail, int(lim) - used) if avail is not None else int(lim) - used
    except Exception:
        pass
    return avail


def cpu_baseline_generated(n, v, m, dtype, budget_s, label, also=()):
    """Large workloads (c3, c4 shard): X cannot be taken from the GPU run (it is generated on the device), so an iid
    Gaussian X of the same shape is drawn on the host (fewer variables if the host's memory cannot hold it, and never
    more than 10^5 - BASELINE.md section 3: the iteration cost is linear in n_variables and the scale is stated), and the
    oracle's loop (reference :136-155) is timed for at least 3 iterations, one per annealing stage while the budget lasts.
    also: further (label, n_hidden, n_variables of that workload) timed on the same host matrix; the return value is then
    a list, the first entry being the one described by the positional arguments."""
    import numpy as np
    from oracle import corex_oracle as O
    es = np.dtype(dtype).itemsize
    avail = _host_memory_available()
    v_cpu = min(v, 100000)
    if avail is not None:
        fit = int(0.6 * avail / (n * es))
        if fit < v_cpu:
            v_cpu = max(1000, fit // 1000 * 1000)
    host_before = _host_facts()
    t_gen = time.perf_counter()
    x = _host_gaussian(n, v_cpu, dtype, seed=1)
    t_gen = time.perf_counter() - t_gen
    results = []
    for lab, mm, v_work in [(label, m, v)] + list(also):
        with _BlasPool() as pool:
            threads, vendor = pool.threads, pool.vendor
            w = O.initial_weights(0, mm, v_cpu, dtype)
            w /= (10.0 * O.norm(x, w, 0))[:, np.newaxis]
            mo = O.moments_ns(x, w, 0, quick=True)
            eps, n_it, trials, invalid, t_iter, t_stage, stages = 0, 0, 0, 0, 0.0, 0.0, 0
            t_begin = time.perf_counter()
            for stage, eps_new in enumerate(O.anneal_schedule(True)):
                t0 = time.perf_counter()
                eps_old, eps = eps, eps_new
                if stage > 0:
                    w = O.rescale_for_stage(w, mo["uj"], eps_old, eps)
                mo = O.moments_ns(x, w, eps, quick=False)
                t1 = time.perf_counter()
                w, mo, info = O.update_ns(x, w, mo, eps, 0.0)
                t2 = time.perf_counter()
                t_stage += t1 - t0
                t_iter += t2 - t1
                n_it += 1
                stages += 1
                trials += info["n_trials"]
                invalid += info["n_invalid"]
                if n_it >= 3 and (time.perf_counter() - t_begin) > budget_s:
                    break
        its = n_it / t_iter
        scale = float(v_cpu) / float(v_work)
        results.append({
            "value": its * scale, "unit": "iterations/s", "cores": threads, "kind": "port", "host": host_before,
            "sample": "%s: %d iterations of the oracle loop (reference linearcorex.py:136-155), the first one of each of the "
                      "first %d annealing stages, on host-generated iid Gaussian X %d x %d %s, n_hidden %d; %.1f s in the "
                      "iterations (%.2f s each), %.1f s in the %d stage changes, %.1f s to draw X; NumPy %s / %s, BLAS threads=%d"
                      % (lab, n_it, stages, n, v_cpu, np.dtype(dtype).name, mm, t_iter, t_iter / n_it, t_stage, stages, t_gen,
                         np.__version__, vendor, threads),
            "n_variables_timed": v_cpu, "n_variables_workload": v_work,
            "scaled_linearly_in_n_variables_by": scale, "measured_iterations_per_sec_at_timed_size": its,
            "trials_per_iteration": trials / max(1, n_it),
            "x_passes_per_iteration": (2 * n_it + 2 * trials - invalid) / max(1, n_it),
            "not_the_same_work_as_value": "the oracle makes the reference's 2 + 2T - E passes over X per iteration (T trials, E "
                                          "early exits; first iterations of the stages), the device path elides the second _sig "
                                          "pass and merges passes (config.x_passes_per_iteration): a reported baseline, not a ratio"})
    del x
    return results if also else results[0]


# ------------------------------------------------------------------------------------------------------
# committed profile figures (PMC traffic, rocprofv3 kernel time): quoted only while they describe THIS library
# ------------------------------------------------------------------------------------------------------


# ------------------------------------------------------------------------------------------------------
# the measurement
# ------------------------------------------------------------------------------------------------------
def make_model(workload, comm, world, rank, local_rank, line_search, keep_x=False, f32_gemm=None):
    import numpy as np
    from linearcorex_amd import Corex
    n, v_per, m, tag = WORKLOADS[workload]
    dtype = np.float64 if tag == "f64" else np.float32
    v_total = v_per * world
    model = Corex(n_hidden=m, seed=0, dtype=dtype, tol=0.0, max_iter=10 ** 9, device=local_rank, comm=comm,
                  line_search=line_search, f32_gemm=f32_gemm if tag == "f32" else None)
    x_host = None
    if n * v_per * 8 <= int(os.environ.get("LCX_BENCH_GENERATE_ABOVE", 4 << 30)):      # (the variable: a test hook, see benchkit/workloads._shrink)
        # Gen-A: iid N(0,1); rank r draws its own columns from RandomState(1 + r)
        from linearcorex_amd.preprocess import preprocess as pp
        x_host = np.random.RandomState(1 + rank).randn(n, v_per)
        xt = pp(x_host.astype(dtype), None, "standard", None)[0]
        be = model._attach_shard(xt, v_total)
        del xt
    else:
        model.n_samples, model.nv = n, v_total
        model._cols = model._comm.shard(v_total)
        be = model._make_backend(n, v_per)
        be.generate_x(1, 0, 1, model._cols[0])
        model.theta = (np.zeros(1), np.ones(1))
    return model, be, (x_host if keep_x else None)


def measure(args, comm, world, rank, local_rank, workload, steps, warmup, line_search, keep_x=False, repeats=0,
            kernel_timing=True, min_repeats=1, f32_gemm=None):
    """Walk the 7-stage schedule; per stage: the stage change (timed on its own), then a window of exactly `steps`
    iterations (barrier + synchronize on both sides).  Repeat the walk from the same start until MIN_TIMED_SECONDS
    of windows have been timed."""
    import numpy as np
    import torch
    model, be, x_host = make_model(workload, comm, world, rank, local_rank, line_search, keep_x,
                                   f32_gemm if f32_gemm is not None else args.f32_gemm)

    def sync_local():
        be.synchronize()
        torch.cuda.synchronize()

    def sync():
        sync_local()
        if comm is not None:
            comm.barrier()
            torch.cuda.synchronize()

    def rank_max(values):
        if comm is None:
            return list(values)
        import torch.distributed as dist
        t = torch.tensor(list(values), dtype=torch.float64, device="cuda")
        dist.all_reduce(t, op=dist.ReduceOp.MAX)
        return [float(u) for u in t.cpu()]

    # clocks: about 250 ms of X passes before anything is timed.  The count comes from the workload's size alone - the same on
    # every rank: with live exchange steps every pass carries a collective
    n_, v_, m_, tag_ = WORKLOADS[workload]
    est_ms = max(2.0 * n_ * v_ * m_ / (60e12 if tag_ == "f32" else 30e12), n_ * v_ * (4 if tag_ == "f32" else 8) / 3e12) * 1e3
    n_warm = int(min(4000, max(4, 250.0 / max(est_ms, 1e-3))))
    if comm is not None:
        n_warm = min(n_warm, 300)              # every pass carries a collective (milliseconds each over gloo in the tests)
    be.synchronize()
    t0 = time.perf_counter()
    for _ in range(n_warm):
        be.moments_a(1)
    be.synchronize()
    pass_ms = (time.perf_counter() - t0) * 1e3 / n_warm
    every = args.timing_sample if args.timing_sample > 0 else (1 if pass_ms >= 1.0 else 16)
    timing = kernel_timing and not args.no_kernel_timing
    if timing:
        be.timing_reset()
        be.timing_sample(every)

    stat_keys = ("trials", "invalid_trials", "moment_evals")
    totals = dict.fromkeys(stat_keys, 0)

    def walk(record):
        np.random.seed(0)                      # the same start on every walk: identical trajectories
        model.ws = np.zeros((0, 0))
        model.history = {}
        sched = model._init_weights()
        win, chg, loc = [], [], []
        for i_eps, eps in enumerate(sched):
            sync()
            t0 = time.perf_counter()
            model._begin_stage(i_eps, eps)
            sync()
            chg.append(time.perf_counter() - t0)
            if i_eps == 0:
                for k in range(warmup):
                    model._iterate(more=k + 1 < warmup)
                sync()
            before = {k: model.stats.get(k, 0) for k in stat_keys}
            if timing and record:
                be.timing_enable(True)
            t0 = time.perf_counter()
            for k in range(steps):
                # more: the engine may start iteration k+1 before returning (as in a real fit); never past the window's
                # end, so that exactly `steps` iterations of work lie between t0 and t1
                model._iterate(more=k + 1 < steps)
            sync_local()
            t_loc = time.perf_counter()          # this rank's own stream has drained; the barrier below then waits for the slowest one
            sync()
            t1 = time.perf_counter()
            if timing and record:
                be.timing_enable(False)
            if record:
                for k in stat_keys:
                    totals[k] += model.stats.get(k, 0) - before[k]
            win.append(t1 - t0)
            loc.append(t_loc - t0)
        local_windows.append(loc)
        return rank_max(win), rank_max(chg)

    local_windows = []
    first_win, first_chg = walk(record=False)
    local_windows.clear()
    est = sum(first_win)
    if repeats <= 0:
        repeats = 1 if est >= MIN_TIMED_SECONDS else min(60, int(math.ceil(MIN_TIMED_SECONDS / max(est, 1e-6))))
        repeats = max(repeats, min_repeats)
    wins, chgs = [], []
    for _ in range(repeats):
        w, c = walk(record=True)
        wins.append(w)
        chgs.append(c)
    wins, chgs = np.asarray(wins), np.asarray(chgs)
    stage_med = np.median(wins, axis=0)
    n_stages = wins.shape[1]
    timed_iters = repeats * n_stages * steps
    per_step = float(stage_med.sum() / (n_stages * steps))
    walk_totals = wins.sum(axis=1)
    res = {
        "per_step_s": per_step, "its_per_s": 1.0 / per_step, "geo": be.geometry(),
        "timing": be.timing_read() if timing else {},
        "x_passes": (be.timing_passes() / timed_iters) if timing else None,
        "trials": totals["trials"] / timed_iters, "invalid": totals["invalid_trials"] / timed_iters,
        "final_tc": float(model.tc), "x_host": x_host, "every": every,
        "kernel_names": {"gemm_nt": be.kernel_name(0), "gemm_tn": be.kernel_name(1), "gemm_nt2": be.kernel_name(2)},
        "passes_by_site": ({k: v / timed_iters for k, v in be.timing_passes_by_kind().items()} if timing else None),
        "windows": {
            "stages": n_stages, "steps_per_window": steps, "walks_timed": repeats, "walks_discarded_as_warmup": 1,
            "timed_iterations": timed_iters, "timed_seconds": float(wins.sum()),
            "ms_per_step_by_stage": [float(s / steps * 1e3) for s in stage_med],
            "ms_per_step_walk_min_median_max": [float(np.min(walk_totals) / (n_stages * steps) * 1e3),
                                                float(np.median(walk_totals) / (n_stages * steps) * 1e3),
                                                float(np.max(walk_totals) / (n_stages * steps) * 1e3)],
            "ms_per_step_by_walk": [float(t / (n_stages * steps) * 1e3) for t in walk_totals],
            "stage_change_ms_median": float(np.median(chgs[:, 1:]) * 1e3) if n_stages > 1 else None,
            "first_stage_start_ms_median": float(np.median(chgs[:, 0]) * 1e3),
            "iterations_per_sec_incl_stage_changes": float(n_stages * steps / (stage_med.sum() + np.median(chgs, axis=0).sum())),
        },
        "bytes_resident": be.bytes_resident() if hasattr(be, "bytes_resident") else None,
        "f32_gemm": getattr(model, "f32_gemm", "mfma"),
        "exchange": dict(be.exchange_info(), transport=getattr(model, "_engine_exchange", None),
                         selftest_seconds_per_y_allreduce=getattr(comm, "selftest_seconds", None),
                         line_search_in_library=bool(getattr(model, "_iterated_in_library", False))),
    }
    local_per_step = float(np.median(np.asarray(local_windows), axis=0).sum() / (n_stages * steps)) if local_windows else per_step
    res["exchange_profile"] = exchange_profile(be, comm, rank, timing, timed_iters, local_per_step, every)
    return res, model, be


EXCHANGE_SITE_WHAT = {"y": "[Y | W.W^T] behind X.W^T (reference :247 -> :259)", "direction": "[Y_g | Bj] of _sig / the merged [Y' | W'.W'^T | Y_g] (:210, :302)",
                      "scalars": "TC sums + tangent + H of an evaluation (:294, :301-305)", "small": "Bj before the merged pass, W'.W'^T of a trial by linearity, a restored H"}


def exchange_profile(be, comm, rank, timing, timed_iters, local_per_step, every):
    """Where an iteration's time goes besides the X passes when several ranks share the variables axis: the all-reduces the library
    issued in the timed windows by site (HIP events on the stream that carries each collective: an all-reduce's duration includes its
    wait for a slower rank), their count, and every rank's own ms_per_step up to the moment ITS stream drained (the window itself ends at
    a barrier, so its length is the slowest rank's by construction).  None without exchange steps inside the library."""
    import numpy as np
    import torch
    sites = be.timing_exchange_read() if (timing and hasattr(be, "timing_exchange_read")) else {}
    if not any(v[0] for v in sites.values()):
        return None
    ms_it = {k: (issued / timed_iters) * (ms / n) for k, (issued, n, ms) in sites.items() if n}
    prof = {"allreduces_per_iteration": sum(v[0] for v in sites.values()) / timed_iters,
            "allreduces_per_iteration_by_site": {k: v[0] / timed_iters for k, v in sites.items() if v[0]},
            "exchange_ms_per_iteration": dict(ms_it, total=sum(ms_it.values())),
            "avg_us_by_site": {k: 1e3 * ms / n for k, (issued, n, ms) in sites.items() if n},
            "timed_every_nth": every, "sites": {k: EXCHANGE_SITE_WHAT[k] for k in ms_it},
            "ms_per_step_this_rank_until_its_stream_drained": local_per_step * 1e3}
    vals = [local_per_step * 1e3, prof["exchange_ms_per_iteration"]["total"]]
    if comm is not None and comm.world > 1:
        import torch.distributed as dist
        t = torch.tensor(vals, dtype=torch.float64, device="cuda")
        parts = [torch.empty_like(t) for _ in range(comm.world)]
        dist.all_gather(parts, t)
        allv = np.asarray([[float(u) for u in p.cpu()] for p in parts])
    else:
        allv = np.asarray([vals])
    prof["ms_per_step_by_rank"] = [float(u) for u in allv[:, 0]]
    prof["ms_per_step_rank_min_median_max"] = [float(np.min(allv[:, 0])), float(np.median(allv[:, 0])), float(np.max(allv[:, 0]))]
    prof["slowest_rank"] = int(np.argmax(allv[:, 0]))
    # the all-reduce time of the FASTEST-waiting rank is the transfer itself; what the others show on top is rank skew
    prof["exchange_ms_per_iteration_rank_min_median_max"] = [float(np.min(allv[:, 1])), float(np.median(allv[:, 1])), float(np.max(allv[:, 1]))]
    return prof


def roofline_of(workload, r, world):
    """Roofline object for the dominant X-streaming kernel of a measurement: algorithmic bytes / flops
    of one launch on this rank / the mean launch duration from HIP events on the engine's stream."""
    import numpy as np
    n, v_per, m, tag = WORKLOADS[workload]
    es = 8 if tag == "f64" else 4
    alg_bytes = es * (n * v_per + m * v_per + n * m)
    alg_flops = 2.0 * n * v_per * m
    # the merged pass (gemm_nt2) contracts X with 2 x n_hidden columns in one read of X: twice the flops, the X bytes once
    site_bytes = {"gemm_nt2": es * (n * v_per + 2 * m * v_per + 2 * n * m)}
    site_flops = {"gemm_nt2": 2.0 * alg_flops}
    kernels = {}
    for name, (cnt, ms) in r["timing"].items():
        if cnt:
            avg = ms / cnt * 1e-3
            kernels[name] = {"launches": cnt, "avg_us": avg * 1e6, "GBps": site_bytes.get(name, alg_bytes) / avg / 1e9,
                             "TFLOPs": site_flops.get(name, alg_flops) / avg / 1e12}
    if not kernels:
        return None
    # use sites -> kernel function (rocprofv3 row).  X.B^T ("gemm_nt", :247/:210) and X^T.Y ("gemm_tn", :259/:211)
    # usually run the same instantiation (the first one on the transposed copy of X).
    by_fn = {}
    for name, k in kernels.items():
        fn = r["kernel_names"][name]
        d = by_fn.setdefault(fn, {"launches": 0, "total_us": 0.0, "use_sites": []})
        d["launches"] += k["launches"]
        d["total_us"] += k["launches"] * k["avg_us"]
        d["use_sites"].append(name)
    for fn, d in by_fn.items():
        avg = d["total_us"] / d["launches"] * 1e-6
        wide = d["use_sites"] == ["gemm_nt2"]
        d.update(avg_us=avg * 1e6, GBps=(site_bytes["gemm_nt2"] if wide else alg_bytes) / avg / 1e9,
                 TFLOPs=(site_flops["gemm_nt2"] if wide else alg_flops) / avg / 1e12)
    dom = max(by_fn, key=lambda k: by_fn[k]["total_us"])
    dom_wide = by_fn[dom]["use_sites"] == ["gemm_nt2"]
    if dom_wide:
        alg_bytes, alg_flops = site_bytes["gemm_nt2"], site_flops["gemm_nt2"]
    intensity = alg_flops / alg_bytes
    mfma_peak = FP64_MFMA_PEAK_TFLOPS if tag == "f64" else FP32_MFMA_PEAK_TFLOPS
    split = r.get("f32_gemm") == "split" and "gemm_split_kernel" in dom
    if split:
        # float32 products on the bf16 pipe: 6 bf16 MFMA products per float32 product - the matrix roof of the USEFUL float32 flops
        # is the dense bf16 peak / 6 (417 TF/s), and 64 factors then sit under the HBM roof
        mfma_peak = BF16_MFMA_PEAK_TFLOPS / SPLIT_PRODUCTS
    prof_name = workload + ("_split" if split else "")       # the profiles of the split mode are a run of their own (tools/gpu_prof.sh)
    traffic, tinfo = load_pmc_traffic(prof_name, dom)
    if intensity < mfma_peak * 1e12 / (HBM_PEAK_GBS * 1e9):
        roofline = {"bound": "hbm", "achieved": by_fn[dom]["GBps"], "peak": HBM_PEAK_GBS, "unit": "GB/s",
                    "frac": by_fn[dom]["GBps"] / HBM_PEAK_GBS, "traffic": traffic}
    else:
        roofline = {"bound": "mfma", "achieved": by_fn[dom]["TFLOPs"], "peak": mfma_peak, "unit": "TFLOP/s",
                    "frac": by_fn[dom]["TFLOPs"] / mfma_peak, "traffic": traffic}
    mfma_ceiling = MEASURED_CEILINGS["mfma_f64_TFLOPs" if tag == "f64" else "mfma_f32_TFLOPs"]
    if split:
        roofline["f32_gemm"] = ("split: every operand = 3 bf16 parts (exact), %d of 9 partial products, float32 accumulation; TFLOP/s "
                                "figures count the float32 flops of the contraction once" % SPLIT_PRODUCTS)
        roofline["bf16_pipe_TFLOPs"] = SPLIT_PRODUCTS * by_fn[dom]["TFLOPs"]
        roofline["frac_of_bf16_mfma_peak"] = SPLIT_PRODUCTS * by_fn[dom]["TFLOPs"] / BF16_MFMA_PEAK_TFLOPS
        roofline["x_float32_mfma_peak"] = by_fn[dom]["TFLOPs"] / FP32_MFMA_PEAK_TFLOPS
    roofline.update(workload=workload, achieved_GBps=by_fn[dom]["GBps"], achieved_TFLOPs=by_fn[dom]["TFLOPs"],
                    frac_of_measured_hbm_read_ceiling=by_fn[dom]["GBps"] / MEASURED_CEILINGS["hbm_read_GBps"],
                    frac_of_measured_mfma_ceiling=by_fn[dom]["TFLOPs"] / mfma_ceiling,
                    measured_ceilings=MEASURED_CEILINGS, kernel=dom, avg_launch_us=by_fn[dom]["avg_us"],
                    launches=by_fn[dom]["launches"], timed_every_nth_launch=r["every"],
                    algorithmic_bytes_per_launch=alg_bytes, algorithmic_flops_per_launch=alg_flops,
                    arithmetic_intensity_flop_per_byte=intensity, use_sites=kernels)
    roofline.update(tinfo)
    roofline["by_kernel"] = by_fn
    # per use site, whatever function is the dominant row (on the panel-major copy X.B^T and X^T.Y run on two kernel functions,
    # gemm_cr / gemm_ct, and the merged pass on a third): every site's fraction of the same peak
    peak_site = HBM_PEAK_GBS if roofline["bound"] == "hbm" else mfma_peak
    roofline["frac_by_site"] = {name: (k["GBps"] if roofline["bound"] == "hbm" else k["TFLOPs"]) / peak_site for name, k in kernels.items()}
    roofline["kernel_by_site"] = {name: r["kernel_names"][name] for name in kernels}
    if "gemm_nt2" in kernels and not dom_wide:
        # the merged 2 x n_hidden-column pass (one read of X, twice the flops): its own fraction of the same peak
        k2 = kernels["gemm_nt2"]
        roofline["merged_pass"] = {"kernel": r["kernel_names"]["gemm_nt2"], "avg_launch_us": k2["avg_us"], "launches": k2["launches"],
                                   "achieved_TFLOPs": k2["TFLOPs"], "frac": k2["TFLOPs"] / mfma_peak}
    rp_us, rp_src = load_rocprof_avg(prof_name, dom)
    roofline.update(rocprofv3_avg_kernel_us=rp_us, rocprofv3_source=rp_src)
    # whole-iteration view: algorithmic bytes / flops of the X passes an iteration makes over the iteration time
    if r["x_passes"]:
        ps = r.get("passes_by_site") or {}
        base_b, base_f = es * (n * v_per + m * v_per + n * m), 2.0 * n * v_per * m
        it_bytes = sum(ps.get(k, 0.0) * site_bytes.get(k, base_b) for k in ("gemm_nt", "gemm_tn", "gemm_nt2"))
        it_flops = sum(ps.get(k, 0.0) * site_flops.get(k, base_f) for k in ("gemm_nt", "gemm_tn", "gemm_nt2"))
        in_pass = sum(ps.get(k, 0.0) * kernels[k]["avg_us"] for k in kernels) * 1e-6
        roofline["iteration"] = {"x_passes": r["x_passes"], "x_passes_by_site": ps,
                                 "achieved_GBps": it_bytes / r["per_step_s"] / 1e9,
                                 "achieved_TFLOPs": it_flops / r["per_step_s"] / 1e12,
                                 "fraction_of_step_inside_the_x_passes": in_pass / r["per_step_s"]}
        # the whole step against the same roof: the algorithmic flops (bytes) of the X passes an iteration executes / ms_per_step / peak
        # - launch gaps, the small kernels and the exchange steps all count against it.  `frac` above is the dominant kernel alone
        roofline["step_frac"] = ((it_bytes / r["per_step_s"] / 1e9) if roofline["bound"] == "hbm" else (it_flops / r["per_step_s"] / 1e12)) / peak_site
    # the slowest of the pass sites (`frac` is the dominant kernel's, which on the merged-pass workloads is the BEST of the three)
    roofline["frac_min_site"] = min(roofline["frac_by_site"].values())
    return roofline


def config_of(workload, r, world, line_search, force_exchange=False):
    n, v_per, m, tag = WORKLOADS[workload]
    v_total = v_per * world
    return {"workload": "%s: synthetic Gaussian X %d samples x %d variables%s, n_hidden=%d, %s%s; 7-stage annealing, one "
                        "timed window of %d iterations per stage"
                        % (workload, n, v_total, " (%d per GPU, variable-sharded)" % v_per if world > 1 else "", m, tag,
                           " (%s)" % DESCRIPTION[workload] if workload in DESCRIPTION else "", r["windows"]["steps_per_window"]),
            "n_samples": n, "n_variables_total": v_total, "n_variables_per_gpu": v_per, "n_hidden": m,
            "fit_iterations_per_sec": r["its_per_s"],
            "line_search_trials_per_iteration": r["trials"], "invalid_trials_per_iteration": r["invalid"],
            "line_search": line_search, "f32_gemm": r.get("f32_gemm") if tag == "f32" else None,
            "x_passes_per_iteration": r["x_passes"],
            "x_passes_per_iteration_reference_shaped": 2 + 2 * r["trials"] - r["invalid"],
            "windows": r["windows"], "launch_geometry": r["geo"], "final_TC": r["final_tc"],
            "bytes_resident": r["bytes_resident"], "force_exchange": bool(force_exchange),
            # who issues the all-reduces of the exchange steps: "rccl" = the library, on a communicator the handle owns
            # (include/lcx.h lcx_comm_init); "hook" = the library through the caller's transport; None / "caller" = the
            # host-sequenced path (LCX_EXCHANGE=torch); kind "none" = one rank, no exchange steps
            "exchange": r.get("exchange"), "exchange_profile": r.get("exchange_profile")}




def other_gemm_name(args):
    return "f32_gemm_split" if args.f32_gemm == "mfma" else "f32_gemm_mfma"


def other_gemm_block(args, comm, world, rank, local_rank, workload, steps, warmup, line_search, r_head):
    """The same workload with the OTHER arithmetic of the float32 X passes (lcx_set_f32_gemm): reported beside `value`, never as it.
    With the default --f32-gemm mfma this is the bf16 split: same iterations, same trajectory to float32 rounding (the final TC
    of both is in the block), X passes 1.3-1.5 x faster."""
    other = "split" if args.f32_gemm == "mfma" else "mfma"
    r7, model7, be7 = measure(args, comm, world, rank, local_rank, workload, steps, warmup, line_search, f32_gemm=other)
    rl = roofline_of(workload, r7, world) or {}
    blk = {"f32_gemm": r7["f32_gemm"], "fit_iterations_per_sec": r7["its_per_s"] * world, "ms_per_step": r7["per_step_s"] * 1e3,
           "speedup_vs_value_mode": r7["its_per_s"] / r_head["its_per_s"],
           "x_passes_per_iteration": r7["x_passes"], "line_search_trials_per_iteration": r7["trials"],
           "final_TC": r7["final_tc"], "final_TC_value_mode": r_head["final_tc"],
           "final_TC_relative_difference": abs(r7["final_tc"] - r_head["final_tc"]) / max(1.0, abs(r_head["final_tc"])),
           "ms_per_step_walk_min_median_max": r7["windows"]["ms_per_step_walk_min_median_max"],
           "roofline": {k: rl.get(k) for k in ("bound", "achieved", "peak", "unit", "frac", "kernel", "avg_launch_us", "launches", "achieved_GBps",
                                               "achieved_TFLOPs", "bf16_pipe_TFLOPs", "frac_of_bf16_mfma_peak", "x_float32_mfma_peak",
                                               "frac_by_site", "kernel_by_site", "f32_gemm")},
           "pass_ms": {k: v["avg_us"] / 1e3 for k, v in (rl.get("use_sites") or {}).items()}}
    be7.close()
    model7._backend = None
    return blk


def linear_mode_block(workload, r3, world):
    """The same iterations with the linear trial mode (SURVEY.md section 7 'linearity shortcut'): trials cost no
    pass over X.  Reported beside the reference-shaped figure, never as `value`; the roofline is computed from the flops
    this mode actually executes."""
    rl = roofline_of(workload, r3, world) or {}
    it = rl.get("iteration") or {}
    # (whole-job figure like `value`: every rank's shard iterations)
    return {"fit_iterations_per_sec": r3["its_per_s"] * world, "ms_per_step": r3["per_step_s"] * 1e3,
            "x_passes_per_iteration": r3["x_passes"], "line_search_trials_per_iteration": r3["trials"],
            "final_TC": r3["final_tc"], "refresh_every": 16,
            "ms_per_step_walk_min_median_max": r3["windows"]["ms_per_step_walk_min_median_max"],
            "roofline": {k: rl.get(k) for k in ("bound", "achieved", "peak", "unit", "frac", "kernel", "avg_launch_us", "launches")},
            "iteration_executed_TFLOPs": it.get("achieved_TFLOPs"), "iteration_executed_GBps": it.get("achieved_GBps")}


def planted_groups(seed, n_variables, n_groups, col_offset=0):
    """Group of every column of lcx_generate_x(kind=1) (moment_kernels.hpp generate_kernel: mix64(seed*31 + column) mod groups)."""
    import numpy as np
    with np.errstate(over="ignore"):
        z = np.uint64(seed) * np.uint64(31) + (np.arange(n_variables, dtype=np.uint64) + np.uint64(col_offset))
        z = z + np.uint64(0x9E3779B97F4A7C15)
        z = (z ^ (z >> np.uint64(30))) * np.uint64(0xBF58476D1CE4E5B9)
        z = (z ^ (z >> np.uint64(27))) * np.uint64(0x94D049BB133111EB)
        z = z ^ (z >> np.uint64(31))
    return (z % np.uint64(n_groups)).astype(np.int64)


def cluster_purity(clusters, groups, n_groups):
    """fraction of variables whose cluster's majority planted group is their own"""
    import numpy as np
    table = np.zeros((int(clusters.max()) + 1, n_groups), np.int64)
    np.add.at(table, (clusters, groups), 1)
    return float(table.max(axis=1).sum() / len(clusters))


def convergence_block(n, v, m, dtype, device, kind, max_iter, seed=1, f32_gemm=None):
    """BASELINE.json's second figure: wall-clock of a whole fit() of a generated workload - data generation and
    standardisation on the device, 7 annealing stages each to |dTC| < tol = 1e-5 (reference defaults :72-74, :152-155)
    or `max_iter` iterations, final detail moments and factor sort (:160-163)."""
    import numpy as np
    from linearcorex_amd import Corex
    t0 = time.perf_counter()
    mdl = Corex(n_hidden=m, seed=0, dtype=dtype, device=device, max_iter=max_iter, f32_gemm=f32_gemm)
    mdl.fit_generated(n, v, seed=seed, kind=kind, n_groups=m)
    t1 = time.perf_counter()
    n_it = len(mdl.history["TC"])
    blk = {"f32_gemm": getattr(mdl, "f32_gemm", None) if np.dtype(dtype) == np.float32 else None,
           "data": "planted: %d latent groups + unit noise per variable (lcx_generate_x kind 1)" % m if kind == 1
                   else "iid N(0,1) (lcx_generate_x kind 0): no structure to converge to",
           "seconds": t1 - t0, "iterations": n_it, "TC": float(mdl.tc), "tol": 1e-5,
           "max_iter_per_stage": max_iter, "iterations_by_stage": list(mdl.stage_iterations),
           "stages_converged_before_the_cap": int(sum(1 for k in mdl.stage_iterations if k < max_iter)),
           "iterations_per_sec_incl_setup": n_it / (t1 - t0),
           "trials_per_iteration": mdl.stats["trials"] / max(1, n_it)}
    if kind == 1:
        blk["cluster_purity_vs_planted_groups"] = cluster_purity(mdl.clusters(), planted_groups(seed, v, m), m)
        blk["distinct_clusters"] = int(len(np.unique(mdl.clusters())))
    mdl._backend.close()
    del mdl
    return blk


def covariance_block(model, be, label):
    """get_covariance() (reference :443-451) of the resident solution: seconds (median of 3 calls, each into a freshly
    allocated NumPy matrix) and the device time of the product kernels."""
    import numpy as np
    secs, dev = [], []
    cov = None
    for _ in range(3):
        del cov
        t0 = time.perf_counter()
        cov = model.get_covariance()
        secs.append(time.perf_counter() - t0)
        d = be.last_covariance_device_seconds() if hasattr(be, "last_covariance_device_seconds") else None
        if d:
            dev.append(d)
    nv, nbytes = cov.shape[0], cov.nbytes
    sec, dev_s = float(np.median(secs)), (float(np.median(dev)) if dev else None)
    out = {"workload": label, "n_variables": int(nv), "dtype": cov.dtype.name, "output_bytes": int(nbytes),
           "seconds": sec, "seconds_each_call": secs, "host_GBps": nbytes / sec / 1e9,
           "kernel_seconds": dev_s, "kernel_write_GBps": (nbytes / dev_s / 1e9) if dev_s else None,
           "diag_is_var": bool(np.allclose(np.diag(cov), np.asarray(model.theta[1], dtype=cov.dtype) ** 2))}
    del cov
    return out


# ------------------------------------------------------------------------------------------------------
# the stdout line: compact by construction.  The driver keeps a few KB of stdout; round 3's 21 KB line could not be parsed.
# ------------------------------------------------------------------------------------------------------


class _Park:
    """Keeps the ranks that do not time the CPU baseline off the cores: they sleep-poll for a file rank 0 creates when it is done.
    LCX_TEST_PARK=spin (test hook) makes them busy-wait instead - what waiting in an RCCL barrier amounts to - so that the difference
    can be measured on a rehearsal."""

    def __init__(self, world, rank):
        self.world, self.rank = world, rank
        # named by the rendezvous port alone (every attempt of the rank launchers has its own): the ranks need not share a parent process -
        # under a foreign launcher each is the child of its own supervisor.  Rank 0 clears a stale file here, BEFORE the barrier behind
        # which the others start to poll
        self.path = os.path.join("/tmp", "lcx_bench_park_%s" % os.environ.get("MASTER_PORT", "0"))
        self.spin = os.environ.get("LCX_TEST_PARK") == "spin"
        self.how = ("busy-waiting (LCX_TEST_PARK=spin: emulates ranks waiting in an RCCL barrier)" if self.spin
                    else "parked: sleeping poll (0.2 s) for a file rank 0 creates when the baseline is done")
        if rank == 0 and os.path.exists(self.path):
            os.remove(self.path)

    def release(self):
        if self.world > 1:
            with open(self.path, "w") as f:
                f.write("done\n")

    def wait(self, limit_s=3600.0):
        t_end, t_next = time.time() + limit_s, 0.0
        while time.time() < t_end:
            if self.spin:
                if time.time() < t_next:
                    continue
                t_next = time.time() + 0.2
            else:
                time.sleep(0.2)
            if os.path.exists(self.path):
                return
        raise SystemExit("bench.py: rank %d waited %d s for rank 0's CPU baseline" % (self.rank, int(limit_s)))

    def cleanup(self):
        if self.rank == 0 and os.path.exists(self.path):
            os.remove(self.path)


def main():
    args = parse()
    if args.gpus > 1 and "WORLD_SIZE" not in os.environ:
        return spawn_ranks(args)
    if args.gpus > 1 and not os.environ.get("LCX_BENCH_WORKER") and int(os.environ.get("WORLD_SIZE", "1")) == args.gpus:
        # a rank of somebody else's launcher (the driver's torch.distributed.run): supervise the real rank as a child, so that a hung
        # first contact costs an attempt of the transport ladder instead of the job (benchkit/launch.py)
        return supervise_rank(args)
    # ONE JSON line on stdout, nothing else: RCCL prints a version banner to the C-level stdout (buffered, so it lands
    # after anything Python printed).  Everything written to fd 1 from here on goes to stderr; rank 0 writes the JSON line
    # to the saved descriptor at the end.
    sys.stdout.flush()
    real_stdout = os.dup(1)
    os.dup2(2, 1)
    world = int(os.environ.get("WORLD_SIZE", "1"))
    rank = int(os.environ.get("RANK", "0"))
    local_rank = int(os.environ.get("LOCAL_RANK", "0"))
    if world != args.gpus:
        sys.stderr.write("bench.py: --gpus %d but WORLD_SIZE=%d\n" % (args.gpus, world))
        return 2

    # torch first: its wheel carries its own HIP / HSA runtime, and liblcx_hip.so must resolve against the copy that is
    # already loaded (two HSA runtimes in one process leave the second one without devices).  Importing is not a GPU call.
    import numpy as np
    import torch
    # then the library, before anything initialises the GPU: a stale library is compiled here (hipcc is a child
    # process), except under a profiler whose preloaded library would be inherited by the compiler
    import __graft_entry__ as ge
    if ge._stale() and any(k in os.environ for k in ("ROCP_TOOL_LIBRARIES", "ROCPROFILER_REGISTER_FORCE_LOAD")):
        sys.stderr.write("bench.py: liblcx_hip.so is stale and a profiler is attached - run `python __graft_entry__.py` first\n")
        return 3
    ge.build(probe=False)
    if not torch.cuda.is_available():
        raise SystemExit("bench.py needs a GPU (the HIP path is the only product path)")
    # test hooks: LCX_BENCH_DEVICE pins every rank to one device and LCX_BENCH_BACKEND=gloo replaces RCCL, so that the
    # multi-rank code of this file can be exercised on a one-GPU box (RCCL refuses two ranks per device)
    if os.environ.get("LCX_BENCH_DEVICE"):
        local_rank = int(os.environ["LCX_BENCH_DEVICE"])
    backend = os.environ.get("LCX_BENCH_BACKEND", "nccl")
    torch.cuda.set_device(local_rank)
    comm = None
    lean = bool(os.environ.get("LCX_BENCH_LEAN"))           # a fall-back attempt of the rank launcher: headline, same-shard pre-run, CPU baseline
    if world > 1:
        import torch.distributed as dist
        from linearcorex_amd.comm import Comm, _FirstContactWatchdog
        # torch.distributed's own first contact (its RCCL communicator comes up inside the first collective - the one Comm() issues)
        # is bounded like the library's: a rank stuck here exits 3 and the launcher starts a fresh set (benchkit/launch.py)
        with _FirstContactWatchdog(rank) as dog:
            dog.step("torch.distributed.init_process_group(%r)" % backend)
            if backend == "nccl":
                dist.init_process_group("nccl", device_id=torch.device("cuda", local_rank))
            else:
                dist.init_process_group(backend)
            dog.step("first collective on the process group (Comm(): shard bounds check)")
            comm = Comm()
        if comm.world != args.gpus:
            sys.stderr.write("bench.py: the process group has %d ranks, --gpus %d\n" % (comm.world, args.gpus))
            return 2
    if world == 1 and args.force_exchange:
        import torch.distributed as dist
        os.environ.setdefault("MASTER_ADDR", "127.0.0.1")
        if "MASTER_PORT" not in os.environ:
            import socket
            with socket.socket() as sk:
                sk.bind(("127.0.0.1", 0))
                os.environ["MASTER_PORT"] = str(sk.getsockname()[1])
        dist.init_process_group("nccl", rank=0, world_size=1, device_id=torch.device("cuda", local_rank))
        from linearcorex_amd.comm import Comm
        comm = Comm(always_exchange=True)

    auto = args.workload == "auto"
    head = ("c3" if world == 1 else "c4shard") if auto else args.workload
    if os.environ.get("LCX_BENCH_HEAD"):                 # test hook: a small headline workload on a shared one-GPU box
        head = os.environ["LCX_BENCH_HEAD"]
        _adhoc(head)
    n, v_per, m, tag = WORKLOADS[head]
    dtype = np.float64 if tag == "f64" else np.float32

    # ---- several ranks: first the SAME shard on every GPU without any exchange (each rank on its own, no collective in the
    # measurement): the one-GPU point of the weak-scaling series of THIS workload, measured in the same job.  (The N=1 line
    # of bench.py headlines another workload, configs[2], so value(N) / (N x value(1)) across lines is not an efficiency.)
    same_shard_single = None
    if world > 1 and args.extras:
        # bounded: at most 10 iterations per window, one timed walk (the --gpus 8 job has to stay within minutes)
        r1, m1, b1 = measure(args, None, 1, rank, local_rank, head, min(args.steps, 10), min(args.warmup, 3), args.line_search,
                             repeats=max(1, args.repeats))
        b1.close()
        m1._backend = None
        del m1, b1
        import torch.distributed as dist
        t = torch.tensor([r1["its_per_s"], -r1["its_per_s"]], dtype=torch.float64, device="cuda")
        dist.all_reduce(t, op=dist.ReduceOp.MIN)
        same_shard_single = {"iterations_per_sec_slowest_rank": float(t[0]), "iterations_per_sec_fastest_rank": float(-t[1]),
                             "ms_per_step_this_rank": r1["per_step_s"] * 1e3, "x_passes_per_iteration": r1["x_passes"],
                             "what": "every rank fits the same %s shard alone, no exchange steps: the one-GPU point of this "
                                     "workload's weak-scaling series, measured in this job" % head}

    # ---- headline: the reference-shaped iteration (every line-search trial re-evaluates the moments with two passes
    # over X, linearcorex.py:321) ----
    r, model, be = measure(args, comm, world, rank, local_rank, head, args.steps, args.warmup, args.line_search,
                           keep_x=True, repeats=args.repeats, min_repeats=3 if args.extras else 1)
    roofline = roofline_of(head, r, world)
    cfg = config_of(head, r, world, args.line_search, args.force_exchange)
    x_head = r.pop("x_host")
    if same_shard_single is not None:
        cfg["single_gpu_same_shard"] = same_shard_single
        cfg["weak_scaling_vs_same_shard"] = r["its_per_s"] / same_shard_single["iterations_per_sec_slowest_rank"]
        if cfg.get("exchange_profile"):
            # the same shard without any exchange step, slowest rank: ms_per_step - this - the exchange sites = launch gaps of the exchange path
            cfg["exchange_profile"]["compute_only_ms_per_step"] = 1e3 / same_shard_single["iterations_per_sec_slowest_rank"]
    if args.extras and world == 1 and comm is None and v_per <= 20000:
        cfg["get_covariance"] = covariance_block(model, be, head)
    be.close()
    model._backend = None
    del model, be
    generated = x_head is None
    if args.extras and generated and not lean:
        # the other line searches of the same workload, reported beside the headline, never as `value`: "exact" = the
        # reference-shaped iteration (every trial two passes over X, :321), "exact-y" = trials after the first one of an
        # iteration take X.w_update^T by linearity (lcx_set_trial_reuse), "linear" = trials cost no pass over X
        names = {"exact": "reference_shaped", "exact-y": "later_trials_by_linearity", "linear": "linear_trial_mode"}
        for ls in ("exact", "exact-y", "linear"):
            if ls == args.line_search:
                continue
            r3, model3, be3 = measure(args, comm, world, rank, local_rank, head, args.steps, args.warmup, ls)
            cfg[names[ls]] = dict(linear_mode_block(head, r3, world), line_search=ls,
                                  refresh_every=16 if ls == "linear" else None)
            be3.close()
            model3._backend = None
            del model3, be3
    if args.extras and generated and tag == "f32" and not lean:
        cfg[other_gemm_name(args)] = other_gemm_block(args, comm, world, rank, local_rank, head, args.steps, args.warmup, args.line_search, r)
    if args.extras and world == 1 and comm is None and generated and args.convergence_max_iter > 0:
        # BASELINE.json's second figure for the headline workload.  On the iid matrix of the throughput run there is nothing to
        # converge to (every stage runs into the cap): that run is labelled as capped, and the convergence measurement proper is
        # the planted matrix of the same shape, each stage to |dTC| < 1e-5
        cfg["fit_to_convergence"] = convergence_block(n, v_per, m, dtype, local_rank, 0, args.convergence_max_iter)
        cfg["fit_to_convergence"]["capped"] = True
        cfg["fit_to_convergence_planted"] = convergence_block(n, v_per, m, dtype, local_rank, 1, args.convergence_planted_max_iter,
                                                              f32_gemm=args.f32_gemm if tag == "f32" else None)
        if tag == "f32" and other_gemm_name(args) in cfg:
            # the same whole fit with the other arithmetic of the X passes: iterations, TC and the recovered clusters beside it
            other = "split" if args.f32_gemm == "mfma" else "mfma"
            cfg[other_gemm_name(args)]["fit_to_convergence_planted"] = convergence_block(
                n, v_per, m, dtype, local_rank, 1, args.convergence_planted_max_iter, f32_gemm=other)

    out = {
        "metric": "corex_fit_iterations_per_sec",
        "value": r["its_per_s"] * world,
        "unit": "fit iterations/s (x GPUs: every rank iterates over its own %d x %d x %d shard)" % (n, v_per, m)
                if world > 1 else "fit iterations/s at (n_samples, n_variables, n_hidden) = (%d, %d, %d)" % (n, v_per, m),
        "n_gpus": world, "steps": args.steps, "warmup": args.warmup,
        "ms_per_step": r["per_step_s"] * 1e3,
        "higher_is_better": True, "scaling": "weak", "vs_baseline": None,
        "dtype": tag if r.get("f32_gemm") != "split" else "f32 as 3 x bf16 (exact split, 6 partial products, f32 accumulate)",
        "data": "synthetic",
        "config": cfg,
        "roofline": roofline,
        "cpu_baseline": None,
    }

    if args.extras and lean:
        out["config"]["lean"] = "fall-back attempt of the rank launcher: no c2_weak block, no other line searches, no split block"
    if args.extras and not lean:
        # ---- nested block: BASELINE.json configs[1] (HBM-bound), 5k variables per GPU, its own fixed protocol ----
        c2_steps, c2_warm = 30, 5
        c2_ls = args.line_search if args.line_search != "linear" else "exact"
        r2, model2, be2 = measure(args, comm, world, rank, local_rank, "c2", c2_steps, c2_warm, c2_ls, keep_x=True)
        x2 = r2.pop("x_host")
        blk = config_of("c2", r2, world, c2_ls)
        blk["value"] = r2["its_per_s"] * world
        blk["ms_per_step"] = r2["per_step_s"] * 1e3
        blk["dtype"] = "f64"
        blk["roofline"] = roofline_of("c2", r2, world)
        if world == 1 and comm is None:
            blk["get_covariance"] = covariance_block(model2, be2, "c2")
        be2.close()
        model2._backend = None
        del model2, be2
        if world == 1:
            # the other line searches on the same X (several ranks: skipped, the --gpus N job stays within minutes)
            names = {"exact": "reference_shaped", "exact-y": "later_trials_by_linearity", "linear": "linear_trial_mode"}
            for ls in ("exact", "exact-y", "linear"):
                if ls == c2_ls:
                    continue
                r3, model3, be3 = measure(args, comm, world, rank, local_rank, "c2", c2_steps, c2_warm, ls, kernel_timing=True)
                blk[names[ls]] = {"fit_iterations_per_sec": r3["its_per_s"], "ms_per_step": r3["per_step_s"] * 1e3,
                                  "x_passes_per_iteration": r3["x_passes"], "line_search": ls,
                                  "line_search_trials_per_iteration": r3["trials"], "final_TC": r3["final_tc"]}
                be3.close()
                model3._backend = None
                del model3, be3
        if world == 1 and comm is None and x2 is not None:
            # BASELINE.json's second figure: wall-clock of a whole fit() to |dTC| < 1e-5 per annealing stage (reference
            # defaults :72-74), including the upload + on-device preprocess and the final detail moments
            from linearcorex_amd import Corex
            t0 = time.perf_counter()
            mdl = Corex(n_hidden=32, seed=0, dtype=np.float64, device=local_rank).fit(x2)
            t1 = time.perf_counter()
            blk["fit_to_convergence"] = {
                "seconds": t1 - t0, "iterations": len(mdl.history["TC"]), "trials": int(mdl.stats["trials"]), "TC": float(mdl.tc), "tol": 1e-5,
                "iterations_per_sec_incl_setup": len(mdl.history["TC"]) / (t1 - t0),
                "trials_per_iteration": mdl.stats["trials"] / max(1, len(mdl.history["TC"]))}
            mdl._backend.close()
            del mdl
            # BASELINE.json configs[4] stand-in shape: get_covariance() of a 20 000-variable model
            mdl = Corex(n_hidden=30, seed=0, dtype=np.float64, device=local_rank, max_iter=3)
            mdl.fit(np.random.RandomState(5).randn(400, 20000))
            out["config"]["get_covariance_c5_standin"] = covariance_block(mdl, mdl._backend, "c5 stand-in: 400 x 20000, n_hidden=30, f64")
            mdl._backend.close()
            del mdl
        if rank == 0 and world == 1 and comm is None and args.cpu_iters_per_stage > 0 and x2 is not None:
            blk["cpu_baseline"] = cpu_baseline_resident(x2, 32, np.float64, args.cpu_iters_per_stage)
            # "full convergence (tol=1e-5) for configs 1-2" on the host cores, beside fit_to_convergence
            blk["cpu_fit_to_convergence"] = cpu_fit_to_convergence(x2, 32, np.float64)
            if "fit_to_convergence" in blk:
                blk["fit_to_convergence"]["cpu_over_device_wall_clock"] = (blk["cpu_fit_to_convergence"]["seconds"]
                                                                           / blk["fit_to_convergence"]["seconds"])
        out["config"]["c2" if world == 1 else "c2_weak"] = blk
        del x2
        if rank == 0 and world == 1 and comm is None and auto and args.cpu_iters_per_stage > 0:
            out["config"]["c1"] = c1_block(local_rank)

        # ---- N=1, default workload: the one-GPU point of the weak-scaling series the --gpus N lines headline (configs[3]'s
        # shard), so that the series 1 -> 8 is self-contained in the driver's records ----
        c4 = None
        if auto and world == 1 and comm is None and head == "c3":
            c4_ls = args.line_search
            r4, model4, be4 = measure(args, comm, world, rank, local_rank, "c4shard", 10, 3, c4_ls)
            c4 = config_of("c4shard", r4, world, c4_ls)
            c4["value"] = r4["its_per_s"]
            c4["ms_per_step"] = r4["per_step_s"] * 1e3
            c4["dtype"] = "f32"
            c4["roofline"] = roofline_of("c4shard", r4, world)
            c4["what"] = ("the workload `python bench.py --gpus N` headlines for N > 1, on one GPU without exchange steps: "
                          "value(N) / (N x this value) is the weak-scaling efficiency of that series")
            be4.close()
            model4._backend = None
            del model4, be4
            names = {"exact": "reference_shaped", "exact-y": "later_trials_by_linearity", "linear": "linear_trial_mode"}
            for ls in ("exact", "exact-y", "linear"):
                if ls == c4_ls:
                    continue
                r5, model5, be5 = measure(args, comm, world, rank, local_rank, "c4shard", 10, 3, ls)
                c4[names[ls]] = dict(linear_mode_block("c4shard", r5, world), line_search=ls,
                                     refresh_every=16 if ls == "linear" else None)
                be5.close()
                model5._backend = None
                del model5, be5
            c4[other_gemm_name(args)] = other_gemm_block(args, comm, world, rank, local_rank, "c4shard", 10, 3, c4_ls, r4)
            out["config"]["c4shard"] = c4
            # ---- configs[3] as ONE problem on this one GPU: 50 000 x 1 000 000 x 128 float32, 200 GB of X.  Two resident copies
            # do not fit 288 GB, so the engine keeps the row-major copy only and X.B^T runs on gemm_cr (chosen by itself).  The
            # matrix is the one the 8-rank run shards: this is that run's single-process counterpart.  Short windows (an
            # iteration is ~0.45 s); not a headline
            if args.c4full_steps > 0:
                # 209 GB on one GPU: if the device cannot give that (another tenant, fragmentation) the block reports the error
                # and the line goes out without it - the headline does not depend on it
                try:
                    r6, model6, be6 = measure(args, comm, world, rank, local_rank, "c4full", args.c4full_steps, 1, "exact", repeats=1)
                    c6 = config_of("c4full", r6, world, "exact")
                    c6["value"] = r6["its_per_s"]
                    c6["ms_per_step"] = r6["per_step_s"] * 1e3
                    c6["dtype"] = "f32"
                    c6["roofline"] = roofline_of("c4full", r6, world)
                    c6["kernels"] = r6["kernel_names"]
                    be6.close()
                    model6._backend = None
                    del model6, be6
                    out["config"]["c4_unsharded_one_gpu"] = c6
                except Exception as e:          # noqa: BLE001
                    out["config"]["c4_unsharded_one_gpu"] = {"error": "%s: %s" % (type(e).__name__, e)}

    c4 = out["config"].get("c4shard") if isinstance(out["config"].get("c4shard"), dict) else None
    parked = None
    if args.extras and args.cpu_seconds > 0 and (world == 1 and comm is None or world > 1):
        # ---- CPU baseline of the headline workload: rank 0.  The other ranks must not compete for the cores meanwhile: a rank
        # waiting in an RCCL barrier sits in a device synchronisation, which the HIP runtime services by busy-waiting - N-1 spinning
        # threads inside the cgroup whose quota the BLAS pool is sized by.  They are parked on a sleeping poll instead (a file in
        # /tmp that rank 0 creates when the baseline is done; one node by the contract of --gpus N) ----
        park = parked = _Park(world, rank)
        if world > 1:
            torch.cuda.synchronize()
            comm.barrier()                      # everybody's GPU work is done; nothing is enqueued past this point
        if rank == 0:
            try:
                if x_head is not None:
                    out["cpu_baseline"] = cpu_baseline_resident(x_head, m, dtype, max(1, args.cpu_iters_per_stage))
                elif c4 is not None:
                    n4, v4, m4, _ = WORKLOADS["c4shard"]
                    both = cpu_baseline_generated(n, v_per, m, dtype, args.cpu_seconds, head, also=[("c4shard", m4, v4)])
                    out["cpu_baseline"], c4["cpu_baseline"] = both
                else:
                    out["cpu_baseline"] = cpu_baseline_generated(n, v_per, m, dtype, args.cpu_seconds, head)
                if world > 1:
                    out["cpu_baseline"]["sample"] += ("; one shard's iteration (n_variables per GPU = %d): `value` counts every "
                                                      "rank's shard iterations, so the two are in the same unit" % v_per)
                    out["cpu_baseline"]["other_ranks_while_timed"] = park.how
            finally:
                park.release()
        else:
            park.wait()

    out["series"] = series_of(out, head, world)

    # tear the process group down first and push out whatever C-level stdio still buffers (now on stderr), so that the JSON
    # line is the last thing this job writes even when the caller merges the two streams
    import ctypes
    libc = ctypes.CDLL(None)
    if comm is not None:
        import torch.distributed as dist
        dist.barrier()
        if parked is not None:
            parked.cleanup()
        libc.fflush(None)
        dist.destroy_process_group()
    libc.fflush(None)
    sys.stdout.flush()
    sys.stderr.flush()
    if rank == 0:
        emit(out, args, real_stdout)
    return 0


if __name__ == "__main__":
    sys.exit(main())
