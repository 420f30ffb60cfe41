#!/usr/bin/env python3
"""bench.py - fit iterations/sec of the Linear CorEx hot path on MI355X (BASELINE.json metric).

A "step" is one fixed-point iteration of the fit loop (reference `_update_ns` + its book-keeping,
linearcorex.py:137-151) inside the reference's 7-stage annealing schedule (:119-134); stage changes
that fall in the timed window are part of it.  Workload at N=1: BASELINE.json configs[1] -
synthetic Gaussian X, 10k samples x 5k variables, n_hidden=32, float64.  With N>1 ranks the
n_variables axis is sharded, 5k variables per GPU (weak scaling): the unit counted in `value` is
"one iteration over a 10k x 5k x 32 block", so N ranks finish N units per iteration.

    python bench.py --gpus 1 --steps 70 --warmup 7
    python -m torch.distributed.run --nproc-per-node N ... bench.py --gpus N ...

Prints ONE JSON line on rank 0.
"""
import argparse
import json
import math
import os
import sys
import time

import numpy as np

ROOT = os.path.dirname(os.path.abspath(__file__))
sys.path.insert(0, ROOT)

HBM_PEAK_GBS = 8000.0          # MI355X_MICROARCH.md: HBM3E 8.0 TB/s spec
FP64_MFMA_PEAK_TFLOPS = 78.6
FP32_MFMA_PEAK_TFLOPS = 157.3
# What this chip has been seen to sustain (profiles/r01_read_probe_c2.txt, r01_mfma_peak.txt, the c3f64* workloads:
# gemm_ct streams float64 X at 6.47 TB/s and runs v_mfma_f64_16x16x4 at 59.9 TF/s): a read-only
# stream of a 400 MB matrix 6.2 TB/s; pure-MFMA loops: v_mfma_f64_4x4x4 72 TF/s, v_mfma_f64_16x16x4 47.6 TF/s (real kernels reach
# 58-60 TF/s with it on large shards), v_mfma_f32_16x16x4
# 151 TF/s.  Reported beside the spec-based fraction, never instead of it.
MEASURED_CEILINGS = {"hbm_read_GBps": 6470.0, "mfma_f64_TFLOPs": 72.0, "mfma_f64_16x16x4_TFLOPs": 59.9,
                     "mfma_f32_TFLOPs": 151.0}

WORKLOADS = {
    # name: (n_samples, n_variables per GPU, n_hidden, dtype)
    "c2": (10000, 5000, 32, "f64"),      # BASELINE.json configs[1]
    "c3": (50000, 100000, 64, "f32"),    # configs[2] (MFMA roofline run; X generated on device)
    "c4shard": (50000, 125000, 128, "f32"),  # configs[3], one GPU's shard
    "c3f64": (50000, 50000, 64, "f64"),  # large float64 shards (gemm_ct on float64; not BASELINE lines)
    "c3f64m32": (50000, 50000, 32, "f64"),
    "c3f64m128": (50000, 50000, 128, "f64"),
    "c2f32": (10000, 5000, 32, "f32"),   # config-2 shape in the reference's own precision (not a BASELINE line)
    "c2m64": (10000, 5000, 64, "f64"), "c2m64f32": (10000, 5000, 64, "f32"), "c2m128f32": (10000, 5000, 128, "f32"),
    "mid32": (20000, 20000, 32, "f64"), "mid32f32": (20000, 20000, 32, "f32"), "mid64f32": (20000, 20000, 64, "f32"),
    "c5": (400, 20000, 30, "f64"),       # configs[4] stand-in shape (few samples, many variables; not a bench line)
    "c5f32": (400, 20000, 30, "f32"),
    "tiny": (2000, 640, 8, "f64"),       # plumbing check
}


def _adhoc(name):
    """'NxVxM:f32' -> WORKLOADS entry (probing shapes outside BASELINE.json)."""
    if name not in WORKLOADS:
        dims, tag = name.split(":")
        n, v, m = (int(t) for t in dims.split("x"))
        assert tag in ("f32", "f64")
        WORKLOADS[name] = (n, v, m, tag)


def parse():
    ap = argparse.ArgumentParser()
    ap.add_argument("--gpus", type=int, default=1)
    ap.add_argument("--steps", type=int, default=70)
    ap.add_argument("--warmup", type=int, default=7)
    ap.add_argument("--workload", default="c2",
                    help="one of %s, or an ad-hoc shard shape NxVxM:f32|f64 (not a BASELINE line)" % ", ".join(sorted(WORKLOADS)))
    ap.add_argument("--cpu-iters-per-stage", type=int, default=40,
                    help="bounded CPU-baseline sample: oracle iterations per annealing stage (0 = skip)")
    ap.add_argument("--no-kernel-timing", action="store_true")
    ap.add_argument("--timing-sample", type=int, default=5,
                    help="HIP-event pairs around every n-th X pass of the timed region (a pair costs ~5 us of stream time; "
                         "1 = every pass)")
    ap.add_argument("--no-convergence", dest="convergence", action="store_false",
                    help="skip the wall-clock-to-TC-convergence measurement (a full fit at tol=1e-5 on the same data, "
                         "reported under config.fit_to_convergence; 1 GPU, workload c2 only)")
    ap.add_argument("--force-exchange", action="store_true",
                    help="1 GPU only: run the multi-rank device path (world>1 kernels + RCCL all-reduces in a group of "
                         "one rank) to measure the fixed cost of the exchange steps")
    ap.add_argument("--line-search", default="exact", choices=["linear", "exact"],
                    help="exact (default, the headline): reference-shaped, every trial makes 2 passes over X "
                         "(linearcorex.py:321); linear: trials cost no pass over X")
    ap.add_argument("--no-also-linear", dest="also_linear", action="store_false",
                    help="skip the second measurement in linear trial mode (reported under config.linear_trial_mode)")
    args = ap.parse_args()
    _adhoc(args.workload)
    return args


def cpu_baseline(x, m, dtype, iters_per_stage):
    """The NumPy oracle (a port of the reference path, pinned to it bit for bit) on the host cores,
    same X, same schedule, bounded to iters_per_stage iterations per annealing stage."""
    from oracle import corex_oracle as O
    try:
        from threadpoolctl import threadpool_info
        threads = max([p.get("num_threads", 1) for p in threadpool_info()] or [1])
    except Exception:
        threads = os.cpu_count() or 1
    xt = O.preprocess(np.asarray(x, dtype=dtype))[0]
    stamps = []
    t0 = time.perf_counter()
    res = O.fit_ns_preprocessed(xt, m, seed=0, dtype=dtype, max_iter=iters_per_stage, tol=0.0,
                                on_iteration=lambda *a: stamps.append(time.perf_counter()), finish=False)
    t1 = time.perf_counter()
    n_it = len(res.history_tc)
    return {"value": n_it / (t1 - t0), "unit": "iterations/s", "cores": int(threads), "kind": "port",
            "sample": "%d iterations (%d per annealing stage x 7) of the same workload, NumPy %s/BLAS threads=%d, "
                      "%.1f s" % (n_it, iters_per_stage, np.__version__, threads, t1 - t0),
            "trials_per_iteration": res.n_trials / max(1, n_it)}


def load_pmc_traffic(workload, kernel):
    """HBM bytes per launch of `kernel` from the rocprofv3 PMC passes of this same command
    (tools/pmc_traffic.py writes profiles/pmc_traffic_<workload>.json on the GPU box; FETCH_SIZE is
    doubled there as MI355X_MICROARCH.md prescribes for gfx950).  None if no such profile is committed."""
    path = os.path.join(ROOT, "profiles", "pmc_traffic_%s.json" % workload)
    try:
        with open(path) as f:
            d = json.load(f)
        k = d["kernels"][kernel]
        return k["hbm_bytes_per_launch"], os.path.relpath(path, ROOT), k.get("mfma_util")
    except Exception:
        return None, None, None


def load_rocprof_avg(workload, kernel):
    """Average duration of `kernel` in the committed rocprofv3 --kernel-trace --stats summary of this same command
    (profiles/r01_rocprof_kernel_stats_<workload>.csv), reported beside the live HIP-event figure.  The event pair
    brackets the dispatch as well, so it reads a few microseconds above the profiler's begin-to-end kernel time."""
    import csv
    path = os.path.join(ROOT, "profiles", "r01_rocprof_kernel_stats_%s.csv" % workload)
    try:
        with open(path) as f:
            for row in csv.DictReader(f):
                if kernel.replace(" ", "") in row["Name"].replace(" ", ""):
                    return float(row["AverageNs"]) / 1e3, os.path.relpath(path, ROOT)
    except Exception:
        pass
    return None, None


def measure(args, comm, world, rank, local_rank, line_search, keep_x=False):
    """Warm up, then time exactly args.steps fit iterations (barrier + synchronize on both sides)."""
    import torch
    from linearcorex_amd import Corex
    n, v_per, m, tag = WORKLOADS[args.workload]
    dtype = np.float64 if tag == "f64" else np.float32
    v_total = v_per * world
    total_steps = args.warmup + args.steps
    per_stage = int(math.ceil(total_steps / 7.0))

    model = Corex(n_hidden=m, seed=0, dtype=dtype, tol=0.0, max_iter=10 ** 9, device=local_rank, comm=comm,
                  line_search=line_search)
    x_host = None
    if n * v_per * 8 <= (4 << 30):
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

    def sync():
        be.synchronize()
        torch.cuda.synchronize()
        if comm is not None:
            comm.barrier()
            torch.cuda.synchronize()

    state = {"step": 0, "t0": None, "t1": None}
    sched = model._init_weights()

    def one_step():
        if state["step"] == args.warmup:
            sync()
            if not args.no_kernel_timing:
                be.timing_reset()
                be.timing_sample(args.timing_sample)
                be.timing_enable(True)
            model.stats.update(trials=0, invalid_trials=0, moment_evals=0, refreshes=0)
            state["t0"] = time.perf_counter()
        model._iterate()
        state["step"] += 1
        if state["step"] == total_steps:
            sync()
            state["t1"] = time.perf_counter()
            be.timing_enable(False)

    for i_eps, eps in enumerate(sched):
        if state["step"] >= total_steps:
            break
        model._begin_stage(i_eps, eps)
        for _ in range(per_stage):
            if state["step"] >= total_steps:
                break
            one_step()
    elapsed = state["t1"] - state["t0"]
    if comm is not None:
        import torch.distributed as dist
        t = torch.tensor([elapsed], dtype=torch.float64, device="cuda")
        dist.all_reduce(t, op=dist.ReduceOp.MAX)
        elapsed = float(t.item())
    res = {"elapsed": elapsed, "its_per_s": args.steps / elapsed, "geo": be.geometry(),
           "timing": be.timing_read() if not args.no_kernel_timing else {},
           "x_passes": be.timing_passes() if not args.no_kernel_timing else None,
           "trials": model.stats["trials"] / max(1, args.steps),
           "invalid": model.stats["invalid_trials"] / max(1, args.steps),
           "final_tc": float(model.tc), "per_stage": per_stage, "x_host": x_host if keep_x else None,
           "kernel_names": {"gemm_nt": be.kernel_name(0), "gemm_tn": be.kernel_name(1)}}
    be.close()
    model._backend = None
    return res


def main():
    args = parse()
    # ONE JSON line on stdout, nothing else: RCCL prints a version banner to the C-level stdout (buffered, so it lands
    # after anything Python printed).  Everything written to fd 1 from here on goes to stderr; rank 0 writes the JSON line
    # to the saved descriptor at the end.
    sys.stdout.flush()
    real_stdout = os.dup(1)
    os.dup2(2, 1)
    world = int(os.environ.get("WORLD_SIZE", "1"))
    rank = int(os.environ.get("RANK", "0"))
    local_rank = int(os.environ.get("LOCAL_RANK", "0"))
    import torch
    if not torch.cuda.is_available():
        raise SystemExit("bench.py needs a GPU (the HIP path is the only product path)")
    # test hooks: LCX_BENCH_DEVICE pins every rank to one device and LCX_BENCH_BACKEND=gloo replaces RCCL, so that the
    # multi-rank code of this file can be exercised on a one-GPU box (RCCL refuses two ranks per device)
    if os.environ.get("LCX_BENCH_DEVICE"):
        local_rank = int(os.environ["LCX_BENCH_DEVICE"])
    backend = os.environ.get("LCX_BENCH_BACKEND", "nccl")
    torch.cuda.set_device(local_rank)
    comm = None
    if world > 1:
        import torch.distributed as dist
        if backend == "nccl":
            dist.init_process_group("nccl", device_id=torch.device("cuda", local_rank))
        else:
            dist.init_process_group(backend)
        from linearcorex_amd.comm import Comm
        comm = Comm()
    assert world == args.gpus or world == 1, "launch with torch.distributed.run for --gpus > 1"
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

    import __graft_entry__ as ge
    ge.build()

    n, v_per, m, tag = WORKLOADS[args.workload]
    dtype = np.float64 if tag == "f64" else np.float32
    v_total = v_per * world
    es = np.dtype(dtype).itemsize

    # headline: the reference-shaped iteration (every line-search trial re-evaluates the moments
    # with two passes over X, linearcorex.py:321)
    r = measure(args, comm, world, rank, local_rank, args.line_search, keep_x=True)
    elapsed, its_per_s, timing, geo = r["elapsed"], r["its_per_s"], r["timing"], r["geo"]
    trials, invalid, per_stage = r["trials"], r["invalid"], r["per_stage"]

    # algorithmic bytes / flops of ONE launch of an X-streaming GEMM on this rank:
    alg_bytes = es * (n * v_per + m * v_per + n * m)
    alg_flops = 2.0 * n * v_per * m
    kernels = {}
    for name, (cnt, ms) in timing.items():
        if cnt:
            avg = ms / cnt * 1e-3
            kernels[name] = {"launches": cnt, "avg_us": avg * 1e6, "GBps": alg_bytes / avg / 1e9,
                             "TFLOPs": alg_flops / avg / 1e12}
    # use sites -> kernel function (rocprofv3 row).  X.B^T ("gemm_nt", :247/:210) and X^T.Y ("gemm_tn",
    # :259/:211) usually run the same instantiation (the first one on the transposed copy of X).
    by_fn = {}
    for name, k in kernels.items():
        fn = r["kernel_names"][name]
        d = by_fn.setdefault(fn, {"launches": 0, "total_us": 0.0, "use_sites": []})
        d["launches"] += k["launches"]
        d["total_us"] += k["launches"] * k["avg_us"]
        d["use_sites"].append(name)
    for fn, d in by_fn.items():
        avg = d["total_us"] / d["launches"] * 1e-6
        d.update(avg_us=avg * 1e6, GBps=alg_bytes / avg / 1e9, TFLOPs=alg_flops / avg / 1e12)
    roofline = None
    if kernels:
        dom = max(by_fn, key=lambda k: by_fn[k]["total_us"])
        use_sites = kernels
        kernels = by_fn
        intensity = alg_flops / alg_bytes
        mfma_peak = FP64_MFMA_PEAK_TFLOPS if tag == "f64" else FP32_MFMA_PEAK_TFLOPS
        traffic, traffic_src, mfma_util = load_pmc_traffic(args.workload, dom)
        if intensity < mfma_peak * 1e12 / (HBM_PEAK_GBS * 1e9):
            roofline = {"bound": "hbm", "achieved": kernels[dom]["GBps"], "peak": HBM_PEAK_GBS, "unit": "GB/s",
                        "frac": kernels[dom]["GBps"] / HBM_PEAK_GBS, "traffic": traffic}
        else:
            roofline = {"bound": "mfma", "achieved": kernels[dom]["TFLOPs"], "peak": mfma_peak, "unit": "TFLOP/s",
                        "frac": kernels[dom]["TFLOPs"] / mfma_peak, "traffic": traffic}
        mfma_ceiling = MEASURED_CEILINGS["mfma_f64_TFLOPs" if tag == "f64" else "mfma_f32_TFLOPs"]
        roofline.update(achieved_GBps=kernels[dom]["GBps"], achieved_TFLOPs=kernels[dom]["TFLOPs"],
                        frac_of_measured_hbm_read_ceiling=kernels[dom]["GBps"] / MEASURED_CEILINGS["hbm_read_GBps"],
                        frac_of_measured_mfma_ceiling=kernels[dom]["TFLOPs"] / mfma_ceiling,
                        measured_ceilings=MEASURED_CEILINGS)
        roofline.update(kernel=dom, avg_launch_us=kernels[dom]["avg_us"], launches=kernels[dom]["launches"],
                        timed_every_nth_launch=args.timing_sample,
                        algorithmic_bytes_per_launch=alg_bytes, algorithmic_flops_per_launch=alg_flops,
                        traffic_source=traffic_src, mfma_util_pmc=mfma_util, use_sites=use_sites)
        rp_us, rp_src = load_rocprof_avg(args.workload, dom)
        roofline.update(rocprofv3_avg_kernel_us=rp_us, rocprofv3_source=rp_src)

    extra = None
    if args.also_linear and args.line_search == "exact":
        # same iterations with the linear trial mode: reported beside the headline, never as it
        r2 = measure(args, comm, world, rank, local_rank, "linear")
        extra = {"fit_iterations_per_sec": r2["its_per_s"], "ms_per_step": r2["elapsed"] / args.steps * 1e3,
                 "x_passes_per_iteration": (r2["x_passes"] / max(1, args.steps))
                 if r2["x_passes"] is not None else None,
                 "line_search_trials_per_iteration": r2["trials"], "final_TC": r2["final_tc"]}

    out = {
        "metric": "corex_fit_iterations_per_sec",
        "value": its_per_s * world,
        "unit": "iterations/s (10k x 5k x 32 block-iterations; = fit iterations/s at 1 GPU)" if args.workload == "c2"
                else "iterations/s x GPUs",
        "n_gpus": world, "steps": args.steps, "warmup": args.warmup,
        "ms_per_step": elapsed / args.steps * 1e3,
        "higher_is_better": True, "scaling": "weak", "vs_baseline": None,
        "dtype": tag, "data": "synthetic",
        "config": {"workload": "%s: synthetic Gaussian X %d samples x %d variables%s, n_hidden=%d, %s, "
                               "7-stage annealing, %d iterations per stage"
                               % (args.workload, n, v_total, " (%d per GPU, variable-sharded)" % v_per if world > 1 else "",
                                  m, tag, per_stage),
                   "n_samples": n, "n_variables_total": v_total, "n_variables_per_gpu": v_per, "n_hidden": m,
                   "fit_iterations_per_sec": its_per_s,
                   "line_search_trials_per_iteration": trials, "invalid_trials_per_iteration": invalid,
                   "line_search": args.line_search,
                   "x_passes_per_iteration": (r["x_passes"] / max(1, args.steps)) if r["x_passes"] is not None
                   else None,
                   "x_passes_per_iteration_reference_shaped": 2 + 2 * trials - invalid,
                   "launch_geometry": geo, "final_TC": r["final_tc"], "force_exchange": bool(args.force_exchange),
                   "linear_trial_mode": extra},
        "roofline": roofline,
    }
    x_host = r["x_host"]
    if world == 1 and args.convergence and args.workload == "c2" and x_host is not None and comm is None:
        # BASELINE.json's second figure: wall-clock of a whole fit() to |dTC| < 1e-5 per annealing stage
        # (reference defaults :72-74), including the upload + on-device preprocess and the final detail moments
        from linearcorex_amd import Corex
        t0 = time.perf_counter()
        mdl = Corex(n_hidden=m, seed=0, dtype=dtype, device=local_rank).fit(x_host)
        t1 = time.perf_counter()
        out["config"]["fit_to_convergence"] = {
            "seconds": t1 - t0, "iterations": len(mdl.history["TC"]), "TC": float(mdl.tc), "tol": 1e-5,
            "iterations_per_sec_incl_setup": len(mdl.history["TC"]) / (t1 - t0),
            "trials_per_iteration": mdl.stats["trials"] / max(1, len(mdl.history["TC"]))}
        mdl._backend.close()
    if rank == 0 and world == 1 and args.cpu_iters_per_stage > 0 and x_host is not None:
        out["cpu_baseline"] = cpu_baseline(x_host, m, dtype, args.cpu_iters_per_stage)
    elif rank == 0:
        out["cpu_baseline"] = None
    # tear the process group down first and push out whatever C-level stdio still buffers (now on stderr), so that the JSON
    # line is the last thing this job writes even when the caller merges the two streams
    import ctypes
    libc = ctypes.CDLL(None)
    if comm is not None:
        import torch.distributed as dist
        dist.barrier()
        libc.fflush(None)
        dist.destroy_process_group()
    libc.fflush(None)
    sys.stdout.flush()
    sys.stderr.flush()
    if rank == 0:
        os.write(real_stdout, (json.dumps(out) + "\n").encode())


if __name__ == "__main__":
    main()
