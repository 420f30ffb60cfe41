"""The record bench.py prints: ONE compact JSON line on stdout (the contract keys, `roofline`, `cpu_baseline`, `series`, scalar riders of
the nested measurements - a few KB, budgeted) and the full record in a side file + one BENCH_DETAIL line on stderr.  The driver keeps
only a few KB of stdout; round 3's 21 KB line could not be parsed."""
import json
import math
import os
import sys

from .workloads import WORKLOADS

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))


def _r(x, digits=6):
    """floats to `digits` significant digits (bytes on the line, not precision anybody reads)"""
    if isinstance(x, bool) or x is None:
        return x
    if isinstance(x, float):
        if not math.isfinite(x):
            return None
        return float("%.*g" % (digits, x))
    if isinstance(x, (list, tuple)):
        return [_r(v, digits) for v in x]
    if isinstance(x, dict):
        return {k: _r(v, digits) for k, v in x.items()}
    return x


def _pick(d, *path, default=None):
    for k in path:
        if not isinstance(d, dict) or k not in d:
            return default
        d = d[k]
    return d


ROOFLINE_LINE_KEYS = ("bound", "achieved", "peak", "unit", "frac", "step_frac", "frac_min_site", "traffic", "kernel", "avg_launch_us", "launches",
                      "algorithmic_bytes_per_launch", "algorithmic_flops_per_launch", "rocprofv3_avg_kernel_us",
                      "rocprofv3_source", "traffic_source", "traffic_profile_matches_library", "mfma_util_pmc")


def compact_line(out, detail_path):
    """The record the driver parses: contract keys + full `roofline` (scalars) + `cpu_baseline` (one-sentence sample) +
    scalar riders of the nested measurements.  Everything else lives in the detail record."""
    cfg = out["config"]
    c = {k: cfg.get(k) for k in ("workload", "n_samples", "n_variables_total", "n_variables_per_gpu", "n_hidden", "line_search",
                                 "x_passes_per_iteration", "x_passes_per_iteration_reference_shaped",
                                 "line_search_trials_per_iteration")}
    w = cfg.get("windows") or {}
    c["windows"] = {k: w.get(k) for k in ("walks_timed", "timed_iterations", "timed_seconds", "ms_per_step_walk_min_median_max")}
    c["exchange"] = _pick(cfg, "exchange", "transport") or _pick(cfg, "exchange", "kind")
    if _pick(cfg, "exchange", "selftest_seconds_per_y_allreduce"):
        c["y_allreduce_selftest_ms"] = 1e3 * _pick(cfg, "exchange", "selftest_seconds_per_y_allreduce")
    c["bytes_resident_total"] = _pick(cfg, "bytes_resident", "total")
    c["bytes_resident_x"] = _pick(cfg, "bytes_resident", "x")
    c["x_layout"] = _pick(cfg, "bytes_resident", "x_layout")
    if cfg.get("force_exchange"):
        c["force_exchange"] = True
    xp = cfg.get("exchange_profile")
    if xp:
        # several ranks (or --force-exchange): where the time outside the X passes goes - all-reduces by site, rank skew, the same shard alone
        c["exchange_ms_per_iteration"] = _r(xp.get("exchange_ms_per_iteration"), 4)
        c["allreduces_per_iteration"] = _r(xp.get("allreduces_per_iteration"), 4)
        c["ms_per_step_rank_min_median_max"] = _r(xp.get("ms_per_step_rank_min_median_max"), 5)
        c["exchange_ms_rank_min_median_max"] = _r(xp.get("exchange_ms_per_iteration_rank_min_median_max"), 4)
        if xp.get("compute_only_ms_per_step") is not None:
            c["compute_only_ms_per_step"] = _r(xp["compute_only_ms_per_step"], 5)
    riders = {
        # the other line searches of the same workload, reported beside `value`
        "reference_shaped_value": _pick(cfg, "reference_shaped", "fit_iterations_per_sec"),
        "exact_y_value": _pick(cfg, "later_trials_by_linearity", "fit_iterations_per_sec"),
        "linear_value": _pick(cfg, "linear_trial_mode", "fit_iterations_per_sec"),
        # the X passes of the float32 workload on the bf16 matrix pipe (exact 3-way split, lcx_set_f32_gemm): beside `value`
        "f32_gemm": cfg.get("f32_gemm"),
        "f32_gemm_split_value": _pick(cfg, "f32_gemm_split", "fit_iterations_per_sec"),
        "f32_gemm_split_final_TC_rel_diff": _pick(cfg, "f32_gemm_split", "final_TC_relative_difference"),
        "f32_gemm_split_roofline_bound": _pick(cfg, "f32_gemm_split", "roofline", "bound"),
        "f32_gemm_split_roofline_frac": _pick(cfg, "f32_gemm_split", "roofline", "frac"),
        "f32_gemm_split_fit_to_convergence_planted_seconds": _pick(cfg, "f32_gemm_split", "fit_to_convergence_planted", "seconds"),
        "f32_gemm_split_fit_to_convergence_planted_iterations": _pick(cfg, "f32_gemm_split", "fit_to_convergence_planted", "iterations"),
        "f32_gemm_mfma_value": _pick(cfg, "f32_gemm_mfma", "fit_iterations_per_sec"),
        "merged_pass_roofline_frac": _pick(out, "roofline", "merged_pass", "frac"),
        "xbt_pass_roofline_frac": _pick(out, "roofline", "frac_by_site", "gemm_nt"),
        "xty_pass_roofline_frac": _pick(out, "roofline", "frac_by_site", "gemm_tn"),
        "fit_to_convergence_planted_seconds": _pick(cfg, "fit_to_convergence_planted", "seconds"),
        "fit_to_convergence_planted_iterations": _pick(cfg, "fit_to_convergence_planted", "iterations"),
        "weak_scaling_vs_same_shard": cfg.get("weak_scaling_vs_same_shard"),
        "single_gpu_same_shard_value": _pick(cfg, "single_gpu_same_shard", "iterations_per_sec_slowest_rank"),
    }
    for name in ("c2", "c2_weak", "c4shard", "c4_unsharded_one_gpu"):
        b = cfg.get(name)
        if isinstance(b, dict) and "value" in b:
            riders[name + "_value"] = b.get("value")
            riders[name + "_line_search"] = b.get("line_search")
            riders[name + "_ms_per_step"] = b.get("ms_per_step")
            riders[name + "_roofline_frac"] = _pick(b, "roofline", "frac")
            riders[name + "_roofline_step_frac"] = _pick(b, "roofline", "step_frac")
            riders[name + "_roofline_frac_min_site"] = _pick(b, "roofline", "frac_min_site")
            riders[name + "_xbt_pass_roofline_frac"] = _pick(b, "roofline", "frac_by_site", "gemm_nt")
            riders[name + "_xty_pass_roofline_frac"] = _pick(b, "roofline", "frac_by_site", "gemm_tn")
            riders[name + "_roofline_bound"] = _pick(b, "roofline", "bound")
            riders[name + "_reference_shaped_value"] = _pick(b, "reference_shaped", "fit_iterations_per_sec")
            riders[name + "_f32_gemm_split_value"] = _pick(b, "f32_gemm_split", "fit_iterations_per_sec")
            riders[name + "_cpu_baseline_value"] = _pick(b, "cpu_baseline", "value")
            riders[name + "_fit_to_convergence_seconds"] = _pick(b, "fit_to_convergence", "seconds")
            riders[name + "_cpu_fit_to_convergence_seconds"] = _pick(b, "cpu_fit_to_convergence", "seconds")
            riders[name + "_exchange_ms_per_iteration"] = _pick(b, "exchange_profile", "exchange_ms_per_iteration", "total")
            riders[name + "_allreduces_per_iteration"] = _pick(b, "exchange_profile", "allreduces_per_iteration")
    # BASELINE.json configs[0] (big5, 2000 x 50): whole fits, device and oracle wall clock - the latency-bound end of the path
    for tag in ("f32", "f64"):
        riders["c1_fit_seconds" + ("" if tag == "f32" else "_f64")] = _pick(cfg, "c1", tag, "fit_seconds")
        riders["c1_cpu_fit_seconds" + ("" if tag == "f32" else "_f64")] = _pick(cfg, "c1", tag, "cpu_fit_seconds")
    riders["c1_iterations"] = _pick(cfg, "c1", "f32", "iterations")
    riders["c1_cpu_iterations"] = _pick(cfg, "c1", "f32", "cpu_iterations")
    c.update({k: v for k, v in riders.items() if v is not None})
    c["detail"] = detail_path
    rl = out.get("roofline")
    if rl:
        it = rl.get("iteration") or {}
        rl = dict({k: rl.get(k) for k in ROOFLINE_LINE_KEYS},
                  iteration={k: it.get(k) for k in ("x_passes", "achieved_TFLOPs", "achieved_GBps",
                                                    "fraction_of_step_inside_the_x_passes")})
    cb = out.get("cpu_baseline")
    if cb:
        cb = {k: cb.get(k) for k in ("value", "unit", "cores", "kind", "sample", "trials_per_iteration", "x_passes_per_iteration",
                                     "n_variables_timed", "scaled_linearly_in_n_variables_by") if cb.get(k) is not None}
        cb["sample"] = cb["sample"].split(";")[0][:300]          # the first sentence; the rest is in the detail record
    line = {k: out[k] for k in ("metric", "value", "unit", "n_gpus", "steps", "warmup", "ms_per_step", "higher_is_better",
                                "scaling", "vs_baseline", "dtype", "data")}
    line.update(config=c, roofline=rl, cpu_baseline=cb)
    if out.get("series"):
        line["series"] = out["series"]
    return _r(line)


def emit(out, args, real_stdout):
    """Full record -> side file + one BENCH_DETAIL line on stderr; compact record -> the one stdout line."""
    rel = args.detail_out or os.path.join("gpurun_out", "bench_detail.json" if out["n_gpus"] == 1
                                          else "bench_detail_gpus%d.json" % out["n_gpus"])
    path = rel if os.path.isabs(rel) else os.path.join(ROOT, rel)
    full = json.dumps(out)
    try:
        os.makedirs(os.path.dirname(path), exist_ok=True)
        with open(path, "w") as f:
            f.write(full + "\n")
    except OSError as e:
        sys.stderr.write("bench.py: could not write %s: %s\n" % (path, e))
        rel = None
    sys.stderr.write("BENCH_DETAIL " + full + "\n")
    sys.stderr.flush()
    line = compact_line(out, rel)
    text = json.dumps(line, separators=(",", ":"))
    # (the rank launcher adds `exchange_attempts` to the line it relays: LCX_BENCH_LINE_RESERVE bytes of the budget are its)
    budget = args.max_line_bytes - int(os.environ.get("LCX_BENCH_LINE_RESERVE", "0") or 0)
    # a budget, not a hope: shed riders, then the sample sentence, until the line fits
    droppable = [k for k in list(line["config"]) if k.endswith(("_cpu_baseline_value", "_line_search", "_roofline_bound", "_ms_per_step",
                                                                "_pass_roofline_frac", "_f64", "_cpu_iterations", "_allreduces_per_iteration"))]
    while len(text) > budget and droppable:
        line["config"].pop(droppable.pop())
        text = json.dumps(line, separators=(",", ":"))
    if len(text) > budget and line.get("cpu_baseline"):
        line["cpu_baseline"]["sample"] = line["cpu_baseline"]["sample"][:120]
        text = json.dumps(line, separators=(",", ":"))
    os.write(real_stdout, (text + "\n").encode())


def series_of(out, head, world):
    """The weak-scaling series, readable from the lines alone (top level, every line): which workload the series is measured on,
    one GPU's rate inside THIS job, the same shard on one GPU without exchange steps, and their ratio.  The N = 1 line headlines
    configs[2] (`value`), the N > 1 lines headline N x the configs[3] shard, so value(N) / (N x value(1)) is NOT an efficiency -
    `series.per_gpu_value / series.n1_value_same_workload` is:
        N = 1, default job:   from the nested config.c4shard block (the workload the --gpus N lines headline); efficiency 1.0
        N = 1, one workload:  that workload is its own one-GPU point
        N > 1:                per_gpu_value = value / N; n1_value_same_workload = the same shard measured on every GPU alone in this
                              job (config.single_gpu_same_shard, slowest rank); null with --no-extras"""
    cfg = out["config"]
    if world == 1:
        c4 = cfg.get("c4shard")
        if isinstance(c4, dict) and c4.get("value"):
            wl, per_gpu = "c4shard", c4["value"]
        else:
            wl, per_gpu = head, out["value"]
        n1, eff = per_gpu, 1.0
    else:
        wl, per_gpu = head, out["value"] / world
        n1 = _pick(cfg, "single_gpu_same_shard", "iterations_per_sec_slowest_rank")
        eff = (per_gpu / n1) if n1 else None
    n, v_per, m, tag = WORKLOADS[wl]
    return {"workload": wl, "shard": "%d x %d x %d %s per GPU" % (n, v_per, m, tag), "n_gpus": world, "per_gpu_value": per_gpu,
            "n1_value_same_workload": n1, "efficiency": eff, "unit": "fit iterations/s per GPU"}
