"""bench.py contract checks on the GPU box: exactly ONE line on stdout and it is the JSON record - also when RCCL is
live (it prints a version banner to the C-level stdout) and when several ranks run (two ranks sharing GPU 0 through
gloo exercise the multi-rank code of bench.py; RCCL itself refuses two ranks per device).  `--gpus 2` without a
launcher must start two ranks itself and report n_gpus == 2."""
import json
import os
import subprocess
import sys

import pytest

from tests.test_distributed_cpu import ROOT, free_port

pytestmark = pytest.mark.gpu

FAST = ["--steps", "3", "--warmup", "2", "--no-extras", "--repeats", "1"]
SMALL = {"LCX_BENCH_HEAD": "tiny"}          # a small headline workload instead of BASELINE configs[2] / [3]


def _one_json_line(stdout):
    lines = [ln for ln in stdout.decode(errors="replace").splitlines() if ln.strip()]
    assert len(lines) == 1, lines
    return json.loads(lines[0])


def test_single_rank_rccl_stdout_is_one_json_line():
    p = subprocess.run([sys.executable, os.path.join(ROOT, "bench.py"), "--force-exchange", "--workload", "c2"] + FAST, cwd=ROOT,
                       stdout=subprocess.PIPE, stderr=subprocess.PIPE, timeout=300)
    assert p.returncode == 0, p.stderr.decode(errors="replace")[-3000:]
    d = _one_json_line(p.stdout)
    assert d["n_gpus"] == 1 and d["config"]["force_exchange"] is True
    assert d["roofline"]["kernel"].startswith("lcx::gemm_") and 0.0 < d["roofline"]["frac"] < 1.0
    assert d["roofline"]["bound"] == "hbm" and d["dtype"] == "f64"
    assert d["value"] > 0 and d["steps"] == 3
    w = d["config"]["windows"]
    assert w["stages"] == 7 and w["steps_per_window"] == 3 and w["timed_iterations"] == 21


def test_default_line_is_the_c3_line_with_the_c2_block():
    """The driver's command: the headline must be BASELINE configs[2] (MFMA-bound), c2 rides along as a nested block, the
    CPU baselines are real, and the figure must not depend on the driver's --steps 20 --warmup 5."""
    p = subprocess.run([sys.executable, os.path.join(ROOT, "bench.py"), "--gpus", "1", "--steps", "20", "--warmup", "5",
                        "--cpu-seconds", "5"], cwd=ROOT, stdout=subprocess.PIPE, stderr=subprocess.PIPE, timeout=1500)
    assert p.returncode == 0, p.stderr.decode(errors="replace")[-3000:]
    d = _one_json_line(p.stdout)
    assert d["config"]["workload"].startswith("c3:") and d["dtype"] == "f32" and d["n_gpus"] == 1
    assert d["roofline"]["bound"] == "mfma" and d["roofline"]["unit"] == "TFLOP/s" and 0.3 < d["roofline"]["frac"] < 1.0
    assert d["roofline"]["kernel"].startswith("lcx::gemm_ct")
    assert d["config"]["windows"]["timed_seconds"] >= 0.5
    assert d["cpu_baseline"]["value"] > 0 and d["cpu_baseline"]["kind"] == "port" and d["cpu_baseline"]["cores"] >= 1
    c2 = d["config"]["c2"]
    assert c2["roofline"]["bound"] == "hbm" and c2["windows"]["timed_seconds"] >= 0.5
    assert c2["cpu_baseline"]["value"] > 0
    lo, med, hi = c2["windows"]["ms_per_step_walk_min_median_max"]
    assert med / lo < 1.1 and hi / lo < 2.0, (lo, med, hi)      # (one of ~10 walks may catch a host hiccup: the figure is the per-stage median)
    assert c2["get_covariance"]["n_variables"] == 5000 and c2["get_covariance"]["seconds"] > 0
    assert d["config"]["get_covariance_c5_standin"]["n_variables"] == 20000
    # the headline is timed over >= 3 walks of the schedule and says how far they are apart
    w = d["config"]["windows"]
    assert w["walks_timed"] >= 3
    lo, med, hi = w["ms_per_step_walk_min_median_max"]
    assert lo <= med <= hi and med / lo < 1.05 and hi / lo < 1.3, (lo, med, hi)
    # the one-GPU point of the weak-scaling series that --gpus N headlines
    c4 = d["config"]["c4shard"]
    assert c4["value"] > 0 and c4["n_hidden"] == 128 and c4["n_variables_per_gpu"] == 125000
    assert c4["roofline"]["bound"] == "mfma" and 0.3 < c4["roofline"]["frac"] < 1.0
    assert c4["cpu_baseline"]["value"] > 0 and c4["cpu_baseline"]["n_variables_timed"] <= 100000
    # configs[3] unsharded on this one GPU (single resident copy, gemm_cr)
    c6 = d["config"]["c4_unsharded_one_gpu"]
    assert c6["value"] > 1.0 and c6["n_variables_total"] == 1000000 and "gemm_cr_kernel" in c6["kernels"]["gemm_nt"]
    assert c6["bytes_resident"]["total"] < 250e9
    # linear trial mode, reported beside the reference-shaped figure at the sizes where a trial costs two long passes
    for blk in (d["config"], c4):
        mid = blk["later_trials_by_linearity"]               # line_search="exact-y": between the reference-shaped and the linear mode
        assert mid["fit_iterations_per_sec"] > blk["fit_iterations_per_sec"] if "fit_iterations_per_sec" in blk else True
        assert blk["linear_trial_mode"]["x_passes_per_iteration"] < mid["x_passes_per_iteration"] < blk["x_passes_per_iteration"]
        lin = blk["linear_trial_mode"]
        assert lin["fit_iterations_per_sec"] > 0 and 1.9 < lin["x_passes_per_iteration"] < 2.6
        assert lin["roofline"]["bound"] == "mfma"
    # a convergence measurement that converges: planted data, every annealing stage below tol before the cap
    cv = d["config"]["fit_to_convergence_planted"]
    assert cv["stages_converged_before_the_cap"] == 7 and cv["seconds"] > 0
    assert cv["cluster_purity_vs_planted_groups"] > 0.99
    assert d["config"]["fit_to_convergence"]["capped"] is True


def _two_rank_env():
    return dict(os.environ, LCX_BENCH_DEVICE="0", LCX_BENCH_BACKEND="gloo", OMP_NUM_THREADS="2", OPENBLAS_NUM_THREADS="2",
                LCX_WAIT_TIMEOUT_MS="60000", **SMALL)


def test_two_ranks_one_gpu_stdout_is_one_json_line():
    cmd = [sys.executable, "-m", "torch.distributed.run", "--nnodes=1", "--nproc-per-node", "2", "--master-addr", "127.0.0.1",
           "--master-port", str(free_port()), os.path.join(ROOT, "bench.py"), "--gpus", "2"] + FAST
    p = subprocess.run(cmd, cwd=ROOT, env=_two_rank_env(), stdout=subprocess.PIPE, stderr=subprocess.PIPE, timeout=240)
    assert p.returncode == 0, p.stderr.decode(errors="replace")[-3000:]
    d = _one_json_line(p.stdout)
    assert d["n_gpus"] == 2 and d["scaling"] == "weak"
    assert d["config"]["n_variables_total"] == 2 * d["config"]["n_variables_per_gpu"]
    assert d["cpu_baseline"] is None


def test_gpus_2_without_a_launcher_starts_two_ranks():
    """`python bench.py --gpus 2` (what the driver types): bench.py launches the ranks as children and relays rank 0's line."""
    p = subprocess.run([sys.executable, os.path.join(ROOT, "bench.py"), "--gpus", "2"] + FAST, cwd=ROOT, env=_two_rank_env(),
                       stdout=subprocess.PIPE, stderr=subprocess.PIPE, timeout=300)
    assert p.returncode == 0, p.stderr.decode(errors="replace")[-3000:]
    d = _one_json_line(p.stdout)
    assert d["n_gpus"] == 2 and d["value"] > 0
    assert d["config"]["n_variables_total"] == 2 * d["config"]["n_variables_per_gpu"]


def test_world_size_mismatch_is_an_error():
    env = dict(os.environ, WORLD_SIZE="1", RANK="0", LOCAL_RANK="0")
    p = subprocess.run([sys.executable, os.path.join(ROOT, "bench.py"), "--gpus", "2"] + FAST, cwd=ROOT, env=env,
                       stdout=subprocess.PIPE, stderr=subprocess.PIPE, timeout=120)
    assert p.returncode != 0 and not p.stdout.strip()


def test_two_ranks_full_line_has_the_same_shard_reference_and_the_c2_block():
    """The line the driver's multi-GPU run produces (extras on): the headline shard measured on every GPU alone first
    (`config.single_gpu_same_shard`: the one-GPU point of this workload's weak-scaling series), then sharded; the nested
    config-2 block with 5 000 variables per GPU."""
    cmd = [sys.executable, os.path.join(ROOT, "bench.py"), "--gpus", "2", "--steps", "3", "--warmup", "2", "--repeats", "1"]
    p = subprocess.run(cmd, cwd=ROOT, env=_two_rank_env(), stdout=subprocess.PIPE, stderr=subprocess.PIPE, timeout=600)
    assert p.returncode == 0, p.stderr.decode(errors="replace")[-3000:]
    d = _one_json_line(p.stdout)
    assert d["n_gpus"] == 2
    # the multi-GPU line carries a CPU baseline too (rank 0, one shard's iteration) and a roofline
    assert d["cpu_baseline"]["value"] > 0 and d["cpu_baseline"]["kind"] == "port" and d["cpu_baseline"]["cores"] >= 1
    assert d["roofline"] is not None and d["roofline"]["frac"] > 0
    ref = d["config"]["single_gpu_same_shard"]
    assert ref["iterations_per_sec_slowest_rank"] > 0 and d["config"]["weak_scaling_vs_same_shard"] > 0
    c2 = d["config"]["c2_weak"]
    assert c2["n_variables_total"] == 10000 and c2["n_variables_per_gpu"] == 5000 and c2["roofline"]["bound"] == "hbm"
    assert c2["linear_trial_mode"]["fit_iterations_per_sec"] > 0
