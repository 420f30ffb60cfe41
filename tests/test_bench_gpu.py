"""bench.py contract checks on the GPU box: exactly ONE line on stdout and it is the JSON record - also when RCCL is
live (it prints a version banner to the C-level stdout) and when several ranks run (two ranks sharing GPU 0 through
gloo exercise the multi-rank code of bench.py; RCCL itself refuses two ranks per device).  `--gpus 2` without a
launcher must start two ranks itself and report n_gpus == 2."""
import json
import os
import subprocess
import sys
import time

import pytest

from tests.test_distributed_cpu import ROOT, free_port

pytestmark = pytest.mark.gpu

FAST = ["--steps", "3", "--warmup", "2", "--no-extras", "--repeats", "1"]
SMALL = {"LCX_BENCH_HEAD": "tiny"}          # a small headline workload instead of BASELINE configs[2] / [3]


LINE_BUDGET = 4096          # bytes of the stdout line (the driver keeps a few KB of stdout; round 3's 21 KB line was not parsed)


def _one_json_line(stdout):
    lines = [ln for ln in stdout.decode(errors="replace").splitlines() if ln.strip()]
    assert len(lines) == 1, lines
    assert len(lines[0]) <= LINE_BUDGET, len(lines[0])
    d = json.loads(lines[0])
    for k in ("metric", "value", "unit", "n_gpus", "steps", "warmup", "ms_per_step", "higher_is_better", "scaling", "vs_baseline",
              "dtype", "data", "config", "roofline", "cpu_baseline"):
        assert k in d, k
    assert isinstance(d["config"]["workload"], str)
    return d


def _detail(path, stderr=None):
    """the full record: the side file, and the same record as the BENCH_DETAIL line on stderr"""
    with open(path) as f:
        full = json.load(f)
    if stderr is not None:
        marked = [ln for ln in stderr.decode(errors="replace").splitlines() if ln.startswith("BENCH_DETAIL ")]
        assert len(marked) == 1 and json.loads(marked[0][len("BENCH_DETAIL "):]) == full
    return full


def test_single_rank_rccl_stdout_is_one_json_line():
    p = subprocess.run([sys.executable, os.path.join(ROOT, "bench.py"), "--force-exchange", "--workload", "c2"] + FAST, cwd=ROOT,
                       stdout=subprocess.PIPE, stderr=subprocess.PIPE, timeout=300)
    assert p.returncode == 0, p.stderr.decode(errors="replace")[-3000:]
    d = _one_json_line(p.stdout)
    assert d["n_gpus"] == 1 and d["config"]["force_exchange"] is True
    assert d["roofline"]["kernel"].startswith("lcx::gemm_") and 0.0 < d["roofline"]["frac"] < 1.0
    assert d["roofline"]["bound"] == "hbm" and d["dtype"] == "f64"
    assert d["value"] > 0 and d["steps"] == 3
    assert d["config"]["windows"]["timed_iterations"] == 21
    w = _detail(os.path.join(ROOT, d["config"]["detail"]), p.stderr)["config"]["windows"]
    assert w["stages"] == 7 and w["steps_per_window"] == 3 and w["timed_iterations"] == 21


def _series_ok(line):
    """the weak-scaling series is readable from the line alone (top level): per_gpu_value / n1_value_same_workload = efficiency"""
    sr = line["series"]
    assert set(("workload", "n_gpus", "per_gpu_value", "n1_value_same_workload", "efficiency")) <= set(sr)
    assert sr["n_gpus"] == line["n_gpus"] and sr["per_gpu_value"] > 0
    if sr["n1_value_same_workload"] is not None:
        assert sr["efficiency"] == pytest.approx(sr["per_gpu_value"] / sr["n1_value_same_workload"], rel=1e-4)
    if line["n_gpus"] > 1:
        assert sr["per_gpu_value"] == pytest.approx(line["value"] / line["n_gpus"], rel=1e-4)
    return sr


def test_c3_headline_alone(tmp_path):
    """BASELINE configs[2] as the driver's command measures it, without the nested blocks (`--no-extras --repeats 1`): the MFMA-bound
    headline, its roofline object from live HIP events, the line inside the driver's stdout budget.  The nested blocks of the
    default job (c2, the config-4 shard, the opt-in line searches, the convergence runs, CPU baselines: ~2 minutes, what the
    driver's own bench step runs at round end) are checked on the CPU side against the committed record of such a run
    (tests/test_host_logic_cpu.py::test_bench_default_record_*)."""
    detail = str(tmp_path / "detail.json")
    t0 = time.time()
    p = subprocess.run([sys.executable, os.path.join(ROOT, "bench.py"), "--gpus", "1", "--steps", "20", "--warmup", "5", "--workload", "c3",
                        "--no-extras", "--repeats", "1", "--detail-out", detail], cwd=ROOT, stdout=subprocess.PIPE, stderr=subprocess.PIPE,
                       timeout=600)
    wall = time.time() - t0
    assert p.returncode == 0, p.stderr.decode(errors="replace")[-3000:]
    line = _one_json_line(p.stdout)
    assert line["config"]["workload"].startswith("c3:") and line["dtype"] == "f32" and line["n_gpus"] == 1
    rl = line["roofline"]
    assert rl["bound"] == "mfma" and rl["unit"] == "TFLOP/s" and 0.3 < rl["frac"] < 1.0
    assert rl["kernel"].startswith("lcx::gemm_c") and rl["avg_launch_us"] > 0 and rl["launches"] > 0
    assert abs(rl["achieved"] / rl["peak"] - rl["frac"]) < 1e-4
    c = line["config"]
    assert c["detail"] == detail and c["windows"]["walks_timed"] == 1 and c["windows"]["timed_iterations"] == 140
    assert abs(line["value"] * line["ms_per_step"] / 1e3 - 1.0) < 1e-3
    assert wall < 200, wall
    sr = _series_ok(line)
    assert sr["workload"] == "c3" and sr["efficiency"] == 1.0 and sr["per_gpu_value"] == pytest.approx(line["value"], rel=1e-5)
    d = _detail(detail, p.stderr)
    assert d["value"] == pytest.approx(line["value"], rel=1e-5) and d["roofline"]["frac"] == pytest.approx(rl["frac"], rel=1e-5)
    assert 3.0 < d["config"]["x_passes_per_iteration"] < 3.6 and d["config"]["bytes_resident"]["x_layout"].startswith("panel")


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
    sr = _series_ok(d)            # --no-extras: no same-shard pre-run, the one-GPU point is absent and says so
    assert sr["workload"] == "tiny" and sr["n1_value_same_workload"] is None and sr["efficiency"] is None


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


def test_two_ranks_full_line_has_the_same_shard_reference_and_the_c2_block(tmp_path):
    """The line the driver's multi-GPU run produces (extras on): the headline shard measured on every GPU alone first
    (`config.single_gpu_same_shard`: the one-GPU point of this workload's weak-scaling series), then sharded; the nested
    config-2 block with 5 000 variables per GPU."""
    detail = str(tmp_path / "detail2.json")
    cmd = [sys.executable, os.path.join(ROOT, "bench.py"), "--gpus", "2", "--steps", "3", "--warmup", "2", "--repeats", "1",
           "--detail-out", detail]
    p = subprocess.run(cmd, cwd=ROOT, env=_two_rank_env(), stdout=subprocess.PIPE, stderr=subprocess.PIPE, timeout=600)
    assert p.returncode == 0, p.stderr.decode(errors="replace")[-3000:]
    line = _one_json_line(p.stdout)
    assert line["n_gpus"] == 2 and line["cpu_baseline"]["value"] > 0 and line["roofline"]["frac"] > 0
    assert line["config"]["single_gpu_same_shard_value"] > 0 and line["config"]["weak_scaling_vs_same_shard"] > 0
    assert line["config"]["c2_weak_value"] > 0
    # the series from the line alone: per-GPU rate of this job / the same shard on one GPU alone in this job
    sr = _series_ok(line)
    assert sr["workload"] == "tiny" and sr["n1_value_same_workload"] == pytest.approx(line["config"]["single_gpu_same_shard_value"], rel=1e-4)
    assert sr["efficiency"] == pytest.approx(line["config"]["weak_scaling_vs_same_shard"], rel=1e-4)
    d = _detail(detail)
    assert d["n_gpus"] == 2
    # the multi-GPU line carries a CPU baseline too (rank 0, one shard's iteration) and a roofline
    assert d["cpu_baseline"]["value"] > 0 and d["cpu_baseline"]["kind"] == "port" and d["cpu_baseline"]["cores"] >= 1
    assert d["roofline"] is not None and d["roofline"]["frac"] > 0
    ref = d["config"]["single_gpu_same_shard"]
    assert ref["iterations_per_sec_slowest_rank"] > 0 and d["config"]["weak_scaling_vs_same_shard"] > 0
    c2 = d["config"]["c2_weak"]
    assert c2["n_variables_total"] == 10000 and c2["n_variables_per_gpu"] == 5000 and c2["roofline"]["bound"] == "hbm"
    assert "linear_trial_mode" not in c2          # (several ranks: the other line searches of c2 are skipped)
