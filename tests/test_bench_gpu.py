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
    # started the way the driver starts it: torch.distributed.run launches the ranks, each supervises its worker (round 6: the first 8-rank
    # rehearsal in this form hung - the ranks that wait out the CPU baseline looked for rank 0's signal under their own parent's pid)
    cmd = [sys.executable, "-m", "torch.distributed.run", "--nnodes=1", "--nproc-per-node", "2", "--master-addr", "127.0.0.1",
           "--master-port", str(free_port()), os.path.join(ROOT, "bench.py"), "--gpus", "2", "--steps", "3", "--warmup", "2", "--repeats", "1",
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
    # where the time outside the X passes goes: the all-reduces by site (HIP events on the engine's stream), their count, every rank's own
    # ms_per_step, and the same shard without exchange steps - on the line and in the record
    c = line["config"]
    assert c["exchange"] == "hook" and c["allreduces_per_iteration"] >= 3.0 and c["compute_only_ms_per_step"] > 0
    xs = c["exchange_ms_per_iteration"]
    assert set(xs) >= {"y", "direction", "scalars", "total"} and all(v >= 0 for v in xs.values())
    assert xs["total"] == pytest.approx(sum(v for k, v in xs.items() if k != "total"), rel=1e-3)
    lo, med, hi = c["ms_per_step_rank_min_median_max"]
    assert 0 < lo <= med <= hi <= line["ms_per_step"] * 1.05
    assert c["c2_weak_exchange_ms_per_iteration"] > 0
    xp = d["config"]["exchange_profile"]
    assert len(xp["ms_per_step_by_rank"]) == 2 and xp["slowest_rank"] in (0, 1)
    assert xp["allreduces_per_iteration_by_site"]["scalars"] >= 1.0 and xp["avg_us_by_site"]["y"] > 0
    # 1 + 2T collectives per iteration: Y + scalars per evaluated trial, one for the direction
    t = d["config"]["line_search_trials_per_iteration"]
    assert xp["allreduces_per_iteration"] == pytest.approx(1 + 2 * t, abs=0.35)
    assert d["cpu_baseline"]["other_ranks_while_timed"].startswith("parked") and d["cpu_baseline"]["host"]["usable_cores"] >= 1
    assert [a["reason"] for a in line["exchange_attempts"]] == ["ok"]


def test_default_job_live_every_nested_block(tmp_path):
    """The driver's bench command - `python bench.py` with every nested block, CPU legs included - run LIVE, on shrunk workloads
    (LCX_BENCH_SHRINK: n_variables / 25, so the whole job takes about a minute; the kernels the real shapes select are covered by
    test_c3_headline_alone and the parity tests).  What the committed-record tests on the CPU side can only re-parse is produced here:
    the c2 block with both convergence legs, the c1 block (BASELINE configs[0], device and oracle wall clock), the config-4 shard
    block, the opt-in line searches and arithmetic, both CPU baselines, the whole-step roofline fraction, the series."""
    detail = str(tmp_path / "detail.json")
    env = dict(os.environ, LCX_BENCH_SHRINK="25", LCX_BENCH_GENERATE_ABOVE=str(10 ** 8))
    t0 = time.time()
    p = subprocess.run([sys.executable, os.path.join(ROOT, "bench.py"), "--steps", "3", "--warmup", "2", "--cpu-seconds", "2",
                        "--cpu-iters-per-stage", "2", "--convergence-max-iter", "3", "--convergence-planted-max-iter", "40",
                        "--detail-out", detail], cwd=ROOT, env=env, stdout=subprocess.PIPE, stderr=subprocess.PIPE, timeout=600)
    wall = time.time() - t0
    assert p.returncode == 0, p.stderr.decode(errors="replace")[-3000:]
    line = _one_json_line(p.stdout)
    c, rl = line["config"], line["roofline"]
    assert c["workload"].startswith("c3:") and "SHRUNK" in c["workload"] and line["n_gpus"] == 1 and line["value"] > 0
    # roofline: the dominant kernel, the slowest pass site and the whole step against the same roof
    assert 0 < rl["step_frac"] < 1 and 0 < rl["frac_min_site"] <= rl["frac"] < 1 and rl["step_frac"] <= rl["frac"] * 1.2
    assert line["cpu_baseline"]["value"] > 0 and line["cpu_baseline"]["kind"] == "port" and line["cpu_baseline"]["cores"] >= 1
    # riders of every nested block
    for k in ("c2_value", "c2_roofline_frac", "c2_roofline_step_frac", "c2_cpu_baseline_value", "c2_fit_to_convergence_seconds",
              "c2_cpu_fit_to_convergence_seconds", "c1_fit_seconds", "c1_cpu_fit_seconds", "c4shard_value", "c4shard_roofline_frac",
              "c4shard_roofline_step_frac", "c4shard_roofline_frac_min_site", "exact_y_value", "linear_value",
              "fit_to_convergence_planted_seconds"):
        assert c.get(k, 0) > 0, k
    sr = _series_ok(line)
    assert sr["workload"] == "c4shard" and sr["efficiency"] == 1.0 and sr["per_gpu_value"] == pytest.approx(c["c4shard_value"], rel=1e-4)
    d = _detail(detail, p.stderr)["config"]
    c2, c1, c4 = d["c2"], d["c1"], d["c4shard"]
    # BASELINE.md section 3's convergence leg: the same X, the same tolerance, device and oracle side by side - and they agree
    dev, cpu = c2["fit_to_convergence"], c2["cpu_fit_to_convergence"]
    assert dev["iterations"] == cpu["iterations"] and dev["trials"] == cpu["trials"] and abs(dev["TC"] - cpu["TC"]) < 1e-6 * max(1.0, abs(cpu["TC"]))
    assert dev["seconds"] > 0 and cpu["seconds"] > 0 and cpu["cores"] >= 1 and cpu["host"]["usable_cores"] >= 1
    assert c2["cpu_baseline"]["host"]["loadavg_1_5_15"] is not None
    for tag in ("f32", "f64"):
        b = c1[tag]
        assert b["fit_seconds"] > 0 and b["cpu_fit_seconds"] > 0 and b["same_clusters"] is True
    assert c1["f64"]["iterations"] == c1["f64"]["cpu_iterations"] == 261 and c1["f64"]["trials"] == c1["f64"]["cpu_trials"]
    assert abs(c1["f32"]["iterations"] - c1["f32"]["cpu_iterations"]) <= 0.06 * c1["f32"]["cpu_iterations"]
    assert c4["cpu_baseline"]["value"] > 0 and c4["roofline"]["step_frac"] > 0 and "f32_gemm_split" in c4
    for blk in (d, c4):
        assert blk["later_trials_by_linearity"]["fit_iterations_per_sec"] > 0 and blk["linear_trial_mode"]["fit_iterations_per_sec"] > 0
    assert d["fit_to_convergence_planted"]["iterations"] > 0 and d["get_covariance_c5_standin"]["n_variables"] == 20000
    assert wall < 240, wall


def test_a_hung_first_contact_ends_in_a_line_from_the_next_transport(tmp_path):
    """The first multi-GPU job must not be lost to a bad first contact: the driver's `torch.distributed.run ... bench.py --gpus 2` whose
    rank 1 never arrives in ncclCommInitRank (LCX_TEST_HANG_COMM_INIT=1; two gloo ranks on this one GPU, the RCCL negotiation forced) -
    the watchdog inside the workers ends attempt 1 (exit code 3, stacks on stderr), the rank supervisors start FRESH workers on
    LCX_EXCHANGE=hook, and the job ends in ONE JSON line that records both attempts.  (`python bench.py --gpus 2`, where bench.py is
    the launcher itself, takes the same ladder in spawn_ranks: CPU tests + test_gpus_2_without_a_launcher_starts_two_ranks.)  (A rank set that nothing ends from the inside is killed at the launcher's
    wall-clock budget: tests/test_host_logic_cpu.py::test_rank_launcher_kills_a_hung_rank_set_and_its_detached_children.)"""
    env = dict(_two_rank_env(), LCX_TEST_FORCE_RCCL_NEGOTIATION="1", LCX_TEST_HANG_COMM_INIT="1", LCX_FIRST_CONTACT_TIMEOUT_S="6",
               LCX_BENCH_ATTEMPT_S="100")
    # launched the way the driver launches a multi-GPU run: the ranks come from torch.distributed.run, each is a supervisor of its worker
    cmd = [sys.executable, "-m", "torch.distributed.run", "--nnodes=1", "--nproc-per-node", "2", "--master-addr", "127.0.0.1",
           "--master-port", str(free_port()), os.path.join(ROOT, "bench.py"), "--gpus", "2"] + FAST
    t0 = time.time()
    p = subprocess.run(cmd, cwd=ROOT, env=env, stdout=subprocess.PIPE, stderr=subprocess.PIPE, timeout=300)
    wall = time.time() - t0
    err = p.stderr.decode(errors="replace")
    assert p.returncode == 0, err[-3000:]
    d = _one_json_line(p.stdout)
    att = d["exchange_attempts"]
    assert [a["transport"] for a in att] == ["caller (LCX_BENCH_BACKEND=gloo)", "hook"], att
    assert att[0]["rc"] not in (0, None) and "worker exited with rc 3" in att[0]["reason"] and att[0]["seconds"] < 60
    assert att[1]["rc"] == 0 and att[1]["reason"] == "ok"
    assert d["n_gpus"] == 2 and d["value"] > 0 and d["config"]["exchange"] == "hook"
    assert "first contact with the exchange transport did not finish" in err and "ncclCommInitRank" in err and "File " in err
    assert wall < 120, wall


def test_the_whole_ladder_against_real_rccl_failures():
    """Two ranks on this ONE GPU with the default (RCCL) backend, launched as the driver launches them: torch's RCCL group refuses two ranks
    per device ("Duplicate GPU detected"), so the engine, hook and torch rungs each die in their first collective with RCCL's own error -
    real failures of the real transport, not injected ones - and the last rung (hook over gloo, fresh workers on a rendezvous of their own)
    delivers the line with all four attempts on it."""
    env = dict(os.environ, LCX_BENCH_DEVICE="0", LCX_FIRST_CONTACT_TIMEOUT_S="25", LCX_BENCH_ATTEMPT_S="120", OMP_NUM_THREADS="2",
               OPENBLAS_NUM_THREADS="2", LCX_WAIT_TIMEOUT_MS="60000", **SMALL)
    for k in ("LCX_BENCH_BACKEND", "LCX_EXCHANGE", "LCX_BENCH_LADDER"):
        env.pop(k, None)
    cmd = [sys.executable, "-m", "torch.distributed.run", "--nnodes=1", "--nproc-per-node", "2", "--master-addr", "127.0.0.1",
           "--master-port", str(free_port()), os.path.join(ROOT, "bench.py"), "--gpus", "2"] + FAST
    t0 = time.time()
    p = subprocess.run(cmd, cwd=ROOT, env=env, stdout=subprocess.PIPE, stderr=subprocess.PIPE, timeout=400)
    err = p.stderr.decode(errors="replace")
    assert p.returncode == 0, err[-3000:]
    d = _one_json_line(p.stdout)
    att = d["exchange_attempts"]
    assert [a["transport"] for a in att] == ["engine", "hook", "torch", "gloo"], att
    assert all(a["rc"] not in (0, None) and a["seconds"] < 60 for a in att[:3]) and att[3]["rc"] == 0 and att[3]["reason"] == "ok"
    assert d["n_gpus"] == 2 and d["value"] > 0 and d["config"]["exchange"] == "hook"
    assert "Duplicate GPU detected" in err
    assert time.time() - t0 < 120
