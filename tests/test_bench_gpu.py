"""bench.py contract checks on the GPU box: exactly ONE line on stdout and it is the JSON record - also when RCCL is
live (it prints a version banner to the C-level stdout) and when several ranks run (two ranks sharing GPU 0 through
gloo exercise the multi-rank code of bench.py; RCCL itself refuses two ranks per device)."""
import json
import os
import subprocess
import sys

import pytest

from tests.test_distributed_cpu import ROOT, free_port

pytestmark = pytest.mark.gpu

FAST = ["--steps", "7", "--warmup", "7", "--no-also-linear", "--cpu-iters-per-stage", "0", "--no-convergence"]


def _one_json_line(stdout):
    lines = [ln for ln in stdout.decode(errors="replace").splitlines() if ln.strip()]
    assert len(lines) == 1, lines
    return json.loads(lines[0])


def test_single_rank_rccl_stdout_is_one_json_line():
    p = subprocess.run([sys.executable, os.path.join(ROOT, "bench.py"), "--force-exchange"] + FAST, cwd=ROOT,
                       stdout=subprocess.PIPE, stderr=subprocess.PIPE, timeout=300)
    assert p.returncode == 0, p.stderr.decode(errors="replace")[-3000:]
    d = _one_json_line(p.stdout)
    assert d["n_gpus"] == 1 and d["config"]["force_exchange"] is True
    assert d["roofline"]["kernel"].startswith("lcx::gemm_") and 0.0 < d["roofline"]["frac"] < 1.0
    assert d["value"] > 0 and d["steps"] == 7


def test_two_ranks_one_gpu_stdout_is_one_json_line():
    env = dict(os.environ, LCX_BENCH_DEVICE="0", LCX_BENCH_BACKEND="gloo", OMP_NUM_THREADS="2", OPENBLAS_NUM_THREADS="2")
    cmd = [sys.executable, "-m", "torch.distributed.run", "--nnodes=1", "--nproc-per-node", "2", "--master-addr", "127.0.0.1",
           "--master-port", str(free_port()), os.path.join(ROOT, "bench.py"), "--gpus", "2"] + FAST
    p = None
    for attempt in range(2):                      # see tests/test_distributed_gpu.py: two gloo ranks on one GPU can stall
        try:
            p = subprocess.run(cmd, cwd=ROOT, env=env, stdout=subprocess.PIPE, stderr=subprocess.PIPE, timeout=180)
            break
        except subprocess.TimeoutExpired as e:
            sys.stderr.write("bench.py with two gloo ranks did not finish in 180 s:\n%s\n" % (e.stderr or b"").decode(errors="replace")[-3000:])
            p = None
    assert p is not None, "bench.py with two gloo ranks stalled twice"
    assert p.returncode == 0, p.stderr.decode(errors="replace")[-3000:]
    d = _one_json_line(p.stdout)
    assert d["n_gpus"] == 2 and d["scaling"] == "weak"
    assert d["config"]["n_variables_total"] == 2 * d["config"]["n_variables_per_gpu"]
    assert d["cpu_baseline"] is None
