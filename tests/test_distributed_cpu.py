"""N>1 path on CPU: world_size-2 (and 3) gloo groups running the product's host driver with the
n_variables axis sharded, exchange steps as torch.distributed all-reduces.  The result must equal
the single-process oracle (same iteration count; float64 agreement to ~1e-10: only the summation
order of the all-reduced quantities changes)."""
import os
import socket
import subprocess
import sys

import numpy as np
import pytest

from oracle import corex_oracle as O

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))


def free_port():
    s = socket.socket()
    s.bind(("127.0.0.1", 0))
    p = s.getsockname()[1]
    s.close()
    return p


def check_covariance(got, ref, tol=1e-8):
    """get_covariance over the ranks (reference :443-455): the whole matrix (small models) and a block of rows that straddles a
    shard boundary against the single-process oracle."""
    cov_ref = ref.get_covariance()
    scale = float(np.max(np.abs(cov_ref)))
    r0, rows = int(got["cov_row0"]), got["cov_rows"]
    assert rows.shape[1] == cov_ref.shape[1] and np.max(np.abs(rows - cov_ref[r0:r0 + rows.shape[0]])) < tol * scale
    if got["cov"].ndim == 2:
        assert np.max(np.abs(got["cov"] - cov_ref)) < tol * scale
        assert np.allclose(np.diag(got["cov"]), np.asarray(ref.theta[1]) ** 2, rtol=1e-12, atol=0)     # fill_diagonal (:449 / :454)


def launch(world, out_dir, n, v, m, mode="exact", extra_env=None):
    port = free_port()
    procs = []
    for r in range(world):
        env = dict(os.environ, RANK=str(r), WORLD_SIZE=str(world), LOCAL_RANK=str(r), MASTER_ADDR="127.0.0.1",
                   MASTER_PORT=str(port), OMP_NUM_THREADS="2", OPENBLAS_NUM_THREADS="2", **(extra_env or {}))
        procs.append(subprocess.Popen([sys.executable, os.path.join(ROOT, "tests", "_dist_worker.py"), str(out_dir),
                                       str(n), str(v), str(m), mode], env=env, cwd=ROOT,
                                      stdout=subprocess.PIPE, stderr=subprocess.STDOUT))
    outs = []
    for p in procs:
        try:
            out, _ = p.communicate(timeout=240)
        except subprocess.TimeoutExpired:
            for q in procs:
                q.kill()
            raise
        outs.append(out.decode(errors="replace"))
    for p, o in zip(procs, outs):
        assert p.returncode == 0, o[-3000:]


@pytest.mark.parametrize("world,mode", [(2, "exact"), (3, "exact"), (2, "linear"), (8, "exact")])
def test_sharded_fit_matches_oracle(world, mode, tmp_path):
    n, v, m = 300, 203, 4               # 203 variables: uneven shards
    launch(world, tmp_path, n, v, m, mode)
    got = np.load(os.path.join(tmp_path, "dist_result.npz"))
    assert int(got["world"]) == world
    x, _ = O.gen_planted(n, v, m, seed=2)
    ref = O.fit_ns(x, m, seed=0, dtype=np.float64, keep_x=True)
    h, h_ref = got["history"], np.asarray(ref.history_tc)
    assert len(h) == len(h_ref)
    assert np.max(np.abs(h - h_ref) / np.maximum(1, np.abs(h_ref))) < 1e-9
    assert np.array_equal(got["clusters"], ref.clusters())
    assert got["ws"].shape == (m, v)
    assert np.max(np.abs(got["ws"] - ref.ws)) < 1e-8
    assert np.max(np.abs(got["transform"] - ref.transform(ref.x_tilde))) < 1e-8
    assert np.max(np.abs(got["rho"] - ref.moments["rho"])) < 1e-8
    assert np.max(np.abs(got["xz"] - ref.moments["X_i Z_j"])) < 1e-8
    assert np.max(np.abs(got["si"] - ref.moments["Si"])) < 1e-8
    assert np.max(np.abs(got["tcs"] - ref.moments["TCs"])) < 1e-8
    assert int(got["trials"]) == ref.n_trials
    check_covariance(got, ref)


def test_caller_chosen_uneven_shards(tmp_path):
    """Comm(bounds=...): the caller's column boundaries instead of the balanced split - three ranks holding 1, 149 and 53 of 203
    variables (a rank with a single variable), same fit as the oracle's."""
    n, v, m = 300, 203, 4
    launch(3, tmp_path, n, v, m, "exact", extra_env={"LCX_TEST_BOUNDS": "0,1,150,203"})
    got = np.load(os.path.join(tmp_path, "dist_result.npz"))
    x, _ = O.gen_planted(n, v, m, seed=2)
    ref = O.fit_ns(x, m, seed=0, dtype=np.float64, keep_x=True)
    h, h_ref = got["history"], np.asarray(ref.history_tc)
    assert len(h) == len(h_ref) and np.max(np.abs(h - h_ref) / np.maximum(1, np.abs(h_ref))) < 1e-9
    assert np.max(np.abs(got["ws"] - ref.ws)) < 1e-8 and np.array_equal(got["clusters"], ref.clusters())
    assert np.max(np.abs(got["rho"] - ref.moments["rho"])) < 1e-8
    assert int(got["trials"]) == ref.n_trials
    check_covariance(got, ref)


def test_sharded_synergistic_fit_matches_oracle(tmp_path):
    """discourage_overlap=False over two ranks: Y all-reduce, the m+3 sums, H."""
    n, v, m = 300, 203, 4
    launch(2, tmp_path, n, v, m, "syn")
    got = np.load(os.path.join(tmp_path, "dist_result.npz"))
    x, _ = O.gen_planted(n, v, m, seed=2)
    ref = O.fit_syn(x, m, seed=0, dtype=np.float64, keep_x=True)
    h, h_ref = got["history"], np.asarray(ref.history_tc)
    assert len(h) == len(h_ref)
    assert np.max(np.abs(h - h_ref) / np.maximum(1, np.abs(h_ref))) < 1e-9
    assert np.max(np.abs(got["ws"] - ref.ws)) < 1e-8
    assert np.max(np.abs(got["transform"] - ref.transform(ref.x_tilde))) < 1e-8
    assert np.max(np.abs(got["rho"] - ref.moments["rho"])) < 1e-8
    assert np.max(np.abs(got["xz"] - ref.moments["X_i Z_j"])) < 1e-8
    assert np.max(np.abs(got["si"] - ref.moments["Si"])) < 1e-8
    assert np.max(np.abs(got["tcs"] - ref.moments["TCs"])) < 1e-8
    check_covariance(got, ref)


@pytest.mark.parametrize("mode", ["exact", "syn"])
def test_sharded_moments_need_an_explicit_collective(mode, tmp_path):
    """Above the eager-gather size the per-variable moments stay on their shards: dict access raises (no hidden collective),
    pickling on one rank works without them, `gather_moments` on every rank brings them in - and the result is the oracle's."""
    n, v, m = 300, 203, 4
    launch(2, tmp_path, n, v, m, mode, extra_env={"LCX_EAGER_GATHER_ELEMS": "0"})
    got = np.load(os.path.join(tmp_path, "dist_result.npz"))
    x, _ = O.gen_planted(n, v, m, seed=2)
    ref = (O.fit_syn if mode == "syn" else O.fit_ns)(x, m, seed=0, dtype=np.float64, keep_x=True)
    assert np.max(np.abs(got["rho"] - ref.moments["rho"])) < 1e-8
    assert np.max(np.abs(got["xz"] - ref.moments["X_i Z_j"])) < 1e-8


def test_shard_boundaries_are_validated():
    """Comm(bounds=...): `world + 1` increasing boundaries from 0, every rank at least one column, ending at the data's n_variables -
    anything else is refused before a handle exists (one-rank gloo group in a child process)."""
    code = r'''
import os, sys
sys.path.insert(0, %r)
import torch.distributed as dist
dist.init_process_group("gloo", rank=0, world_size=1, init_method="tcp://127.0.0.1:%d")
from linearcorex_amd.comm import Comm
for bad in ([0, 5, 9], [1, 9], [0, 0], [0]):
    try:
        Comm(bounds=bad)
        raise SystemExit("accepted %%r" %% (bad,))
    except ValueError:
        pass
c = Comm(bounds=[0, 9])
assert c.shard(9) == (0, 9) and c.shard(9, 0) == (0, 9)
try:
    c.shard(10)
    raise SystemExit("boundaries that end at 9 accepted for 10 variables")
except ValueError:
    pass
assert Comm().shard(10) == (0, 10)
dist.destroy_process_group()
print("BOUNDS_OK")
''' % (ROOT, free_port())
    p = subprocess.run([sys.executable, "-c", code], capture_output=True, text=True, timeout=120)
    assert p.returncode == 0 and "BOUNDS_OK" in p.stdout, (p.stdout[-500:], p.stderr[-2000:])


def test_shard_boundaries_must_agree_across_the_ranks(tmp_path):
    """Comm(bounds=...) "the same on every rank" is checked, not assumed: two gloo ranks with different boundaries - or one with
    boundaries and one without, or one malformed - all raise on BOTH ranks, at construction, instead of hanging in a gather later;
    matching boundaries pass."""
    code = r'''
import os, sys
sys.path.insert(0, %r)
import torch.distributed as dist
rank = int(sys.argv[1])
dist.init_process_group("gloo", rank=rank, world_size=2, init_method="tcp://127.0.0.1:%d")
from linearcorex_amd.comm import Comm
cases = [([0, 4, 9], [0, 5, 9]), ([0, 4, 9], None), ([0, 4, 9], [0, 9, 4]), (None, [0, 4, 10])]
for k, pair in enumerate(cases):
    try:
        Comm(bounds=pair[rank])
        raise SystemExit("case %%d accepted on rank %%d" %% (k, rank))
    except ValueError as e:
        assert "differs across the ranks" in str(e) or "bounds must be" in str(e) or "another rank" in str(e), str(e)
c = Comm(bounds=[0, 4, 9])
assert c.shard(9) == ((0, 4) if rank == 0 else (4, 9))
assert Comm().shard(9) == ((0, 4) if rank == 0 else (4, 9))
dist.barrier()
dist.destroy_process_group()
print("AGREE_OK")
''' % (ROOT, free_port())
    procs = [subprocess.Popen([sys.executable, "-c", code, str(r)], stdout=subprocess.PIPE, stderr=subprocess.PIPE, text=True) for r in range(2)]
    for p in procs:
        out, err = p.communicate(timeout=180)
        assert p.returncode == 0 and "AGREE_OK" in out, (out[-500:], err[-2000:])
