"""Worker of tests/test_distributed_cpu.py: one rank of a world_size-N gloo group running the
product's `Corex` driver over the NumPy backend double.  argv: out_dir n v m"""
import os
import sys

import numpy as np

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)

import torch.distributed as dist  # noqa: E402

from linearcorex_amd import Corex  # noqa: E402
from linearcorex_amd.comm import Comm  # noqa: E402
from oracle import corex_oracle as O  # noqa: E402
from tests.shard_double import ShardDouble  # noqa: E402


def main():
    # a rank that is still running after this many seconds prints the Python stack of every thread (the launching
    # test kills the ranks and shows this output)
    import faulthandler
    faulthandler.dump_traceback_later(int(os.environ.get("LCX_TEST_DUMP_AFTER", "240")), exit=False)
    out_dir, n, v, m = sys.argv[1], int(sys.argv[2]), int(sys.argv[3]), int(sys.argv[4])
    mode = sys.argv[5] if len(sys.argv) > 5 else "exact"
    backend = sys.argv[6] if len(sys.argv) > 6 else "double"
    max_iter = int(sys.argv[7]) if len(sys.argv) > 7 else 10000
    syn = mode == "syn"                     # discourage_overlap=False (reference :336-384)
    if syn:
        mode = "exact"
    trace = os.environ.get("LCX_TEST_TRACE", "0") not in ("", "0")
    t_start = __import__("time").time()

    def mark(what):
        if trace:
            print("[rank %s +%.2fs] %s" % (os.environ.get("RANK"), __import__("time").time() - t_start, what), flush=True)

    mark("before rendezvous")
    dist.init_process_group("gloo")
    bounds = os.environ.get("LCX_TEST_BOUNDS")          # explicit (uneven) shard boundaries instead of the balanced split
    comm = Comm(bounds=[int(t) for t in bounds.split(",")] if bounds else None)
    mark("rendezvous done")
    x, _ = O.gen_planted(n, v, m, seed=2)
    dt = np.float32 if os.environ.get("LCX_TEST_DTYPE") == "f32" else np.float64
    if backend == "hip":
        # every rank drives its own engine handle on GPU 0; the exchange tensors are CUDA tensors and the
        # collectives go through gloo (RCCL refuses two ranks on one device) - same host code as bench.py
        model = Corex(n_hidden=m, seed=0, dtype=dt, comm=comm, line_search=mode, device=0, max_iter=max_iter,
                      discourage_overlap=not syn)
    else:
        model = Corex(n_hidden=m, seed=0, dtype=np.float64, comm=comm, line_search=mode, max_iter=max_iter,
                      discourage_overlap=not syn,
                      _backend_factory=lambda ns, nv, mm, dt: ShardDouble(ns, nv, mm, dt))
    mark("model constructed")
    y_resident = model.fit_transform(x)     # one pass over the resident shard + the same all-reduce as transform
    mark("fit done: %d iterations" % len(model.history["TC"]))
    c0, c1 = comm.shard(v)
    assert model._backend.nv == c1 - c0
    y = model.transform(x)
    assert y_resident.shape == y.shape and np.max(np.abs(y_resident - y)) < (1e-11 if dt == np.float64 else 2e-5) * max(1.0, float(np.max(np.abs(y))))
    xr = model.predict(y[:50])              # sharded columns of the product, gathered: a collective like transform
    if os.environ.get("LCX_EAGER_GATHER_ELEMS") == "0":
        # sharded moments were NOT put together at the end of fit: touching one must raise (never a hidden collective that
        # the other ranks do not join), pickling must work without them, and the explicit collective brings them in
        import pickle
        try:
            model.moments["rho"]
            raise SystemExit("moments['rho'] on a sharded model did not raise")
        except RuntimeError:
            pass
        assert "rho" not in model.moments and "TC" in model.moments
        if comm.rank == 0:
            back = pickle.loads(pickle.dumps(model))
            assert "rho" not in back.moments and back.ws.shape == (m, v)
        model.gather_moments(["rho", "X_i Z_j", "Si"])
        assert "rho" in model.moments
    elif comm.rank == 0:
        # the usual pattern: one rank alone looks at the results / saves the model (vis_corex.py:549) - no collective may
        # hide behind that
        import pickle
        back = pickle.loads(pickle.dumps(model))
        assert back.moments["rho"].shape == (m, v)
        assert model.mis.shape == (m, v) and model.moments["MI"].shape == (m, v)
    rho = model.moments["rho"]
    xz = model.moments["X_i Z_j"]
    si = model.moments["Si"]
    # get_covariance over the ranks (a collective): the whole matrix while it is small, and a block of rows that straddles a
    # shard boundary - every rank gets the same (rows, n_variables) array
    cov = model.get_covariance() if v <= 1000 else np.zeros(1)
    b = comm.shard(v, 0)[1]
    cov_rows = model.get_covariance(rows=(max(0, b - 100), min(v, b + 156)))
    assert cov_rows.shape == (min(v, b + 156) - max(0, b - 100), v)
    if comm.rank == 0:
        np.savez(os.path.join(out_dir, "dist_result.npz"), history=np.asarray(model.history["TC"], np.float64),
                 ws=model.ws, clusters=model.clusters(), transform=y, predict=xr, rho=rho, xz=xz, si=si, cov=cov, cov_rows=cov_rows, cov_row0=max(0, b - 100),
                 tcs=model.tcs, world=comm.world, trials=model.stats["trials"],
                 transport=str(getattr(model, "_engine_exchange", None)), f32_gemm=str(getattr(model, "f32_gemm", "mfma")),
                 kernel=str(model._backend.kernel_name(0)) if hasattr(model._backend, "kernel_name") else "",
                 in_library=np.array(bool(getattr(model, "_iterated_in_library", False))),
                 calls=np.array(len(getattr(model._backend, "calls", []))))
    mark("results gathered")
    dist.barrier()
    dist.destroy_process_group()
    mark("process group destroyed")


if __name__ == "__main__":
    main()
