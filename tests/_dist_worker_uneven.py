"""Worker of tests/test_distributed_gpu.py::test_uneven_shards_many_ranks: one rank of a gloo group of 4 or 8, every rank with its
own engine handle on GPU 0 holding an UNEVEN column block (Comm(bounds=...): a rank with fewer than 16 variables = less than one
panel, one with fewer than a block's 256 / 512 columns, widths that no tile divides).  The loop of the reference (:124-159) runs for
a fixed number of iterations per annealing stage with the exchange steps and the line search inside the library (hook transport),
LCX_CHECK_RANKS=1.  argv: out_dir n m iters dtype bounds(comma separated)"""
import os
import sys

import numpy as np

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)


def planted(n, v, m, dtype):
    from oracle import corex_oracle as O
    from linearcorex_amd.preprocess import preprocess as pp
    x, _ = O.gen_planted(n, v, min(m, 24), seed=77)
    return pp(x.astype(dtype), None, "standard", None)[0]


def run_loop(model, iters):
    for i_eps, eps in enumerate(model._init_weights()):
        model._begin_stage(i_eps, eps)
        for k in range(iters):
            model._iterate(more=k + 1 < iters)
    return np.asarray(model.history["TC"], np.float64)


def main():
    import faulthandler
    faulthandler.dump_traceback_later(int(os.environ.get("LCX_TEST_DUMP_AFTER", "240")), exit=False)
    out_dir, n, m, iters = sys.argv[1], int(sys.argv[2]), int(sys.argv[3]), int(sys.argv[4])
    dtype = np.float32 if sys.argv[5] == "f32" else np.float64
    bounds = [int(t) for t in sys.argv[6].split(",")]
    v = bounds[-1]
    import torch.distributed as dist
    from linearcorex_amd import Corex
    from linearcorex_amd.comm import Comm
    dist.init_process_group("gloo")
    comm = Comm(bounds=bounds)
    assert comm.world == len(bounds) - 1
    xt = planted(n, v, m, dtype)
    c0, c1 = comm.shard(v)
    model = Corex(n_hidden=m, seed=0, dtype=dtype, tol=0.0, device=0, comm=comm)
    be = model._attach_shard(np.ascontiguousarray(xt[:, c0:c1]), v)
    # the first-contact test of the transport ran inside bind_engine with all `world` ranks (closed-form sums use the rank count)
    assert comm.selftest_seconds is not None and comm.selftest_seconds > 0
    again = be.comm_selftest(comm.rank)
    h = run_loop(model, iters)
    ws = model._gather(be.get_ws(0))
    rho = model._gather(be.get_moment(0, "rho"))
    model.ws = ws                               # (no `_finish` in this loop: transform_fitted wants the weights in place)
    y = model.transform_fitted()
    info = be.exchange_info()
    if comm.rank == 0:
        np.savez(os.path.join(out_dir, "dist_uneven.npz"), history=h, ws=ws, rho=rho, y=y, trials=model.stats["trials"],
                 world=comm.world, selftest_seconds=np.array([comm.selftest_seconds, again]),
                 kernels=np.array([be.kernel_name(0), be.kernel_name(1)]), allreduces=info["allreduces_issued"],
                 in_library=np.array(bool(getattr(model, "_iterated_in_library", False))),
                 transport=str(getattr(model, "_engine_exchange", None)))
    dist.barrier()
    dist.destroy_process_group()


if __name__ == "__main__":
    main()
