"""Worker of tests/test_distributed_gpu.py::test_sharded_float32_large_kernels_match_single_gpu: one rank of a gloo group,
every rank with its own engine handle on GPU 0 holding its column block of a float32 problem.  The loop of the reference
(:124-159) is driven for a fixed number of iterations per annealing stage, WITHOUT the final factor sort (whose order is a
near-tie after so few iterations).  argv: out_dir n v m iters_per_stage"""
import os
import sys

import numpy as np

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)


def planted_f32(n, v, m):
    from oracle import corex_oracle as O
    from linearcorex_amd.preprocess import preprocess as pp
    x, _ = O.gen_planted(n, v, m, seed=51)
    return pp(x.astype(np.float32), None, "standard", None)[0]


def run_loop(model, iters):
    for i_eps, eps in enumerate(model._init_weights()):
        model._begin_stage(i_eps, eps)
        for k in range(iters):
            model._iterate(more=k + 1 < iters)
    return np.asarray(model.history["TC"], np.float64)


def main():
    import faulthandler
    faulthandler.dump_traceback_later(int(os.environ.get("LCX_TEST_DUMP_AFTER", "240")), exit=False)
    out_dir, n, v, m, iters = sys.argv[1], int(sys.argv[2]), int(sys.argv[3]), int(sys.argv[4]), int(sys.argv[5])
    import torch.distributed as dist
    from linearcorex_amd import Corex
    from linearcorex_amd.comm import Comm
    dist.init_process_group("gloo")
    comm = Comm()
    xt = planted_f32(n, v, m)
    c0, c1 = comm.shard(v)
    model = Corex(n_hidden=m, seed=0, dtype=np.float32, tol=0.0, device=0, comm=comm,
                  line_search=os.environ.get("LCX_TEST_LINE_SEARCH", "exact"))
    be = model._attach_shard(np.ascontiguousarray(xt[:, c0:c1]), v)
    names = (be.kernel_name(0), be.kernel_name(1), be.kernel_name(2))
    be.timing_enable(True)
    h = run_loop(model, iters)
    merged_passes = be.timing_passes_by_kind()["gemm_nt2"]
    be.timing_enable(False)
    ws = model._gather(be.get_ws(0))
    rho = model._gather(be.get_moment(0, "rho"))
    if comm.rank == 0:
        np.savez(os.path.join(out_dir, "dist_f32.npz"), history=h, ws=ws, rho=rho, trials=model.stats["trials"],
                 world=comm.world, kernel_nt=names[0], kernel_tn=names[1], kernel_merged=names[2], merged_passes=merged_passes,
                 in_library=np.array(bool(getattr(model, "_iterated_in_library", False))),
                 transport=str(getattr(model, "_engine_exchange", None)))
    dist.barrier()
    dist.destroy_process_group()


if __name__ == "__main__":
    main()
