#!/usr/bin/env python3
"""Stress the two-ranks-on-one-GPU launch of tests/test_distributed_gpu.py (GPU box).

The round-1 GPU suite saw that launch stall right after the gloo rendezvous in 2 of ~12 full-suite runs, never in
isolation.  This script repeats it under suite-like conditions (the parent process keeps a HIP context with a resident
engine handle, as pytest does when the in-process GPU tests ran first) with every diagnostic on:
  * LCX_WAIT_TIMEOUT_MS: the engine's bounded wait for a state publication fails with expected / seen sequence numbers,
  * faulthandler: a rank still alive after DUMP seconds prints the Python stack of every thread,
  * per-run wall time, so that slow-but-finishing runs show up as well.

    python tools/dist_stress.py --reps 30 --timeout 60 --dump-after 30 [--no-parent-context] [--mode exact]
"""
import argparse
import os
import socket
import subprocess
import sys
import time

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)


def free_port():
    s = socket.socket()
    s.bind(("127.0.0.1", 0))
    p = s.getsockname()[1]
    s.close()
    return p


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--reps", type=int, default=30)
    ap.add_argument("--timeout", type=int, default=60)
    ap.add_argument("--dump-after", type=int, default=30)
    ap.add_argument("--mode", default="exact")
    ap.add_argument("--shape", default="400x331x5")
    ap.add_argument("--no-parent-context", action="store_true")
    ap.add_argument("--out", default=os.path.join(ROOT, "gpurun_out", "dist_stress"))
    args = ap.parse_args()
    os.makedirs(args.out, exist_ok=True)
    n, v, m = (int(t) for t in args.shape.split("x"))
    keep = None
    if not args.no_parent_context:
        import numpy as np
        from linearcorex_amd import Corex
        keep = Corex(n_hidden=4, seed=0, dtype=np.float64, max_iter=5, device=0).fit(np.random.RandomState(0).randn(300, 200))
        print("parent holds a HIP context (resident handle)", flush=True)
    print("cpus:", os.cpu_count(), "affinity:", len(os.sched_getaffinity(0)), flush=True)
    times, bad = [], 0
    for it in range(args.reps):
        port = free_port()
        t0 = time.time()
        procs = []
        for r in range(2):
            env = dict(os.environ, RANK=str(r), WORLD_SIZE="2", LOCAL_RANK="0", MASTER_ADDR="127.0.0.1", MASTER_PORT=str(port),
                       OMP_NUM_THREADS="2", OPENBLAS_NUM_THREADS="2", LCX_TEST_DUMP_AFTER=str(args.dump_after),
                       LCX_CHECK_RANKS="1", LCX_WAIT_TIMEOUT_MS=str(1000 * max(5, args.dump_after - 10)),
                       LCX_TEST_TRACE="1")
            procs.append(subprocess.Popen([sys.executable, os.path.join(ROOT, "tests", "_dist_worker.py"), args.out, str(n), str(v),
                                           str(m), args.mode, "hip", "25"], env=env, cwd=ROOT, stdout=subprocess.PIPE,
                                          stderr=subprocess.STDOUT))
        outs, timed_out = [], False
        for p in procs:
            try:
                o, _ = p.communicate(timeout=5 if timed_out else args.timeout)
            except subprocess.TimeoutExpired:
                timed_out = True
                for q in procs:
                    q.kill()
                o, _ = p.communicate()
            outs.append(o.decode(errors="replace"))
        dt = time.time() - t0
        times.append(dt)
        rcs = [p.returncode for p in procs]
        status = "TIMEOUT" if timed_out else ("ok" if all(rc == 0 for rc in rcs) else "FAIL rc=%s" % rcs)
        print("run %2d: %6.1f s  %s" % (it, dt, status), flush=True)
        if status != "ok" or dt > 20:
            bad += 1
            for r, o in enumerate(outs):
                print("---- rank %d output (tail) ----\n%s" % (r, o[-6000:]), flush=True)
    times.sort()
    print("summary: %d runs, %d bad/slow, median %.1f s, max %.1f s" % (len(times), bad, times[len(times) // 2], times[-1]))
    del keep
    return 1 if bad else 0


if __name__ == "__main__":
    sys.exit(main())
