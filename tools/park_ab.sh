#!/bin/bash
# What the ranks that do NOT time the CPU baseline may do meanwhile (bench.py, _Park), measured on the GPU box's host cores.
#   bash tools/park_ab.sh ranks     the real thing: `bench.py --gpus 2` rehearsed on one GPU (lean job), the waiting rank parked (sleeping poll,
#                                   what bench.py does) or busy-waiting (LCX_TEST_PARK=spin: what waiting in an RCCL barrier amounts to), alternating
#   bash tools/park_ab.sh host      the same question without a GPU in the loop, up to 7 waiters (the 8-GPU job): bench.cpu_baseline_generated of a
#                                   config-4-shaped matrix alone / beside k sleeping pollers / beside k spinning processes, alternating on one box
# -> gpurun_out/r06_park_ab_<what>.txt
cd ${GRAFT_REPO_ROOT:-$(pwd)}; mkdir -p gpurun_out
case "$1" in
  ranks)
    python3 __graft_entry__.py || exit 1
    for mode in sleep spin sleep spin; do
      if [ $mode = spin ]; then export LCX_TEST_PARK=spin; else unset LCX_TEST_PARK; fi
      LCX_BENCH_LEAN=1 LCX_BENCH_DEVICE=0 LCX_BENCH_BACKEND=gloo timeout 900 python bench.py --gpus 2 --steps 5 --warmup 2 --repeats 1 --detail-out gpurun_out/r06_park_$mode.json > gpurun_out/r06_park_line.json 2>gpurun_out/r06_park.err || { tail -5 gpurun_out/r06_park.err; }
      python - <<PY
import json
d=json.load(open("gpurun_out/r06_park_$mode.json")); cb=d["cpu_baseline"]
print("park=$mode  cpu_baseline %.4f it/s at timed size %.4f  cores %d  loadavg %s  others: %s" % (cb["value"], cb["measured_iterations_per_sec_at_timed_size"], cb["cores"], cb["host"]["loadavg_1_5_15"], cb["other_ranks_while_timed"][:40]), flush=True)
PY
    done | tee gpurun_out/r06_park_ab_ranks.txt ;;
  host)
    python - <<'PY' | tee gpurun_out/r06_park_ab_host.txt
import os, subprocess, sys, time
sys.path.insert(0, os.getcwd())
import numpy as np
import bench
SPIN = "import time\nwhile True:\n    pass\n"
SLEEP = "import os, time\nwhile not os.path.exists('/nonexistent/lcx_park'):\n    time.sleep(0.2)\n"
print("# usable cores %d of %d logical; loadavg at start %s" % (bench._usable_cores(), os.cpu_count(), os.getloadavg()), flush=True)
for rnd in range(3):
    for what, k in (("alone", 0), ("sleeping pollers", 1), ("spinning", 1), ("sleeping pollers", 7), ("spinning", 7)):
        procs = [subprocess.Popen([sys.executable, "-c", SPIN if what == "spinning" else SLEEP]) for _ in range(k)]
        time.sleep(1.0)
        try:
            r = bench.cpu_baseline_generated(50000, 20000, 128, np.float32, 6.0, "c4-shaped, 20000 variables")
        finally:
            for p in procs:
                p.kill()
            for p in procs:
                p.wait()
        print("round %d  %-18s x %d   %.4f it/s at the timed size  (BLAS threads %d, loadavg before %s)"
              % (rnd, what, k, r["measured_iterations_per_sec_at_timed_size"], r["cores"], r["host"]["loadavg_1_5_15"]), flush=True)
PY
    ;;
  *) echo "usage: park_ab.sh ranks|host"; exit 2 ;;
esac
