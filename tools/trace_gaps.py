#!/usr/bin/env python3
"""Where does a fit iteration's time go between the kernels?  Reads a rocprofv3 --kernel-trace CSV of a bench.py run and
prints, per kernel function: launches, mean duration, mean idle gap BEFORE it (start - end of the previous dispatch on the
device), and the share of the whole timeline; then the totals (busy, idle).  Gaps above --cut microseconds (host pauses
between phases of the script) are left out of the idle statistics.

    python tools/trace_gaps.py gpurun_out/prof_c2/c2_kernel_trace.csv [--cut 200] [--skip-first 400]
"""
import argparse
import csv
import re
import sys
from collections import OrderedDict


def short(name):
    name = re.sub(r"^void\s+", "", name.strip().strip('"'))
    name = re.sub(r"\(.*$", "", name)
    return name[:70]


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("csv")
    ap.add_argument("--cut", type=float, default=200.0)
    ap.add_argument("--skip-first", type=int, default=0, help="dispatches to skip (setup)")
    a = ap.parse_args()
    rows = []
    with open(a.csv, newline="") as f:
        for r in csv.DictReader(f):
            rows.append((int(r["Start_Timestamp"]), int(r["End_Timestamp"]), short(r["Kernel_Name"])))
    rows.sort()
    rows = rows[a.skip_first:]
    stats = OrderedDict()
    busy = idle = 0.0
    prev_end = None
    n_cut = 0
    for st, en, name in rows:
        d = stats.setdefault(name, {"n": 0, "dur": 0.0, "gap": 0.0, "ngap": 0})
        d["n"] += 1
        d["dur"] += (en - st) / 1e3
        busy += (en - st) / 1e3
        if prev_end is not None:
            g = (st - prev_end) / 1e3
            if g > a.cut:
                n_cut += 1
            else:
                g = max(g, 0.0)
                d["gap"] += g
                d["ngap"] += 1
                idle += g
        prev_end = max(prev_end or en, en)
    tot = busy + idle
    print("%-70s %7s %10s %10s %8s" % ("kernel", "calls", "avg us", "gap us", "% time"))
    for name, d in sorted(stats.items(), key=lambda kv: -(kv[1]["dur"] + kv[1]["gap"])):
        print("%-70s %7d %10.2f %10.2f %7.2f%%" % (name, d["n"], d["dur"] / d["n"], d["gap"] / max(1, d["ngap"]),
                                                    100.0 * (d["dur"] + d["gap"]) / tot))
    print("busy %.1f ms, idle between dependent launches %.1f ms (%.1f%%), %d gaps above %.0f us left out"
          % (busy / 1e3, idle / 1e3, 100.0 * idle / tot, n_cut, a.cut))
    return 0


if __name__ == "__main__":
    sys.exit(main())
