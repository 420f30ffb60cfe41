#!/usr/bin/env python3
"""Per kernel: where its waves spend their cycles (rocprofv3 --pmc SQ_WAVE_CYCLES SQ_WAIT_ANY SQ_WAIT_INST_ANY SQ_ACTIVE_INST_ANY
SQ_WAIT_INST_LDS SQ_ACTIVE_INST_LDS SQ_VALU_MFMA_BUSY_CYCLES SQ_LDS_BANK_CONFLICT GRBM_GUI_ACTIVE).  argv[1]: the output directory."""
import collections
import csv
import glob
import sys

acc = collections.defaultdict(lambda: collections.defaultdict(lambda: [0, 0.0]))
for f in glob.glob(sys.argv[1] + "/**/*counter_collection.csv", recursive=True):
    for r in csv.DictReader(open(f)):
        k = r["Kernel_Name"][:96] + " grid=" + r.get("Grid_Size", "?") + " vgpr=" + r.get("VGPR_Count", "?")
        a = acc[k][r["Counter_Name"]]
        a[0] += 1
        a[1] += float(r["Counter_Value"])
for k, d in sorted(acc.items()):
    if "gemm_" not in k:
        continue
    v = {c: a[1] / a[0] for c, a in d.items()}
    wc = v.get("SQ_WAVE_CYCLES", 1)
    print("%s\n   wave cycles %.3e: waiting (waitcnt/barrier) %.1f%%, issue-stalled %.1f%% (LDS issue %.1f%%), issuing %.1f%%; LDS active %.1f%%, "
          "LDS bank conflict cycles %.2e; MFMA busy / (GUI active x 1024 SIMDs) %.3f"
          % (k, wc, 100 * v.get("SQ_WAIT_ANY", 0) / wc, 100 * v.get("SQ_WAIT_INST_ANY", 0) / wc, 100 * v.get("SQ_WAIT_INST_LDS", 0) / wc,
             100 * v.get("SQ_ACTIVE_INST_ANY", 0) / wc, 100 * v.get("SQ_ACTIVE_INST_LDS", 0) / wc, v.get("SQ_LDS_BANK_CONFLICT", 0),
             v.get("SQ_VALU_MFMA_BUSY_CYCLES", 0) / (v.get("GRBM_GUI_ACTIVE", 1) / 8.0 * 1024.0)))
