#!/bin/bash
# where do the waves of the config-2 float64 pass (gemm_tn4) spend their cycles? SQ wave-state counters on the probe variants
R=${GRAFT_REPO_ROOT:-$(pwd)}
cd /tmp && export TMPDIR=/tmp
mkdir -p $R/gpurun_out
rocprofv3 --kernel-trace --pmc SQ_WAVE_CYCLES SQ_WAIT_ANY SQ_WAIT_INST_ANY SQ_ACTIVE_INST_ANY SQ_WAIT_INST_LDS SQ_ACTIVE_INST_LDS SQ_VALU_MFMA_BUSY_CYCLES SQ_LDS_BANK_CONFLICT GRBM_GUI_ACTIVE --output-format csv -d $R/gpurun_out/pmc_probe6 -o p6 -- $R/tools/gemm_probe6 occ > $R/gpurun_out/pmc_probe6.log 2>&1
tail -3 $R/gpurun_out/pmc_probe6.log
python3 - <<PY
import csv, glob, collections
acc = collections.defaultdict(lambda: collections.defaultdict(lambda: [0, 0.0]))
for f in glob.glob("$R/gpurun_out/pmc_probe6/**/*counter_collection.csv", recursive=True):
    for r in csv.DictReader(open(f)):
        k = r["Kernel_Name"][:64] + " grid=" + r.get("Grid_Size", "?") + " vgpr=" + r.get("VGPR_Count", "?") + " lds=" + r.get("LDS_Block_Size", "?")
        a = acc[k][r["Counter_Name"]]
        a[0] += 1; a[1] += float(r["Counter_Value"])
out = open("$R/gpurun_out/r03_pmc_wave_states_c2.txt", "w")
for k, d in sorted(acc.items()):
    if "gemm_tn4" not in k: continue
    v = {c: a[1] / a[0] for c, a in d.items()}
    wc = v.get("SQ_WAVE_CYCLES", 1)
    line = "%s\n   wave cycles %.3e: waiting (waitcnt/barrier) %.1f%%, issue-stalled %.1f%% (LDS issue %.1f%%), issuing %.1f%%; LDS active %.1f%%, LDS bank conflict cycles %.2e; MFMA busy / (GUI active x 1024 SIMDs) %.3f\n" % (
        k, wc, 100 * v.get("SQ_WAIT_ANY", 0) / wc, 100 * v.get("SQ_WAIT_INST_ANY", 0) / wc, 100 * v.get("SQ_WAIT_INST_LDS", 0) / wc,
        100 * v.get("SQ_ACTIVE_INST_ANY", 0) / wc, 100 * v.get("SQ_ACTIVE_INST_LDS", 0) / wc, v.get("SQ_LDS_BANK_CONFLICT", 0),
        v.get("SQ_VALU_MFMA_BUSY_CYCLES", 0) / (v.get("GRBM_GUI_ACTIVE", 1) / 8.0 * 1024.0))
    print(line); out.write(line)
PY
find $R/gpurun_out/pmc_probe6 -name "*.csv" -size +1M -delete
