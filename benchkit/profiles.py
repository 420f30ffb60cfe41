"""Committed profile figures (PMC traffic, rocprofv3 kernel time): quoted by bench.py only while they describe THIS library (the hash
of the library sources recorded with the profile equals the running library's)."""
import json
import os
import sys

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
if ROOT not in sys.path:
    sys.path.insert(0, ROOT)


def _lib_src_hash():
    try:
        import __graft_entry__ as ge
        return ge._src_hash()
    except Exception:
        return None


def load_pmc_traffic(workload, kernel):
    """HBM bytes per launch of `kernel` from the rocprofv3 PMC passes of this same command (tools/pmc_traffic.py writes
    profiles/pmc_traffic_<workload>.json on the GPU box together with the hash of the library sources it profiled;
    FETCH_SIZE is doubled there as MI355X_MICROARCH.md prescribes for gfx950).  A profile taken from other sources than
    the library that is running is not quoted."""
    path = os.path.join(ROOT, "profiles", "pmc_traffic_%s.json" % workload)
    info = {"traffic_source": None, "mfma_util_pmc": None, "traffic_profile_matches_library": None}
    try:
        with open(path) as f:
            d = json.load(f)
        k = d["kernels"][kernel]
    except Exception:
        return None, info
    info["traffic_source"] = os.path.relpath(path, ROOT)
    info["traffic_profile_lib_src_hash"] = d.get("lib_src_hash")
    ok = d.get("lib_src_hash") is not None and d.get("lib_src_hash") == _lib_src_hash()
    info["traffic_profile_matches_library"] = ok
    if not ok:
        return None, info
    info["mfma_util_pmc"] = k.get("mfma_util")
    info["mfma_flops_per_launch_pmc"] = k.get("mfma_flops_per_launch")
    return k.get("hbm_bytes_per_launch"), info


def load_rocprof_avg(workload, kernel):
    """Average duration of `kernel` in the committed rocprofv3 --kernel-trace --stats summary of this same command,
    quoted beside the live HIP-event figure while the summary's recorded source hash equals the running library's."""
    import csv
    import glob
    best = None
    for path in sorted(glob.glob(os.path.join(ROOT, "profiles", "r*_rocprof_kernel_stats_%s.csv" % workload))):
        meta = path[:-4] + ".meta.json"
        try:
            with open(meta) as f:
                if json.load(f).get("lib_src_hash") != _lib_src_hash():
                    continue
            with open(path) as f:
                for row in csv.DictReader(f):
                    if kernel.replace(" ", "") in row["Name"].replace(" ", ""):
                        best = (float(row["AverageNs"]) / 1e3, os.path.relpath(path, ROOT))
        except Exception:
            continue
    return best if best else (None, None)
