#!/usr/bin/env python3
"""Turn rocprofv3 PMC passes (FETCH_SIZE and WRITE_SIZE, collected in separate runs of the SAME
bench.py command) into HBM bytes per launch for every kernel, and write
profiles/pmc_traffic_<workload>.json, which bench.py reads for `roofline.traffic`.

Units and corrections (MI355X_MICROARCH.md, "HBM"): FETCH_SIZE / WRITE_SIZE are in KiB; on gfx950
FETCH_SIZE reports exactly half of the bytes of a wide (16 B/lane) coalesced streaming read, so it is
doubled; WRITE_SIZE is uncalibrated and taken as is.

    python tools/pmc_traffic.py --workload c2 --fetch DIR_OR_CSV --write DIR_OR_CSV [--out profiles/...]
"""
import argparse
import csv
import glob
import json
import os
import re
import sys


def find_csv(path):
    if os.path.isfile(path):
        return [path]
    return sorted(glob.glob(os.path.join(path, "**", "*counter_collection.csv"), recursive=True))


def short(name):
    name = re.sub(r"^void\s+", "", name.strip().strip('"'))
    depth, out = 0, []
    for ch in name:                      # cut the argument list (first '(' outside <...>)
        if ch == "<":
            depth += 1
        elif ch == ">":
            depth -= 1
        elif ch == "(" and depth == 0:
            break
        out.append(ch)
    return "".join(out).strip()


def collect(path, counter):
    acc = {}
    for f in find_csv(path):
        with open(f, newline="") as fh:
            for row in csv.DictReader(fh):
                if row.get("Counter_Name") != counter:
                    continue
                k = short(row["Kernel_Name"])
                d = acc.setdefault(k, {"launches": 0, "sum": 0.0})
                d["launches"] += 1
                d["sum"] += float(row["Counter_Value"])
    return acc


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--workload", required=True)
    ap.add_argument("--fetch", required=True)
    ap.add_argument("--write", required=True)
    ap.add_argument("--mfma", default=None, help="pass with SQ_VALU_MFMA_BUSY_CYCLES GRBM_GUI_ACTIVE SQ_INSTS_VALU_MFMA_MOPS_F64/_F32")
    ap.add_argument("--command", default="")
    ap.add_argument("--lib-src-hash", default=None, help="hash of the library sources that were profiled (__graft_entry__._src_hash)")
    ap.add_argument("--out", default=None)
    a = ap.parse_args()
    fetch, write = collect(a.fetch, "FETCH_SIZE"), collect(a.write, "WRITE_SIZE")
    kernels = {}
    for k in sorted(set(fetch) | set(write)):
        f, w = fetch.get(k), write.get(k)
        fk = f["sum"] / f["launches"] if f else None
        wk = w["sum"] / w["launches"] if w else None
        rd = None if fk is None else 2.0 * fk * 1024.0
        wr = None if wk is None else wk * 1024.0
        kernels[k] = {"launches_fetch_pass": f["launches"] if f else 0, "launches_write_pass": w["launches"] if w else 0,
                      "FETCH_SIZE_KiB_per_launch_raw": fk, "WRITE_SIZE_KiB_per_launch_raw": wk,
                      "hbm_read_bytes_per_launch": rd, "hbm_write_bytes_per_launch": wr,
                      "hbm_bytes_per_launch": None if rd is None and wr is None else (rd or 0.0) + (wr or 0.0)}
    if a.mfma:
        # MFMA pipe utilisation per kernel: busy cycles summed over the SIMDs / (active cycles x 256 CUs x 4 SIMDs).
        # SQ_VALU_MFMA_BUSY_CYCLES books the nominal pipe occupancy (64 cycles per v_mfma_f64_16x16x4, 32 per
        # v_mfma_f32_16x16x4): on gfx950 the f64 instruction really issues every ~104 cycles (tools/mfma_peak.hip),
        # so a float64 kernel cannot exceed ~0.62 by this counter.
        # MOPS counters are in units of 512 flop (rocprofv3 -L: MfmaFlopsF64 = SQ_INSTS_VALU_MFMA_MOPS_F64 * 512)
        busy, act = collect(a.mfma, "SQ_VALU_MFMA_BUSY_CYCLES"), collect(a.mfma, "GRBM_GUI_ACTIVE")
        m64, m32 = collect(a.mfma, "SQ_INSTS_VALU_MFMA_MOPS_F64"), collect(a.mfma, "SQ_INSTS_VALU_MFMA_MOPS_F32")
        mb16 = collect(a.mfma, "SQ_INSTS_VALU_MFMA_MOPS_BF16")       # the split float32 contraction: 6 bf16 products per float32 product
        for k in busy:
            d = kernels.setdefault(k, {})
            n = busy[k]["launches"]
            b = busy[k]["sum"] / n
            # the CSV carries the SUM over the 8 XCD instances of GRBM_GUI_ACTIVE (rocprofv3's own MfmaUtil takes the max)
            g = act[k]["sum"] / act[k]["launches"] / 8.0 if k in act else None
            d["mfma_busy_cycles_per_launch"] = b
            d["gui_active_cycles_per_launch"] = g
            d["mfma_util"] = (b / (g * 1024.0)) if g else None
            flops = 512.0 * ((m64[k]["sum"] / m64[k]["launches"] if k in m64 else 0.0) +
                             (m32[k]["sum"] / m32[k]["launches"] if k in m32 else 0.0))
            d["mfma_flops_per_launch"] = flops
            if k in mb16 and mb16[k]["sum"] > 0:
                d["mfma_bf16_flops_per_launch"] = 512.0 * mb16[k]["sum"] / mb16[k]["launches"]
    lib_hash = a.lib_src_hash
    if lib_hash is None:
        try:
            sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
            import __graft_entry__ as ge
            lib_hash = ge._src_hash()
        except Exception:
            lib_hash = None
    out = {"workload": a.workload, "command": a.command, "lib_src_hash": lib_hash,
           "corrections": "FETCH_SIZE x2 (gfx950 wide coalesced reads are tallied at half size), KiB -> bytes; "
                          "WRITE_SIZE as reported (uncalibrated)",
           "kernels": kernels}
    path = a.out or os.path.join(os.path.dirname(os.path.dirname(os.path.abspath(__file__))), "profiles",
                                 "pmc_traffic_%s.json" % a.workload)
    with open(path, "w") as fh:
        json.dump(out, fh, indent=1, sort_keys=True)
    top = sorted(kernels.items(), key=lambda kv: -(kv[1].get("hbm_bytes_per_launch") or 0))[:8]
    for k, v in top:
        print("%-80s %8.1f MB/launch (rd %s wr %s) mfma_util %s flops %s" % (
            k[:80], (v.get("hbm_bytes_per_launch") or 0) / 1e6, v.get("hbm_read_bytes_per_launch"),
            v.get("hbm_write_bytes_per_launch"), v.get("mfma_util"), v.get("mfma_flops_per_launch")))
    return 0


if __name__ == "__main__":
    sys.exit(main())
