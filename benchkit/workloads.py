"""The workload table of bench.py (BASELINE.json `configs` and the ad-hoc shapes), and the hardware peaks the roofline is priced
against (/opt/skills/guides/MI355X_MICROARCH.md)."""

HBM_PEAK_GBS = 8000.0          # MI355X_MICROARCH.md: HBM3E 8.0 TB/s spec
FP64_MFMA_PEAK_TFLOPS = 78.6
FP32_MFMA_PEAK_TFLOPS = 157.3
BF16_MFMA_PEAK_TFLOPS = 2500.0   # dense; the split float32 contraction issues 6 bf16 products per float32 product
SPLIT_PRODUCTS = 6
# What this chip has been seen to sustain (profiles/r01_read_probe_c2.txt, r01_mfma_peak.txt, the c3f64* workloads):
# reported beside the spec-based fraction, never instead of it.
MEASURED_CEILINGS = {"hbm_read_GBps": 6470.0, "mfma_f64_TFLOPs": 72.0, "mfma_f64_16x16x4_TFLOPs": 59.9,
                     "mfma_f32_TFLOPs": 151.0}
MIN_TIMED_SECONDS = 0.6        # one walk of the schedule shorter than this is repeated (median per stage)

WORKLOADS = {
    # name: (n_samples, n_variables per GPU, n_hidden, dtype)
    "c2": (10000, 5000, 32, "f64"),      # BASELINE.json configs[1]
    "c3": (50000, 100000, 64, "f32"),    # configs[2] (MFMA roofline run; X generated on device)
    "c4shard": (50000, 125000, 128, "f32"),  # configs[3], one GPU's shard
    "c4full": (50000, 1000000, 128, "f32"),  # configs[3] unsharded on ONE GPU: single-copy mode (gemm_cr), 200 GB of X
    "c3f64": (50000, 50000, 64, "f64"),  # large float64 shards (gemm_ct on float64; not BASELINE lines)
    "c3f64m32": (50000, 50000, 32, "f64"),
    "c3f64m128": (50000, 50000, 128, "f64"),
    "c2f32": (10000, 5000, 32, "f32"),   # config-2 shape in the reference's own precision (not a BASELINE line)
    "c2m64": (10000, 5000, 64, "f64"), "c2m64f32": (10000, 5000, 64, "f32"), "c2m128f32": (10000, 5000, 128, "f32"),
    "mid32": (20000, 20000, 32, "f64"), "mid32f32": (20000, 20000, 32, "f32"), "mid64f32": (20000, 20000, 64, "f32"),
    "c5": (400, 20000, 30, "f64"),       # configs[4] stand-in shape (few samples, many variables; not a bench line)
    "c5f32": (400, 20000, 30, "f32"),
    "tiny": (2000, 640, 8, "f64"),       # plumbing check
}
DESCRIPTION = {
    "c2": "BASELINE.json configs[1]", "c3": "BASELINE.json configs[2], the MFMA roofline run",
    "c4shard": "BASELINE.json configs[3], one GPU's shard of the 1M-variable problem",
    "c4full": "BASELINE.json configs[3] unsharded: the whole 1M-variable problem on one GPU, one resident copy of X",
}


def _shrink():
    """LCX_BENCH_SHRINK=k (test hook): the BASELINE workloads with n_variables / k, so that the WHOLE default job - every nested block,
    CPU legs included - runs live in the GPU suite in about a minute (tests/test_bench_gpu.py).  The shapes then select other kernels
    than the real configs: the hook tests bench.py, not the kernels, and says so on the line."""
    import os
    k = int(os.environ.get("LCX_BENCH_SHRINK", "0") or 0)
    if k > 1:
        for name in ("c2", "c3", "c4shard"):
            n, v, m, tag = WORKLOADS[name]
            WORKLOADS[name] = (n, max(256, v // k // 64 * 64), m, tag)
            DESCRIPTION[name] = DESCRIPTION[name] + " SHRUNK: n_variables / %d (LCX_BENCH_SHRINK, a test of bench.py - not a BASELINE figure)" % k
    return k


SHRUNK = _shrink()


def _adhoc(name):
    """'NxVxM:f32' -> WORKLOADS entry (probing shapes outside BASELINE.json)."""
    if name not in WORKLOADS and name != "auto":
        dims, tag = name.split(":")
        n, v, m = (int(t) for t in dims.split("x"))
        assert tag in ("f32", "f64")
        WORKLOADS[name] = (n, v, m, tag)
