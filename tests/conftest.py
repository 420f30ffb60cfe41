"""pytest configuration: markers, import path, shared fixtures."""
import os
import sys

import numpy as np
import pytest

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
if ROOT not in sys.path:
    sys.path.insert(0, ROOT)

GOLDEN = os.path.join(ROOT, "tests", "golden")


def pytest_configure(config):
    config.addinivalue_line("markers", "gpu: needs a real MI355X (run with -m gpu on the GPU box)")


def xpass_names_ok(name_xbt, name_xty, ct=None, panel=True):
    """The kernel functions behind the two X passes of a shard on the stream-K pair: gemm_cr<.., true, true> / gemm_ct<.., true, true>
    on the panel-major copy - or, when the suite is run with LCX_F32_GEMM=split (the float32 passes on the bf16 pipe by default),
    gemm_split_kernel<CT, ..> for both."""
    name_xbt, name_xty = str(name_xbt), str(name_xty)
    if "gemm_split_kernel" in name_xbt or "gemm_split_kernel" in name_xty:
        ok = os.environ.get("LCX_F32_GEMM") == "split" and "gemm_split_kernel" in name_xbt and "gemm_split_kernel" in name_xty
        return ok and (ct is None or (("gemm_split_kernel<%d," % ct) in name_xbt and ("gemm_split_kernel<%d," % ct) in name_xty))
    ok = "gemm_cr_kernel" in name_xbt and "gemm_ct_kernel" in name_xty
    if ct is not None:
        ok = ok and ("_kernel<float, %d," % ct) in name_xbt and ("_kernel<float, %d," % ct) in name_xty
    if panel:
        ok = ok and name_xbt.endswith("true, true>") and name_xty.endswith("true, true>")
    return ok


def load_golden(name):
    return np.load(os.path.join(GOLDEN, name + ".npz"))


@pytest.fixture(scope="session")
def g1():
    return load_golden("g1_big5")


@pytest.fixture(scope="session")
def g2_small():
    return load_golden("g2_planted_small")


@pytest.fixture(scope="session")
def g2_mid():
    return load_golden("g2_planted_mid")


@pytest.fixture(scope="session")
def g3():
    return load_golden("g3_c2_step")


@pytest.fixture(scope="session")
def g4():
    return load_golden("g4_edges")


@pytest.fixture(scope="session")
def g5():
    return load_golden("g5_outliers")


@pytest.fixture(scope="session")
def g6():
    return load_golden("g6_adni_missing")


def _set_line_search(request, monkeypatch):
    monkeypatch.setenv("LCX_LINE_SEARCH", request.param)
    return request.param


@pytest.fixture(params=["exact"])
def ls(request, monkeypatch):
    """The line search of an end-to-end fixture, set through the environment so that models built by the CLI and by child
    processes follow it too.  Round 4 ran EVERY end-to-end fixture under "exact" (reference-shaped: each back-tracking trial makes
    two passes over X, linearcorex.py:321) and "exact-y" (trials after the first one take X.w_update^T by linearity,
    lcx_set_trial_reuse) at the same bars to decide the default; the decision is made ("exact"), so the
    matrix is reduced to one representative per fixture family: `ls_both` below (big5, the whole config-2 fit) and the two-rank
    run of tests/test_distributed_gpu.py keep "exact-y" at the same bars."""
    return _set_line_search(request, monkeypatch)


@pytest.fixture(params=["exact", "exact-y"])
def ls_both(request, monkeypatch):
    return _set_line_search(request, monkeypatch)
