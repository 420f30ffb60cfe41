"""Synergistic branch (discourage_overlap=False; reference linearcorex.py:336-384, :452-455) on the device against
the fixture generated from the reference (g8_syn.npz) and the oracle.

float64 engine vs the float64-lifted reference: step level 1e-9, end to end same iteration count and 1e-6.
float32 engine vs the reference verbatim (which runs x in float32 but W and the moments in float64, :121):
TC within 1e-3 relative, clusters bit-exact on planted data."""
import numpy as np
import pytest

from oracle import corex_oracle as O
from tests.conftest import load_golden
from tests.test_parity_gpu import relerr

pytestmark = pytest.mark.gpu


def _kn(k):
    return k.replace(" ", "_").replace("^", "p").replace("|", "g")


def test_syn_step_level():
    g = load_golden("g8_syn")
    from linearcorex_amd.backend import HipBackend
    p = "planted_f64_step_"
    xt, w_in = g["planted_f64_x_tilde"], g[p + "w_in"]
    n, v = xt.shape
    m = w_in.shape[0]
    be = HipBackend(n, v, m, np.float64, 0)
    be.upload_x(xt)
    be.set_ws(w_in)
    be.moments_a(0); be.syn_moments_b(0, 1.0); be.syn_moments_c(0)
    st = be.read_state(0)
    assert abs(st[0] - float(g[p + "in_TC"])) < 1e-9 * max(1.0, abs(float(g[p + "in_TC"])))
    for key, name in (("syn X_i Y_j", "X_i Y_j"), ("cy", "cy"), ("Y_j^2", "Y_j^2"), ("ry", "ry"), ("rho", "rho"),
                      ("syn X_i Z_j", "X_i Z_j"), ("syn X_i^2 | Y", "X_i^2 | Y")):
        assert relerr(be.get_moment(0, key), g[p + "in_" + _kn(name)]) < 1e-9, key
    sums = be.read_sbuf(m + 3)
    iyx = 0.5 * np.log(g[p + "in_" + _kn("Y_j^2")])
    assert relerr(sums[:m] - iyx, g[p + "in_TCs"]) < 1e-9
    assert abs((sums[m + 2] - sums[m + 1]) - float(g[p + "in_additivity"])) < 1e-8
    be.syn_update_a(); be.syn_update_b(float(g[p + "eta"]))
    assert relerr(be.get_ws(1), g[p + "w_out"]) < 1e-10
    be.moments_a(1); be.syn_moments_b(1, 1.0); be.syn_moments_c(1)
    assert abs(be.read_state(1)[0] - float(g[p + "out_TC"])) < 1e-9 * max(1.0, abs(float(g[p + "out_TC"])))
    be.close()


@pytest.mark.parametrize("tag", ["f32", "f64"])
def test_syn_end_to_end(tag, g1):
    from linearcorex_amd import Corex
    g = load_golden("g8_syn")
    dt = np.float32 if tag == "f32" else np.float64
    for name, x, m in (("planted", O.gen_planted(400, 300, 5, seed=4)[0], 5), ("big5", g1["x_raw"].astype(np.float64), 5)):
        out = Corex(n_hidden=m, seed=0, dtype=dt, device=0, discourage_overlap=False).fit(x)
        p = "%s_%s_" % (name, tag)
        h, h_ref = np.asarray(out.history["TC"], np.float64), g[p + "history_tc"]
        # `moments` lists the 16 keys of the reference's dict (:348-373) in its order, before any device array has been read
        from tests.test_host_logic_cpu import reference_moment_keys
        assert sorted(out.moments) == sorted(reference_moment_keys(g, p[:-1])) and len(out.moments.keys()) == 16
        assert list(out.moments) == list(out.moments._ORDER_SYN) and len(out.moments._stored()) < 16
        if tag == "f64":
            assert len(h) == len(h_ref), name
            assert relerr(h, h_ref) < 1e-6
            assert relerr(out.ws, g[p + "ws"]) < 1e-6
            assert relerr(out.get_covariance(), g[p + "cov"]) < 1e-6           # north_star tolerance, syn branch
            assert relerr(out.get_covariance(rows=(3, 11)), g[p + "cov"][3:11]) < 1e-6
            assert relerr(out.transform(x), g[p + "transform"]) < 1e-6
            assert np.array_equal(out.clusters(), g[p + "clusters"])
            for k in ("TCs", "rho", "X_i Z_j", "X_i Y_j", "X_i^2 | Y", "cy", "ry", "Qij", "Qi", "Si", "MI", "Y_j^2"):
                assert relerr(out.moments[k], g[p + "mom_" + _kn(k)]) < 1e-6, k
            assert abs(out.moments["additivity"] - float(g[p + "mom_additivity"])) < 1e-6
        else:
            # the run ends on a long tail of ~1e-5 TC increments against tol = 1e-5: a float32 engine (the
            # reference holds W and the moments of this branch in float64) leaves it earlier or later
            assert abs(len(h) - len(h_ref)) <= 0.3 * len(h_ref)
            assert abs(h[-1] - h_ref[-1]) < 1e-3 * abs(h_ref[-1])
            assert relerr(out.get_covariance(), g[p + "cov"]) < 5e-3
            if name == "planted":
                assert np.array_equal(out.clusters(), g[p + "clusters"])


def test_syn_two_ranks_one_gpu(tmp_path):
    import os
    from tests.test_distributed_gpu import launch_hip
    import tests.test_distributed_gpu as tdg
    n, v, m = 400, 331, 5
    launch_hip(2, tmp_path, n, v, m, "syn")
    got = np.load(os.path.join(tmp_path, "dist_result.npz"))
    x, _ = O.gen_planted(n, v, m, seed=2)
    ref = O.fit_syn(x, m, seed=0, dtype=np.float64, keep_x=True, max_iter=tdg.MAX_ITER)
    h, h_ref = got["history"], np.asarray(ref.history_tc)
    assert len(h) == len(h_ref)
    assert np.max(np.abs(h - h_ref) / np.maximum(1, np.abs(h_ref))) < 1e-8
    assert np.max(np.abs(got["ws"] - ref.ws)) < 1e-7
    assert np.max(np.abs(got["xz"] - ref.moments["X_i Z_j"])) < 1e-7
    assert np.max(np.abs(got["tcs"] - ref.moments["TCs"])) < 1e-7
