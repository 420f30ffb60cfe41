"""NumPy test double of the backend interface (`linearcorex_amd.backend.HipBackend`).

TEST INFRASTRUCTURE ONLY - it lets the CPU suite exercise the product's *host* logic (the `Corex`
driver: line search, annealing stages, warm start, pickling, and the n_variables sharding with its
all-reduce exchange steps over gloo) without a GPU.  The arithmetic is the oracle's, regrouped at
the same dependency levels as the C ABI (include/lcx.h): every quantity that needs a sum over all
variables is written to an exchange buffer between the *_a/_b/_c calls, exactly like the HIP path.
The product never imports this module.
"""
import numpy as np
import torch


class ShardDouble:
    def __init__(self, n_samples, nv_local, n_hidden, dtype=np.float64):
        self.n, self.nv, self.m = int(n_samples), int(nv_local), int(n_hidden)
        self.dtype = np.dtype(dtype)
        self.m_pad = self.m
        self.ndiv = self.n           # the reference's self.n_samples (set_sample_divisor: another fit's count)
        self.generation = 0
        self.x = None
        self.w = [np.zeros((self.m, self.nv), self.dtype), np.zeros((self.m, self.nv), self.dtype)]
        self.mo = [dict(), dict()]
        self.state = [np.zeros(8), np.zeros(8)]
        self.ybuf = np.zeros(self.n * self.m + self.m * self.m, self.dtype)
        # layout of include/lcx.h (lcx_read_sbuf): [0..8) scalars, [8, 8+m^2) H, then the detail sums
        self.sbuf = np.zeros(8 + self.m * self.m + self.m + 8, np.float64)
        self.SB_H, self.SB_DET = 8, 8 + self.m * self.m
        self.grad = self.update = self.sig_grad = None
        self.calls = []

    # ---- plumbing ----------------------------------------------------------------------------
    def close(self):
        pass

    def set_sample_divisor(self, n_samples):
        self.ndiv = float(n_samples)

    def geometry(self):
        return {"m_pad": self.m}

    def synchronize(self):
        pass

    def set_world(self, world):
        self.world = world

    def exchange_tensors(self):
        return torch.from_numpy(self.ybuf), torch.from_numpy(self.sbuf)

    def stream_context(self):
        import contextlib
        return contextlib.nullcontext()

    def sbuf_ranges(self):
        return (0, self.SB_DET), (self.SB_DET, self.m + 3)

    def read_sbuf(self, count):
        return self.sbuf[self.SB_DET:self.SB_DET + count].copy()

    def upload_x(self, x):
        assert x.shape == (self.n, self.nv)
        self.x = np.ascontiguousarray(x, self.dtype)

    def upload_preprocess(self, x_raw, gaussianize, missing_values=None, theta=None):
        from linearcorex_amd.preprocess import preprocess as pp
        x, th, n_obs = pp(np.asarray(x_raw, self.dtype), theta, gaussianize, missing_values)
        if th is None:
            th = (np.zeros(self.nv, self.dtype), np.ones(self.nv, self.dtype))
        self.upload_x(np.asarray(x, self.dtype))
        return th, n_obs, float(np.max(np.abs(x)))

    def project_raw(self, x_raw, gaussianize, theta):
        from linearcorex_amd.preprocess import preprocess as pp
        x = pp(np.asarray(x_raw, self.dtype), theta, gaussianize, None)[0]
        return self.project(x)

    def set_ws(self, w):
        self.w[0] = np.array(w, self.dtype)
        self.generation += 1

    def get_ws(self, which=0):
        return self.w[which].copy()

    def permute_factors(self, order):
        self.w[0] = self.w[0][np.asarray(order)]
        self.generation += 1

    def _y(self):
        return self.ybuf[:self.n * self.m].reshape(self.n, self.m)

    def _tail(self):
        return self.ybuf[self.n * self.m:]

    # ---- moments (ref :236-275) -----------------------------------------------------------------
    def moments_a(self, which):
        w = self.w[which]
        self._y()[:] = self.x.dot(w.T)
        self._tail()[:] = w.dot(w.T).ravel()
        self.calls.append("moments_a")

    def moments_b(self, which, eps, quick):
        y = self._y().copy()
        self._finish_moments(which, eps, quick, y, None)
        self.calls.append("moments_b")

    def _finish_moments(self, which, eps, quick, y, d):
        """d = X^T.Y of the shard (nv x m); None -> one pass over X."""
        w, n = self.w[which], self.ndiv
        gw = self._tail().reshape(self.m, self.m).copy()
        dt = self.dtype.type
        c1, c2 = dt(1 - eps ** 2), dt(eps ** 2)
        ry = c1 * y.T.dot(y) / dt(n) + c2 * gw
        uj = np.diag(ry).copy()
        np.fill_diagonal(ry, 1)
        st = self.state[which]
        st[1] = uj.max()
        st[4] = np.sum(np.log(1 - uj)) if np.all(uj < 1) else np.nan
        invalid = bool(quick and uj.max() >= 1.0)
        st[2] = 1.0 if invalid else 0.0
        mo = {"uj": uj, "ry": ry, "wmag": np.diag(gw).copy(), "Y": y}
        if not invalid:
            if d is None:
                d = self.x.T.dot(y)
            mo["D"] = d
            rho = c1 * d.T / dt(n) + c2 * w
            inv = 1.0 / (1.0 - rho ** 2)
            rir = rho * inv
            qij = ry.dot(rir)
            si = np.sum(rho * rir, axis=0)
            q2 = np.einsum("ki,ki->i", rir, qij - si * rho)
            mo.update(rho=rho, invrho=inv, rhoinvrho=rir, Qij=qij, Si=si)
            mo["Qi-Si^2"] = q2
            with np.errstate(all="ignore"):
                self.sbuf[0] = np.sum(np.log(1 + si).astype(np.float64))
                self.sbuf[1] = np.sum(np.log(1 + q2).astype(np.float64))
                self.sbuf[self.SB_H:self.SB_DET] = np.dot(rir / (1 + q2), rir.T).ravel()     # H partial of this set
        self.mo[which] = mo
        if which == 0:
            self.generation += 1

    # linear trial mode (include/lcx.h lcx_trial_linear_a/_b)
    def trial_linear_a(self, eta):
        self.make_trial(eta)
        w = self.w[1]
        self._tail()[:] = w.dot(w.T).ravel()
        self._ytrial = self.mo[0]["Y"] + self.dtype.type(eta) * self.ydir
        self.calls.append("trial_linear_a")

    def trial_linear_b(self, eps, eta):
        d = self.mo[0]["D"] + self.dtype.type(eta) * self.ddir
        self._finish_moments(1, eps, True, self._ytrial, d)
        self.calls.append("trial_linear_b")

    def moments_c(self, which):
        st = self.state[which]
        if st[2] != 0:
            st[0] = np.nan
        else:
            st[0] = float(self.dtype.type(self.sbuf[0] - 0.5 * self.sbuf[1] + 0.5 * st[4]))
        st[3] = self.sbuf[2]                      # update_tangent of the direction in flight
        self.calls.append("moments_c")

    def moments_detail(self, which):
        mo, m = self.mo[which], self.m
        rho = mo["rho"]
        mi = -0.5 * np.log1p(-rho ** 2)
        xz = np.linalg.solve(mo["ry"], rho).T
        x2y = (1.0 - np.einsum("ij,ji->i", xz, rho)).clip(1e-6)
        mo.update(MI=mi)
        mo["X_i Z_j"] = xz
        mo["X_i^2 | Y"] = x2y
        d = self.SB_DET
        self.sbuf[d:d + m] = mi.sum(axis=1)
        self.sbuf[d + m] = mi.max(axis=0).sum()
        self.sbuf[d + m + 1] = (-0.5 * np.log(x2y)).sum()
        self.sbuf[d + m + 2] = mi.sum()

    # ---- update (ref :290-305) -------------------------------------------------------------------
    def update_a(self):
        mo = self.mo[0]
        rir = mo["rhoinvrho"]
        self.sbuf[self.SB_H:self.SB_DET] = np.dot(rir / (1 + mo["Qi-Si^2"]), rir.T).ravel()
        self.calls.append("update_a")

    def update_b(self, eps):
        mo, w = self.mo[0], self.w[0]
        h = self.sbuf[self.SB_H:self.SB_DET].reshape(self.m, self.m).astype(self.dtype)
        np.fill_diagonal(h, 0)
        rj = 1.0 - mo["uj"][:, np.newaxis]
        rho, inv, rir = mo["rho"], mo["invrho"], mo["rhoinvrho"]
        grad = w / rj
        grad -= 2 * inv * rir / (1 + mo["Si"])
        grad += inv ** 2 * ((1 + rho ** 2) * mo["Qij"] - 2 * rho * mo["Si"]) / (1 + mo["Qi-Si^2"])
        grad += np.dot(h, w)
        self.grad = grad
        self._y()[:] = self.x.dot(grad.T)
        self._tail()[:self.m] = np.sum(rho * grad, axis=1)
        mo["H"] = h
        self.calls.append("update_b")

    def update_c(self, eps):
        mo, w = self.mo[0], self.w[0]
        dt = self.dtype.type
        yg = self._y().copy()
        bj = self._tail()[:self.m].copy()[:, np.newaxis]
        sg = dt(1 - eps ** 2) * self.x.T.dot(yg).T / dt(self.ndiv) + dt(eps ** 2) * self.grad
        rj = 1.0 - mo["uj"][:, np.newaxis]
        self.update = -rj * (self.grad - 2.0 * w / (2 - rj) * bj)
        self.sig_grad = sg
        c = 2.0 * bj / (2 - rj)                                   # (m, 1)
        self.ydir = (-rj * (yg.T - c * mo["Y"].T)).T                # Y(update), n x m
        self.ddir = (-rj * (self.x.T.dot(yg).T - c * mo["D"].T)).T  # D(update), nv x m
        self.sbuf[2] = float(np.einsum("ji,ji", sg.astype(np.float64), self.update.astype(np.float64)))
        self.calls.append("update_c")

    def update_d(self):
        pass

    # ---- synergistic branch (ref :336-384), cut at the same exchange points as include/lcx.h --------------
    def syn_moments_b(self, which, yscale):
        w, n, m = self.w[which], self.ndiv, self.m
        y = self._y().copy()
        xy = self.x.T.dot(y) / n                                     # nv x m   (:355)
        cy = y.T.dot(y) / n + yscale ** 2 * np.eye(m)                # == ws.dot(X_i Y_j) + yscale^2 I (:356)
        yj2 = np.diag(cy).copy()
        sd = np.sqrt(yj2)
        ry = cy / (sd * sd[:, np.newaxis])
        rho = (xy / sd).T
        xz = np.linalg.solve(cy, xy.T).T
        x2y = (1.0 - np.einsum("ij,ij->i", xz, xy)).clip(1e-6)
        mi = -0.5 * np.log1p(-rho ** 2)
        self.mo[which] = {"syn X_i Y_j": xy, "cy": cy, "Y_j^2": yj2, "ry": ry, "rho": rho, "syn X_i Z_j": xz,
                          "syn X_i^2 | Y": x2y, "MI": mi, "D": xy * n}
        d = self.SB_DET
        self.sbuf[d:d + m] = mi.sum(axis=1)
        self.sbuf[d + m] = 0.0
        self.sbuf[d + m + 1] = (-0.5 * np.log(x2y)).sum()
        self.sbuf[d + m + 2] = mi.sum()
        self.state[which][4] = np.sum(0.5 * np.log(yj2) - 0.5 * np.log(yscale ** 2))
        self.state[which][2] = 0.0
        if which == 0:
            self.generation += 1
        self.calls.append("syn_moments_b")

    def syn_moments_c(self, which):
        self.state[which][0] = self.sbuf[self.SB_DET + self.m + 1] - self.state[which][4]
        self.calls.append("syn_moments_c")

    def syn_update_a(self):
        mo = self.mo[0]
        xz, x2y = mo["syn X_i Z_j"], mo["syn X_i^2 | Y"]
        self.sbuf[self.SB_H:self.SB_DET] = (1.0 / x2y * xz.T).dot(xz).ravel()
        self.calls.append("syn_update_a")

    def syn_update_b(self, eta):
        mo = self.mo[0]
        h = self.sbuf[self.SB_H:self.SB_DET].reshape(self.m, self.m).copy()
        np.fill_diagonal(h, 0)
        r = mo["syn X_i Z_j"].T / mo["syn X_i^2 | Y"]
        self.w[1] = ((1.0 - eta) * self.w[0] + eta * (r - np.dot(h, self.w[0]))).astype(self.dtype)
        self.calls.append("syn_update_b")

    def covariance_syn(self, std):
        mo = self.mo[0]
        cov = np.einsum("ij,kj->ik", mo["syn X_i Z_j"], mo["syn X_i Y_j"])
        np.fill_diagonal(cov, 1)
        return std[:, np.newaxis] * std * cov

    def make_trial(self, eta):
        self.w[1] = self.w[0] + self.dtype.type(eta) * self.update

    def accept_trial(self):
        self.w.reverse()
        self.mo.reverse()
        self.state.reverse()
        self.generation += 1

    def rescale_ws(self, eps_old, eps_new):
        mo, w = self.mo[0], self.w[0]
        wmag = mo["wmag"][:, np.newaxis]
        delta = (eps_new ** 2 - eps_old ** 2) / (1.0 - eps_new ** 2) * wmag / mo["uj"].reshape((-1, 1))
        a = np.sqrt((1.0 - eps_old ** 2) / ((1.0 - eps_new ** 2) * (1.0 + delta)))
        self.w[0] = w * (0.001 * np.floor(1000.0 * a)).astype(self.dtype)
        self.generation += 1

    def init_scale_ws(self):
        y = self._y()
        uj = np.einsum("lj,lj->j", y, y) / self.ndiv
        self.w[0] = self.w[0] / (10.0 * np.sqrt(uj))[:, np.newaxis].astype(self.dtype)
        self.generation += 1

    # ---- readback / outputs ---------------------------------------------------------------------
    def read_state(self, which):
        return self.state[which].copy()

    def get_moment(self, which, name, eps=0.0):
        if name in ("grad", "update", "sig_grad"):
            return getattr(self, name).copy()
        if name in ("MI", "X_i Z_j", "X_i^2 | Y") and name not in self.mo[which]:
            keep = self.sbuf.copy()
            self.moments_detail(which)
            self.sbuf[:] = keep
        return np.array(self.mo[which][name], copy=True)

    def set_moment(self, which, name, value):
        self.mo[which][name] = np.array(value, self.dtype)

    def covariance(self, eps, std):
        mo = self.mo[0]
        z = mo["rhoinvrho"] / (1 + mo["Si"])
        cov = np.dot(z.T, z) / (1.0 - eps ** 2)
        np.fill_diagonal(cov, 1)
        return std[:, np.newaxis] * std * cov

    def covariance_rows(self, eps, std, row0, nrows, syn=False):
        cov = self.covariance_syn(std) if syn else self.covariance(eps, std)
        return cov[row0:row0 + nrows]

    def project(self, x):
        return np.asarray(x, self.dtype).dot(self.w[0].T)

    def predict(self, y, xz=None, syn=False, gaussianize=None, theta=None):
        from oracle import corex_oracle as O
        if xz is None:
            xz = self.get_moment(0, "syn X_i Z_j" if syn else "X_i Z_j")
        return O.predict(np.asarray(xz, self.dtype), np.asarray(y, self.dtype), theta, gaussianize)

    def invert(self, x, gaussianize, theta):
        from oracle import corex_oracle as O
        return O.invert(np.asarray(x, self.dtype), theta, gaussianize)

    def project_resident(self):
        return self.x.dot(self.w[0].T)
