"""The reference's module-level helpers (`g`, `g_inv`, `mean_impute`, `random_impute`; linearcorex.py:483-524) and the host
form of `Corex.preprocess(x)` (:397-429), which returns a host array by contract.

`fit`, `transform`, `predict` and `invert` do NOT come through here: they preprocess on the device (lcx_upload_preprocess /
lcx_project_raw / lcx_predict / lcx_invert, include/lcx.h).  What is kept are the reference's exact conventions for callers that use
these functions directly:
  * 'standard'  : mean / sqrt(sum((x-mean)^2)/n_obs) with per-column n_obs when values are missing,
                  std clipped at 1e-10 (:411-415);
  * 'outliers'  : np.std(ddof=0) scaling followed by the tanh tail squash `g` (:418-423);
  * 'empirical' : rank-based gaussianisation (:424-426);
  * 'none'      : pass-through.
"""
from __future__ import annotations

import numpy as np


def g(x, t=4):
    """Suppress outliers of a standard normal: identity on [-t, t], tanh beyond (:483-487)."""
    core = np.clip(x, -t, t)
    return core + np.tanh(x - core)


def g_inv(x, t=4):
    """Inverse of `g` (:490-494)."""
    core = np.clip(x, -t, t)
    return core + np.arctanh(np.clip(x - core, -1 + 1e-10, 1 - 1e-10))


def mean_impute(x, v):
    """Replace cells equal to the sentinel `v` (or NaN) by the column mean of the observed cells;
    returns (imputed copy, observed count per column) (:497-510)."""
    x = np.array(x, copy=True)
    if not np.isnan(v):
        x = np.where(x == v, np.nan, x)
    n_obs = np.zeros(x.shape[1], dtype=np.int64)
    for c in range(x.shape[1]):
        col = x[:, c]
        seen = np.isfinite(col)
        col[np.isnan(col)] = np.mean(col[seen])
        n_obs[c] = seen.sum()
    return x, n_obs


def random_impute(x, v):
    """Replace missing cells by random draws from the observed cells of the column (:513-524)."""
    x = np.array(x, copy=True)
    if not np.isnan(v):
        x = np.where(x == v, np.nan, x)
    for c in range(x.shape[1]):
        col = x[:, c]
        miss = np.where(np.isnan(col))[0]
        col[miss] = np.random.choice(col[np.isfinite(col)], size=len(miss))
    return x


def preprocess(x, theta, gaussianize, missing_values, verbose=False):
    """theta=None: estimate (mean, std) from x ("fit"); returns (x_tilde, theta, n_obs)."""
    if missing_values is not None:
        x, n_obs = mean_impute(x, missing_values)
    else:
        n_obs = len(x)
    if gaussianize == 'standard':
        if theta is None:
            mean = np.mean(x, axis=0)
            std = np.sqrt(np.sum((x - mean) ** 2, axis=0) / n_obs).clip(1e-10)
            theta = (mean, std)
        x = (x - theta[0]) / theta[1]
        if verbose and np.max(np.abs(x)) > 6:
            print("Warning: outliers more than 6 stds away from mean. Consider using gaussianize='outliers'")
    elif gaussianize == 'outliers':
        if theta is None:
            theta = (np.mean(x, axis=0), np.std(x, axis=0, ddof=0).clip(1e-10))
        x = g((x - theta[0]) / theta[1])
    elif gaussianize == 'empirical':
        from scipy.stats import norm, rankdata
        print("Warning: correct inversion/transform of empirical gauss transform not implemented.")
        x = np.array([norm.ppf((rankdata(col) - 0.5) / len(col)) for col in x.T]).T
    return x, theta, n_obs
