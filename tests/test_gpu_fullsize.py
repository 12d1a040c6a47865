"""Full-size (BASELINE c3: n = 1e6, d = 128, D = 4096, G = 1024) check of the primal fit through size-independent
identities - the oracle cannot run at this size, the algebra can:

* normal equations of the re-solve:  (gamma* c I + A) beta = b  with A, b from the Gram hook, c = 1 / (n (D + 1));
* the two routes to the selected column agree: the EVD sweep's LOO residuals and leverages reproduce the Cholesky
  route's residuals,  e_loo (1 - leverage) = residuals  (regression, no clipping; leverage_i = s_i^2 phi_i M^-1 phi_i^H
  with the normalised s, ``_neo_ls_svm.py:136-150,169``);
* residuals = Re(phi beta) - y on sampled rows (feature-map hook), sigma from the stored factor (predict hook) ties
  loo_std to the leverage:  leverage = s^2 sigma^2,  loo_std^2 = sigma^2 + (s sigma^2)^2 / (1 - leverage)  (``:184-187``);
* loo_errors[gamma*] = sum_i s_i |e_i|;  eigenvalues positive, ascending, trace(A) / c = sum(lam).
"""
from __future__ import annotations

import sys
from pathlib import Path

import numpy as np
import pytest

ROOT = Path(__file__).resolve().parent.parent
sys.path.insert(0, str(ROOT))

pytestmark = pytest.mark.gpu


def test_c3_identities():
    import bench
    import neo_ls_svm_amd as hp

    cfg = bench.CONFIGS["c3"]
    n, d, D = cfg["n"], cfg["d"], cfg["D"]
    ctx = hp.default_context()
    X, y = bench.synth(n, d, 0, n)
    s = np.ones(n)
    shift, scale, B = bench.affine_params(n, d, D, ctx=ctx)  # pre-step on the 2e5-row prefix
    dX = ctx.to_device(X)
    r = hp.primal_fit(dX, y, s, shift, scale, B, False, ctx=ctx)
    D1 = D + 1
    c = 1.0 / (n * D1)
    sn = s / s.sum()
    g, lam = r["gamma"], r["lam"]
    assert r["gammas"].shape == (1024,) and np.all(np.isfinite(r["loo_errors_gammas"]))
    assert r["opt"] == int(np.argmin(r["loo_errors_gammas"]))

    # eigenvalues and normal equations
    A, b = hp.gram(dX, y, s, shift, scale, B, ctx=ctx)
    assert np.all(np.diff(lam) >= 0) and lam[0] > -1e-9 * lam[-1]
    assert abs(np.trace(A).real / c - lam.sum()) <= 1e-9 * lam.sum()
    resid = (g * c) * r["beta"] + A @ r["beta"] - b
    assert np.linalg.norm(resid) <= 1e-9 * np.linalg.norm(b)

    # EVD route vs Cholesky route, every row
    lev = r["loo_leverage"]
    assert np.all(lev > 0) and np.all(lev < 1)
    lhs = r["loo_residuals"] * (1.0 - lev)
    assert np.max(np.abs(lhs - r["residuals"])) <= 1e-8 * np.max(np.abs(r["residuals"]))
    assert abs(np.sum(sn * np.abs(r["loo_residuals"])) - r["loo_errors_gammas"][r["opt"]]) <= 1e-10 * r["loo_errors_gammas"][r["opt"]]

    # sampled rows through the inference hooks
    idx = np.random.default_rng(1).choice(n, size=2048, replace=False)
    Xs = np.ascontiguousarray(X[idx])
    yhat, sigma = hp.primal_predict(Xs, shift, scale, B, beta=r["beta"], L=r["L"], ctx=ctx)
    assert np.max(np.abs((yhat - y[idx]) - r["residuals"][idx])) <= 1e-9 * np.max(np.abs(r["residuals"]))
    phi = hp.featuremap(Xs[:256], shift, scale, B, ctx=ctx)
    assert np.max(np.abs((phi @ r["beta"]).real - yhat[:256])) <= 1e-10 * np.max(np.abs(yhat))
    s2 = sigma**2
    assert np.max(np.abs(sn[idx] ** 2 * s2 - lev[idx])) <= 1e-8 * np.max(lev)
    loo_var = s2 + (sn[idx] * s2) ** 2 / (1.0 - lev[idx])
    assert np.max(np.abs(np.sqrt(loo_var) - r["loo_std"][idx])) <= 1e-8 * np.max(r["loo_std"])


def test_c4_dual_identities():
    """Dual path at BASELINE c4 (n = 1e4, r = 256): residuals_ = F alpha - y (``_neo_ls_svm.py:313-317``) reproduced by the
    inference hook on the training rows (k(x_i, X) alpha + sum(alpha) = (F alpha)_i with F = rbf + 1), and the re-solve's
    normal equations  (gamma* diag(sn^-2) + F) alpha = y  through that same product."""
    import neo_ls_svm_amd as hp

    n, r_ = 10_000, 256
    rng = np.random.default_rng(4)
    Xt = rng.standard_normal((n, r_)) * 0.25
    w = rng.standard_normal(r_) / np.sqrt(r_)
    y = np.sin(4 * Xt @ w) + 0.1 * rng.standard_normal(n)
    s = rng.uniform(0.5, 2.0, n)
    r = hp.dual_fit(Xt, y, s, False)
    assert r["opt"] == int(np.argmin(r["loo_errors_gammas"])) and np.all(np.isfinite(r["loo_residuals"]))
    yhat, sigma = hp.dual_predict(Xt, Xt, alpha=r["alpha"], L=r["L"])
    assert np.max(np.abs((yhat - y) - r["residuals"])) <= 1e-9 * np.max(np.abs(y))
    sn = s / s.sum()
    sn = sn / np.median(np.abs(sn))  # :253-254
    lhs = r["gamma"] * r["alpha"] / sn**2 + yhat
    assert np.max(np.abs(lhs - y)) <= 1e-8 * np.max(np.abs(y))
    assert np.all(np.isfinite(sigma)) and np.all(sigma >= 0)


def test_c5_sigma_grid_full_16x32():
    """BASELINE config 5 exactly as ``bench.py --config c5`` runs it: n = 1e6, d = 128, D = 4096, the 16 sigmas
    ``logspace(1/4, 4, 16)`` x the 32 gammas ``gamma_grid(1024)[::33]``.  Every sigma - the extremes 1/4 (arguments x 4) and 4
    (features nearly collinear: the most ill-conditioned normal equations of the grid) included - is also fitted UNCONDITIONALLY with
    its Cholesky factor, and must satisfy the identities of ``test_c3_identities`` that need no oracle:
      argmin of its own curve; trace(A) / c = sum(lam); LOO error = sum s |e_loo| at gamma*; e_loo (1 - leverage) = residuals_ on all
      rows; the normal equations (gamma* c I + A) beta = b; beta == cho_solve(L_, b) (``_neo_ls_svm.py:176-178``).
    The early-out driver's 16 x 32 table must equal the table of those curves bit for bit and name the same winner."""
    import scipy.linalg as sla

    import bench
    import neo_ls_svm_amd as hp

    cfg = bench.CONFIGS["c5"]
    n, d, D = cfg["n"], cfg["d"], cfg["D"]
    ctx = hp.default_context()
    X, y = bench.synth(n, d, 0, n)
    s = np.ones(n)
    sn = s / s.sum()
    shift, scale, B = bench.affine_params(n, d, D, ctx=ctx)
    dX, dy, ds = ctx.to_device(X), ctx.to_device(y), ctx.to_device(s)
    del X
    gammas = hp.gamma_grid(1024)[::33]
    sigmas = np.logspace(np.log10(0.25), np.log10(4.0), cfg["sigmas"])
    assert gammas.size == 32 and sigmas.size == 16
    grid = hp.primal_fit_sigma_grid(dX, dy, ds, shift, scale, B, False, sigmas, gammas=gammas, ctx=ctx)
    c = 1.0 / (n * (D + 1))
    curves, minima, report = [], [], []
    best_full = None
    for k, sg in enumerate(sigmas):
        r = hp.primal_fit(dX, y, s, shift, scale, B / sg, False, gammas=gammas, ctx=ctx)
        assert np.array_equal(grid["loo_errors"][k], r["loo_errors_gammas"]), k
        assert np.array_equal(grid["objective"][k], r["objective"]), k
        assert r["opt"] == int(np.argmin(r["loo_errors_gammas"]))
        e_opt = r["loo_errors_gammas"][r["opt"]]
        assert abs(np.sum(sn * np.abs(r["loo_residuals"])) - e_opt) <= 1e-10 * e_opt
        lev = r["loo_leverage"]
        assert np.all(lev > 0) and np.all(lev < 1)
        assert np.max(np.abs(r["loo_residuals"] * (1.0 - lev) - r["residuals"])) <= 1e-8 * np.max(np.abs(r["residuals"]))
        A, b = hp.gram(dX, y, s, shift, scale, B / sg, ctx=ctx)
        lam = r["lam"]
        assert np.all(np.diff(lam) >= 0) and lam[0] > -1e-9 * lam[-1]
        assert abs(np.trace(A).real / c - lam.sum()) <= 1e-9 * lam.sum()
        resid = np.linalg.norm((r["gamma"] * c) * r["beta"] + A @ r["beta"] - b) / np.linalg.norm(b)
        assert resid <= 1e-9, (k, resid)
        beta_chol = sla.cho_solve((r["L"], False), b)
        dbeta = np.linalg.norm(r["beta"] - beta_chol) / np.linalg.norm(beta_chol)
        assert dbeta <= 1e-9, (k, dbeta)
        report.append((float(sg), int(r["opt"]), float(e_opt), float(lam[0]), float(lam[-1]), float(resid), float(dbeta)))
        curves.append(r["loo_errors_gammas"])
        minima.append(r["objective"].min())
        if best_full is None or r["objective"].min() < best_full[0]:
            best_full = (r["objective"].min(), k, {key: r[key].copy() for key in ("beta", "loo_residuals", "loo_leverage", "loo_std", "residuals")}, r["opt"])
        del r, A
    k_best = int(np.argmin(minima))
    assert grid["sigma_index"] == k_best == best_full[1] and grid["gamma_index"] == best_full[3]
    best = grid["best"]
    assert best is not None
    for key in ("beta", "loo_residuals", "loo_leverage", "loo_std", "residuals"):
        assert np.array_equal(best[key], best_full[2][key]), key
    out = ROOT / "gpurun_out"
    out.mkdir(exist_ok=True)
    with open(out / "c5_full_grid_report.txt", "w") as fh:
        fh.write("sigma gamma_index loo_error lam_min lam_max normal_eq_residual beta_vs_cho_solve\n")
        for row in report:
            fh.write(" ".join(f"{v:.6g}" for v in row) + "\n")
        fh.write(f"winner sigma_index {k_best} gamma_index {best_full[3]}\n")
