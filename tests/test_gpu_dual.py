"""GPU parity tests of the dual path (D1-D6) against the reference fixtures and the NumPy oracle."""

from __future__ import annotations

import numpy as np
import pytest
from conftest import DUAL_CASES, relerr, signed_targets

import neolssvm_oracle as orc

pytestmark = pytest.mark.gpu

TOL = 1e-5


@pytest.fixture(scope="module")
def hp():
    import neo_ls_svm_amd as pkg

    pkg.default_context()
    return pkg


@pytest.mark.parametrize("name", DUAL_CASES)
def test_dual_fit_matches_reference_fixture(name, golden_loader, hp):
    g = golden_loader(name)
    nz = g["nz"]
    y, s, is_clf = signed_targets(g)[nz], g["s"][nz], g["task"] == "clf"
    r = hp.dual_fit(g["Xt"], y, s, is_clf)
    assert np.array_equal(r["gammas"], g["gammas"])
    assert relerr(r["loo_errors_gammas"], g["loo_errors_gammas"]) < TOL
    assert r["opt"] == int(g["opt"])
    assert relerr(r["alpha"], g["alpha"]) < TOL
    assert relerr(r["loo_residuals"], g["loo_residuals"]) < TOL
    assert relerr(r["loo_yhat"], g["loo_yhat"]) < TOL
    assert relerr(r["residuals"], g["residuals"]) < TOL
    assert relerr(r["loo_std"], g["loo_std"]) < TOL
    assert abs(r["loo_score"] - float(g["loo_score"])) < 1e-9
    yq, sq = hp.dual_predict(g["Xqt"], g["Xt"], alpha=r["alpha"], L=r["L"])
    assert relerr(yq, g["decision_function"]) < TOL
    assert relerr(sq, g["predict_std"]) < TOL
    # the stored factor is scipy's cho_factor(lower=False) layout
    o = orc.dual_fit_reduced(g["Xt"], y, s, is_clf)
    iu = np.triu_indices(y.size)
    assert relerr(r["L"][iu], o["L"][iu]) < TOL
    assert relerr(r["lam"], o["lam"]) < 1e-8


@pytest.mark.parametrize("task", ["reg", "clf"])
def test_dual_fit_vs_oracle_seeded(task, hp):
    rng = np.random.default_rng(9)
    n, r_ = 777, 37  # ragged sizes
    Xt = rng.standard_normal((n, r_)) * 0.4
    w = rng.standard_normal(r_)
    y = np.sin(Xt @ w) + 0.1 * rng.standard_normal(n) if task == "reg" else np.where(Xt @ w + 0.2 * rng.standard_normal(n) > 0, 1.0, -1.0)
    s = rng.uniform(0.3, 2.0, n)
    o = orc.dual_fit_reduced(Xt, y, s, task == "clf")
    r = hp.dual_fit(Xt, y, s, task == "clf", gamma_index=o["opt"])
    assert relerr(r["loo_errors_gammas"], o["loo_errors_gammas"]) < 1e-6
    assert relerr(r["alpha"], o["alpha"]) < TOL
    assert relerr(r["loo_residuals"], o["loo_residuals"]) < 1e-6
    assert relerr(r["loo_std"], o["loo_std"]) < 1e-6
    assert abs(r["loo_score"] - o["loo_score"]) < 1e-8
    Xq = rng.standard_normal((300, r_)) * 0.4
    yq, sq = hp.dual_predict(Xq, Xt, alpha=o["alpha"], L=o["L"])
    assert relerr(yq, orc.dual_decision_function(Xq, Xt, o["alpha"])) < 1e-10
    assert relerr(sq, orc.dual_predict_std(Xq, Xt, o["L"], o["L_lower"])) < 1e-7


def test_dual_rejects_zero_weights(hp):
    rng = np.random.default_rng(0)
    Xt, y = rng.standard_normal((40, 3)), rng.standard_normal(40)
    s = np.ones(40)
    s[3] = 0.0
    with pytest.raises(ValueError):
        hp.dual_fit(Xt, y, s, False)


@pytest.mark.parametrize("cplx", [False, True])
@pytest.mark.parametrize("n", [1, 5, 31, 32, 33, 63, 64, 65, 127, 128, 129, 300, 640, 1000, 1025, 1537])
def test_own_cholesky_factorisation_matches_numpy(n, cplx, hp):
    """L_ comes from the library's own Cholesky factorisations - real (dual fit, csrc/nls_potrf.h: 128 x 128 leaf in LDS, blocked forward
    substitution, rank-128 update) and complex (primal fit, csrc/nls_zpotrf.h: 32 x 32 leaf, row-wise forward substitution, rank-32 update through
    real planes): against numpy.linalg.cholesky on sizes around the leaf / panel / tile edges."""
    rng = np.random.default_rng(n)
    M = rng.standard_normal((n, n + 3)) + (1j * rng.standard_normal((n, n + 3)) if cplx else 0)
    A = M @ M.conj().T / n + 0.5 * np.eye(n)
    L = hp.cholesky(A)
    L0 = np.linalg.cholesky(A)
    assert L.dtype == L0.dtype
    assert np.max(np.abs(L - L0)) <= 1e-12 * np.max(np.abs(L0))
    assert np.max(np.abs(L @ L.conj().T - A)) <= 1e-13 * n * np.max(np.abs(A))
    if cplx:
        assert np.all(np.diag(L).imag == 0.0) and np.all(np.diag(L).real > 0.0)


@pytest.mark.parametrize("n", [1025, 1537, 2600])
def test_real_cholesky_lookahead_is_bit_identical(n, hp, monkeypatch):
    """The real factorisation's look-ahead (the next diagonal block and panel factored on a side stream beside the trailing update,
    csrc/nls_dual.hip: potrf_lower_real; active from n > 1024) only reorders independent work: the factor is the same bit for bit with
    NLS_POTRF_LOOKAHEAD=0, and repeated calls agree."""
    rng = np.random.default_rng(n)
    M = rng.standard_normal((n, n + 3))
    A = M @ M.T / n + 0.5 * np.eye(n)
    L1 = hp.cholesky(A)
    L2 = hp.cholesky(A)
    monkeypatch.setenv("NLS_POTRF_LOOKAHEAD", "0")
    L0 = hp.cholesky(A)
    assert np.array_equal(L1, L0) and np.array_equal(L2, L0)
    assert np.max(np.abs(L0 @ L0.T - A)) <= 1e-13 * n * np.max(np.abs(A))


@pytest.mark.parametrize("cplx", [False, True])
def test_own_cholesky_reports_the_first_bad_pivot(cplx, hp):
    """LAPACK semantics: info = 1-based index of the first non-positive pivot -> LinAlgError."""
    rng = np.random.default_rng(3)
    n = 400
    M = rng.standard_normal((n, n)) + (1j * rng.standard_normal((n, n)) if cplx else 0)
    A = M @ M.conj().T / n + np.eye(n)
    v = np.linalg.eigh(A[:251, :251])[1][:, 0]
    B = A.copy()
    B[:251, :251] -= 1.001 * np.linalg.eigvalsh(A[:251, :251])[0] * np.outer(v, v.conj())  # the leading 251 x 251 minor just indefinite: find the index
    first_bad = next(k for k in range(1, n + 1) if np.linalg.eigvalsh(B[:k, :k])[0] <= 0)
    with pytest.raises(np.linalg.LinAlgError) as err:
        hp.cholesky(B)
    assert f"pivot {first_bad} " in str(err.value)


def test_own_complex_cholesky_is_the_one_the_fit_returns(golden_loader, hp, monkeypatch):
    """The factor L_ of the primal fit against the reference's own (``primal_reg_n3000_d20_D256.L``, scipy ``cho_factor(lower=False)`` layout),
    through the library's factorisation and - NLS_POTRF=rocsolver - through rocSOLVER / rocBLAS: both to 1e-12."""
    g = golden_loader("primal_reg_n3000_d20_D256")
    iu = np.triu_indices(int(g["D"]) + 1)
    for mode in (None, "rocsolver"):
        if mode:
            monkeypatch.setenv("NLS_POTRF", mode)
        r = hp.primal_fit(g["X"], g["y"], g["s"], g["shift"], g["scale"], g["B"], False, gamma_index=int(g["opt"]))
        assert relerr(r["L"][iu], g["L"][iu]) < 1e-12, mode
        assert relerr(r["beta"], g["beta"]) < 1e-9, mode


@pytest.mark.parametrize("gi", [0, 127])
def test_alpha_is_the_cholesky_resolve(gi, golden_loader, hp):
    """``_neo_ls_svm.py:313-314``: alpha = cho_solve(cho_factor(gamma* diag(sn^-2) + K), y): with the factor requested the returned
    pair satisfies it to rounding, at both edges of the gamma grid; without it alpha comes from the eigendecomposition and agrees
    far inside the parity bar."""
    import scipy.linalg as sla

    g = golden_loader("dual_reg_n1000_d32_w")
    nz = g["nz"]
    Xt, y, s = g["Xt"], signed_targets(g)[nz], g["s"][nz]
    r = hp.dual_fit(Xt, y, s, False, gamma_index=gi)
    a_ref = sla.cho_solve((r["L"], False), y)
    assert np.linalg.norm(r["alpha"] - a_ref) <= 1e-10 * np.linalg.norm(a_ref)
    r2 = hp.dual_fit(Xt, y, s, False, gamma_index=gi, want_L=False)
    assert np.linalg.norm(r2["alpha"] - r["alpha"]) <= 1e-7 * np.linalg.norm(r["alpha"])
    assert np.array_equal(r2["loo_residuals"], r["loo_residuals"]) and np.array_equal(r2["loo_std"], r["loo_std"])
