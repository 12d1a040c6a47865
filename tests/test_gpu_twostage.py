"""GPU tests of the two-stage eigendecomposition (dense -> band -> tridiagonal, two back-transformations; csrc/nls_sb.h, nls_chase.h,
nls_q2.h) that ``nls_eigh_only`` / the fits take under ``NLS_EVD=twostage``: every stage through its own hook against NumPy, the
whole decomposition against ``numpy.linalg.eigh``, the fall-back on degenerate panels, and the two fits on reference fixtures."""
from __future__ import annotations

import sys
from pathlib import Path

import numpy as np
import pytest
from conftest import load_golden, relerr, signed_targets

ROOT = Path(__file__).resolve().parent.parent
sys.path.insert(0, str(ROOT / "tools"))
from twostage_proto import apply_q2_naive  # noqa: E402  (NumPy prototype: reflector-by-reflector product)

pytestmark = pytest.mark.gpu


@pytest.fixture(scope="module")
def hp():
    import neo_ls_svm_amd as pkg

    pkg.default_context()
    return pkg


def _herm(n, cplx, seed, spd=False):
    rng = np.random.default_rng(seed)
    M = rng.standard_normal((n, n)) + (1j * rng.standard_normal((n, n)) if cplx else 0)
    return M @ M.conj().T / n if spd else (M + M.conj().T) / 2


def _band(Aout, bw):
    L = np.tril(Aout) - np.tril(Aout, -bw - 1)
    B = L + np.tril(L, -1).conj().T
    B[np.diag_indices_from(B)] = B[np.diag_indices_from(B)].real
    return B


CASES = [(False, 32), (False, 64), (True, 32)]


@pytest.mark.parametrize("cplx,bw", CASES)
@pytest.mark.parametrize("n", [40, 100, 257, 700])
def test_stages_against_numpy(n, cplx, bw, hp):
    rng = np.random.default_rng(n)
    A = _herm(n, cplx, 10 + n)
    ev = np.linalg.eigvalsh(A)
    scale = np.max(np.abs(ev))
    # stage 1: the band matrix is unitarily similar to A; nothing is left below the band except the reflectors
    Aout, tau1, failed, nred = hp.twostage_stage(1, A, bw)
    assert not failed and nred == max(0, n - bw - 1)
    Bd = _band(Aout, bw)
    assert np.max(np.abs(np.linalg.eigvalsh(Bd) - ev)) <= 1e-13 * n * scale
    # stage 2: the tridiagonal matrix has the band matrix's spectrum, and the stored reflectors reproduce it: Q2^H B Q2 = T
    d, e, V2, timed_out = hp.twostage_stage(2, np.tril(Bd), bw)
    assert not timed_out
    T = np.diag(d) + np.diag(e, 1) + np.diag(e, -1)
    assert np.max(np.abs(np.linalg.eigvalsh(T) - ev)) <= 1e-13 * n * scale
    Q2 = apply_q2_naive(V2, bw, np.eye(n, dtype=A.dtype))
    assert np.max(np.abs(Q2.conj().T @ Bd @ Q2 - T)) <= 1e-13 * n * scale
    # stage 3: the blocked (diamond, MFMA) application equals the reflector-by-reflector product, also on a ragged column count
    Z = (rng.standard_normal((n, 37)) + (1j * rng.standard_normal((n, 37)) if cplx else 0)).astype(A.dtype)
    assert np.max(np.abs(hp.twostage_stage(3, V2, bw, aux=Z) - apply_q2_naive(V2, bw, Z.copy()))) <= 1e-13 * n


def test_second_back_transformation_window_sizes(hp, monkeypatch):
    """The number of sweep groups kept in the LDS window (NLS_Q2_GROUPS) must not change the result."""
    n, bw = 333, 32
    A = _herm(n, True, 5)
    _, _, V2, _ = hp.twostage_stage(2, np.tril(A) - np.tril(A, -bw - 1), bw)
    Z = np.eye(n, dtype=np.complex128)[:, :50]
    ref = apply_q2_naive(V2, bw, Z.copy())
    for g in ("1", "2", "3", "8"):
        monkeypatch.setenv("NLS_Q2_GROUPS", g)
        assert np.max(np.abs(hp.twostage_stage(3, V2, bw, aux=Z) - ref)) <= 1e-13 * n, g


@pytest.mark.parametrize("cplx,bw", CASES)
@pytest.mark.parametrize("n", [5, 33, 34, 65, 66, 130, 257, 1025])
def test_eigh_two_stage_matches_numpy(n, cplx, bw, hp, monkeypatch):
    monkeypatch.setenv("NLS_EVD", "twostage")
    monkeypatch.setenv("NLS_SB_BW", str(bw))
    A = _herm(n, cplx, 200 + n, spd=True)
    lam, Q = hp.eigh(A)
    lam0 = np.linalg.eigvalsh(A)
    assert np.all(np.diff(lam) >= 0)
    assert np.max(np.abs(lam - lam0)) <= 1e-12 * lam0[-1] * max(n, 8)
    assert np.max(np.abs(A @ Q - Q * lam[None, :])) <= 1e-12 * lam0[-1] * max(n, 8)
    assert np.max(np.abs(Q.conj().T @ Q - np.eye(n))) <= 1e-12 * max(n, 8)


def test_eigh_two_stage_at_path_sizes(hp, monkeypatch):
    """Complex n = 4097 (c3's D + 1) and real n = 6500: clustered low end like the path's matrices."""
    monkeypatch.setenv("NLS_EVD", "twostage")
    for n, cplx in ((4097, True), (6500, False)):
        rng = np.random.default_rng(n)
        k = n // 2
        M = rng.standard_normal((n, k)) + (1j * rng.standard_normal((n, k)) if cplx else 0)
        A = M @ M.conj().T / k + 1e-3 * np.eye(n)
        lam, Q = hp.eigh(A)
        nrm = lam[-1]
        assert abs(lam.sum() - np.trace(A).real) <= 1e-11 * n * nrm
        assert np.max(np.abs(A @ Q - Q * lam[None, :])) <= 1e-11 * n * nrm
        assert np.max(np.abs(Q.conj().T @ Q - np.eye(n))) <= 1e-11 * n
        assert abs(lam[0] - 1e-3) < 1e-8 * nrm


def test_degenerate_panels_are_rescued_or_fall_back_to_the_one_stage_reduction(hp, monkeypatch):
    """Panels that CholeskyQR cannot orthogonalise raise a flag.  A diagonal matrix (zero panels) ends in the one-stage panel; a matrix
    with exactly dependent columns goes through at the second attempt (panels perturbed by 1e-13 of their norm); both counters move,
    the results are right."""
    monkeypatch.setenv("NLS_EVD", "twostage")
    ctx = hp.default_context()
    fb, rs = ctx.lib.nls_twostage_fallbacks(ctx.handle), ctx.lib.nls_twostage_rescues(ctx.handle)
    lam, Q = hp.eigh(np.diag(np.arange(1.0, 301.0)))
    assert np.array_equal(lam, np.arange(1.0, 301.0)) and np.allclose(np.abs(Q), np.eye(300), atol=1e-14)
    assert ctx.lib.nls_twostage_fallbacks(ctx.handle) == fb + 1
    rng = np.random.default_rng(1)
    M = rng.standard_normal((260, 3))
    A = M @ M.T  # rank 3: the first panel's 32 columns are exactly dependent
    lam, Q = hp.eigh(A)
    assert np.max(np.abs(lam - np.linalg.eigvalsh(A))) <= 1e-11 * lam[-1] and np.max(np.abs(A @ Q - Q * lam)) <= 1e-11 * lam[-1]
    assert np.max(np.abs(Q.T @ Q - np.eye(260))) <= 1e-12
    assert ctx.lib.nls_twostage_rescues(ctx.handle) + ctx.lib.nls_twostage_fallbacks(ctx.handle) >= rs + fb + 2
    # identity + low rank: the reduction runs out of rank in the middle of a panel (condition number 1e16)
    n = 1400
    M = rng.standard_normal((n, n // 2 + 8))
    A = M @ M.T / n + np.eye(n)
    for bw in ("32", "64"):
        monkeypatch.setenv("NLS_SB_BW", bw)
        lam, Q = hp.eigh(A)
        assert np.max(np.abs(lam - np.linalg.eigvalsh(A))) <= 1e-11 * lam[-1] and np.max(np.abs(A @ Q - Q * lam)) <= 1e-10 * lam[-1]
        assert np.max(np.abs(Q.T @ Q - np.eye(n))) <= 1e-11


@pytest.mark.parametrize("cplx", [False, True])
def test_three_pass_panels_equal_the_adaptive_two_pass_ones(cplx, hp, monkeypatch):
    """Well-conditioned panels skip the second CholeskyQR pass by default; NLS_SB_ADAPTIVE=0 keeps all three: same decomposition to rounding."""
    monkeypatch.setenv("NLS_EVD", "twostage")
    n = 700
    A = _herm(n, cplx, 3)
    lam0 = np.linalg.eigvalsh(A)
    scale = np.max(np.abs(lam0))
    for mode in ("1", "0"):
        monkeypatch.setenv("NLS_SB_ADAPTIVE", mode)
        lam, Q = hp.eigh(A)
        assert np.max(np.abs(lam - lam0)) <= 1e-13 * n * scale
        assert np.max(np.abs(A @ Q - Q * lam[None, :])) <= 1e-13 * n * scale
        assert np.max(np.abs(Q.conj().T @ Q - np.eye(n))) <= 1e-13 * n


@pytest.mark.parametrize("cplx", [False, True])
def test_third_pass_factor_by_series_equals_the_elimination(cplx, hp, monkeypatch):
    """The reconstruction kernel takes the Cholesky factor of the third Gram matrix G3 = I + E from its series when |E| < 1e-8 (always, in
    practice) and by elimination otherwise; NLS_SB_SERIES=0 forces the elimination: same decomposition to rounding, with two and with three passes."""
    monkeypatch.setenv("NLS_EVD", "twostage")
    n = 700
    A = _herm(n, cplx, 5)
    lam0 = np.linalg.eigvalsh(A)
    scale = np.max(np.abs(lam0))
    for adaptive in ("1", "0"):
        monkeypatch.setenv("NLS_SB_ADAPTIVE", adaptive)
        for series in ("1", "0"):
            monkeypatch.setenv("NLS_SB_SERIES", series)
            lam, Q = hp.eigh(A)
            assert np.max(np.abs(lam - lam0)) <= 1e-13 * n * scale
            assert np.max(np.abs(A @ Q - Q * lam[None, :])) <= 1e-13 * n * scale
            assert np.max(np.abs(Q.conj().T @ Q - np.eye(n))) <= 1e-13 * n


def test_fits_through_the_two_stage_reduction_match_the_reference(hp, monkeypatch):
    """The primal and the dual fit with every eigendecomposition forced through the two-stage path: same parity bar as the default."""
    monkeypatch.setenv("NLS_EVD", "twostage")
    g = load_golden("primal_reg_n3000_d20_D256")
    r = hp.primal_fit(g["X"], g["y"], g["s"], g["shift"], g["scale"], g["B"], False)
    assert r["opt"] == int(g["opt"])
    for k in ("beta", "loo_residuals", "loo_leverage", "loo_std", "residuals", "loo_errors_gammas", "lam"):
        assert relerr(r[k], g[k]) < 1e-9, k
    g = load_golden("dual_reg_n1000_d32_w")
    nz = g["nz"]
    y = signed_targets(g)[nz]
    r = hp.dual_fit(g["Xt"], y, g["s"][nz], g["task"] == "clf")
    assert r["opt"] == int(g["opt"])
    for k in ("alpha", "loo_residuals", "loo_std", "residuals", "loo_errors_gammas"):
        assert relerr(r[k], g[k]) < 1e-9, k


@pytest.mark.parametrize("cplx", [False, True])
def test_a_rejected_chase_falls_back_to_the_one_stage_reduction(cplx, hp, monkeypatch):
    """The band -> tridiagonal chase hands data between workgroups inside one launch; its result is checked through the invariants of
    a unitary similarity (trace, Frobenius norm) and a time-out word.  A rejected chase (here: injected) must not fail the
    eigendecomposition: the matrix is restored and the one-stage panel delivers the same decomposition; the fall-back is counted."""
    monkeypatch.setenv("NLS_EVD", "twostage")
    ctx = hp.default_context()
    n = 520
    A = _herm(n, cplx, 5)
    lam0, Q0 = hp.eigh(A)  # through the two-stage reduction; the invariants check passes
    st = ctx.evd_stage_ms()
    assert st is not None and st["kind"].startswith("two-stage") and st["n"] == n and st["total"] > 0
    fb = ctx.lib.nls_twostage_fallbacks(ctx.handle)
    monkeypatch.setenv("NLS_CHASE_INJECT_FAILURE", "1")
    lam, Q = hp.eigh(A)
    assert ctx.lib.nls_twostage_fallbacks(ctx.handle) == fb + 1
    st = ctx.evd_stage_ms()
    assert st["kind"].startswith("one-stage") and set(st) >= {"tridiagonalisation", "stedc", "back_transformation", "total"}
    scale = np.max(np.abs(lam0))
    assert np.max(np.abs(lam - lam0)) <= 1e-13 * n * scale
    assert np.max(np.abs(A @ Q - Q * lam[None, :])) <= 1e-13 * n * scale
    assert np.max(np.abs(Q.conj().T @ Q - np.eye(n))) <= 1e-13 * n
    monkeypatch.delenv("NLS_CHASE_INJECT_FAILURE")
    lam, _ = hp.eigh(A)
    assert ctx.lib.nls_twostage_fallbacks(ctx.handle) == fb + 1 and np.array_equal(lam, lam0)
