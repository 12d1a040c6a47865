"""GPU tests of the eigendecomposition stage (P4 / D2): the three-kernels-per-column tridiagonalisation against
LAPACK's zhetrd / dsytrd (same conventions, so d, e, tau and the reflectors agree to rounding) and the full
eigendecomposition against numpy.linalg.eigh."""
from __future__ import annotations

import numpy as np
import pytest
import scipy.linalg.lapack as lapack

pytestmark = pytest.mark.gpu


@pytest.fixture(scope="module")
def hp():
    import neo_ls_svm_amd as pkg

    pkg.default_context()
    return pkg


def _hermitian(n, cplx, seed, spd=False):
    rng = np.random.default_rng(seed)
    M = rng.standard_normal((n, n)) + (1j * rng.standard_normal((n, n)) if cplx else 0)
    A = M @ M.conj().T / n if spd else (M + M.conj().T) / 2
    return A


@pytest.mark.parametrize("cplx", [True, False])
@pytest.mark.parametrize("n", [1, 2, 3, 31, 32, 33, 64, 65, 130, 257, 700])
def test_tridiagonalisation_matches_lapack(n, cplx, hp):
    A = _hermitian(n, cplx, 100 + n)
    d, e, tau, R = hp.tridiagonalize(A)
    f = lapack.zhetrd if cplx else lapack.dsytrd
    c, d0, e0, tau0, info = f(np.asfortranarray(A), lower=1)
    assert info == 0
    scale = np.max(np.abs(A))
    assert np.max(np.abs(d - d0)) <= 1e-12 * scale * max(n, 8)
    if n > 1:
        assert np.max(np.abs(e - e0)) <= 1e-12 * scale * max(n, 8)
        assert np.max(np.abs(tau - tau0)) <= 1e-11 * max(n, 8)
        il = np.tril_indices(n, -2)
        assert np.max(np.abs(R[il] - c[il]), initial=0.0) <= 1e-11 * max(n, 8)
        assert np.allclose(np.diag(R, -1).real, e) and np.allclose(np.diag(R).real, d)
    # the tridiagonal matrix has the spectrum of A
    T = np.diag(d) + np.diag(e, 1) + np.diag(e, -1)
    assert np.max(np.abs(np.linalg.eigvalsh(T) - np.linalg.eigvalsh(A))) <= 1e-12 * scale * max(n, 8)


def test_tridiagonalisation_reads_only_the_lower_triangle(hp):
    A = _hermitian(97, True, 5)
    junk = np.triu(np.full_like(A, 7.0 + 3.0j), 1)
    d, e, _, _ = hp.tridiagonalize(np.tril(A) + junk)
    d0, e0, _, _ = hp.tridiagonalize(A)
    assert np.array_equal(d, d0) and np.array_equal(e, e0)


def test_tridiagonalisation_is_bit_reproducible(hp):
    A = _hermitian(300, True, 6)
    r1, r2 = hp.tridiagonalize(A), hp.tridiagonalize(A)
    assert all(np.array_equal(a, b) for a, b in zip(r1, r2))


def test_identity_and_diagonal_inputs(hp):
    """Columns that are already reduced take the tau = 0 branch of larfg."""
    d, e, tau, _ = hp.tridiagonalize(np.diag(np.arange(1.0, 41.0)))
    assert np.array_equal(d, np.arange(1.0, 41.0)) and not e.any() and not tau.any()
    lam, Q = hp.eigh(np.eye(50, dtype=np.complex128))
    assert np.allclose(lam, 1.0) and np.allclose(Q.conj().T @ Q, np.eye(50), atol=1e-13)


@pytest.mark.parametrize("cplx", [True, False])
@pytest.mark.parametrize("n", [5, 64, 257, 1025])
def test_eigh_matches_numpy(n, cplx, hp):
    A = _hermitian(n, cplx, 200 + n, spd=True)
    lam, Q = hp.eigh(A)
    lam0 = np.linalg.eigvalsh(A)
    scale = lam0[-1]
    assert np.all(np.diff(lam) >= 0)
    assert np.max(np.abs(lam - lam0)) <= 1e-12 * scale * max(n, 8)
    assert np.max(np.abs(Q.conj().T @ Q - np.eye(n))) <= 1e-12 * max(n, 8)
    assert np.max(np.abs(A @ Q - Q * lam[None, :])) <= 1e-12 * scale * max(n, 8)


def test_eigh_rocsolver_fallback_agrees(hp, monkeypatch):
    A = _hermitian(200, True, 9, spd=True)
    lam, _ = hp.eigh(A)
    monkeypatch.setenv("NLS_EVD", "rocsolver")
    lam_r, Qr = hp.eigh(A)
    assert np.max(np.abs(lam - lam_r)) <= 1e-12 * lam[-1] * 200
    assert np.max(np.abs(A @ Qr - Qr * lam_r[None, :])) <= 1e-12 * lam[-1] * 200


def test_argument_errors(hp):
    with pytest.raises(ValueError):
        hp.tridiagonalize(np.zeros((3, 4)))
    with pytest.raises(ValueError):
        hp.eigh(np.zeros((0, 0)))


@pytest.mark.parametrize("kernels", ["2", "3"])
@pytest.mark.parametrize("cplx", [True, False])
def test_both_panel_variants_match_lapack(kernels, cplx, hp, monkeypatch):
    """Two kernels per column (default up to n = 6144) and three (above): same reduction, same LAPACK conventions."""
    monkeypatch.setenv("NLS_TRD_KERNELS", kernels)
    for n in (33, 200, 517):
        A = _hermitian(n, cplx, 300 + n)
        d, e, tau, _ = hp.tridiagonalize(A)
        f = lapack.zhetrd if cplx else lapack.dsytrd
        _, d0, e0, tau0, info = f(np.asfortranarray(A), lower=1)
        scale = np.max(np.abs(A))
        assert info == 0
        assert np.max(np.abs(d - d0)) <= 1e-12 * scale * n and np.max(np.abs(e - e0)) <= 1e-12 * scale * n
        assert np.max(np.abs(tau - tau0)) <= 1e-11 * n
    lam, Q = hp.eigh(_hermitian(300, cplx, 77, spd=True))
    assert np.max(np.abs(Q.conj().T @ Q - np.eye(300))) <= 1e-11


def test_large_n_configuration_on_a_small_matrix(hp, monkeypatch):
    """Above n = 6144 the panel runs three kernels per column with four row groups per dot block; force that here."""
    monkeypatch.setenv("NLS_TRD_KERNELS", "3")
    monkeypatch.setenv("NLS_TRD_DOTGROUPS", "4")
    for cplx in (True, False):
        A = _hermitian(517, cplx, 41)
        d, e, tau, _ = hp.tridiagonalize(A)
        f = lapack.zhetrd if cplx else lapack.dsytrd
        _, d0, e0, tau0, _ = f(np.asfortranarray(A), lower=1)
        assert np.max(np.abs(d - d0)) <= 1e-12 * 517 * np.max(np.abs(A)) and np.max(np.abs(e - e0)) <= 1e-12 * 517 * np.max(np.abs(A))
        assert np.max(np.abs(tau - tau0)) <= 1e-11 * 517


def test_rank2k_rocblas_knob(hp, monkeypatch):
    A = _hermitian(260, True, 8)
    d0, e0, _, _ = hp.tridiagonalize(A)
    monkeypatch.setenv("NLS_TRD_RANK2K", "rocblas")
    d1, e1, _, _ = hp.tridiagonalize(A)
    assert np.max(np.abs(d0 - d1)) <= 1e-12 * 260 and np.max(np.abs(e0 - e1)) <= 1e-12 * 260
