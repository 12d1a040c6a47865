"""The library's own divide-and-conquer eigensolver for symmetric tridiagonal matrices (csrc/nls_stedc.h; the tridiagonal stage of the
eigendecompositions at ``_neo_ls_svm.py:120`` and ``:265``) against numpy on the families of matrices that stress its parts: deflation of both
kinds, the secular solver next to poles, leaf / level edges, exact ties."""

from __future__ import annotations

import numpy as np
import pytest

pytestmark = pytest.mark.gpu


@pytest.fixture(scope="module")
def hp():
    import neo_ls_svm_amd as pkg

    pkg.default_context()
    return pkg


def _check(hp, d, e, tol=2e-14):
    n = d.size
    lam, Q = hp.stedc(d, e)
    T = np.diag(d) + np.diag(e, 1) + np.diag(e, -1)
    ref = np.linalg.eigvalsh(T)
    nrm = max(np.max(np.abs(ref)), 1e-300)
    assert np.all(np.diff(lam) >= 0)
    assert np.max(np.abs(lam - ref)) <= tol * n * nrm
    assert np.max(np.abs(T @ Q - Q * lam[None, :])) <= tol * n * nrm
    assert np.max(np.abs(Q.T @ Q - np.eye(n))) <= tol * n
    return lam, Q


@pytest.mark.parametrize("n", [1, 2, 3, 5, 31, 32, 33, 63, 64, 65, 100, 128, 129, 257, 511, 512, 513, 700, 1025, 2500])
def test_random_tridiagonal(n, hp):
    rng = np.random.default_rng(n)
    _check(hp, rng.standard_normal(n), rng.standard_normal(max(n - 1, 0)))


def test_structured_families(hp):
    rng = np.random.default_rng(1)
    lam, Q = _check(hp, np.arange(1.0, 201.0), np.zeros(199))  # diagonal: every merge deflates everything
    assert np.array_equal(lam, np.arange(1.0, 201.0)) and np.allclose(np.abs(Q), np.eye(200), atol=0)
    _check(hp, np.full(300, 2.0), np.ones(299))  # Toeplitz (2, 1): z components decay to zero at the block edges
    wd = np.abs(np.arange(-50, 51)).astype(float)
    _check(hp, wd, np.ones(100))  # Wilkinson W101: pairs of eigenvalues agreeing to 1e-14
    glued = np.concatenate([wd] * 4)
    ge = np.concatenate([np.ones(100), [1e-8], np.ones(100), [1e-8], np.ones(100), [1e-8], np.ones(100)])
    _check(hp, glued, ge)  # glued Wilkinson: clusters of four
    _check(hp, np.ones(257), 1e-9 * rng.standard_normal(256))  # identity + tiny couplings: close poles rotated into one
    _check(hp, np.ones(300), np.zeros(299))  # exact ties everywhere
    _check(hp, 10.0 ** np.linspace(0, -14, 400), 10.0 ** np.linspace(-1, -15, 399))  # graded over 14 decades
    _check(hp, np.zeros(129), np.ones(128))  # zero diagonal: symmetric spectrum


def test_tridiagonalised_rbf_kernel(hp):
    """The dual path's spectrum: a few large eigenvalues, most of them clustered near the bottom (heavy deflation)."""
    import scipy.linalg as sla

    rng = np.random.default_rng(2)
    X = rng.standard_normal((900, 8)) * 0.4
    K = np.exp(-0.5 * ((X[:, None, :] - X[None, :, :]) ** 2).sum(-1)) + 1.0
    H = sla.hessenberg(K)
    _check(hp, np.diag(H).copy(), np.diag(H, 1).copy())


def test_rocsolver_variant_agrees(hp, monkeypatch):
    rng = np.random.default_rng(3)
    d, e = rng.standard_normal(700), rng.standard_normal(699)
    lam, _ = hp.stedc(d, e)
    monkeypatch.setenv("NLS_STEDC", "rocsolver")
    lam2, Q2 = hp.stedc(d, e)
    assert np.max(np.abs(lam - lam2)) <= 1e-13 * np.max(np.abs(lam))
    assert np.max(np.abs(Q2.T @ Q2 - np.eye(700))) <= 1e-11
