"""CPU checks of tools/twostage_proto.py - the NumPy statement of the two-stage tridiagonalisation (band reduction by CholeskyQR3 +
Householder reconstruction, bulge chase, diamond-blocked second back-transformation) whose pieces the GPU tests use as their checker
(tests/test_gpu_twostage.py compares the HIP stages with `apply_q2_naive` and with numpy.linalg.eigh)."""
import sys
from pathlib import Path

import numpy as np
import pytest

sys.path.insert(0, str(Path(__file__).resolve().parent.parent / "tools"))
import twostage_proto as tp  # noqa: E402


def _herm(n, cplx, seed):
    rng = np.random.default_rng(seed)
    M = rng.standard_normal((n, n)) + (1j * rng.standard_normal((n, n)) if cplx else 0)
    return (M + M.conj().T) / 2


@pytest.mark.parametrize("cplx", [False, True])
@pytest.mark.parametrize("n,b,g", [(23, 4, 2), (41, 8, 3), (64, 8, 1), (70, 16, 2)])
def test_prototype_eigendecomposition_matches_numpy(n, b, g, cplx):
    A = _herm(n, cplx, n + b)
    lam, Q, _ = tp.eigh_two_stage(A, b, g)
    lam0 = np.linalg.eigvalsh(A)
    scale = np.max(np.abs(lam0))
    assert np.max(np.abs(lam - lam0)) <= 1e-12 * n * scale
    assert np.max(np.abs(A @ Q - Q * lam[None, :])) <= 1e-12 * n * scale
    assert np.max(np.abs(Q.conj().T @ Q - np.eye(n))) <= 1e-12 * n


@pytest.mark.parametrize("cplx", [False, True])
def test_diamond_blocked_back_transformation_equals_the_reflector_product(cplx):
    n, b = 45, 8
    A = _herm(n, cplx, 7)
    Ab, panels = tp.sy2sb(A.copy(), b)
    d, e, V2 = tp.sb2st(Ab, b)
    T = np.diag(d) + np.diag(e, 1) + np.diag(e, -1)
    assert np.max(np.abs(np.linalg.eigvalsh(T) - np.linalg.eigvalsh(A))) <= 1e-12 * n * np.max(np.abs(A))
    rng = np.random.default_rng(3)
    C = (rng.standard_normal((n, 11)) + (1j * rng.standard_normal((n, 11)) if cplx else 0)).astype(A.dtype)
    ref = tp.apply_q2_naive(V2, b, C.copy())
    for g in (1, 2, 4):
        assert np.max(np.abs(tp.apply_q2_diamond(V2, b, g, C.copy()) - ref)) <= 1e-13 * n
