"""The compressed gamma sweep (csrc/nls_lib.hip: sweep_compression) is an identity to rounding: for the reference's
1024-point grid (``_neo_ls_svm.py:146``) 1 / (gamma_g + lam) = sum_q W[q, g] / (node_q + lam) for every lam >= 0.  Host
arithmetic only (the library's own routine through its C ABI test hook): runs without a GPU."""
from __future__ import annotations

import ctypes as C

import numpy as np
import pytest

import neolssvm_oracle as orc


def weights(gammas):
    from neo_ls_svm_amd import _lib

    lib = _lib.load_library()
    g = np.ascontiguousarray(gammas, dtype=np.float64)
    nodes, W, ok = np.empty(128), np.empty((128, g.size)), C.c_int()
    assert lib.nls_sweep_weights(g.ctypes.data, g.size, nodes.ctypes.data, W.ctypes.data, C.byref(ok)) == 0
    return (nodes, W) if ok.value else None


def test_reference_grid_is_reproduced_to_rounding():
    gam = orc.gamma_grid(1024)
    nodes, W = weights(gam)
    assert np.all(nodes > 0) and nodes.min() >= gam[0] * (1 - 1e-12) and nodes.max() <= gam[-1] * (1 + 1e-12)
    assert np.abs(W).sum(0).max() < 3.5  # Lebesgue sum: rounding errors are not amplified
    assert np.allclose(W.sum(0), 1.0, atol=1e-13)  # constants are reproduced
    lam = np.concatenate([[0.0, -1e-12, -1e-9, 1e-300], np.logspace(-16, 8, 3000)])
    R = 1.0 / (gam[None, :] + lam[:, None])
    Rn = 1.0 / (nodes[None, :] + lam[:, None])
    assert np.max(np.abs(Rn @ W - R) / R) < 1e-14
    # rounding-negative eigenvalues: exact to rounding down to -gamma_min / 8, which is the library's threshold for keeping the
    # compressed products (nls_lib.hip); beyond it the pole approaches the first piece and the direct products are taken
    neg = -gam[0] * np.array([1e-6, 1e-3, 1 / 64, 1 / 16, 1 / 8])
    Rneg = 1.0 / (gam[None, :] + neg[:, None])
    assert np.max(np.abs((1.0 / (nodes[None, :] + neg[:, None])) @ W - Rneg) / Rneg) < 1e-14
    half = np.array([-gam[0] / 2])
    Rh = 1.0 / (gam[None, :] + half[:, None])
    assert 1e-12 < np.max(np.abs((1.0 / (nodes[None, :] + half[:, None])) @ W - Rh) / Rh) < 1e-8  # why the threshold is not 1/2


@pytest.mark.parametrize("name", ["primal_reg_n3000_d20_D256", "primal_clf_n3000_d16_D256_wz"])
def test_compressed_sweep_reproduces_the_loo_curve(name, golden_loader):
    """End to end on a fixture: the per-gamma LOO errors through the compressed products equal the direct ones."""
    from conftest import relerr, signed_targets

    g = golden_loader(name)
    y, is_clf = signed_targets(g), g["task"] == "clf"
    phi = orc.feature_map(g["X"], g["shift"], g["scale"], g["B"])
    A, b, sn = orc.primal_gram(phi, y, g["s"])
    c = 1.0 / phi.size
    lam, Q = np.linalg.eigh(A / c)
    P = phi @ Q
    U, Gm = np.real(P * ((Q.conj().T @ b) / c)[None, :]), P.real**2 + P.imag**2
    gam = g["gammas"]
    nodes, W = weights(gam)
    Rn = 1.0 / (nodes[None, :] + lam[:, None])
    e = ((U @ Rn) @ W - y[:, None]) / (1 - (sn[:, None] ** 2) * ((Gm @ Rn) @ W) / c)
    if is_clf:
        e = orc.clip_classifier_residuals(e, y)
    assert relerr(sn @ np.abs(e), g["loo_errors_gammas"]) < 1e-12
    assert relerr(e[:, int(g["opt"])], g["loo_residuals"]) < 1e-11


def test_grids_that_take_the_direct_product():
    assert weights(orc.gamma_grid(128)) is None  # the dual path's grid / the 32-point sub-grid: too short to gain
    assert weights(orc.gamma_grid(1024)[::-1]) is None  # not increasing
    g = orc.gamma_grid(1024).copy()
    g[5] = -1.0
    assert weights(g) is None  # non-positive entry
    assert weights(np.logspace(-9, 2, 1024)) is None  # spans more than e^17.2: pieces longer than the validated 4.3
    lin = np.linspace(1e-3, 5.0, 700)  # an increasing positive grid that is not log-spaced (pieces by value, not by index)
    nodes, W = weights(lin)
    lam = np.logspace(-8, 3, 500)
    R = 1.0 / (lin[None, :] + lam[:, None])
    assert np.max(np.abs((1.0 / (nodes[None, :] + lam[:, None])) @ W - R) / R) < 1e-13
