"""The package's own affine pre-step vs the shift_/scale_/A_ the reference produced (CPU only)."""

from __future__ import annotations

import numpy as np
import pytest
from conftest import DUAL_CASES, PRIMAL_CASES, relerr, signed_targets

from neo_ls_svm_amd import _prestep


@pytest.mark.parametrize("name", PRIMAL_CASES + DUAL_CASES)
def test_separator_matches_reference(name, golden_loader):
    g = golden_loader(name)
    X, y, s = g["X"], signed_targets(g), g["s"]
    if g["kind"] == "dual":  # the reference drops zero-weight rows first (_neo_ls_svm.py:388-389)
        nz = g["nz"]
        X, y, s = X[nz], y[nz], s[nz]
    sw = s if bool(g["has_weights"]) else None
    shift, scale, A = _prestep.fit_affine_separator(X, y, sw)
    assert relerr(np.ravel(shift), g["shift"]) < 1e-12
    assert relerr(np.ravel(scale), g["scale"]) < 1e-12
    assert A.shape == g["A_sep"].shape
    assert relerr(A, g["A_sep"]) < 1e-9


def test_target_bins_regression_and_classes():
    rng = np.random.default_rng(0)
    y = rng.standard_normal(5000)
    b = _prestep.target_bins(y)
    assert b.min() == 0 and 4 <= b.max() + 1 <= 64
    assert np.all(np.diff(b[np.argsort(y)]) >= 0)  # bins are monotone in y
    counts = np.bincount(b)
    assert counts.sum() == 5000 and counts.max() <= 0.3 * 5000  # the merged centre bin may exceed max_bin_size
    assert np.array_equal(_prestep.target_bins(np.array([3.0, 1.0, 3.0, 1.0, 1.0])), np.array([1, 0, 1, 0, 0]))


def test_weighted_median_toy():
    # a = (0, 1, 1), w = (2, 1, 1): the reference's definition gives 0.5 (_weighted_quantile.py:66-68)
    med = _prestep.weighted_median_columns(np.array([[0.0], [1.0], [1.0]]), np.array([2.0, 1.0, 1.0]))
    assert med.shape == (1, 1) and abs(med[0, 0] - 0.5) < 1e-15
