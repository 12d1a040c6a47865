"""The package's own affine pre-step vs the shift_/scale_/A_ the reference produced (CPU only)."""

from __future__ import annotations

import numpy as np
import pytest
from conftest import DUAL_CASES, PLUGIN_CASES, PRIMAL_CASES, relerr, signed_targets

from neo_ls_svm_amd import _prestep


@pytest.mark.parametrize("name", PRIMAL_CASES + DUAL_CASES)
def test_separator_matches_reference(name, golden_loader):
    g = golden_loader(name)
    X, y, s = g["X"], signed_targets(g), g["s"]
    if g["kind"] == "dual":  # the reference drops zero-weight rows first (_neo_ls_svm.py:388-389)
        nz = g["nz"]
        X, y, s = X[nz], y[nz], s[nz]
    sw = s if bool(g["has_weights"]) else None
    if PLUGIN_CASES.get(name) == "orf_normalizer":  # the fixture's affine map is an AffineNormalizer: shift / scale only, A = None
        shift, scale = _prestep.fit_affine_normalizer(X, y, sw)
        assert relerr(np.ravel(shift), g["shift"]) < 1e-12 and relerr(np.ravel(scale), g["scale"]) < 1e-12
        assert g["B"].shape == (X.shape[1], int(g["D"])) and np.array_equal(g["B"], g["Z"])  # B = Z: nothing to fold
        return
    shift, scale, A = _prestep.fit_affine_separator(X, y, sw)
    assert relerr(np.ravel(shift), g["shift"]) < 1e-12
    assert relerr(np.ravel(scale), g["scale"]) < 1e-12
    assert A.shape == g["A_sep"].shape
    assert relerr(A, g["A_sep"]) < 1e-9


def test_plain_rff_frequencies_are_the_reference_stream(golden_loader):
    """RandomFourierFeatures (``_feature_maps.py:120-127``): Z = RandomState(42).randn(d', D), no QR, no chi rescaling; B = A Z."""
    g = golden_loader("primal_reg_n2000_d12_RFF256")
    Z = np.random.RandomState(42).randn(*g["Z"].shape)
    assert np.array_equal(Z, g["Z"])
    assert relerr(g["A_sep"] @ Z, g["B"]) < 1e-13


def test_foreign_feature_maps_are_translated_or_refused():
    """The primal / dual plug-in points (``_neo_ls_svm.py:62-75,380-394``): upstream's classes are translated by name and public
    parameters, own classes pass, anything else raises TypeError - never a silent substitution."""
    from neo_ls_svm_amd import estimator as est

    class OrthogonalRandomFourierFeatures:  # stands in for upstream's class (same name, public parameters)
        def __init__(self):
            self.num_features, self.random_state, self.affine_feature_map = 192, 7, None

    class RandomFourierFeatures(OrthogonalRandomFourierFeatures):
        pass

    fm = est._as_own_feature_map(OrthogonalRandomFourierFeatures())
    assert type(fm) is est.OrthogonalRandomFourierFeatures and (fm.num_features, fm.random_state) == (192, 7)
    fm = est._as_own_feature_map(RandomFourierFeatures())
    assert type(fm) is est.RandomFourierFeatures and fm.orthogonal is None and not fm.orthogonal_default
    own = est.RandomFourierFeatures(num_features=64, orthogonal=True)
    assert est._as_own_feature_map(own) is own

    class Nystroem:
        num_features = 10

    with pytest.raises(TypeError, match="primal_feature_map"):
        est._as_own_feature_map(Nystroem())
    with pytest.raises(TypeError, match="not an affine map"):
        est._as_own_affine_map(object())

    class MyAffine:  # a caller's affine map: fit + shift / scale / A attributes
        def fit(self, X, y=None, sample_weight=None):
            self.shift_, self.scale_, self.A_ = X.mean(0), X.std(0), None
            return self

    a = MyAffine()
    assert est._as_own_affine_map(a) is a
    X = np.random.default_rng(0).standard_normal((50, 3))
    fitted = est._fit_affine(a, X, None, None, None)
    sh, sc, A = est._affine_params(fitted)
    assert np.allclose(sh, X.mean(0)) and A is None

    class NotAffine:
        def fit(self, X, y=None, sample_weight=None):
            return self

    with pytest.raises(TypeError, match="exposes no shift"):
        est._fit_affine(NotAffine(), X, None, None, None)
    fixed = est.AffineFeatureMap(scale=np.array([1.0, 2.0, 4.0]), shift=np.zeros(3), A=np.eye(3)[:, :2])
    assert np.allclose(fixed.fit(X).transform(X), (X / [1.0, 2.0, 4.0])[:, :2])
    with pytest.raises(ValueError):
        est.AffineFeatureMap(scale=np.array([1.0, 0.0, 4.0]), shift=np.zeros(3)).fit(X)


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


def test_affine_feature_map_surface_outside_the_hot_path():
    """``append_features`` / ``inverse_transform`` / ``get_feature_names_out`` of the reference's ``AffineFeatureMap``
    (``_affine_feature_map.py:26-38,90-136``), mirrored on the host: appended columns come first and invert exactly; without a matrix the map
    inverts exactly; with one, through its pseudo-inverse; names as upstream spells them.  A random-feature map refuses an appending map."""
    from neo_ls_svm_amd import AffineFeatureMap, AffineSeparator, RandomFourierFeatures

    rng = np.random.default_rng(2)
    X = rng.standard_normal((40, 5))
    shift, scale = rng.standard_normal(5), rng.uniform(0.5, 2.0, 5)
    plain = AffineFeatureMap(scale=scale, shift=shift).fit(X)
    T = plain.transform(X)
    assert np.allclose(T, (X - shift) / scale) and np.allclose(plain.inverse_transform(T), X, atol=1e-13)
    assert list(plain.get_feature_names_out(list("abcde"))) == [f"{c}_shifted_scaled" for c in "abcde"]
    A = rng.standard_normal((5, 7))  # wider than tall: full row rank, the pseudo-inverse undoes it
    wide = AffineFeatureMap(scale=scale, shift=shift, A=A).fit(X)
    Tw = wide.transform(X)
    assert Tw.shape == (40, 7) and np.allclose(Tw, ((X - shift) / scale) @ A)
    assert np.allclose(wide.inverse_transform(Tw), X, atol=1e-10)
    assert list(wide.get_feature_names_out(list("abcde"))) == ["a,b,c,d,e_affine_map"] * 7
    app = AffineFeatureMap(scale=scale, shift=shift, A=A, append_features=True).fit(X)
    Ta = app.transform(X)
    assert Ta.shape == (40, 12) and np.array_equal(Ta[:, :5], X) and np.allclose(Ta[:, 5:], Tw)
    assert np.array_equal(app.inverse_transform(Ta), X)
    assert list(app.get_feature_names_out(list("abcde"))) == list("abcde") + ["a,b,c,d,e_affine_map"] * 7
    tall = AffineFeatureMap(scale=scale, shift=shift, A=A[:, :3]).fit(X)  # the memory-order switch of the reference (A.shape[1] < A.shape[0])
    assert np.allclose(tall.transform(X), ((X - shift) / scale) @ A[:, :3])
    # sklearn plumbing: the parameter round-trips through get_params / clone
    from sklearn.base import clone

    assert clone(AffineSeparator(append_features=True)).append_features is True
    with pytest.raises(TypeError, match="append_features"):
        RandomFourierFeatures(affine_feature_map=app, num_features=8).fit(X, rng.standard_normal(40))
