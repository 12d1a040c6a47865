"""Pin the CPU oracle to fixtures captured from the imported reference (CPU only, no GPU)."""

from __future__ import annotations

import numpy as np
import pytest
from conftest import DUAL_CASES, PLUGIN_CASES, PRIMAL_CASES, relerr, signed_targets

import neolssvm_oracle as orc

TOL = 1e-8  # oracle vs reference, float64 both sides; the HIP parity bar is 1e-5


@pytest.mark.parametrize("name", PRIMAL_CASES)
def test_orf_frequencies_and_fold_bitwise(name, golden_loader):
    g = golden_loader(name)
    kind = PLUGIN_CASES.get(name)
    A_sep = None if kind == "orf_normalizer" else g["A_sep"]  # an AffineNormalizer has no matrix: Z acts on the d inputs themselves
    d_in = g["X"].shape[1] if A_sep is None else A_sep.shape[1]
    Z = (orc.rff_frequencies if kind == "rff" else orc.orf_frequencies)(d_in, int(g["D"]), seed=42)
    assert np.array_equal(Z, g["Z"])
    assert np.array_equal(orc.fold_projection(A_sep, Z), g["B"])


@pytest.mark.parametrize("name", PRIMAL_CASES)
def test_feature_map_matches_reference_transform(name, golden_loader):
    g = golden_loader(name)
    phi = orc.feature_map(g["Xq"][:8], g["shift"], g["scale"], g["B"])
    assert relerr(phi, g["phi_q"]) < 1e-14


@pytest.mark.parametrize("name", PRIMAL_CASES)
@pytest.mark.parametrize("schedule", ["faithful", "streamed"])
def test_primal_fit_matches_reference(name, schedule, golden_loader):
    g = golden_loader(name)
    y, is_clf = signed_targets(g), g["task"] == "clf"
    if schedule == "faithful":
        phi = orc.feature_map(g["X"], g["shift"], g["scale"], g["B"])
        r = orc.primal_fit_faithful(phi, y, g["s"], is_clf)
    else:
        r = orc.primal_fit_streamed(g["X"], y, g["s"], g["shift"], g["scale"], g["B"], is_clf, row_tile=1024)
    assert np.array_equal(r["gammas"], g["gammas"])
    assert relerr(r["lam"], g["lam"]) < TOL
    assert relerr(r["loo_errors_gammas"], g["loo_errors_gammas"]) < TOL
    assert r["opt"] == int(g["opt"])
    assert r["gamma"] == float(g["gamma"])
    assert relerr(r["beta"], g["beta"]) < TOL
    assert relerr(r["loo_residuals"], g["loo_residuals"]) < TOL
    assert relerr(r["loo_yhat"], g["loo_yhat"]) < TOL
    assert relerr(r["loo_leverage"], g["loo_leverage"]) < TOL
    assert relerr(r["loo_std"], g["loo_std"]) < 1e-7
    assert relerr(r["residuals"], g["residuals"]) < TOL
    assert abs(r["loo_error"] - float(g["loo_error"])) < TOL * abs(float(g["loo_error"]))
    assert abs(r["loo_score"] - float(g["loo_score"])) < 1e-10
    assert r["L_lower"] == bool(g["L_lower"])


def test_primal_intermediates_match_reference(golden_loader):
    g = golden_loader("primal_reg_n3000_d20_D256")
    phi = orc.feature_map(g["X"], g["shift"], g["scale"], g["B"])
    A, b, sn = orc.primal_gram(phi, g["y"], g["s"])
    assert relerr(A * phi.size, g["A_over_c"]) < 1e-12
    r = orc.primal_fit_faithful(phi, g["y"], g["s"], False)
    assert relerr(r["L"][np.triu_indices(A.shape[0])], g["L"][np.triu_indices(A.shape[0])]) < TOL


@pytest.mark.parametrize("name", PRIMAL_CASES)
def test_primal_inference_matches_reference(name, golden_loader):
    g = golden_loader(name)
    yq = orc.primal_decision_function(g["Xq"], g["shift"], g["scale"], g["B"], g["beta"])
    assert relerr(yq, g["decision_function"]) < 1e-10
    y, is_clf = signed_targets(g), g["task"] == "clf"
    r = orc.primal_fit_streamed(g["X"], y, g["s"], g["shift"], g["scale"], g["B"], is_clf)
    sq = orc.primal_predict_std(g["Xq"], g["shift"], g["scale"], g["B"], r["L"], r["L_lower"])
    assert relerr(sq, g["predict_std"]) < 1e-7


@pytest.mark.parametrize("name", DUAL_CASES)
@pytest.mark.parametrize("schedule", ["faithful", "reduced"])
def test_dual_fit_matches_reference(name, schedule, golden_loader):
    g = golden_loader(name)
    if schedule == "faithful" and g["Xt"].shape[0] > 600:
        pytest.skip("n x G x n tensor kept small in the CPU suite")
    nz = g["nz"]
    y, is_clf = signed_targets(g)[nz], g["task"] == "clf"
    fit = orc.dual_fit_faithful if schedule == "faithful" else orc.dual_fit_reduced
    r = fit(g["Xt"], y, g["s"][nz], is_clf)
    assert np.array_equal(r["gammas"], g["gammas"])
    assert relerr(r["loo_errors_gammas"], g["loo_errors_gammas"]) < 1e-7
    assert r["opt"] == int(g["opt"])
    assert relerr(r["alpha"], g["alpha"]) < 1e-7
    assert relerr(r["loo_residuals"], g["loo_residuals"]) < 1e-7
    assert relerr(r["residuals"], g["residuals"]) < 1e-7
    assert relerr(r["loo_std"], g["loo_std"]) < 1e-6
    assert abs(r["loo_score"] - float(g["loo_score"])) < 1e-10
    yq = orc.dual_decision_function(g["Xqt"], g["Xt"], r["alpha"])
    assert relerr(yq, g["decision_function"]) < 1e-7
    sq = orc.dual_predict_std(g["Xqt"], g["Xt"], r["L"], r["L_lower"])
    assert relerr(sq, g["predict_std"]) < 1e-6


def test_rbf_gram_equals_sklearn():
    from sklearn.metrics.pairwise import rbf_kernel

    rng = np.random.default_rng(3)
    X, Y = rng.standard_normal((70, 9)), rng.standard_normal((31, 9))
    assert np.array_equal(orc.rbf_gram(X), rbf_kernel(X, gamma=0.5))
    assert np.allclose(orc.rbf_gram(X, Y), rbf_kernel(X, Y, gamma=0.5), rtol=0, atol=1e-15)


def test_sigma_grid_matches_reference(golden_loader):
    sg = golden_loader("sigma_grid_reg_n3000")
    g = golden_loader(sg["base"])
    gam = orc.gamma_grid(1024)[::33]
    assert gam.size == 32
    for k, sigma in enumerate(sg["sigmas"]):
        r = orc.primal_fit_streamed(g["X"], g["y"], g["s"], g["shift"], g["scale"], g["B"] / sigma, False, gammas=gam)
        assert relerr(r["loo_errors_gammas"], sg["loo_errors"][k]) < TOL


def test_ames_shaped_case_pins_prestep_and_oracle(golden_loader):
    """BASELINE config 1 (plumbing): wide d with one-hot, all-zero and constant columns.  The package's pre-step
    reproduces the reference's shift_/scale_/A_ (incl. the eps clamp, ``_affine_normalizer.py:88``) and the oracle its fit."""
    from neo_ls_svm_amd import _prestep, hotpath

    g = golden_loader("primal_reg_ames_n2930_d301_D512")
    shift, scale, A = _prestep.fit_affine_separator(g["X"], g["y"], None)
    assert relerr(np.ravel(shift), g["shift"]) < 1e-12 and relerr(np.ravel(scale), g["scale"]) < 1e-12
    assert np.sum(np.abs(np.ravel(scale)) == np.finfo(np.float64).eps) == 18
    assert A.shape[1] == int(g["Z_shape"][0])
    B = A @ hotpath.orf_frequencies(A.shape[1], 512, 42)
    assert relerr(B, g["B"]) < 1e-9
    r = orc.primal_fit_streamed(g["X"], g["y"], g["s"], g["shift"], g["scale"], g["B"], False)
    assert r["opt"] == int(g["opt"]) and r["gamma"] == float(g["gamma"])
    for k in ("lam", "loo_errors_gammas", "beta", "loo_residuals", "loo_leverage", "residuals"):
        assert relerr(r[k], g[k]) < TOL, k
    assert relerr(r["loo_std"], g["loo_std"]) < 1e-7
    assert relerr(orc.primal_decision_function(g["Xq"], g["shift"], g["scale"], g["B"], r["beta"]), g["decision_function"]) < TOL


EXACT_C_CASES = ["primal_reg_n400_d8_D192_exactC", "primal_clf_n300_d6_D128_exactC"]


@pytest.mark.parametrize("name", EXACT_C_CASES)
def test_exact_complexity_matrix_and_generalised_branch(name, golden_loader):
    """the oracle's restatement of the exact complexity matrix (``_feature_maps.py:46-55``) and of the
    generalised-EVD branch (``_neo_ls_svm.py:122-124,131,139``) against fixtures from the reference run with
    ``fast_approx=False``; the product's host-side matrix builder agrees with both."""
    from neo_ls_svm_amd import hotpath

    g = golden_loader(name)
    Cm = orc.exact_complexity_matrix(g["Z"])
    assert relerr(Cm, g["C"]) < 1e-13
    assert relerr(hotpath.exact_complexity_matrix(g["Z"]), g["C"]) < 1e-13
    y, is_clf = signed_targets(g), g["task"] == "clf"
    phi = orc.feature_map(g["X"], g["shift"], g["scale"], g["B"])
    r = orc.primal_fit_faithful(phi, y, g["s"], is_clf, C=g["C"])
    assert r["opt"] == int(g["opt"]) and r["gamma"] == float(g["gamma"])
    assert relerr(r["loo_errors_gammas"], g["loo_errors_gammas"]) < TOL
    for k in ("beta", "loo_residuals", "loo_leverage", "residuals", "loo_yhat"):
        assert relerr(r[k], g[k]) < TOL, k
    assert relerr(r["loo_std"], g["loo_std"]) < 1e-7
    assert abs(r["loo_score"] - float(g["loo_score"])) < 1e-10
    iu = np.triu_indices(phi.shape[1])
    assert relerr(r["L"][iu], g["L"][iu]) < TOL
