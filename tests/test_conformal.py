"""Conformal quantile layer against fixtures captured from the reference
(tests/golden/make_golden_conformal.py).  CPU part: the LP restatement and the two-level calibration fed with the
reference's own calibration split; GPU part: NeoLSSVM.predict_quantiles / predict_interval end to end."""
from __future__ import annotations

import importlib.util
import sys
import types
from pathlib import Path

import numpy as np
import pytest

ROOT = Path(__file__).resolve().parent.parent
GOLD = Path(__file__).resolve().parent / "golden"


def _conformal_module():
    """neo_ls_svm_amd.conformal without importing the package (which would load the HIP library)."""
    name = "_nls_conformal_standalone"
    if name in sys.modules:
        return sys.modules[name]
    spec = importlib.util.spec_from_file_location(name, ROOT / "neo_ls_svm_amd" / "conformal.py")
    mod = importlib.util.module_from_spec(spec)
    sys.modules[name] = mod
    spec.loader.exec_module(mod)
    return mod


def test_coherent_quantile_lp_matches_reference():
    c = _conformal_module()
    g = np.load(GOLD / "conformal_lp_n400.npz")
    r = c.CoherentLinearQuantileRegressor(quantiles=tuple(g["quantiles"])).fit(g["Xc"], g["yc"], sample_weight=g["w"].copy())
    assert r.β_.shape == g["beta"].shape and r.β_full_.shape == g["beta_full"].shape
    np.testing.assert_allclose(r.β_full_, g["beta_full"], rtol=1e-7, atol=1e-9)
    np.testing.assert_allclose(r.predict(g["Xn"]), g["pred"], rtol=1e-7, atol=1e-9)
    np.testing.assert_allclose(r.intercept_clip(g["Xc"], g["yc"]), g["clip"], rtol=1e-6, atol=1e-9)
    # predictions of consecutive (buffered) ranks never cross on the training rows
    P = np.hstack([g["Xc"], np.ones((len(g["Xc"]), 1))]) @ r.β_full_
    assert np.all(np.diff(P, axis=1) >= -1e-9)
    r1 = c.CoherentLinearQuantileRegressor(quantiles=(0.5,), fit_intercept=False).fit(g["Xc"], g["yc"])
    np.testing.assert_allclose(r1.β_, g["beta_single"], rtol=1e-7, atol=1e-9)
    assert r1.predict(g["Xn"]).shape == g["pred_single"].shape
    np.testing.assert_allclose(r1.predict(g["Xn"]), g["pred_single"], rtol=1e-7, atol=1e-9)


def test_lp_argument_errors():
    c = _conformal_module()
    X, y = np.ones((8, 1)), np.arange(8.0)
    with pytest.raises(AssertionError):
        c.coherent_quantile_lp(X, y, quantiles=np.array([0.9, 0.1]))
    with pytest.raises(AssertionError):
        c.coherent_quantile_lp(X, y, quantiles=np.array([0.1, 0.9]), sample_weight=-np.ones(8))


def _stub_model(g, task):
    m = types.SimpleNamespace(_estimator_type="regressor" if task == "reg" else "classifier",
                              conformal_l1_={"Δŷ": {}, "Δŷ/ŷ": {}}, conformal_l2_={"Δŷ": {}, "Δŷ/ŷ": {}})
    for k in ("nonconformity_calib_l1_", "nonconformity_calib_l2_", "ŷ_calib_l1_", "ŷ_calib_l2_", "residuals_calib_l1_",
              "residuals_calib_l2_", "sample_weight_calib_l1_", "sample_weight_calib_l2_"):
        setattr(m, k, np.array(g["calib_" + k.replace("ŷ", "yhat")]))
    return m


@pytest.mark.parametrize("name", ["conformal_reg_n2400_d12_D128", "conformal_dual_reg_n600_d10"])
def test_conformal_layer_on_reference_calibration_split(name):
    """Given the reference's LOO calibration split and its yhat / sigma on the query rows, the restated two-level
    calibration reproduces its quantiles (regressor: quantile = yhat + offset)."""
    c = _conformal_module()
    g = np.load(GOLD / f"{name}.npz")
    m = _stub_model(g, "reg")
    yhat, sigma = g["yhat_q"], g["sigma_q"]
    scale = np.max(np.abs(g["q_default"]))
    cases = [("q_default", (0.025, 0.5, 0.975), "accuracy"), ("interval_90", (0.05, 0.95), "coverage")]
    if "dual" in name:  # the five-rank LP on 1440 calibration rows takes ~40 s on the CPU: run it on the 400-row split
        cases.append(("q_five", (0.05, 0.25, 0.5, 0.75, 0.95), "accuracy"))
    else:
        cases.append(("q_cov", (0.1, 0.9), "coverage"))
    for key, qs, prio in cases:
        got = yhat[:, None] + c.conformal_delta_quantiles(m, yhat, sigma, qs, prio)
        assert np.max(np.abs(got - g[key])) <= 1e-7 * scale, key
        assert np.all(np.diff(got, axis=1) >= -1e-9)  # coherent: quantiles do not cross
    assert (0.025, 0.5, 0.975) in m.conformal_l1_["Δŷ"] and (0.05, 0.95) in m.conformal_l2_["Δŷ/ŷ"]  # cached per tuple


def test_conformal_layer_classifier_offsets():
    c = _conformal_module()
    g = np.load(GOLD / "conformal_clf_n2400_d12_D128.npz")
    m = _stub_model(g, "clf")
    d = c.conformal_delta_quantiles(m, g["yhat_q"], g["sigma_q"], (0.025, 0.5, 0.975))
    assert d.shape == (len(g["yhat_q"]), 3) and np.all(np.diff(d, axis=1) >= -1e-9)


# ---- GPU: end to end through the estimator -------------------------------------------------------------
@pytest.mark.gpu
@pytest.mark.parametrize("name", ["conformal_reg_n2400_d12_D128", "conformal_clf_n2400_d12_D128", "conformal_dual_reg_n600_d10"])
def test_predict_quantiles_end_to_end(name):
    import neo_ls_svm_amd as hp

    g = np.load(GOLD / f"{name}.npz")
    dual = str(g["kind"]) == "dual"
    m = hp.NeoLSSVM(primal_feature_map=hp.OrthogonalRandomFourierFeatures(num_features=int(g["D"])), dual=dual).fit(g["X"], g["y"])
    Xq = g["Xq"]
    yhat, sigma = m._yhat_sigma(np.ascontiguousarray(Xq))
    np.testing.assert_allclose(yhat, g["yhat_q"], rtol=0, atol=1e-6 * np.max(np.abs(g["yhat_q"])))
    np.testing.assert_allclose(sigma, g["sigma_q"], rtol=0, atol=1e-6 * np.max(np.abs(g["sigma_q"])))
    tol = 2e-5  # LP vertices are stable under the 1e-9 differences of the LOO inputs; isotonic steps are not bit-stable
    for key, kw in (("q_default", {}), ("q_five", dict(quantiles=(0.05, 0.25, 0.5, 0.75, 0.95))),
                    ("q_cov", dict(quantiles=(0.1, 0.9), priority="coverage"))):
        got = m.predict_quantiles(Xq, **kw)
        assert got.shape == g[key].shape
        assert np.max(np.abs(got - g[key])) <= tol * max(1.0, np.max(np.abs(g[key]))), key
    got = m.predict_interval(Xq, coverage=0.9)
    assert np.max(np.abs(got - g["interval_90"])) <= tol * max(1.0, np.max(np.abs(g["interval_90"])))
    got = m.predict(Xq, coverage=0.8)
    assert np.max(np.abs(got - g["predict_cov"])) <= tol * max(1.0, np.max(np.abs(g["predict_cov"])))


@pytest.mark.gpu
def test_predict_quantiles_dataframe():
    import pandas as pd

    import neo_ls_svm_amd as hp

    g = np.load(GOLD / "conformal_clf_n2400_d12_D128.npz")
    cols = [f"f{i}" for i in range(g["X"].shape[1])]
    m = hp.NeoLSSVM(primal_feature_map=hp.OrthogonalRandomFourierFeatures(num_features=int(g["D"])), dual=False)
    m.fit(pd.DataFrame(g["X"], columns=cols), pd.Series(g["y"]))
    Xq = pd.DataFrame(g["Xq"][:50], columns=cols, index=pd.RangeIndex(100, 150, name="row"))
    df = m.predict_quantiles(Xq)
    assert df.shape == (100, 3) and df.columns.name == "quantile" and list(df.index.names) == ["class", "row"]
    np.testing.assert_allclose(df.loc[m.classes_[1]].to_numpy(), g["q_default"][:50, :, 1], atol=2e-5)
