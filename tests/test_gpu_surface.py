"""The sklearn-ecosystem surface the reference's own suite exercises, on the GPU estimator.

``/root/reference/tests/test_neo_ls_svm.py:16-67`` puts ``NeoLSSVM()`` at the end of ``make_pipeline(...)``, wraps it in
``OneVsRestClassifier`` for multiclass targets, checks the coherence of predicted quantiles and the coverage of predicted intervals
(``>= 0.97 x`` the nominal coverage); ``:70-108`` round-trips pandas input (Series / DataFrame out, same index, same numbers as the NumPy
path).  The reference's datasets come from OpenML (no network here), so the same checks run on synthetic tables of the same kinds.
float32 input (``_feature_maps.py:197-200`` branches on it: complex64 features) is WIDENED to float64 here - this library computes in
float64 throughout - so the fit must equal the float64 fit of the widened array exactly."""

from __future__ import annotations

import json
from pathlib import Path

import numpy as np
import pytest

pytestmark = pytest.mark.gpu

# What the unmodified reference measures on the same two tables (tests/golden/make_surface_reference.py, run in the build container).
REFERENCE = json.loads((Path(__file__).resolve().parent / "golden" / "surface_reference.json").read_text())


def _regression_table(n=4000, seed=0):
    from sklearn.datasets import make_friedman1

    X, y = make_friedman1(n_samples=n, n_features=10, noise=1.0, random_state=seed)
    return X * np.array([1, 10, 100, 1, 1, 5, 1, 1, 1, 1.0]) + 3.0, 4.0 * y - 20.0  # un-normalised columns: the pipeline's scaler and the pre-step matter


def _binary_table(n=4000, seed=1):
    from sklearn.datasets import make_moons

    rng = np.random.default_rng(seed)
    X2, y = make_moons(n_samples=n, noise=0.25, random_state=seed)
    X = np.hstack([X2, rng.standard_normal((n, 4))]) * np.array([3.0, 1.0, 1.0, 20.0, 1.0, 1.0])
    return X, np.where(y == 1, "pos", "neg")  # string labels: classes_ / predict must hand them back


def _split(X, y, seed=42):
    from sklearn.model_selection import train_test_split

    return train_test_split(X, y, test_size=0.15, random_state=seed)


def test_pipeline_regression_quantiles_and_coverage():
    """``make_pipeline(StandardScaler(), NeoLSSVM())`` on a regression table: better than SVR and a linear model (the reference's
    ``neo_ls_svm_score > svm_score``), quantiles monotone in the level, intervals ordered.  Coverage: on THIS table the reference's own
    intervals cover 0.64 / 0.76 / 0.87 / 0.94 of the test rows - short of its suite's ``0.97 x nominal`` bar, which is a statement about its
    OpenML datasets - so the check here is the stronger one: the same score and the same coverage as the reference measured
    (``tests/golden/surface_reference.json``), to one test row."""
    from sklearn.linear_model import Ridge
    from sklearn.pipeline import make_pipeline
    from sklearn.preprocessing import StandardScaler
    from sklearn.svm import SVR

    from neo_ls_svm_amd import NeoLSSVM

    Xtr, Xte, ytr, yte = _split(*_regression_table())
    pipe = make_pipeline(StandardScaler(), NeoLSSVM()).fit(Xtr, ytr)
    model = pipe.steps[-1][1]
    assert model.primal_ and model._estimator_type == "regressor"  # n = 3400 > 1024: the primal path (_neo_ls_svm.py:377)
    score = pipe.score(Xte, yte)
    assert score > make_pipeline(StandardScaler(), SVR()).fit(Xtr, ytr).score(Xte, yte)
    assert score > make_pipeline(StandardScaler(), Ridge()).fit(Xtr, ytr).score(Xte, yte) + 0.05
    ref = REFERENCE["regression"]
    assert len(yte) == ref["n_test"] and abs(score - ref["score"]) < 1e-6
    q = pipe.predict(Xte, quantiles=np.linspace(0.1, 0.9, 3))
    assert q.shape == (len(yte), 3)
    for j in range(q.shape[1] - 1):
        assert np.all(q[:, j] <= q[:, j + 1])
    for want, ref_cov in zip(REFERENCE["coverages"], ref["coverage"]):
        iv = pipe.predict(Xte, coverage=want)
        assert iv.shape == (len(yte), 2) and np.all(iv[:, 0] <= iv[:, 1])
        covered = (iv[:, 0] <= yte) & (yte <= iv[:, 1])
        assert abs(covered.mean() - ref_cov) <= 1.5 / len(yte), (want, covered.mean(), ref_cov)
        assert covered.mean() >= 0.9 * want


def test_pipeline_binary_classification_quantiles_and_coverage():
    """The same for a binary target with string labels: probabilities in [0, 1], per-class quantiles monotone, interval coverage as the
    reference measures it (``tests/test_neo_ls_svm.py:53-62``)."""
    from sklearn.linear_model import LogisticRegression
    from sklearn.pipeline import make_pipeline
    from sklearn.preprocessing import StandardScaler
    from sklearn.svm import SVC

    from neo_ls_svm_amd import NeoLSSVM

    Xtr, Xte, ytr, yte = _split(*_binary_table())
    pipe = make_pipeline(StandardScaler(), NeoLSSVM()).fit(Xtr, ytr)
    model = pipe.steps[-1][1]
    assert model._estimator_type == "classifier" and list(model.classes_) == ["neg", "pos"]
    pred = pipe.predict(Xte)
    assert set(np.unique(pred)) <= {"neg", "pos"}
    score = pipe.score(Xte, yte)
    assert score > make_pipeline(StandardScaler(), LogisticRegression()).fit(Xtr, ytr).score(Xte, yte) + 0.02
    assert score >= make_pipeline(StandardScaler(), SVC()).fit(Xtr, ytr).score(Xte, yte) - 0.02
    ref = REFERENCE["binary"]
    assert len(yte) == ref["n_test"] and abs(score - ref["score"]) <= 1.5 / len(yte)
    proba = pipe.predict_proba(Xte)
    assert proba.shape == (len(yte), 2) and np.allclose(proba.sum(axis=1), 1.0) and proba.min() >= 0.0
    q = pipe.predict(Xte, quantiles=np.linspace(0.1, 0.9, 3))
    assert q.shape == (len(yte), 3, 2)
    for j in range(q.shape[1] - 1):
        for k in range(q.shape[2]):
            assert np.all(q[:, j, k] <= q[:, j + 1, k])
    is_neg = yte == model.classes_[0]
    for want, ref_cov in zip(REFERENCE["coverages"], ref["coverage"]):
        iv = pipe.predict(Xte, coverage=want)
        assert iv.shape == (len(yte), 2, 2) and iv.min() >= 0.0 and iv.max() <= 1.0
        assert np.all(iv[:, 0, 0] <= iv[:, 1, 0]) and np.all(iv[:, 0, 1] <= iv[:, 1, 1])
        covered = (np.any(iv[:, :, 0] > 0.5, axis=1) & is_neg) | (np.any(iv[:, :, 1] > 0.5, axis=1) & ~is_neg)
        assert covered.mean() >= 0.97 * want, (want, covered.mean())  # the reference's bar (it holds on this table for the reference too)
        assert abs(covered.mean() - ref_cov) <= 1.5 / len(yte), (want, covered.mean(), ref_cov)


def test_one_vs_rest_multiclass():
    """``OneVsRestClassifier(NeoLSSVM())`` (``tests/test_neo_ls_svm.py:28-29``): three classes, one binary model per class (cloned by
    sklearn, each through the whole pre-step + fit), scores above a linear baseline, ``predict_proba`` rows sum to one."""
    from sklearn.datasets import make_blobs
    from sklearn.linear_model import LogisticRegression
    from sklearn.multiclass import OneVsRestClassifier

    from neo_ls_svm_amd import NeoLSSVM

    rng = np.random.default_rng(3)
    Xb, blob = make_blobs(n_samples=3600, centers=9, n_features=5, cluster_std=1.6, random_state=3)
    y = blob % 3  # each class is a union of three blobs: not linearly separable
    X = np.hstack([Xb, rng.standard_normal((Xb.shape[0], 2))])
    Xtr, Xte, ytr, yte = _split(X, y)
    ovr = OneVsRestClassifier(NeoLSSVM()).fit(Xtr, ytr)
    assert len(ovr.estimators_) == 3 and all(e._estimator_type == "classifier" and e.primal_ for e in ovr.estimators_)
    assert list(ovr.classes_) == [0, 1, 2]
    score = ovr.score(Xte, yte)
    assert score > LogisticRegression(max_iter=500).fit(Xtr, ytr).score(Xte, yte) + 0.1, score
    proba = ovr.predict_proba(Xte)
    assert proba.shape == (len(yte), 3) and np.allclose(proba.sum(axis=1), 1.0) and proba.min() >= 0.0
    assert np.mean(np.argmax(proba, axis=1) == ovr.predict(Xte)) > 0.98


@pytest.mark.parametrize("task", ["regression", "binary"])
def test_pandas_in_pandas_out(task):
    """``tests/test_neo_ls_svm.py:70-108``: a DataFrame in gives a Series (vector methods) or a DataFrame (quantiles, intervals, class
    probabilities) out, carrying the input's index and the very numbers of the NumPy path."""
    import pandas as pd

    from neo_ls_svm_amd import NeoLSSVM

    Xa, ya = _regression_table(2400) if task == "regression" else _binary_table(2400)
    idx = pd.Index(np.arange(len(ya)) * 7 + 11, name="row")
    X = pd.DataFrame(Xa, columns=[f"f{j}" for j in range(Xa.shape[1])], index=idx)
    y = pd.Series(ya, index=idx, name="target")
    Xtr, Xte, ytr, yte = _split(X, y)
    binary = task == "binary"
    model = NeoLSSVM().fit(Xtr, ytr)
    for name in ["decision_function", "predict", "predict_std"] + ([] if binary else ["predict_proba"]):
        method = getattr(model, name)
        out_np, out_pd = method(np.asarray(Xte)), method(Xte)
        assert isinstance(out_pd, pd.Series), name
        assert np.all(np.asarray(out_pd) == out_np) and out_pd.index.equals(Xte.index), name
    for name in ["predict_quantiles", "predict_interval"] + (["predict_proba"] if binary else []):
        method = getattr(model, name)
        out_np, out_pd = method(np.asarray(Xte)), method(Xte)
        tensor = out_np.ndim == 3
        if tensor:  # classifier quantiles: one block of rows per class, as the reference stacks them
            out_np = np.vstack([out_np[:, :, k] for k in range(out_np.shape[2])])
        assert isinstance(out_pd, pd.DataFrame), name
        assert np.all(np.asarray(out_pd) == out_np), name
        assert tensor or out_pd.index.equals(Xte.index), name
    # the keyword forms of predict go through the same methods
    assert isinstance(model.predict(Xte, quantiles=(0.25, 0.75)), pd.DataFrame)
    assert isinstance(model.predict(Xte, coverage=0.8), pd.DataFrame)


@pytest.mark.parametrize("dual", [False, True])
def test_float32_input_is_widened(dual):
    """float32 ``X`` (the reference switches to complex64 features, ``_feature_maps.py:197-200``; 5e-4 away from its own complex128 path,
    BASELINE.md): accepted, widened, and the fit IS the float64 fit of the widened array - every fitted attribute, bit for bit."""
    from neo_ls_svm_amd import NeoLSSVM

    Xa, ya = _regression_table(2600 if not dual else 700, seed=5)
    X32 = Xa.astype(np.float32)
    a = NeoLSSVM(dual=dual).fit(X32, ya)
    b = NeoLSSVM(dual=dual).fit(X32.astype(np.float64), ya)
    assert a.dual_ == dual and a.γ_ == b.γ_
    coef_a, coef_b = (a.α̂_, b.α̂_) if dual else (a.β̂_, b.β̂_)
    assert coef_a.dtype == coef_b.dtype and np.array_equal(coef_a, coef_b)
    assert np.array_equal(a.loo_residuals_, b.loo_residuals_) and np.array_equal(a.loo_errors_γs_, b.loo_errors_γs_)
    Xq = X32[:300]
    assert np.array_equal(np.asarray(a.predict(Xq)), np.asarray(b.predict(Xq.astype(np.float64))))
    assert np.array_equal(np.asarray(a.predict_std(Xq)), np.asarray(b.predict_std(Xq.astype(np.float64))))
    assert np.asarray(a.predict(Xq)).dtype == ya.dtype
