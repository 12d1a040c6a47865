"""Golden fixtures for the conformal quantile layer, from the UNMODIFIED reference.

    python tests/golden/make_golden_conformal.py [/root/reference]

Same mechanism as make_golden.py (identity ``numba`` stand-in in a temp dir, reference imported from
``<reference>/src``, only data written).  Per case: the training problem, the query rows, the reference's
calibration split (``*_calib_l1_/l2_``), its yhat / sigma on the query rows and its ``predict_quantiles`` /
``predict_interval`` outputs; plus one stand-alone coherent-quantile-regression problem with its coefficients.
"""
from __future__ import annotations

import os
import sys
import tempfile
from pathlib import Path

import numpy as np

HERE = Path(__file__).resolve().parent
REF = Path(sys.argv[1] if len(sys.argv) > 1 else "/root/reference")
sys.dont_write_bytecode = True
_shim = tempfile.mkdtemp(prefix="numba_shim_")
with open(os.path.join(_shim, "numba.py"), "w") as fh:
    fh.write("def jit(*a, **k):\n    if a and callable(a[0]) and not k:\n        return a[0]\n    return lambda f: f\nnjit = jit\nprange = range\n")
sys.path[:0] = [_shim, str(REF / "src")]

from neo_ls_svm import NeoLSSVM  # noqa: E402
from neo_ls_svm._coherent_linear_quantile_regressor import CoherentLinearQuantileRegressor  # noqa: E402
from neo_ls_svm._feature_maps import OrthogonalRandomFourierFeatures  # noqa: E402


def synth(n, d, task, seed):
    rng = np.random.default_rng(seed)
    X = rng.standard_normal((n, d))
    w = rng.standard_normal(d) / np.sqrt(d)
    y = np.sin(X @ w) + 0.1 * rng.standard_normal(n) if task == "reg" else (X @ w + 0.3 * rng.standard_normal(n) > 0).astype(np.float64)
    return rng, X, y


def model_case(name, n, d, D, task, seed, dual=False, nq=193):
    rng, X, y = synth(n, d, task, seed)
    Xq = rng.standard_normal((nq, d))
    m = NeoLSSVM(primal_feature_map=OrthogonalRandomFourierFeatures(num_features=D), dual=dual).fit(X, y)
    calib = {k: np.array(getattr(m, k)) for k in (
        "nonconformity_calib_l1_", "nonconformity_calib_l2_", "ŷ_calib_l1_", "ŷ_calib_l2_", "residuals_calib_l1_",
        "residuals_calib_l2_", "sample_weight_calib_l1_", "sample_weight_calib_l2_")}
    out = dict(kind="dual" if dual else "primal", task=task, X=X, y=y, Xq=Xq, D=D,
               yhat_q=np.asarray(m.decision_function(Xq)), sigma_q=np.asarray(m.predict_std(Xq)),
               q_default=np.asarray(m.predict_quantiles(Xq)),
               q_five=np.asarray(m.predict_quantiles(Xq, quantiles=(0.05, 0.25, 0.5, 0.75, 0.95))),
               q_cov=np.asarray(m.predict_quantiles(Xq, quantiles=(0.1, 0.9), priority="coverage")),
               interval_90=np.asarray(m.predict_interval(Xq, coverage=0.9)),
               predict_cov=np.asarray(m.predict(Xq, coverage=0.8)))
    out.update({"calib_" + k.replace("ŷ", "yhat"): v for k, v in calib.items()})
    np.savez_compressed(HERE / f"{name}.npz", **out)
    print(name, {k: getattr(v, "shape", v) for k, v in out.items() if k.startswith("q_") or k.startswith("interval")})


def lp_case(name, n, seed):
    rng = np.random.default_rng(seed)
    sig = np.abs(rng.standard_normal(n)) + 0.1
    mag = np.abs(rng.standard_normal(n))
    Xc = np.column_stack([sig, mag])
    yc = sig * rng.standard_normal(n) + 0.05 * mag
    w = rng.uniform(0.5, 2.0, n)
    q = (0.05, 0.5, 0.95)
    r = CoherentLinearQuantileRegressor(quantiles=q).fit(Xc, yc, sample_weight=w.copy())
    r1 = CoherentLinearQuantileRegressor(quantiles=(0.5,), fit_intercept=False).fit(Xc, yc)
    Xn = np.abs(rng.standard_normal((37, 2)))
    np.savez_compressed(HERE / f"{name}.npz", Xc=Xc, yc=yc, w=w, quantiles=np.asarray(q), beta=r.β_, beta_full=r.β_full_, Xn=Xn,
                        pred=r.predict(Xn), clip=r.intercept_clip(Xc, yc), beta_single=r1.β_, pred_single=r1.predict(Xn))
    print(name, r.β_.shape, r.β_full_.shape)


if __name__ == "__main__":
    lp_case("conformal_lp_n400", 400, 11)
    model_case("conformal_reg_n2400_d12_D128", 2400, 12, 128, "reg", 21)
    model_case("conformal_clf_n2400_d12_D128", 2400, 12, 128, "clf", 22)
    model_case("conformal_dual_reg_n600_d10", 600, 10, 128, "reg", 23, dual=True)
