"""Generate the golden fixtures in this directory by importing the UNMODIFIED reference.

Run in the build container only (the reference never travels to the GPU box):

    python tests/golden/make_golden.py [/root/reference]

What it does
* puts a 6-line identity stand-in for ``numba`` (absent from this image; it only decorates three
  helper loops outside the hot path) into a temp dir on ``sys.path`` and imports ``neo_ls_svm``
  from ``<reference>/src`` with bytecode writing disabled;
* fits ``NeoLSSVM`` on seeded synthetic problems and stores inputs, the pre-step outputs the hot
  path consumes (``shift_``, ``scale_``, folded ``A_``; the un-folded separator matrix and ``Z_``)
  and every hot-path output (the fitted ``*_`` attributes, ``decision_function`` / ``predict_std`` /
  ``predict`` on query rows);
* records the reference's own intermediates (A/c handed to ``eigh``, its eigenvalues, the matrix
  handed to ``cho_factor``) by wrapping those two names inside the reference module's namespace.

Only data is written (``*.npz``); no reference source is copied.
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
    fh.write(
        "def jit(*a, **k):\n"
        "    if a and callable(a[0]) and not k:\n"
        "        return a[0]\n"
        "    return lambda f: f\n"
        "njit = jit\nprange = range\n"
    )
sys.path[:0] = [_shim, str(REF / "src")]

import neo_ls_svm._neo_ls_svm as ref_mod  # noqa: E402
from neo_ls_svm import NeoLSSVM  # noqa: E402
from neo_ls_svm._affine_separator import AffineSeparator  # noqa: E402
from neo_ls_svm._affine_normalizer import AffineNormalizer  # noqa: E402
from neo_ls_svm._feature_maps import OrthogonalRandomFourierFeatures, RandomFourierFeatures  # noqa: E402

_captured: dict = {}
_orig_eigh, _orig_cho = ref_mod.eigh, ref_mod.cho_factor


def _eigh_spy(a, *args, **kw):
    lam, Q = _orig_eigh(a, *args, **kw)
    _captured["eigh_in"], _captured["lam"] = np.array(a), np.array(lam)
    return lam, Q


def _cho_spy(a, *args, **kw):
    _captured["cho_in"] = np.array(a)
    return _orig_cho(a, *args, **kw)


ref_mod.eigh, ref_mod.cho_factor = _eigh_spy, _cho_spy


def synth(n, d, task, seed):
    """The generator behind SURVEY.md 8(d): X ~ N(0,1), y = sin(Xw) + noise or a noisy half-space."""
    rng = np.random.default_rng(seed)
    X = rng.standard_normal((n, d))
    w = rng.standard_normal(d) / np.sqrt(d)
    if task == "reg":
        y = np.sin(X @ w) + 0.1 * rng.standard_normal(n)
    else:
        y = (X @ w + 0.3 * rng.standard_normal(n) > 0).astype(np.float64)
    return rng, X, y


def weights(rng, n, kind):
    if kind == "unit":
        return None
    s = rng.uniform(0.2, 3.0, size=n)
    if kind == "zeros":
        s[rng.choice(n, size=n // 20, replace=False)] = 0.0
    return s


def primal_case(name, n, d, D, task, wkind, seed, nq=257, store_A=False, fm=None):
    """``fm``: the primal feature map handed to ``NeoLSSVM`` (default: ``OrthogonalRandomFourierFeatures(num_features=D)``) - the
    plug-in point of ``_neo_ls_svm.py:62-75,380-394``: plain ``RandomFourierFeatures`` or a map with a caller-chosen affine map."""
    rng, X, y = synth(n, d, task, seed)
    s = weights(rng, n, wkind)
    Xq = rng.standard_normal((nq, d))
    _captured.clear()
    fm = OrthogonalRandomFourierFeatures(num_features=D) if fm is None else fm
    m = NeoLSSVM(primal_feature_map=fm, dual=False).fit(X, y, sample_weight=s)
    afm = m.primal_feature_map_.affine_feature_map
    sep = AffineSeparator().fit(X, m.classes_.searchsorted(y) * 2.0 - 1 if task == "clf" else y, s)
    if not isinstance(afm, AffineSeparator):
        sep.A_ = np.zeros((0, 0))  # no separator inside this map
    out = dict(
        kind="primal",
        task=task,
        X=X,
        y=y,
        s=np.ones(n) if s is None else s,
        has_weights=s is not None,
        Xq=Xq,
        D=D,
        shift=np.ravel(afm.shift_),
        scale=np.ravel(afm.scale_),
        B=afm.A_,
        A_sep=sep.A_,
        Z=m.primal_feature_map_.Z_,
        lam=_captured["lam"],
        gammas=m.γs_,
        loo_errors_gammas=m.loo_errors_γs_,
        gamma=m.γ_,
        opt=int(np.argmin(np.abs(m.γs_ - m.γ_))),
        beta=m.β̂_,
        loo_residuals=m.loo_residuals_,
        loo_yhat=m.loo_ŷ_,
        loo_leverage=m.loo_leverage_,
        loo_error=m.loo_error_,
        loo_score=m.loo_score_,
        loo_std=m.loo_std_,
        residuals=m.residuals_,
        L=m.L_[0] if store_A else np.zeros(0),
        L_lower=bool(m.L_[1]),
        decision_function=m.decision_function(Xq),
        predict_std=m.predict_std(Xq),
        predict=m.predict(Xq),
        phi_q=m.primal_feature_map_.transform(Xq[:8]),
    )
    if store_A:
        out["A_over_c"] = _captured["eigh_in"]
        out["cho_in"] = _captured["cho_in"]
    np.savez_compressed(HERE / f"{name}.npz", **out)
    print(name, "gamma", m.γ_, "opt", out["opt"], "loo_score", m.loo_score_, "r", afm.A_.shape)


def dual_case(name, n, d, task, wkind, seed, nq=129):
    rng, X, y = synth(n, d, task, seed)
    s = weights(rng, n, wkind)
    Xq = rng.standard_normal((nq, d))
    m = NeoLSSVM(dual=True).fit(X, y, sample_weight=s)
    sw = np.ones(n) if s is None else s
    nz = sw > 0
    afm = m.dual_feature_map_
    out = dict(
        kind="dual",
        task=task,
        X=X,
        y=y,
        s=sw,
        nz=nz,
        has_weights=s is not None,
        Xq=Xq,
        shift=np.ravel(afm.shift_),
        scale=np.ravel(afm.scale_),
        A_sep=afm.A_,
        Xt=m.X_,
        Xqt=afm.transform(Xq),
        gammas=m.γs_,
        loo_errors_gammas=m.loo_errors_γs_,
        gamma=m.γ_,
        opt=int(np.argmin(np.abs(m.γs_ - m.γ_))),
        alpha=m.α̂_,
        loo_residuals=m.loo_residuals_,
        loo_yhat=m.loo_ŷ_,
        loo_error=m.loo_error_,
        loo_score=m.loo_score_,
        loo_std=m.loo_std_,
        residuals=m.residuals_,
        L_lower=bool(m.L_[1]),
        decision_function=m.decision_function(Xq),
        predict_std=m.predict_std(Xq),
        predict=m.predict(Xq),
    )
    np.savez_compressed(HERE / f"{name}.npz", **out)
    print(name, "gamma", m.γ_, "opt", out["opt"], "loo_score", m.loo_score_, "r", m.X_.shape)


def ames_like(n=2930, seed=8):
    """An ames_housing-shaped table (BASELINE config 1; the real one needs ``fetch_openml``): ~40 numeric columns of
    mixed scale (areas, years, counts, skewed and zero-inflated ones), ~260 one-hot columns from 43 categoricals with
    unbalanced levels (some rare enough to be all-zero in a class bin), one constant column; log-price-like target."""
    rng = np.random.default_rng(seed)
    z = rng.standard_normal((n, 6))  # latent factors
    num = []
    for j in range(40):
        load = rng.standard_normal(6) * (rng.random(6) < 0.5)
        v = z @ load + rng.standard_normal(n)
        kind = j % 5
        if kind == 0:
            v = np.exp(0.5 * v) * 1500.0  # areas
        elif kind == 1:
            v = np.round(1970 + 20 * v)  # years
        elif kind == 2:
            v = np.clip(np.round(2 + v), 0, None)  # counts
        elif kind == 3:
            v = np.where(rng.random(n) < 0.7, 0.0, np.exp(v) * 100.0)  # zero-inflated
        num.append(v)
    cats = []
    ncols = 0
    while ncols < 260:
        levels = int(rng.integers(2, 13))
        levels = min(levels, 260 - ncols) if 260 - ncols >= 2 else 260 - ncols
        p = rng.dirichlet(np.full(levels, 0.4))
        c = rng.choice(levels, size=n, p=p)
        cats.append(np.eye(levels)[c])
        ncols += levels
    X = np.column_stack(num + cats + [np.full(n, 3.0)])
    price = 12.0 + 0.25 * z[:, 0] - 0.15 * z[:, 1] + 0.1 * np.tanh(z[:, 2] * z[:, 3]) + 0.05 * (cats[0] @ rng.standard_normal(cats[0].shape[1]))
    y = np.exp(price + 0.1 * rng.standard_normal(n))
    return rng, np.ascontiguousarray(X), y


def ames_case(name, nq=129):
    """Default ``NeoLSSVM()`` (n > 1024 -> primal, D = 512) on the ames-shaped table: the plumbing configuration."""
    rng, X, y = ames_like()
    n = X.shape[0]
    Xq = X[rng.choice(n, size=nq, replace=False)] + 0.0
    Xq[:, :40] *= 1.0 + 0.01 * rng.standard_normal((nq, 40))
    _captured.clear()
    m = NeoLSSVM().fit(X, y)
    assert m.primal_
    afm = m.primal_feature_map_.affine_feature_map
    out = dict(
        kind="primal", task="reg", X=X, y=y, s=np.ones(n), has_weights=False, Xq=Xq, D=512,
        shift=np.ravel(afm.shift_), scale=np.ravel(afm.scale_), B=afm.A_, Z_shape=np.array(m.primal_feature_map_.Z_.shape),
        lam=_captured["lam"], gammas=m.γs_, loo_errors_gammas=m.loo_errors_γs_, gamma=m.γ_,
        opt=int(np.argmin(np.abs(m.γs_ - m.γ_))), beta=m.β̂_, loo_residuals=m.loo_residuals_, loo_yhat=m.loo_ŷ_,
        loo_leverage=m.loo_leverage_, loo_error=m.loo_error_, loo_score=m.loo_score_, loo_std=m.loo_std_,
        residuals=m.residuals_, L=np.zeros(0), L_lower=bool(m.L_[1]), decision_function=m.decision_function(Xq),
        predict_std=m.predict_std(Xq), predict=m.predict(Xq), phi_q=m.primal_feature_map_.transform(Xq[:8]),
    )  # fmt: skip
    np.savez_compressed(HERE / f"{name}.npz", **out)
    print(name, "gamma", m.γ_, "opt", out["opt"], "loo_score", m.loo_score_, "X", X.shape, "r", afm.A_.shape,
          "min|scale|", np.abs(afm.scale_).min())


def exact_complexity_case(name, n, d, D, task, seed, nq=65):
    """the exact complexity matrix (``_ztz_prod_sinc_zmz`` with ``fast_approx=False``,
    ``_feature_maps.py:46-55``) and the generalised-EVD branch it sends ``_optimize_beta_gamma`` down
    (``eigh(a=A, b=C)`` + LU, ``_neo_ls_svm.py:122-124,131,139``).  Unreachable upstream (``fast_approx=True`` is
    hard-wired at ``_feature_maps.py:133``); here the flag is flipped by wrapping the module-level function."""
    import neo_ls_svm._feature_maps as fm_mod

    orig = fm_mod._ztz_prod_sinc_zmz
    fm_mod._ztz_prod_sinc_zmz = lambda Z, fast_approx=False: orig(Z, fast_approx=False)
    try:
        rng, X, y = synth(n, d, task, seed)
        Xq = rng.standard_normal((nq, d))
        _captured.clear()
        fm = OrthogonalRandomFourierFeatures(num_features=D)
        m = NeoLSSVM(primal_feature_map=fm, dual=False).fit(X, y)
        Cm = np.array(m.primal_feature_map_.complexity_matrix)
        assert not np.all(np.diag(np.diag(Cm)) == Cm)
        afm = m.primal_feature_map_.affine_feature_map
        out = dict(
            kind="primal", task=task, X=X, y=y, s=np.ones(n), has_weights=False, Xq=Xq, D=D,
            shift=np.ravel(afm.shift_), scale=np.ravel(afm.scale_), B=afm.A_, Z=m.primal_feature_map_.Z_, C=Cm,
            gammas=m.γs_, loo_errors_gammas=m.loo_errors_γs_, gamma=m.γ_, opt=int(np.argmin(np.abs(m.γs_ - m.γ_))),
            beta=m.β̂_, loo_residuals=m.loo_residuals_, loo_yhat=m.loo_ŷ_, loo_leverage=m.loo_leverage_,
            loo_error=m.loo_error_, loo_score=m.loo_score_, loo_std=m.loo_std_, residuals=m.residuals_,
            L=m.L_[0], L_lower=bool(m.L_[1]), cho_in=_captured["cho_in"], decision_function=m.decision_function(Xq),
            predict_std=m.predict_std(Xq), predict=m.predict(Xq),
        )  # fmt: skip
        np.savez_compressed(HERE / f"{name}.npz", **out)
        print(name, "gamma", m.γ_, "opt", out["opt"], "loo_score", m.loo_score_, "Z", m.primal_feature_map_.Z_.shape,
              "cond(C)", np.linalg.cond(Cm))
    finally:
        fm_mod._ztz_prod_sinc_zmz = orig


def sigma_case(name, base, sigmas):
    """gamma x sigma grid semantics of SURVEY.md 8(d) c5: B / sigma_k, 32-point sub-grid [::33]."""
    z = np.load(HERE / f"{base}.npz")
    X, y, s, shift, scale, B = z["X"], z["y"], z["s"], z["shift"], z["scale"], z["B"]
    is_clf = str(z["task"]) == "clf"
    yy = np.where(y == y.max(), 1.0, -1.0) if is_clf else y
    table = []
    for sg in sigmas:
        Tm = (X - shift[None, :]) @ ((B / sg) / scale[:, None])
        phi = np.empty((X.shape[0], B.shape[1] + 1), dtype=np.complex128)
        phi[:, :-1] = np.exp(-1j * Tm) / np.sqrt(B.shape[1])
        phi[:, -1] = 1
        m = NeoLSSVM(dual=False)
        m._estimator_type = "classifier" if is_clf else "regressor"
        m._optimize_β̂_γ(phi, yy, s, np.eye(phi.shape[1]))
        table.append(m.loo_errors_γs_[::33].copy())
    np.savez_compressed(HERE / f"{name}.npz", base=base, sigmas=np.asarray(sigmas), loo_errors=np.asarray(table))
    print(name, np.asarray(table).shape)


if __name__ == "__main__":
    if len(sys.argv) > 2 and sys.argv[2] == "ames":  # only the ames-shaped case (added in round 2)
        ames_case("primal_reg_ames_n2930_d301_D512")
        raise SystemExit(0)
    if len(sys.argv) > 2 and sys.argv[2] == "plugins":  # the feature-map plug-in point (added in round 4)
        primal_case("primal_reg_n2000_d12_RFF256", 2000, 12, 256, "reg", "uniform", seed=11, nq=65, fm=RandomFourierFeatures(num_features=256))
        primal_case("primal_clf_n1500_d10_ORF128_normalizer", 1500, 10, 128, "clf", "unit", seed=12, nq=65,
                    fm=OrthogonalRandomFourierFeatures(affine_feature_map=AffineNormalizer(), num_features=128))
        raise SystemExit(0)
    if len(sys.argv) > 2 and sys.argv[2] == "exactC":  # only the exact-complexity-matrix cases (added in round 2)
        exact_complexity_case("primal_reg_n400_d8_D192_exactC", 400, 8, 192, "reg", seed=9)
        exact_complexity_case("primal_clf_n300_d6_D128_exactC", 300, 6, 128, "clf", seed=10)
        raise SystemExit(0)
    primal_case("primal_reg_n3000_d20_D256", 3000, 20, 256, "reg", "unit", seed=0, store_A=True)
    primal_case("primal_reg_n5000_d16_D256_w", 5000, 16, 256, "reg", "uniform", seed=1)
    primal_case("primal_clf_n3000_d16_D256_wz", 3000, 16, 256, "clf", "zeros", seed=2)
    primal_case("primal_clf_n2500_d24_D192", 2500, 24, 192, "clf", "unit", seed=3)
    primal_case("primal_reg_n2000_d48_D32", 2000, 48, 32, "reg", "unit", seed=4)  # D < d branch
    dual_case("dual_reg_n300_d12", 300, 12, "reg", "unit", seed=5)
    dual_case("dual_clf_n500_d20_wz", 500, 20, "clf", "zeros", seed=6)
    dual_case("dual_reg_n1000_d32_w", 1000, 32, "reg", "uniform", seed=7)
    sigma_case("sigma_grid_reg_n3000", "primal_reg_n3000_d20_D256", [0.5, 1.0, 2.0])
    ames_case("primal_reg_ames_n2930_d301_D512")
    exact_complexity_case("primal_reg_n400_d8_D192_exactC", 400, 8, 192, "reg", seed=9)
    exact_complexity_case("primal_clf_n300_d6_D128_exactC", 300, 6, 128, "clf", seed=10)
    primal_case("primal_reg_n2000_d12_RFF256", 2000, 12, 256, "reg", "uniform", seed=11, nq=65, fm=RandomFourierFeatures(num_features=256))
    primal_case("primal_clf_n1500_d10_ORF128_normalizer", 1500, 10, 128, "clf", "unit", seed=12, nq=65,
                fm=OrthogonalRandomFourierFeatures(affine_feature_map=AffineNormalizer(), num_features=128))
