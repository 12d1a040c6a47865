"""Conformal quantile calibration on top of the hot path's leave-one-out outputs (host side, <= 1440 rows).

SURVEY.md 8(f) row 3.  The GPU part of ``predict_quantiles`` is ONE fused ``nls_primal_predict`` /
``nls_dual_predict`` call that returns yhat and sigma together (the reference makes two feature-map passes,
``_neo_ls_svm.py:571-572``); what is left is a small linear program on the calibration split, restated here:

* :func:`coherent_quantile_lp` - the coherent linear quantile regression LP of
  ``_coherent_linear_quantile_regressor.py:24-184``: pinball loss for every (buffered) quantile rank, a vanishing
  L1 term, and the constraint that consecutive quantile predictions do not cross on the training rows.
* :class:`CoherentLinearQuantileRegressor` - its estimator wrapper (``:187-272``).
* :func:`fit_conformal_level` / :func:`conformal_delta_quantiles` - the two-level calibration and the
  absolute-vs-relative choice of ``NeoLSSVM._lazily_fit_conformal_predictor`` / ``predict_quantiles``
  (``_neo_ls_svm.py:489-533``, ``:566-594``).
"""

from __future__ import annotations

import numpy as np
from scipy import sparse
from scipy.optimize import linprog
from sklearn.base import BaseEstimator, RegressorMixin
from sklearn.utils.validation import check_array, check_consistent_length, check_is_fitted, check_X_y

__all__ = ["coherent_quantile_lp", "CoherentLinearQuantileRegressor", "fit_conformal_level", "conformal_delta_quantiles"]


def _buffered_ranks(quantiles: np.ndarray, buffer: int) -> np.ndarray:
    """Insert ``buffer`` evenly spaced auxiliary ranks between consecutive requested ranks (``:63-68``)."""
    m = len(quantiles)
    grid = np.linspace(0, m - 1, (m - 1) * (1 + buffer) + 1)
    return np.interp(grid, np.arange(m), quantiles).astype(quantiles.dtype)


def coherent_quantile_lp(X, y, *, quantiles, sample_weight=None, coherence_buffer=3):
    """Solve  min sum_j sum_i w_i pinball_{q_j}(y_i - x_i b_j) / Q + alpha |b|_1   s.t.  X b_j <= X b_{j+1}.

    Unknowns, in this order (the order fixes which optimal vertex HiGHS returns, so it is part of the contract):
    b (Q x f, rank-major), t >= |b| (Q x f), over-shoot u >= 0 (Q x n), under-shoot v >= 0 (Q x n) with
    x_i b_j - y_i = u_ji - v_ji.  Returns (b for the requested ranks [f x m], b for all buffered ranks [f x Q]).
    """
    n, f = X.shape
    ranks = _buffered_ranks(np.asarray(quantiles), coherence_buffer)
    Q = len(ranks)
    if not np.array_equal(ranks, np.sort(ranks)):
        raise AssertionError("Quantile ranks must be sorted.")
    if sample_weight is not None and not np.all(sample_weight >= 0):
        raise AssertionError("Sample weights must be >= 0.")
    w = np.ones(n, dtype=y.dtype) if sample_weight is None else sample_weight
    w /= np.sum(w)  # in place, as the reference does (:76): a caller's weight array comes back normalised
    alpha = np.sqrt(np.finfo(y.dtype).eps) / (Q * f)
    nb, nr = Q * f, Q * n

    cost = np.concatenate([np.zeros(nb, dtype=y.dtype), np.full(nb, alpha, dtype=y.dtype), np.kron((1 - ranks) / Q, w), np.kron(ranks / Q, w)])
    I_r, I_b = sparse.identity(nr, dtype=X.dtype, format="csr"), sparse.identity(nb, dtype=X.dtype, format="csr")
    # residual definition: one block row per rank
    A_eq = sparse.hstack([sparse.kron(sparse.identity(Q, dtype=X.dtype), X), sparse.csr_matrix((nr, nb), dtype=X.dtype), -I_r, I_r])
    b_eq = np.tile(y, Q)
    # |b| <= t, and no crossing between neighbouring ranks: (u_j - v_j) - (u_{j+1} - v_{j+1}) <= 0
    step = sparse.diags([1, -1], offsets=[0, 1], shape=(Q - 1, Q), dtype=X.dtype)
    cross = sparse.kron(step, sparse.identity(n, dtype=X.dtype))
    pad_r = sparse.csr_matrix((nb, 2 * nr), dtype=X.dtype)
    pad_b = sparse.csr_matrix(((Q - 1) * n, 2 * nb), dtype=X.dtype)
    A_ub = sparse.vstack([sparse.hstack([I_b, -I_b, pad_r]), sparse.hstack([-I_b, -I_b, pad_r]), sparse.hstack([pad_b, cross, -cross])])
    b_ub = np.zeros(A_ub.shape[0], dtype=X.dtype)
    bounds = [(None, None)] * nb + [(0, None)] * (nb + 2 * nr)
    sol = linprog(c=cost, A_ub=A_ub, b_ub=b_ub, A_eq=A_eq, b_eq=b_eq, bounds=bounds, method="highs")
    if sol.x is None:
        raise RuntimeError(f"coherent quantile regression LP failed: {sol.message}")
    beta_full = sol.x[:nb].astype(y.dtype).reshape(Q, f).T
    return beta_full[:, :: coherence_buffer + 1], beta_full


class CoherentLinearQuantileRegressor(RegressorMixin, BaseEstimator):
    """Linear model for several quantile ranks whose predictions never cross on the training rows."""

    def __init__(self, *, quantiles=(0.025, 0.5, 0.975), fit_intercept=True, coherence_buffer=3):
        self.quantiles = quantiles
        self.fit_intercept = fit_intercept
        self.coherence_buffer = coherence_buffer

    def _design(self, X):
        return np.hstack([X, np.ones((X.shape[0], 1), dtype=X.dtype)]) if self.fit_intercept else X

    def fit(self, X, y, *, sample_weight=None):
        X, y = check_X_y(X, y, dtype=(np.float64, np.float32), y_numeric=True)
        self.n_features_in_ = X.shape[1]
        self.y_dtype_ = X.dtype if np.issubdtype(y.dtype, np.integer) else y.dtype
        if np.issubdtype(y.dtype, np.datetime64) or np.issubdtype(y.dtype, np.timedelta64):
            X, y = X.astype(np.float64), y.astype(np.float64)
        y = y.astype(X.dtype)
        if sample_weight is not None:
            check_consistent_length(y, sample_weight)
            sample_weight = np.asarray(sample_weight).astype(y.dtype)
        self.β_, self.β_full_ = coherent_quantile_lp(
            self._design(X), y, quantiles=np.asarray(self.quantiles).astype(y.dtype), sample_weight=sample_weight,
            coherence_buffer=self.coherence_buffer,
        )
        return self

    def predict(self, X):
        check_is_fitted(self)
        X = check_array(X, dtype=self.β_.dtype)
        yhat = self._design(X) @ self.β_
        return yhat[:, 0] if yhat.shape[1] == 1 else yhat

    def intercept_clip(self, X, y):
        """[2 x m] interval by which each rank's intercept may move without crossing a neighbouring (buffered) rank."""
        check_is_fitted(self)
        X, y = check_X_y(X, y, dtype=self.β_.dtype, y_numeric=True)
        R = self._design(X) @ self.β_full_ - y[:, None]
        gap_down = np.max(R[:, :-1] - R[:, 1:], axis=0)  # <= 0: room towards the rank below
        gap_up = np.min(R[:, 1:] - R[:, :-1], axis=0)    # >= 0: room towards the rank above
        clip = np.vstack([np.concatenate(([-np.inf], gap_down)), np.concatenate((gap_up, [np.inf]))])
        clip[:, clip[0] >= clip[1]] = 0
        return clip[:, :: self.coherence_buffer + 1]


def _calibration_design(nonconformity, yhat, is_regressor):
    cols = [nonconformity[:, None]]
    if is_regressor:
        cols.append(np.abs(yhat)[:, None])
    return np.hstack(cols)


def fit_conformal_level(model, relative: bool, quantiles: np.ndarray):
    """Level 1 (coherent quantile regression of the LOO residuals on [sigma_loo, |yhat_loo|]) and level 2 (a clipped
    per-rank bias from the second split): ``_neo_ls_svm.py:500-532``.  ``model`` carries the ``*_calib_l1_/l2_`` arrays."""
    is_reg = model._estimator_type == "regressor"
    eps = np.finfo(model.ŷ_calib_l1_.dtype).eps

    def problem(sigma, yhat, resid):
        scale = np.maximum(np.abs(yhat), eps) if relative else 1
        return _calibration_design(sigma, yhat, is_reg), -resid / scale

    X1, y1 = problem(model.nonconformity_calib_l1_, model.ŷ_calib_l1_, model.residuals_calib_l1_)
    cqr = CoherentLinearQuantileRegressor(quantiles=quantiles).fit(X1, y1, sample_weight=model.sample_weight_calib_l1_)
    bias = np.zeros(quantiles.shape, dtype=model.ŷ_calib_l1_.dtype)
    if len(model.ŷ_calib_l2_) >= 128:
        X2, y2 = problem(model.nonconformity_calib_l2_, model.ŷ_calib_l2_, model.residuals_calib_l2_)
        pred2 = cqr.predict(X2)
        pred2 = pred2[:, None] if pred2.ndim == 1 else pred2
        clip = cqr.intercept_clip(np.vstack([X1, X2]), np.hstack([y1, y2]))
        for j, q in enumerate(quantiles):
            bias[j] = np.clip(np.quantile(y2 - pred2[:, j], q), clip[0, j], clip[1, j])
    return cqr, bias


def conformal_delta_quantiles(model, yhat, sigma, quantiles, priority="accuracy"):
    """Offsets to add to yhat, [m x len(quantiles)]: per row the less dispersed of the absolute and the relative
    conformal model (``_neo_ls_svm.py:566-594``).  Fitted levels are cached on ``model.conformal_l1_/l2_``."""
    quantiles = np.asarray(quantiles)
    key = tuple(quantiles)
    fitted = []
    for name, relative in (("Δŷ", False), ("Δŷ/ŷ", True)):
        if key not in model.conformal_l1_[name]:
            model.conformal_l1_[name][key], model.conformal_l2_[name][key] = fit_conformal_level(model, relative, quantiles)
        fitted.append((model.conformal_l1_[name][key], model.conformal_l2_[name][key]))
    (cqr_abs, bias_abs), (cqr_rel, bias_rel) = fitted
    if priority == "coverage":  # only let the level-2 bias widen the interval (in place on the cached bias, as upstream)
        up, down = quantiles >= 0.5, quantiles <= 0.5
        for b in (bias_abs, bias_rel):
            b[up] = np.maximum(b[up], 0)
            b[down] = np.minimum(b[down], 0)
    Xc = _calibration_design(sigma, yhat, model._estimator_type == "regressor")

    def pred(cqr):
        p = cqr.predict(Xc)
        return p[:, None] if p.ndim == 1 else p

    both = np.dstack([pred(cqr_abs) + bias_abs[None, :], np.abs(yhat)[:, None] * (pred(cqr_rel) + bias_rel[None, :])])
    pick = np.argmin(np.std(both, axis=1), axis=-1)
    return both[np.arange(both.shape[0]), :, pick]
