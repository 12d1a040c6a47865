"""CPU oracle for the neo-ls-svm fit/predict hot path.  TEST INFRASTRUCTURE ONLY.

This module is a NumPy restatement of the reference's algorithm for the path named by
BASELINE.json (ORF feature map -> primal Hermitian normal equations -> EVD gamma-sweep of the
leave-one-out residuals -> Cholesky re-solve; dual kernel path with its EVD sweep).  It is the
*checker* for the HIP library: only ``tests/``, ``__graft_entry__.smoke()`` and the ``cpu_baseline``
leg of ``bench.py`` may import it.  The product (``neo_ls_svm_amd``) never does and fails loudly when
the HIP extension is missing.

Parity status: PINNED.  Every function here is checked against fixtures captured by importing the
unmodified reference in the build container (``tests/golden/make_golden.py`` writes them,
``tests/test_oracle_golden.py`` checks them).  The reference's own tests hold no numeric vectors
for this path, so the imported-reference fixtures are the only pin.

All citations ``file:line`` are into the reference tree (``src/neo_ls_svm/...``).

Two schedules of the primal solver are provided:

* ``primal_fit_faithful``  - the reference's schedule: a materialised phi and the same five
  zgemm-class products as ``_neo_ls_svm.py:112-187`` (used for parity pinning and as the honest
  "reference" CPU baseline where phi fits in host RAM).
* ``primal_fit_streamed``  - algebraically identical, simplified schedule (one rotation P = phi Q,
  h = s^2 |P|^2 / c, phi beta = Re(P v)), row tiled so n = 1e6 streams in bounded memory.  This is
  also the schedule the HIP library implements, stage by stage.
"""

from __future__ import annotations

import numpy as np
import scipy.linalg as sla

__all__ = [
    "gamma_grid",
    "orf_frequencies",
    "rff_frequencies",
    "fold_projection",
    "affine_project",
    "feature_map",
    "exact_complexity_matrix",
    "primal_gram",
    "primal_fit_faithful",
    "primal_fit_streamed",
    "primal_decision_function",
    "primal_predict_std",
    "rbf_gram",
    "dual_fit_faithful",
    "dual_fit_reduced",
    "dual_decision_function",
    "dual_predict_std",
    "select_gamma",
    "weighted_scores",
]


# --------------------------------------------------------------------------------------------
# Shared helpers
# --------------------------------------------------------------------------------------------
def gamma_grid(num: int, dtype=np.float64) -> np.ndarray:
    """The reference's gamma grid: ``_neo_ls_svm.py:146`` (num=1024, primal), ``:270`` (128, dual)."""
    return np.logspace(np.log10(1e-6), np.log10(20), num, dtype=dtype)


def orf_frequencies(d: int, D: int, seed=42, dtype=np.float64) -> np.ndarray:
    """Orthogonal random frequency matrix Z (d x D).

    Follows ``_feature_maps.py:209-223``: legacy ``RandomState(seed).randn`` draw, each block of d
    columns replaced by the Q factor of its QR decomposition, then every column rescaled by the
    square root of a chi-square(d) draw from the same generator.
    """
    gen = seed if isinstance(seed, np.random.RandomState) else np.random.RandomState(seed)
    Z = gen.randn(d, D).astype(dtype)
    start = 0
    while start < D:
        stop = min(start + d, D)
        q, _ = np.linalg.qr(Z[:, start : start + d])
        Z[:, start:stop] = q
        start += d
    chi = np.sqrt(gen.chisquare(d, size=(1, D)).astype(dtype))
    return Z * chi


def rff_frequencies(d: int, D: int, seed=42, dtype=np.float64) -> np.ndarray:
    """Plain random Fourier frequencies: ``_feature_maps.py:120-127`` - Z = RandomState(seed).randn(d, D), nothing else."""
    gen = seed if isinstance(seed, np.random.RandomState) else np.random.RandomState(seed)
    return gen.randn(d, D).astype(dtype)


def fold_projection(A_sep: np.ndarray | None, Z: np.ndarray) -> np.ndarray:
    """Fold Z into the separator's matrix: ``_feature_maps.py:147-150`` (A_ <- A @ Z, or Z)."""
    return Z if A_sep is None else A_sep @ Z


def affine_project(X: np.ndarray, shift: np.ndarray, scale: np.ndarray, B: np.ndarray) -> np.ndarray:
    """T = (X - shift) diag(1/scale) B, with the reference's memory-order switch.

    ``_affine_feature_map.py:81-89``: when B has fewer columns than rows the reference evaluates
    ``X @ Bs - shift @ Bs`` instead of ``(X - shift) @ Bs`` (Bs = B / scale^T).
    """
    shift = np.reshape(shift, (1, -1))
    scale = np.reshape(scale, (1, -1))
    Bs = B / scale.T
    if B.shape[1] < B.shape[0]:
        T = X @ Bs - shift @ Bs
    else:
        T = (X - shift) @ Bs
    return T.astype(X.dtype)


def feature_map(X: np.ndarray, shift: np.ndarray, scale: np.ndarray, B: np.ndarray) -> np.ndarray:
    """phi(X) in C^{n x (D+1)}: ``_feature_maps.py:195-202``.

    phi[:, :D] = exp(-i T) / sqrt(D) and phi[:, D] = 1 (bias column).
    """
    T = affine_project(X, shift, scale, B)
    D = B.shape[1]
    phi = np.empty((X.shape[0], D + 1), dtype=np.complex128)
    np.exp(-1j * T, out=phi[:, :D])
    phi[:, :D] /= np.sqrt(D)
    phi[:, D] = 1.0
    return phi


def select_gamma(e: np.ndarray, s: np.ndarray, is_clf: bool):
    """Per-gamma LOO error vector and the index the reference picks.

    ``_neo_ls_svm.py:158-165`` (and ``:295-302`` for the dual): errs = s @ |e|; the regressor takes
    its argmin, the classifier the argmin of s @ [|e| >= 1] + s @ max(0, |e| - 1) + errs.
    ``e`` must already carry the classifier clipping of ``:153-155``.
    """
    abs_e = np.abs(e)
    errs = s @ abs_e
    if is_clf:
        objective = s @ (abs_e >= 1) + s @ np.maximum(0, abs_e - 1) + errs
    else:
        objective = errs
    return errs, int(np.argmin(objective)), objective


def clip_classifier_residuals(e: np.ndarray, y: np.ndarray) -> np.ndarray:
    """Zero residuals on the correct side of the margin: ``_neo_ls_svm.py:153-155`` / ``:180-182``."""
    ycol = y if e.ndim == 1 else y[:, None]
    e = e.copy()
    e[(ycol > 0) & (e > 0)] = 0
    e[(ycol < 0) & (e < 0)] = 0
    return e


def weighted_scores(y: np.ndarray, yhat: np.ndarray, s: np.ndarray, is_clf: bool) -> float:
    """``accuracy_score(y, sign(yhat), sample_weight=s)`` / ``r2_score(y, yhat, sample_weight=s)``.

    ``_neo_ls_svm.py:171-174``.  Written out so the oracle does not need sklearn.
    """
    if is_clf:
        return float(np.sum(s * (np.sign(yhat) == y)) / np.sum(s))
    ybar = np.sum(s * y) / np.sum(s)
    return float(1.0 - np.sum(s * (y - yhat) ** 2) / np.sum(s * (y - ybar) ** 2))


# --------------------------------------------------------------------------------------------
# Primal path
# --------------------------------------------------------------------------------------------
def primal_gram(phi: np.ndarray, y: np.ndarray, s: np.ndarray):
    """Weighted Hermitian normal equations: ``_neo_ls_svm.py:110-114,127``.

    Returns (A, b, s_normalised) with A = (S phi)^H (S phi) Hermitianised and b = (S phi)^H (s y).
    """
    sn = s / np.sum(s)
    F = sn[:, None] * phi
    A = F.conj().T @ F
    A = (A + A.conj().T) / 2
    b = F.conj().T @ (sn * y)
    return A, b, sn


def exact_complexity_matrix(Z: np.ndarray) -> np.ndarray:
    """The exact complexity matrix: the slow branch of ``_ztz_prod_sinc_zmz`` (``_feature_maps.py:46-55``,
    ``fast_approx=False``) embedded the way ``complexity_matrix`` embeds it (``:131-134``: identity of size D + 1 with
    the D x D block replaced).  C_ij = (1/d) (Z'Z)_ij prod_k sin(Z_ki - Z_kj) / (Z_ki - Z_kj), factors with
    |Z_ki - Z_kj| <= eps skipped; only the lower triangle is multiplied and then mirrored (``:54``)."""
    dp, D = Z.shape
    Cm = Z.T @ Z
    eps = np.finfo(Z.dtype).eps
    il = np.tril_indices(D, -1)
    for k in range(dp):
        dz = Z[k, il[0]] - Z[k, il[1]]
        f = np.ones_like(dz)
        m = np.abs(dz) > eps
        f[m] = np.sin(dz[m]) / dz[m]
        Cm[il] *= f
    Cm = (np.tril(Cm) + np.tril(Cm, -1).T) / dp
    out = np.eye(D + 1, dtype=Z.dtype)
    out[:-1, :-1] = Cm
    return out


def primal_fit_faithful(phi: np.ndarray, y: np.ndarray, s: np.ndarray, is_clf: bool, gammas=None, C=None) -> dict:
    """The reference's own schedule of ``_optimize_beta_gamma``.

    ``_neo_ls_svm.py:110-187``.  C = None stands for I_{D+1}, what ``complexity_matrix`` returns for RFF/ORF
    (``_feature_maps.py:133-134``): the diagonal-C branch (``:119-121``) with the normalised diagonal
    c = 1 / phi.size (``:117-118``).  A non-diagonal C takes the generalised branch (``:122-124,131,139``):
    ``eigh(a=A, b=C)`` and an LU solve with C Q (the branch is unreachable upstream).
    """
    n, D1 = phi.shape
    gammas = gamma_grid(1024, y.dtype) if gammas is None else np.asarray(gammas, dtype=np.float64)
    A, b, sn = primal_gram(phi, y, s)
    F = sn[:, None] * phi
    c = 1.0 / phi.size
    if C is None:
        Cn = c * np.eye(D1)
        lam, Q = sla.eigh(A / c)  # :120
        QHc = Q.conj().T / c  # :121
        modes = Q * (QHc @ b)[None, :]  # :128-129  beta as a function of gamma is modes @ r(gamma)
        h = np.ascontiguousarray(np.real((F @ Q) * (QHc @ F.conj().T).T))  # :136-137,143
    else:
        cd = np.diag(C)
        Cn = C / np.mean(np.abs(cd)) / phi.size  # :117
        lam, Q = sla.eigh(a=A, b=Cn)  # :123
        lu = sla.lu_factor(Cn @ Q)  # :124
        modes = Q * sla.lu_solve(lu, b)[None, :]  # :131
        h = np.ascontiguousarray(np.real((F @ Q) * sla.lu_solve(lu, F.conj().T).T))  # :139
    phib = np.ascontiguousarray(np.real(phi @ modes))  # :134,142
    r = 1.0 / (gammas[None, :] + lam[:, None])  # :147
    with np.errstate(divide="ignore", invalid="ignore"):
        e = (phib @ r - y[:, None]) / (1 - h @ r)  # :149
        yloo = y[:, None] + e  # :150 (before clipping)
    if is_clf:
        e = clip_classifier_residuals(e, y)
    errs, opt, objective = select_gamma(e, sn, is_clf)
    out = {
        "A": A,
        "b": b,
        "lam": lam,
        "gammas": gammas,
        "loo_errors_gammas": errs,
        "objective": objective,
        "opt": opt,
        "gamma": float(gammas[opt]),
        "loo_residuals": e[:, opt],
        "loo_yhat": y + e[:, opt],
        "loo_leverage": h @ r[:, opt],  # :169
        "loo_error": float(errs[opt]),
        "loo_score": weighted_scores(y, yloo[:, opt], sn, is_clf),
    }
    M = gammas[opt] * Cn + A  # :177 (C normalised: c I for the identity)
    L = sla.cho_factor(M)
    beta = sla.cho_solve(L, b)  # :178
    res = np.real(phi @ beta) - y  # :179
    if is_clf:
        res = clip_classifier_residuals(res, y)
    sigma2 = np.real(np.sum(phi * sla.cho_solve(L, phi.conj().T).T, axis=1))  # :184
    loo_sigma2 = sigma2 + (sn * sigma2) ** 2 / (1 - out["loo_leverage"])  # :186
    out.update(beta=beta, L=L[0], L_lower=bool(L[1]), residuals=res, loo_std=np.sqrt(loo_sigma2), s_norm=sn)
    return out


def primal_fit_streamed(
    X: np.ndarray,
    y: np.ndarray,
    s: np.ndarray,
    shift: np.ndarray,
    scale: np.ndarray,
    B: np.ndarray,
    is_clf: bool,
    gammas=None,
    row_tile: int = 8192,
    gamma_index: int | None = None,
    timings: dict | None = None,
) -> dict:
    """Simplified, row-tiled schedule of the primal fit (P0 excluded): SURVEY.md 8(a) P1-P9.

    Algebra (verified against ``primal_fit_faithful`` in tests): with A/c = Q diag(lam) Q^H,
    P = phi Q, v = Q^H b / c:
        (phi beta(gamma))_i = sum_j Re(P_ij v_j) / (gamma + lam_j)            (:128-134)
        h_i(gamma)         = s_i^2 sum_j |P_ij|^2 / (c (gamma + lam_j))       (:136-140)
        sigma2_i           = sum_j |P_ij|^2 / (c (gamma* + lam_j))            (:184, Sherman-Morrison free)
    Only n-vectors and G-vectors survive a tile, so memory is O(row_tile * D).
    ``gamma_index`` forces the selected grid index (used to compare at the reference's argmin).
    """
    import time

    n, d = X.shape
    D = B.shape[1]
    D1 = D + 1
    gammas = gamma_grid(1024, y.dtype) if gammas is None else np.asarray(gammas, dtype=np.float64)
    G = gammas.size
    sn = s / np.sum(s)
    c = 1.0 / (n * D1)
    tm = {} if timings is None else timings

    def tick(key, t0):
        tm[key] = tm.get(key, 0.0) + (time.perf_counter() - t0)

    # Pass 1: A = sum_i s_i^2 phi_i^H phi_i, b = sum_i s_i^2 y_i phi_i^H.
    A = np.zeros((D1, D1), dtype=np.complex128)
    b = np.zeros(D1, dtype=np.complex128)
    for r0 in range(0, n, row_tile):
        r1 = min(n, r0 + row_tile)
        t0 = time.perf_counter()
        phi = feature_map(X[r0:r1], shift, scale, B)
        tick("feature_map", t0)
        t0 = time.perf_counter()
        F = sn[r0:r1, None] * phi
        A += F.conj().T @ F
        b += F.conj().T @ (sn[r0:r1] * y[r0:r1])
        tick("gram", t0)
    A = (A + A.conj().T) / 2
    t0 = time.perf_counter()
    lam, Q = sla.eigh(A / c)
    tick("eigh", t0)
    v = (Q.conj().T @ b) / c
    R = 1.0 / (gammas[None, :] + lam[:, None])  # D1 x G

    # Pass 2: per-gamma error sums; keep numerator / leverage-sum matrices when they fit.
    keep = n * G * 16 <= 6e9
    num_all = np.empty((n, G)) if keep else None
    hs_all = np.empty((n, G)) if keep else None
    errs = np.zeros(G)
    cnt = np.zeros(G)
    hinge = np.zeros(G)

    def tile_quantities(r0, r1, Rm):
        t0 = time.perf_counter()
        phi = feature_map(X[r0:r1], shift, scale, B)
        tick("feature_map", t0)
        t0 = time.perf_counter()
        P = phi @ Q
        U = np.ascontiguousarray(np.real(P * v[None, :]))  # np.real is a strided view (:141-143)
        Gm = np.real(P) ** 2 + np.imag(P) ** 2
        tick("rotate", t0)
        t0 = time.perf_counter()
        num = U @ Rm
        hs = (Gm @ Rm) / c
        tick("sweep", t0)
        return num, hs

    for r0 in range(0, n, row_tile):
        r1 = min(n, r0 + row_tile)
        num, hs = tile_quantities(r0, r1, R)
        if keep:
            num_all[r0:r1] = num
            hs_all[r0:r1] = hs
        with np.errstate(divide="ignore", invalid="ignore"):
            e = (num - y[r0:r1, None]) / (1 - (sn[r0:r1, None] ** 2) * hs)
        if is_clf:
            e = clip_classifier_residuals(e, y[r0:r1])
        abs_e = np.abs(e)
        errs += sn[r0:r1] @ abs_e
        if is_clf:
            cnt += sn[r0:r1] @ (abs_e >= 1)
            hinge += sn[r0:r1] @ np.maximum(0, abs_e - 1)
    objective = cnt + hinge + errs if is_clf else errs
    opt = int(np.argmin(objective)) if gamma_index is None else int(gamma_index)

    # Column of the selected gamma.
    if keep:
        num_o, hs_o = num_all[:, opt], hs_all[:, opt]
    else:
        num_o, hs_o = np.empty(n), np.empty(n)
        for r0 in range(0, n, row_tile):
            r1 = min(n, r0 + row_tile)
            nm, hh = tile_quantities(r0, r1, R[:, opt : opt + 1])
            num_o[r0:r1], hs_o[r0:r1] = nm[:, 0], hh[:, 0]
    lev = sn**2 * hs_o
    with np.errstate(divide="ignore", invalid="ignore"):
        e_raw = (num_o - y) / (1 - lev)
    e_opt = clip_classifier_residuals(e_raw, y) if is_clf else e_raw
    t0 = time.perf_counter()
    M = gammas[opt] * c * np.eye(D1) + A
    L = sla.cho_factor(M)
    beta = sla.cho_solve(L, b)
    tick("cholesky", t0)
    res = np.empty(n)
    for r0 in range(0, n, row_tile):
        r1 = min(n, r0 + row_tile)
        res[r0:r1] = np.real(feature_map(X[r0:r1], shift, scale, B) @ beta) - y[r0:r1]
    if is_clf:
        res = clip_classifier_residuals(res, y)
    sigma2 = hs_o
    loo_sigma2 = sigma2 + (sn * sigma2) ** 2 / (1 - lev)
    return {
        "A": A,
        "b": b,
        "lam": lam,
        "gammas": gammas,
        "loo_errors_gammas": errs,
        "objective": objective,
        "opt": opt,
        "gamma": float(gammas[opt]),
        "loo_residuals": e_opt,
        "loo_yhat": y + e_opt,
        "loo_leverage": lev,
        "loo_error": float(errs[opt]),
        "loo_score": weighted_scores(y, y + e_raw, sn, is_clf),
        "beta": beta,
        "L": L[0],
        "L_lower": bool(L[1]),
        "residuals": res,
        "loo_std": np.sqrt(loo_sigma2),
        "s_norm": sn,
        "timings": tm,
    }


def primal_decision_function(X, shift, scale, B, beta, row_tile: int = 16384) -> np.ndarray:
    """yhat = Re(phi(X) beta): ``_neo_ls_svm.py:661-665``."""
    out = np.empty(X.shape[0])
    for r0 in range(0, X.shape[0], row_tile):
        r1 = min(X.shape[0], r0 + row_tile)
        out[r0:r1] = np.real(feature_map(X[r0:r1], shift, scale, B) @ beta)
    return out


def primal_predict_std(X, shift, scale, B, L, lower: bool = False, row_tile: int = 8192) -> np.ndarray:
    """sigma = sqrt(Re sum phi o cho_solve(L, phi^H)^T): ``_neo_ls_svm.py:464-469,477``."""
    out = np.empty(X.shape[0])
    for r0 in range(0, X.shape[0], row_tile):
        r1 = min(X.shape[0], r0 + row_tile)
        phi = feature_map(X[r0:r1], shift, scale, B)
        out[r0:r1] = np.real(np.sum(phi * sla.cho_solve((L, lower), phi.conj().T).T, axis=1))
    return np.sqrt(out)


# --------------------------------------------------------------------------------------------
# Dual path
# --------------------------------------------------------------------------------------------
def rbf_gram(X: np.ndarray, Y: np.ndarray | None = None, gamma: float = 0.5) -> np.ndarray:
    """exp(-gamma ||x - y||^2) the way sklearn's ``rbf_kernel`` evaluates it.

    sklearn ``euclidean_distances(squared=True)``: row norms + the GEMM expansion
    ``xx - 2 x.y + yy``, clipped at zero, exact-zero diagonal when Y is X; called from
    ``_neo_ls_svm.py:257-261,321,474,669``.
    """
    same = Y is None
    Y = X if same else Y
    xx = np.einsum("ij,ij->i", X, X)[:, None]
    yy = np.einsum("ij,ij->i", Y, Y)[None, :]
    d2 = -2.0 * (X @ Y.T)
    d2 += xx
    d2 += yy
    np.maximum(d2, 0, out=d2)
    if same:
        np.fill_diagonal(d2, 0.0)
    d2 *= -gamma
    return np.exp(d2, out=d2)


def _dual_common(Xt, y, s):
    """``_neo_ls_svm.py:252-268`` with rho = 1 (the only call site, ``:404``), so K = F."""
    sn_sum = s / np.sum(s)
    sn = sn_sum / np.median(np.abs(sn_sum))
    F = rbf_gram(Xt) + 1.0  # :261
    lam, Q = np.linalg.eigh(sn[:, None] * F * sn[None, :])  # :265
    return sn_sum, sn, F, lam, Q


def _dual_finish(Xt, y, s1, sn, F, gammas, yloo, is_clf):
    """Selection, score, Cholesky re-solve, residuals and sigma: ``_neo_ls_svm.py:287-323``."""
    e = yloo - y[:, None]
    if is_clf:
        e = clip_classifier_residuals(e, y)
    errs, opt, objective = select_gamma(e, s1, is_clf)
    M = gammas[opt] * np.diag(sn**-2.0) + F  # :313 with rho = 1
    L = sla.cho_factor(M)
    alpha = sla.cho_solve(L, y)
    res = F @ alpha - y
    if is_clf:
        res = clip_classifier_residuals(res, y)
    Kp = F - 1.0  # rbf kernel without the bias term (:321)
    sigma2 = 1.0 - np.sum(Kp * sla.cho_solve(L, Kp.T).T, axis=1)
    return {
        "gammas": gammas,
        "loo_errors_gammas": errs,
        "objective": objective,
        "opt": opt,
        "gamma": float(gammas[opt]),
        "loo_residuals": e[:, opt],
        "loo_yhat": y + e[:, opt],
        "loo_error": float(errs[opt]),
        "loo_score": weighted_scores(y, yloo[:, opt], s1, is_clf),
        "alpha": alpha,
        "L": L[0],
        "L_lower": bool(L[1]),
        "residuals": res,
        "loo_std": np.sqrt(sigma2),
        "s_norm": s1,
    }


def dual_fit_faithful(Xt: np.ndarray, y: np.ndarray, s: np.ndarray, is_clf: bool, gammas=None) -> dict:
    """The reference's schedule of ``_optimize_alpha_gamma`` incl. the n x G x n tensor (small n only).

    ``_neo_ls_svm.py:252-323``.
    """
    gammas = gamma_grid(128, Xt.dtype) if gammas is None else np.asarray(gammas, dtype=np.float64)
    s1, sn, F, lam, Q = _dual_common(Xt, y, s)
    snQ = sn[:, None] * Q
    modes = snQ * (Q.T @ (sn * y))[None, :]  # :268
    Rg = 1.0 / (gammas[:, None] + lam[None, :])  # G x n
    H = np.einsum("ij,gj,jk->igk", snQ, Rg, snQ.T, optimize="optimal")  # :272-278
    for g in range(gammas.size):
        hd = np.diag(H[:, g, :]).copy()
        hd[hd == 0] = np.finfo(Xt.dtype).eps  # :281
        H[:, g, :] = H[:, g, :] / -hd[:, None]
    F0 = F.copy()
    np.fill_diagonal(F0, 0)
    a_g = modes @ Rg.T  # :285
    yloo = np.sum(F0[:, None, :] * H, axis=2) * a_g + F0 @ a_g  # :286
    out = _dual_finish(Xt, y, s1, sn, F, gammas, yloo, is_clf)
    out["lam"] = lam
    return out


def dual_fit_reduced(Xt: np.ndarray, y: np.ndarray, s: np.ndarray, is_clf: bool, gammas=None) -> dict:
    """Same result as ``dual_fit_faithful`` in 2 n^3 + O(n^2 G) instead of 2 G n^3.

    With W = sn o Q, M = F0 W, R = 1/(gamma + lam):  t = (W o M) R, hd = (W o W) R,
    alpha_g = (W o (Q^T sn y)^T) R and yloo = -(t / hd) o alpha_g + F0 alpha_g.
    """
    gammas = gamma_grid(128, Xt.dtype) if gammas is None else np.asarray(gammas, dtype=np.float64)
    s1, sn, F, lam, Q = _dual_common(Xt, y, s)
    W = sn[:, None] * Q
    F0 = F.copy()
    np.fill_diagonal(F0, 0)
    Mx = F0 @ W
    R = 1.0 / (gammas[None, :] + lam[:, None])  # n x G
    t = (W * Mx) @ R
    hd = (W * W) @ R
    hd[hd == 0] = np.finfo(Xt.dtype).eps
    a_g = (W * (Q.T @ (sn * y))[None, :]) @ R
    yloo = -(t / hd) * a_g + F0 @ a_g
    out = _dual_finish(Xt, y, s1, sn, F, gammas, yloo, is_clf)
    out["lam"] = lam
    return out


def dual_decision_function(Xq: np.ndarray, Xt: np.ndarray, alpha: np.ndarray) -> np.ndarray:
    """k(x, X) alpha + 1' alpha: ``_neo_ls_svm.py:666-671`` (inputs already affine-transformed)."""
    return rbf_gram(Xq, Xt) @ alpha + np.sum(alpha)


def dual_predict_std(Xq: np.ndarray, Xt: np.ndarray, L: np.ndarray, lower: bool = False) -> np.ndarray:
    """sqrt(1 - sum K o cho_solve(L, K^T)^T): ``_neo_ls_svm.py:470-477``."""
    K = rbf_gram(Xq, Xt)
    return np.sqrt(1.0 - np.sum(K * sla.cho_solve((L, lower), K.T).T, axis=1))

