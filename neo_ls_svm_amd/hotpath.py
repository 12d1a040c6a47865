"""Host-side mirror of the reference's solver seam, calling the HIP library through the C ABI.

Functions here have the argument meaning of the reference's private solver methods
(``NeoLSSVM._optimize_beta_gamma`` ``_neo_ls_svm.py:77-189``, ``_optimize_alpha_gamma`` ``:191-325``) and
of ``RandomFourierFeatures.transform`` (``_feature_maps.py:153-203``); results come back as a dict
keyed like the fitted attributes the reference sets as side effects (``:146-187``).

All arithmetic happens on the GPU.  NumPy is used for argument marshalling only.
"""

from __future__ import annotations

import ctypes as C

import numpy as np

from . import _lib
from ._hostpool import factor_output
from . import _prestep
from ._prestep import blas_threads
from ._lib import Context, DeviceArray, DualFitArgs, Factor, Group, GroupFactor, PrimalFitArgs, SigmaGrid, default_context, default_group

__all__ = [
    "gamma_grid",
    "orf_frequencies",
    "exact_complexity_matrix",
    "featuremap",
    "gram",
    "rotate",
    "bin_stats",
    "rank_codes",
    "tridiagonalize",
    "eigh",
    "stedc",
    "primal_fit",
    "primal_fit_sharded",
    "primal_fit_sigma_grid",
    "primal_predict",
    "dual_fit",
    "dual_predict",
    "timings_dict",
]


def gamma_grid(num: int) -> np.ndarray:
    """``np.logspace(log10(1e-6), log10(20), num)``: ``_neo_ls_svm.py:146`` (1024) / ``:270`` (128)."""
    return np.logspace(np.log10(1e-6), np.log10(20), num, dtype=np.float64)


def orf_frequencies(d: int, D: int, random_state=42) -> np.ndarray:
    """Orthogonal random frequencies Z (d x D), host side (P0): ``_feature_maps.py:209-223``.

    O(d^2 D) on a d x D matrix; it stays on the host so that the legacy ``RandomState`` stream is the reference's
    (Z then agrees with the reference's to the rounding of LAPACK's QR, which varies with the BLAS thread count).
    """
    gen = random_state if isinstance(random_state, np.random.RandomState) else np.random.RandomState(random_state)
    Z = gen.randn(d, D)
    starts = list(range(0, D, d))

    def block_q(j):
        q, _ = np.linalg.qr(Z[:, j : j + d])
        return q

    # The ceil(D / d) blocks are independent.  A d x d QR on 64+ BLAS threads spends its time in thread hand-offs (0.20 s per 896 x 896 block on
    # the GPU box's host against 0.03 s on 8 threads), and five of them in a row are 0.16 s of a 2.4 s fit: with single-threaded BLAS the
    # blocks run side by side on host threads instead (LAPACK releases the GIL).  The rounding pattern of a block does not depend on how many
    # run at once, only on the BLAS thread count of its own call - fixed at one here when there are several blocks.
    if len(starts) > 1:
        with blas_threads(1):  # (the pre-step's persistent pool: a pool per call costs ~6 ms of thread starts)
            qs = list(_prestep.host_pool().map(block_q, starts))
    else:
        with blas_threads(8):
            qs = [block_q(0)]
    for j, q in zip(starts, qs):
        w = min(d, D - j)
        Z[:, j : j + w] = q[:, :w]
    Z *= np.sqrt(gen.chisquare(d, size=(1, D)))
    return Z


def exact_complexity_matrix(Z) -> np.ndarray:
    """The exact complexity matrix of a random-Fourier map with frequencies Z (d' x D): the slow branch of
    ``_ztz_prod_sinc_zmz`` (``_feature_maps.py:40-55``) embedded as ``complexity_matrix`` does (``:129-135``):
    C[:D, :D] = (Z'Z o prod_k sinc(Z_ki - Z_kj)) / d' (plain sin(x)/x, 1 where |x| <= eps), C[D, D] = 1.
    The reference never reaches it (``fast_approx=True`` is hard-wired); passing it to ``primal_fit`` as
    ``complexity_matrix`` runs the generalised-EVD branch (``_neo_ls_svm.py:122-124``).  Host NumPy, O(d' D^2)."""
    Z = np.asarray(Z, dtype=np.float64)
    dp, D = Z.shape
    Cm = Z.T @ Z
    eps = np.finfo(np.float64).eps
    for k in range(dp):
        dz = Z[k][:, None] - Z[k][None, :]
        with np.errstate(invalid="ignore", divide="ignore"):
            Cm *= np.where(np.abs(dz) > eps, np.sin(dz) / dz, 1.0)
    Cm = (np.tril(Cm) + np.tril(Cm, -1).T) / dp
    out = np.eye(D + 1)
    out[:D, :D] = Cm
    return out


def _f64(a, name, shape=None):
    if isinstance(a, DeviceArray):
        if shape is not None and tuple(a.shape) != tuple(shape):
            raise ValueError(f"{name} has shape {a.shape}, expected {shape}")
        return a
    a = np.ascontiguousarray(a, dtype=np.float64)
    if shape is not None and a.shape != tuple(shape):
        raise ValueError(f"{name} has shape {a.shape}, expected {shape}")
    return a


def _map_params(shift, scale, B, d):
    B = _f64(B, "B")
    if B.ndim != 2 or B.shape[0] != d:
        raise ValueError(f"B must be (d={d}, D), got {B.shape}")
    shift = _f64(np.ravel(np.broadcast_to(np.asarray(shift, dtype=np.float64).ravel(), (d,))), "shift", (d,))
    scale = _f64(np.ravel(np.broadcast_to(np.asarray(scale, dtype=np.float64).ravel(), (d,))), "scale", (d,))
    return shift, scale, B


def timings_dict(t: np.ndarray) -> dict:
    return {k: float(t[i]) for k, i in _lib.TIMING_NAMES.items()}


def featuremap(X, shift, scale, B, ctx: Context | None = None, out: DeviceArray | None = None):
    """phi(X) = [exp(-i ((X - shift)/scale) B)/sqrt(D), 1] as complex128 (n x (D+1)), or into ``out`` (device)."""
    ctx = ctx or default_context()
    X = _f64(X, "X")
    n, d = X.shape
    shift, scale, B = _map_params(shift, scale, B, d)
    D = B.shape[1]
    if out is None:
        phi = np.empty((n, D + 1), dtype=np.complex128)
        target = phi.ctypes.data
    else:
        phi, target = out, out.ptr
    ctx._check(
        ctx.lib.nls_featuremap(ctx.handle, _lib._ptr(X), n, d, shift.ctypes.data, scale.ctypes.data, B.ctypes.data, D, target)
    )
    return phi


def gram(X, y, s, shift, scale, B, ctx: Context | None = None):
    """(A, b) of ``_neo_ls_svm.py:110-114,127`` with phi generated on the fly (never materialised)."""
    ctx = ctx or default_context()
    X = _f64(X, "X")
    n, d = X.shape
    y, s = _f64(y, "y", (n,)), _f64(s, "s", (n,))
    shift, scale, B = _map_params(shift, scale, B, d)
    D1 = B.shape[1] + 1
    A = np.empty((D1, D1), dtype=np.complex128)
    b = np.empty(D1, dtype=np.complex128)
    ctx._check(
        ctx.lib.nls_gram_only(
            ctx.handle, _lib._ptr(X), _lib._ptr(y), _lib._ptr(s), n, d, shift.ctypes.data, scale.ctypes.data,
            B.ctypes.data, B.shape[1], A.ctypes.data, b.ctypes.data,
        )
    )  # fmt: skip
    return A, b


def rotate(X, shift, scale, B, Q, v, ctx: Context | None = None, want_outputs: bool = True):
    """(U, Gm) = (Re(P o v), |P|^2) with P = phi(X) Q: the K4 kernel in isolation (``_neo_ls_svm.py:128-143``)."""
    ctx = ctx or default_context()
    X = _f64(X, "X")
    n, d = X.shape
    shift, scale, B = _map_params(shift, scale, B, d)
    D1 = B.shape[1] + 1
    Q = np.ascontiguousarray(Q, dtype=np.complex128)
    v = np.ascontiguousarray(v, dtype=np.complex128)
    if Q.shape != (D1, D1) or v.shape != (D1,):
        raise ValueError("Q must be (D+1, D+1) and v (D+1,)")
    U = np.empty((n, D1)) if want_outputs else None
    Gm = np.empty((n, D1)) if want_outputs else None
    ctx._check(
        ctx.lib.nls_rotate_only(
            ctx.handle, _lib._ptr(X), n, d, shift.ctypes.data, scale.ctypes.data, B.ctypes.data, B.shape[1],
            Q.ctypes.data, v.ctypes.data, _lib._ptr(U), _lib._ptr(Gm),
        )
    )  # fmt: skip
    return U, Gm


def tridiagonalize(A, ctx: Context | None = None):
    """(d, e, tau, reflectors) of the Householder tridiagonalisation A = Q T Q^H, LAPACK ``zhetrd`` / ``dsytrd``
    conventions with ``uplo='L'`` - the first stage of the eigendecompositions at ``_neo_ls_svm.py:120`` and ``:265``.
    Only the lower triangle of the (complex Hermitian or real symmetric) matrix is read."""
    ctx = ctx or default_context()
    A = np.asarray(A)
    cplx = np.iscomplexobj(A)
    Af = np.asfortranarray(A, dtype=np.complex128 if cplx else np.float64).copy(order="F")
    n = Af.shape[0]
    if Af.shape != (n, n) or n < 1:
        raise ValueError("A must be a non-empty square matrix")
    d, e = np.empty(n), np.zeros(max(n - 1, 1))
    tau = np.zeros(max(n - 1, 1), dtype=Af.dtype)
    ctx._check(ctx.lib.nls_tridiag_only(ctx.handle, Af.ctypes.data, n, int(cplx), d.ctypes.data, e.ctypes.data, tau.ctypes.data))
    return d, e[: n - 1], tau[: n - 1], Af


def eigh(A, ctx: Context | None = None):
    """(eigenvalues ascending, eigenvectors in columns) of a Hermitian / real symmetric matrix given by its lower
    triangle: ``scipy.linalg.eigh`` at ``_neo_ls_svm.py:120`` / ``numpy.linalg.eigh`` at ``:265``."""
    ctx = ctx or default_context()
    A = np.asarray(A)
    cplx = np.iscomplexobj(A)
    Af = np.asfortranarray(A, dtype=np.complex128 if cplx else np.float64).copy(order="F")
    n = Af.shape[0]
    if Af.shape != (n, n) or n < 1:
        raise ValueError("A must be a non-empty square matrix")
    lam = np.empty(n)
    ctx._check(ctx.lib.nls_eigh_only(ctx.handle, Af.ctypes.data, n, int(cplx), lam.ctypes.data))
    return lam, Af


def stedc(d, e, ctx: Context | None = None):
    """(eigenvalues ascending, eigenvectors in columns) of the symmetric tridiagonal matrix diag(d) + offdiag(e): the tridiagonal stage of both
    eigendecompositions alone (test / profiling hook ``nls_stedc_only``)."""
    ctx = ctx or default_context()
    d = np.array(d, dtype=np.float64)
    e = np.ascontiguousarray(e, dtype=np.float64)
    n = d.size
    if n < 1 or e.size != max(n - 1, 0):
        raise ValueError("d must have n >= 1 entries and e n - 1")
    Q = np.empty((n, n), order="F")
    ctx._check(ctx.lib.nls_stedc_only(ctx.handle, d.ctypes.data, e.ctypes.data if n > 1 else None, n, Q.ctypes.data))
    return d, Q


def cholesky(A, ctx: Context | None = None):
    """Lower Cholesky factor of a real symmetric / complex Hermitian positive definite matrix by the library's own factorisations (test /
    profiling hooks ``nls_cholesky_only``: the dual path's, ``nls_zcholesky_only``: the primal path's); only the lower triangle is read.  Raises
    ``numpy.linalg.LinAlgError`` with the index of the first non-positive pivot, as ``numpy.linalg.cholesky`` would."""
    ctx = ctx or default_context()
    A = np.asarray(A)
    cplx = np.iscomplexobj(A)
    if A.ndim != 2 or A.shape[0] != A.shape[1] or A.shape[0] < 1:
        raise ValueError("A must be a non-empty square matrix")
    n = A.shape[0]
    Af = np.asfortranarray(np.tril(A), dtype=np.complex128 if cplx else np.float64)
    info = C.c_int(0)
    fn = ctx.lib.nls_zcholesky_only if cplx else ctx.lib.nls_cholesky_only
    ctx._check(fn(ctx.handle, Af.ctypes.data, n, C.byref(info)))
    if info.value != 0:
        raise np.linalg.LinAlgError(f"matrix is not positive definite: pivot {info.value} <= 0")
    return np.tril(Af)


def twostage_stage(stage: int, A, bw: int, aux=None, ctx: Context | None = None):
    """One stage of the two-stage reduction behind ``eigh`` on host data (test / profiling hook, ``nls_twostage_stage``):
    1: dense -> band ``(A_out, tau1, failed, columns_reduced)``; 2: band -> tridiagonal ``(d, e, V2, timed_out)``;
    3: ``aux <- Q2 aux`` with the chase reflectors ``A`` = V2."""
    ctx = ctx or default_context()
    A = np.asarray(A)
    cplx = np.iscomplexobj(A)
    dt = np.complex128 if cplx else np.float64
    Af = np.asfortranarray(A, dtype=dt).copy(order="F")
    n = Af.shape[0]
    info = np.zeros(2, dtype=np.int32)
    if stage == 1:
        tau1 = np.zeros(n, dtype=dt)
        ctx._check(ctx.lib.nls_twostage_stage(ctx.handle, 1, Af.ctypes.data, n, int(cplx), bw, tau1.ctypes.data, None, None, 0, info.ctypes.data))
        return Af, tau1, bool(info[0]), int(info[1])
    if stage == 2:
        V2 = np.zeros((n, n), dtype=dt, order="F")
        d, e = np.empty(n), np.zeros(max(n - 1, 1))
        ctx._check(ctx.lib.nls_twostage_stage(ctx.handle, 2, Af.ctypes.data, n, int(cplx), bw, V2.ctypes.data, d.ctypes.data, e.ctypes.data, 0, info.ctypes.data))
        return d, e[: n - 1], V2, bool(info[0])
    Cm = np.asfortranarray(aux, dtype=dt).copy(order="F")
    ctx._check(ctx.lib.nls_twostage_stage(ctx.handle, 3, Af.ctypes.data, n, int(cplx), bw, Cm.ctypes.data, None, None, Cm.shape[1], info.ctypes.data))
    return Cm


def bin_stats(X, labels, sample_weight=None, ctx: Context | None = None):
    """(centers, spreads), each nbins x d: per class bin the weighted median and weighted mean absolute deviation of
    every input column (``_affine_normalizer.py:72-79``) on the GPU (``nls_bin_stats_labels``: the rows are grouped by bin with a stable
    radix sort of (label, row) pairs on the device - numpy's ``argsort(labels, kind="stable")`` order - then one segmented radix sort of all
    d x nbins segments and one scan per segment)."""
    ctx = ctx or default_context()
    X = _f64(ctx.held(X), "X")
    n, d = X.shape
    labels = np.asarray(labels)
    lo, hi = int(labels.min()), int(labels.max())
    nbins = hi - lo + 1
    sw = np.ones(n) if sample_weight is None else np.ascontiguousarray(sample_weight, dtype=np.float64)
    lab32 = np.ascontiguousarray(labels - lo, dtype=np.int32)
    centers, spreads = np.empty((nbins, d)), np.empty((nbins, d))
    ctx._check(
        ctx.lib.nls_bin_stats_labels(ctx.handle, _lib._ptr(X), sw.ctypes.data, n, d, lab32.ctypes.data, nbins, centers.ctypes.data, spreads.ctypes.data)
    )
    return centers, spreads


def rank_codes(y, ctx: Context | None = None):
    """(inverse, number of distinct values) of ``numpy.unique(y, return_inverse=True)`` on the GPU (``nls_rank_codes``): the first step of the
    target quantiser (``_quantizer.py:246-253``) - a radix sort of n doubles instead of 48 ms of host sorting at n = 1e6."""
    ctx = ctx or default_context()
    y = np.ascontiguousarray(y, dtype=np.float64).ravel()
    inv = np.empty(y.size, dtype=np.int64)
    nu = C.c_int64()
    ctx._check(ctx.lib.nls_rank_codes(ctx.handle, y.ctypes.data, y.size, inv.ctypes.data, C.byref(nu)))
    return inv, int(nu.value)


def _primal_args(X, y, s, shift, scale, B, is_classifier, gammas, gamma_index, ctx, want_L, want_rows, sweep_only, finish_below,
                 complexity_matrix, residuals_from_sweep=False):
    """Marshal one ``nls_primal_fit_args``: returns (args, out dict of the output arrays, keep-alive tuple)."""
    X = _f64(ctx.held(X), "X")
    if len(X.shape) != 2:
        raise ValueError("X must be 2-D")
    n, d = X.shape
    y, s = _f64(y, "y", (n,)), _f64(s, "s", (n,))
    shift, scale, B = _map_params(shift, scale, B, d)
    D = B.shape[1]
    D1 = D + 1
    gammas = gamma_grid(1024) if gammas is None else np.ascontiguousarray(gammas, dtype=np.float64)
    G = gammas.size
    out = {
        "beta": np.empty(D1, dtype=np.complex128),
        "lam": np.empty(D1),
        "loo_errors_gammas": np.empty(G),
        "objective": np.empty(G),
    }
    if want_L:
        pool_ctx = ctx.contexts[0] if isinstance(ctx, Group) else ctx  # (rank 0 downloads the factor)
        out["L"] = factor_output((D1, D1), np.complex128, pool_ctx)  # only the upper triangle is defined (cho_factor layout; large factors arrive as the triangle alone)
    if want_rows:
        for k in ("loo_residuals", "loo_leverage", "loo_std", "residuals"):
            out[k] = np.empty(n)
    score = C.c_double()
    opt = C.c_int32()
    tm = np.zeros(_lib.NUM_TIMINGS)
    a = PrimalFitArgs()
    a.X, a.y, a.s = _lib._ptr(X), _lib._ptr(y), _lib._ptr(s)
    a.shift, a.scale, a.B, a.gammas = shift.ctypes.data, scale.ctypes.data, B.ctypes.data, gammas.ctypes.data
    a.n, a.d, a.D, a.G = n, d, D, G
    a.is_classifier = 1 if is_classifier else 0
    a.gamma_index_in = -1 if gamma_index is None else int(gamma_index)
    a.flags = (_lib.FIT_SWEEP_ONLY if sweep_only else 0) | (_lib.FIT_FINISH_IF_BELOW if finish_below is not None else 0)
    a.flags |= _lib.FIT_RESIDUALS_FROM_SWEEP if residuals_from_sweep else 0
    a.finish_below = float(finish_below) if finish_below is not None else 0.0
    Cm = None
    if complexity_matrix is not None:
        Cm = np.ascontiguousarray(complexity_matrix, dtype=np.float64)
        if Cm.shape != (D1, D1) or not np.allclose(Cm, Cm.T, rtol=1e-12, atol=0):
            raise ValueError(f"complexity_matrix must be a symmetric ({D1}, {D1}) matrix")
        a.Cmat = Cm.ctypes.data
    finished = C.c_int32(1)
    a.finished = C.addressof(finished)
    a.beta, a.lam = out["beta"].ctypes.data, out["lam"].ctypes.data
    a.L = out["L"].ctypes.data if want_L else None
    a.loo_errors, a.objective = out["loo_errors_gammas"].ctypes.data, out["objective"].ctypes.data
    if want_rows:
        a.loo_residuals, a.loo_leverage = out["loo_residuals"].ctypes.data, out["loo_leverage"].ctypes.data
        a.loo_std, a.residuals = out["loo_std"].ctypes.data, out["residuals"].ctypes.data
    a.loo_score = C.addressof(score)
    a.gamma_index = C.addressof(opt)
    a.timings = tm.ctypes.data
    return a, out, {"X": X, "y": y, "s": s, "shift": shift, "scale": scale, "B": B, "gammas": gammas, "Cm": Cm, "score": score, "opt": opt,
                    "finished": finished, "tm": tm}  # fmt: skip


def _primal_result(out, keep, want_rows):
    gammas, opt, finished = keep["gammas"], keep["opt"], keep["finished"]
    out["gammas"] = gammas
    out["opt"] = int(opt.value)
    out["gamma"] = float(gammas[opt.value])
    out["loo_error"] = float(out["loo_errors_gammas"][opt.value])
    out["finished"] = bool(finished.value)
    out["timings"] = timings_dict(keep["tm"])
    if not out["finished"]:  # the curve only: drop the buffers P8 / P9 would have filled
        for k in ("beta", "L", "loo_residuals", "loo_leverage", "loo_std", "residuals"):
            out.pop(k, None)
        return out
    out["loo_score"] = float(keep["score"].value)
    out["L_lower"] = False
    if want_rows and not isinstance(keep["y"], DeviceArray):
        out["loo_yhat"] = keep["y"] + out["loo_residuals"]
    return out


def primal_fit(
    X,
    y,
    s,
    shift,
    scale,
    B,
    is_classifier: bool,
    gammas=None,
    gamma_index: int | None = None,
    ctx: Context | Group | None = None,
    want_L: bool = True,
    want_rows: bool = True,
    sweep_only: bool = False,
    finish_below: float | None = None,
    complexity_matrix=None,
    residuals_from_sweep: bool = False,
) -> dict:
    """Primal LS-SVM fit with the full gamma sweep (P1-P9).

    X (n x d), y (n), s (n) may be NumPy arrays or ``DeviceArray`` s already resident in HBM.  Returns
    a dict with the reference's attribute names (ASCII): beta, gamma, gammas, opt, loo_errors_gammas,
    loo_residuals, loo_yhat (host input y only), loo_leverage, loo_error, loo_score, L (scipy ``cho_factor``
    format, lower=False), residuals, loo_std, lam, timings.

    ``ctx``: a ``Context`` (one GPU; inside a communicator: this rank's row block of a process-per-GPU launch) or a ``Group``
    (several GPUs in this one call, ``nls_group_primal_fit``: rows sharded over the group's devices inside the library, outputs
    as on one GPU).

    ``sweep_only`` stops after the gamma selection (P1-P7); ``finish_below=t`` runs the Cholesky re-solve and the row
    outputs only when the selected objective is below t (``out["finished"]`` says which) - how a gamma x sigma grid
    avoids finishing sigmas that cannot win.  ``complexity_matrix``: None = identity (the reference's only reachable
    case), else a (D+1) x (D+1) symmetric positive definite matrix -> generalised-EVD branch (``_neo_ls_svm.py:122-124``).
    ``residuals`` is Re(phi beta) - y of the returned beta as the reference computes it (``_neo_ls_svm.py:178-182``: one more pass over
    the feature planes when beta is the Cholesky re-solve); ``residuals_from_sweep=True`` takes the sweep table's column instead
    (the eigendecomposition's beta at gamma*, equal to ~1e-9 relative, no extra pass).
    """
    ctx = ctx or default_context()
    a, out, keep = _primal_args(X, y, s, shift, scale, B, is_classifier, gammas, gamma_index, ctx, want_L, want_rows, sweep_only, finish_below,
                                complexity_matrix, residuals_from_sweep)  # fmt: skip
    fn = ctx.lib.nls_group_primal_fit if isinstance(ctx, Group) else ctx.lib.nls_primal_fit
    ctx._check(fn(ctx.handle, C.byref(a)))
    return _primal_result(out, keep, want_rows)


def primal_fit_sharded(X, y, s, shift, scale, B, is_classifier: bool, devices, **kw) -> dict:
    """``primal_fit`` with the rows sharded over ``devices`` (GPU ordinals) in ONE call of ONE process: the process-wide ``Group`` of that
    device tuple (one context and one host thread per device inside the library, RCCL between them) - SURVEY.md 8(b), 8(e)."""
    return primal_fit(X, y, s, shift, scale, B, is_classifier, ctx=default_group(devices), **kw)


def primal_fit_sigma_grid(
    X,
    y,
    s,
    shift,
    scale,
    B,
    is_classifier: bool,
    sigmas,
    gammas=None,
    ctx: Context | Group | None = None,
    rank: int = 0,
    world: int = 1,
    merge_ctx: Context | None = None,
    want_L: bool = True,
) -> dict:
    """gamma x sigma leave-one-out grid (BASELINE config 5) - ONE C call (``nls_primal_fit_grid``; with a ``Group``:
    ``nls_group_primal_fit_grid``, the sigmas dealt over the group's devices).

    The reference fixes the kernel bandwidth in closed form (``_affine_separator.py:200-209``); the grid extends the search with
    multipliers sigma_k that divide B (T / sigma_k).  For every sigma one fit runs P2-P7 on the gamma grid - ONE eigendecomposition
    per sigma is the factorisation all gammas reuse - and the (sigma, gamma) pair with the smallest selection objective wins.  The
    library visits this rank's sigmas nearest to 1 first, finishes (P8 / P9) only a sigma that beats the incumbent, and resolves
    ties as documented at ``nls_sigma_grid`` in the header; the winner's full result is returned under ``"best"`` (on the rank
    that owns it).

    Default gammas: the 32-point grid ``gamma_grid(1024)[::33]`` (exactly a sub-grid of the reference's 1024 points).
    Process-per-GPU launches: sigmas are dealt round-robin over ``world`` ranks, every rank holding all rows (no collective in the
    data path); ``merge_ctx`` - a SECOND context of this rank that has joined the communicator - merges the small tables.  (A fitting
    context that has joined a communicator turns every fit into a row-sharded collective, so the library refuses that.)
    """
    ctx = ctx or default_context()
    sigmas = np.ascontiguousarray(sigmas, dtype=np.float64)
    gammas = gamma_grid(1024)[::33] if gammas is None else np.ascontiguousarray(gammas, dtype=np.float64)
    a, out, keep = _primal_args(X, y, s, shift, scale, B, is_classifier, gammas, None, ctx, want_L, True, False, None, None)
    S, G = sigmas.size, gammas.size
    table, objective, seconds = np.zeros((S, G)), np.zeros((S, G)), np.zeros(S)
    k_opt, g_opt, best_valid, nfin = C.c_int32(), C.c_int32(), C.c_int32(), C.c_int32()
    tm = np.zeros(_lib.NUM_TIMINGS)
    gr = SigmaGrid()
    gr.sigmas, gr.Sg, gr.rank, gr.world = sigmas.ctypes.data, S, int(rank), int(world)
    gr.merge = merge_ctx.handle if merge_ctx is not None else None
    gr.loo_errors, gr.objective, gr.seconds = table.ctypes.data, objective.ctypes.data, seconds.ctypes.data
    gr.sigma_index, gr.gamma_index = C.addressof(k_opt), C.addressof(g_opt)
    gr.best_valid, gr.finished_count, gr.timings = C.addressof(best_valid), C.addressof(nfin), tm.ctypes.data
    fn = ctx.lib.nls_group_primal_fit_grid if isinstance(ctx, Group) else ctx.lib.nls_primal_fit_grid
    ctx._check(fn(ctx.handle, C.byref(a), C.byref(gr)))
    best = None
    if best_valid.value:
        keep["finished"].value = 1
        keep["opt"].value = g_opt.value
        best = _primal_result(out, keep, True)
        best["timings"] = timings_dict(tm)
    return {
        "sigmas": sigmas,
        "gammas": gammas,
        "loo_errors": table,
        "objective": objective,
        "sigma_index": int(k_opt.value),
        "gamma_index": int(g_opt.value),
        "sigma": float(sigmas[k_opt.value]),
        "gamma": float(gammas[g_opt.value]),
        "seconds_per_sigma": seconds,
        "timings": timings_dict(tm),
        "finished_count": int(nfin.value),
        "best": best,
    }


def primal_predict(X, shift, scale, B, beta=None, L=None, ctx: Context | Group | None = None, factor: Factor | GroupFactor | None = None):
    """(yhat, sigma): ``decision_function`` ``_neo_ls_svm.py:661-665`` and ``predict_std`` ``:464-469,477``.

    Pass ``beta`` for yhat and/or, for sigma, ``L`` (upper factor as returned by ``primal_fit``; uploaded and inverted on
    every call) or ``factor`` (``Factor(ctx, L)``: the inverse kept on the device across calls).
    """
    ctx = ctx or default_context()
    X = _f64(X, "X")
    m, d = X.shape
    shift, scale, B = _map_params(shift, scale, B, d)
    D = B.shape[1]
    yhat = sigma = None
    if beta is not None:
        beta = np.ascontiguousarray(beta, dtype=np.complex128)
        if beta.shape != (D + 1,):
            raise ValueError(f"beta must have shape ({D + 1},)")
        yhat = np.empty(m)
    if factor is not None:
        if factor.ctx is not ctx or factor.D != D or not factor.handle:
            raise ValueError("factor belongs to another context / group / feature count, or is closed")
        L = None
        sigma = np.empty(m)
    elif L is not None:
        L = np.ascontiguousarray(L, dtype=np.complex128)
        if L.shape != (D + 1, D + 1):
            raise ValueError(f"L must have shape ({D + 1}, {D + 1})")
        sigma = np.empty(m)
    fn = ctx.lib.nls_group_primal_predict if isinstance(ctx, Group) else ctx.lib.nls_primal_predict  # (a Group shards the query rows)
    ctx._check(
        fn(
            ctx.handle, _lib._ptr(X), m, d, shift.ctypes.data, scale.ctypes.data, B.ctypes.data, D,
            _lib._ptr(beta), _lib._ptr(L), factor.handle if factor is not None else None, _lib._ptr(yhat), _lib._ptr(sigma),
        )
    )  # fmt: skip
    return yhat, sigma


def dual_fit(
    Xt, y, s, is_classifier: bool, gammas=None, gamma_index: int | None = None, ctx: Context | None = None,
    want_L: bool = True,
) -> dict:  # fmt: skip
    """Dual LS-SVM fit (D1-D5) on affine-transformed rows ``Xt``; weights must be strictly positive."""
    ctx = ctx or default_context()
    Xt = _f64(Xt, "Xt")
    n, r = Xt.shape
    y, s = _f64(y, "y", (n,)), _f64(s, "s", (n,))
    gammas = gamma_grid(128) if gammas is None else np.ascontiguousarray(gammas, dtype=np.float64)
    G = gammas.size
    out = {
        "alpha": np.empty(n),
        "lam": np.empty(n),
        "loo_errors_gammas": np.empty(G),
        "objective": np.empty(G),
        "loo_residuals": np.empty(n),
        "loo_std": np.empty(n),
        "residuals": np.empty(n),
    }
    if want_L:
        out["L"] = factor_output((n, n), np.float64, ctx)  # only the upper triangle is defined (cho_factor layout; large factors arrive as the triangle alone)
    score, opt = C.c_double(), C.c_int32()
    tm = np.zeros(_lib.NUM_TIMINGS)
    a = DualFitArgs()
    a.Xt, a.y, a.s, a.gammas = _lib._ptr(Xt), _lib._ptr(y), _lib._ptr(s), gammas.ctypes.data
    a.n, a.r, a.G = n, r, G
    a.is_classifier = 1 if is_classifier else 0
    a.gamma_index_in = -1 if gamma_index is None else int(gamma_index)
    a.alpha, a.lam = out["alpha"].ctypes.data, out["lam"].ctypes.data
    a.L = out["L"].ctypes.data if want_L else None
    a.loo_errors, a.objective = out["loo_errors_gammas"].ctypes.data, out["objective"].ctypes.data
    a.loo_residuals, a.loo_std, a.residuals = (
        out["loo_residuals"].ctypes.data,
        out["loo_std"].ctypes.data,
        out["residuals"].ctypes.data,
    )
    a.loo_score, a.gamma_index, a.timings = C.addressof(score), C.addressof(opt), tm.ctypes.data
    ctx._check(ctx.lib.nls_dual_fit(ctx.handle, C.byref(a)))
    out["gammas"] = gammas
    out["opt"] = int(opt.value)
    out["gamma"] = float(gammas[opt.value])
    out["loo_error"] = float(out["loo_errors_gammas"][opt.value])
    out["loo_score"] = float(score.value)
    out["L_lower"] = False
    if not isinstance(y, DeviceArray):
        out["loo_yhat"] = y + out["loo_residuals"]
    out["timings"] = timings_dict(tm)
    return out


def dual_predict(Xq, Xt, alpha=None, L=None, ctx: Context | None = None):
    """(yhat, sigma) of the dual model: ``_neo_ls_svm.py:666-671`` and ``:470-477``."""
    ctx = ctx or default_context()
    Xq, Xt = _f64(Xq, "Xq"), _f64(Xt, "Xt")
    m, r = Xq.shape
    n = Xt.shape[0]
    if Xt.shape[1] != r:
        raise ValueError("Xq and Xt must have the same number of columns")
    yhat = sigma = None
    if alpha is not None:
        alpha = _f64(alpha, "alpha", (n,))
        yhat = np.empty(m)
    if L is not None:
        L = _f64(L, "L", (n, n))
        sigma = np.empty(m)
    ctx._check(
        ctx.lib.nls_dual_predict(
            ctx.handle, _lib._ptr(Xq), m, _lib._ptr(Xt), n, r, _lib._ptr(alpha), _lib._ptr(L), _lib._ptr(yhat),
            _lib._ptr(sigma),
        )
    )  # fmt: skip
    return yhat, sigma
