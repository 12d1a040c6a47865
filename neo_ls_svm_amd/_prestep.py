"""Supervised affine pre-step: produces ``shift_``, ``scale_`` and the separator matrix ``A_``.

This is the host-side (NumPy) counterpart of the reference's L2c layer, which runs *before* the hot path and
feeds it.  It restates, in this package's own code, what these reference functions
compute so that ``NeoLSSVM.fit`` can stand alone where upstream ``neo_ls_svm`` is not installed:

    sample bins of the target      _quantizer.py:98-171 (hist_quantized_ecdf), :246-253
    weighted per-bin medians       _weighted_quantile.py:35-77
    shift / scale                  _affine_normalizer.py:50-117
    separator matrix A and lambda  _affine_separator.py:107-210

The random draws use ``numpy.random.RandomState`` in the same order as the reference, so for equal inputs the
outputs agree with it to rounding (pinned by ``tests/test_prestep.py`` against the golden fixtures).
It is O(n d log n) sort/select work on the CPU - SURVEY.md 8(f) lists it as the next row to move to the GPU.
"""

from __future__ import annotations

import os

import numpy as np
from threadpoolctl import ThreadpoolController

__all__ = ["target_bins", "weighted_median_columns", "fit_affine_normalizer", "fit_affine_separator", "blas_threads"]

_controller = None


_pool = None


def host_pool():
    """A process-wide pool of eight host threads for the pre-step's independent small pieces (the separator's bins, the ORF blocks' QRs).
    Persistent: starting threads costs ~2 ms each - 6 of a 100 ms fit at n = 1e5 when a pool is made per call - and the workers spend
    their time in NumPy / LAPACK calls that release the GIL."""
    global _pool
    if _pool is None:
        from concurrent.futures import ThreadPoolExecutor

        _pool = ThreadPoolExecutor(max_workers=8, thread_name_prefix="nls-prestep")
    return _pool


class _Inline:
    """``submit`` that runs the call at once on the calling thread (the pre-step without its pipeline: NLS_PRESTEP_PIPELINE=0)."""

    class _Done:
        def __init__(self, value):
            self.value = value

        def result(self):
            return self.value

    def submit(self, fn, *args):
        return self._Done(fn(*args))


def _pipelined() -> bool:
    """The BLAS-free halves of the separator's direction step - the sampling cdfs of its draws and the element-wise halves of its distance
    matrices - on pool threads beside the calling thread's products.  Same numbers either way (NLS_PRESTEP_PIPELINE=0 / 1 forces).  On by
    default only with at least 32 host cores: with few, the pool threads and the BLAS's own (spinning) workers take each other's cores -
    measured at n = 1e5: 39 -> 25-31 ms per call on the GPU box's 256 cores (before the cdfs moved to the pool too), ~100 -> ~120 ms on 8."""
    e = os.environ.get("NLS_PRESTEP_PIPELINE")
    return e == "1" if e in ("0", "1") else (os.cpu_count() or 1) >= 32


def blas_threads(limit: int):
    """Context manager capping the BLAS threads (threadpoolctl).  The controller is built ONCE: ``threadpool_limits(...)`` re-discovers the
    process's shared objects on every call - ~70 ms each with the ROCm libraries loaded, more than the whole pre-step of a 1e5-row fit."""
    global _controller
    if _controller is None:
        _controller = ThreadpoolController()
    return _controller.limit(limits=limit, user_api="blas")


# --------------------------------------------------------------------------------------------
# Target binning (ECDF quantisation)
# --------------------------------------------------------------------------------------------
def _first_break(ratio_lo, ratio_hi, ratio):
    """Index of the first step at which the running [max lower, min upper] slope band excludes ``ratio`` (or -1)."""
    lo = np.maximum(np.fmax.accumulate(ratio_lo), 0.0)
    hi = np.fmin.accumulate(ratio_hi)
    bad = ~((lo <= ratio) & (ratio <= hi))
    hit = np.flatnonzero(bad)
    return int(hit[0]) if hit.size else -1


def _scan_right(xs, cum, knot, tol, cap):
    """Grow a bin [knot, nxt) to the right until its linear ECDF fit breaks ``tol`` or it exceeds ``cap``.

    ``_quantizer.py:18-44`` (a sequential scan there); here the same decisions from prefix min / max arrays.
    xs/cum carry the -inf/0 and +inf/max sentinels at both ends.  Returns (nxt, count) exactly as the scan does.
    """
    last = len(xs) - 1
    if knot + 1 > last:
        return knot, 0
    base = cum[knot - 1] if knot > 0 else 0
    head = cum[knot:last]  # cum[nxt - 1] for nxt = knot + 1 .. last
    nxt_cap = knot + 1 + int(np.searchsorted(head, base + cap, side="right"))  # first nxt whose count exceeds cap
    stop = min(nxt_cap, last + 1)  # candidates for the slope test: nxt = knot + 2 .. stop - 1
    nxt_tol = -1
    if stop > knot + 2:
        dx = xs[knot + 1:stop - 1] - xs[knot]
        dy = cum[knot + 1:stop - 1] - cum[knot]
        v = _first_break((dy - tol) / dx, (dy + tol) / dx, dy / dx)
        if v >= 0:
            nxt_tol = knot + 2 + v
    if nxt_tol >= 0:
        nxt = nxt_tol
    elif nxt_cap <= last:
        nxt = nxt_cap
    else:
        nxt = last
    return nxt, int(cum[nxt - 1] - base)


def _scan_left(xs, cum, knot, tol, cap):
    """Mirror image of ``_scan_right``: ``_quantizer.py:47-73``."""
    if knot - 1 < 0:
        return knot, 0
    top = cum[knot - 1]
    # count(prv) = top - (cum[prv - 1] if prv > 0 else 0) for prv = knot - 1 .. 0; it exceeds cap below prv_cap
    below = np.concatenate(([0], cum[:knot - 1]))  # index prv -> cum[prv - 1] (0 for prv = 0), prv = 0 .. knot - 1
    prv_cap = int(np.searchsorted(below, top - cap, side="left")) - 1  # largest prv with count > cap, or -1
    start = max(prv_cap + 1, 0)  # candidates for the slope test: prv = knot - 2 .. start (descending)
    prv_tol = -1
    if knot - 2 >= start:
        idx = np.arange(knot - 2, start - 1, -1)
        dx = xs[knot - 1] - xs[idx]
        dy = top - cum[idx]
        v = _first_break((dy - tol) / dx, (dy + tol) / dx, dy / dx)
        if v >= 0:
            prv_tol = knot - 2 - v
    if prv_tol >= 0:
        prv = prv_tol
    elif prv_cap >= 0:
        prv = prv_cap
    else:
        prv = 0
    return prv, int(top - (cum[prv - 1] if prv > 0 else 0))


def _ecdf_bin_edges(values, max_bin_error=0.0125, max_bin_size=0.125, merge_bin_size=0.025):
    """Variable-width bin edges from both ends of the ECDF towards the middle: ``_quantizer.py:98-171``."""
    n = len(values)
    tol, cap, merge = int(max_bin_error * n), int(max_bin_size * n), int(merge_bin_size * n)
    ux, counts = np.unique(values, return_counts=True)
    cum = np.cumsum(counts)
    xs = np.concatenate(([-np.inf], ux, [np.inf]))
    cs = np.concatenate(([0], cum, [np.iinfo(cum.dtype).max]))
    left, right = 1, len(xs) - 1
    edges_l, edges_r = [ux[0]], [ux[-1]]
    edges = None
    while left < right:
        left_prev, right_prev = left, right
        left, _ = _scan_right(xs, cs, left, tol, cap)
        right, _ = _scan_left(xs, cs, right, tol, cap)
        edges_l.append((xs[left] + xs[left - 1]) / 2 if left > 0 else xs[left])
        edges_r.insert(0, (xs[right] + xs[right - 1]) / 2 if right > 0 else xs[right])
        if left == right:
            edges = edges_l + edges_r[1:]
            break
        if left > right:
            edges = edges_l[:-1] + edges_r[1:]
            break
        if cs[right - 1] - cs[left - 1] <= merge:
            mid_l = int(np.floor((left + right) / 2))
            mid_r = int(np.ceil((left + right) / 2))
            edges = edges_l[:-1] + [(xs[mid_l] + xs[mid_r]) / 2] + edges_r[1:]
            break
        del left_prev, right_prev
    if edges is None:  # pragma: no cover - the loop always terminates through one of the breaks
        edges = edges_l + edges_r[1:]
    return np.asarray(edges, dtype=np.float64)


def target_bins(y: np.ndarray, unique=None) -> np.ndarray:
    """Integer class-bin label per sample: ``sample_bins_quantized_ecdf``, ``_quantizer.py:246-253``.

    Few distinct targets (<= ceil(sqrt(n))) are their own bins (classification); otherwise the target's
    ECDF is quantised into dynamically sized bins.  ``unique(y) -> (inverse, number of distinct values)`` replaces
    ``numpy.unique(y, return_inverse=True)`` (``hotpath.rank_codes``: the same codes from a radix sort on the GPU).
    """
    y = np.asarray(y)
    memo = _BINS_MEMO.get("last")  # fit() bins the same target twice (normaliser, separator): reuse the labels
    if memo is not None and memo[0].shape == y.shape and memo[0].dtype == y.dtype and np.array_equal(memo[0], y):
        return memo[1]
    if unique is not None and y.dtype == np.float64 and y.size > 4096:
        inv, nuniq = unique(y)
        uniq = range(nuniq)
    else:
        uniq, inv = np.unique(y, return_inverse=True)
    if len(uniq) <= np.ceil(np.sqrt(len(y))):
        labels = inv
    else:
        edges = _ecdf_bin_edges(inv)  # the reference quantises the rank codes, not the raw values
        labels = np.clip(np.searchsorted(edges, inv, side="right") - 1, 0, len(edges) - 2).astype(np.intp)
    _BINS_MEMO["last"] = (y.copy(), labels)
    return labels


_BINS_MEMO: dict = {}


# --------------------------------------------------------------------------------------------
# Weighted medians
# --------------------------------------------------------------------------------------------
def weighted_median_columns(Xb: np.ndarray, wb: np.ndarray) -> np.ndarray:
    """Weighted 0.5-quantile of every column: ``weighted_quantile(a, w, 0.5, axis=0)``, ``_weighted_quantile.py:35-63``.

    The estimate is the average of the two interpolants through (lower cumulative weight, value) and
    (upper cumulative weight, value).
    """
    A = np.ascontiguousarray(Xb.T)  # one row per feature
    W = np.broadcast_to(np.ravel(wb)[None, :], A.shape)
    order = np.argsort(A, axis=1)
    A = np.take_along_axis(A, order, axis=1)
    W = np.take_along_axis(W, order, axis=1)
    cum = np.cumsum(W, axis=1)
    total = cum[:, [-1]].copy()
    p_lo, p_hi = (cum - W) / total, cum / total
    out = np.empty(A.shape[0], dtype=A.dtype)
    for j in range(A.shape[0]):
        out[j] = (np.interp(0.5, p_lo[j], A[j]) + np.interp(0.5, p_hi[j], A[j])) / 2
    return out[None, :]


# --------------------------------------------------------------------------------------------
# Shift / scale
# --------------------------------------------------------------------------------------------
def _split_bins(X, y, sw):
    labels = target_bins(y)
    masks = [labels == i for i in range(np.min(labels), np.max(labels) + 1)]
    X_bins = [X[m, :] for m in masks]
    n_bins = [np.sum(sw[m]) for m in masks]
    s_bins = [sw[np.newaxis, m] / np.sum(sw[m]) for m in masks]
    return masks, X_bins, n_bins, s_bins


def fit_affine_normalizer(X: np.ndarray, y: np.ndarray, sample_weight: np.ndarray | None = None, stats=None, unique=None):
    """(shift, scale), each 1 x d: ``AffineNormalizer.fit``, ``_affine_normalizer.py:50-117``.

    Per-bin weighted medians and mean absolute deviations; every pair of bins votes for a separating
    threshold (shift) and a spread (scale) with weight sqrt((n_i + n_j) (1/2 + separability)).
    ``stats(X, labels, sw) -> (centers, spreads)`` (nbins x d each) replaces the NumPy sort-based statistics -
    ``hotpath.bin_stats`` computes them on the GPU.
    """
    X = np.asarray(X)
    y = np.ravel(np.asarray(y)).astype(X.dtype)
    sw = (np.ones(y.shape) if sample_weight is None else np.ravel(np.asarray(sample_weight))).astype(y.dtype)
    d = X.shape[1]
    if stats is not None:
        labels = target_bins(y, unique)
        nb = int(labels.max() - labels.min()) + 1
        if nb <= 1:
            return np.zeros((1, d), dtype=X.dtype), np.ones((1, d), dtype=X.dtype)
        cen, spr = stats(X, labels, sw)
        n_bins = np.bincount(labels - labels.min(), weights=sw, minlength=nb)
        centers = [cen[b][None, :] for b in range(nb)]
        spreads = [spr[b][None, :] for b in range(nb)]
    else:
        _, X_bins, n_bins, s_bins = _split_bins(X, y, sw)
        if len(X_bins) <= 1:
            return np.zeros((1, d), dtype=X.dtype), np.ones((1, d), dtype=X.dtype)
        centers = [weighted_median_columns(Xb, sb) for Xb, sb in zip(X_bins, s_bins)]
        spreads = [sb @ np.abs(Xb - mu) for Xb, sb, mu in zip(X_bins, s_bins, centers)]
    eps = np.finfo(X.dtype).eps
    sign = np.zeros((1, d), dtype=X.dtype)
    wsum = np.zeros((1, d), dtype=X.dtype)
    shift = np.zeros((1, d), dtype=X.dtype)
    scale = np.zeros((1, d), dtype=X.dtype)
    for i in range(len(centers) - 1):
        for j in range(i + 1, len(centers)):
            dmu = centers[j] - centers[i]
            ssum = np.maximum(spreads[i] + spreads[j], eps)
            w = np.sqrt((n_bins[i] + n_bins[j]) * (0.5 + np.abs(dmu) / ssum))
            alpha = np.clip(spreads[i] / ssum, 1e-6, 1.0 - 1e-6)
            shift = shift + w * (centers[i] + alpha * dmu)
            scale = scale + w * ssum
            sign += w * np.sign(dmu)
            wsum += w
    sign /= wsum
    shift, scale = shift / wsum, scale / wsum
    scale[np.sign(sign) < 0] = -scale[np.sign(sign) < 0]
    return shift, scale


# --------------------------------------------------------------------------------------------
# Separator matrix
# --------------------------------------------------------------------------------------------
def _sq_dists(P, Q):
    """||p_i - q_j||^2 by the expansion (|p|^2 - 2 p.q) + |q|^2, in the reference's order of operations (``_affine_separator.py:24-29``:
    the nearest-neighbour argmin over near-ties depends on the rounding) but in ONE buffer: the three n x m temporaries of the plain
    expression cost more than the product itself at these sizes (384 ... 1536 rows)."""
    return _sq_dists_finish(P @ Q.T, P, Q)


def _sq_dists_finish(D, P, Q):
    """The element-wise half of ``_sq_dists`` on the product D = P Q^T (in place): no BLAS call in here, so it may run on a pool thread
    beside the calling thread's products."""
    D *= 2.0  # exact
    np.subtract(np.sum(P * P, axis=1, keepdims=True), D, out=D)
    D += np.sum(Q * Q, axis=1, keepdims=True).T
    return D


def _nearest_finish(D, P, Q):
    idx = np.argmin(_sq_dists_finish(D, P, Q), axis=1, keepdims=True)
    return np.take_along_axis(Q, idx, axis=0)


def _nearest_rows(P, Q):
    """Rows of Q nearest to each row of P: ``_affine_separator.py:24-29``."""
    return _nearest_finish(P @ Q.T, P, Q)


def _right_singular_vectors(M):
    """Singular values (descending) and right singular vectors via an eigendecomposition: ``:32-51``."""
    if M.shape[0] >= M.shape[1]:
        e, V = np.linalg.eigh(M.conj().T @ M)
        return np.sqrt(np.abs(e))[::-1], V[:, ::-1]
    e, U = np.linalg.eigh(M @ M.conj().T)
    sv = np.sqrt(np.abs(e))[::-1]
    U = U[:, ::-1]
    keep = sv > 0
    sv, U = sv[keep], U[:, keep]
    return sv, (M.conj().T @ U) / sv[np.newaxis, :]


def fit_affine_separator(
    X: np.ndarray,
    y: np.ndarray,
    sample_weight: np.ndarray | None = None,
    *,
    rank_threshold: float = 2e-2,
    edge_sample_size: int = 384,
    edge_search_multiplier: int = 4,
    random_state=42,
    normalizer=None,
    unique=None,
):
    """(shift, scale, A): ``AffineSeparator.fit``, ``_affine_separator.py:107-210``.  A is None for one bin.

    For every class bin: sample edge points of the bin and of its complement (weighted draws), pair each
    with its nearest neighbour on the other side, and keep the leading right singular vectors of the
    difference matrix.  The concatenated directions are scaled by lambda = sqrt(2 log(f/g) / (f - g)) with f / g
    the mean inter- / intra-bin squared distances of the edge samples.

    Only the sampled rows are ever gathered and normalised (the reference materialises X per bin and per
    complement, ~n d nbins doubles of copies); the random stream and the probabilities are the reference's, so
    the draws are identical.  ``normalizer(X, y, sw) -> (shift, scale)`` overrides the shift/scale step (the GPU
    implementation plugs in here).
    """
    X = np.asarray(X)
    y = np.ravel(np.asarray(y)).astype(X.dtype)
    shift, scale = (normalizer or fit_affine_normalizer)(X, y, sample_weight)
    # (the edge-sample products are a few hundred rows wide: on a 64-thread BLAS they spend their time in thread hand-offs - 13 ms per
    # 1536 x 128 x 1536 product against 3 ms on 8 threads; since round 6 the bins run side by side, one BLAS thread each)
    with blas_threads(8):
        return _separator_directions(X, y, sample_weight, shift, scale, rank_threshold, edge_sample_size, edge_search_multiplier, random_state, unique)


def _separator_directions(X, y, sample_weight, shift, scale, rank_threshold, edge_sample_size, edge_search_multiplier, random_state, unique=None):
    sw = (np.ones(y.shape) if sample_weight is None else np.ravel(np.asarray(sample_weight))).astype(y.dtype)
    labels = target_bins(y, unique)
    ids = [np.flatnonzero(labels == i) for i in range(np.min(labels), np.max(labels) + 1)]
    if len(ids) <= 1:
        return shift, scale, None
    n_bins = [np.sum(sw[ix]) for ix in ids]
    p_bins = [sw[ix] / np.sum(sw[ix]) for ix in ids]

    def rows(ix):  # normalised rows, gathered on demand
        return ((X[ix, :] - shift) / scale).astype(X.dtype)

    m = int(edge_sample_size * 4 / 3) if len(ids) == 2 else edge_sample_size
    gen = random_state if isinstance(random_state, np.random.RandomState) else np.random.RandomState(random_state)
    # The draws first, in the reference's order (one random stream: seeds, outside candidates, inside candidates per bin).  The reference calls
    # RandomState.choice(len, size, p), which is  cdf = p.cumsum(); cdf /= cdf[-1]; idx = cdf.searchsorted(random_sample(size), "right")
    # (numpy/random/mtrand.pyx, legacy sampling with replacement): the cdf halves - a concatenation of n indices, three n-long element-wise
    # passes and a cumsum per bin, half of this function's time at n = 1e5 - consume no random numbers, so they are prepared side by side on
    # the pool and only the random_sample calls follow the stream's order.  Same draws, bit for bit (the fixtures pin B).
    pool = host_pool() if _pipelined() else _Inline()

    def cdf_of(p):
        cdf = p.cumsum()
        cdf /= cdf[-1]
        return cdf

    def prepare(i):  # (no BLAS in here)
        rest = np.concatenate([ix for j, ix in enumerate(ids) if j != i])
        s_rest = sw[rest]
        return rest, cdf_of(s_rest / np.sum(s_rest)), cdf_of(np.asarray(p_bins[i], dtype=np.float64))

    prepared = [pool.submit(prepare, i) for i in range(len(ids))]
    draws = []
    for i in range(len(ids)):
        rest, cdf_rest, cdf_in = prepared[i].result()
        seeds_ix = ids[i][cdf_in.searchsorted(gen.random_sample(m), side="right")]
        cand_ix = rest[cdf_rest.searchsorted(gen.random_sample(m * edge_search_multiplier), side="right")]
        cand_in_ix = ids[i][cdf_in.searchsorted(gen.random_sample(m * edge_search_multiplier), side="right")]
        draws.append((seeds_ix, cand_ix, cand_in_ix))

    # ... then the bins pipelined over host threads.  Every BLAS / LAPACK call stays on THIS thread with the caller's thread count: the
    # nearest-neighbour argmin sits on near-ties whose outcome follows the rounding of the distance product, and that follows how the product
    # is split over BLAS threads (under a one-thread product half of the ames-shaped fixture's B changes) - and several multi-threaded BLAS calls
    # at once fight over the cores (measured: 47 -> 650 ms).  What runs beside them on the pool is the element-wise half of each distance
    # matrix (scale, two broadcast adds, argmin, gather: 0.8 of each 1 ms call at these sizes): the same operations on the same numbers.
    nb_ = len(ids)
    first = []
    for seeds_ix, cand_ix, _ in draws:
        P, Q = rows(seeds_ix), rows(cand_ix)
        first.append(pool.submit(_nearest_finish, P @ Q.T, P, Q))
    second, edge_out = [], []
    for k, (_, _, cand_in_ix) in enumerate(draws):
        outside = first[k].result()
        edge_out.append(outside)
        Q = rows(cand_in_ix)
        second.append(pool.submit(_nearest_finish, outside @ Q.T, outside, Q))
    edge_in, dirs = [], []
    for k in range(nb_):
        inside = second[k].result()
        edge_in.append(inside)
        sv, V = _right_singular_vectors(inside - edge_out[k])
        dirs.append(V[:, : int(np.sum(sv > rank_threshold * sv[0]))])
    A = np.hstack(dirs)
    n_inter, n_intra = m * (m + 1) / 2, m * (m - 1) / 2

    def spread(D, P, Q, k, norm):
        return np.sum(np.tril(_sq_dists_finish(D, P, Q), k=k)) / norm

    futs = []
    for k in range(nb_):
        ein, eout = edge_in[k] @ A, edge_out[k] @ A
        futs.append((pool.submit(spread, ein @ eout.T, ein, eout, 0, n_inter), pool.submit(spread, ein @ ein.T, ein, ein, -1, n_intra)))
    parts = [(fi.result(), fa.result()) for fi, fa in futs]
    inter = intra = 0.0
    for (pi, pa), nb in zip(parts, n_bins):  # (summed in bin order, as the reference's loop does)
        inter += nb * pi
        intra += nb * pa
    inter /= sum(n_bins)
    intra /= sum(n_bins)
    lam = np.sqrt(2 * np.log(inter / intra) / (inter - intra)) if intra > 0 else 1
    return shift, scale, A * lam
