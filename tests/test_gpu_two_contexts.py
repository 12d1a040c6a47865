"""Two fitting contexts of ONE process on ONE GPU, fits in flight at the same time from two host threads.

Round 3 recorded silently wrong Cholesky factors (and ``zpotrf info != 0``) in this mode (``profiles/r03_sigma_overlap.log``).  Round 4 traced
it to rocSOLVER / rocBLAS: ``rocsolver_zpotrf`` + ``rocblas_ztrsm`` / ``zherk`` on two handles at the same time corrupt each other's results
(``tools/probe_rocsolver_concurrency.cpp``, ``profiles/r04_two_contexts.md``); the fit path no longer calls them (own factorisation,
``csrc/nls_zpotrf.h``).  The bar here: every output of every concurrent fit equals, bit for bit, the same fit run alone."""

from __future__ import annotations

import sys
import threading
from pathlib import Path

import numpy as np
import pytest

ROOT = Path(__file__).resolve().parent.parent
sys.path.insert(0, str(ROOT))

pytestmark = pytest.mark.gpu


def _run_two(ctxs, jobs, fit):
    """jobs: list of job keys; thread t takes jobs[t::2] on ctxs[t].  Returns ({key: result}, errors)."""
    results, errors = {}, []

    def worker(t):
        for k in jobs[t::2]:
            try:
                results[k] = fit(ctxs[t], k)
            except Exception as exc:  # noqa: BLE001
                errors.append((t, k, repr(exc)))

    ths = [threading.Thread(target=worker, args=(t,)) for t in range(2)]
    for th in ths:
        th.start()
    for th in ths:
        th.join()
    return results, errors


def test_two_contexts_primal_fits_in_flight_equal_sequential_fits():
    """20 fits of the c3e shape (n = 125 000 rows: what one rank of an 8-GPU c3 fit holds; d = 128, D = 4096, 32 gammas), each at a different
    kernel width and each with its factor L_, ten per context, two in flight."""
    import bench
    import neo_ls_svm_amd as hp

    n, d, D = 125_000, 128, 4096
    ctxs = [hp.Context(0), hp.Context(0)]
    try:
        X, y = bench.synth(n, d, 0, n)
        s = np.ones(n)
        shift, scale, B = bench.affine_params(n, d, D, ctx=ctxs[0])
        dX, dy, ds = ctxs[0].to_device(X), ctxs[0].to_device(y), ctxs[0].to_device(s)
        gammas = hp.gamma_grid(1024)[::33]
        sigmas = np.logspace(np.log10(0.3), np.log10(3.0), 20)
        keys = ("beta", "lam", "loo_errors_gammas", "loo_residuals", "loo_leverage", "loo_std", "residuals")
        iu = np.triu_indices(D + 1)

        def fit(ctx, k):
            r = hp.primal_fit(dX, dy, ds, shift, scale, B / sigmas[k], False, gammas=gammas, ctx=ctx)
            out = {key: r[key].copy() for key in keys}
            out["L"], out["opt"] = r["L"][iu].copy(), r["opt"]
            return out

        jobs = list(range(20))
        ref = {k: fit(ctxs[k % 2], k) for k in jobs}  # one at a time (each on the context that will run it concurrently)
        got, errors = _run_two(ctxs, jobs, fit)
        assert not errors, errors
        for k in jobs:
            assert got[k]["opt"] == ref[k]["opt"], k
            for key in list(keys) + ["L"]:
                assert np.array_equal(got[k][key], ref[k][key]), (k, key, float(np.max(np.abs(got[k][key] - ref[k][key]))))
    finally:
        for c in ctxs:
            c.close()


def test_two_contexts_dual_fits_in_flight_equal_sequential_fits():
    """The dual path the same way: 8 fits at n = 2500 (one-stage real eigendecomposition, own Cholesky factorisation), four per context."""
    import neo_ls_svm_amd as hp

    ctxs = [hp.Context(0), hp.Context(0)]
    try:
        rng = np.random.default_rng(7)
        n, r_ = 2500, 24
        Xt = rng.standard_normal((n, r_)) * 0.4
        w = rng.standard_normal(r_)
        s = rng.uniform(0.5, 2.0, n)
        ys = [np.sin((1.0 + 0.2 * k) * Xt @ w) + 0.1 * rng.standard_normal(n) for k in range(8)]
        keys = ("alpha", "lam", "loo_errors_gammas", "loo_residuals", "loo_std", "residuals")
        iu = np.triu_indices(n)

        def fit(ctx, k):
            r = hp.dual_fit(Xt * (1.0 + 0.1 * k), ys[k], s, False, ctx=ctx)
            out = {key: r[key].copy() for key in keys}
            out["L"], out["opt"] = r["L"][iu].copy(), r["opt"]
            return out

        jobs = list(range(8))
        ref = {k: fit(ctxs[k % 2], k) for k in jobs}
        got, errors = _run_two(ctxs, jobs, fit)
        assert not errors, errors
        for k in jobs:
            assert got[k]["opt"] == ref[k]["opt"], k
            for key in list(keys) + ["L"]:
                assert np.array_equal(got[k][key], ref[k][key]), (k, key)
    finally:
        for c in ctxs:
            c.close()
