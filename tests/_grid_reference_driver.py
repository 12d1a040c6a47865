"""TEST INFRASTRUCTURE: the gamma x sigma grid driver of BASELINE config 5 restated in Python over a ``fit`` callable.

The product runs the grid inside the library (``nls_primal_fit_grid`` / ``nls_group_primal_fit_grid``, ``csrc/nls_group.hip``); this is the
round-4 host driver kept as the checker of that C code: the GPU tests run both over the same ``primal_fit`` and compare tables, indices and
the winner bit for bit; the CPU tests compare the library's bookkeeping hooks (``nls_grid_visiting_order``, ``nls_grid_select``) with it."""

from __future__ import annotations

import numpy as np


def visiting_order(sigmas, rank=0, world=1):
    """This rank's sigmas, nearest to 1 first (``_affine_separator.py:200-209``: sigma = 1 is the separator's own bandwidth)."""
    sigmas = np.asarray(sigmas, dtype=np.float64)
    return sorted(range(rank, sigmas.size, world), key=lambda i: (abs(np.log(sigmas[i])), i))


def select(objective, owned, incumbent=None):
    """(sigma_index, gamma_index): first minimum over the owned rows of the row minima; ``incumbent`` (unmerged single rank): a tie that
    includes it goes to it."""
    objective = np.asarray(objective, dtype=np.float64)
    owned = np.asarray(owned, dtype=bool)
    with np.errstate(all="ignore"):
        import warnings

        with warnings.catch_warnings():
            warnings.simplefilter("ignore")
            col_min = np.where(owned, np.nanmin(np.where(owned[:, None], objective, np.inf), axis=1), np.inf)
    k_opt = int(np.argmin(col_min))
    if incumbent is not None and col_min[incumbent] == col_min[k_opt]:
        k_opt = incumbent
    return k_opt, int(np.argmin(objective[k_opt]))


def grid(fit, B, sigmas, gammas, rank=0, world=1, allreduce_sum=None):
    """``fit(B_scaled, finish_below) -> dict`` with loo_errors_gammas, objective, opt, finished, timings (the keys of ``hotpath.primal_fit``)."""
    sigmas = np.asarray(sigmas, dtype=np.float64)
    S, G = sigmas.size, len(gammas)
    table, objective, seconds = np.zeros((S, G)), np.zeros((S, G)), np.zeros(S)
    best = None
    finished_count = 0
    for k in visiting_order(sigmas, rank, world):
        r = fit(B / sigmas[k], None if best is None else best[0])
        table[k], objective[k], seconds[k] = r["loo_errors_gammas"], r["objective"], r["timings"]["total"]
        score = r["objective"][r["opt"]]
        finished_count += bool(r["finished"])
        if r["finished"] and (best is None or score < best[0]):
            best = (score, k, r)
    merged = allreduce_sum is not None and world > 1
    if merged:
        m = allreduce_sum(np.concatenate([table.ravel(), objective.ravel(), seconds]))
        table, objective, seconds = m[: S * G].reshape(S, G), m[S * G : 2 * S * G].reshape(S, G), m[2 * S * G :]
        owned = np.ones(S, dtype=bool)
    else:
        owned = np.zeros(S, dtype=bool)
        owned[rank::world] = True
        table[~owned], objective[~owned] = np.nan, np.nan
    k_opt, g_opt = select(objective, owned, None if merged or best is None else best[1])
    return {
        "loo_errors": table, "objective": objective, "sigma_index": k_opt, "gamma_index": g_opt, "seconds_per_sigma": seconds,
        "finished_count": finished_count, "best": best[2] if (best is not None and best[1] == k_opt) else None,
    }  # fmt: skip
