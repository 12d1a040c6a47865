"""Round-4 investigation of profiles/r03_sigma_overlap.log: two fitting contexts of ONE process on ONE GPU, two host threads, fits in flight
at the same time - must give the bits of the same fits run one after the other, or fail loudly.

    python tools/dev_two_context.py <n> <iterations> [ws_limit_GB]

Every fit is a full ``primal_fit`` (with the factor L_) of the c5 workload shape (d = 128, D = 4096, 32 gammas) at a different sigma."""

from __future__ import annotations

import sys
import threading
import time
import traceback
from pathlib import Path

import numpy as np

ROOT = Path(__file__).resolve().parent.parent
sys.path.insert(0, str(ROOT))

import bench  # noqa: E402
import neo_ls_svm_amd as hp  # noqa: E402


def main():
    n, iters = int(sys.argv[1]), int(sys.argv[2])
    limit = float(sys.argv[3]) if len(sys.argv) > 3 else 0.0
    d, D = (128, 4096) if n >= 50_000 else (32, 512)
    ctxs = [hp.Context(0), hp.Context(0)]
    if limit > 0:
        for c in ctxs:
            c._check(c.lib.nls_set_workspace_limit(c.handle, int(limit * 2**30)))
    X, y = bench.synth(n, d, 0, n)
    s = np.ones(n)
    shift, scale, B = bench.affine_params(n, d, D, ctx=ctxs[0])
    dX, dy, ds = ctxs[0].to_device(X), ctxs[0].to_device(y), ctxs[0].to_device(s)
    gammas = hp.gamma_grid(1024)[::33]
    sigmas = np.logspace(np.log10(0.25), np.log10(4.0), 16)
    order = [int(k) for k in np.argsort(np.abs(np.log(sigmas)))][: 2 * iters]
    keys = ("beta", "lam", "loo_errors_gammas", "loo_residuals", "loo_leverage", "loo_std", "residuals")

    def fit(ctx, k):
        r = hp.primal_fit(dX, dy, ds, shift, scale, B / sigmas[k], False, gammas=gammas, ctx=ctx)
        out = {key: r[key].copy() for key in keys}
        out["L"] = r["L"][np.triu_indices(D + 1)].copy()
        out["opt"] = r["opt"]
        return out

    t0 = time.perf_counter()
    ref = {k: fit(ctxs[0], k) for k in order}
    t_seq = time.perf_counter() - t0
    print(f"sequential: {len(order)} fits in {t_seq:.2f} s", flush=True)
    results, errors = {}, []

    def worker(t):
        for k in order[t::2]:
            try:
                results[k] = fit(ctxs[t], k)
            except Exception as exc:  # noqa: BLE001
                errors.append((t, k, repr(exc)))
                traceback.print_exc()

    t0 = time.perf_counter()
    ths = [threading.Thread(target=worker, args=(t,)) for t in range(2)]
    for th in ths:
        th.start()
    for th in ths:
        th.join()
    t_par = time.perf_counter() - t0
    print(f"two in flight: {t_par:.2f} s, errors: {errors}", flush=True)
    bad = 0
    for k in order:
        if k not in results:
            bad += 1
            continue
        for key in list(keys) + ["L"]:
            if not np.array_equal(results[k][key], ref[k][key]):
                a, b = results[k][key], ref[k][key]
                print(f"MISMATCH sigma index {k} {key}: max rel {np.max(np.abs(a - b)) / np.max(np.abs(b)):.3e}", flush=True)
                bad += 1
        if results[k]["opt"] != ref[k]["opt"]:
            print(f"MISMATCH sigma index {k} opt {results[k]['opt']} vs {ref[k]['opt']}")
            bad += 1
    print(f"n = {n}: {'IDENTICAL' if bad == 0 and not errors else f'{bad} mismatches, {len(errors)} errors'}", flush=True)
    for c in ctxs:
        c.close()


if __name__ == "__main__":
    main()
