"""Why does the SAME nls_primal_fit call take ~84 ms inside NeoLSSVM.fit at c2 and ~41 ms back to back?

One c2-size problem resident in HBM; the solver call is timed (wall + the library's own HIP-event stage times) after different kinds of
70 ms gaps: none, time.sleep (GPU and host idle), a host NumPy/BLAS load like the pre-step's (threads = the estimator's), the same load
single-threaded, and a sleep during which a trickle of tiny GPU work keeps the queue busy.  Output: one line per variant (median of 7).

    python tools/probe_c2_idle.py > gpurun_out/r05_c2_idle.log
"""

from __future__ import annotations

import statistics
import sys
import threading
import time
from pathlib import Path

import numpy as np

ROOT = Path(__file__).resolve().parent.parent
sys.path.insert(0, str(ROOT))

import bench  # noqa: E402
import neo_ls_svm_amd as hp  # noqa: E402
from neo_ls_svm_amd._prestep import blas_threads  # noqa: E402


def main():
    cfg = bench.CONFIGS["c2"]
    n, d, D = cfg["n"], cfg["d"], cfg["D"]
    ctx = hp.Context(0)
    X, y = bench.synth(n, d, 0, n)
    s = np.ones(n)
    shift, scale, B = bench.affine_params(n, d, D, ctx=ctx)
    dX, dy, ds = ctx.to_device(X), ctx.to_device(y), ctx.to_device(s)
    gam = hp.gamma_grid(1024)
    A = np.random.default_rng(0).standard_normal((1536, 64))
    tiny = ctx.to_device(np.ones(1024))

    def fit():
        t0 = time.perf_counter()
        r = hp.primal_fit(dX, dy, ds, shift, scale, B, False, gammas=gam, ctx=ctx)
        return time.perf_counter() - t0, r["timings"]

    def gap_none():
        pass

    def gap_sleep():
        time.sleep(0.07)

    def gap_blas(threads):
        def g():
            t0 = time.perf_counter()
            with blas_threads(threads):
                while time.perf_counter() - t0 < 0.07:
                    (A @ A.T).sum()
        return g

    def gap_sleep_gpu_trickle():
        stop = threading.Event()

        def trickle():
            while not stop.is_set():
                tiny.to_host()  # a 8 KB D2H copy: a packet on the queue every ~20 us
        th = threading.Thread(target=trickle)
        th.start()
        time.sleep(0.07)
        stop.set()
        th.join()

    def gap_numpy_single():
        t0 = time.perf_counter()
        v = np.arange(200_000, dtype=np.float64)
        while time.perf_counter() - t0 < 0.07:
            np.unique(v)

    for _ in range(3):
        fit()
    variants = [("back to back", gap_none), ("sleep 70 ms", gap_sleep), ("host BLAS 8 threads 70 ms", gap_blas(8)),
                ("host BLAS default threads 70 ms", gap_blas(None)), ("host BLAS 1 thread 70 ms", gap_blas(1)),
                ("numpy single-thread 70 ms", gap_numpy_single), ("sleep 70 ms + GPU trickle", gap_sleep_gpu_trickle)]  # fmt: skip
    keys = ("featuremap", "gram", "evd", "rotate", "sweep", "loo", "cholesky", "download")
    print(f"{'gap before the call':36s} {'wall ms':>8s} {'lib total':>9s} " + " ".join(f"{k:>10s}" for k in keys))
    for name, gap in variants:
        walls, tms = [], []
        for _ in range(7):
            gap()
            w, tm = fit()
            walls.append(w)
            tms.append(tm)
        med = lambda f: statistics.median(f(t) for t in tms)  # noqa: E731
        print(f"{name:36s} {1e3 * statistics.median(walls):8.2f} {1e3 * med(lambda t: t['total']):9.2f} "
              + " ".join(f"{1e3 * med(lambda t, k=k: t[k]):10.2f}" for k in keys), flush=True)
    ctx.close()


if __name__ == "__main__":
    main()
