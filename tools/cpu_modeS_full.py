"""ONE full-size CPU run of the oracle's row-streamed schedule (Mode S, ``oracle/neolssvm_oracle.py::primal_fit_streamed``) at BASELINE config 3
- n = 1e6, d = 128, D = 4096, G = 1024 - so that the extrapolation ``bench.py``'s ``cpu_baseline`` makes on every run (a real oracle fit on the
first 65 536 rows, its n-proportional stages x n / 65 536, ``eigh`` / Cholesky unscaled) is checked once against the real thing (VERDICT r04
weak #9).  Off the bench path (~10-15 minutes of host time); needs no GPU work beyond the pre-step's bin statistics.

    python tools/cpu_modeS_full.py > profiles/r05_cpu_modeS_c3_full.json      (on the GPU box: same host as the bench's cpu_baseline)
"""

from __future__ import annotations

import json
import os
import sys
import time
from pathlib import Path

import numpy as np

ROOT = Path(__file__).resolve().parent.parent
sys.path[:0] = [str(ROOT), str(ROOT / "oracle")]

import bench  # noqa: E402
import neolssvm_oracle as orc  # noqa: E402  (test / measurement infrastructure: the CPU baseline, never the product)
from threadpoolctl import threadpool_limits  # noqa: E402


def main():
    cfg = bench.CONFIGS[sys.argv[1] if len(sys.argv) > 1 else "c3"]
    n, d, D, G = cfg["n"], cfg["d"], cfg["D"], cfg["G"]
    import neo_ls_svm_amd as hp

    ctx = hp.Context(0)
    shift, scale, B = bench.affine_params(n, d, D, ctx=ctx)
    ctx.close()
    gammas = hp.gamma_grid(G)
    threads, blas = bench._blas_info()
    out = {"workload": cfg["name"], "cores": int(threads), "host_cpus": os.cpu_count(), "blas": blas, "cores_note": bench._blas_threads_note(int(threads))}
    with threadpool_limits(limits=int(threads), user_api="blas"):
        n_s = 65_536
        Xs, ys = bench.synth(n, d, 0, n_s)
        tm_s = {}
        t0 = time.perf_counter()
        orc.primal_fit_streamed(Xs, ys, np.ones(n_s), shift, scale, B, False, gammas=gammas, row_tile=8192, timings=tm_s)
        t_s = time.perf_counter() - t0
        serial = tm_s.get("eigh", 0.0) + tm_s.get("cholesky", 0.0)
        est = (t_s - serial) * n / n_s + serial
        del Xs
        X, y = bench.synth(n, d, 0, n)
        tm = {}
        t0 = time.perf_counter()
        r = orc.primal_fit_streamed(X, y, np.ones(n), shift, scale, B, False, gammas=gammas, row_tile=8192, timings=tm)
        t_full = time.perf_counter() - t0
    out.update(
        seconds_full_size=t_full,
        fits_per_s_full_size=1.0 / t_full,
        stage_seconds_full_size={k: round(v, 3) for k, v in tm.items()},
        gamma_index_full_size=int(r["opt"]),
        sample_rows=n_s,
        seconds_sample_fit=t_s,
        stage_seconds_sample={k: round(v, 3) for k, v in tm_s.items()},
        seconds_extrapolated_from_sample=est,
        extrapolation_over_measured=est / t_full,
        note="extrapolated = (sample fit - eigh - Cholesky) x n / 65 536 + eigh + Cholesky: what bench.py's cpu_baseline reports at c3 / c5",
    )
    print(json.dumps(out, indent=1))


if __name__ == "__main__":
    main()
