"""Headline benchmark: primal fits/s with the full gamma sweep on synthetic n x d data (BASELINE.json).

    python bench.py [--gpus N] [--steps K] [--warmup W] [--config c3|c2|c5|...] [--no-cpu-baseline]

One step = one ``nls_primal_fit`` call (P1-P9: feature map, Hermitian Gram, EVD, rotation, gamma sweep over G = 1024,
selection, Cholesky re-solve, residuals, LOO sigma) with X, y, s already resident in HBM.  Everything is ctypes + the
HIP library; there is no PyTorch anywhere.  N > 1: one process per GPU - either launched by a launcher that sets
RANK / LOCAL_RANK / WORLD_SIZE / MASTER_PORT (``python -m torch.distributed.run ... bench.py --gpus N``; only its
environment variables are used) or, with no such environment, spawned by this script itself before any GPU call.
The n rows are sharded over the ranks (strong scaling at fixed n) and the library's exchange points run as RCCL
collectives on its own stream (``nls_comm_init_rank``; the 128-byte communicator id travels through a file in /tmp).
Rank 0 prints ONE JSON line.
"""

from __future__ import annotations

import argparse
import json
import os
import subprocess
import sys
import time
from pathlib import Path

ROOT = Path(__file__).resolve().parent
sys.path.insert(0, str(ROOT))
# rocBLAS's own process-wide switch (real GEMMs through hipBLASLt: 5.5 ms of a c4 fit): exported by this launcher for its own process and the
# ranks it spawns - the library itself does not touch the environment; a caller's explicit choice stands
os.environ.setdefault("ROCBLAS_USE_HIPBLASLT", "1")

CONFIGS = {
    # BASELINE.json configs[2] - the configuration the metric is quoted on
    "c3": dict(n=1_000_000, d=128, D=4096, G=1024, name="synthetic regression n=1e6 d=128 D=4096 ORF, primal, G=1024"),
    # BASELINE.json configs[1]
    "c2": dict(n=100_000, d=64, D=1024, G=1024, name="synthetic regression n=1e5 d=64 D=1024 ORF, primal, G=1024"),
    # one eighth of c3: what one rank of an 8-GPU row-sharded c3 fit computes locally (scaling diagnostics)
    "c3e": dict(n=125_000, d=128, D=4096, G=1024, name="synthetic regression n=1.25e5 d=128 D=4096 ORF, primal, G=1024"),
    # BASELINE.json configs[4]: gamma x sigma grid (32 x 16); one eigendecomposition per sigma serves all 32 gammas.
    # N > 1 shards the SIGMAS (every rank holds all rows, no collective in the data path) -> "weak"-style replicas.
    "c5": dict(n=1_000_000, d=128, D=4096, G=32, sigmas=16, name="gamma x sigma LOO grid 32 x 16, n=1e6 d=128 D=4096 ORF, primal"),
    # one row chunk of c3 (profiling: same kernels, same D, 1/4 of the rows)
    "c3q": dict(n=262_144, d=128, D=4096, G=1024, name="synthetic regression n=262144 d=128 D=4096 ORF, primal, G=1024"),
    # BASELINE.json configs[3]: dual path, binary classification, explicit n x n kernel + eigendecomposition gamma-sweep (G = 128)
    "c4": dict(n=10_000, d=256, G=128, dual=True, name="synthetic binary classification n=1e4 d=256, dual path (n x n RBF kernel, EVD gamma-sweep G=128)"),
    # c3's shape at n / D = 16 rows per feature: with this generator the LOO-optimal gamma is INTERIOR to the grid (index ~ 675 of 1024; at c2 / c3
    # - 98 / 244 rows per feature at 10 % noise - it is the smallest grid point), so gamma selection and the Cholesky re-solve run at an interior
    # gamma*.  A secondary line (profiles/), not the headline.
    "c3i": dict(n=65_536, d=128, D=4096, G=1024, name="synthetic regression n=65536 d=128 D=4096 ORF (16 rows per feature: interior LOO optimum), primal, G=1024"),
    # config 5 in miniature (tests: sigma sharding through the native communicator at world 8 on one GPU)
    "c5s": dict(n=20_000, d=32, D=512, G=32, sigmas=16, name="gamma x sigma LOO grid 32 x 16, n=2e4 d=32 D=512 ORF, primal"),
    # small plumbing configuration for quick checks
    "c0": dict(n=20_000, d=32, D=512, G=1024, name="synthetic regression n=2e4 d=32 D=512 ORF, primal, G=1024"),
}
FP64_MFMA_PEAK_TFLOPS = 78.6  # MI355X dense fp64 matrix peak; measured issue rate 78.0 (profiles/r01_probe_mfma.log)
HBM_PEAK_GBS = 8000.0  # MI355X HBM3E (MI355X_MICROARCH.md); a plain 16-byte copy reaches 4.6 TB/s (profiles/r01_probe_hw.log)
K1_VALU_PER_FEATURE = 36.6  # vector instructions per feature in k_featuremap's epilogue (34.9 sincos + scale + stores' addresses, 1.7 range test): tools/k1_valu_count.py
PRESTEP_PREFIX = 200_000  # for n = 1e6 the affine pre-step is fitted on a fixed 2e5-row prefix


def synth(n, d, lo, hi):
    """Rows [lo, hi) of the SURVEY 8(d) generator (default_rng(0); X ~ N(0,1); y = sin(Xw) + 0.1 eps)."""
    import numpy as np

    rng = np.random.default_rng(0)
    w = rng.standard_normal(d) / np.sqrt(d)
    # Row blocks come from independent child streams so that every rank can build just its shard.
    blk = 65536
    Xs, ys = [], []
    for b0 in range(lo - lo % blk, hi, blk):
        r = np.random.default_rng([0, b0 // blk])
        Xb = r.standard_normal((min(blk, n - b0), d))
        yb = np.sin(Xb @ w) + 0.1 * r.standard_normal(Xb.shape[0])
        a, e = max(lo, b0) - b0, min(hi, b0 + blk) - b0
        Xs.append(Xb[a:e])
        ys.append(yb[a:e])
    return np.ascontiguousarray(np.vstack(Xs)), np.concatenate(ys)


def affine_params(n, d, D, ctx=None):
    """(shift, scale, B) as SURVEY.md 8(d) prescribes: the package's own supervised pre-step (AffineSeparator + ORF
    frequencies, RandomState(42)) fitted on the first min(n, 2e5) rows of the workload.  Deterministic: every rank
    computes the same parameters."""
    from neo_ls_svm_amd import OrthogonalRandomFourierFeatures

    p = min(n, PRESTEP_PREFIX)
    Xp, yp = synth(n, d, 0, p)
    fm = OrthogonalRandomFourierFeatures(num_features=D).fit(Xp, yp, None, ctx=ctx)
    shift, scale, B = fm.map_params
    return shift.copy(), scale.copy(), B.copy()


def _blas_info():
    try:
        from threadpoolctl import threadpool_info

        infos = [i for i in threadpool_info() if i.get("user_api") == "blas"]
        return max([i.get("num_threads", 1) for i in infos] or [1]), ",".join(sorted({str(i.get("internal_api")) for i in infos})) or "unknown"
    except Exception:
        return os.cpu_count(), "unknown"


def _blas_threads_note(threads):
    """Why the CPU baseline runs on `threads` BLAS threads and not on every CPU of the box."""
    cpus = os.cpu_count() or 1
    if threads >= cpus:
        return f"all {cpus} CPUs"
    try:
        from threadpoolctl import threadpool_info

        ver = ",".join(sorted({f"{i.get('internal_api')} {i.get('version')} ({i.get('threading_layer')}, {i.get('architecture')})" for i in threadpool_info() if i.get("user_api") == "blas"}))
    except Exception:
        ver = "unknown"
    return (f"{threads} of {cpus} CPUs: the thread count NumPy's bundled BLAS ({ver}) starts with on this box - its build-time cap (the OpenBLAS wheel "
            f"is compiled with NUM_THREADS = 64; threadpoolctl cannot raise it) - not a measured knee")


def _parity(gpu, ref, rows):
    """BASELINE.json's second metric half: GPU fit against the float64 oracle fit on the SAME rows -
    max |e - e_ref| / max |e_ref| on ``loo_residuals_`` at the reference's argmin, ||beta - beta_ref|| / ||beta_ref||,
    max-rel on the whole per-gamma error curve (``_neo_ls_svm.py:146-167``).  Bar: 1e-5."""
    import numpy as np

    def mx(a, b):
        return float(np.max(np.abs(np.asarray(a) - np.asarray(b))) / max(float(np.max(np.abs(b))), 1e-300))

    return {
        "rows": int(rows),
        "argmin_equal": bool(gpu["opt"] == ref["opt"]),
        "argmin_gpu_ref": [int(gpu["opt"]), int(ref["opt"])],
        "loo_residuals_max_rel_err": mx(gpu["loo_residuals"], ref["loo_residuals"]),
        "beta_rel_err": float(np.linalg.norm(gpu["beta"] - ref["beta"]) / np.linalg.norm(ref["beta"])),
        "loo_errors_max_rel_err": mx(gpu["loo_errors_gammas"], ref["loo_errors_gammas"]),
        "loo_leverage_max_rel_err": mx(gpu["loo_leverage"], ref["loo_leverage"]),
        "loo_std_max_rel_err": mx(gpu["loo_std"], ref["loo_std"]),
        "residuals_max_rel_err": mx(gpu["residuals"], ref["residuals"]),
        "bar": 1e-5,
    }


def cpu_baseline(cfg, shift, scale, B, gammas, gpu_fit, gpu_full=None):
    """The oracle (NumPy restatement of the reference's algorithm, ``kind: port``) timed on this box's host cores, and the
    parity of the GPU fit against it on the same rows.

    * Where the reference's own schedule fits host RAM (n (D+1) 16 B of phi and its five zgemm-class products: c2),
      Mode R (``primal_fit_faithful``, the schedule of ``_neo_ls_svm.py:112-187``) and Mode S (``primal_fit_streamed``, the
      simplified row-streamed schedule the HIP library implements) are both run at FULL size; parity = the timed GPU
      step's outputs (``gpu_full``) against Mode S on all rows.
    * Otherwise (c3 / c5: phi alone is 65.5 GB) a REAL oracle fit - ``primal_fit_streamed``: feature map, Gram, ``eigh`` of the
      (D+1)^2 matrix, rotation, full gamma sweep, selection, ``cho_factor`` / ``cho_solve`` - runs on a bounded row sample of the
      workload at full d, D, G; its n-proportional stage times are scaled to n rows and the n-independent ``eigh`` /
      Cholesky added unscaled; parity = ``gpu_fit(X, y, s)`` on the same sample against that oracle fit.
    BLAS threads are pinned with threadpoolctl and stated.  Returns (cpu_baseline dict, parity dict).
    """
    import numpy as np
    from threadpoolctl import threadpool_limits

    sys.path.insert(0, str(ROOT / "oracle"))
    import neolssvm_oracle as orc

    threads, blas = _blas_info()
    n, d, D = cfg["n"], cfg["d"], cfg["D"]
    D1 = D + 1
    out = {"unit": "fits/s", "cores": int(threads), "host_cpus": os.cpu_count(), "blas": blas, "kind": "port", "cores_note": _blas_threads_note(int(threads))}
    with threadpool_limits(limits=int(threads), user_api="blas"):
        full_fits_ram = n * D1 <= 2e8  # phi, S phi, h, phi beta(.) of the faithful schedule: 4 x 16 B per entry (c2: 6.6 GB)
        if full_fits_ram:
            X, y = synth(n, d, 0, n)
            s = np.ones(n)
            t0 = time.perf_counter()
            rS = orc.primal_fit_streamed(X, y, s, shift, scale, B, False, gammas=gammas)
            tS = time.perf_counter() - t0
            t0 = time.perf_counter()
            phi = orc.feature_map(X, shift, scale, B)
            rR = orc.primal_fit_faithful(phi, y, s, False, gammas=gammas)
            tR = time.perf_counter() - t0
            del phi
            out.update(
                value=1.0 / tR,
                mode="R (reference-faithful schedule, materialised phi, transform included), full size",
                seconds_mode_R=tR,
                seconds_mode_S=tS,
                value_mode_S=1.0 / tS,
                sample=f"all {n} rows, both schedules at full size; argmin R/S {rR['opt']}/{rS['opt']}",
            )
            g = gpu_full if gpu_full is not None and "loo_residuals" in gpu_full else gpu_fit(X, y, s, None)
            if g["opt"] != rS["opt"]:
                g = gpu_fit(X, y, s, int(rS["opt"]))
            par = _parity(g, rS, n)
            par["argmin_equal"] = bool((gpu_full or g)["opt"] == rS["opt"])
            par["against"] = "oracle primal_fit_streamed on ALL rows of the timed workload"
            return out, par
        n_s = min(n, 65_536)
        X, y = synth(n, d, 0, n_s)
        s = np.ones(n_s)
        tm = {}
        t0 = time.perf_counter()
        o = orc.primal_fit_streamed(X, y, s, shift, scale, B, False, gammas=gammas, row_tile=8192, timings=tm)
        t_fit = time.perf_counter() - t0
        serial = tm.get("eigh", 0.0) + tm.get("cholesky", 0.0)  # n-independent: eigh of the (D+1)^2 matrix, cho_factor + cho_solve
        est = (t_fit - serial) * n / n_s + serial
        out.update(
            value=1.0 / est,
            mode="S (simplified row-streamed schedule): a real oracle fit on a row sample; row stages scaled to n, eigh / Cholesky unscaled",
            sample=f"oracle primal_fit_streamed (feature map x3, Gram, eigh({D1}), rotation, sweep over G = {len(gammas)}, selection, "
            f"cho_factor / cho_solve, residuals) on the first {n_s} of {n} rows at full d, D, G: {t_fit:.2f} s, of which eigh "
            f"{tm.get('eigh', 0.0):.2f} s + Cholesky {tm.get('cholesky', 0.0):.2f} s do not scale with n; the rest x{n / n_s:.2f}",
            seconds_estimated=est,
            seconds_sample_fit=t_fit,
            stage_seconds={k: round(v, 3) for k, v in tm.items()},
            note="the reference itself cannot run this size (phi alone is 65.5 GB); Mode R at the largest size it can run "
            "(c2) is in profiles/ (bench.py --config c2); the x n / n_s extrapolation of the row stages is checked once against a FULL-size "
            "Mode-S run of c3 on the same kind of host: profiles/r05_cpu_modeS_c3_full.json (tools/cpu_modeS_full.py)",
        )
    try:  # the one full-size run of this schedule on the same kind of host (tools/cpu_modeS_full.py): how far off the extrapolation is
        full = json.loads((ROOT / "profiles" / "r05_cpu_modeS_c3_full.json").read_text())
        if full.get("workload") == cfg["name"]:
            out["full_size_check"] = {
                "source": "profiles/r05_cpu_modeS_c3_full.json",
                "seconds_measured_full_size": full["seconds_full_size"],
                "seconds_extrapolated_in_that_run": full["seconds_extrapolated_from_sample"],
                "extrapolation_over_measured": full["extrapolation_over_measured"],
                "reading": "the extrapolated figure UNDER-states the CPU time (feature-map and rotation stages fall out of the host's caches at full size): value is conservative",
            }
    except Exception:
        pass
    g = gpu_fit(X, y, s, None)
    argmin_equal = g["opt"] == o["opt"]
    if not argmin_equal:
        g = gpu_fit(X, y, s, int(o["opt"]))
    par = _parity(g, o, n_s)
    par["argmin_equal"] = bool(argmin_equal)
    par["against"] = f"oracle primal_fit_streamed on the first {n_s} rows of the workload at full d, D, G (the same fit the cpu_baseline times)"
    return out, par


def synth_clf(n, d):
    """SURVEY 8(d) classification generator: y = (X w + 0.3 eps > 0)."""
    import numpy as np

    rng = np.random.default_rng(0)
    X = rng.standard_normal((n, d))
    w = rng.standard_normal(d) / np.sqrt(d)
    y01 = (X @ w + 0.3 * rng.standard_normal(n) > 0).astype(np.float64)
    return X, y01


def end_to_end_fit(cfg, ctx, dual=False):
    """Wall time of the user-visible ``NeoLSSVM.fit`` on the bench workload ("report end-to-end fit() separately"):
    host X, y in; input validation, the supervised affine pre-step on ALL rows (AffineSeparator: GPU bin statistics + the
    separator's small products; ORF frequency matrix), one upload, the solver call that ``value`` times, download of every
    fitted attribute, the calibration split of ``_neo_ls_svm.py:405-441``.  The estimator runs on the bench's own context
    (one fitting context per process and GPU).  One untimed call first (workspace arena), then the timed one."""
    import neo_ls_svm_amd as hp
    from neo_ls_svm_amd import _lib

    n, d = cfg["n"], cfg["d"]
    _lib.set_default_context(ctx)
    if dual:
        X, y01 = synth_clf(n, d)
        y = y01
        est = hp.NeoLSSVM(dual=True, device=ctx.device)
    else:
        X, y = synth(n, d, 0, n)
        est = hp.NeoLSSVM(primal_feature_map=hp.OrthogonalRandomFourierFeatures(num_features=cfg["D"]), dual=False, device=ctx.device)
    est.fit(X, y)
    import gc

    gc.collect()  # as timeit does: a generation-2 collection of the interpreter (~ 80 ms with sklearn / scipy imported) is not part of a fit
    gc_was_on = gc.isenabled()
    gc.disable()
    try:
        t0 = time.perf_counter()
        est.fit(X, y)
        t = time.perf_counter() - t0
    finally:
        if gc_was_on:
            gc.enable()
    return {
        "seconds": t,
        "fits_per_s": 1.0 / t,
        "stage_seconds": {k: round(v, 4) for k, v in est.fit_wall_.items()},
        "solver_seconds_inside": round(est.fit_timings_["total"], 4),
        "solver_stage_ms": {k: round(1e3 * v, 3) for k, v in est.fit_timings_.items() if k in (
            "upload", "featuremap", "gram", "evd", "rotate", "sweep", "loo", "cholesky", "residuals", "download")},
        "evd_stage_ms": ctx.evd_stage_ms(),
        "gamma_index": int(list(est.γs_).index(est.γ_)) if est.γ_ in est.γs_ else None,
        "note": "NeoLSSVM.fit(X, y) from host arrays, pre-step fitted on all rows (the timed solver step above uses the pre-step of the "
        "first 2e5 rows, SURVEY 8(d)); second of two calls, the interpreter's cyclic garbage collector paused during the timed call (timeit's convention)",
    }


def run_dual(args, cfg):
    """BASELINE config 4: one step = one ``nls_dual_fit`` (D1-D5: RBF kernel, EVD of sn K sn, reduced LOO sweep over G = 128,
    selection, Cholesky re-solve, residuals, sigma) on the affine-transformed rows X_, resident in HBM.  The dual path does
    not shard (the n x n EVD): N > 1 runs N independent replicas ("replicas only")."""
    import numpy as np

    import neo_ls_svm_amd as hp

    rank = int(os.environ.get("RANK", "0"))
    local_rank = int(os.environ.get("LOCAL_RANK", "0"))
    world = int(os.environ.get("WORLD_SIZE", "1"))
    n, d, G = cfg["n"], cfg["d"], cfg["G"]
    dev = int(os.environ.get("NLS_BENCH_DEVICE", local_rank))  # (tests: every rank on one device, through the stand-in communicator)
    ctx = hp.Context(dev)
    cctx = None
    if world > 1:
        from neo_ls_svm_amd.distributed import init_from_env

        cctx = hp.Context(dev)
        init_from_env(cctx)
    X, y01 = synth_clf(n, d)
    y = np.where(y01 == 1.0, 1.0, -1.0)
    s = np.ones(n)
    sep = hp.AffineSeparator().fit(X, y, s, ctx=ctx)  # the package's own supervised pre-step: X -> X_
    Xt = np.ascontiguousarray(sep.transform(X))
    gammas = hp.gamma_grid(G)
    dX, dy, ds = ctx.to_device(Xt), ctx.to_device(y), ctx.to_device(s)

    def barrier():
        ctx.synchronize()
        if cctx is not None:
            cctx.comm_barrier()

    def step(X_=dX, y_=dy, s_=ds):
        return hp.dual_fit(X_, y_, s_, True, gammas=gammas, ctx=ctx)

    # the L_ outputs of the loop: two host buffers reserved once, like the inputs (a C caller allocates - and may register - its output buffer
    # once; the Python mirror would otherwise create them during the first steps and page-lock them when they are first recycled)
    hp.reserve_factor_outputs((n, n), np.float64, ctx, 2)
    r = None
    for _ in range(args.warmup):
        r = step()  # (held like the timed results: the loop is in its steady state, two output buffers alternating, before the clock starts)
    barrier()
    t0 = time.perf_counter()
    stage = {}
    for _ in range(args.steps):
        r = step()
        for k, v in r["timings"].items():
            stage[k] = stage.get(k, 0.0) + v
    barrier()
    elapsed = time.perf_counter() - t0
    if cctx is not None:
        elapsed = float(cctx.comm_allreduce([elapsed], "max")[0])
    pcie = None
    if world == 1:
        tp = time.perf_counter()
        step(Xt, y, s)
        ctx.synchronize()
        pcie = time.perf_counter() - tp
    if rank == 0:
        r_ = Xt.shape[1]
        mw_tflops = stage["rotate_flops"] / max(stage["rotate"], 1e-12) / 1e12  # M = F0 W: 2 n^3 per fit
        alg = 2.0 * n * n * r_ + 2.0 * n**3 + 10.0 * n * n * G  # SURVEY 8(d) dual F (+ the variance product of D5)
        out = {
            "metric": "fits/sec (dual path, full gamma-sweep), n=1e4 d=256",
            "unit_note": "one step = one dual fit",
            "value": world * args.steps / elapsed,
            "unit": "fits/s",
            "n_gpus": world,
            "steps": args.steps,
            "warmup": args.warmup,
            "ms_per_step": 1e3 * elapsed / args.steps,
            "higher_is_better": True,
            "scaling": "weak",
            "vs_baseline": None,
            "dtype": "f64",
            "data": "synthetic",
            "value_pcie_inclusive": None if pcie is None else 1.0 / pcie,
            "config": {
                "workload": cfg["name"],
                "n": n, "d": d, "r": int(r_), "G": G,
                "parallelism": f"{world} independent replicas (the n x n EVD does not shard)" if world > 1 else "single GPU",
                "affine": "package pre-step (AffineSeparator) fitted on all rows",
                "gamma_index": r["opt"],
                "loo_score": r["loo_score"],
                "outputs": "every fitted attribute downloaded inside the timed region; L_ into one of two host buffers reserved (page-locked) before the warm-up, as a C caller's reused output buffer would be",
            },
            "roofline": {
                "kernel": "k_gemm (M = F0 W, the 2 n^3 product of the reduced sweep; stage time includes its n^2 helper kernels)",
                "bound": "mfma",
                "achieved": mw_tflops,
                "peak": FP64_MFMA_PEAK_TFLOPS,
                "unit": "TFLOP/s",
                "frac": mw_tflops / FP64_MFMA_PEAK_TFLOPS,
                "traffic": None,
                "avg_launch_ms": 1e3 * stage["rotate"] / max(stage["rotate_launches"], 1.0),
                "whole_fit_algorithmic_tflops": alg * args.steps / elapsed / 1e12,
                "note": "the eigendecomposition (stage evd) is O(n^3) too but reported as wall-time share (SURVEY 8(d)); its own stage split is in "
                "evd_stage_ms",
            },
            "stage_ms_per_step": {
                k: round(1e3 * stage.get(k, 0.0) / args.steps, 3)
                for k in ("upload", "gram", "evd", "rotate", "sweep", "loo", "cholesky", "residuals", "download", "total")
            },
            "evd_stage_ms": ctx.evd_stage_ms(),
        }
        if not args.no_cpu_baseline and world == 1:
            from threadpoolctl import threadpool_limits

            sys.path.insert(0, str(ROOT / "oracle"))
            import neolssvm_oracle as orc

            ctx.release_workspace()
            threads, blas = _blas_info()
            with threadpool_limits(limits=int(threads), user_api="blas"):
                t0 = time.perf_counter()
                o = orc.dual_fit_reduced(Xt, y, s, True, gammas=gammas)
                tc = time.perf_counter() - t0
            out["cpu_baseline"] = {
                "value": 1.0 / tc, "unit": "fits/s", "cores": int(threads), "host_cpus": os.cpu_count(), "blas": blas, "kind": "port",
                "sample": f"the oracle's reduced schedule (dual_fit_reduced: numpy eigh + one 2 n^3 product + cho_factor / cho_solve with n right-hand "
                f"sides) at FULL size, all {n} rows: {tc:.1f} s; the reference's own n x G x n schedule (102 GB here) cannot run this size",
                "argmin_cpu_gpu": [int(o["opt"]), int(r["opt"])],
                "gpu_over_cpu": (args.steps / elapsed) * tc,
            }  # fmt: skip
            g = r if r["opt"] == o["opt"] else hp.dual_fit(Xt, y, s, True, gammas=gammas, gamma_index=int(o["opt"]), ctx=ctx)

            def mx(a, b):
                return float(np.max(np.abs(np.asarray(a) - np.asarray(b))) / max(float(np.max(np.abs(b))), 1e-300))

            out["parity"] = {  # _neo_ls_svm.py:270-323: the GPU fit against the oracle's fit on all rows
                "rows": n, "argmin_equal": bool(r["opt"] == o["opt"]), "argmin_gpu_ref": [int(r["opt"]), int(o["opt"])],
                "loo_residuals_max_rel_err": mx(g["loo_residuals"], o["loo_residuals"]),
                "alpha_rel_err": float(np.linalg.norm(g["alpha"] - o["alpha"]) / np.linalg.norm(o["alpha"])),
                "loo_errors_max_rel_err": mx(g["loo_errors_gammas"], o["loo_errors_gammas"]),
                "loo_std_max_rel_err": mx(g["loo_std"], o["loo_std"]),
                "residuals_max_rel_err": mx(g["residuals"], o["residuals"]),
                "bar": 1e-5, "against": "oracle dual_fit_reduced on ALL rows of the timed workload",
            }  # fmt: skip
        else:
            out["cpu_baseline"] = None
            out["parity"] = None
        if world == 1 and not args.no_end_to_end:
            ctx.release_workspace()
            out["end_to_end"] = end_to_end_fit(cfg, ctx, dual=True)
            out["value_end_to_end"] = out["end_to_end"]["fits_per_s"]
        print(json.dumps(out), flush=True)
    if cctx is not None:
        cctx.close()
    ctx.close()


def spawn_ranks(args) -> int:
    """``python bench.py --gpus N`` without a launcher: start N copies of this script, one per GPU, before anything here
    has touched the GPU; relay rank 0's JSON line."""
    import socket

    with socket.socket() as sk:
        sk.bind(("127.0.0.1", 0))
        port = sk.getsockname()[1]
    procs = []
    for rank in range(args.gpus):
        env = dict(os.environ, RANK=str(rank), LOCAL_RANK=str(rank), WORLD_SIZE=str(args.gpus), MASTER_ADDR="127.0.0.1",
                   MASTER_PORT=str(port), HSA_ENABLE_IPC_MODE_LEGACY=os.environ.get("HSA_ENABLE_IPC_MODE_LEGACY", "0"))  # fmt: skip
        procs.append(subprocess.Popen([sys.executable, str(Path(__file__).resolve())] + sys.argv[1:], env=env,
                                      stdout=subprocess.PIPE if rank == 0 else subprocess.DEVNULL, text=True))  # fmt: skip
    out0, _ = procs[0].communicate()
    codes = [procs[0].returncode] + [p.wait() for p in procs[1:]]
    sys.stdout.write(out0)
    return max(abs(c) for c in codes)


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--gpus", type=int, default=1)
    ap.add_argument("--steps", type=int, default=3)
    ap.add_argument("--warmup", type=int, default=1)
    ap.add_argument("--config", default="c3", choices=sorted(CONFIGS))
    ap.add_argument("--no-cpu-baseline", action="store_true")
    ap.add_argument("--no-end-to-end", action="store_true", help="skip the NeoLSSVM.fit wall-time leg (pre-step + solver + calibration split)")
    ap.add_argument("--as-rank", type=int, default=0, help="with --of W: the virtual rank whose share of the sharded fit this ONE process measures")
    ap.add_argument("--of", type=int, default=0, help="virtual world size W (> 1): one rank's share of a W-GPU row-sharded fit of the configuration, on one GPU - "
                    "its row block, the rank-0 tridiagonal solve (as rank 0), its own eigenvector column block, every exchange with its real payload through a "
                    "one-rank RCCL communicator (enqueue + local pass timed, no link)")  # fmt: skip
    args = ap.parse_args()
    cfg = CONFIGS[args.config]
    virt = args.of if args.of > 1 else 0
    if virt and (args.gpus > 1 or "sigmas" in cfg or cfg.get("dual") or not 0 <= args.as_rank < virt):
        raise SystemExit("--as-rank / --of: one process (--gpus 1), a row-sharded primal configuration, 0 <= rank < W")
    if "WORLD_SIZE" not in os.environ and args.gpus > 1:
        raise SystemExit(spawn_ranks(args))
    rank = int(os.environ.get("RANK", "0"))
    local_rank = int(os.environ.get("LOCAL_RANK", "0"))
    world = int(os.environ.get("WORLD_SIZE", "1"))
    args.gpus = world
    if cfg.get("dual"):
        return run_dual(args, cfg)

    import numpy as np

    import neo_ls_svm_amd as hp
    from neo_ls_svm_amd.distributed import init_from_env

    n, d, D, G = cfg["n"], cfg["d"], cfg["D"], cfg["G"]
    grid_mode = "sigmas" in cfg
    dev = int(os.environ.get("NLS_BENCH_DEVICE", local_rank))  # (tests: every rank on one device, through the stand-in communicator)
    ctx = hp.Context(dev)
    # Row sharding: the fitting context itself joins the communicator.  Sigma sharding (c5): every rank fits all rows on
    # its own, so the communicator lives on a second context that only serves the barrier / the merge of the small tables.
    use_comm = world > 1 or os.environ.get("NLS_BENCH_FORCE_COMM") == "1" or virt > 0
    cctx = None
    if use_comm:
        cctx = hp.Context(dev) if grid_mode else ctx
        init_from_env(cctx)

    lo, hi = (0, n) if grid_mode else ((n * rank) // world, (n * (rank + 1)) // world)
    if virt:
        lo, hi = (n * args.as_rank) // virt, (n * (args.as_rank + 1)) // virt
    shift, scale, B = affine_params(n, d, D, ctx=ctx)
    X, y = synth(n, d, lo, hi)
    s = np.ones(hi - lo)
    gammas = hp.gamma_grid(1024)[::33] if grid_mode else hp.gamma_grid(G)
    dX, dy, ds = ctx.to_device(X), ctx.to_device(y), ctx.to_device(s)

    def barrier():
        ctx.synchronize()
        if cctx is not None:
            cctx.comm_barrier()

    def step(X_=dX, y_=dy, s_=ds):
        if grid_mode:
            sig = np.logspace(np.log10(0.25), np.log10(4.0), cfg["sigmas"])
            # ONE library call per grid (nls_primal_fit_grid): visiting order, early-out, tie rules and - through the communicator-only
            # context - the merge of the small tables all happen behind the C ABI
            g = hp.primal_fit_sigma_grid(X_, y_, s_, shift, scale, B, False, sig, gammas=gammas, ctx=ctx, rank=rank, world=world,
                                         merge_ctx=cctx if world > 1 else None)  # fmt: skip
            best = g["best"] or {}
            return {"opt": g["gamma_index"], "sigma_index": g["sigma_index"], "loo_score": best.get("loo_score"), "timings": g["timings"],
                    "finished_count": g["finished_count"]}  # fmt: skip
        # row-sharded fit: every rank ends with the same beta / lam / curve; the factor L_ (an output only) is produced and downloaded by rank 0
        return hp.primal_fit(X_, y_, s_, shift, scale, B, False, gammas=gammas, ctx=ctx, want_L=(rank == 0 and (not virt or args.as_rank == 0)))

    if grid_mode or rank == 0:
        # the L_ outputs of the loop: two page-locked host buffers reserved once, like the inputs (a C caller allocates - and may register - its
        # output buffer once; the Python mirror would otherwise create them during the first steps and page-lock them when they are first recycled)
        hp.reserve_factor_outputs((D + 1, D + 1), np.complex128, ctx, 2)
    r = None
    full = None
    if virt:  # one COMPLETE fit of the same rows first: the virtual rank takes its peers' eigenvector blocks from what it leaves on the device
        ctx.comm_set_virtual_rank(capture=True)
        full = step()
        ctx.comm_set_virtual_rank(args.as_rank, virt)
    for _ in range(args.warmup):
        r = step()  # (held like the timed results: the loop is in its steady state before the clock starts)
    barrier()
    t0 = time.perf_counter()
    stage = {}
    for _ in range(args.steps):
        r = step()
        for k, v in r["timings"].items():
            stage[k] = stage.get(k, 0.0) + v
    barrier()
    elapsed = time.perf_counter() - t0
    if cctx is not None:
        elapsed = float(cctx.comm_allreduce([elapsed], "max")[0])

    pcie = None
    if world == 1 and not grid_mode and not virt:  # the same step with host-resident inputs: one pageable H2D copy of X inside the call
        tp = time.perf_counter()
        step(X, y, s)
        ctx.synchronize()
        pcie = time.perf_counter() - tp

    if rank == 0:
        Kf, Np = -(-D // 128) * 128, -(-(D + 1) // 64) * 64
        rot_launches = max(stage["rotate_launches"], 1.0)
        rot_rows = stage["rotate_flops"] / (8.0 * (D + 1) ** 2)  # rows through K4 over the timed steps
        rot_exec_tflops = 6.0 * rot_rows * Kf * Np / max(stage["rotate"], 1e-12) / 1e12
        rot_alg_tflops = stage["rotate_flops"] / max(stage["rotate"], 1e-12) / 1e12
        # Traffic past L2 comes from separate rocprofv3 PMC passes (it cannot be read live); per row because every launch
        # streams (rows x panels) with the same reuse pattern.
        traffic = k1_traffic = traffic_src = mfma_busy = mfma_busy_src = gram_pmc = None
        for src in ("r06_pmc_summary.json", "r05_pmc_summary.json", "r04_pmc_summary.json"):  # the newest counter passes committed (tools/evidence_pass.sh <tag> pmc)
            try:
                pmc = json.loads((ROOT / "profiles" / src).read_text())
                pr = pmc["k_rotate3"]
                if pr["D"] == D and pr["d"] == d and pr.get("fetch_bytes_x2"):
                    if traffic is None:
                        traffic = (pr["fetch_bytes_x2"] + pr["write_bytes"]) * (rot_rows / rot_launches) / pr["rows_per_launch"]
                        traffic_src = f"profiles/{src}"
                    if mfma_busy is None and pr.get("mfma_busy"):
                        mfma_busy, mfma_busy_src = pr["mfma_busy"], f"profiles/{src}"
                    if gram_pmc is None and pmc.get("k_gram3", {}).get("fetch_bytes_x2"):
                        gram_pmc = dict(pmc["k_gram3"], source=f"profiles/{src}")
                pk = pmc.get("k_featuremap")
                if k1_traffic is None and pk and pk["D"] == D and pk["d"] == d and pk.get("hbm_bytes_per_row"):
                    k1_traffic = pk["hbm_bytes_per_row"]
            except Exception:
                pass
        if traffic is None and D == 4096 and d == 128:
            try:  # round 3's passes of the XCD-patch order that is the default now (8 x 5 at D = 4096): 333 440 rows per launch
                pr = json.loads((ROOT / "profiles" / "r03_pmc_rotate.json").read_text())["p8x5"]["k_rotate3"]
                traffic = (pr["fetch_bytes_x2_per_launch"] + pr["write_bytes_per_launch"]) * (rot_rows / rot_launches) / 333440.0
                traffic_src = "profiles/r03_pmc_rotate.json (p8x5)"
            except Exception:
                pass
        if k1_traffic is None:
            try:
                pk = json.loads((ROOT / "profiles" / "r02_pmc_summary.json").read_text()).get("k_featuremap")
                if pk and pk["D"] == D and pk["d"] == d:
                    k1_traffic = pk["hbm_bytes_per_row"]
            except Exception:
                pass
        fm_launches = max(stage["featuremap_launches"], 1.0)
        fm_rows = stage["featuremap_flops"] / (2.0 * d * D)
        fm_bytes = fm_rows * (8.0 * d + 16.0 * (D + 1))  # 8 n d + 16 n (D+1) algorithmic bytes
        fm_gbs = fm_bytes / max(stage["featuremap"], 1e-12) / 1e9
        whole_alg = (stage["rotate_flops"] + stage["gram_flops"] + stage["sweep_flops"] + stage["featuremap_flops"]) / elapsed / 1e12
        # flops the kernels actually execute: Hermitian half + 3M Gram, 3M rotation, compressed sweep when the
        # library takes it (G > 256 on the reference's grid), the K = d feature-map product; EVD / Cholesky time is inside `elapsed`
        rows_all = stage["gram_flops"] / (4.0 * (D + 1) ** 2)
        sweep_exec = (4.0 * Np * 128 + 4.0 * 128 * G) if G > 256 else 4.0 * Np * G
        whole_exec = (3.0 * rows_all * Kf * Kf + 6.0 * rot_rows * Kf * Np + rot_rows * sweep_exec + stage["featuremap_flops"]) / elapsed / 1e12
        out = {
            "metric": "fits/sec (full gamma-sweep), n=1e6 d=128 D=4096" if args.config == "c3"
            else ("gamma x sigma grids/sec (16 sigma x 32 gamma), n=1e6 d=128 D=4096" if grid_mode else f"fits/sec (full gamma-sweep), {args.config}"),
            "unit_note": "one step = the whole 16 x 32 grid (16 fits)" if grid_mode else "one step = one fit",
            "value": args.steps / elapsed,
            "unit": "grids/s" if grid_mode else "fits/s",
            "n_gpus": world,
            "steps": args.steps,
            "warmup": args.warmup,
            "ms_per_step": 1e3 * elapsed / args.steps,
            "higher_is_better": True,
            "scaling": "weak" if grid_mode else "strong",
            "vs_baseline": None,
            "dtype": "f64",
            "data": "synthetic",
            "value_pcie_inclusive": None if pcie is None else 1.0 / pcie,
            "config": {
                "workload": cfg["name"],
                "n": n, "d": d, "D": D, "G": G,
                "rows_per_gpu": hi - lo,
                "parallelism": (f"sigma-shard x{world}, rows replicated" if grid_mode else f"row-shard x{world}, RCCL all-reduce of A||b") if world > 1
                else ("single GPU" if not virt else
                      f"VIRTUAL rank {args.as_rank} of {virt} on one GPU: rows [{lo}, {hi}) of {n}; tridiagonal eigensolver {'run (rank 0)' if args.as_rank == 0 else 'not run'}; "
                      f"back-transformation of this rank's eigenvector columns only; the status votes and every exchange (weight sums, {8 * (2 * (Kf // 128) * (Kf // 128 + 1) // 2 * 128 * 128 + 4 * Kf + 8) / 1e6:.0f} MB "
                      f"all-reduce of A||b, {8 * (D + 1) ** 2 / 1e6:.0f} MB broadcast of the real eigenvectors, all-gather of {virt} column blocks, error vectors, beta) "
                      "go through a ONE-rank RCCL communicator with their real payloads - enqueue and local pass timed, no xGMI link; the peers' blocks are a "
                      "device copy of a complete fit's (results = the complete fit's, checked below)"),
                "affine": f"package pre-step (AffineSeparator + ORF RandomState 42) fitted on the first {min(n, PRESTEP_PREFIX)} rows (SURVEY 8d)",
                "gamma_index": r["opt"],
                "sigma_index": r.get("sigma_index"),
                "loo_score": r["loo_score"],
                "outputs": "every fitted attribute downloaded inside the timed region; L_ into one of two host buffers reserved (page-locked) before the warm-up, as a C caller's reused output buffer would be",
                "generator": "SURVEY 8(d) (X ~ N(0,1), w ~ N(0,1)/sqrt(d), y = sin(Xw) + 0.1 eps) with w from default_rng(0) and the rows of "
                "block k (65 536 rows) from the child stream default_rng([0, k]) - not the single default_rng(0) stream - so that a rank "
                "generates just its shard",
            },
            "roofline": {
                "kernel": "k_rotate3",
                "bound": "mfma",
                "achieved": rot_exec_tflops,
                "peak": FP64_MFMA_PEAK_TFLOPS,
                "unit": "TFLOP/s",
                "frac": rot_exec_tflops / FP64_MFMA_PEAK_TFLOPS,
                "frac_is": "executed-MFMA utilisation (flops the kernel executes / time / peak); frac_algorithmic is SURVEY 8(d)'s figure",
                "frac_algorithmic": rot_alg_tflops / FP64_MFMA_PEAK_TFLOPS,
                "traffic": traffic,
                "from_profiles": {
                    "what": "NOT measured by this run: counter values read from the committed rocprofv3 --pmc summaries under profiles/ (separate passes "
                    "of the same kernels; counters cannot be read live) and scaled to this run's rows per launch.  `traffic` above repeats the value "
                    "because the bench contract names that key; everything else in `roofline` is measured live with HIP events",
                    "mfma_busy": mfma_busy,
                    "mfma_busy_source": None if mfma_busy is None else f"{mfma_busy_src}: SQ_VALU_MFMA_BUSY_CYCLES / (GRBM_GUI_ACTIVE / 8 x 1024 matrix pipes), its own rocprofv3 --pmc pass",
                    "traffic": traffic,
                    "traffic_unit": "bytes/launch beyond L2 (2 x FETCH_SIZE + WRITE_SIZE, Infinity-Cache hits included)",
                    "traffic_source": traffic_src,
                    "traffic_over_algorithmic": None if traffic is None else traffic / ((rot_rows / rot_launches) * 16.0 * (Kf + Np) + 16.0 * Kf * Np),
                },
                "algorithmic_bytes_per_launch": (rot_rows / rot_launches) * 16.0 * (Kf + Np) + 16.0 * Kf * Np,
                "note": "achieved / frac = EXECUTED MFMA flops (3M complex product: 6 rows Kf Np, Kf = ceil(D/128)*128, Np = ceil((D+1)/64)*64) / kernel "
                "time: the matrix-pipe utilisation.  frac_algorithmic = SURVEY 8(d)'s algorithmic 8 rows (D+1)^2 flops of the four-product form / time / "
                "peak: it exceeds the executed figure by algorithmic_gain (3 instead of 4 real products per complex product, D+1 padded to Np) and may "
                "exceed 1 - the kernel does LESS arithmetic than the formula prices, with results equal to the reference's to 1e-13 (parity)",
                "algorithmic_tflops": rot_alg_tflops,
                "algorithmic_gain": rot_alg_tflops / rot_exec_tflops,
                "avg_launch_ms": 1e3 * stage["rotate"] / rot_launches,
                "executed_flops_per_launch": 6.0 * rot_rows * Kf * Np / rot_launches,
                "whole_fit_algorithmic_tflops": whole_alg,
                "whole_fit_executed_tflops": whole_exec,
                "whole_fit_executed_frac": whole_exec / (FP64_MFMA_PEAK_TFLOPS * world),
            },
            "roofline_gram": {
                "kernel": "k_gram3 (Hermitian lower block triangle, 3M, split-K over rows)",
                "bound": "mfma",
                "achieved": 3.0 * rows_all * Kf * Kf / max(stage["gram"], 1e-12) / 1e12,
                "peak": FP64_MFMA_PEAK_TFLOPS,
                "unit": "TFLOP/s",
                "frac": 3.0 * rows_all * Kf * Kf / max(stage["gram"], 1e-12) / 1e12 / FP64_MFMA_PEAK_TFLOPS,
                "frac_is": "executed-MFMA utilisation over the whole gram stage (k_gram3 + slab reduction + border sums); 3 rows Kf^2 executed flops",
                "frac_algorithmic": stage["gram_flops"] / max(stage["gram"], 1e-12) / 1e12 / FP64_MFMA_PEAK_TFLOPS,
                "avg_launch_ms": 1e3 * stage["gram"] / max(stage["gram_launches"], 1.0),
                "algorithmic_bytes_per_launch": (rows_all / max(stage["gram_launches"], 1.0)) * 16.0 * Kf + 8.0 * Kf * Kf,
                "traffic": None if not gram_pmc else (gram_pmc["fetch_bytes_x2"] + gram_pmc["write_bytes"]) * (rows_all / max(stage["gram_launches"], 1.0)) / gram_pmc["rows_per_launch"],
                "from_profiles": {
                    "what": "NOT measured by this run: committed rocprofv3 --pmc summaries (see roofline.from_profiles.what)",
                    "mfma_busy": None if not gram_pmc else gram_pmc.get("mfma_busy"),
                    "traffic": None if not gram_pmc else (gram_pmc["fetch_bytes_x2"] + gram_pmc["write_bytes"]) * (rows_all / max(stage["gram_launches"], 1.0)) / gram_pmc["rows_per_launch"],
                    "traffic_over_algorithmic": None if not gram_pmc else (gram_pmc["fetch_bytes_x2"] + gram_pmc["write_bytes"]) / (gram_pmc["rows_per_launch"] * 16.0 * Kf + 8.0 * Kf * Kf),
                    "traffic_source": None if not gram_pmc else f"{gram_pmc['source']} ({gram_pmc.get('order')})",
                },
            },
            "roofline_k1": {
                "kernel": "k_featuremap (+ k_shift_pad)",
                "bound": "hbm",
                "achieved": fm_gbs,
                "peak": HBM_PEAK_GBS,
                "unit": "GB/s",
                "frac": fm_gbs / HBM_PEAK_GBS,
                "traffic": None if k1_traffic is None else k1_traffic * fm_rows / fm_launches,
                "from_profiles": {"what": "NOT measured by this run (see roofline.from_profiles.what)", "traffic": None if k1_traffic is None else k1_traffic * fm_rows / fm_launches,
                                  "traffic_over_algorithmic": None if k1_traffic is None else k1_traffic * fm_rows / fm_bytes},
                "avg_launch_ms": 1e3 * stage["featuremap"] / fm_launches,
                "algorithmic_bytes_per_launch": fm_bytes / fm_launches,
                # The bound that actually binds K1: the fp64 matrix pipe and the fp64 vector ALU share one datapath (profiles/r02_probe_f64_coexec.log),
                # so the K = d product and the sincos epilogue ADD: t >= 2 rows dk Kf / MFMA peak + (VALU instructions per feature) rows Kf / issue rate.
                "datapath": (lambda dk, mat_s, valu_s: {
                    "bound": "fp64 datapath (MFMA + VALU, serialised)",
                    "matrix_ms_per_launch": 1e3 * mat_s / fm_launches,
                    "valu_ms_per_launch": 1e3 * valu_s / fm_launches,
                    "valu_instructions_per_feature": K1_VALU_PER_FEATURE,
                    "valu_issue_rate_note": "256 CUs x 4 SIMDs x one wave64 fp64 instruction per 4 cycles at 2.4 GHz = 3.93e13 lane-operations/s; count from the ISA: tools/k1_valu_count.py",
                    "bound_ms_per_launch": 1e3 * (mat_s + valu_s) / fm_launches,
                    "frac": (mat_s + valu_s) / max(stage["featuremap"], 1e-12),
                })(-(-d // 16) * 16, 2.0 * fm_rows * (-(-d // 16) * 16) * Kf / (FP64_MFMA_PEAK_TFLOPS * 1e12), K1_VALU_PER_FEATURE * fm_rows * Kf / 3.93e13),
            },
            "stage_ms_per_step": {
                k: round(1e3 * stage.get(k, 0.0) / args.steps, 3)
                for k in ("upload", "featuremap", "gram", "allreduce", "evd", "rotate", "sweep", "loo", "cholesky", "residuals", "download", "total")
            },
            "evd_stage_ms": ctx.evd_stage_ms(),
        }
        if virt:
            bd = float(np.max(np.abs(r["beta"] - full["beta"])) / np.max(np.abs(full["beta"])))
            out["virtual_rank"] = {
                "rank": args.as_rank, "of": virt, "global_n": n, "projected_fits_per_s_before_links": out["value"],
                "equals_complete_fit": {"argmin_equal": bool(r["opt"] == full["opt"]), "beta_max_rel_diff": bd,
                                        "loo_errors_max_rel_diff": float(np.max(np.abs(r["loo_errors_gammas"] - full["loo_errors_gammas"])) / np.max(np.abs(full["loo_errors_gammas"])))},
                "note": "value = 1 / (this rank's time per fit): what a W-GPU node would reach if the links were free; n_gpus stays 1 (one GPU ran)",
            }  # fmt: skip
            args.no_cpu_baseline = args.no_end_to_end = True
        if not args.no_cpu_baseline and world == 1:
            ctx.release_workspace()

            def gpu_fit(Xh, yh, sh, gamma_index):
                return hp.primal_fit(Xh, yh, sh, shift, scale, B, False, gammas=gammas, gamma_index=gamma_index, ctx=ctx)

            out["cpu_baseline"], out["parity"] = cpu_baseline(cfg, shift, scale, B, gammas, gpu_fit, None if grid_mode else r)
            out["cpu_baseline"]["gpu_over_cpu"] = out["value"] * (cfg.get("sigmas", 1)) / out["cpu_baseline"]["value"]
        else:
            out["cpu_baseline"] = None
            out["parity"] = None
        if world == 1 and not args.no_end_to_end:
            ctx.release_workspace()
            r.pop("L", None)  # (hand the loop's output buffer back to the pool: the estimator's two fits then find their buffers there)
            out["end_to_end"] = None if grid_mode else end_to_end_fit(cfg, ctx)
            if out["end_to_end"]:
                out["value_end_to_end"] = out["end_to_end"]["fits_per_s"]
        print(json.dumps(out), flush=True)
    if cctx is not None and cctx is not ctx:
        cctx.close()
    ctx.close()


if __name__ == "__main__":
    main()
