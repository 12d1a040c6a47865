"""Development check of the two-stage EVD on the GPU: prints per-stage errors (no asserts) so that ONE run localises a bug."""
import os
import sys
import time
from pathlib import Path

ROOT = Path(__file__).resolve().parent.parent
sys.path[:0] = [str(ROOT), str(ROOT / "tools")]
import numpy as np

import neo_ls_svm_amd as hp
from twostage_proto import apply_q2_naive

hp.default_context()
rng = np.random.default_rng(0)


def herm(n, cplx, spd=False):
    M = rng.standard_normal((n, n)) + (1j * rng.standard_normal((n, n)) if cplx else 0)
    return M @ M.conj().T / n if spd else (M + M.conj().T) / 2


def band_of(Aout, bw):
    n = Aout.shape[0]
    L = np.tril(Aout) - np.tril(Aout, -bw - 1)
    return L + np.tril(L, -1).conj().T


sizes = [int(x) for x in os.environ.get("SIZES", "40,100,257,700").split(",")]
for cplx in (False, True):
    for bw in (32, 64):
        if cplx and bw == 64:
            continue
        for n in sizes:
            A = herm(n, cplx)
            t0 = time.time()
            Aout, tau1, failed, nred = hp.twostage_stage(1, A, bw)
            Bd = band_of(Aout, bw)
            Bd[np.diag_indices(n)] = Bd[np.diag_indices(n)].real
            ev0 = np.linalg.eigvalsh(A)
            err1 = np.max(np.abs(np.linalg.eigvalsh(Bd) - ev0)) / np.max(np.abs(ev0))
            # stage 2 on the exact band
            d, e, V2, tmo = hp.twostage_stage(2, np.tril(Bd), bw)
            T = np.diag(d) + np.diag(e, 1) + np.diag(e, -1)
            err2 = np.max(np.abs(np.linalg.eigvalsh(T) - np.linalg.eigvalsh(Bd))) / np.max(np.abs(ev0))
            # stage 3 against the reflector-by-reflector product
            Zt = (rng.standard_normal((n, 21)) + (1j * rng.standard_normal((n, 21)) if cplx else 0)).astype(A.dtype)
            ref = apply_q2_naive(V2, bw, Zt.copy())
            got = hp.twostage_stage(3, V2, bw, aux=Zt)
            err3 = np.max(np.abs(got - ref))
            # Q2^H Bd Q2 == T ?
            I = np.eye(n, dtype=A.dtype)
            Q2 = apply_q2_naive(V2, bw, I.copy())
            err2b = np.max(np.abs(Q2.conj().T @ Bd @ Q2 - T)) / np.max(np.abs(ev0))
            print(f"cplx={int(cplx)} bw={bw} n={n}: stage1 failed={failed} nred={nred} eig err {err1:.1e} | stage2 tmo={tmo} eig err {err2:.1e} "
                  f"Q2^H B Q2 - T {err2b:.1e} | stage3 vs naive {err3:.1e}  ({time.time() - t0:.1f}s)", flush=True)

os.environ["NLS_EVD"] = "twostage"
for cplx in (False, True):
    for bw in (32, 64):
        if cplx and bw == 64:
            continue
        os.environ["NLS_SB_BW"] = str(bw)
        for n in [5, 33, 34, 65, 66, 130] + sizes:
            A = herm(n, cplx, spd=True)
            lam, Q = hp.eigh(A)
            lam0 = np.linalg.eigvalsh(A)
            print(f"eigh twostage cplx={int(cplx)} bw={bw} n={n}: lam err {np.max(np.abs(lam - lam0)) / lam0[-1]:.1e}  resid {np.max(np.abs(A @ Q - Q * lam)) / lam0[-1]:.1e} "
                  f" orth {np.max(np.abs(Q.conj().T @ Q - np.eye(n))):.1e}", flush=True)
ctx = hp.default_context()
print("fallbacks so far", ctx.lib.nls_twostage_fallbacks(ctx.handle))
lam, Q = hp.eigh(np.diag(np.arange(1.0, 301.0)))
print("diagonal matrix: lam err", np.max(np.abs(lam - np.arange(1.0, 301.0))), "fallbacks", ctx.lib.nls_twostage_fallbacks(ctx.handle))
