import os, sys
from pathlib import Path
ROOT = Path(__file__).resolve().parent.parent
sys.path[:0] = [str(ROOT), str(ROOT / "tools")]
import numpy as np
import neo_ls_svm_amd as hp
from twostage_proto import apply_q2_naive
rng = np.random.default_rng(0)
for cplx in (False, True):
    for n, bw in [(66, 64), (100, 64), (200, 64), (700, 64), (40, 32), (100, 32), (333, 32), (700, 32)]:
        if cplx and bw == 64:
            continue
        M = rng.standard_normal((n, n)) + (1j * rng.standard_normal((n, n)) if cplx else 0); A = (M + M.conj().T) / 2
        Bd = np.tril(A) - np.tril(A, -bw - 1)
        d, e, V2, tmo = hp.twostage_stage(2, Bd, bw)
        Zt = (rng.standard_normal((n, 37)) + (1j * rng.standard_normal((n, 37)) if cplx else 0)).astype(A.dtype)
        ref = apply_q2_naive(V2, bw, Zt.copy())
        for G in ("1", "2", "8"):
            os.environ["NLS_Q2_GROUPS"] = G
            got = hp.twostage_stage(3, V2, bw, aux=Zt)
            D = np.abs(got - ref)
            badrows = np.where(D.max(1) > 1e-10)[0]
            badcols = np.where(D.max(0) > 1e-10)[0]
            print(f"cplx={int(cplx)} n={n} bw={bw} G={G}: max err {D.max():.2e}; bad rows {badrows[:8]} ({len(badrows)}), bad cols {badcols[:8]} ({len(badcols)})", flush=True)
