"""Which stage of the two-stage EVD raises the fall-back flag on time_evd.py's test matrix?  (development probe)"""
import os, sys
from pathlib import Path
sys.path.insert(0, str(Path(__file__).resolve().parent.parent))
import numpy as np
import neo_ls_svm_amd as hp

n = int(sys.argv[1]) if len(sys.argv) > 1 else 6500
bw = int(sys.argv[2]) if len(sys.argv) > 2 else 32
rng = np.random.default_rng(0)
M = rng.standard_normal((n, n // 2 + 8))
A = M @ M.T / n + np.eye(n)
Aout, tau1, failed, nred = hp.twostage_stage(1, A, bw)
print(f"n={n} bw={bw}: stage 1 failed={failed} nred={nred} finite={np.isfinite(Aout).all()}")
L = np.tril(Aout) - np.tril(Aout, -bw - 1)
Bd = L + np.tril(L, -1).T
ev0 = np.linalg.eigvalsh(A)
print("band eig err", np.max(np.abs(np.linalg.eigvalsh(Bd) - ev0)))
d, e, V2, tmo = hp.twostage_stage(2, np.tril(Bd), bw)
T_ = np.diag(d) + np.diag(e, 1) + np.diag(e, -1)
print("stage 2 timed out", tmo, "tridiagonal eig err", np.max(np.abs(np.linalg.eigvalsh(T_) - ev0)))
