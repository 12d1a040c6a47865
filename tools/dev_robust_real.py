"""Development check: the default (two-stage from n = 6000) real eigendecomposition on structured matrices against numpy."""
import sys, time
from pathlib import Path
sys.path.insert(0, str(Path(__file__).resolve().parent.parent))
import numpy as np
import neo_ls_svm_amd as hp

n = int(sys.argv[1]) if len(sys.argv) > 1 else 6200
rng = np.random.default_rng(1)
X1 = rng.standard_normal((n, 3))            # smooth kernel on 3-d data: numerically low rank
X2 = rng.standard_normal((n, 64))
def rbf(X, s):
    sq = (X * X).sum(1)
    return np.exp(-0.5 * np.maximum(sq[:, None] + sq[None, :] - 2 * X @ X.T, 0) / s**2)
cases = {
    "rbf d=3 wide (low rank)": rbf(X1, 3.0),
    "rbf d=64": rbf(X2, 8.0),
    "ones (rank 1)": np.ones((n, n)),
    "identity + 1e-8 noise": np.eye(n) + 1e-8 * (lambda M: (M + M.T) / 2)(rng.standard_normal((n, n))),
    "graded 1e-12 .. 1": (lambda Q, d: (Q * d) @ Q.T)(np.linalg.qr(rng.standard_normal((n, n)))[0], np.logspace(-12, 0, n)),
    "block diagonal": np.kron(np.eye(n // 100), rng.standard_normal((100, 100)) @ np.ones((100, 100)) * 0 + np.cov(rng.standard_normal((100, 300))))[:n, :n],
}
ctx = hp.default_context()
for name, A in cases.items():
    A = np.ascontiguousarray((A + A.T) / 2)
    f0, r0 = ctx.lib.nls_twostage_fallbacks(ctx.handle), ctx.lib.nls_twostage_rescues(ctx.handle)
    t = time.time(); lam, Q = hp.eigh(A); el = time.time() - t
    lam0 = np.linalg.eigvalsh(A)
    sc = max(np.max(np.abs(lam0)), 1e-300)
    print(f"{name:28s} n={A.shape[0]}: {el*1e3:7.1f} ms  lam err {np.max(np.abs(lam - lam0))/sc:.1e}  resid {np.max(np.abs(A @ Q - Q * lam))/sc:.1e}  orth {np.max(np.abs(Q.T @ Q - np.eye(A.shape[0]))):.1e}"
          f"  rescues +{ctx.lib.nls_twostage_rescues(ctx.handle) - r0} fallbacks +{ctx.lib.nls_twostage_fallbacks(ctx.handle) - f0}", flush=True)
