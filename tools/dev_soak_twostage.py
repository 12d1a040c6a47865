"""Development soak: many two-stage eigendecompositions in a row (hand-off protocol of the chase under repetition); residual checks on the device-side result."""
import sys, time
from pathlib import Path
sys.path.insert(0, str(Path(__file__).resolve().parent.parent))
import numpy as np
import neo_ls_svm_amd as hp

ctx = hp.default_context()
rng = np.random.default_rng(7)
t0 = time.time()
worst = 0.0
for it in range(int(sys.argv[1]) if len(sys.argv) > 1 else 30):
    n = int(rng.choice([6000, 6100, 7001, 8192, 10000]))
    M = rng.standard_normal((n, 40))
    A = M @ M.T / 40 + np.diag(rng.uniform(0.5, 2.0, n))
    lam, Q = hp.eigh(A)
    # cheap checks: trace, a few random residual columns, orthogonality of a column sample
    cols = rng.choice(n, 8, replace=False)
    res = np.abs(A @ Q[:, cols] - Q[:, cols] * lam[cols]).max() / abs(lam).max()
    orth = np.abs(Q[:, cols].T @ Q - np.eye(n)[cols]).max()
    tr = abs(lam.sum() - np.trace(A)) / abs(lam).max()
    worst = max(worst, res, orth, tr / n)
    assert res < 1e-12 and orth < 1e-11 and tr < 1e-10 * n, (n, res, orth, tr)
print(f"soak ok: worst {worst:.1e}, {time.time() - t0:.0f} s, rescues {ctx.lib.nls_twostage_rescues(ctx.handle)} fallbacks {ctx.lib.nls_twostage_fallbacks(ctx.handle)}")
