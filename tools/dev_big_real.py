"""Development check: the default real eigendecomposition at a size well beyond c4 (residual and orthogonality only)."""
import sys, time
from pathlib import Path
sys.path.insert(0, str(Path(__file__).resolve().parent.parent))
import numpy as np
import neo_ls_svm_amd as hp

n = int(sys.argv[1]) if len(sys.argv) > 1 else 16000
rng = np.random.default_rng(2)
X = rng.standard_normal((n, 48))
sq = (X * X).sum(1)
A = np.exp(-0.5 * np.maximum(sq[:, None] + sq[None, :] - 2 * X @ X.T, 0) / 36.0)
ctx = hp.default_context()
t = time.time(); lam, Q = hp.eigh(A); el = time.time() - t
sc = abs(lam).max()
R = A @ Q - Q * lam
print(f"n={n}: {el:.2f} s  resid {np.abs(R).max()/sc:.1e}  orth {np.abs(Q.T @ Q - np.eye(n)).max():.1e}  trace err {abs(lam.sum() - np.trace(A))/sc:.1e}  lam range {lam[0]:.2e} .. {lam[-1]:.2e}"
      f"  rescues {ctx.lib.nls_twostage_rescues(ctx.handle)} fallbacks {ctx.lib.nls_twostage_fallbacks(ctx.handle)}")
