"""BASELINE config 4: dual path, synthetic binary classification n=1e4, d=256 (the reference cannot run it: its
H_loo tensor alone is 102 GB).  Times nls_dual_fit and checks it against the oracle's reduced schedule on a subsample."""
import sys, time
from pathlib import Path
ROOT = Path(__file__).resolve().parent.parent
sys.path[:0] = [str(ROOT), str(ROOT / "oracle")]
import os
os.environ.setdefault("ROCBLAS_USE_HIPBLASLT", "1")  # the launcher's export
import numpy as np
import neo_ls_svm_amd as hp

n, d = int(sys.argv[1]) if len(sys.argv) > 1 else 10_000, int(sys.argv[2]) if len(sys.argv) > 2 else 256
rng = np.random.default_rng(0)
X = rng.standard_normal((n, d)); w = rng.standard_normal(d) / np.sqrt(d)
y = np.where(X @ w + 0.3 * rng.standard_normal(n) > 0, 1.0, -1.0)
Xt = X * (0.3 / np.sqrt(d) * 4)  # stands in for the separator's output (unit-ish bandwidth)
s = np.ones(n)
for rep in range(3):
    t = time.time(); r = hp.dual_fit(Xt, y, s, True); dt = time.time() - t
    tm = r["timings"]
    print(f"dual fit n={n} r={d}: {dt:.3f} s  [kernel {tm['gram']*1e3:.1f} evd {tm['evd']*1e3:.1f} M-gemm {tm['rotate']*1e3:.1f} sweep {tm['sweep']*1e3:.1f} "
          f"loo {tm['loo']*1e3:.1f} chol+solves {tm['cholesky']*1e3:.1f} ms] opt={r['opt']} loo_score={r['loo_score']:.4f}")
if n <= 3000:
    import neolssvm_oracle as orc
    o = orc.dual_fit_reduced(Xt, y, s, True)
    print("  vs oracle: alpha", np.max(np.abs(r["alpha"] - o["alpha"])) / np.max(np.abs(o["alpha"])), "opt", r["opt"], o["opt"])
yq, sq = hp.dual_predict(Xt[:2000], Xt, alpha=r["alpha"], L=r["L"])
t = time.time(); yq, sq = hp.dual_predict(Xt[:2000], Xt, alpha=r["alpha"], L=r["L"]); print(f"  dual predict+std 2000 rows: {time.time()-t:.3f} s, train acc {np.mean(np.sign(yq)==y[:2000]):.3f}")
