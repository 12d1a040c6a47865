import sys, time
sys.path.insert(0, '/root/repo')
import os
os.environ.setdefault("ROCBLAS_USE_HIPBLASLT", "1")  # the launcher's export
import numpy as np
import neo_ls_svm_amd as hp
n, d = 10000, 256
rng = np.random.default_rng(0)
X = rng.standard_normal((n, d)); w = rng.standard_normal(d) / np.sqrt(d)
y = np.where(X @ w + 0.3 * rng.standard_normal(n) > 0, 1.0, -1.0)
Xt = X * (0.3 / np.sqrt(d) * 4)
s = np.ones(n)
for want_L in (True, True, False, False):
    t = time.time(); r = hp.dual_fit(Xt, y, s, True, want_L=want_L); dt = time.time() - t
    tm = r["timings"]
    print(f"want_L={want_L}: wall {dt*1e3:.1f} ms, C total {tm['total']*1e3:.1f}, stages: " + " ".join(f"{k} {v*1e3:.1f}" for k, v in tm.items() if v and k not in ("total",) and not k.endswith(("flops", "launches", "chunk"))))
