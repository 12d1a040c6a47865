"""Times the eigendecomposition stage in isolation (nls_eigh_only; includes the host <-> device copies of A)."""
import sys, time
from pathlib import Path
sys.path.insert(0, str(Path(__file__).resolve().parent.parent))
import os
os.environ.setdefault("ROCBLAS_USE_HIPBLASLT", "1")  # the launcher's export
import numpy as np
import neo_ls_svm_amd as hp

n = int(sys.argv[1]) if len(sys.argv) > 1 else 4097
cplx = (sys.argv[2] if len(sys.argv) > 2 else "c") == "c"
reps = int(sys.argv[3]) if len(sys.argv) > 3 else 3
rng = np.random.default_rng(0)
M = rng.standard_normal((n, n // 2 + 8)) + (1j * rng.standard_normal((n, n // 2 + 8)) if cplx else 0)
A = M @ M.conj().T / n + np.eye(n)
hp.eigh(A[:64, :64])
for _ in range(reps):
    t = time.time(); lam, Q = hp.eigh(A); el = time.time() - t
    print(f"eigh n={n} {'complex' if cplx else 'real'}: {el*1e3:.1f} ms (with copies), lam range {lam[0]:.3g} .. {lam[-1]:.3g}")
