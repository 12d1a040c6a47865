import sys, time
sys.path.insert(0, '.')
import numpy as np
import neo_ls_svm_amd as hp
ctx = hp.Context(0)
rng = np.random.default_rng(0)
for n in (4097, 1025):
    M = rng.standard_normal((n, n)) + 1j * rng.standard_normal((n, n))
    A = M @ M.conj().T / n + np.eye(n)
    for rep in range(3):
        t = time.perf_counter(); L = hp.cholesky(A, ctx=ctx); dt = time.perf_counter() - t
    print(n, "hook wall (with copies)", round(dt * 1e3, 2), "ms", flush=True)
ctx.close()
