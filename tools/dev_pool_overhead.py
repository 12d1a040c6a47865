"""Development probe: where the wall time of hp.dual_fit goes around the C call when L_ comes from the page-locked pool."""
import sys, time, gc
from pathlib import Path
sys.path.insert(0, str(Path(__file__).resolve().parent.parent))
import numpy as np
import neo_ls_svm_amd as hp
from neo_ls_svm_amd import _hostpool as pool

orig = pool.factor_output
def timed(shape, dtype, ctx=None):
    t = time.perf_counter(); a = orig(shape, dtype, ctx); dt = time.perf_counter() - t
    print(f"    factor_output {shape}: {dt*1e3:.2f} ms, registered {len(pool._registered)}, pooled {pool._pooled_bytes()>>20} MB", flush=True)
    return a
import neo_ls_svm_amd.hotpath as hot
hot.factor_output = timed

n, d = int(sys.argv[1]) if len(sys.argv) > 1 else 10000, 64
rng = np.random.default_rng(0)
X = rng.standard_normal((n, d)) * 0.3
y = np.sign(X[:, 0] + 0.1 * rng.standard_normal(n)); s = np.ones(n)
ctx = hp.Context(0)
r = None
for i in range(5):
    t = time.perf_counter(); r = hp.dual_fit(X, y, s, True, ctx=ctx); wall = time.perf_counter() - t
    print(f"fit {i}: wall {wall*1e3:.1f} ms, library total {r['timings']['total']*1e3:.1f} ms", flush=True)
t = time.perf_counter(); del r; gc.collect(); print(f"dropping the last result: {(time.perf_counter()-t)*1e3:.2f} ms")
t = time.perf_counter(); pool.release(); print(f"pool.release(): {(time.perf_counter()-t)*1e3:.2f} ms")
