"""Development probe: the "reuse" page-locking policy of the output pool in a loop of c3e-size primal fits."""
import sys, time
from pathlib import Path
sys.path.insert(0, str(Path(__file__).resolve().parent.parent))
import numpy as np
import neo_ls_svm_amd as hp
from neo_ls_svm_amd import _hostpool as pool
import bench
n, d, D = 125000, 128, 4096
policy = {"never": False, "reuse": "reuse", "always": True}[sys.argv[1] if len(sys.argv) > 1 else "reuse"]
pool.pin_large_outputs(policy)
print("policy", policy)
ctx = hp.Context(0)
shift, scale, B = bench.affine_params(n, d, D, ctx=ctx)
X, y = bench.synth(n, d, 0, n); s = np.ones(n)
dX, dy, ds = ctx.to_device(X), ctx.to_device(y), ctx.to_device(s)
r = None
for i in range(7):
    t = time.perf_counter(); r = hp.primal_fit(dX, dy, ds, shift, scale, B, False, ctx=ctx); w = time.perf_counter() - t
    tm = r["timings"]
    print(f"fit {i}: wall {w*1e3:.1f} ms library {tm['total']*1e3:.1f} download {tm['download']*1e3:.2f}; registered {len(pool._registered)} last rc {pool.LAST_REGISTER_RC} err {ctx.last_error() if hasattr(ctx,'last_error') else ''}", flush=True)
