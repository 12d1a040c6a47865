"""Does a host-side pause before a fit slow the (launch-latency-bound) eigendecomposition down?  c2-size fits with and without a pause."""
import sys, time
from pathlib import Path
sys.path.insert(0, str(Path(__file__).resolve().parent.parent))
import numpy as np
import bench, neo_ls_svm_amd as hp
cfg = bench.CONFIGS["c2"]; n, d, D = cfg["n"], cfg["d"], cfg["D"]
ctx = hp.Context(0)
shift, scale, B = bench.affine_params(n, d, D, ctx=ctx)
X, y = bench.synth(n, d, 0, n); s = np.ones(n)
dX, dy, ds = ctx.to_device(X), ctx.to_device(y), ctx.to_device(s)
def fit(host=False):
    r = hp.primal_fit(X if host else dX, y if host else dy, s if host else ds, shift, scale, B, False, ctx=ctx)
    return r["timings"], ctx.evd_stage_ms()
for _ in range(3): fit()
for label, pause, host in (("back to back", 0.0, False), ("0.15 s sleep before", 0.15, False), ("0.15 s numpy work before", -0.15, False), ("host inputs, back to back", 0.0, True), ("host inputs, sleep", 0.15, True)):
    out = []
    for _ in range(5):
        if pause > 0: time.sleep(pause)
        elif pause < 0:
            t0 = time.perf_counter(); a = np.random.randn(400, 400)
            while time.perf_counter() - t0 < -pause: a = a @ a; a /= np.abs(a).max()
        tm, ev = fit(host)
        out.append((round(tm["total"] * 1e3, 1), round(tm["evd"] * 1e3, 1), ev["tridiagonalisation"], ev["stedc"]))
    print(label, "(total, evd, trd, stedc ms):", out, flush=True)
