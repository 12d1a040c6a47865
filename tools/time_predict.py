"""Inference rates of the primal model at the c3 shape (d = 128, D = 4096): decision_function and predict_std (P10 / P11)."""
import sys, time
from pathlib import Path
sys.path.insert(0, str(Path(__file__).resolve().parent.parent))
import numpy as np
import neo_ls_svm_amd as hp
import bench

d, D = 128, 4096
ctx = hp.default_context()
shift, scale, B = bench.affine_params(200_000, d, D, ctx=ctx)
rng = np.random.default_rng(0)
X, y = bench.synth(200_000, d, 0, 200_000)
r = hp.primal_fit(X, y, np.ones(len(y)), shift, scale, B, False)
for m in (20_000, 1_000_000):
    Xq = rng.standard_normal((m, d))
    dXq = ctx.to_device(Xq)
    hp.primal_predict(dXq, shift, scale, B, beta=r["beta"])
    t = time.time(); yh, _ = hp.primal_predict(dXq, shift, scale, B, beta=r["beta"]); t1 = time.time() - t
    t = time.time(); yh2, _ = hp.primal_predict(Xq, shift, scale, B, beta=r["beta"]); t1h = time.time() - t
    ms = min(m, 200_000)
    L = np.ascontiguousarray(r["L"])
    t = time.time(); y3, sg = hp.primal_predict(Xq[:ms], shift, scale, B, beta=r["beta"], L=L); t2c = time.time() - t  # factor inverted on the fly
    f = hp.Factor(ctx, L)
    hp.primal_predict(Xq[:100], shift, scale, B, factor=f)
    t = time.time(); _, sg = hp.primal_predict(Xq[:ms], shift, scale, B, beta=r["beta"], factor=f); t2 = time.time() - t  # factor handle
    f.close()
    print(f"m={m}: decision_function {t1*1e3:.1f} ms resident X ({m/t1/1e6:.2f} M rows/s), {t1h*1e3:.1f} ms host X; "
          f"yhat + sigma on {ms} rows {t2*1e3:.1f} ms ({ms/t2/1e6:.3f} M rows/s; {t2c*1e3:.1f} ms with L passed: 268 MB upload + ztrtri); "
          f"fused vs plane path max diff {np.max(np.abs(yh[:ms] - y3)):.2e}")
