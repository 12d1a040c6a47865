"""Development probe: where the wall time of a dual c4 fit goes outside the library's own stage timers."""
import os, sys, time
from pathlib import Path
sys.path.insert(0, str(Path(__file__).resolve().parent.parent))
import numpy as np
import neo_ls_svm_amd as hp
import bench

n, d, G = 10000, 256, 128
ctx = hp.Context(0)
X, y01 = bench.synth_clf(n, d)
y = np.where(y01 == 1.0, 1.0, -1.0); s = np.ones(n)
sep = hp.AffineSeparator().fit(X, y, s, ctx=ctx)
Xt = np.ascontiguousarray(sep.transform(X))
gam = hp.gamma_grid(G)
dX, dy, ds = ctx.to_device(Xt), ctx.to_device(y), ctx.to_device(s)
# (the library reads NLS_PIN_OUTPUT once per process: run this script once per setting)
for label, kw in (("default", {}), ("want_L=False", {"want_L": False})):
    hp.dual_fit(dX, dy, ds, True, gammas=gam, ctx=ctx, **kw)
    ts = []
    for _ in range(3):
        t = time.perf_counter(); r = hp.dual_fit(dX, dy, ds, True, gammas=gam, ctx=ctx, **kw); ts.append(time.perf_counter() - t)
    tm = r["timings"]
    print(f"NLS_PIN_OUTPUT={os.environ.get('NLS_PIN_OUTPUT', '(unset)')} {label:14s}: wall {np.mean(ts)*1e3:7.1f} ms  library total {tm['total']*1e3:7.1f}  "
          f"(evd {tm['evd']*1e3:.1f} cholesky {tm['cholesky']*1e3:.1f} download {tm['download']*1e3:.1f})", flush=True)
