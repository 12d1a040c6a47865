"""cProfile of an end-to-end NeoLSSVM.fit (second call: library and rocSOLVER initialisation excluded)."""
import cProfile, pstats, sys, time
from pathlib import Path
sys.path.insert(0, str(Path(__file__).resolve().parent.parent))
import numpy as np
import neo_ls_svm_amd as hp

n, d, D = (int(a) for a in (sys.argv[1:4] + ["1000000", "128", "4096"][len(sys.argv) - 1:]))
clf = len(sys.argv) > 4 and sys.argv[4] == "clf"
rng = np.random.default_rng(0)
X = rng.standard_normal((n, d)); w = rng.standard_normal(d) / np.sqrt(d)
y = (X @ w + 0.3 * rng.standard_normal(n) > 0) if clf else np.sin(X @ w) + 0.1 * rng.standard_normal(n)
mk = lambda: hp.NeoLSSVM(primal_feature_map=hp.OrthogonalRandomFourierFeatures(num_features=D), dual=False)
t = time.time(); mk().fit(X[:20000], y[:20000]); print(f"warm-up fit (20k rows) {time.time() - t:.2f} s")
pr = cProfile.Profile(); t = time.time(); pr.enable(); m = mk().fit(X, y); pr.disable(); el = time.time() - t
print(f"n={n} d={d} D={D} {'clf' if clf else 'reg'}: fit end to end {el:.2f} s, solver {m.fit_timings_['total']:.2f} s")
print("  solver stages (s):", {k: round(v, 3) for k, v in m.fit_timings_.items() if v and not k.endswith(("flops", "launches"))})
pr2 = cProfile.Profile(); t = time.time(); pr2.enable(); m2 = mk().fit(X, y); pr2.disable()
print(f"  second full-size fit (workspace already allocated): {time.time() - t:.3f} s, solver {m2.fit_timings_['total']:.3f} s, stages {({k: round(v, 4) for k, v in m2.fit_wall_.items()})}")
pstats.Stats(pr2 if len(sys.argv) > 5 and sys.argv[5] == "second" else pr).sort_stats("cumulative").print_stats(40)
