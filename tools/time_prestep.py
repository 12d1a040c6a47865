"""Times the supervised pre-step (CPU NumPy vs GPU nls_bin_stats) and an end-to-end NeoLSSVM.fit."""
import sys, time
from pathlib import Path
sys.path.insert(0, str(Path(__file__).resolve().parent.parent))
import numpy as np
import neo_ls_svm_amd as hp
from neo_ls_svm_amd import _prestep, hotpath

n, d = int(sys.argv[1]) if len(sys.argv) > 1 else 200_000, int(sys.argv[2]) if len(sys.argv) > 2 else 128
rng = np.random.default_rng(0)
X = rng.standard_normal((n, d)); w = rng.standard_normal(d) / np.sqrt(d); y = np.sin(X @ w) + 0.1 * rng.standard_normal(n)
ctx = hp.default_context()
t = time.time(); labels = _prestep.target_bins(y); t_bins = time.time() - t
hotpath.bin_stats(X[:1000], labels[:1000])  # warm up
t = time.time(); cen, spr = hotpath.bin_stats(X, labels); t_gpu = time.time() - t
dX = ctx.to_device(X)
t = time.time(); cen2, spr2 = hotpath.bin_stats(dX, labels); t_gpu_res = time.time() - t
print(f"n={n} d={d}: target_bins {t_bins:.2f} s; bin_stats GPU {t_gpu:.3f} s (X resident: {t_gpu_res:.3f} s)")
if n <= 300_000:
    t = time.time(); sh, sc = _prestep.fit_affine_normalizer(X, y); t_cpu = time.time() - t
    sh2, sc2 = _prestep.fit_affine_normalizer(X, y, stats=lambda A, l, s: hotpath.bin_stats(A, l, s))
    print(f"  normalizer CPU {t_cpu:.2f} s; shift/scale max rel diff {np.max(np.abs(sh - sh2) / np.abs(sh)):.2e} {np.max(np.abs(sc - sc2) / np.abs(sc)):.2e}")
t = time.time(); m = hp.NeoLSSVM(primal_feature_map=hp.OrthogonalRandomFourierFeatures(num_features=int(sys.argv[3]) if len(sys.argv) > 3 else 1024), dual=False).fit(X, y); t_fit = time.time() - t
print(f"  NeoLSSVM.fit end to end {t_fit:.2f} s (solver {m.fit_timings_['total']:.2f} s), loo_score {m.loo_score_:.4f}, gamma index {int(np.argmin(np.abs(m.γs_ - m.γ_)))}")
