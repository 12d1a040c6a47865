"""Development probe: where the wall time of NeoLSSVM.fit goes at c2 (n = 1e5, d = 64, D = 1024), third fit of a process, cProfile."""
import cProfile, pstats, sys, time, gc
from pathlib import Path
sys.path.insert(0, str(Path(__file__).resolve().parent.parent))
import numpy as np
import neo_ls_svm_amd as hp
import bench
X, y = bench.synth(100000, 64, 0, 100000)
mk = lambda: hp.NeoLSSVM(primal_feature_map=hp.OrthogonalRandomFourierFeatures(num_features=1024), dual=False)
est = mk(); est.fit(X, y); est.fit(X, y)
gc.collect(); gc.disable()
pr = cProfile.Profile(); t = time.perf_counter(); pr.enable(); est.fit(X, y); pr.disable(); el = time.perf_counter() - t
print(f"fit {el*1e3:.1f} ms; stages {est.fit_wall_}; inside {est.fit_timings_['total']*1e3:.1f} ms")
print({k: round(v*1e3,2) for k,v in est.fit_timings_.items() if not k.endswith(('flops','launches'))})
print(ctx_evd := hp.default_context().evd_stage_ms())
pstats.Stats(pr).sort_stats("tottime").print_stats(6)
