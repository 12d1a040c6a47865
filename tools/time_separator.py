"""Host time of the separator's direction step (``_prestep._separator_directions``) at c2's shape with and without its pool pipeline
(NLS_PRESTEP_PIPELINE), and of ``orf_frequencies``: no GPU call - run it on the GPU box for that host's figure."""
import os, sys, time
from pathlib import Path
sys.path.insert(0, str(Path(__file__).resolve().parent.parent))
import numpy as np
from neo_ls_svm_amd import _prestep, hotpath

n, d, D = 100_000, 64, 1024
rng = np.random.default_rng(0)
X = rng.standard_normal((n, d)); w = rng.standard_normal(d) / np.sqrt(d); y = np.sin(X @ w) + 0.1 * rng.standard_normal(n)
sh, sc = _prestep.fit_affine_normalizer(X[:20000], y[:20000])
print("host cpus", os.cpu_count())
ref = None
for mode in ("0", "1", "0", "1"):
    os.environ["NLS_PRESTEP_PIPELINE"] = mode
    with _prestep.blas_threads(8):
        ts = []
        for _ in range(6):
            t = time.perf_counter(); r = _prestep._separator_directions(X, y, None, sh, sc, 2e-2, 384, 4, 42); ts.append(round(1e3 * (time.perf_counter() - t), 1))
    ref = r[2] if ref is None else ref
    print(f"pipeline={mode}: ms per call {ts}   identical to the first result: {np.array_equal(r[2], ref)}")
ts = []
for _ in range(5):
    t = time.perf_counter(); hotpath.orf_frequencies(d, D); ts.append(round(1e3 * (time.perf_counter() - t), 1))
print("orf_frequencies ms", ts)
