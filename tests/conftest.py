"""Shared test configuration: markers, paths and fixture loading."""

from __future__ import annotations

import sys
from pathlib import Path

import numpy as np
import pytest

ROOT = Path(__file__).resolve().parent.parent
GOLDEN = ROOT / "tests" / "golden"
for p in (str(ROOT), str(ROOT / "oracle")):
    if p not in sys.path:
        sys.path.insert(0, p)


def pytest_configure(config):
    config.addinivalue_line("markers", "gpu: needs a real MI355X (run with -m gpu on the GPU box)")


PRIMAL_CASES = [
    "primal_reg_n3000_d20_D256",
    "primal_reg_n5000_d16_D256_w",
    "primal_clf_n3000_d16_D256_wz",
    "primal_clf_n2500_d24_D192",
    "primal_reg_n2000_d48_D32",
    # the feature-map plug-in point (_neo_ls_svm.py:62-75): plain RandomFourierFeatures; ORF over an AffineNormalizer (A = None)
    "primal_reg_n2000_d12_RFF256",
    "primal_clf_n1500_d10_ORF128_normalizer",
]
PLUGIN_CASES = {"primal_reg_n2000_d12_RFF256": "rff", "primal_clf_n1500_d10_ORF128_normalizer": "orf_normalizer"}
DUAL_CASES = ["dual_reg_n300_d12", "dual_clf_n500_d20_wz", "dual_reg_n1000_d32_w"]


def load_golden(name: str) -> dict:
    z = np.load(GOLDEN / f"{name}.npz", allow_pickle=False)
    out = {k: z[k] for k in z.files}
    for k in ("kind", "task", "base"):
        if k in out:
            out[k] = str(out[k])
    return out


def signed_targets(g: dict) -> np.ndarray:
    """The +-1 / float target the solver sees (reference ``fit``: ``_neo_ls_svm.py:364-370``)."""
    y = g["y"]
    if g["task"] == "clf":
        return np.where(y == np.max(y), 1.0, -1.0)
    return y.astype(np.float64)


def relerr(a, b) -> float:
    a, b = np.asarray(a), np.asarray(b)
    denom = np.max(np.abs(b))
    return float(np.max(np.abs(a - b)) / (denom if denom > 0 else 1.0))


@pytest.fixture(scope="session")
def golden_loader():
    cache: dict = {}

    def _get(name):
        if name not in cache:
            cache[name] = load_golden(name)
        return cache[name]

    return _get
