"""The NumPy prototype of the library's tridiagonal divide and conquer (tools/stedc_proto.py) against numpy.linalg.eigh (CPU only): the GPU tests
(tests/test_gpu_stedc.py) check the HIP kernels on the same families of matrices."""

from __future__ import annotations

import sys
from pathlib import Path

import numpy as np
import pytest

sys.path.insert(0, str(Path(__file__).resolve().parent.parent / "tools"))
import stedc_proto as sp  # noqa: E402


def families():
    rng = np.random.default_rng(0)
    out = {"random 257": (rng.standard_normal(257), rng.standard_normal(256)), "diagonal": (np.arange(1.0, 101.0), np.zeros(99)), "toeplitz": (np.full(200, 2.0), np.ones(199))}
    wd = np.abs(np.arange(-20, 21)).astype(float)
    out["glued Wilkinson"] = (np.concatenate([wd] * 3), np.concatenate([np.ones(40), [1e-8], np.ones(40), [1e-8], np.ones(40)]))
    out["identity + tiny"] = (np.ones(130), 1e-9 * rng.standard_normal(129))
    out["graded"] = (10.0 ** np.linspace(0, -14, 150), 10.0 ** np.linspace(-1, -15, 149))
    return out


@pytest.mark.parametrize("name", list(families()))
def test_prototype_matches_numpy(name):
    d, e = families()[name]
    n = d.size
    T = np.diag(d) + np.diag(e, 1) + np.diag(e, -1)
    lam, Q = sp.stedc(d, e)
    ref = np.linalg.eigvalsh(T)
    nrm = max(np.max(np.abs(ref)), 1e-300)
    assert np.max(np.abs(lam - ref)) <= 1e-13 * nrm
    assert np.max(np.abs(T @ Q - Q * lam[None, :])) <= 1e-13 * nrm
    assert np.max(np.abs(Q.T @ Q - np.eye(n))) <= 1e-13
