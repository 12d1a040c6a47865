"""The short sincos of the feature-map kernels (csrc/nls_sincos.h) compiled for the host and checked against long double
sin / cos: 25 million arguments over 60 binades plus the neighbourhoods of 2.6 million multiples of pi/2."""
from __future__ import annotations

import shutil
import subprocess
from pathlib import Path

import pytest

ROOT = Path(__file__).resolve().parent.parent


def test_fast_sincos_is_accurate_to_an_ulp(tmp_path):
    if shutil.which("g++") is None:
        pytest.skip("g++ not available")
    exe = tmp_path / "test_sincos"
    subprocess.run(["g++", "-O2", "-mfma", "-I", str(ROOT / "neo_ls_svm_amd" / "csrc"), str(ROOT / "tests" / "csrc" / "test_sincos.cpp"),
                    "-o", str(exe)], check=True)  # fmt: skip
    r = subprocess.run([str(exe)], capture_output=True, text=True, timeout=600)
    print(r.stdout)
    assert r.returncode == 0, r.stdout + r.stderr  # the program's own bar: max abs error < 3e-16 for sin and cos
    assert "nan: nan nan" in r.stdout and "zero: 0 1" in r.stdout
