"""CPU-only checks of the C ABI: the library loads and exports exactly what include/*.h declares."""

from __future__ import annotations

import ctypes
import os
import re
from pathlib import Path

import numpy as np
import pytest

ROOT = Path(__file__).resolve().parent.parent


def declared_symbols():
    text = (ROOT / "include" / "neolssvm_hip.h").read_text()
    text = re.sub(r"/\*.*?\*/", "", text, flags=re.S)
    return sorted(set(re.findall(r"\b(nls_[a-z0-9_]+)\s*\(", text)))


def test_header_declares_the_path():
    syms = declared_symbols()
    for must in ("nls_featuremap", "nls_gram_only", "nls_primal_fit", "nls_primal_predict", "nls_dual_fit", "nls_dual_predict"):
        assert must in syms


def test_library_exports_every_declared_symbol():
    from neo_ls_svm_amd import _lib

    lib = _lib.load_library()
    for name in declared_symbols():
        assert hasattr(lib, name), f"{name} declared in the header but not exported"
        assert name in _lib.SIGNATURES, f"{name} has no ctypes signature"
    assert lib.nls_abi_version() == 1


def test_struct_layout_matches_header():
    from neo_ls_svm_amd import _lib

    # 7 pointers, int64, 5 int32 (+4 pad), 12 pointers
    assert ctypes.sizeof(_lib.PrimalFitArgs) == 7 * 8 + 8 + 5 * 4 + 4 + 12 * 8
    assert _lib.PrimalFitArgs.n.offset == 56 and _lib.PrimalFitArgs.beta.offset == 88
    # 4 pointers, int64, 4 int32, 11 pointers
    assert ctypes.sizeof(_lib.DualFitArgs) == 4 * 8 + 8 + 4 * 4 + 11 * 8
    assert _lib.DualFitArgs.alpha.offset == 56


def test_no_cpu_fallback_without_gpu():
    """On a box without an MI355X the context must refuse loudly, not compute on the CPU."""
    # (torch is deliberately not imported here: loading torch's bundled ROCm after this library's
    # /opt/rocm one in the same process is not supported; see neo_ls_svm_amd/_lib.py.)
    from neo_ls_svm_amd import Context, NlsError

    try:
        ctx = Context(0)
    except NlsError as exc:
        assert "no CPU fallback" in str(exc) or "failed" in str(exc)
    else:
        ctx.close()
        pytest.skip("GPU present")


def test_host_orf_frequencies_match_fixture():
    from conftest import load_golden

    from neo_ls_svm_amd import orf_frequencies

    g = load_golden("primal_reg_n3000_d20_D256")
    Z = orf_frequencies(g["A_sep"].shape[1], int(g["D"]), 42)
    assert np.array_equal(Z, g["Z"])
    g = load_golden("primal_reg_n2000_d48_D32")  # D < d': a single, truncated QR block
    Z = orf_frequencies(g["A_sep"].shape[1], int(g["D"]), 42)
    assert np.array_equal(Z, g["Z"])


def test_3m_engine_agpr_invariant():
    """The 3M engine names AGPRs a0..a191 in inline asm (csrc/nls_gemm3m.h): the compiler must allocate them to the
    kernels and never use an AGPR itself there.  Checked on the ISA hipcc generates (cross-compile, no GPU needed)."""
    import shutil
    import subprocess
    import sys

    if not os.path.exists("/opt/rocm/bin/hipcc") and shutil.which("hipcc") is None:
        pytest.skip("hipcc not available")
    tool = os.path.join(os.path.dirname(os.path.dirname(os.path.abspath(__file__))), "tools", "check_agpr.py")
    r = subprocess.run([sys.executable, tool], capture_output=True, text=True, timeout=600)
    assert r.returncode == 0, r.stdout + r.stderr
