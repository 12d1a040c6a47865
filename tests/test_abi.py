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
    assert lib.nls_abi_version() == 4


def test_struct_layout_matches_header():
    from neo_ls_svm_amd import _lib

    # 7 pointers, int64, 6 int32, Cmat pointer, finish_below double, 13 output pointers
    assert ctypes.sizeof(_lib.PrimalFitArgs) == 7 * 8 + 8 + 6 * 4 + 8 + 8 + 13 * 8
    assert _lib.PrimalFitArgs.n.offset == 56 and _lib.PrimalFitArgs.flags.offset == 84
    assert _lib.PrimalFitArgs.Cmat.offset == 88 and _lib.PrimalFitArgs.finish_below.offset == 96
    assert _lib.PrimalFitArgs.beta.offset == 104
    # 4 pointers, int64, 4 int32, 11 pointers
    assert ctypes.sizeof(_lib.DualFitArgs) == 4 * 8 + 8 + 4 * 4 + 11 * 8
    assert _lib.DualFitArgs.alpha.offset == 56


def test_struct_layout_matches_the_compiler(tmp_path):
    """offsetof / sizeof as gcc lays the header's structs out == the ctypes mirror (field by field)."""
    import subprocess

    from neo_ls_svm_amd import _lib

    def prog(struct, fields):
        lines = "\n".join(f'  printf("{struct}.{f} %zu\\n", offsetof({struct}, {f}));' for f in fields)
        return f'  printf("sizeof_{struct} %zu\\n", sizeof({struct}));\n{lines}\n'

    pf = [f[0] for f in _lib.PrimalFitArgs._fields_]
    df = [f[0] for f in _lib.DualFitArgs._fields_]
    gf = [f[0] for f in _lib.SigmaGrid._fields_]
    src = tmp_path / "layout.c"
    src.write_text(
        '#include <stdio.h>\n#include <stddef.h>\n#include "neolssvm_hip.h"\nint main(void) {\n'
        + prog("nls_primal_fit_args", pf) + prog("nls_dual_fit_args", df) + prog("nls_sigma_grid", gf) + "  return 0;\n}\n"
    )
    exe = tmp_path / "layout"
    subprocess.run(["gcc", "-I", str(ROOT / "include"), str(src), "-o", str(exe)], check=True)
    out = dict(line.split() for line in subprocess.run([str(exe)], capture_output=True, text=True, check=True).stdout.splitlines())
    assert int(out["sizeof_nls_primal_fit_args"]) == ctypes.sizeof(_lib.PrimalFitArgs)
    assert int(out["sizeof_nls_dual_fit_args"]) == ctypes.sizeof(_lib.DualFitArgs)
    for f in pf:
        assert int(out[f"nls_primal_fit_args.{f}"]) == getattr(_lib.PrimalFitArgs, f).offset, f
    for f in df:
        assert int(out[f"nls_dual_fit_args.{f}"]) == getattr(_lib.DualFitArgs, f).offset, f
    assert int(out["sizeof_nls_sigma_grid"]) == ctypes.sizeof(_lib.SigmaGrid)
    for f in gf:
        assert int(out[f"nls_sigma_grid.{f}"]) == getattr(_lib.SigmaGrid, f).offset, f


def test_estimator_hands_its_device_to_the_prestep(monkeypatch):
    """NeoLSSVM(device=k) must run the pre-step's GPU statistics and the solver on context k only (one upload of X):
    no call may fall back to the default context of device 0."""
    import neo_ls_svm_amd.estimator as est
    from neo_ls_svm_amd import _prestep

    asked = []

    class FakeCtx:
        device = 1

        def hold(self, a):
            import contextlib

            return contextlib.nullcontext()

    fake = FakeCtx()

    def fake_default(device=0):
        asked.append(device)
        if device != 1:
            raise AssertionError(f"default_context({device}) requested by an estimator with device=1")
        return fake

    seen = []

    def fake_bin_stats(X, labels, sw=None, ctx=None):
        seen.append(ctx)
        lab = np.asarray(labels)
        w = np.ones(len(lab)) if sw is None else np.asarray(sw)
        cen, spr = [], []
        for b in range(lab.max() + 1):
            m = lab == b
            mu = _prestep.weighted_median_columns(X[m], w[m] / w[m].sum())
            cen.append(mu[0])
            spr.append(((w[m] / w[m].sum())[None, :] @ np.abs(X[m] - mu))[0])
        return np.array(cen), np.array(spr)

    class Stop(Exception):
        pass

    def fake_fit(*a, ctx=None, **k):
        seen.append(ctx)
        raise Stop

    monkeypatch.setattr(est, "default_context", fake_default)
    monkeypatch.setattr(est.hotpath, "bin_stats", fake_bin_stats)
    monkeypatch.setattr(est.hotpath, "primal_fit", fake_fit)
    rng = np.random.default_rng(0)
    X, y = rng.standard_normal((1500, 6)), rng.standard_normal(1500)
    with pytest.raises(Stop):
        est.NeoLSSVM(device=1, dual=False).fit(X, y)
    assert asked and set(asked) == {1}
    assert len(seen) >= 2 and all(c is fake for c in seen)


def test_no_torch_in_the_product():
    """The product and bench.py are torch-free (north star: ctypes + HIP + RCCL); torch appears in test workers only."""
    import ast

    files = list((ROOT / "neo_ls_svm_amd").glob("*.py")) + [ROOT / "bench.py", ROOT / "__graft_entry__.py"]
    for f in files:
        tree = ast.parse(f.read_text())
        for node in ast.walk(tree):
            names = []
            if isinstance(node, ast.Import):
                names = [a.name for a in node.names]
            elif isinstance(node, ast.ImportFrom):
                names = [node.module or ""]
            assert not any(n == "torch" or n.startswith("torch.") for n in names), f"{f} imports torch"


def test_rendezvous_file_exchange(tmp_path, monkeypatch):
    """distributed.exchange_unique_id: rank 0 publishes 128 bytes atomically, the others read exactly those."""
    import threading

    from neo_ls_svm_amd import distributed

    monkeypatch.setenv("NLS_RENDEZVOUS_DIR", str(tmp_path))

    class Ctx:
        def comm_unique_id(self):
            return bytes(range(128))

    got = {}

    def reader(rank):
        got[rank] = distributed.exchange_unique_id(Ctx(), rank, 3, key="t1", timeout=20)

    ts = [threading.Thread(target=reader, args=(r,)) for r in (1, 2)]
    for t in ts:
        t.start()
    got[0] = distributed.exchange_unique_id(Ctx(), 0, 3, key="t1")
    for t in ts:
        t.join()
    assert got[0] == got[1] == got[2] == bytes(range(128))
    with pytest.raises(TimeoutError):
        distributed.exchange_unique_id(Ctx(), 1, 2, key="absent", timeout=0.2)
    # no explicit key: parent-pid file plus a port-keyed, time-stamped fallback for launchers with an intermediate process
    monkeypatch.setenv("MASTER_PORT", "34567")
    assert distributed.exchange_unique_id(Ctx(), 0, 2) == bytes(range(128))
    prim, sec = distributed._rendezvous_files(None)
    assert prim.read_bytes()[:128] == bytes(range(128)) and sec.read_bytes()[:128] == bytes(range(128))
    assert (prim.stat().st_mode & 0o777) == 0o600
    # an explicit key is the caller's own rendezvous: joined whatever its age (a rank may arrive long after rank 0 published) - when it
    # carries THIS launch's nonce ...
    old = bytes(range(1, 129)) + repr(0.0).encode() + b"|" + distributed._launch_nonce(explicit_key=True)
    (tmp_path / "nls_rccl_id_job.1").write_bytes(old)
    os.utime(tmp_path / "nls_rccl_id_job.1", (1.0, 1.0))
    assert distributed.exchange_unique_id(Ctx(), 1, 2, key="job.1", timeout=0.2) == bytes(range(1, 129))
    # ... and NOT when a previous launch that died before its post-barrier unlink left it there (another nonce, or none at all): the
    # ranks of the next launch with the same key keep waiting for their own rank 0 instead of joining a dead communicator id
    for stale in (bytes(range(1, 129)) + repr(0.0).encode() + b"|4242_1_none_0", bytes(range(1, 129)) + repr(0.0).encode()):
        (tmp_path / "nls_rccl_id_job.1").write_bytes(stale)
        with pytest.raises(TimeoutError):
            distributed.exchange_unique_id(Ctx(), 1, 2, key="job.1", timeout=0.2)
    monkeypatch.setenv("NLS_RENDEZVOUS_NONCE", "4242_1_none_0")  # (a launcher may set the nonce itself)
    (tmp_path / "nls_rccl_id_job.1").write_bytes(bytes(range(1, 129)) + repr(0.0).encode() + b"|4242_1_none_0")
    assert distributed.exchange_unique_id(Ctx(), 1, 2, key="job.1", timeout=0.2) == bytes(range(1, 129))
    monkeypatch.delenv("NLS_RENDEZVOUS_NONCE")
    # ... and a key with a dot keeps its whole name while it is being published (the temporary file is <name>.tmp<pid>)
    assert distributed.exchange_unique_id(Ctx(), 0, 2, key="job.1") == bytes(range(128))
    assert (tmp_path / "nls_rccl_id_job.1").read_bytes()[:128] == bytes(range(128)) and not list(tmp_path.glob("*.tmp*"))
    # an automatic (launcher-derived) name left behind by a launch that died after publishing is stale - by the FILE's own
    # modification time, not by a clock reading inside the payload: it is not joined
    age = 10 * distributed._fresh_seconds()
    os.utime(prim, (prim.stat().st_mtime - age, prim.stat().st_mtime - age))
    os.utime(sec, (sec.stat().st_mtime - age, sec.stat().st_mtime - age))
    with pytest.raises(TimeoutError):
        distributed.exchange_unique_id(Ctx(), 1, 2, timeout=0.2)
    monkeypatch.setenv("NLS_RENDEZVOUS_FRESH_SECONDS", str(100 * age))  # read at call time: a wider window accepts it again
    assert distributed.exchange_unique_id(Ctx(), 1, 2, timeout=0.2) == bytes(range(128))
    monkeypatch.delenv("NLS_RENDEZVOUS_FRESH_SECONDS")
    assert distributed.exchange_unique_id(Ctx(), 0, 2) == bytes(range(128))  # published again: fresh
    prim.unlink()  # a rank whose parent pid differs would not find the primary: after 15 s it takes the fresh secondary
    monkeypatch.setattr(distributed.time, "monotonic", iter([0.0, 16.0, 17.0, 18.0]).__next__)
    assert distributed.exchange_unique_id(Ctx(), 1, 2, timeout=100) == bytes(range(128))


def test_rendezvous_explicit_key_joins_ranks_of_different_parents(tmp_path, monkeypatch):
    """An explicit key is the rendezvous of ranks that are NOT children of one launcher (several nodes on a shared directory, per-node
    daemons, hand-started ranks): a reader whose parent pid differs from the publisher's must still join the payload - the default nonce
    under an explicit key holds only what every rank of the launch shares (address, port, run id, restart count)."""
    import threading

    from neo_ls_svm_amd import distributed

    monkeypatch.setenv("NLS_RENDEZVOUS_DIR", str(tmp_path))
    monkeypatch.setenv("MASTER_ADDR", "10.0.0.1")
    monkeypatch.setenv("MASTER_PORT", "29511")

    class Ctx:
        def comm_unique_id(self):
            return bytes(range(128))

    parents = {threading.get_ident(): 1000}
    monkeypatch.setattr(distributed.os, "getppid", lambda: parents.get(threading.get_ident(), 1))
    got = {}

    def reader(rank, ppid):
        parents[threading.get_ident()] = ppid
        got[rank] = distributed.exchange_unique_id(Ctx(), rank, 3, key="multinode", timeout=20)

    ts = [threading.Thread(target=reader, args=(r, 2000 + r)) for r in (1, 2)]  # two readers, two other parents
    for t in ts:
        t.start()
    got[0] = distributed.exchange_unique_id(Ctx(), 0, 3, key="multinode")
    for t in ts:
        t.join()
    assert got == {0: bytes(range(128)), 1: bytes(range(128)), 2: bytes(range(128))}
    # another launch (another port) with the same key does not join this one's leftover
    monkeypatch.setenv("MASTER_PORT", "29512")
    with pytest.raises(TimeoutError):
        distributed.exchange_unique_id(Ctx(), 1, 2, key="multinode", timeout=0.2)
    # without a key the parent pid stays part of the nonce (and of the primary file name)
    assert distributed._launch_nonce() != distributed._launch_nonce(explicit_key=True)


def test_graft_entry_build_runs_and_agrees_on_the_abi_version():
    """``__graft_entry__.build()`` - the driver's "does it build" check - compiles (or finds up to date) the library and asserts the ABI
    version: header, library, Python mirror and that assertion must move together (round 6 bumped three of the four at first)."""
    import re

    import __graft_entry__ as g

    from neo_ls_svm_amd import _lib

    header = (ROOT / "include" / "neolssvm_hip.h").read_text()
    version = int(re.search(r"#define\s+NLS_ABI_VERSION\s+(\d+)", header).group(1))
    assert version == _lib.ABI_VERSION
    g.build()  # raises on a mismatch (and when the sources do not compile)


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


def test_sigma_grid_bookkeeping_hooks():
    """The gamma x sigma grid runs inside the library (``nls_primal_fit_grid``); its bookkeeping - which sigmas a rank visits in which
    order, and the selection over the (merged or unmerged) objective table with numpy's argmin / nanmin semantics and the two tie rules -
    is host arithmetic behind two hooks, checked here against the Python restatement of the round-4 driver (``tests/_grid_reference_driver``).
    The finish-if-below loop itself needs fits: ``tests/test_gpu_primal.py::test_sigma_grid_driver`` runs both drivers on the GPU."""
    import sys

    sys.path.insert(0, str(ROOT / "tests"))
    import _grid_reference_driver as ref
    from neo_ls_svm_amd import _lib

    lib = _lib.load_library()
    rng = np.random.default_rng(0)
    for S in (1, 3, 7, 16):
        sig = np.exp(rng.uniform(-1.4, 1.4, S))
        if S >= 3:
            sig[1] = 1.0 / sig[0]  # |ln sigma| ties: broken by index
        for world in (1, 2, 3, 8):
            for rank in range(world):
                order = np.full(S, -1, dtype=np.int32)
                cnt = lib.nls_grid_visiting_order(sig.ctypes.data, S, rank, world, order.ctypes.data)
                assert list(order[:cnt]) == ref.visiting_order(sig, rank, world)
    assert lib.nls_grid_visiting_order(sig.ctypes.data, 16, 3, 3, order.ctypes.data) == -1  # rank outside the world

    def select(obj, owned, incumbent):
        k, g = ctypes.c_int32(), ctypes.c_int32()
        own = np.ascontiguousarray(owned, dtype=np.uint8)
        rc = lib.nls_grid_select(obj.ctypes.data, own.ctypes.data, obj.shape[0], obj.shape[1], -1 if incumbent is None else incumbent,
                                 ctypes.byref(k), ctypes.byref(g))  # fmt: skip
        assert rc == 0
        return k.value, g.value

    S, G = 7, 5
    for trial in range(200):
        obj = np.ascontiguousarray(rng.integers(0, 4, (S, G)).astype(np.float64))  # many exact ties
        world = int(rng.integers(1, 4))
        rank = int(rng.integers(0, world))
        owned = np.zeros(S, dtype=bool)
        owned[rank::world] = True
        merged = bool(rng.integers(0, 2))
        if merged:
            owned[:] = True
        else:
            obj[~owned] = np.nan
        if trial % 7 == 0:
            obj[int(np.flatnonzero(owned)[0]), 2] = np.nan  # a NaN inside an owned row: nanmin skips it, argmin of the row returns it
        cand = np.flatnonzero(owned)
        incumbent = None if merged else int(rng.choice(cand))
        assert select(obj, owned, incumbent) == ref.select(obj, owned, incumbent), (trial, obj, owned, incumbent)
    # the merged rule does not depend on any rank's incumbent: the smallest tied index
    obj = np.ones((S, G))
    obj[4, 2] = obj[1, 3] = obj[2, 0] = 0.5
    assert select(obj, np.ones(S, dtype=bool), None) == (1, 3)
    assert select(obj, np.ones(S, dtype=bool), 4) == (4, 2)  # unmerged: the finished incumbent keeps a tie


def test_sigma_grid_reference_driver_bookkeeping():
    """The checker itself (``tests/_grid_reference_driver.grid``) over a deterministic stand-in solver: sigmas dealt round-robin over
    ranks, tables merged with a sum (every sigma has one owner), only a sigma that beats the rank's incumbent is finished."""
    import sys

    sys.path.insert(0, str(ROOT / "tests"))
    import _grid_reference_driver as ref

    S, G = 7, 5
    sig = np.linspace(0.5, 2.0, S)
    rng = np.random.default_rng(0)
    curves = rng.uniform(1.0, 2.0, (S, G))
    curves[4, 2] = 0.5  # the joint minimum
    calls = []

    def fake_fit(B, finish_below):
        k = int(np.argmin(np.abs(sig - 1.0 / B[0, 0])))
        obj = curves[k]
        opt = int(np.argmin(obj))
        finished = finish_below is None or obj[opt] < finish_below
        calls.append((k, finish_below, finished))
        r = {"loo_errors_gammas": obj + 10.0, "objective": obj, "opt": opt, "finished": finished, "timings": {"total": 1.0, "gram": 0.5}}
        if finished:
            r["beta"] = np.full(3, float(k))
        return r

    B = np.ones((1, 1))
    world = 3
    gam = np.arange(1.0, G + 1)
    contribs = []
    for rank in range(world):  # pass 1: what every rank hands to the all-reduce
        ref.grid(fake_fit, B, sig, gam, rank, world, allreduce_sum=lambda a: contribs.append(a.copy()) or a)
    total = sum(contribs)
    calls.clear()
    parts = []
    for rank in range(world):  # pass 2: every rank receives the sum
        g = ref.grid(fake_fit, B, sig, gam, rank, world, allreduce_sum=lambda a: total.copy())
        parts.append(g)
        assert (g["sigma_index"], g["gamma_index"]) == (4, 2)
        assert np.allclose(g["objective"], curves) and np.allclose(g["loo_errors"], curves + 10.0)
    assert np.allclose(total[2 * S * G :], 1.0)  # seconds per sigma: every sigma timed exactly once
    owner = 4 % world
    assert parts[owner]["best"] is not None and parts[owner]["best"]["beta"][0] == 4.0
    assert all(parts[r]["best"] is None for r in range(world) if r != owner)
    for rank in range(world):  # a sigma is finished only when it beats the incumbent of its rank
        mine = [c for c in calls if c[0] % world == rank]
        best = np.inf
        for k, bound, finished in mine:
            assert (bound is None) == (best == np.inf)
            assert finished == (curves[k].min() < best)
            best = min(best, curves[k].min()) if finished else best
    g = ref.grid(fake_fit, B, sig, gam)
    assert (g["sigma_index"], g["gamma_index"]) == (4, 2) and g["best"]["beta"][0] == 4.0 and g["finished_count"] >= 1


def test_pooled_factor_outputs_are_recycled_only_when_nobody_holds_them():
    """Large factor outputs (``L_``) come from a pool of anonymous mappings (``_hostpool``): an array and its views keep their mapping; once all are
    gone the NEXT output of that size reuses it (no munmap / mmap / first-touch faults per fit); small outputs are plain arrays; ``release`` unmaps."""
    import gc

    from neo_ls_svm_amd import _hostpool as pool

    pool.release()
    n = 2900  # 2900^2 doubles = 67 MB >= MIN_BYTES
    a = pool.factor_output((n, n), np.float64)
    assert a.shape == (n, n) and a.flags["C_CONTIGUOUS"] and a.flags["WRITEABLE"] and not a.any()
    a[:] = 3.0
    addr = a.__array_interface__["data"][0]
    view = a[5:7]
    b = pool.factor_output((n, n), np.float64)  # a is alive: a second mapping
    assert b.__array_interface__["data"][0] != addr
    del a
    gc.collect()
    assert pool._pooled_bytes() == 0 and view[0, 0] == 3.0  # the view keeps the mapping
    del view
    gc.collect()
    assert pool._pooled_bytes() == n * n * 8
    c = pool.factor_output((n, n), np.float64)
    assert c.__array_interface__["data"][0] == addr  # recycled (its contents are the previous factor's: only the defined triangle is output)
    import pickle

    assert pickle.loads(pickle.dumps(c[:2])).shape == (2, n)
    small = pool.factor_output((10, 10), np.complex128)
    assert small.base is None and not small.any()
    del b, c
    gc.collect()
    pool.release()
    assert pool._pooled_bytes() == 0
