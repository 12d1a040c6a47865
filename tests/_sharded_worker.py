"""Worker of the world_size-2 tests (launched by the test files with RANK / WORLD_SIZE / MASTER_* set).

mode "cpu": no GPU.  Exercises the product's row_shard and a staged all-reduce over gloo, and replays the
            library's exchange points with the oracle standing in for the HIP stages (test infrastructure).
mode "gpu": both ranks share GPU 0 (RCCL refuses two ranks on one device, so the collectives go through the library's
            caller-supplied all-reduce hook, staged over gloo); the real library runs its sharded path.
torch is test infrastructure here (gloo rendezvous); the product itself is torch-free.
Prints "OK <rank>" on success.
"""

from __future__ import annotations

import ctypes
import os
import sys
from pathlib import Path

ROOT = Path(__file__).resolve().parent.parent
sys.path[:0] = [str(ROOT), str(ROOT / "oracle"), str(ROOT / "tests")]

import numpy as np  # noqa: E402

torch = dist = None  # imported by the gloo modes only (main): the native-communicator modes need no torch at all


class FakeLib:
    """Host-memory stand-in for the two memcpy entry points the staged hook uses."""

    @staticmethod
    def nls_memcpy_d2h(_h, dst, src, nbytes):
        ctypes.memmove(dst, src, nbytes)
        return 0

    @staticmethod
    def nls_memcpy_h2d(_h, dst, src, nbytes):
        ctypes.memmove(dst, src, nbytes)
        return 0


class FakeCtx:
    lib, handle, device = FakeLib(), None, 0

    @staticmethod
    def _check(rc):
        assert rc == 0


def make_allreduce(ctx):
    """fn(device_ptr, count): sum over the gloo ranks, staged through host memory (nls_memcpy_d2h / h2d)."""

    def fn(ptr: int, count: int) -> None:
        host = np.empty(count, dtype=np.float64)
        ctx._check(ctx.lib.nls_memcpy_d2h(ctx.handle, host.ctypes.data, ctypes.c_void_p(ptr), host.nbytes))
        t = torch.from_numpy(host)
        dist.all_reduce(t, op=dist.ReduceOp.SUM)
        ctx._check(ctx.lib.nls_memcpy_h2d(ctx.handle, ctypes.c_void_p(ptr), host.ctypes.data, host.nbytes))

    return fn


def problem(n=1800, d=9, D=64, clf=False):
    rng = np.random.default_rng(123)
    X = rng.standard_normal((n, d))
    w = rng.standard_normal(d) / np.sqrt(d)
    y = np.where(X @ w > 0, 1.0, -1.0) if clf else np.sin(X @ w) + 0.1 * rng.standard_normal(n)
    s = rng.uniform(0.2, 2.0, n)
    import neolssvm_oracle as orc

    B = orc.orf_frequencies(d, D) * 0.4
    return X, y, s, rng.standard_normal(d) * 0.1, rng.uniform(0.8, 1.2, d), B


def run_cpu(rank, world):
    import neolssvm_oracle as orc

    from neo_ls_svm_amd.distributed import row_shard

    hook = make_allreduce(FakeCtx())

    def allreduce(a):  # in place on a float64 (or complex128 viewed as float64) host array
        v = a.view(np.float64).reshape(-1)
        hook(v.ctypes.data, v.size)

    # hook data path
    buf = np.arange(10, dtype=np.float64) * (rank + 1)
    allreduce(buf)
    assert np.array_equal(buf, np.arange(10) * sum(range(1, world + 1)))

    for clf in (False, True):
        X, y, s, shift, scale, B = problem(clf=clf)
        n, D1 = X.shape[0], B.shape[1] + 1
        gam = orc.gamma_grid(64)
        ref = orc.primal_fit_streamed(X, y, s, shift, scale, B, clf, gammas=gam)
        lo, hi = row_shard(n, rank, world)
        Xl, yl, sl = X[lo:hi], y[lo:hi], s[lo:hi]
        # exchange point 1: weight sums and n
        sums = np.array([sl.sum(), (sl * yl).sum(), float(hi - lo)])
        allreduce(sums)
        sn = sl / sums[0]
        c = 1.0 / (sums[2] * D1)
        # exchange point 2: A || b
        phi = orc.feature_map(Xl, shift, scale, B)
        F = sn[:, None] * phi
        Ab = np.ascontiguousarray(np.concatenate([F.conj().T @ F, (F.conj().T @ (sn * yl))[:, None]], axis=1))
        allreduce(Ab)
        A, b = (Ab[:, :D1] + Ab[:, :D1].conj().T) / 2, Ab[:, D1]
        lam, Q = np.linalg.eigh(A / c)
        v = (Q.conj().T @ b) / c
        P = phi @ Q
        R = 1.0 / (gam[None, :] + lam[:, None])
        num = np.ascontiguousarray(np.real(P * v[None, :])) @ R
        hs = ((P.real**2 + P.imag**2) @ R) / c
        e = (num - yl[:, None]) / (1 - (sn[:, None] ** 2) * hs)
        if clf:
            e = orc.clip_classifier_residuals(e, yl)
        # exchange point 3: per-gamma error vectors
        ae = np.abs(e)
        vec = np.stack([sn @ ae, sn @ (ae >= 1), sn @ np.maximum(0, ae - 1)])
        allreduce(vec)
        obj = vec[1] + vec[2] + vec[0] if clf else vec[0]
        opt = int(np.argmin(obj))
        assert opt == ref["opt"], (opt, ref["opt"])
        assert np.max(np.abs(vec[0] - ref["loo_errors_gammas"])) < 1e-10 * np.max(ref["loo_errors_gammas"])
        assert np.max(np.abs(e[:, opt] - ref["loo_residuals"][lo:hi])) < 1e-9 * np.max(np.abs(ref["loo_residuals"]))
        import scipy.linalg as sla

        beta = sla.cho_solve(sla.cho_factor(gam[opt] * c * np.eye(D1) + A), b)
        assert np.max(np.abs(beta - ref["beta"])) < 1e-9 * np.max(np.abs(ref["beta"]))


def run_gpu(rank, world):
    import neo_ls_svm_amd as hp
    from neo_ls_svm_amd.distributed import row_shard

    ctx = hp.Context(0)
    ctx.set_allreduce(make_allreduce(ctx), rank, world)
    for clf in (False, True):
        X, y, s, shift, scale, B = problem(n=5000, d=12, D=200, clf=clf)
        lo, hi = row_shard(X.shape[0], rank, world)
        r = hp.primal_fit(X[lo:hi], y[lo:hi], s[lo:hi], shift, scale, B, clf, ctx=ctx)
        A, b = hp.gram(X[lo:hi], y[lo:hi], s[lo:hi], shift, scale, B, ctx=ctx)
        # single-rank reference on the full rows with a hook-free context
        solo = hp.Context(0)
        r1 = hp.primal_fit(X, y, s, shift, scale, B, clf, ctx=solo)
        A1, b1 = hp.gram(X, y, s, shift, scale, B, ctx=solo)
        solo.close()

        def rel(a, bb):
            return float(np.max(np.abs(a - bb)) / np.max(np.abs(bb)))

        assert rel(A, A1) < 1e-12 and rel(b, b1) < 1e-12
        assert r["opt"] == r1["opt"]
        assert rel(r["beta"], r1["beta"]) < 1e-8
        assert rel(r["loo_errors_gammas"], r1["loo_errors_gammas"]) < 1e-10
        assert rel(r["L"][np.triu_indices(B.shape[1] + 1)], r1["L"][np.triu_indices(B.shape[1] + 1)]) < 1e-9
        for k in ("loo_residuals", "loo_leverage", "loo_std", "residuals"):
            assert r[k].shape == (hi - lo,)
            assert rel(r[k], r1[k][lo:hi]) < 1e-8, k
        assert abs(r["loo_score"] - r1["loo_score"]) < 1e-10
    ctx.close()


def run_gpu_native(rank, world, expect_failure=False):
    """All ranks share GPU 0 and join a NATIVE communicator (ctx->comm != NULL: the library's own ncclAllReduce / ncclBroadcast /
    grouped-broadcast call sites), served by the test stand-in for librccl that NLS_RCCL_LIB points at (tests/csrc/rccl_shim.cpp).
    Only rank 0 asks for the factor L_ (as bench.py does); every rank must end with the same beta."""
    import neo_ls_svm_amd as hp
    from neo_ls_svm_amd._lib import NLS_ERR_COMM, NlsError
    from neo_ls_svm_amd.distributed import init_from_env, row_shard

    ctx = hp.Context(0)
    init_from_env(ctx)
    assert ctx.comm_world == world
    got = ctx.comm_allreduce([float(rank + 1), 1.0], "sum")
    assert np.array_equal(got, [world * (world + 1) / 2, float(world)])
    assert ctx.comm_allreduce([float(rank)], "max")[0] == world - 1
    for clf in (False, True):
        X, y, s, shift, scale, B = problem(n=5000, d=12, D=200, clf=clf)
        lo, hi = row_shard(X.shape[0], rank, world)
        if expect_failure:
            try:
                hp.primal_fit(X[lo:hi], y[lo:hi], s[lo:hi], shift, scale, B, clf, ctx=ctx, want_L=(rank == 0))
            except NlsError as exc:  # NLS_ERR_COMM -> NlsError (RuntimeError); the message names the RCCL call
                assert "ncclBroadcast" in str(exc) or "Broadcast" in str(exc), str(exc)
                assert ctx.comm_state == "aborted"  # a failed RCCL call costs the context its communicator
                ctx.close()
                return
            raise AssertionError("the injected ncclBroadcast failure did not surface")
        r = hp.primal_fit(X[lo:hi], y[lo:hi], s[lo:hi], shift, scale, B, clf, ctx=ctx, want_L=(rank == 0))
        A, b = hp.gram(X[lo:hi], y[lo:hi], s[lo:hi], shift, scale, B, ctx=ctx)
        solo = hp.Context(0)  # single-rank reference on all rows, no communicator
        r1 = hp.primal_fit(X, y, s, shift, scale, B, clf, ctx=solo)
        A1, b1 = hp.gram(X, y, s, shift, scale, B, ctx=solo)
        solo.close()

        def rel(a, bb):
            return float(np.max(np.abs(a - bb)) / np.max(np.abs(bb)))

        assert rel(A, A1) < 1e-12 and rel(b, b1) < 1e-12
        assert r["opt"] == r1["opt"]
        assert rel(r["beta"], r1["beta"]) < 1e-8
        assert rel(r["lam"], r1["lam"]) < 1e-9
        assert rel(r["loo_errors_gammas"], r1["loo_errors_gammas"]) < 1e-10
        if rank == 0:
            iu = np.triu_indices(B.shape[1] + 1)
            assert rel(r["L"][iu], r1["L"][iu]) < 1e-9
        else:
            assert "L" not in r
        for k in ("loo_residuals", "loo_leverage", "loo_std", "residuals"):
            assert r[k].shape == (hi - lo,)
            assert rel(r[k], r1[k][lo:hi]) < 1e-8, k
        assert abs(r["loo_score"] - r1["loo_score"]) < 1e-10
        # every rank holds the SAME beta (rank 0's, broadcast): sum over ranks == world x own
        parts = np.concatenate([r["beta"].real, r["beta"].imag])[:64]
        tot = ctx.comm_allreduce(parts, "sum")
        assert np.array_equal(tot, world * parts) or rel(tot, world * parts) < 1e-15
    ctx.close()


def _native_ctx(rank, world):
    import neo_ls_svm_amd as hp
    from neo_ls_svm_amd.distributed import init_from_env

    ctx = hp.Context(0)
    init_from_env(ctx)
    assert ctx.comm_world == world and ctx.comm_state == "joined"
    return hp, ctx


def run_gpu_fault(rank, world):
    """ONE rank fails locally (NLS_FAULT_INJECT="site:rank[:code]": an allocation / launch / factorisation failure as the library sees it) in the
    middle of a sharded fit.  Every rank must return an error - the failed one its own, the others NLS_ERR_COMM naming it - at the next status
    vote, i.e. at once (long before the collective deadline), and the SAME communicator must carry the next fit."""
    import time

    import numpy.linalg as npl

    from neo_ls_svm_amd._lib import NlsError
    from neo_ls_svm_amd.distributed import row_shard

    spec = os.environ["NLS_FAULT_INJECT"]
    site, bad, *code = spec.split(":")
    bad, linalg = int(bad), (code and code[0] == "3")
    hp, ctx = _native_ctx(rank, world)
    X, y, s, shift, scale, B = problem(n=5000, d=12, D=200, clf=False)
    lo, hi = row_shard(X.shape[0], rank, world)
    t0 = time.monotonic()
    try:
        hp.primal_fit(X[lo:hi], y[lo:hi], s[lo:hi], shift, scale, B, False, ctx=ctx, want_L=(rank == 0))
    except (NlsError, npl.LinAlgError) as exc:
        waited, msg = time.monotonic() - t0, str(exc)
        assert waited < 45.0, f"rank {rank} waited {waited:.1f} s: the deadline, not the vote, ended the call"
        if linalg:
            assert isinstance(exc, npl.LinAlgError), (type(exc), msg)  # a property of the shared problem: the same error everywhere
        else:
            assert isinstance(exc, NlsError), (type(exc), msg)
        if rank == bad:
            assert "injected fault" in msg and f"'{site}'" in msg, msg
        else:
            assert f"rank {bad} of {world} failed with" in msg and ("NLS_ERR_LINALG" if linalg else "NLS_ERR_HIP") in msg, msg
    else:
        raise AssertionError(f"rank {rank}: the injected fault at {spec} did not surface")
    assert ctx.comm_state == "joined"  # everybody left at the same vote: nothing is pending on the communicator
    del os.environ["NLS_FAULT_INJECT"]
    r = hp.primal_fit(X[lo:hi], y[lo:hi], s[lo:hi], shift, scale, B, False, ctx=ctx, want_L=(rank == 0))
    solo = hp.Context(0)
    r1 = hp.primal_fit(X, y, s, shift, scale, B, False, ctx=solo)
    solo.close()
    assert r["opt"] == r1["opt"] and np.max(np.abs(r["beta"] - r1["beta"])) < 1e-8 * np.max(np.abs(r1["beta"]))
    ctx.close()


def run_gpu_lost_rank(rank, world, how):
    """ONE rank is lost in a way no vote can carry: `call` - an RCCL call of that rank alone returns an error (NLS_SHIM_FAIL_RANK / _CALL) and the
    rank leaves; `dead` - its process exits before the fit.  The others are inside a collective with a peer that will never come: the
    library's deadline (NLS_COMM_TIMEOUT_S, set short by the test) must end their wait with NLS_ERR_COMM and an aborted communicator, which
    then refuses further collective work at once; a new communicator cannot include the lost rank, so the survivors stop there."""
    import time

    from neo_ls_svm_amd._lib import NlsError
    from neo_ls_svm_amd.distributed import row_shard

    bad = int(os.environ["NLS_TEST_LOST_RANK"])
    deadline = float(os.environ["NLS_COMM_TIMEOUT_S"])
    hp, ctx = _native_ctx(rank, world)
    X, y, s, shift, scale, B = problem(n=5000, d=12, D=200, clf=False)
    lo, hi = row_shard(X.shape[0], rank, world)
    if how == "dead" and rank == bad:
        sys.stdout.flush()
        os._exit(7)  # (no interpreter shutdown: the communicator is simply gone, as after a crash)
    t0 = time.monotonic()
    try:
        hp.primal_fit(X[lo:hi], y[lo:hi], s[lo:hi], shift, scale, B, False, ctx=ctx, want_L=(rank == 0))
    except NlsError as exc:
        waited, msg = time.monotonic() - t0, str(exc)
        if rank == bad:
            assert "failed" in msg and "injected" in msg and waited < 30.0, (waited, msg)
        else:
            assert "did not complete within" in msg and "NLS_COMM_TIMEOUT_S" in msg, msg
            assert 0.5 * deadline < waited < deadline + 60.0, waited
    else:
        raise AssertionError(f"rank {rank}: the fit returned although rank {bad} was lost")
    assert ctx.comm_state == "aborted"
    t0 = time.monotonic()
    try:
        hp.primal_fit(X[lo:hi], y[lo:hi], s[lo:hi], shift, scale, B, False, ctx=ctx, want_L=False)
    except NlsError as exc:
        assert "aborted" in str(exc) and time.monotonic() - t0 < 5.0, str(exc)
    else:
        raise AssertionError("a context with an aborted communicator must refuse a sharded fit")
    ctx.comm_destroy()  # leaving the communicator makes the context a single rank again
    assert ctx.comm_state == "none"
    r = hp.primal_fit(X[:500], y[:500], s[:500], shift, scale, B, False, ctx=ctx, want_L=False)
    assert r["beta"].shape == (B.shape[1] + 1,)
    ctx.close()


def run_gpu_grid_fault(rank, world):
    """The sigma-sharded grid (``nls_primal_fit_grid`` with a merge context): ONE rank's own fits fail (only that process sets NLS_FAULT_INJECT);
    without a vote its peers would wait in the merge's all-reduces for the deadline.  Every rank must return at once, and the same merge
    communicator must carry the next grid."""
    import time

    from neo_ls_svm_amd._lib import NlsError

    bad = int(os.environ["NLS_TEST_LOST_RANK"])
    hp, cctx = _native_ctx(rank, world)  # the communicator-only context of the merge
    ctx = hp.Context(0)
    X, y, s, shift, scale, B = problem(n=3000, d=8, D=64, clf=False)
    sig, gam = np.logspace(-0.5, 0.5, 8), hp.gamma_grid(1024)[::33]
    if rank == bad:
        os.environ["NLS_FAULT_INJECT"] = "sweep"
    t0 = time.monotonic()
    try:
        hp.primal_fit_sigma_grid(X, y, s, shift, scale, B, False, sig, gammas=gam, ctx=ctx, rank=rank, world=world, merge_ctx=cctx, want_L=False)
    except NlsError as exc:
        waited, msg = time.monotonic() - t0, str(exc)
        assert waited < 45.0, f"rank {rank} waited {waited:.1f} s"
        assert ("injected fault" in msg) if rank == bad else (f"rank {bad} of {world} failed with NLS_ERR_HIP" in msg and "merge" in msg), msg
    else:
        raise AssertionError(f"rank {rank}: the grid returned although rank {bad}'s fits failed")
    os.environ.pop("NLS_FAULT_INJECT", None)
    assert cctx.comm_state == "joined"
    g = hp.primal_fit_sigma_grid(X, y, s, shift, scale, B, False, sig, gammas=gam, ctx=ctx, rank=rank, world=world, merge_ctx=cctx, want_L=False)
    g1 = hp.primal_fit_sigma_grid(X, y, s, shift, scale, B, False, sig, gammas=gam, ctx=ctx, want_L=False)  # the whole grid on this rank alone
    assert (g["sigma_index"], g["gamma_index"]) == (g1["sigma_index"], g1["gamma_index"])
    ctx.close()
    cctx.close()


if __name__ == "__main__":
    mode = sys.argv[1]
    rank, world = int(os.environ["RANK"]), int(os.environ["WORLD_SIZE"])
    if mode in ("gpu_rccl", "gpu_rccl_fail"):
        run_gpu_native(rank, world, expect_failure=(mode == "gpu_rccl_fail"))
        print(f"OK {rank}", flush=True)
        sys.exit(0)
    if mode == "gpu_rccl_fault":
        run_gpu_fault(rank, world)
        print(f"OK {rank}", flush=True)
        sys.exit(0)
    if mode == "gpu_grid_fault":
        run_gpu_grid_fault(rank, world)
        print(f"OK {rank}", flush=True)
        sys.exit(0)
    if mode in ("gpu_rccl_lost_call", "gpu_rccl_lost_dead"):
        run_gpu_lost_rank(rank, world, mode.rsplit("_", 1)[1])
        print(f"OK {rank}", flush=True)
        sys.exit(0)
    import torch  # noqa: F811
    import torch.distributed as dist  # noqa: F811

    globals().update(torch=torch, dist=dist)
    dist.init_process_group("gloo", rank=rank, world_size=world)
    try:
        (run_cpu if mode == "cpu" else run_gpu)(rank, world)
        dist.barrier()
        print(f"OK {rank}", flush=True)
    finally:
        dist.destroy_process_group()
