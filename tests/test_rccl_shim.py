"""The library's NATIVE collective path (``ctx->comm != NULL``) with more than one rank, on ONE GPU.

RCCL refuses two ranks per device and the pool's boxes have one GPU, so without this the ``ncclAllReduce`` / ``ncclBroadcast`` /
grouped-broadcast call sites of ``csrc/nls_host.h`` and ``csrc/nls_evd.hip`` would first run with world > 1 on an 8-GPU node.
``tests/csrc/rccl_shim.cpp`` is a test stand-in for ``librccl.so.1`` (shared-memory staging between processes that share GPU 0);
the library finds it through ``NLS_RCCL_LIB``.  The shim itself is checked on the CPU first (host-only build)."""

from __future__ import annotations

import json
import os
import socket
import subprocess
import sys
import tempfile
from pathlib import Path

import pytest

HERE = Path(__file__).resolve().parent
SHIM_SRC = HERE / "csrc" / "rccl_shim.cpp"
SHIM_DIR = HERE / "csrc" / "_shim"
HIPCC = "/opt/rocm/bin/hipcc"
# The stand-in's asynchronous mode holds a rank's stream with a spinning kernel.  HIP maps the streams of ONE process onto 4 hardware queues by
# default, and a stream that shares a queue with a held one does not run: with the ranks' main streams, the stand-in's copy streams and rank 0's
# side streams all on one device (a group: one process) that dead-locks - measured: GPU_MAX_HW_QUEUES=4 hangs, 16 / 32 do not.  (On a real node
# every rank has its own device and holds only its own queues.)
ASYNC_STAND_IN = {"NLS_SHIM_ASYNC": "1", "GPU_MAX_HW_QUEUES": "32"}


def build_shim(host_only: bool) -> Path:
    out = SHIM_DIR / ("librccl_hostonly.so" if host_only else "librccl.so.1")
    if not out.exists() or out.stat().st_mtime < SHIM_SRC.stat().st_mtime:
        SHIM_DIR.mkdir(exist_ok=True)
        # (gfx950 named: the GPU build carries a kernel and may be compiled in a container without a GPU, then travel to the box)
        cmd = [HIPCC, "-shared", "-fPIC", "-O2", "-std=c++17", str(SHIM_SRC), "-o", str(out), "-lrt"] + (["-DSHIM_HOST_ONLY"] if host_only else ["--offload-arch=gfx950"])
        subprocess.run(cmd, check=True, capture_output=True, text=True)
    return out


def _free_port():
    with socket.socket() as s:
        s.bind(("127.0.0.1", 0))
        return s.getsockname()[1]


def _run_ranks(argv, world, env_extra, timeout, expect_exit=None):
    """expect_exit: {rank: exit code} for ranks that are meant to die (they print no OK line)."""
    expect_exit = expect_exit or {}
    port = _free_port()
    procs = []
    for rank in range(world):
        env = dict(os.environ, **env_extra, RANK=str(rank), LOCAL_RANK="0", WORLD_SIZE=str(world), MASTER_ADDR="127.0.0.1", MASTER_PORT=str(port),
                   OMP_NUM_THREADS="2", OPENBLAS_NUM_THREADS="2", HSA_ENABLE_IPC_MODE_LEGACY="0")  # fmt: skip
        procs.append(subprocess.Popen([sys.executable] + argv, env=env, stdout=subprocess.PIPE, stderr=subprocess.STDOUT, text=True))
    outs = []
    for p in procs:
        try:
            out, _ = p.communicate(timeout=timeout)
        except subprocess.TimeoutExpired:
            for q in procs:
                q.kill()
            raise
        outs.append(out)
    for rank, (p, out) in enumerate(zip(procs, outs)):
        if rank in expect_exit:
            assert p.returncode == expect_exit[rank], f"rank {rank}: exit code {p.returncode}, expected {expect_exit[rank]}:\n{out[-3000:]}"
        else:
            assert p.returncode == 0 and f"OK {rank}" in out, f"rank {rank} failed:\n{out[-3000:]}"


@pytest.mark.parametrize("world, fail_at", [(2, 0), (3, 0), (3, 1), (3, 3)])
def test_shim_protocol_cpu(world, fail_at):
    """The stand-in itself: chunked in-place sum, out-of-place max, broadcast from a non-zero root, a group of unequal
    broadcasts, and the injected failure (alone and inside a group) - on host buffers, N processes."""
    lib = build_shim(host_only=True)
    with tempfile.TemporaryDirectory() as td:
        env = {"NLS_SHIM_SLOT_BYTES": "8192", "NLS_SHIM_TIMEOUT_S": "30"}
        if fail_at:
            env["NLS_SHIM_FAIL_BROADCAST"] = str(fail_at)
        _run_ranks([str(HERE / "_shim_worker.py"), str(lib), str(Path(td) / "id")], world, env, timeout=120)


def test_shim_asymmetric_failure_cpu():
    """The stand-in's ASYMMETRIC injection (``NLS_SHIM_FAIL_RANK`` / ``NLS_SHIM_FAIL_CALL``): the 2nd collective call of rank 1 alone returns an
    error; the other ranks are left in the exchange and get the stand-in's own time-out (host-only build: the calls are synchronous)."""
    lib = build_shim(host_only=True)
    with tempfile.TemporaryDirectory() as td:
        env = {"NLS_SHIM_SLOT_BYTES": "8192", "NLS_SHIM_TIMEOUT_S": "3", "NLS_SHIM_FAIL_RANK": "1", "NLS_SHIM_FAIL_CALL": "2"}
        _run_ranks([str(HERE / "_shim_worker.py"), str(lib), str(Path(td) / "id"), "asymmetric"], 3, env, timeout=120)


def test_the_gpu_build_of_the_shim_compiles_and_exports_what_the_library_binds():
    """``csrc/nls_comm.hip`` resolves ten ``nccl*`` symbols (``ncclCommAbort`` and ``ncclCommGetAsyncError`` since round 6): the stand-in must
    have them all, or the library would refuse it at load time on the GPU box."""
    import ctypes

    lib = ctypes.CDLL(str(build_shim(host_only=False)))
    for name in ("ncclGetUniqueId", "ncclCommInitRank", "ncclCommDestroy", "ncclCommAbort", "ncclCommGetAsyncError", "ncclAllReduce",
                 "ncclBroadcast", "ncclGroupStart", "ncclGroupEnd", "ncclGetErrorString"):  # fmt: skip
        assert hasattr(lib, name), name


def _launch_native(mode, world, extra_env=None, timeout=900):
    lib = build_shim(host_only=False)
    with tempfile.TemporaryDirectory() as td:
        env = {"NLS_RCCL_LIB": str(lib), "NLS_RENDEZVOUS_DIR": td, "NLS_SHIM_SLOT_BYTES": str(1 << 20), "NLS_SHIM_TIMEOUT_S": "300", **(extra_env or {})}
        _run_ranks([str(HERE / "_sharded_worker.py"), mode], world, env, timeout)


@pytest.mark.gpu
def test_native_communicator_world2_asynchronous_stand_in():
    """The same fits with the stand-in in its ASYNCHRONOUS mode (``NLS_SHIM_ASYNC=1``: the call returns, a kernel holds the stream, a worker
    thread exchanges - RCCL's shape): every collective wait of the library goes through its polling loop for real."""
    _launch_native("gpu_rccl", 2, {**ASYNC_STAND_IN})


@pytest.mark.gpu
@pytest.mark.parametrize("world", [2, 3, 8])
def test_native_communicator_world_n_one_gpu(world):
    """Row-sharded fits (regression and classification) through the library's own RCCL call sites, world 2, 3 and 8: equal to the
    single-rank fit; ``L_`` only on rank 0; beta identical on every rank; one-stage eigendecomposition (rank-0 ``stedc`` + real
    broadcast + column-split back-transformation + grouped all-gather of unequal blocks)."""
    _launch_native("gpu_rccl", world)


@pytest.mark.gpu
@pytest.mark.parametrize("world", [3, 8])
def test_native_communicator_two_stage_evd(world):
    """The same with the two-stage reduction forced: both back-transformations split by columns, the ranks vote on the
    chase's invariants check (one more all-reduce)."""
    _launch_native("gpu_rccl", world, {"NLS_EVD": "twostage"})


@pytest.mark.gpu
def test_native_communicator_failure_is_an_error_on_every_rank():
    """Symmetric failure injection: the 2nd ``ncclBroadcast`` of every rank (the real eigenvectors of rank 0's tridiagonal solve, after the
    eigenvalues) returns an error: every rank must get NLS_ERR_COMM (``NlsError``) - nobody hangs."""
    _launch_native("gpu_rccl_fail", 2, {"NLS_SHIM_FAIL_BROADCAST": "2"}, timeout=300)


@pytest.mark.gpu
def test_native_communicator_failure_inside_the_group():
    """... and inside the grouped all-gather (broadcasts 3.. of a fit: eigenvalues and real eigenvectors come first): the group is closed,
    the error surfaces, nobody hangs."""
    _launch_native("gpu_rccl_fail", 2, {"NLS_SHIM_FAIL_BROADCAST": "4"}, timeout=300)


@pytest.mark.gpu
def test_native_communicator_failure_world8():
    """World 8, the failure inside the grouped all-gather of eight unequal blocks (the 7th broadcast of every rank)."""
    _launch_native("gpu_rccl_fail", 8, {"NLS_SHIM_FAIL_BROADCAST": "7"}, timeout=600)


# ---- ONE rank fails; the others must not be left waiting (include/neolssvm_hip.h, "Failure of ONE rank ...") -------------
@pytest.mark.gpu
@pytest.mark.parametrize("world, spec", [(2, "gram:1"), (2, "prepare:0"), (2, "cholesky:0:3"), (8, "evd:5"), (8, "backtransform:3"), (8, "sweep:7"),
                                         (8, "select:2")])  # fmt: skip
def test_one_rank_fails_locally_every_rank_returns_at_the_vote(world, spec):
    """A local failure on ONE rank (``NLS_FAULT_INJECT=site:rank[:code]`` - what a failed allocation or launch looks like from inside the
    library) at each stretch of the sharded fit: the rank goes to the next status vote instead of leaving, every rank returns an error
    at once - its own on the failed rank, NLS_ERR_COMM naming rank and code on the others, ``LinAlgError`` everywhere for a factorisation
    failure on rank 0 - and the same communicator carries the next fit."""
    _launch_native("gpu_rccl_fault", world, {"NLS_FAULT_INJECT": spec, "NLS_COMM_TIMEOUT_S": "90", **ASYNC_STAND_IN}, timeout=900)


@pytest.mark.gpu
@pytest.mark.parametrize("world, bad", [(2, 1), (8, 4)])
def test_one_rank_of_the_sigma_grid_fails_every_rank_returns_at_the_vote(world, bad):
    """The gamma x sigma grid shards the SIGMAS; its only exchange is the merge of the small tables on a second, communicator-only
    context.  ONE rank's own fits fail: the status vote on the merge communicator takes every rank out at once (its own error on the
    failed rank, NLS_ERR_COMM naming it on the others), and the next grid on the same communicator names the single-rank winner."""
    _launch_native("gpu_grid_fault", world, {"NLS_TEST_LOST_RANK": str(bad), "NLS_COMM_TIMEOUT_S": "90", **ASYNC_STAND_IN}, timeout=900)


@pytest.mark.gpu
@pytest.mark.parametrize("world, bad, call", [(2, 1, 5), (8, 3, 9)])
def test_one_rank_loses_an_rccl_call_the_others_meet_the_deadline(world, bad, call):
    """The ``call``-th collective of rank ``bad`` ALONE fails inside the communication library (the stand-in's asymmetric injection): that
    rank returns at once; the others sit in a collective whose peer has left and are released by the library's own deadline
    (``NLS_COMM_TIMEOUT_S`` = 10 s here, far below the stand-in's safety net): NLS_ERR_COMM, communicator aborted, further
    collective calls refused, ``nls_comm_destroy`` makes the context a single rank again."""
    env = {"NLS_SHIM_FAIL_RANK": str(bad), "NLS_SHIM_FAIL_CALL": str(call), "NLS_TEST_LOST_RANK": str(bad), "NLS_COMM_TIMEOUT_S": "10", **ASYNC_STAND_IN}
    _launch_native("gpu_rccl_lost_call", world, env, timeout=900)


@pytest.mark.gpu
@pytest.mark.parametrize("world, bad", [(2, 0), (8, 6)])
def test_one_rank_dies_the_others_meet_the_deadline(world, bad):
    """Rank ``bad``'s PROCESS exits (code 7) after joining the communicator and before the fit: the survivors' first status vote never
    completes; the deadline ends it on every survivor."""
    lib = build_shim(host_only=False)
    with tempfile.TemporaryDirectory() as td:
        env = {"NLS_RCCL_LIB": str(lib), "NLS_RENDEZVOUS_DIR": td, "NLS_SHIM_SLOT_BYTES": str(1 << 20), "NLS_SHIM_TIMEOUT_S": "300",
               "NLS_TEST_LOST_RANK": str(bad), "NLS_COMM_TIMEOUT_S": "10", **ASYNC_STAND_IN}  # fmt: skip
        _run_ranks([str(HERE / "_sharded_worker.py"), "gpu_rccl_lost_dead"], world, env, 900, expect_exit={bad: 7})


def _bench_line(out):
    lines = [ln for ln in out.splitlines() if ln.startswith("{")]
    assert len(lines) == 1, out[-3000:]
    return json.loads(lines[0])


@pytest.mark.gpu
@pytest.mark.parametrize("launcher", ["self", "torchrun"])
def test_bench_two_ranks_one_gpu(launcher):
    """``bench.py --gpus 2`` end to end - rendezvous, row shards, the barrier / max-over-ranks timing through the native communicator, rank 0's
    JSON line - with both ranks on GPU 0 and the stand-in communicator: what the driver's scaling run executes, minus the second device.
    Launched by the script itself and by ``python -m torch.distributed.run`` (only its environment is used)."""
    lib = build_shim(host_only=False)
    root = HERE.parent
    with tempfile.TemporaryDirectory() as td:
        env = dict(os.environ, NLS_RCCL_LIB=str(lib), NLS_RENDEZVOUS_DIR=td, NLS_SHIM_SLOT_BYTES=str(1 << 20), NLS_SHIM_TIMEOUT_S="300",
                   NLS_BENCH_DEVICE="0", HSA_ENABLE_IPC_MODE_LEGACY="0", OMP_NUM_THREADS="4", OPENBLAS_NUM_THREADS="4")  # fmt: skip
        for k in ("RANK", "LOCAL_RANK", "WORLD_SIZE", "MASTER_ADDR", "MASTER_PORT"):
            env.pop(k, None)
        args = ["bench.py", "--gpus", "2", "--config", "c2", "--steps", "2", "--warmup", "1", "--no-cpu-baseline", "--no-end-to-end"]
        if launcher == "self":
            cmd = [sys.executable] + args
        else:
            cmd = [sys.executable, "-m", "torch.distributed.run", "--nnodes=1", "--nproc-per-node", "2", "--master-addr", "127.0.0.1",
                   "--master-port", str(_free_port())] + args  # fmt: skip
        p = subprocess.run(cmd, cwd=root, env=env, capture_output=True, text=True, timeout=900)
        assert p.returncode == 0, (p.stdout + p.stderr)[-3000:]
        d = _bench_line(p.stdout)
    assert d["n_gpus"] == 2 and d["steps"] == 2 and d["scaling"] in ("strong", "weak") and d["value"] > 0
    one = subprocess.run([sys.executable] + args[:2] + ["1"] + args[3:], cwd=root, env={k: v for k, v in env.items() if k != "NLS_RCCL_LIB"},
                         capture_output=True, text=True, timeout=900)  # fmt: skip
    assert one.returncode == 0, (one.stdout + one.stderr)[-3000:]
    d1 = _bench_line(one.stdout)
    # the same fit either way: the selected gamma and the LOO score agree (two ranks on ONE device are slower, not different)
    assert d["config"]["workload"] == d1["config"]["workload"]
    assert d["config"]["gamma_index"] == d1["config"]["gamma_index"]
    assert d["config"]["loo_score"] == pytest.approx(d1["config"]["loo_score"], rel=1e-9)
    assert d["config"]["rows_per_gpu"] * 2 == d1["config"]["rows_per_gpu"] and "row-shard x2" in d["config"]["parallelism"]


def _bench_env(td):
    lib = build_shim(host_only=False)
    env = dict(os.environ, NLS_RCCL_LIB=str(lib), NLS_RENDEZVOUS_DIR=td, NLS_SHIM_SLOT_BYTES=str(1 << 20), NLS_SHIM_TIMEOUT_S="300",
               NLS_BENCH_DEVICE="0", HSA_ENABLE_IPC_MODE_LEGACY="0", OMP_NUM_THREADS="2", OPENBLAS_NUM_THREADS="2")  # fmt: skip
    for k in ("RANK", "LOCAL_RANK", "WORLD_SIZE", "MASTER_ADDR", "MASTER_PORT"):
        env.pop(k, None)
    return env


@pytest.mark.gpu
def test_bench_eight_ranks_one_gpu():
    """``bench.py --gpus 8 --config c2`` - the driver's scaling run at its largest world, minus seven devices: eight processes, rendezvous
    through the id file, 12 500 rows per rank, 1025 = 8 * 128 + 1 eigenvector columns over the ranks, the max-over-ranks clock, one N = 8
    line with ``parallelism`` filled; the same fit as one rank computes."""
    root = HERE.parent
    args = ["bench.py", "--gpus", "8", "--config", "c2", "--steps", "2", "--warmup", "1", "--no-cpu-baseline", "--no-end-to-end"]
    with tempfile.TemporaryDirectory() as td:
        env = _bench_env(td)
        p = subprocess.run([sys.executable] + args, cwd=root, env=env, capture_output=True, text=True, timeout=1500)
        assert p.returncode == 0, (p.stdout + p.stderr)[-3000:]
        d = _bench_line(p.stdout)
        one = subprocess.run([sys.executable] + args[:2] + ["1"] + args[3:], cwd=root, env={k: v for k, v in env.items() if k != "NLS_RCCL_LIB"},
                             capture_output=True, text=True, timeout=900)  # fmt: skip
        assert one.returncode == 0, (one.stdout + one.stderr)[-3000:]
        d1 = _bench_line(one.stdout)
    assert d["n_gpus"] == 8 and d["scaling"] == "strong" and d["value"] > 0 and d["metric"] == d1["metric"]
    assert d["config"]["rows_per_gpu"] == 12_500 and "row-shard x8" in d["config"]["parallelism"]
    assert d["config"]["gamma_index"] == d1["config"]["gamma_index"]
    assert d["config"]["loo_score"] == pytest.approx(d1["config"]["loo_score"], rel=1e-9)


@pytest.mark.gpu
def test_bench_sigma_grid_eight_ranks_one_gpu():
    """Config 5's deployment at world 8 in miniature (``--config c5s``: 16 sigma x 32 gamma, n = 2e4): every rank fits its two sigmas on
    all rows through ``nls_primal_fit_grid``, the tables are merged through the NATIVE communicator of a second context
    (``grid->merge``), every rank names the same winner as the single-rank grid."""
    root = HERE.parent
    args = ["bench.py", "--gpus", "8", "--config", "c5s", "--steps", "1", "--warmup", "1", "--no-cpu-baseline", "--no-end-to-end"]
    with tempfile.TemporaryDirectory() as td:
        env = _bench_env(td)
        p = subprocess.run([sys.executable] + args, cwd=root, env=env, capture_output=True, text=True, timeout=1500)
        assert p.returncode == 0, (p.stdout + p.stderr)[-3000:]
        d = _bench_line(p.stdout)
        one = subprocess.run([sys.executable] + args[:2] + ["1"] + args[3:], cwd=root, env={k: v for k, v in env.items() if k != "NLS_RCCL_LIB"},
                             capture_output=True, text=True, timeout=900)  # fmt: skip
        assert one.returncode == 0, (one.stdout + one.stderr)[-3000:]
        d1 = _bench_line(one.stdout)
    assert d["n_gpus"] == 8 and d["scaling"] == "weak" and "sigma-shard x8" in d["config"]["parallelism"]
    assert (d["config"]["sigma_index"], d["config"]["gamma_index"]) == (d1["config"]["sigma_index"], d1["config"]["gamma_index"])
