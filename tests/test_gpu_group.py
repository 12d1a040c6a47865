"""Several GPUs behind one handle and one call (``nls_group_*``; ``hotpath.primal_fit(ctx=Group)``; ``NeoLSSVM(devices=[...])``) and the
gamma x sigma grid behind the C ABI (``nls_primal_fit_grid``) - SURVEY.md 8(b) "multi-GPU is internal to the ctx", 8(e), 8(d) config 5.

The pool's boxes have ONE GPU and RCCL refuses two ranks per device, so the group's members all sit on GPU 0 and the communicator is the
test stand-in for librccl (``tests/csrc/rccl_shim.cpp``, found through ``NLS_RCCL_LIB``): the ranks are host threads of one process here, as in
a real group.  Everything else - ``nls_comm_init_rank`` per member, the all-reduce / broadcast / grouped-broadcast call sites, the row blocks,
the column split of the back-transformation - is the code an 8-GPU node runs.  Each case runs in a subprocess: a process loads one
communication library."""

from __future__ import annotations

import os
import subprocess
import sys
from pathlib import Path

import pytest

HERE = Path(__file__).resolve().parent
pytestmark = pytest.mark.gpu
ASYNC_STAND_IN = {"NLS_SHIM_ASYNC": "1", "GPU_MAX_HW_QUEUES": "32"}  # (see tests/test_rccl_shim.py: one process, many streams, one device)


def _run(mode, world, extra_env=None, timeout=1200):
    from test_rccl_shim import build_shim

    lib = build_shim(host_only=False)
    env = dict(os.environ, NLS_RCCL_LIB=str(lib), NLS_SHIM_SLOT_BYTES=str(4 << 20), NLS_SHIM_TIMEOUT_S="300", HSA_ENABLE_IPC_MODE_LEGACY="0",
               OMP_NUM_THREADS="4", OPENBLAS_NUM_THREADS="4", **(extra_env or {}))  # fmt: skip
    p = subprocess.run([sys.executable, str(HERE / "_group_worker.py"), mode, str(world)], env=env, capture_output=True, text=True, timeout=timeout)
    assert p.returncode == 0 and "OK" in p.stdout, (p.stdout + p.stderr)[-4000:]


@pytest.mark.parametrize("world", [1, 2, 3, 8])
def test_group_fit_and_predict_equal_one_device(world):
    """``primal_fit`` / ``primal_predict`` on a Group of ``world`` members == the same calls on one context: every output incl. ``L_``, row
    outputs concatenated in row order, ragged row blocks, fewer query rows than ranks; n < world is a ValueError before any collective."""
    _run("fit", world)


@pytest.mark.parametrize("world", [3, 8])
def test_estimator_on_several_devices_reproduces_the_fixtures(world):
    """``NeoLSSVM(devices=[...])``: the reference's surface (``_neo_ls_svm.py:327-442``) reaching N GPUs in one ``fit`` - the fixtures the
    single-device estimator is pinned to, and its single-device twin to 1e-9 incl. ``L_``, ``loo_*``, ``predict_std``, quantiles."""
    _run("estimator", world)


@pytest.mark.parametrize("world", [1, 3, 8])
def test_sigma_grid_behind_the_c_abi(world):
    """``nls_primal_fit_grid`` == the Python reference driver bit for bit (tables, indices, finished count, the winner's full result);
    ``nls_group_primal_fit_grid`` (sigmas dealt over the members) == one context; the exact-tie rule; the group row-shards again afterwards."""
    _run("grid", world)


@pytest.mark.parametrize("evd", ["onestage", "twostage"])
def test_group_world8_at_4097_columns(evd):
    """World 8 at D = 4096: the 4097 eigenvector columns split 8 ways (8 * 512 + 1: uneven blocks), rank-0 tridiagonal solve + broadcast,
    grouped all-gather - one- and two-stage eigendecomposition."""
    _run("big", 8, {"NLS_EVD": evd}, timeout=2400)


def test_group_failure_is_an_error_not_a_hang():
    """The 2nd ``ncclBroadcast`` of every member fails (the stand-in's symmetric injection): the group call returns NLS_ERR_COMM naming a rank."""
    _run("fail", 3, {"NLS_SHIM_FAIL_BROADCAST": "2"}, timeout=300)


@pytest.mark.parametrize("world, spec", [(2, "gram:1"), (3, "cholesky:0:3"), (8, "prepare:4"), (8, "evd:0"), (8, "backtransform:6"), (8, "sweep:1"), (8, "select:7")])
def test_one_member_fails_locally_the_group_call_returns(world, spec):
    """One member of the group fails on its own (``NLS_FAULT_INJECT=site:rank[:code]``) at each stretch of the sharded fit: no member is
    left in a collective and the one call returns the failing member's error (``LinAlgError`` for a factorisation failure) - at the next
    status vote, not at the deadline - and the next call on the same group, the same communicator, is right."""
    _run("fault", world, {"NLS_FAULT_INJECT": spec, "NLS_COMM_TIMEOUT_S": "90", **ASYNC_STAND_IN}, timeout=900)


@pytest.mark.parametrize("world, bad, call", [(2, 1, 4), (8, 5, 8), (8, 0, 3)])
def test_one_member_loses_an_rccl_call_the_abort_flag_releases_the_others(world, bad, call):
    """The ``call``-th collective of ONE member fails inside the communication library; that member's thread returns, the others are
    inside a collective it will never join.  The group's abort flag (polled by their bounded waits) releases them at once - with the
    deadline at 90 s the call must still return within seconds - and the group joins a fresh communicator for the next call."""
    _run("lost", world, {"NLS_SHIM_FAIL_RANK": str(bad), "NLS_SHIM_FAIL_CALL": str(call), "NLS_COMM_TIMEOUT_S": "90", **ASYNC_STAND_IN}, timeout=900)
