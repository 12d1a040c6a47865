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
    """The 2nd ``ncclBroadcast`` of every rank fails (stand-in's injection): the group call returns NLS_ERR_COMM naming the rank."""
    _run("fail", 3, {"NLS_SHIM_FAIL_BROADCAST": "2"}, timeout=300)
