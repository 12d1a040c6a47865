"""world_size-2 gloo tests of the row-sharded path (CPU), and the same with the real library on one GPU."""

from __future__ import annotations

import os
import socket
import subprocess
import sys
from pathlib import Path

import pytest

HERE = Path(__file__).resolve().parent


def _free_port():
    with socket.socket() as s:
        s.bind(("127.0.0.1", 0))
        return s.getsockname()[1]


def _launch(mode, world=2, timeout=600, extra_env=None):
    port = _free_port()
    procs = []
    for rank in range(world):
        env = dict(os.environ, **(extra_env or {}), RANK=str(rank), WORLD_SIZE=str(world), MASTER_ADDR="127.0.0.1", MASTER_PORT=str(port),
                   OMP_NUM_THREADS="2", OPENBLAS_NUM_THREADS="2", HSA_ENABLE_IPC_MODE_LEGACY="0")  # fmt: skip
        procs.append(subprocess.Popen([sys.executable, str(HERE / "_sharded_worker.py"), mode], env=env,
                                      stdout=subprocess.PIPE, stderr=subprocess.STDOUT, text=True))  # fmt: skip
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
        assert p.returncode == 0 and f"OK {rank}" in out, f"rank {rank} failed:\n{out[-3000:]}"


def test_row_shard_tiles_the_rows():
    from neo_ls_svm_amd.distributed import row_shard

    for n in (1, 7, 1000, 1_000_000):
        for world in (1, 2, 3, 8):
            blocks = [row_shard(n, r, world) for r in range(world)]
            assert blocks[0][0] == 0 and blocks[-1][1] == n
            assert all(a[1] == b[0] for a, b in zip(blocks, blocks[1:]))
            sizes = [hi - lo for lo, hi in blocks]
            assert max(sizes) - min(sizes) <= 1
    with pytest.raises(ValueError):
        row_shard(10, 2, 2)


def test_sharded_protocol_world2_gloo_cpu():
    """Row shards + the three all-reduce points reproduce the single-process result (oracle stages, gloo)."""
    _launch("cpu")


@pytest.mark.gpu
def test_sharded_library_world2_one_gpu():
    """The library's real sharded path: two ranks share GPU 0, collectives staged over gloo."""
    _launch("gpu")


@pytest.mark.gpu
def test_sharded_library_world2_one_gpu_two_stage_evd():
    """The same with the eigendecomposition forced through the two-stage reduction: the tridiagonal eigenvectors are computed on rank 0
    and broadcast as REAL numbers, each rank back-transforms (both stages) its own column block, the blocks are all-gathered."""
    _launch("gpu", extra_env={"NLS_EVD": "twostage"})
