"""Row sharding over one process per GPU.

The primal path shards by rows: every rank calls ``primal_fit`` on its own contiguous block
of X, y, s and the library all-reduces, through the hook registered here, exactly four things:

    1. {sum s, sum s*y, n}                 -> global weight normalisation and c = 1 / (n_total (D+1))
    2. the tile-packed Hermitian block A||b -> identical normal equations (hence EVD, beta, L) on every rank
    3. the eigenvector matrix Q             -> each rank back-transforms its own column block of the eigenvectors and
                                               contributes zeros elsewhere: an exact all-gather (bit-identical Q)
    4. the per-gamma error vectors          -> identical argmin on every rank
    (+ the two scalars of the LOO score)

Per-row outputs (loo_residuals, loo_leverage, loo_std, residuals) stay sharded.  The collective itself is
``torch.distributed`` - backend ``nccl`` is RCCL over xGMI on the GPU box (zero-copy on the library's device
buffer), ``gloo`` is staged through host memory (CPU tests, or several ranks sharing one GPU).

IMPORTANT: import ``torch`` BEFORE the HIP library is loaded (i.e. before the first ``Context``), so that
both use one copy of the ROCm runtime; ``attach`` checks this.
"""

from __future__ import annotations

import ctypes
import sys

import numpy as np

__all__ = ["row_shard", "make_allreduce", "attach"]


def row_shard(n: int, rank: int, world: int) -> tuple[int, int]:
    """Contiguous, balanced row block [lo, hi) of rank ``rank``; blocks tile [0, n) in rank order."""
    if not (0 <= rank < world):
        raise ValueError(f"rank {rank} outside world {world}")
    return (n * rank) // world, (n * (rank + 1)) // world


class _DeviceView:
    """Zero-copy view of ``count`` doubles at a raw device address for ``torch.as_tensor``."""

    def __init__(self, ptr: int, count: int):
        self.__cuda_array_interface__ = {"shape": (count,), "typestr": "<f8", "data": (ptr, False), "version": 2}


def make_allreduce(ctx, dist=None, group=None):
    """Return ``fn(device_ptr, count)`` that sums ``count`` doubles over the ranks of ``group`` in place."""
    import torch

    dist = dist or torch.distributed
    backend = dist.get_backend(group)

    if backend == "nccl":
        device = torch.device("cuda", ctx.device)

        def fn(ptr: int, count: int) -> None:
            t = torch.as_tensor(_DeviceView(ptr, count), device=device)
            dist.all_reduce(t, op=dist.ReduceOp.SUM, group=group)
            torch.cuda.synchronize(device)

        return fn

    def fn_staged(ptr: int, count: int) -> None:
        host = np.empty(count, dtype=np.float64)
        ctx._check(ctx.lib.nls_memcpy_d2h(ctx.handle, host.ctypes.data, ctypes.c_void_p(ptr), host.nbytes))
        t = torch.from_numpy(host)
        dist.all_reduce(t, op=dist.ReduceOp.SUM, group=group)
        ctx._check(ctx.lib.nls_memcpy_h2d(ctx.handle, ctypes.c_void_p(ptr), host.ctypes.data, host.nbytes))

    return fn_staged


def attach(ctx, dist=None, group=None) -> tuple[int, int]:
    """Register the collective hook on ``ctx`` for the initialised process group; returns (rank, world)."""
    if "torch" not in sys.modules:
        raise RuntimeError("import torch (and init the process group) before creating the neo_ls_svm_amd Context")
    import torch

    dist = dist or torch.distributed
    rank, world = dist.get_rank(group), dist.get_world_size(group)
    ctx.set_allreduce(make_allreduce(ctx, dist, group) if world > 1 else None, rank, world)
    return rank, world
