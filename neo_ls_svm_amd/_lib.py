"""ctypes binding of ``libneolssvm_hip.so`` (C ABI: ``include/neolssvm_hip.h``).

There is no CPU fallback: importing this module without the built library, or creating a context
without an MI355X, raises.  Build with ``python -c "import __graft_entry__ as g; g.build()"`` or
``make -C neo_ls_svm_amd/csrc``.
"""

from __future__ import annotations

import ctypes as C
import os
from pathlib import Path

import numpy as np

_HERE = Path(__file__).resolve().parent
LIB_PATH = Path(os.environ.get("NEOLSSVM_HIP_LIB", _HERE / "libneolssvm_hip.so"))

NLS_OK, NLS_ERR_ARG, NLS_ERR_HIP, NLS_ERR_LINALG, NLS_ERR_COMM = 0, 1, 2, 3, 4
ABI_VERSION = 4
FIT_SWEEP_ONLY, FIT_FINISH_IF_BELOW, FIT_RESIDUALS_FROM_SWEEP = 1, 2, 4
COMM_ID_BYTES = 128
NUM_TIMINGS = 24
TIMING_NAMES = {
    "total": 0,
    "upload": 1,
    "featuremap": 2,
    "gram": 3,
    "allreduce": 4,
    "evd": 5,
    "rotate": 6,
    "sweep": 7,
    "loo": 8,
    "cholesky": 9,
    "residuals": 10,
    "download": 11,
    "rotate_launches": 12,
    "gram_launches": 13,
    "sweep_launches": 14,
    "featuremap_launches": 15,
    "rotate_flops": 16,
    "gram_flops": 17,
    "sweep_flops": 18,
    "featuremap_flops": 19,
    "row_chunk": 20,
}

ALLREDUCE_FN = C.CFUNCTYPE(C.c_int, C.c_void_p, C.c_size_t, C.c_void_p)

_dp = C.POINTER(C.c_double)


class PrimalFitArgs(C.Structure):
    """Mirror of ``nls_primal_fit_args`` (field order is the ABI)."""

    _fields_ = [
        ("X", C.c_void_p),
        ("y", C.c_void_p),
        ("s", C.c_void_p),
        ("shift", C.c_void_p),
        ("scale", C.c_void_p),
        ("B", C.c_void_p),
        ("gammas", C.c_void_p),
        ("n", C.c_int64),
        ("d", C.c_int32),
        ("D", C.c_int32),
        ("G", C.c_int32),
        ("is_classifier", C.c_int32),
        ("gamma_index_in", C.c_int32),
        ("flags", C.c_int32),
        ("Cmat", C.c_void_p),
        ("finish_below", C.c_double),
        ("beta", C.c_void_p),
        ("L", C.c_void_p),
        ("lam", C.c_void_p),
        ("loo_errors", C.c_void_p),
        ("objective", C.c_void_p),
        ("loo_residuals", C.c_void_p),
        ("loo_leverage", C.c_void_p),
        ("loo_std", C.c_void_p),
        ("residuals", C.c_void_p),
        ("loo_score", C.c_void_p),
        ("gamma_index", C.c_void_p),
        ("finished", C.c_void_p),
        ("timings", C.c_void_p),
    ]


class SigmaGrid(C.Structure):
    """Mirror of ``nls_sigma_grid``."""

    _fields_ = [
        ("sigmas", C.c_void_p),
        ("Sg", C.c_int32),
        ("rank", C.c_int32),
        ("world", C.c_int32),
        ("merge", C.c_void_p),
        ("loo_errors", C.c_void_p),
        ("objective", C.c_void_p),
        ("seconds", C.c_void_p),
        ("sigma_index", C.c_void_p),
        ("gamma_index", C.c_void_p),
        ("best_valid", C.c_void_p),
        ("finished_count", C.c_void_p),
        ("timings", C.c_void_p),
    ]


class DualFitArgs(C.Structure):
    """Mirror of ``nls_dual_fit_args``."""

    _fields_ = [
        ("Xt", C.c_void_p),
        ("y", C.c_void_p),
        ("s", C.c_void_p),
        ("gammas", C.c_void_p),
        ("n", C.c_int64),
        ("r", C.c_int32),
        ("G", C.c_int32),
        ("is_classifier", C.c_int32),
        ("gamma_index_in", C.c_int32),
        ("alpha", C.c_void_p),
        ("L", C.c_void_p),
        ("lam", C.c_void_p),
        ("loo_errors", C.c_void_p),
        ("objective", C.c_void_p),
        ("loo_residuals", C.c_void_p),
        ("loo_std", C.c_void_p),
        ("residuals", C.c_void_p),
        ("loo_score", C.c_void_p),
        ("gamma_index", C.c_void_p),
        ("timings", C.c_void_p),
    ]


# name -> (restype, argtypes); must list every symbol include/neolssvm_hip.h declares.
SIGNATURES = {
    "nls_abi_version": (C.c_int, []),
    "nls_ctx_create": (C.c_int, [C.c_int, C.POINTER(C.c_void_p)]),
    "nls_ctx_destroy": (None, [C.c_void_p]),
    "nls_last_error": (C.c_char_p, [C.c_void_p]),
    "nls_set_allreduce": (C.c_int, [C.c_void_p, ALLREDUCE_FN, C.c_void_p, C.c_int, C.c_int]),
    "nls_set_workspace_limit": (C.c_int, [C.c_void_p, C.c_size_t]),
    "nls_ws_release": (C.c_int, [C.c_void_p, C.c_size_t, C.POINTER(C.c_size_t)]),
    "nls_comm_get_unique_id": (C.c_int, [C.c_void_p]),
    "nls_comm_init_rank": (C.c_int, [C.c_void_p, C.c_void_p, C.c_int, C.c_int]),
    "nls_comm_destroy": (C.c_int, [C.c_void_p]),
    "nls_comm_allreduce": (C.c_int, [C.c_void_p, C.c_void_p, C.c_size_t, C.c_int]),
    "nls_comm_set_virtual_rank": (C.c_int, [C.c_void_p, C.c_int, C.c_int, C.c_int]),
    "nls_comm_set_timeout": (C.c_int, [C.c_void_p, C.c_double]),
    "nls_comm_abort": (C.c_int, [C.c_void_p]),
    "nls_comm_state": (C.c_int, [C.c_void_p]),
    "nls_factor_create": (C.c_int, [C.c_void_p, C.c_void_p, C.c_int, C.POINTER(C.c_void_p)]),
    "nls_factor_destroy": (C.c_int, [C.c_void_p, C.c_void_p]),
    "nls_device_malloc": (C.c_int, [C.c_void_p, C.c_size_t, C.POINTER(C.c_void_p)]),
    "nls_device_free": (C.c_int, [C.c_void_p, C.c_void_p]),
    "nls_memcpy_h2d": (C.c_int, [C.c_void_p, C.c_void_p, C.c_void_p, C.c_size_t]),
    "nls_memcpy_d2h": (C.c_int, [C.c_void_p, C.c_void_p, C.c_void_p, C.c_size_t]),
    "nls_synchronize": (C.c_int, [C.c_void_p]),
    "nls_device_info": (C.c_int, [C.c_void_p, C.c_char_p, C.c_int, C.POINTER(C.c_int), C.POINTER(C.c_size_t)]),
    "nls_featuremap": (
        C.c_int,
        [C.c_void_p, C.c_void_p, C.c_int64, C.c_int, C.c_void_p, C.c_void_p, C.c_void_p, C.c_int, C.c_void_p],
    ),
    "nls_gram_only": (
        C.c_int,
        [C.c_void_p, C.c_void_p, C.c_void_p, C.c_void_p, C.c_int64, C.c_int, C.c_void_p, C.c_void_p, C.c_void_p, C.c_int]
        + [C.c_void_p, C.c_void_p],
    ),
    "nls_rotate_only": (
        C.c_int,
        [C.c_void_p, C.c_void_p, C.c_int64, C.c_int, C.c_void_p, C.c_void_p, C.c_void_p, C.c_int, C.c_void_p, C.c_void_p]
        + [C.c_void_p, C.c_void_p],
    ),
    "nls_tridiag_only": (C.c_int, [C.c_void_p, C.c_void_p, C.c_int, C.c_int, C.c_void_p, C.c_void_p, C.c_void_p]),
    "nls_eigh_only": (C.c_int, [C.c_void_p, C.c_void_p, C.c_int, C.c_int, C.c_void_p]),
    "nls_twostage_stage": (C.c_int, [C.c_void_p, C.c_int, C.c_void_p, C.c_int, C.c_int, C.c_int, C.c_void_p, C.c_void_p, C.c_void_p, C.c_int, C.c_void_p]),
    "nls_host_register": (C.c_int, [C.c_void_p, C.c_void_p, C.c_size_t]),
    "nls_host_unregister": (C.c_int, [C.c_void_p, C.c_void_p]),
    "nls_cholesky_only": (C.c_int, [C.c_void_p, C.c_void_p, C.c_int, C.c_void_p]),
    "nls_zcholesky_only": (C.c_int, [C.c_void_p, C.c_void_p, C.c_int, C.c_void_p]),
    "nls_twostage_fallbacks": (C.c_long, [C.c_void_p]),
    "nls_twostage_rescues": (C.c_long, [C.c_void_p]),
    "nls_evd_stage_ms": (C.c_int, [C.c_void_p, C.c_void_p]),
    "nls_stedc_only": (C.c_int, [C.c_void_p, C.c_void_p, C.c_void_p, C.c_int, C.c_void_p]),
    "nls_primal_fit": (C.c_int, [C.c_void_p, C.POINTER(PrimalFitArgs)]),
    "nls_primal_fit_grid": (C.c_int, [C.c_void_p, C.POINTER(PrimalFitArgs), C.POINTER(SigmaGrid)]),
    "nls_grid_select": (C.c_int, [C.c_void_p, C.c_void_p, C.c_int, C.c_int, C.c_int, C.POINTER(C.c_int32), C.POINTER(C.c_int32)]),
    "nls_grid_visiting_order": (C.c_int, [C.c_void_p, C.c_int, C.c_int, C.c_int, C.c_void_p]),
    "nls_group_create": (C.c_int, [C.POINTER(C.c_int), C.c_int, C.POINTER(C.c_void_p)]),
    "nls_group_destroy": (None, [C.c_void_p]),
    "nls_group_last_error": (C.c_char_p, [C.c_void_p]),
    "nls_group_size": (C.c_int, [C.c_void_p]),
    "nls_group_ctx": (C.c_void_p, [C.c_void_p, C.c_int]),
    "nls_group_primal_fit": (C.c_int, [C.c_void_p, C.POINTER(PrimalFitArgs)]),
    "nls_group_primal_fit_grid": (C.c_int, [C.c_void_p, C.POINTER(PrimalFitArgs), C.POINTER(SigmaGrid)]),
    "nls_group_factor_create": (C.c_int, [C.c_void_p, C.c_void_p, C.c_int, C.POINTER(C.c_void_p)]),
    "nls_group_factor_destroy": (C.c_int, [C.c_void_p, C.c_void_p]),
    "nls_group_primal_predict": (
        C.c_int,
        [C.c_void_p, C.c_void_p, C.c_int64, C.c_int, C.c_void_p, C.c_void_p, C.c_void_p, C.c_int, C.c_void_p, C.c_void_p]
        + [C.c_void_p, C.c_void_p, C.c_void_p],
    ),
    "nls_sweep_weights": (C.c_int, [C.c_void_p, C.c_int, C.c_void_p, C.c_void_p, C.POINTER(C.c_int)]),
    "nls_primal_predict": (
        C.c_int,
        [C.c_void_p, C.c_void_p, C.c_int64, C.c_int, C.c_void_p, C.c_void_p, C.c_void_p, C.c_int, C.c_void_p, C.c_void_p]
        + [C.c_void_p, C.c_void_p, C.c_void_p],
    ),
    "nls_bin_stats": (
        C.c_int,
        [C.c_void_p, C.c_void_p, C.c_void_p, C.c_int64, C.c_int, C.c_void_p, C.c_void_p, C.c_int, C.c_void_p, C.c_void_p],
    ),
    "nls_bin_stats_labels": (
        C.c_int,
        [C.c_void_p, C.c_void_p, C.c_void_p, C.c_int64, C.c_int, C.c_void_p, C.c_int, C.c_void_p, C.c_void_p],
    ),
    "nls_rank_codes": (C.c_int, [C.c_void_p, C.c_void_p, C.c_int64, C.c_void_p, C.POINTER(C.c_int64)]),
    "nls_dual_fit": (C.c_int, [C.c_void_p, C.POINTER(DualFitArgs)]),
    "nls_dual_predict": (
        C.c_int,
        [C.c_void_p, C.c_void_p, C.c_int64, C.c_void_p, C.c_int64, C.c_int, C.c_void_p, C.c_void_p, C.c_void_p, C.c_void_p],
    ),
}

_lib = None


def load_library() -> C.CDLL:
    """Load the HIP library once; raise loudly when it is missing (no fallback path exists)."""
    global _lib
    if _lib is not None:
        return _lib
    if not LIB_PATH.exists():
        raise ImportError(
            f"{LIB_PATH} not found: the MI355X HIP library is not built. Run "
            "`python -c 'import __graft_entry__ as g; g.build()'` (or `make -C neo_ls_svm_amd/csrc`). "
            "neo_ls_svm_amd has no CPU fallback."
        )
    # (ROCBLAS_USE_HIPBLASLT=1 - rocBLAS's own, process-wide switch: the real GEMMs of the two-stage eigendecomposition's first back-transformation
    # run at 37.5 instead of 43 ms at n = 1e4 - is the LAUNCHER's to export: loading this library does not change the host process's environment,
    # INTEGRATION.md section 5)
    lib = C.CDLL(str(LIB_PATH))
    for name, (restype, argtypes) in SIGNATURES.items():
        fn = getattr(lib, name)  # AttributeError if the export is missing
        fn.restype = restype
        fn.argtypes = argtypes
    if lib.nls_abi_version() != ABI_VERSION:
        raise ImportError(f"{LIB_PATH} has ABI version {lib.nls_abi_version()}, this package needs {ABI_VERSION}: rebuild the library")
    _lib = lib
    return lib


class NlsError(RuntimeError):
    pass


def raise_for(code: int, message: str):
    """Map C error codes to the exceptions the reference raises."""
    if code == NLS_OK:
        return
    if code == NLS_ERR_ARG:
        raise ValueError(message)
    if code == NLS_ERR_LINALG:
        raise np.linalg.LinAlgError(message)
    raise NlsError(message)


class DeviceArray:
    """A float64 array resident in HBM, owned by a Context (plain pointer + shape, no torch)."""

    def __init__(self, ctx: "Context", shape):
        self.ctx = ctx
        self.shape = tuple(int(x) for x in np.atleast_1d(shape))
        self.nbytes = int(np.prod(self.shape)) * 8
        p = C.c_void_p()
        ctx._check(ctx.lib.nls_device_malloc(ctx.handle, max(self.nbytes, 8), C.byref(p)))
        self.ptr = p.value

    # Lets an array library (cupy, ...) wrap the buffer without a copy.
    @property
    def __cuda_array_interface__(self):
        return {"shape": self.shape, "typestr": "<f8", "data": (self.ptr, False), "version": 2, "strides": None}

    def copy_from_host(self, a: np.ndarray) -> "DeviceArray":
        a = np.ascontiguousarray(a, dtype=np.float64)
        assert a.nbytes == self.nbytes
        self.ctx._check(self.ctx.lib.nls_memcpy_h2d(self.ctx.handle, self.ptr, a.ctypes.data, a.nbytes))
        return self

    def to_host(self) -> np.ndarray:
        out = np.empty(self.shape, dtype=np.float64)
        self.ctx._check(self.ctx.lib.nls_memcpy_d2h(self.ctx.handle, out.ctypes.data, self.ptr, out.nbytes))
        return out

    def free(self):
        if self.ptr:
            self.ctx.lib.nls_device_free(self.ctx.handle, self.ptr)
            self.ptr = None

    def __del__(self):
        try:
            self.free()
        except Exception:
            pass


def _ptr(a):
    """void* of a numpy array / DeviceArray / None."""
    if a is None:
        return None
    if isinstance(a, DeviceArray):
        return a.ptr
    return a.ctypes.data


class _StdoutToStderr:
    """librccl prints a version banner on file descriptor 1 when the first communicator is created (and leaves it in the C
    library's stdout buffer when that is a pipe); a driver that parses this process's stdout - bench.py's one JSON line -
    must not see it.  While active, fd 1 points at stderr; on exit the C buffers are flushed before fd 1 is restored."""

    def __enter__(self):
        import sys

        sys.stdout.flush()
        self.saved = os.dup(1)
        os.dup2(2, 1)
        return self

    def __exit__(self, *exc):
        try:
            C.CDLL(None).fflush(None)
        except Exception:
            pass
        os.dup2(self.saved, 1)
        os.close(self.saved)
        return False


class Factor:
    """Device-resident inverse of one fitted Cholesky factor ``L_`` (``nls_factor_create``): what repeated
    ``predict_std`` calls reuse instead of uploading and inverting (D+1)^2 complex numbers every time.  Explicit state:
    it belongs to the estimator that fitted ``L`` and is dropped (``close``) on refit; nothing is keyed on addresses."""

    def __init__(self, ctx: "Context", L: np.ndarray):
        L = np.ascontiguousarray(L, dtype=np.complex128)
        if L.ndim != 2 or L.shape[0] != L.shape[1] or L.shape[0] < 2:
            raise ValueError("L must be a (D+1) x (D+1) complex matrix")
        self.ctx, self.D = ctx, L.shape[0] - 1
        h = C.c_void_p()
        ctx._check(ctx.lib.nls_factor_create(ctx.handle, L.ctypes.data, self.D, C.byref(h)))
        self.handle = h

    def close(self):
        if getattr(self, "handle", None) and getattr(self.ctx, "handle", None):
            self.ctx.lib.nls_factor_destroy(self.ctx.handle, self.handle)
        self.handle = None

    def __del__(self):
        try:
            self.close()
        except Exception:
            pass


class Context:
    """One context per process and GPU: stream, rocBLAS/rocSOLVER handles, grow-only workspace."""

    def __init__(self, device: int = 0, _borrowed=None):
        self.lib = load_library()
        self._owned = _borrowed is None
        if _borrowed is None:
            h = C.c_void_p()
            rc = self.lib.nls_ctx_create(int(device), C.byref(h))
            if rc != NLS_OK:
                msg = self.lib.nls_last_error(None)
                raise NlsError(f"nls_ctx_create failed: {msg.decode() if msg else rc}")
        else:  # a member context of a Group (nls_group_ctx): the group owns and destroys it
            h = C.c_void_p(_borrowed)
        self.handle = h
        self.device = int(device)
        self.comm_world = 1  # world size of the native communicator this context has joined (1: none, or one rank)
        self._hook = None  # keep the ctypes callback alive

    def _check(self, rc: int):
        if rc != NLS_OK:
            msg = self.lib.nls_last_error(self.handle)
            raise_for(rc, msg.decode() if msg else f"error {rc}")

    def close(self):
        if getattr(self, "handle", None):
            dev, self._hold_cache = getattr(self, "_hold_cache", None), None
            if dev is not None:
                dev.free()
            if self._owned:
                self.lib.nls_ctx_destroy(self.handle)
            self.handle = None

    def __del__(self):
        try:
            self.close()
        except Exception:
            pass

    def synchronize(self):
        self._check(self.lib.nls_synchronize(self.handle))

    def release_workspace(self, min_bytes: int = 0) -> int:
        """Free the workspace buffers of at least ``min_bytes`` each (0: all); returns the bytes still held.  The arena
        only grows between calls (a c3-size fit leaves ~100 GB allocated for the next one); this is the trim."""
        held = C.c_size_t()
        self._check(self.lib.nls_ws_release(self.handle, int(min_bytes), C.byref(held)))
        from . import _hostpool  # the pooled host buffers of large factor outputs go with it

        _hostpool.release()
        dev, self._hold_cache = getattr(self, "_hold_cache", None), None
        if dev is not None:
            dev.free()
        return int(held.value)

    # ---- native RCCL communicator (one process per GPU; no torch) -----------------------------------
    def comm_unique_id(self) -> bytes:
        """Rank 0: a fresh communicator id (128 bytes) to hand to the other ranks (file, socket, env ...)."""
        buf = C.create_string_buffer(COMM_ID_BYTES)
        with _StdoutToStderr():
            rc = self.lib.nls_comm_get_unique_id(buf)
        if rc != NLS_OK:
            msg = self.lib.nls_last_error(None)
            raise NlsError(f"nls_comm_get_unique_id failed: {msg.decode() if msg else rc}")
        return bytes(buf.raw)

    def comm_init(self, unique_id: bytes, rank: int, world: int):
        """Join the communicator (collective over all ranks); from now on ``primal_fit`` treats its rows as this
        rank's block of a row-sharded problem."""
        if len(unique_id) != COMM_ID_BYTES:
            raise ValueError(f"unique_id must be {COMM_ID_BYTES} bytes")
        with _StdoutToStderr():
            rc = self.lib.nls_comm_init_rank(self.handle, C.c_char_p(unique_id), int(rank), int(world))
        self._check(rc)
        self.comm_world = int(world)

    def comm_destroy(self):
        self._check(self.lib.nls_comm_destroy(self.handle))
        self.comm_world = 1

    def comm_allreduce(self, values, op: str = "sum") -> np.ndarray:
        """Sum / max of a few host doubles over the ranks (driver utility: barrier, timing)."""
        v = np.ascontiguousarray(np.atleast_1d(values), dtype=np.float64).copy()
        self._check(self.lib.nls_comm_allreduce(self.handle, v.ctypes.data, v.size, {"sum": 0, "max": 1}[op]))
        return v

    def comm_barrier(self):
        self.comm_allreduce([0.0])

    def comm_set_virtual_rank(self, rank: int = 0, world: int = 0, capture: bool = False):
        """Measurement hook (``nls_comm_set_virtual_rank``; ``bench.py --as-rank r --of W``): this context, the only rank of a native
        communicator, does rank ``rank``'s share of a ``world``-rank sharded fit; ``capture=True`` (with world <= 1) makes the next complete
        fit leave the eigenpairs the virtual ranks take their peers' blocks from."""
        self._check(self.lib.nls_comm_set_virtual_rank(self.handle, int(rank), int(world), 1 if capture else 0))

    def comm_set_timeout(self, seconds: float):
        """Deadline of every collective wait (``nls_comm_set_timeout``): after it the communicator is aborted and the call raises
        ``NlsError`` (NLS_ERR_COMM) instead of waiting for a rank that died or left.  0: ``NLS_COMM_TIMEOUT_S`` / 300 s."""
        self._check(self.lib.nls_comm_set_timeout(self.handle, float(seconds)))

    def comm_abort(self):
        """Give the communicator up now (``ncclCommAbort``); collective calls fail until ``comm_init`` / ``comm_destroy``."""
        self._check(self.lib.nls_comm_abort(self.handle))

    @property
    def comm_state(self) -> str:
        return {0: "none", 1: "joined", 2: "aborted"}.get(int(self.lib.nls_comm_state(self.handle)), "?")

    _EVD_KINDS = {1: "one-stage real", 2: "one-stage complex", 3: "two-stage real", 4: "two-stage complex", 5: "rocsolver heevd / syevd"}

    def evd_stage_ms(self) -> dict | None:
        """Stage times (ms) of this context's most recent eigendecomposition (``nls_evd_stage_ms``), or None when none has run."""
        out = np.zeros(8)
        if self.lib.nls_evd_stage_ms(self.handle, out.ctypes.data) != NLS_OK:
            return None
        kind = int(out[7])
        two = kind in (3, 4)
        d = {"kind": self._EVD_KINDS.get(kind, str(kind)), "n": int(out[6])}
        if kind == 5:
            d["total"] = round(float(out[5]), 3)
            return d
        d["band" if two else "tridiagonalisation"] = round(float(out[0]), 3)
        if two:
            d["chase"] = round(float(out[1]), 3)
        d["stedc"] = round(float(out[2]), 3)
        if two:
            d["q2"] = round(float(out[3]), 3)
        d["q1" if two else "back_transformation"] = round(float(out[4]), 3)
        d["total"] = round(float(out[5]), 3)
        return d

    def device_info(self) -> dict:
        name = C.create_string_buffer(256)
        cus, hbm = C.c_int(), C.c_size_t()
        self._check(self.lib.nls_device_info(self.handle, name, 256, C.byref(cus), C.byref(hbm)))
        return {"name": name.value.decode(), "compute_units": cus.value, "hbm_bytes": hbm.value}

    def to_device(self, a: np.ndarray) -> DeviceArray:
        a = np.ascontiguousarray(a, dtype=np.float64)
        return DeviceArray(self, a.shape).copy_from_host(a)

    def empty(self, shape) -> DeviceArray:
        return DeviceArray(self, shape)

    # ---- one upload per fit ------------------------------------------------------------------------
    def hold(self, a: np.ndarray):
        """Context manager: upload the host matrix ``a`` once and let every hot-path call made inside the ``with``
        block that is handed this very array (same object, or a view of the same memory, shape and strides) use the
        device copy - the estimator's ``fit`` passes X to the pre-step statistics and to the solver."""
        ctx = self

        class _Hold:
            def __enter__(self_inner):
                arr = np.asarray(a)
                if arr.dtype == np.float64 and arr.ndim == 2 and arr.flags.c_contiguous and arr.size:
                    # the device buffer of the previous fit's X is kept and reused when the size matches: a hipMalloc + hipFree of the
                    # matrix per fit cost 15-55 ms (51 MB ... 1 GB), more than the upload itself; Context.release_workspace drops it
                    dev = getattr(ctx, "_hold_cache", None)
                    ctx._hold_cache = None
                    if dev is not None and dev.ptr and dev.nbytes == arr.nbytes:
                        dev.shape = tuple(arr.shape)
                        dev.copy_from_host(arr)
                    else:
                        if dev is not None:
                            dev.free()
                        dev = ctx.to_device(arr)
                    ctx._held = (arr.__array_interface__["data"][0], arr.shape, arr.strides, dev)
                return self_inner

            def __exit__(self_inner, *exc):
                h, ctx._held = getattr(ctx, "_held", None), None
                if h is not None:
                    ctx._hold_cache = h[3]
                return False

        return _Hold()

    def held(self, a):
        """The device copy made by ``hold`` if ``a`` is that host matrix, else ``a`` itself."""
        h = getattr(self, "_held", None)
        if h is not None and isinstance(a, np.ndarray) and a.dtype == np.float64:
            if (a.__array_interface__["data"][0], a.shape, a.strides) == h[:3]:
                return h[3]
        return a

    def set_allreduce(self, fn, rank: int, world: int):
        """fn(device_ptr: int, count: int) -> None sums `count` doubles at `device_ptr` over all ranks."""
        if fn is None or world <= 1:
            self._hook = None
            self._check(self.lib.nls_set_allreduce(self.handle, ALLREDUCE_FN(0), None, 0, 1))
            return

        def _thunk(buf, count, _user):
            try:
                fn(int(buf), int(count))
                return 0
            except Exception as exc:  # surfaced as NLS_ERR_COMM
                import traceback

                traceback.print_exc()
                self._hook_error = exc
                return 1

        self._hook = ALLREDUCE_FN(_thunk)
        self._check(self.lib.nls_set_allreduce(self.handle, self._hook, None, int(rank), int(world)))


class GroupFactor:
    """``Factor`` for a group: U^-1 of one fitted ``L_`` resident on every member device (``nls_group_factor_create``)."""

    def __init__(self, group: "Group", L: np.ndarray):
        L = np.ascontiguousarray(L, dtype=np.complex128)
        if L.ndim != 2 or L.shape[0] != L.shape[1] or L.shape[0] < 2:
            raise ValueError("L must be a (D+1) x (D+1) complex matrix")
        self.ctx, self.D = group, L.shape[0] - 1
        h = C.c_void_p()
        group._check(group.lib.nls_group_factor_create(group.handle, L.ctypes.data, self.D, C.byref(h)))
        self.handle = h

    def close(self):
        if getattr(self, "handle", None) and getattr(self.ctx, "handle", None):
            self.ctx.lib.nls_group_factor_destroy(self.ctx.handle, self.handle)
        self.handle = None

    def __del__(self):
        try:
            self.close()
        except Exception:
            pass


class Group:
    """Several GPUs behind one handle (``nls_group_create``; SURVEY.md 8(b): "multi-GPU is internal to the ctx ... the Python surface is
    identical at 1 and 8 GPUs").  One member context per listed device, joined by an RCCL communicator inside the library; ``hotpath.primal_fit``
    / ``primal_predict`` / ``primal_fit_sigma_grid`` take a Group wherever they take a Context, and ``NeoLSSVM(devices=[...])`` builds one.
    A group call blocks the calling thread (ctypes releases the GIL) while the library drives one host thread per device."""

    def __init__(self, devices):
        self.lib = load_library()
        self.devices = tuple(int(d) for d in devices)
        if not self.devices:
            raise ValueError("devices must name at least one GPU")
        arr = (C.c_int * len(self.devices))(*self.devices)
        h = C.c_void_p()
        with _StdoutToStderr():  # (librccl's version banner, as in Context.comm_init)
            rc = self.lib.nls_group_create(arr, len(self.devices), C.byref(h))
        if rc != NLS_OK:
            msg = self.lib.nls_group_last_error(None)
            raise_for(rc, f"nls_group_create failed: {msg.decode() if msg else rc}")
        self.handle = h
        self.device = self.devices[0]
        self.contexts = [Context(d, _borrowed=self.lib.nls_group_ctx(h, r)) for r, d in enumerate(self.devices)]
        for c in self.contexts:
            c.comm_world = len(self.devices)

    @property
    def size(self) -> int:
        return len(self.devices)

    def _check(self, rc: int):
        if rc != NLS_OK:
            msg = self.lib.nls_group_last_error(self.handle)
            raise_for(rc, msg.decode() if msg else f"error {rc}")

    # the estimator's single-device plumbing (pre-step statistics, the dual path: "replicas only") runs on rank 0's context
    def hold(self, a):
        return self.contexts[0].hold(a)

    def held(self, a):
        return a  # a group fit takes host rows: every rank uploads its own block

    def evd_stage_ms(self):
        return self.contexts[0].evd_stage_ms()

    def synchronize(self):
        for c in self.contexts:
            c.synchronize()

    def release_workspace(self, min_bytes: int = 0) -> int:
        return sum(c.release_workspace(min_bytes) for c in self.contexts)

    def close(self):
        if getattr(self, "handle", None):
            for c in self.contexts:
                c.close()  # (borrowed handles: drops the cached device copies only)
            self.lib.nls_group_destroy(self.handle)
            self.handle = None

    def __del__(self):
        try:
            self.close()
        except Exception:
            pass


_default_ctx: dict[int, Context] = {}
_default_group: dict[tuple, Group] = {}


def default_group(devices) -> Group:
    """The process-wide group of this device tuple (created on first use, like ``default_context``)."""
    key = tuple(int(d) for d in devices)
    if key not in _default_group:
        _default_group[key] = Group(key)
    return _default_group[key]


def set_default_context(ctx: Context) -> None:
    """Make ``ctx`` the context estimators of its device use (``NeoLSSVM(device=...)``): a program that already owns a context
    hands it over instead of letting the estimator create a second one on the same GPU."""
    _default_ctx[ctx.device] = ctx


def default_context(device: int = 0) -> Context:
    if device not in _default_ctx:
        _default_ctx[device] = Context(device)
    return _default_ctx[device]
