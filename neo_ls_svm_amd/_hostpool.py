"""Host buffers for the LARGE outputs of a fit (the Cholesky factor ``L_``: 268 MB at D = 4096, 800 MB at n = 10^4 in the dual path).

A fresh ``np.zeros`` of that size costs the process ~2 10^5 first-touch page faults when the download arrives and a ``munmap`` of as many
pages when the previous result is dropped - 35-40 ms per dual fit at n = 10^4, more than the factorisation itself.  Buffers of at least
``MIN_BYTES`` therefore come from a small pool of anonymous mappings: the array handed out keeps a *lease* object alive as its base; when the
last array (or view) on it is garbage collected the mapping goes back to the pool - still mapped, pages resident - for the next fit.
Smaller outputs are plain ``np.zeros``.

Only the triangle the library defines is meaningful in such an array (scipy's ``cho_factor`` contract: the other triangle "contains random
data"); a recycled buffer holds a previous factor's entries there, a fresh one zeros.  ``release()`` unmaps what the pool holds
(``Context.release_workspace`` calls it).

The arrays start 64 bytes into their mapping on purpose: into a PAGE-ALIGNED pageable destination the HIP runtime locks the user pages in place
for every copy (measured: 12 ms per 268 MB when the pages are resident, 38 ms when they are fresh), into an unaligned one it stages the copy
(0.4 ms visible once the pages are resident) - numpy's own arrays are never page-aligned either.

A mapping can additionally be page-locked ONCE (``nls_host_register``; 16-40 ms per 268 MB, 110-140 ms per 800 MB): the library then sends the
finished block columns with asynchronous copies.  With the unaligned start this buys nothing measurable any more (0.36 against 0.43 ms of visible
download at D = 4096), so the default policy is ``False``; ``pin_large_outputs("reuse")`` page-locks a mapping when it is handed out for the
second time, ``pin_large_outputs(True)`` at creation, and ``reserve(shape, dtype, ctx)`` puts page-locked mappings into the pool ahead of a loop
(``bench.py``).  A failed registration (locked-memory limit) simply leaves the buffer pageable.
"""

from __future__ import annotations

import ctypes
import mmap
import os
import threading

import numpy as np

MIN_BYTES = 64 << 20
MAX_POOLED_PER_SIZE = 2
MAX_POOLED_BYTES = 4 << 30
# The array starts OFFSET bytes into its mapping: a page-aligned pageable destination makes the HIP runtime lock the user pages in place for every copy
# (12 ms per 268 MB when the pages are resident, 38 ms when they are fresh - measured), an unaligned one takes its staged path (2-3 ms), as numpy's own
# arrays do.
OFFSET = 64
PIN_OUTPUTS = False  # False | "reuse" | True, see the module docstring
LAST_REGISTER_RC = None  # return code of the most recent nls_host_register (diagnostic)

_lock = threading.RLock()  # re-entrant: a cyclic-GC pass triggered by an allocation inside a locked region may finalise another lease on this thread
_free: dict[int, list[mmap.mmap]] = {}
_registered: dict[int, object] = {}  # id(mapping) -> the ctypes library that page-locked it


def _address(mm: mmap.mmap) -> int:
    view = ctypes.c_char.from_buffer(mm)
    try:
        return ctypes.addressof(view) + OFFSET
    finally:
        del view


def _close(mm: mmap.mmap) -> None:
    lib = _registered.pop(id(mm), None)
    if lib is not None:
        try:
            lib.nls_host_unregister(None, ctypes.c_void_p(_address(mm)))
        except Exception:
            pass
    mm.close()


def _pooled_bytes() -> int:
    return sum(size * len(v) for size, v in _free.items())


class _Lease:
    """Owner of one mapping while arrays on it are alive (``ndarray.base``)."""

    __slots__ = ("_mm", "_view", "__array_interface__")

    def __init__(self, mm: mmap.mmap, shape, dtype):
        self._mm = mm
        self._view = ctypes.c_char.from_buffer(mm)  # (pins the mapping: it cannot be resized or closed under the arrays)
        self.__array_interface__ = {"data": (ctypes.addressof(self._view) + OFFSET, False), "shape": tuple(shape), "typestr": np.dtype(dtype).str, "version": 3}

    def __del__(self):
        mm, self._view = self._mm, None
        if mm is None:
            return
        try:
            with _lock:
                size = len(mm) - 4096
                lst = _free.setdefault(size, [])
                if len(lst) < MAX_POOLED_PER_SIZE and _pooled_bytes() + size <= MAX_POOLED_BYTES:
                    lst.append(mm)
                    return
            _close(mm)
        except Exception:  # interpreter shutdown: the module globals may be gone; the mapping dies with the process
            pass


def pin_large_outputs(flag=True) -> None:
    """Policy for page-locking the pooled buffers of large factor outputs: ``True`` (at creation), ``"reuse"`` (when a buffer is handed out for
    the second time) or ``False`` (never: the default); see the module docstring."""
    global PIN_OUTPUTS
    PIN_OUTPUTS = "reuse" if flag == "reuse" else bool(flag)


def factor_output(shape, dtype, ctx=None) -> np.ndarray:
    """A writable C-contiguous array for a factor output: pooled mapping when it is large (page-locked through ``ctx`` after ``pin_large_outputs(True)``),
    ``np.zeros`` otherwise."""
    nbytes = int(np.prod(shape)) * np.dtype(dtype).itemsize
    if nbytes < MIN_BYTES or os.environ.get("NLS_HOST_POOL", "1") == "0":  # (NLS_HOST_POOL=0: always a fresh array)
        return np.zeros(shape, dtype=dtype)
    with _lock:
        lst = _free.get(nbytes)
        mm = lst.pop() if lst else None
    recycled = mm is not None
    if mm is None:
        mm = mmap.mmap(-1, nbytes + 4096)  # anonymous, zero-filled on first touch (+ a page: the array starts OFFSET bytes in)
        try:  # transparent huge pages where the system allows them on request (what numpy asks for its own large arrays): 512 x fewer faults / TLB entries
            mm.madvise(mmap.MADV_HUGEPAGE)
        except (AttributeError, OSError, ValueError):
            pass
    if (PIN_OUTPUTS is True or (PIN_OUTPUTS == "reuse" and recycled)) and ctx is not None and id(mm) not in _registered and getattr(ctx, "handle", None):
        global LAST_REGISTER_RC
        try:
            LAST_REGISTER_RC = ctx.lib.nls_host_register(ctx.handle, ctypes.c_void_p(_address(mm)), ctypes.c_size_t(nbytes))
            if LAST_REGISTER_RC == 0:
                _registered[id(mm)] = ctx.lib
        except Exception as exc:  # pragma: no cover
            LAST_REGISTER_RC = repr(exc)
    return np.asarray(_Lease(mm, shape, dtype))


def reserve(shape, dtype, ctx=None, count: int = 2) -> int:
    """Put ``count`` mappings for outputs of this shape into the pool ahead of time - page-locked when a context is given - so that a loop of
    fits starts in its steady state (what a C caller does when it allocates and registers its output buffer once).  Returns how many were added."""
    nbytes = int(np.prod(shape)) * np.dtype(dtype).itemsize
    if nbytes < MIN_BYTES or os.environ.get("NLS_HOST_POOL", "1") == "0":
        return 0
    added = 0
    for _ in range(count):
        with _lock:
            if len(_free.get(nbytes, [])) >= MAX_POOLED_PER_SIZE or _pooled_bytes() + nbytes > MAX_POOLED_BYTES:
                break
        mm = mmap.mmap(-1, nbytes + 4096)
        try:
            mm.madvise(mmap.MADV_HUGEPAGE)
        except (AttributeError, OSError, ValueError):
            pass
        if ctx is not None and getattr(ctx, "handle", None):
            try:
                if ctx.lib.nls_host_register(ctx.handle, ctypes.c_void_p(_address(mm)), ctypes.c_size_t(nbytes)) == 0:
                    _registered[id(mm)] = ctx.lib
            except Exception:
                pass
        with _lock:
            _free.setdefault(nbytes, []).append(mm)
        added += 1
    return added


def release() -> None:
    """Unmap every pooled buffer (arrays still alive keep theirs)."""
    with _lock:
        held = [mm for lst in _free.values() for mm in lst]
        _free.clear()
    for mm in held:
        _close(mm)
