"""CPU self-test worker of the test stand-in for librccl (tests/csrc/rccl_shim.cpp built with -DSHIM_HOST_ONLY): N processes
drive its entry points through ctypes on host buffers.  argv: <shim.so> <id file> ; RANK / WORLD_SIZE from the environment."""

from __future__ import annotations

import ctypes as C
import os
import sys
import time

import numpy as np


class UniqueId(C.Structure):
    _fields_ = [("internal", C.c_char * 128)]


def main():
    lib = C.CDLL(sys.argv[1])
    idfile = sys.argv[2]
    rank, world = int(os.environ["RANK"]), int(os.environ["WORLD_SIZE"])
    lib.ncclCommInitRank.argtypes = [C.POINTER(C.c_void_p), C.c_int, UniqueId, C.c_int]
    lib.ncclAllReduce.argtypes = [C.c_void_p, C.c_void_p, C.c_size_t, C.c_int, C.c_int, C.c_void_p, C.c_void_p]
    lib.ncclBroadcast.argtypes = [C.c_void_p, C.c_void_p, C.c_size_t, C.c_int, C.c_int, C.c_void_p, C.c_void_p]
    lib.ncclCommDestroy.argtypes = [C.c_void_p]
    lib.ncclGetErrorString.restype = C.c_char_p
    uid = UniqueId()
    if rank == 0:
        assert lib.ncclGetUniqueId(C.byref(uid)) == 0
        with open(idfile + ".tmp", "wb") as f:
            f.write(bytes(uid))
        os.replace(idfile + ".tmp", idfile)
    else:
        t0 = time.monotonic()
        while not os.path.exists(idfile):
            assert time.monotonic() - t0 < 60
            time.sleep(0.01)
        C.memmove(C.byref(uid), open(idfile, "rb").read(), 128)
    comm = C.c_void_p()
    assert lib.ncclCommInitRank(C.byref(comm), world, uid, rank) == 0
    DOUBLE, SUM, MAX = 8, 0, 2
    if len(sys.argv) > 3 and sys.argv[3] == "asymmetric":  # NLS_SHIM_FAIL_RANK / NLS_SHIM_FAIL_CALL: one rank's call fails, the others time out
        lib.ncclCommGetAsyncError.argtypes = [C.c_void_p, C.POINTER(C.c_int)]
        lib.ncclCommAbort.argtypes = [C.c_void_p]
        bad, call = int(os.environ["NLS_SHIM_FAIL_RANK"]), int(os.environ["NLS_SHIM_FAIL_CALL"])
        a = np.ones(16)
        for k in range(1, call + 1):
            t0 = time.monotonic()
            rc = lib.ncclAllReduce(a.ctypes.data, a.ctypes.data, a.size, DOUBLE, SUM, comm, None)
            if k < call:
                assert rc == 0 and np.all(a == world**k)
            elif rank == bad:
                assert rc != 0 and b"injected" in lib.ncclGetErrorString(rc) and time.monotonic() - t0 < 1.0
            else:
                assert rc != 0 and b"time-out" in lib.ncclGetErrorString(rc) and time.monotonic() - t0 > 2.0
        assert lib.ncclCommAbort(comm) == 0
        print(f"OK {rank}", flush=True)
        return
    fail_at = int(os.environ.get("NLS_SHIM_FAIL_BROADCAST", "0"))

    # all-reduce (sum) larger than a slot (NLS_SHIM_SLOT_BYTES is small in the test): chunking; in place
    n = 5000
    a = np.arange(n, dtype=np.float64) * (rank + 1)
    assert lib.ncclAllReduce(a.ctypes.data, a.ctypes.data, n, DOUBLE, SUM, comm, None) == 0
    assert np.array_equal(a, np.arange(n) * sum(range(1, world + 1)))
    # max, out of place
    b, out = np.array([float(rank), -float(rank), 3.5]), np.zeros(3)
    assert lib.ncclAllReduce(b.ctypes.data, out.ctypes.data, 3, DOUBLE, MAX, comm, None) == 0
    assert np.array_equal(out, [world - 1, 0.0, 3.5])
    nb = 0
    # broadcast from a non-zero root, in place
    root = world - 1
    v = np.full(3000, float(rank))
    rc = lib.ncclBroadcast(v.ctypes.data, v.ctypes.data, v.size, DOUBLE, root, comm, None)
    nb += 1
    if fail_at == nb:
        assert rc != 0 and b"injected" in lib.ncclGetErrorString(rc)
        print(f"OK {rank}", flush=True)
        lib.ncclCommDestroy(comm)
        return
    assert rc == 0 and np.all(v == root)
    # grouped broadcasts of unequal blocks (the library's all-gather): rank r owns [offs[r], offs[r + 1])
    offs = [0]
    for r in range(world):
        offs.append(offs[-1] + 700 * (r + 1) + 13)
    g = np.zeros(offs[-1])
    g[offs[rank] : offs[rank + 1]] = rank + 1
    assert lib.ncclGroupStart() == 0
    bad = 0
    for r in range(world):
        p = g.ctypes.data + 8 * offs[r]
        rc = lib.ncclBroadcast(p, p, offs[r + 1] - offs[r], DOUBLE, r, comm, None)
        nb += 1
        if rc != 0:
            bad = rc
            break
    end = lib.ncclGroupEnd()
    if fail_at and fail_at <= nb:
        assert bad != 0 and end == 0
        print(f"OK {rank}", flush=True)
        lib.ncclCommDestroy(comm)
        return
    assert bad == 0 and end == 0
    for r in range(world):
        assert np.all(g[offs[r] : offs[r + 1]] == r + 1), r
    lib.ncclCommDestroy(comm)
    print(f"OK {rank}", flush=True)


if __name__ == "__main__":
    main()
