// TEST INFRASTRUCTURE: a stand-in `librccl.so.1` for N ranks - processes, or host threads of one process (a group) - that share ONE GPU.
//
// RCCL refuses two ranks on one device, and the pool's boxes have one GPU each, so the library's NATIVE collective
// path (ctx->comm != NULL: ncclAllReduce / ncclBroadcast / grouped broadcasts, csrc/nls_host.h, csrc/nls_evd.hip)
// would otherwise run for the first time on an 8-GPU node.  This shim implements exactly the entry points
// csrc/nls_comm.hip resolves, with RCCL's semantics as far as the library relies on them:
//   * collectives are ordered on the stream they are given (here: the stream is drained, the data is staged through
//     a POSIX shared-memory segment, every rank reduces all slots in rank order - identical bits on every rank);
//   * calls between ncclGroupStart / ncclGroupEnd are deferred to ncclGroupEnd;
//   * in-place and out-of-place buffers, ncclDouble with ncclSum / ncclMax, arbitrary roots and unequal counts per call.
// Found by the library through NLS_RCCL_LIB (or LD_LIBRARY_PATH, which precedes the RUNPATH of libneolssvm_hip.so).
// Failure injection: NLS_SHIM_FAIL_BROADCAST=k makes the k-th ncclBroadcast call of every rank return
// ncclInternalError (all ranks issue the same sequence, so they fail at the same point and nobody is left waiting).
// A barrier that is not completed within NLS_SHIM_TIMEOUT_S (default 120) returns ncclSystemError instead of hanging.
//
// Build: hipcc -shared -fPIC -O2 rccl_shim.cpp -o _shim/librccl.so.1   (tests/test_rccl_shim.py does it on demand)
#include <fcntl.h>
#include <hip/hip_runtime.h>
#include <rccl/rccl.h>
#include <sched.h>
#include <sys/mman.h>
#include <sys/stat.h>
#include <unistd.h>

#include <atomic>
#include <chrono>
#include <cstdio>
#include <cstdlib>
#include <cstring>
#include <functional>
#include <vector>

// -DSHIM_HOST_ONLY: the buffers are host memory (the CPU self-test of this file, tests/test_rccl_shim.py::test_shim_protocol_cpu)
#ifdef SHIM_HOST_ONLY
#define SHIM_COPY(dst, src, bytes, kind) (std::memcpy((dst), (src), (bytes)), hipSuccess)
#define SHIM_SYNC(stream) hipSuccess
#else
#define SHIM_COPY(dst, src, bytes, kind) hipMemcpy((dst), (src), (bytes), (kind))
#define SHIM_SYNC(stream) hipStreamSynchronize(stream)
#endif

namespace {

constexpr size_t HEADER_BYTES = 4096;
constexpr int MAX_RANKS = 16;

struct Header {
  std::atomic<int> arrived;
  std::atomic<int> generation;
  std::atomic<int> attached;
  std::atomic<int> poisoned;  // a rank gave up on a barrier: everybody fails fast from now on
  size_t slot_bytes;
};

struct Comm {
  int rank = 0, world = 1;
  char name[64] = {0};
  size_t map_bytes = 0;
  char* base = nullptr;
  Header* hdr = nullptr;
  size_t slot_bytes = 0;
  std::vector<char> host;  // staging for this rank's contribution / the result
  int broadcast_calls = 0;  // per communicator rank (ranks may be threads of one process: a group, include/neolssvm_hip.h)
  char* slot(int r) const { return base + HEADER_BYTES + (size_t)r * slot_bytes; }
};

thread_local int g_group_depth = 0;
thread_local std::vector<std::function<ncclResult_t()>> g_deferred;

double now_s() { return std::chrono::duration<double>(std::chrono::steady_clock::now().time_since_epoch()).count(); }
double timeout_s() {
  const char* e = std::getenv("NLS_SHIM_TIMEOUT_S");
  return e ? std::atof(e) : 120.0;
}
size_t slot_bytes_default() {
  const char* e = std::getenv("NLS_SHIM_SLOT_BYTES");
  return e ? (size_t)std::atoll(e) : ((size_t)8 << 20);
}

ncclResult_t barrier(Comm* c) {
  Header* h = c->hdr;
  if (h->poisoned.load()) return ncclSystemError;
  const int gen = h->generation.load();
  if (h->arrived.fetch_add(1) + 1 == c->world) {
    h->arrived.store(0);
    h->generation.fetch_add(1);
    return ncclSuccess;
  }
  const double t0 = now_s();
  while (h->generation.load() == gen) {
    if (h->poisoned.load()) return ncclSystemError;
    if (now_s() - t0 > timeout_s()) {
      h->poisoned.store(1);
      return ncclSystemError;
    }
    sched_yield();
  }
  return ncclSuccess;
}

size_t dtype_bytes(ncclDataType_t t) { return t == ncclDouble ? 8 : 0; }

ncclResult_t run_allreduce(const void* send, void* recv, size_t count, ncclRedOp_t op, Comm* c, hipStream_t stream) {
  if (SHIM_SYNC(stream) != hipSuccess) return ncclUnhandledCudaError;
  const size_t per = c->slot_bytes / 8;
  std::vector<double> acc(std::min(count, per));
  for (size_t off = 0; off < count; off += per) {
    const size_t m = std::min(per, count - off);
    if (SHIM_COPY(c->slot(c->rank), static_cast<const double*>(send) + off, m * 8, hipMemcpyDeviceToHost) != hipSuccess) return ncclUnhandledCudaError;
    ncclResult_t r = barrier(c);
    if (r != ncclSuccess) return r;
    for (size_t i = 0; i < m; ++i) {  // rank order: the same bits on every rank
      double v = reinterpret_cast<const double*>(c->slot(0))[i];
      for (int k = 1; k < c->world; ++k) {
        const double w = reinterpret_cast<const double*>(c->slot(k))[i];
        v = op == ncclMax ? (w > v ? w : v) : v + w;
      }
      acc[i] = v;
    }
    r = barrier(c);  // everybody has read the slots
    if (r != ncclSuccess) return r;
    if (SHIM_COPY(static_cast<double*>(recv) + off, acc.data(), m * 8, hipMemcpyHostToDevice) != hipSuccess) return ncclUnhandledCudaError;
  }
  return ncclSuccess;
}

ncclResult_t run_broadcast(const void* send, void* recv, size_t count, int root, Comm* c, hipStream_t stream) {
  if (SHIM_SYNC(stream) != hipSuccess) return ncclUnhandledCudaError;
  const size_t per = c->slot_bytes / 8;
  for (size_t off = 0; off < count; off += per) {
    const size_t m = std::min(per, count - off);
    if (c->rank == root &&
        SHIM_COPY(c->slot(root), static_cast<const double*>(send) + off, m * 8, hipMemcpyDeviceToHost) != hipSuccess)
      return ncclUnhandledCudaError;
    ncclResult_t r = barrier(c);
    if (r != ncclSuccess) return r;
    if (c->rank != root || recv != send) {
      if (SHIM_COPY(static_cast<double*>(recv) + off, c->slot(root), m * 8, hipMemcpyHostToDevice) != hipSuccess) return ncclUnhandledCudaError;
    }
    r = barrier(c);
    if (r != ncclSuccess) return r;
  }
  return ncclSuccess;
}

ncclResult_t submit(std::function<ncclResult_t()> fn) {
  if (g_group_depth > 0) {
    g_deferred.push_back(std::move(fn));
    return ncclSuccess;
  }
  return fn();
}

}  // namespace

extern "C" {

ncclResult_t ncclGetUniqueId(ncclUniqueId* id) {
  if (!id) return ncclInvalidArgument;
  static std::atomic<int> counter{0};
  std::memset(id->internal, 0, NCCL_UNIQUE_ID_BYTES);
  std::snprintf(id->internal, 64, "/nls_rccl_shim_%d_%d_%ld", (int)getpid(), counter.fetch_add(1), (long)(now_s() * 1e3) % 1000000007L);
  const size_t bytes = HEADER_BYTES + MAX_RANKS * slot_bytes_default();
  const int fd = shm_open(id->internal, O_CREAT | O_EXCL | O_RDWR, 0600);
  if (fd < 0) return ncclSystemError;
  if (ftruncate(fd, (off_t)bytes) != 0) {
    close(fd);
    return ncclSystemError;
  }
  void* p = mmap(nullptr, HEADER_BYTES, PROT_READ | PROT_WRITE, MAP_SHARED, fd, 0);
  close(fd);
  if (p == MAP_FAILED) return ncclSystemError;
  Header* h = new (p) Header();
  h->arrived.store(0);
  h->generation.store(0);
  h->attached.store(0);
  h->poisoned.store(0);
  h->slot_bytes = slot_bytes_default();
  munmap(p, HEADER_BYTES);
  std::fprintf(stderr, "[rccl shim] test stand-in for librccl: communicator id %s\n", id->internal);
  return ncclSuccess;
}

ncclResult_t ncclCommInitRank(ncclComm_t* comm, int nranks, ncclUniqueId id, int rank) {
  if (!comm || nranks < 1 || nranks > MAX_RANKS || rank < 0 || rank >= nranks) return ncclInvalidArgument;
  Comm* c = new Comm();
  c->rank = rank;
  c->world = nranks;
  std::memcpy(c->name, id.internal, 63);
  const int fd = shm_open(c->name, O_RDWR, 0600);
  if (fd < 0) {
    delete c;
    return ncclSystemError;
  }
  struct stat st;
  fstat(fd, &st);
  c->map_bytes = (size_t)st.st_size;
  c->base = static_cast<char*>(mmap(nullptr, c->map_bytes, PROT_READ | PROT_WRITE, MAP_SHARED, fd, 0));
  close(fd);
  if (c->base == MAP_FAILED) {
    delete c;
    return ncclSystemError;
  }
  c->hdr = reinterpret_cast<Header*>(c->base);
  c->slot_bytes = c->hdr->slot_bytes;
  c->hdr->attached.fetch_add(1);
  const ncclResult_t r = barrier(c);  // like RCCL: returns once every rank has joined
  if (r != ncclSuccess) return r;
  if (rank == 0) shm_unlink(c->name);  // everybody has it mapped: the name can go
  *comm = reinterpret_cast<ncclComm_t>(c);
  return ncclSuccess;
}

ncclResult_t ncclCommDestroy(ncclComm_t comm) {
  Comm* c = reinterpret_cast<Comm*>(comm);
  if (!c) return ncclSuccess;
  if (c->base) munmap(c->base, c->map_bytes);
  delete c;
  return ncclSuccess;
}

ncclResult_t ncclAllReduce(const void* sendbuff, void* recvbuff, size_t count, ncclDataType_t datatype, ncclRedOp_t op, ncclComm_t comm,
                           hipStream_t stream) {
  Comm* c = reinterpret_cast<Comm*>(comm);
  if (!c || !sendbuff || !recvbuff || dtype_bytes(datatype) == 0 || (op != ncclSum && op != ncclMax)) return ncclInvalidArgument;
  return submit([=] { return run_allreduce(sendbuff, recvbuff, count, op, c, stream); });
}

ncclResult_t ncclBroadcast(const void* sendbuff, void* recvbuff, size_t count, ncclDataType_t datatype, int root, ncclComm_t comm,
                           hipStream_t stream) {
  Comm* c = reinterpret_cast<Comm*>(comm);
  if (!c || !recvbuff || dtype_bytes(datatype) == 0 || root < 0 || root >= c->world) return ncclInvalidArgument;
  ++c->broadcast_calls;
  if (const char* e = std::getenv("NLS_SHIM_FAIL_BROADCAST"))
    if (std::atoi(e) == c->broadcast_calls) return ncclInternalError;
  return submit([=] { return run_broadcast(sendbuff, recvbuff, count, root, c, stream); });
}

ncclResult_t ncclGroupStart() {
  ++g_group_depth;
  return ncclSuccess;
}

ncclResult_t ncclGroupEnd() {
  if (g_group_depth <= 0) return ncclInvalidUsage;
  if (--g_group_depth > 0) return ncclSuccess;
  ncclResult_t first = ncclSuccess;
  for (auto& fn : g_deferred) {
    const ncclResult_t r = fn();
    if (r != ncclSuccess && first == ncclSuccess) first = r;
  }
  g_deferred.clear();
  return first;
}

const char* ncclGetErrorString(ncclResult_t r) {
  switch (r) {
    case ncclSuccess: return "no error";
    case ncclUnhandledCudaError: return "unhandled HIP error (shim)";
    case ncclSystemError: return "system error / barrier time-out (shim)";
    case ncclInternalError: return "internal error (shim: injected)";
    case ncclInvalidArgument: return "invalid argument (shim)";
    case ncclInvalidUsage: return "invalid usage (shim)";
    default: return "unknown result code (shim)";
  }
}

}  // extern "C"
