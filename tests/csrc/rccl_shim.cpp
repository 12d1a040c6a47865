// TEST INFRASTRUCTURE: a stand-in `librccl.so.1` for N ranks - processes, or host threads of one process (a group) - that share ONE GPU.
//
// RCCL refuses two ranks on one device, and the pool's boxes have one GPU each, so the library's NATIVE collective
// path (ctx->comm != NULL: ncclAllReduce / ncclBroadcast / grouped broadcasts, csrc/nls_host.h, csrc/nls_evd.hip)
// would otherwise run for the first time on an 8-GPU node.  This shim implements exactly the entry points
// csrc/nls_comm.hip resolves, with RCCL's semantics as far as the library relies on them:
//   * collectives are ordered on the stream they are given; the data is staged through a POSIX shared-memory segment and every rank
//     reduces all slots in rank order (identical bits on every rank).  Two modes:
//       - default (synchronous): the call drains the stream and does the exchange before it returns.  Nothing of the shim is ever pending
//         on the device, so the functional tests - N ranks on ONE GPU - cannot dead-lock on a device-wide synchronisation (a hipFree of a
//         growing workspace, a hipDeviceSynchronize of the caller) that would wait for a peer's pending collective, which waits for us;
//         RCCL itself never meets that: it refuses two ranks per device;
//       - NLS_SHIM_ASYNC=1 (the failure-handling tests): ASYNCHRONOUS like RCCL - the call returns at once, the caller's stream is held by
//         a small kernel that spins on a host flag (RCCL's kernels spin on their peers the same way) and a worker thread of the
//         communicator does the exchange once the stream has reached the call, then releases the stream.  A rank whose peer never
//         arrives then sits in `hipStreamQuery == hipErrorNotReady`, exactly what the library's deadline (comm_wait) has to deal with.
//         Run it with GPU_MAX_HW_QUEUES=32: a held stream holds its hardware queue, and by default the streams of one process share 4;
//   * calls between ncclGroupStart / ncclGroupEnd are deferred to ncclGroupEnd;
//   * in-place and out-of-place buffers, ncclDouble with ncclSum / ncclMax, arbitrary roots and unequal counts per call;
//   * ncclCommAbort ends this rank's pending collectives (the stream is released) WITHOUT telling the peers - as with RCCL, they find out
//     through their own deadline; ncclCommGetAsyncError reports a failed exchange.
// Found by the library through NLS_RCCL_LIB (or LD_LIBRARY_PATH, which precedes the RUNPATH of libneolssvm_hip.so).
// Failure injection:
//   NLS_SHIM_FAIL_BROADCAST=k             the k-th ncclBroadcast call of EVERY rank returns ncclInternalError (symmetric);
//   NLS_SHIM_FAIL_RANK=r NLS_SHIM_FAIL_CALL=k   the k-th collective call (all-reduce or broadcast) of rank r ALONE returns ncclInternalError:
//                                         the other ranks are left in the collective (asymmetric - the case the library's deadline and the
//                                         group's abort flag exist for).
// Safety nets of the shim itself: a barrier that is not completed within NLS_SHIM_TIMEOUT_S (default 120) gives up (ncclSystemError through
// ncclCommGetAsyncError) and the spin kernel leaves after the same time, so a test can never leave the GPU busy for good.
//
// Build: hipcc -shared -fPIC -O2 rccl_shim.cpp -o _shim/librccl.so.1   (tests/test_rccl_shim.py does it on demand)
// -DSHIM_HOST_ONLY: the buffers are host memory and the calls run synchronously (the CPU self-test of the protocol, tests/test_rccl_shim.py::test_shim_protocol_cpu)
#include <fcntl.h>
#include <hip/hip_runtime.h>
#include <rccl/rccl.h>
#include <sched.h>
#include <sys/mman.h>
#include <sys/stat.h>
#include <unistd.h>

#include <atomic>
#include <chrono>
#include <condition_variable>
#include <cstdio>
#include <cstdlib>
#include <cstring>
#include <deque>
#include <functional>
#include <mutex>
#include <thread>
#include <vector>

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

struct Job {
  std::function<ncclResult_t()> fn;
  hipEvent_t ready = nullptr;
  unsigned seq = 0;
};

struct Comm {
  int rank = 0, world = 1;
  char name[64] = {0};
  size_t map_bytes = 0;
  char* base = nullptr;
  Header* hdr = nullptr;
  size_t slot_bytes = 0;
  int broadcast_calls = 0;  // per communicator rank (ranks may be threads of one process: a group, include/neolssvm_hip.h)
  int calls = 0;            // collective calls of this rank (NLS_SHIM_FAIL_CALL)
  std::atomic<int> async_err{0};
  std::atomic<bool> aborted{false};
  // the asynchronous engine (GPU build)
  int device = 0;
  std::thread worker;
  std::mutex m;
  std::condition_variable cv;
  std::deque<Job> q;
  bool stop = false;
  unsigned* flag = nullptr;  // host-coherent: the sequence number of the last finished collective
  unsigned next_seq = 0;
  hipStream_t copy = nullptr;
  char* slot(int r) const { return base + HEADER_BYTES + (size_t)r * slot_bytes; }
};

struct Deferred {
  Comm* c;
  hipStream_t stream;
  std::function<ncclResult_t()> fn;
};
thread_local int g_group_depth = 0;
thread_local std::vector<Deferred> g_deferred;

bool async_mode() {
#ifdef SHIM_HOST_ONLY
  return false;
#else
  static const bool on = [] { const char* e = std::getenv("NLS_SHIM_ASYNC"); return e && e[0] == '1'; }();
  return on;
#endif
}
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
  if (h->poisoned.load() || c->aborted.load()) return ncclSystemError;
  const int gen = h->generation.load();
  if (h->arrived.fetch_add(1) + 1 == c->world) {
    h->arrived.store(0);
    h->generation.fetch_add(1);
    return ncclSuccess;
  }
  const double t0 = now_s();
  while (h->generation.load() == gen) {
    if (h->poisoned.load()) return ncclSystemError;
    if (c->aborted.load()) return ncclSystemError;  // (this rank was aborted: it leaves; the peers keep waiting, as with RCCL)
    if (now_s() - t0 > timeout_s()) {
      h->poisoned.store(1);
      return ncclSystemError;
    }
    sched_yield();
  }
  return ncclSuccess;
}

size_t dtype_bytes(ncclDataType_t t) { return t == ncclDouble ? 8 : 0; }

bool copy_bytes(Comm* c, void* dst, const void* src, size_t bytes, hipMemcpyKind kind) {
#ifdef SHIM_HOST_ONLY
  (void)c;
  (void)kind;
  std::memcpy(dst, src, bytes);
  return true;
#else
  return hipMemcpyAsync(dst, src, bytes, kind, c->copy) == hipSuccess && hipStreamSynchronize(c->copy) == hipSuccess;
#endif
}

ncclResult_t run_allreduce(const void* send, void* recv, size_t count, ncclRedOp_t op, Comm* c) {
  const size_t per = c->slot_bytes / 8;
  std::vector<double> acc(std::min(count, per));
  for (size_t off = 0; off < count; off += per) {
    const size_t m = std::min(per, count - off);
    if (!copy_bytes(c, c->slot(c->rank), static_cast<const double*>(send) + off, m * 8, hipMemcpyDeviceToHost)) return ncclUnhandledCudaError;
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
    if (!copy_bytes(c, static_cast<double*>(recv) + off, acc.data(), m * 8, hipMemcpyHostToDevice)) return ncclUnhandledCudaError;
  }
  return ncclSuccess;
}

ncclResult_t run_broadcast(const void* send, void* recv, size_t count, int root, Comm* c) {
  const size_t per = c->slot_bytes / 8;
  for (size_t off = 0; off < count; off += per) {
    const size_t m = std::min(per, count - off);
    if (c->rank == root && !copy_bytes(c, c->slot(root), static_cast<const double*>(send) + off, m * 8, hipMemcpyDeviceToHost)) return ncclUnhandledCudaError;
    ncclResult_t r = barrier(c);
    if (r != ncclSuccess) return r;
    if (c->rank != root || recv != send) {
      if (!copy_bytes(c, static_cast<double*>(recv) + off, c->slot(root), m * 8, hipMemcpyHostToDevice)) return ncclUnhandledCudaError;
    }
    r = barrier(c);
    if (r != ncclSuccess) return r;
  }
  return ncclSuccess;
}

#ifndef SHIM_HOST_ONLY
// Holds the caller's stream until the worker has finished collective `seq` (or for max_ticks of the 100 MHz clock at most).
__global__ void k_shim_wait(const unsigned* flag, unsigned seq, long long max_ticks) {
  const long long t0 = wall_clock64();
  for (;;) {
    const unsigned v = __hip_atomic_load(flag, __ATOMIC_ACQUIRE, __HIP_MEMORY_SCOPE_SYSTEM);
    if ((int)(v - seq) >= 0) break;
    if (wall_clock64() - t0 > max_ticks) break;
    __builtin_amdgcn_s_sleep(64);
  }
}

void worker_main(Comm* c) {
  (void)hipSetDevice(c->device);
  for (;;) {
    Job j;
    {
      std::unique_lock<std::mutex> lk(c->m);
      c->cv.wait(lk, [&] { return c->stop || !c->q.empty(); });
      if (c->q.empty()) return;  // stop, and drained
      j = std::move(c->q.front());
      c->q.pop_front();
    }
    ncclResult_t r = ncclSuccess;
    if (hipEventSynchronize(j.ready) != hipSuccess) r = ncclUnhandledCudaError;  // the stream has reached the call: the send buffer is final
    (void)hipEventDestroy(j.ready);
    if (r == ncclSuccess) r = (c->aborted.load() || c->async_err.load()) ? ncclSystemError : j.fn();
    if (r != ncclSuccess && c->async_err.load() == 0) c->async_err.store((int)r);
    __atomic_store_n(c->flag, j.seq, __ATOMIC_RELEASE);  // the stream goes on, result or not (an error is reported by ncclCommGetAsyncError)
  }
}

void stop_worker(Comm* c) {
  {
    std::lock_guard<std::mutex> lk(c->m);
    c->stop = true;
  }
  c->cv.notify_all();
  if (c->worker.joinable()) c->worker.join();
}
#endif

// One collective of communicator c on `stream`: inline (synchronous mode, host-only build) or through the worker (NLS_SHIM_ASYNC=1).
ncclResult_t launch(Comm* c, hipStream_t stream, std::function<ncclResult_t()> fn) {
#ifdef SHIM_HOST_ONLY
  (void)stream;
  return fn();
#else
  if (!async_mode()) {
    if (hipStreamSynchronize(stream) != hipSuccess) return ncclUnhandledCudaError;
    return fn();
  }
  if (c->aborted.load()) return ncclInvalidUsage;
  if (c->async_err.load() != 0) return (ncclResult_t)c->async_err.load();  // a communicator in error state takes no more work
  Job j;
  j.fn = std::move(fn);
  j.seq = ++c->next_seq;
  if (hipEventCreateWithFlags(&j.ready, hipEventDisableTiming) != hipSuccess || hipEventRecord(j.ready, stream) != hipSuccess) return ncclUnhandledCudaError;
  hipLaunchKernelGGL(k_shim_wait, dim3(1), dim3(1), 0, stream, c->flag, j.seq, (long long)(timeout_s() * 1e8));
  if (hipGetLastError() != hipSuccess) return ncclUnhandledCudaError;
  {
    std::lock_guard<std::mutex> lk(c->m);
    c->q.push_back(std::move(j));
  }
  c->cv.notify_one();
  return ncclSuccess;
#endif
}

ncclResult_t submit(Comm* c, hipStream_t stream, std::function<ncclResult_t()> fn) {
  if (g_group_depth > 0) {
    g_deferred.push_back(Deferred{c, stream, std::move(fn)});
    return ncclSuccess;
  }
  return launch(c, stream, std::move(fn));
}

// NLS_SHIM_FAIL_RANK / NLS_SHIM_FAIL_CALL: this call of this rank alone fails at the API
bool injected_rank_failure(Comm* c) {
  ++c->calls;
  const char* er = std::getenv("NLS_SHIM_FAIL_RANK");
  const char* ek = std::getenv("NLS_SHIM_FAIL_CALL");
  return er && ek && std::atoi(er) == c->rank && std::atoi(ek) == c->calls;
}

void release(Comm* c) {
#ifndef SHIM_HOST_ONLY
  stop_worker(c);
  // An ABORTED communicator keeps its flag page and copy stream (a leak of test infrastructure): hipHostFree waits for the whole device, and on
  // the one device the ranks share a peer's stream may be held by its spinning kernel - waiting for this very rank.
  if (!c->aborted.load()) {
    if (c->flag) (void)hipHostFree(c->flag);
    if (c->copy) (void)hipStreamDestroy(c->copy);
  }
#endif
  if (c->base) munmap(c->base, c->map_bytes);
  delete c;
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
#ifndef SHIM_HOST_ONLY
  if (hipGetDevice(&c->device) != hipSuccess || hipStreamCreateWithFlags(&c->copy, hipStreamNonBlocking) != hipSuccess ||
      hipHostMalloc(reinterpret_cast<void**>(&c->flag), 64, hipHostMallocCoherent | hipHostMallocMapped) != hipSuccess) {
    release(c);
    return ncclUnhandledCudaError;
  }
  *c->flag = 0;
  if (async_mode()) c->worker = std::thread(worker_main, c);
#endif
  const ncclResult_t r = barrier(c);  // like RCCL: returns once every rank has joined
  if (r != ncclSuccess) {
    release(c);
    return r;
  }
  if (rank == 0) shm_unlink(c->name);  // everybody has it mapped: the name can go
  *comm = reinterpret_cast<ncclComm_t>(c);
  return ncclSuccess;
}

ncclResult_t ncclCommDestroy(ncclComm_t comm) {
  Comm* c = reinterpret_cast<Comm*>(comm);
  if (!c) return ncclSuccess;
  release(c);  // (drains the queue first)
  return ncclSuccess;
}

// Local, like RCCL's: the pending collectives of THIS rank end (their stream is released), the peers are not told.
ncclResult_t ncclCommAbort(ncclComm_t comm) {
  Comm* c = reinterpret_cast<Comm*>(comm);
  if (!c) return ncclSuccess;
  c->aborted.store(true);
  release(c);
  return ncclSuccess;
}

ncclResult_t ncclCommGetAsyncError(ncclComm_t comm, ncclResult_t* asyncError) {
  Comm* c = reinterpret_cast<Comm*>(comm);
  if (!c || !asyncError) return ncclInvalidArgument;
  *asyncError = (ncclResult_t)c->async_err.load();
  return ncclSuccess;
}

ncclResult_t ncclAllReduce(const void* sendbuff, void* recvbuff, size_t count, ncclDataType_t datatype, ncclRedOp_t op, ncclComm_t comm,
                           hipStream_t stream) {
  Comm* c = reinterpret_cast<Comm*>(comm);
  if (!c || !sendbuff || !recvbuff || dtype_bytes(datatype) == 0 || (op != ncclSum && op != ncclMax)) return ncclInvalidArgument;
  if (injected_rank_failure(c)) return ncclInternalError;
  return submit(c, stream, [=] { return run_allreduce(sendbuff, recvbuff, count, op, c); });
}

ncclResult_t ncclBroadcast(const void* sendbuff, void* recvbuff, size_t count, ncclDataType_t datatype, int root, ncclComm_t comm,
                           hipStream_t stream) {
  Comm* c = reinterpret_cast<Comm*>(comm);
  if (!c || !recvbuff || dtype_bytes(datatype) == 0 || root < 0 || root >= c->world) return ncclInvalidArgument;
  ++c->broadcast_calls;
  if (const char* e = std::getenv("NLS_SHIM_FAIL_BROADCAST"))
    if (std::atoi(e) == c->broadcast_calls) return ncclInternalError;
  if (injected_rank_failure(c)) return ncclInternalError;
  return submit(c, stream, [=] { return run_broadcast(sendbuff, recvbuff, count, root, c); });
}

ncclResult_t ncclGroupStart() {
  ++g_group_depth;
  return ncclSuccess;
}

ncclResult_t ncclGroupEnd() {
  if (g_group_depth <= 0) return ncclInvalidUsage;
  if (--g_group_depth > 0) return ncclSuccess;
  ncclResult_t first = ncclSuccess;
  for (auto& d : g_deferred) {
    const ncclResult_t r = launch(d.c, d.stream, std::move(d.fn));
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
