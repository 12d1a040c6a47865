// Host-side plumbing shared by the translation units of libneolssvm_hip.so: context, error
// reporting, grow-only workspace, pointer classification, stage timing, collective hook.
#pragma once
#include <hip/hip_runtime.h>
#include <rccl/rccl.h>  // types only: librccl is loaded with dlopen on first use (nls_comm.hip)
#include <rocblas/rocblas.h>
#include <rocsolver/rocsolver.h>

#include <unistd.h>

#include <algorithm>
#include <atomic>
#include <chrono>
#include <cmath>
#include <cstdarg>
#include <cstdio>
#include <cstring>
#include <map>
#include <string>
#include <thread>
#include <vector>

#include "../../include/neolssvm_hip.h"

// ------------------------------------------------------------------------------------------------
// Context
// ------------------------------------------------------------------------------------------------
struct DevBuf {
  void* p = nullptr;
  size_t bytes = 0;
};

// Entry points of librccl resolved at run time (nls_comm.hip): a single-GPU process never loads the library.
struct RcclApi {
  decltype(&ncclGetUniqueId) GetUniqueId = nullptr;
  decltype(&ncclCommInitRank) CommInitRank = nullptr;
  decltype(&ncclCommDestroy) CommDestroy = nullptr;
  decltype(&ncclCommAbort) CommAbort = nullptr;
  decltype(&ncclCommGetAsyncError) CommGetAsyncError = nullptr;
  decltype(&ncclAllReduce) AllReduce = nullptr;
  decltype(&ncclBroadcast) Broadcast = nullptr;
  decltype(&ncclGroupStart) GroupStart = nullptr;
  decltype(&ncclGroupEnd) GroupEnd = nullptr;
  decltype(&ncclGetErrorString) GetErrorString = nullptr;
};
const RcclApi* rccl_api(std::string* why);

// Device-resident inverse Cholesky factor for predict_std (nls_factor_create): the B-operand planes of U^-1.
struct nls_factor {
  nls_ctx* owner = nullptr;
  int D = 0;
  double *Mr = nullptr, *Mi = nullptr, *mbr = nullptr, *mbi = nullptr, *zero = nullptr;
};

struct nls_ctx {
  int device = 0;
  hipStream_t stream = nullptr;
  rocblas_handle blas = nullptr;
  bool k1_table = false;           // NLS_K1_SINCOS=table: the feature map's epilogue takes the table form of sincos (measured slower: nls_kernels.h)
  double* sintab = nullptr;        // device copy of the feature map's (sin, cos) table (nls_sincos.h), built at context creation
  hipStream_t stream2 = nullptr;   // side stream (created on first use): the Cholesky factor L_ and its download run beside the residual pass
  rocblas_handle blas2 = nullptr;
  hipStream_t copy_stream = nullptr;  // block columns of a Cholesky factor travel to the host on it while the rest is still being factored
  hipStream_t copy_lane[3] = {nullptr, nullptr, nullptr};  // further lanes of the pageable download (nls_dual.hip: helper threads)
  std::vector<hipEvent_t> blk_ev;     // one event per finished block column
  hipEvent_t side_ev[4] = {nullptr, nullptr, nullptr, nullptr};  // fork, potrf done, download done (timing of the side stream)
  hipEvent_t la_ev[2] = {nullptr, nullptr};  // look-ahead of the real Cholesky factorisation (nls_dual.hip: potrf_lower_real): fork, join
  std::string err;
  std::map<std::string, DevBuf> ws;  // grow-only named workspace
  size_t ws_limit = 0;
  size_t hbm_bytes = 0;
  int cus = 0;
  nls_allreduce_fn allreduce = nullptr;
  void* allreduce_user = nullptr;
  int rank = 0, world = 1;
  bool solo = false;  // set by the group's sigma-sharded grid: the context fits alone although it has joined the group's communicator
  ncclComm_t comm = nullptr;  // native RCCL communicator (nls_comm_init_rank); takes precedence over the hook
  double* comm_scratch = nullptr;  // a few doubles on the device for nls_comm_allreduce and the status votes (allocated when the context joins)
  // Failure handling of the collective path ("surface HIP/RCCL errors as status codes"; see comm_wait / comm_vote below)
  bool comm_broken = false;   // the communicator was aborted (deadline, asynchronous error, a failed group member): collective calls fail until
                              // nls_comm_init_rank joins a new one (nls_comm_destroy: back to a single rank)
  bool voted_out = false;     // the last call ended through a status vote: every rank left at the same point, the communicator is intact
  bool vote_victim = false;   // ... and this rank was fine itself (its error names the rank that was not)
  double comm_timeout_s = 0;  // > 0: set with nls_comm_set_timeout; else NLS_COMM_TIMEOUT_S, else 300 s
  std::atomic<int>* abort_flag = nullptr;  // member of a group: 1 + rank of a member that failed outside a vote (the others stop waiting)
  // Measurement hook (nls_comm_set_virtual_rank; bench.py --as-rank r --of W): the context, inside a ONE-rank communicator, does the work of
  // rank virt_rank of virt_world ranks - its row block is the caller's business; here: the rank-0 tridiagonal solve (or not), its own
  // column block of the back-transformation, every exchange with its real payload pushed through the communicator - and takes the other
  // ranks' eigenvector blocks from the copy a first, complete call left behind (virt_Q), so that the results stay those of a real fit.
  int virt_rank = 0, virt_world = 0;
  int virt_n = 0;  // order of the matrix whose eigenpairs the workspace "virt.*" holds (0: none yet)
  bool virt_capture = false;  // the next collective eigendecompositions leave their (lam, real tridiagonal eigenvectors, Q) there
  size_t ws_bytes = 0;        // bytes currently held by the workspace arena
  long twostage_rescues = 0;    // eigendecompositions whose band reduction met a degenerate panel and succeeded at the second, perturbed attempt
  long twostage_fallbacks = 0;  // eigendecompositions whose band reduction met a degenerate panel and fell back to the one-stage panel
  std::vector<nls_factor*> factors;
  // stage times of the most recent eigendecomposition (nls_evd_stage_ms): events 0..5 bracket the five stages
  hipEvent_t evd_ev[6] = {nullptr, nullptr, nullptr, nullptr, nullptr, nullptr};
  int evd_kind = 0, evd_n = 0;
  // XCD patch shape of k_rotate3 (NLS_ROT_PATCH=RxC; 0x0 = plain order).  Unset: a padding-free 8-row patch, with or without a communicator
  // (launch_rotate in nls_lib.hip has the counters)
  int rot_pr = 0, rot_pc = 0;
  bool rot_patch_set = false;
  // k_gram3 tile order: 0 = round-robin (plain: the default everywhere since round 6), 1 = contiguous run of the (split, half tile) list per XCD,
  // 2 = XCD patches, -1 = unset (NLS_GRAM_ORDER=plain / contiguous / patch).  The other orders trade traffic past L2 (17 % / 40 % less) for
  // 1.4-2.9 % of kernel time (profiles/r04_pmc_summary.md, profiles/r05_gram_orders.md, profiles/r06_evd_world8.md).
  int gram_order = -1;
  int k1_stagger_ticks = 0;  // NLS_K1_STAGGER_US: period over which the first-round workgroups of K1 are spread (k1_stagger)
  int rot_kstagger = 0;  // NLS_ROT_KSTAGGER=S: K-walk phase (tr + tc) % S slices per workgroup (see mainloop_3m)
  bool no_resident = false;  // NLS_NO_RESIDENT_PLANES=1: recompute the feature planes per phase even when they would fit
  // stage timing
  struct Span {
    hipEvent_t a, b;
    int stage;
  };
  std::vector<Span> spans;
  std::vector<hipEvent_t> event_pool;
  size_t events_used = 0;
};

extern std::string g_create_error;

// Helper threads of a call (the dual fit's download lanes) must not write ctx->err while the calling thread may be failing too: a thread that
// sets this sink gets its messages there and the caller reports them after the join.
static thread_local std::string* tls_err_sink = nullptr;

static int fail(nls_ctx* ctx, int code, const char* fmt, ...) {
  char buf[1024];
  va_list ap;
  va_start(ap, fmt);
  vsnprintf(buf, sizeof(buf), fmt, ap);
  va_end(ap);
  if (tls_err_sink)
    *tls_err_sink = buf;
  else if (ctx)
    ctx->err = buf;
  else
    g_create_error = buf;
  return code;
}

#define HIPCHK(ctx, call)                                                                                 \
  do {                                                                                                    \
    hipError_t e__ = (call);                                                                              \
    if (e__ != hipSuccess)                                                                                \
      return fail(ctx, NLS_ERR_HIP, "%s failed: %s (%s:%d)", #call, hipGetErrorString(e__), __FILE__, __LINE__); \
  } while (0)

#define BLASCHK(ctx, call)                                                                      \
  do {                                                                                          \
    rocblas_status s__ = (call);                                                                \
    if (s__ != rocblas_status_success)                                                          \
      return fail(ctx, NLS_ERR_HIP, "%s failed: rocblas_status %d (%s:%d)", #call, (int)s__, __FILE__, __LINE__); \
  } while (0)

#define NLSCHK(call)         \
  do {                       \
    int rc__ = (call);       \
    if (rc__ != NLS_OK) return rc__; \
  } while (0)

static inline long round_up(long x, long m) { return (x + m - 1) / m * m; }

static int ws_get(nls_ctx* ctx, const char* name, size_t bytes, void** out) {
  DevBuf& b = ctx->ws[name];
  if (b.bytes < bytes) {
    if (b.p) HIPCHK(ctx, hipFree(b.p));
    ctx->ws_bytes -= b.bytes;
    b.p = nullptr;
    b.bytes = 0;
    // An explicitly set limit is a hard bound (with 10 % slack for the small buffers the chunk planner does not count).
    if (ctx->ws_limit && (double)(ctx->ws_bytes + bytes) > 1.1 * (double)ctx->ws_limit + (double)(64u << 20))
      return fail(ctx, NLS_ERR_ARG, "workspace '%s' (%zu bytes) would take the context to %zu bytes, past the limit of %zu set with "
                  "nls_set_workspace_limit", name, bytes, ctx->ws_bytes + bytes, ctx->ws_limit);
    hipError_t e = hipMalloc(&b.p, bytes);
    if (e != hipSuccess)
      return fail(ctx, NLS_ERR_HIP, "hipMalloc(%zu bytes) for workspace '%s' failed: %s", bytes, name, hipGetErrorString(e));
    b.bytes = bytes;
    ctx->ws_bytes += bytes;
  }
  *out = b.p;
  return NLS_OK;
}
template <class T>
static int ws_get_t(nls_ctx* ctx, const char* name, size_t count, T** out) {
  void* p = nullptr;
  NLSCHK(ws_get(ctx, name, count * sizeof(T), &p));
  *out = reinterpret_cast<T*>(p);
  return NLS_OK;
}

static bool is_device_ptr(const void* p) {
  if (!p) return false;
  hipPointerAttribute_t attr;
  hipError_t e = hipPointerGetAttributes(&attr, p);
  if (e != hipSuccess) {
    (void)hipGetLastError();  // clear the sticky error of an unregistered host pointer
    return false;
  }
  return attr.type == hipMemoryTypeDevice || attr.type == hipMemoryTypeManaged;
}

// Returns a device pointer for `src` (count doubles): itself when already resident, else a staged copy.
static int resident(nls_ctx* ctx, const char* name, const double* src, size_t count, const double** out) {
  if (is_device_ptr(src)) {
    *out = src;
    return NLS_OK;
  }
  double* d = nullptr;
  NLSCHK(ws_get_t(ctx, name, count, &d));
  HIPCHK(ctx, hipMemcpyAsync(d, src, count * sizeof(double), hipMemcpyHostToDevice, ctx->stream));
  *out = d;
  return NLS_OK;
}

// Stage timing with HIP events on the library's stream.
static int span_begin(nls_ctx* ctx, int stage) {
  auto take = [&](hipEvent_t* ev) -> int {
    if (ctx->events_used == ctx->event_pool.size()) {
      hipEvent_t e;
      HIPCHK(ctx, hipEventCreate(&e));
      ctx->event_pool.push_back(e);
    }
    *ev = ctx->event_pool[ctx->events_used++];
    return NLS_OK;
  };
  nls_ctx::Span sp;
  NLSCHK(take(&sp.a));
  NLSCHK(take(&sp.b));
  sp.stage = stage;
  HIPCHK(ctx, hipEventRecord(sp.a, ctx->stream));
  ctx->spans.push_back(sp);
  return NLS_OK;
}
static int span_end(nls_ctx* ctx) {
  HIPCHK(ctx, hipEventRecord(ctx->spans.back().b, ctx->stream));
  return NLS_OK;
}
static int spans_collect(nls_ctx* ctx, double* timings) {
  HIPCHK(ctx, hipStreamSynchronize(ctx->stream));
  for (auto& sp : ctx->spans) {
    float ms = 0.f;
    HIPCHK(ctx, hipEventElapsedTime(&ms, sp.a, sp.b));
    if (timings) timings[sp.stage] += ms * 1e-3;
  }
  ctx->spans.clear();
  ctx->events_used = 0;
  return NLS_OK;
}
struct SpanGuard {  // RAII so early returns still close the span
  nls_ctx* c;
  bool open;
  SpanGuard(nls_ctx* ctx, int stage) : c(ctx), open(span_begin(ctx, stage) == NLS_OK) {
    static const bool marks = [] { const char* m = std::getenv("NLS_HOST_MARKS"); return m && m[0] == '1'; }();  // diagnostic: when the HOST reaches each stage
    if (marks) std::fprintf(stderr, "[nls host] stage %d enqueued at %.3f ms\n", stage, 1e3 * std::fmod(std::chrono::duration<double>(std::chrono::steady_clock::now().time_since_epoch()).count(), 1000.0));
  }
  ~SpanGuard() {
    if (open) span_end(c);
  }
};

// ------------------------------------------------------------------------------------------------
// Collectives of the row-sharded primal fit.  With a native communicator they are RCCL calls enqueued on the
// library's stream, each followed by a bounded wait (comm_wait); with only the caller's all-reduce hook the broadcast
// and the all-gather are expressed as sums (zeros outside the owned part), which is exact.
// ------------------------------------------------------------------------------------------------
#define RCCLCHK(ctx, api, call)                                                                              \
  do {                                                                                                       \
    ncclResult_t r__ = (call);                                                                               \
    if (r__ != ncclSuccess)                                                                                  \
      return fail(ctx, NLS_ERR_COMM, "%s failed: %s (%s:%d)", #call, (api)->GetErrorString(r__), __FILE__, __LINE__); \
  } while (0)

// A native communicator always takes the collective path, also with one rank (the RCCL calls then run on the device
// buffers for real: how the single-GPU box validates them); a hook only matters with more than one rank.  A context whose
// communicator was aborted still counts as multi-rank: its collectives fail (NLS_ERR_COMM) instead of silently fitting the shard alone.
static inline bool multi_rank(const nls_ctx* ctx) {
  return !ctx->solo && (ctx->comm != nullptr || ctx->comm_broken || (ctx->world > 1 && ctx->allreduce));
}
// Rank / world as the WORK is divided (the communicator's own, or the virtual ones of the measurement hook).
static inline int work_rank(const nls_ctx* ctx) { return ctx->virt_world > 1 ? ctx->virt_rank : ctx->rank; }
static inline int work_world(const nls_ctx* ctx) { return ctx->virt_world > 1 ? ctx->virt_world : ctx->world; }

// ---- failure handling ---------------------------------------------------------------------------------------------------------------
// The reference has nothing distributed, so the contract is SURVEY.md section 5's: a failure is a status code on EVERY rank, never a hang.
//   1. comm_vote: before each exchange the ranks all-reduce one status slot per rank.  A rank whose local work failed (allocation, launch,
//      argument, factorisation) does not return early - it skips to the next vote and says so; everybody then leaves the call at the same
//      point (the failed rank with its own code, the others with NLS_ERR_COMM naming it) and the communicator stays usable.
//   2. comm_wait: the host never blocks inside the runtime behind a collective.  It polls the stream and gives up - aborting the
//      communicator (ncclCommAbort), which ends the collective's kernel - on an asynchronous RCCL error, when another member of the same
//      group has failed, or after NLS_COMM_TIMEOUT_S (a peer process died, left the call elsewhere, or is stuck).  Every collective is
//      followed by this wait, so no later blocking call (a pageable download) can sit behind an unfinished collective.
static double comm_timeout(const nls_ctx* ctx) {
  if (ctx->comm_timeout_s > 0) return ctx->comm_timeout_s;
  if (const char* e = std::getenv("NLS_COMM_TIMEOUT_S")) {
    const double v = std::atof(e);
    if (v > 0) return v;
  }
  return 300.0;
}

static double wall();

// Gives the communicator up: ncclCommAbort ends the kernels of the pending collectives; the context keeps its rank / world and refuses
// collective work until it joins a new communicator.
static int comm_give_up(nls_ctx* ctx, const char* fmt, ...) {
  char why[768];
  va_list ap;
  va_start(ap, fmt);
  vsnprintf(why, sizeof(why), fmt, ap);
  va_end(ap);
  if (ctx->comm) {
    const RcclApi* api = rccl_api(nullptr);
    if (api && api->CommAbort) (void)api->CommAbort(ctx->comm);  // (a library without it: the handle is leaked rather than destroyed under a running kernel)
    ctx->comm = nullptr;
    const double t0 = wall();  // the aborted kernels leave the stream; bounded - a stream that stays busy is reported, not waited for
    while (hipStreamQuery(ctx->stream) == hipErrorNotReady && wall() - t0 < 10.0) usleep(200);
    (void)hipGetLastError();
  }
  ctx->comm_broken = true;
  return fail(ctx, NLS_ERR_COMM, "rank %d of %d: %s - communicator aborted (join a new one with nls_comm_init_rank)", ctx->rank, ctx->world, why);
}

static bool comm_trace() {  // NLS_COMM_TRACE=1 (diagnostic): one line on stderr per collective wait - rank, what, how long
  const char* e = std::getenv("NLS_COMM_TRACE");
  return e && e[0] == '1';
}

static int comm_wait(nls_ctx* ctx, const char* what) {
  if (!ctx->comm) {
    HIPCHK(ctx, hipStreamSynchronize(ctx->stream));
    return NLS_OK;
  }
  const RcclApi* api = rccl_api(nullptr);
  const double t0 = wall(), limit = comm_timeout(ctx);
  const bool trace = comm_trace();
  if (trace) std::fprintf(stderr, "[nls comm] rank %d of %d waits for %s\n", ctx->rank, ctx->world, what);
  for (unsigned long polls = 1;; ++polls) {
    const hipError_t q = hipStreamQuery(ctx->stream);
    if (q == hipSuccess) {
      if (trace) std::fprintf(stderr, "[nls comm] rank %d of %d: %s done after %.3f ms\n", ctx->rank, ctx->world, what, 1e3 * (wall() - t0));
      return NLS_OK;
    }
    (void)hipGetLastError();  // (hipErrorNotReady must not reach the next launch check)
    if (q != hipErrorNotReady) return comm_give_up(ctx, "%s: the stream failed (%s)", what, hipGetErrorString(q));
    if ((polls & 31) != 0) continue;
    if (ctx->abort_flag) {
      const int who = ctx->abort_flag->load(std::memory_order_acquire);
      if (who != 0) return comm_give_up(ctx, "%s: rank %d of the group failed", what, who - 1);
    }
    if (api && api->CommGetAsyncError) {
      ncclResult_t ar = ncclSuccess;
      if (api->CommGetAsyncError(ctx->comm, &ar) == ncclSuccess && ar != ncclSuccess && ar != ncclInProgress)
        return comm_give_up(ctx, "%s: asynchronous RCCL error: %s", what, api->GetErrorString(ar));
    }
    const double waited = wall() - t0;
    if (waited > limit)
      return comm_give_up(ctx, "%s did not complete within %g s (NLS_COMM_TIMEOUT_S / nls_comm_set_timeout): a peer rank has failed, has left the call or is stuck",
                          what, limit);
    if (waited > 5e-3)
      usleep(100);
    else if (waited > 3e-4)
      std::this_thread::yield();
  }
}

#define RCCL_ENQUEUE(ctx, api, call)                                                                       \
  do {                                                                                                     \
    ncclResult_t r__ = (call);                                                                             \
    if (r__ != ncclSuccess) return comm_give_up(ctx, "%s failed: %s (%s:%d)", #call, (api)->GetErrorString(r__), __FILE__, __LINE__); \
  } while (0)

static inline int comm_refuse_broken(nls_ctx* ctx) {
  return fail(ctx, NLS_ERR_COMM, "rank %d of %d: the communicator of this context was aborted after an earlier failure - join a new one "
              "(nls_comm_init_rank) or leave it (nls_comm_destroy)", ctx->rank, ctx->world);
}

static int do_allreduce(nls_ctx* ctx, double* dbuf, size_t count) {
  if (!multi_rank(ctx)) return NLS_OK;
  if (ctx->comm_broken) return comm_refuse_broken(ctx);
  if (ctx->comm) {
    const RcclApi* api = rccl_api(nullptr);
    RCCL_ENQUEUE(ctx, api, api->AllReduce(dbuf, dbuf, count, ncclDouble, ncclSum, ctx->comm, ctx->stream));
    return comm_wait(ctx, "ncclAllReduce");
  }
  HIPCHK(ctx, hipStreamSynchronize(ctx->stream));
  int rc = ctx->allreduce(dbuf, count, ctx->allreduce_user);
  if (rc != 0) return fail(ctx, NLS_ERR_COMM, "all-reduce hook returned %d", rc);
  return NLS_OK;
}

// dbuf[0:count] of rank `root` to every rank.
static int do_broadcast(nls_ctx* ctx, double* dbuf, size_t count, int root) {
  if (!multi_rank(ctx)) return NLS_OK;
  if (ctx->comm_broken) return comm_refuse_broken(ctx);
  if (ctx->comm) {
    const RcclApi* api = rccl_api(nullptr);
    RCCL_ENQUEUE(ctx, api, api->Broadcast(dbuf, dbuf, count, ncclDouble, root, ctx->comm, ctx->stream));
    return comm_wait(ctx, "ncclBroadcast");
  }
  if (ctx->rank != root) HIPCHK(ctx, hipMemsetAsync(dbuf, 0, count * sizeof(double), ctx->stream));
  return do_allreduce(ctx, dbuf, count);
}

// Rank r owns dbuf[offs[r] : offs[r + 1]) (offs has world + 1 entries, in doubles); afterwards every rank holds all blocks.
static int do_allgather_blocks(nls_ctx* ctx, double* dbuf, const std::vector<size_t>& offs) {
  if (!multi_rank(ctx)) return NLS_OK;
  if (ctx->comm_broken) return comm_refuse_broken(ctx);
  if (ctx->comm) {
    const RcclApi* api = rccl_api(nullptr);
    RCCL_ENQUEUE(ctx, api, api->GroupStart());
    ncclResult_t first_bad = ncclSuccess;
    const int W = work_world(ctx);
    const bool virt = ctx->virt_world > 1;  // (measurement hook: the payloads of all W blocks through the one-rank communicator)
    for (int r = 0; r < W && first_bad == ncclSuccess; ++r)
      if (offs[r + 1] > offs[r])
        first_bad = api->Broadcast(dbuf + offs[r], dbuf + offs[r], offs[r + 1] - offs[r], ncclDouble, virt ? 0 : r, ctx->comm, ctx->stream);
    const ncclResult_t end = api->GroupEnd();  // always: a failed call must not leave the group open
    if (first_bad != ncclSuccess) return comm_give_up(ctx, "ncclBroadcast (all-gather of blocks) failed: %s", api->GetErrorString(first_bad));
    if (end != ncclSuccess) return comm_give_up(ctx, "ncclGroupEnd failed: %s", api->GetErrorString(end));
    return comm_wait(ctx, "all-gather of blocks (grouped ncclBroadcast)");
  }
  const size_t lo = offs[work_rank(ctx)], hi = offs[work_rank(ctx) + 1], tot = offs[work_world(ctx)];
  if (lo > 0) HIPCHK(ctx, hipMemsetAsync(dbuf, 0, lo * sizeof(double), ctx->stream));
  if (tot > hi) HIPCHK(ctx, hipMemsetAsync(dbuf + hi, 0, (tot - hi) * sizeof(double), ctx->stream));
  return do_allreduce(ctx, dbuf, tot);
}

static const char* nls_code_name(int code) {
  switch (code) {
    case NLS_OK: return "NLS_OK";
    case NLS_ERR_ARG: return "NLS_ERR_ARG";
    case NLS_ERR_HIP: return "NLS_ERR_HIP";
    case NLS_ERR_LINALG: return "NLS_ERR_LINALG";
    case NLS_ERR_COMM: return "NLS_ERR_COMM";
    default: return "an unknown code";
  }
}

// Status vote (see above): `local_rc` is what this rank's work since the previous exchange returned.  One slot per rank, summed: every rank
// learns WHO failed and with which code.  Returns NLS_OK when everybody is fine; else this rank's own code (its message stands), or
// NLS_ERR_COMM naming the first failed rank.  Single rank: local_rc.
static int comm_vote(nls_ctx* ctx, int local_rc, const char* where) {
  if (!multi_rank(ctx)) return local_rc;
  if (ctx->voted_out)  // an earlier vote of this call has failed: every rank is on its way out, nobody votes again
    return local_rc != NLS_OK ? local_rc : fail(ctx, NLS_ERR_COMM, "internal error: a vote %s after a failed vote", where);
  if (ctx->comm_broken) return local_rc != NLS_OK ? local_rc : comm_refuse_broken(ctx);
  const int W = work_world(ctx);
  if (!ctx->comm_scratch || W > NLS_COMM_UTIL_MAX) return local_rc != NLS_OK ? local_rc : fail(ctx, NLS_ERR_COMM, "status vote %s: no vote buffer", where);
  std::vector<double> hv((size_t)W, 0.0);
  hv[(size_t)work_rank(ctx)] = (double)local_rc;
  const std::string own = ctx->err;  // (a failing vote must not bury the local message)
  int rc = NLS_OK;
  if (hipMemcpyAsync(ctx->comm_scratch, hv.data(), sizeof(double) * W, hipMemcpyHostToDevice, ctx->stream) != hipSuccess ||
      hipStreamSynchronize(ctx->stream) != hipSuccess)  // hv is a local; nothing collective is pending here
    rc = comm_give_up(ctx, "status vote %s: upload failed", where);
  if (rc == NLS_OK) rc = do_allreduce(ctx, ctx->comm_scratch, (size_t)W);
  if (rc == NLS_OK && (hipMemcpyAsync(hv.data(), ctx->comm_scratch, sizeof(double) * W, hipMemcpyDeviceToHost, ctx->stream) != hipSuccess ||
                       hipStreamSynchronize(ctx->stream) != hipSuccess))
    rc = comm_give_up(ctx, "status vote %s: download failed", where);
  if (rc != NLS_OK) {  // the vote itself failed: the communicator is gone (comm_give_up)
    if (local_rc == NLS_OK) return rc;
    ctx->err = own;
    return local_rc;
  }
  int first = -1, nbad = 0;
  for (int r = 0; r < W; ++r)
    if (hv[(size_t)r] != 0.0) {
      if (first < 0) first = r;
      ++nbad;
    }
  if (first < 0) return NLS_OK;
  ctx->voted_out = true;
  if (local_rc != NLS_OK) return local_rc;
  ctx->vote_victim = true;
  // (a factorisation / eigensolver failure is a property of the shared problem, not of the rank that met it: the same code everywhere)
  const int code = (int)hv[(size_t)first] == NLS_ERR_LINALG ? NLS_ERR_LINALG : NLS_ERR_COMM;
  return fail(ctx, code, "rank %d of %d failed with %s %s%s; every rank left the call there (this is rank %d; the communicator stays usable)", first, W,
              nls_code_name((int)hv[(size_t)first]), where, nbad > 1 ? " (and other ranks with it)" : "", ctx->rank);
}

// Test hook: NLS_FAULT_INJECT="site", "site:rank" or "site:rank:code" makes the named step of the sharded fit fail on that rank (every rank
// without one) with NLS_ERR_HIP (or the given code: 3 = NLS_ERR_LINALG) - the asymmetric, local failure the votes exist for.  Read per call.
// Sites: prepare, gram, evd, backtransform, sweep, select, cholesky.
static int fault_point(nls_ctx* ctx, const char* site) {
  const char* e = std::getenv("NLS_FAULT_INJECT");
  if (!e || !e[0]) return NLS_OK;
  const size_t ls = std::strlen(site);
  if (std::strncmp(e, site, ls) != 0 || (e[ls] != 0 && e[ls] != ':')) return NLS_OK;
  int code = NLS_ERR_HIP;
  if (e[ls] == ':') {
    if (std::atoi(e + ls + 1) != ctx->rank) return NLS_OK;
    if (const char* c2 = std::strchr(e + ls + 1, ':')) code = std::atoi(c2 + 1);
  }
  if (code < NLS_ERR_ARG || code > NLS_ERR_COMM) code = NLS_ERR_HIP;
  return fail(ctx, code, "injected fault at '%s' on rank %d (NLS_FAULT_INJECT)", site, ctx->rank);
}

// Page-locks a caller's host output buffer for the duration of a call so that its (large) download runs at full PCIe rate instead of
// through the runtime's pageable staging (L_: 268 MB at c3, 800 MB at c4).  The registration is issued where the host would otherwise
// wait for the GPU, so its cost hides behind kernels.  NLS_PIN_OUTPUT=0 disables it; a failed registration is simply not used.
struct HostPin {
  void* p = nullptr;
  void pin(void* ptr, size_t bytes) {
    static const bool off = [] { const char* m = std::getenv("NLS_PIN_OUTPUT"); return m && m[0] == '0'; }();
    // measured (profiles/r03_tail.md): 800 MB (dual c4) 32.7 -> 14.5 ms; 268 MB (primal c3) no gain over the pageable path -> large buffers only
    if (off || !ptr || bytes < ((size_t)512 << 20)) return;
    hipPointerAttribute_t attr;
    if (hipPointerGetAttributes(&attr, ptr) == hipSuccess && attr.type != hipMemoryTypeUnregistered) return;  // device or already pinned
    (void)hipGetLastError();
    if (hipHostRegister(ptr, bytes, hipHostRegisterDefault) == hipSuccess)
      p = ptr;
    else
      (void)hipGetLastError();
  }
  ~HostPin() {
    if (p) (void)hipHostUnregister(p);
  }
};

// True for a host pointer that is already page-locked (hipHostMalloc / hipHostRegister - e.g. through nls_host_register): copies into it are
// asynchronous and run at the full PCIe rate.
static inline bool is_pinned_host(const void* ptr) {
  if (!ptr) return false;
  hipPointerAttribute_t attr;
  if (hipPointerGetAttributes(&attr, ptr) != hipSuccess) {
    (void)hipGetLastError();
    return false;
  }
  return attr.type == hipMemoryTypeHost;
}

// Touches every page of a caller's pageable OUTPUT buffer from helper threads, early in a call, while the host would otherwise wait for the GPU:
// a fresh 800 MB array costs ~ 10^5 first-touch page faults (the kernel zeroes each page), which would otherwise land in the download at the end
// of the call.  Contents are preserved (each page's first byte is read and written back).  Device and page-locked pointers are left alone.
struct Prefault {
  std::thread t[4];
  void start(void* ptr, size_t bytes) {
    if (!ptr || bytes < ((size_t)64 << 20)) return;
    hipPointerAttribute_t attr;
    if (hipPointerGetAttributes(&attr, ptr) == hipSuccess && attr.type != hipMemoryTypeUnregistered) return;
    (void)hipGetLastError();
    constexpr size_t PAGE = 4096;
    const size_t per = (bytes / 4 + PAGE - 1) / PAGE * PAGE;
    for (int i = 0; i < 4; ++i) {
      const size_t lo = std::min(bytes, (size_t)i * per), hi = std::min(bytes, lo + per);
      if (lo >= hi) break;
      volatile char* p = static_cast<volatile char*>(ptr);
      t[i] = std::thread([p, lo, hi] {
        for (size_t off = lo; off < hi; off += PAGE) p[off] = p[off];
      });
    }
  }
  void join() {
    for (auto& th : t)
      if (th.joinable()) th.join();
  }
  ~Prefault() { join(); }
};

// Stage marks of the eigendecomposition (nls_evd_stage_ms): mark i closes stage i - 1.  A stage that does not exist in the path taken is
// marked twice at the same point (0 ms).
static inline void evd_mark(nls_ctx* ctx, int i) {
  if (!ctx->evd_ev[i] && hipEventCreate(&ctx->evd_ev[i]) != hipSuccess) {
    ctx->evd_ev[i] = nullptr;
    return;
  }
  (void)hipEventRecord(ctx->evd_ev[i], ctx->stream);
}

static double wall() {
  return std::chrono::duration<double>(std::chrono::steady_clock::now().time_since_epoch()).count();
}

// Download of a Cholesky factor: the device holds the column-major LOWER factor, which is, byte for byte, the row-major UPPER factor scipy's
// cho_factor(lower=False) returns - and only that triangle is defined output (scipy leaves "random data" in the other one).  Into page-locked
// memory (HostPin: the 800 MB of the dual fit) only the triangle travels: block columns of 256, rows from the block's first row down - half
// the bytes of the square, 22.7 -> 7.7 ms at n = 10^4.  Pageable memory takes the square in one copy (strided copies through the staging
// buffers are no faster: 13.6 against 11.9 ms for the 268 MB of the primal fit); the caller may rely on the upper triangle only.
static int download_factor(nls_ctx* ctx, void* host, const void* dev, int n, long ld_dev, size_t elem_bytes, hipStream_t stream, bool pinned) {
  const size_t spitch = (size_t)ld_dev * elem_bytes, dpitch = (size_t)n * elem_bytes;  // (device: leading dimension ld_dev >= n; host: dense)
  if (!pinned) {
    if (ld_dev == n)
      HIPCHK(ctx, hipMemcpyAsync(host, dev, (size_t)n * n * elem_bytes, hipMemcpyDeviceToHost, stream));
    else
      HIPCHK(ctx, hipMemcpy2DAsync(host, dpitch, dev, spitch, dpitch, (size_t)n, hipMemcpyDeviceToHost, stream));
    return NLS_OK;
  }
  constexpr int NBD = 256;
  for (int j0 = 0; j0 < n; j0 += NBD) {
    const int w = std::min(NBD, n - j0);
    HIPCHK(ctx, hipMemcpy2DAsync(static_cast<char*>(host) + ((size_t)j0 + (size_t)j0 * n) * elem_bytes, dpitch,
                                 static_cast<const char*>(dev) + ((size_t)j0 + (size_t)j0 * ld_dev) * elem_bytes, spitch, (size_t)(n - j0) * elem_bytes, (size_t)w,
                                 hipMemcpyDeviceToHost, stream));
  }
  return NLS_OK;
}

// Pipelined download of a Cholesky factor: the factorisation records blk_ev[b] on its own stream when block column b (columns b nbk ...) is final
// and no longer read; the copy stream then (optionally conjugates and) sends rows b nbk .. n of those columns - the defined triangle only - while the
// following block columns are still being factored.  Into pageable memory each copy blocks the calling thread until it is done: issue these
// AFTER everything else has been enqueued (the thread would wait for the result anyway).
static int ensure_copy_stream(nls_ctx* ctx, int nblk) {
  if (!ctx->copy_stream) HIPCHK(ctx, hipStreamCreateWithFlags(&ctx->copy_stream, hipStreamNonBlocking));
  for (auto& ls : ctx->copy_lane)
    if (!ls) HIPCHK(ctx, hipStreamCreateWithFlags(&ls, hipStreamNonBlocking));
  while ((int)ctx->blk_ev.size() < nblk) {
    hipEvent_t e = nullptr;
    HIPCHK(ctx, hipEventCreateWithFlags(&e, hipEventDisableTiming));
    ctx->blk_ev.push_back(e);
  }
  return NLS_OK;
}
__global__ void k_conj_block(double2* A, long ld, long r0, long c0, long rows, long cols) {
  const long idx = blockIdx.x * (long)blockDim.x + threadIdx.x;
  if (idx >= rows * cols) return;
  double2* p = A + (r0 + idx % rows) + (c0 + idx / rows) * ld;
  p->y = -p->y;
}
// (b_first, b_step, stream: a lane of the block columns on a stream of its own - the pageable download of nls_dual.hip runs two of them.)
static int download_block_columns(nls_ctx* ctx, void* host, void* dev, int n, long ld_dev, size_t elem_bytes, int nbk, bool conj, int b_first = 0,
                                  int b_step = 1, hipStream_t lane = nullptr) {
  const int nblk = (n + nbk - 1) / nbk;
  const size_t spitch = (size_t)ld_dev * elem_bytes, dpitch = (size_t)n * elem_bytes;
  hipStream_t cs = lane ? lane : ctx->copy_stream;
  for (int b = b_first; b < nblk; b += b_step) {
    const long k0 = (long)b * nbk, w = std::min<long>(nbk, n - k0), rows = n - k0;
    HIPCHK(ctx, hipStreamWaitEvent(cs, ctx->blk_ev[b], 0));
    if (conj) {
      hipLaunchKernelGGL(k_conj_block, dim3((unsigned)((rows * w + 255) / 256)), dim3(256), 0, cs, static_cast<double2*>(dev), ld_dev, k0, k0, rows, w);
      HIPCHK(ctx, hipGetLastError());
    }
    HIPCHK(ctx, hipMemcpy2DAsync(static_cast<char*>(host) + ((size_t)k0 + (size_t)k0 * n) * elem_bytes, dpitch,
                                 static_cast<const char*>(dev) + ((size_t)k0 + (size_t)k0 * ld_dev) * elem_bytes, spitch, (size_t)rows * elem_bytes, (size_t)w,
                                 hipMemcpyDeviceToHost, cs));
  }
  return NLS_OK;
}
// infos[b] (relative to block b, 0 = fine) -> out = the first failure as a global 1-based index
__global__ void k_merge_block_info(const int* infos, int nblk, int nbk, int* out) {
  if (threadIdx.x != 0 || blockIdx.x != 0) return;
  int r = 0;
  for (int b = 0; b < nblk && r == 0; ++b)
    if (infos[b] != 0) r = b * nbk + infos[b];
  *out = r;
}

static int check_info(nls_ctx* ctx, rocblas_int* dinfo, const char* what) {
  rocblas_int info = 0;
  HIPCHK(ctx, hipMemcpyAsync(&info, dinfo, sizeof(info), hipMemcpyDeviceToHost, ctx->stream));
  HIPCHK(ctx, hipStreamSynchronize(ctx->stream));
  if (info != 0) return fail(ctx, NLS_ERR_LINALG, "%s: info = %d (matrix not positive definite / no convergence)", what, (int)info);
  return NLS_OK;
}

