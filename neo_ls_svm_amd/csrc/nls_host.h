// Host-side plumbing shared by the translation units of libneolssvm_hip.so: context, error
// reporting, grow-only workspace, pointer classification, stage timing, collective hook.
#pragma once
#include <hip/hip_runtime.h>
#include <rocblas/rocblas.h>
#include <rocsolver/rocsolver.h>

#include <algorithm>
#include <chrono>
#include <cmath>
#include <cstdarg>
#include <cstdio>
#include <cstring>
#include <map>
#include <string>
#include <vector>

#include "../../include/neolssvm_hip.h"

// ------------------------------------------------------------------------------------------------
// Context
// ------------------------------------------------------------------------------------------------
struct DevBuf {
  void* p = nullptr;
  size_t bytes = 0;
};

struct nls_ctx {
  int device = 0;
  hipStream_t stream = nullptr;
  rocblas_handle blas = nullptr;
  std::string err;
  std::map<std::string, DevBuf> ws;  // grow-only named workspace
  size_t ws_limit = 0;
  size_t hbm_bytes = 0;
  int cus = 0;
  nls_allreduce_fn allreduce = nullptr;
  void* allreduce_user = nullptr;
  int rank = 0, world = 1;
  // XCD patch shape of k_rotate3 (NLS_ROT_PATCH=RxC; 0x0 = plain order, the default: patches raise the L2 hit rate
  // from 0.57 to 0.78 and halve the fabric traffic but run 1-4 % slower, profiles/r01_pmc_summary.md)
  int rot_pr = 0, rot_pc = 0;
  // nls_primal_predict keeps the inverse-factor planes of the last L it was given (fingerprint: host address, size,
  // a checksum of the diagonal and of one entry per row), so repeated predict_std calls skip the 268 MB upload and ztrtri
  const void* pred_L = nullptr;
  int pred_D1 = 0;
  double pred_hash = 0.0;
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

static int fail(nls_ctx* ctx, int code, const char* fmt, ...) {
  char buf[1024];
  va_list ap;
  va_start(ap, fmt);
  vsnprintf(buf, sizeof(buf), fmt, ap);
  va_end(ap);
  if (ctx)
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
    b.p = nullptr;
    b.bytes = 0;
    hipError_t e = hipMalloc(&b.p, bytes);
    if (e != hipSuccess)
      return fail(ctx, NLS_ERR_HIP, "hipMalloc(%zu bytes) for workspace '%s' failed: %s", bytes, name, hipGetErrorString(e));
    b.bytes = bytes;
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
  SpanGuard(nls_ctx* ctx, int stage) : c(ctx), open(span_begin(ctx, stage) == NLS_OK) {}
  ~SpanGuard() {
    if (open) span_end(c);
  }
};

static int do_allreduce(nls_ctx* ctx, double* dbuf, size_t count) {
  if (ctx->world <= 1 || !ctx->allreduce) return NLS_OK;
  HIPCHK(ctx, hipStreamSynchronize(ctx->stream));
  int rc = ctx->allreduce(dbuf, count, ctx->allreduce_user);
  if (rc != 0) return fail(ctx, NLS_ERR_COMM, "all-reduce hook returned %d", rc);
  return NLS_OK;
}

static double wall() {
  return std::chrono::duration<double>(std::chrono::steady_clock::now().time_since_epoch()).count();
}

static int check_info(nls_ctx* ctx, rocblas_int* dinfo, const char* what) {
  rocblas_int info = 0;
  HIPCHK(ctx, hipMemcpyAsync(&info, dinfo, sizeof(info), hipMemcpyDeviceToHost, ctx->stream));
  HIPCHK(ctx, hipStreamSynchronize(ctx->stream));
  if (info != 0) return fail(ctx, NLS_ERR_LINALG, "%s: info = %d (matrix not positive definite / no convergence)", what, (int)info);
  return NLS_OK;
}

