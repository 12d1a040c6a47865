// Host side of libneolssvm_hip.so: context, workspace, stage orchestration of the primal path and the C ABI of
// include/neolssvm_hip.h.  Device code lives in nls_gemm.h / nls_gemm3m.h / nls_kernels.h; the dual path is nls_dual.hip.
#include <functional>
#include <mutex>

#include "nls_host.h"
#include "nls_kernels.h"
#include "nls_dual_kernels.h"
#include "nls_zpotrf.h"

using namespace nls;

std::string g_create_error;

// ------------------------------------------------------------------------------------------------
// Shared stage helpers
// ------------------------------------------------------------------------------------------------
struct MapParams {  // device-resident parameters of the affine + ORF map and the padded sizes derived from D
  int d = 0, dk = 0;    // input columns, padded to the K slice
  int D = 0, D1 = 0;    // features, features + bias
  int Kf = 0;           // plane columns: ceil(D / 128) * 128
  int Np = 0;           // rotation output columns: ceil((D + 1) / 64) * 64
  double* shift = nullptr;  // d
  double* Bs = nullptr;     // dk x Kf
};

static int upload_map(nls_ctx* ctx, const double* shift, const double* scale, const double* B, int d, int D,
                      MapParams* mp) {
  if (d <= 0 || D <= 0) return fail(ctx, NLS_ERR_ARG, "d and D must be positive (d=%d, D=%d)", d, D);
  if (!shift || !scale || !B) return fail(ctx, NLS_ERR_ARG, "shift, scale and B must not be NULL");
  for (int k = 0; k < d; ++k) {
    if (scale[k] == 0.0 || !std::isfinite(scale[k]))
      return fail(ctx, NLS_ERR_ARG, "scale[%d] must be finite and non-zero", k);  // _affine_feature_map.py:53-54
    if (!std::isfinite(shift[k])) return fail(ctx, NLS_ERR_ARG, "shift[%d] must be finite", k);
  }
  mp->d = d;
  mp->dk = (int)round_up(d, BK);
  mp->D = D;
  mp->D1 = D + 1;
  mp->Kf = (int)round_up(D, BN);
  mp->Np = (int)round_up(D + 1, m3::BN3);  // 64-wide rotation tiles; also the K of the sweep GEMM (multiple of 16)
  double *dB = nullptr, *dscale = nullptr;
  NLSCHK(ws_get_t(ctx, "map.shift", (size_t)d, &mp->shift));
  NLSCHK(ws_get_t(ctx, "map.scale", (size_t)d, &dscale));
  NLSCHK(ws_get_t(ctx, "map.B", (size_t)d * D, &dB));
  NLSCHK(ws_get_t(ctx, "map.Bs", (size_t)mp->dk * mp->Kf, &mp->Bs));
  HIPCHK(ctx, hipMemcpyAsync(mp->shift, shift, sizeof(double) * d, hipMemcpyHostToDevice, ctx->stream));
  HIPCHK(ctx, hipMemcpyAsync(dscale, scale, sizeof(double) * d, hipMemcpyHostToDevice, ctx->stream));
  HIPCHK(ctx, hipMemcpyAsync(dB, B, sizeof(double) * (size_t)d * D, hipMemcpyHostToDevice, ctx->stream));
  const long tot = (long)mp->dk * mp->Kf;
  hipLaunchKernelGGL(k_build_Bs, dim3((unsigned)((tot + 255) / 256)), dim3(256), 0, ctx->stream, dB, dscale, d, D,
                     mp->dk, mp->Kf, mp->Bs);
  HIPCHK(ctx, hipGetLastError());
  return NLS_OK;
}

constexpr size_t SMEM_REAL = 2 * 2 * TILE_DOUBLES * sizeof(double);  // double-buffered A, B of the real tile engine

// K1 into split planes for `rows` rows starting at Xchunk.
static int launch_featuremap_planes(nls_ctx* ctx, const MapParams& mp, const double* Xchunk, long rows, long rows_pad,
                                    const double* rowscale, double* Fc, double* Fs) {
  FeatureMapParams p{};
  p.X = Xchunk;
  double* Xs = nullptr;
  NLSCHK(ws_get_t(ctx, "fm.Xs", (size_t)rows_pad * mp.dk, &Xs));
  hipLaunchKernelGGL(k_shift_pad, dim3((unsigned)((rows_pad * mp.dk + 255) / 256)), dim3(256), 0, ctx->stream, Xchunk, mp.shift, rows, mp.d,
                     rows_pad, mp.dk, Xs);
  p.Xs = Xs;
  p.shift = mp.shift;
  p.Bs = mp.Bs;
  p.rowscale = rowscale;
  p.rows = rows;
  p.d = mp.d;
  p.dk = mp.dk;
  p.D = mp.D;
  p.Kf = mp.Kf;
  p.inv_sqrt_D = 1.0 / std::sqrt((double)mp.D);
  p.Fc = Fc;
  p.Fs = Fs;
  p.phi = nullptr;
  p.stagger_ticks = ctx->k1_stagger_ticks;
  p.sc = sincos_coef();
  p.tc = sincos_tab_coef();
  p.sintab = ctx->k1_table ? ctx->sintab : nullptr;
  dim3 grid((unsigned)(mp.Kf / BN), (unsigned)(rows_pad / BM));
  if (p.sintab)
    hipLaunchKernelGGL((k_featuremap<false, true>), grid, dim3(Cfg4::NTHREADS), SMEM_REAL, ctx->stream, p);
  else
    hipLaunchKernelGGL((k_featuremap<false, false>), grid, dim3(Cfg4::NTHREADS), SMEM_REAL, ctx->stream, p);
  HIPCHK(ctx, hipGetLastError());
  return NLS_OK;
}

// K4: U, Gm for `rows_pad` rows of planes against the rotation planes Mr, Mi (+ bias row mbr, mbi).
static int launch_rotate(nls_ctx* ctx, const MapParams& mp, const double* Fc, const double* Fs, const double* Mr, const double* Mi,
                         const double* mbr, const double* mbi, const double* vr, const double* vi, double* U, double* Gm,
                         const double* inv_rs, long rows_pad) {
  const long tiles_r = rows_pad / BM, tiles_c = mp.Np / m3::BN3;
  // XCD patch order (NLS_ROT_PATCH=RxC; 0x0: plain).  Round-3 counters (profiles/r03_pmc_summary.md, 333 440 rows, 65 column tiles): a patch whose
  // column count DIVIDES the number of column tiles has no padding blocks and costs nothing - 8 x 5: 500 GB past L2 against 932 GB plain
  // (11.9 x the algorithmic bytes against 21.7 x, L2 hit 0.76 against 0.56) at 465.6 against 463.8 ms; the 4 x 8 patch of round 2 (every ninth
  // patch 7/8 empty) took 485.8 ms.  The fabric is shared by the ranks of a node, so with a communicator the order is patched when such a
  // divisor exists.  Round 4: the same order on one GPU too (the time is the same and half the traffic past L2 is half the fabric power);
  // NLS_ROT_PATCH=0x0 restores the plain order.
  int pr = ctx->rot_pr, pc = ctx->rot_pc;
  if (!ctx->rot_patch_set) {
    pr = pc = 0;
    for (int c : {5, 6, 7, 8, 4, 9, 10, 11, 12, 13})
      if (tiles_c % c == 0) {
        pr = 8;
        pc = c;
        break;
      }
  }
  const long grid = pr > 0 ? xcd_patch_grid(tiles_r, tiles_c, pr, pc) : tiles_r * tiles_c;
  hipLaunchKernelGGL(k_rotate3, dim3((unsigned)grid), dim3(m3::NT3), m3::SMEM3, ctx->stream, Fc, Fs, mp.Kf, Mr, Mi, mbr, mbi, mp.Np,
                     vr, vi, U, Gm, inv_rs, tiles_r, pr, pc, ctx->rot_kstagger);
  HIPCHK(ctx, hipGetLastError());
  return NLS_OK;
}

// Rows per chunk so that the per-chunk buffers stay within budget.  `per_row` = bytes of chunk buffers per row.
static long pick_row_chunk_bytes(nls_ctx* ctx, long n, size_t per_row, size_t fixed_bytes) {
  const size_t limit = ctx->ws_limit ? ctx->ws_limit : (size_t)(0.6 * (double)ctx->hbm_bytes);
  size_t avail = limit > 2 * fixed_bytes ? limit - fixed_bytes : limit / 2;
  avail = std::min(avail, (size_t)48 << 30);  // larger chunks buy nothing once launches are >~100 ms
  long rc_max = (long)(avail / per_row) / BM * BM;
  rc_max = std::max<long>(rc_max, BM);
  const long n_pad = round_up(n, BM);
  const long nchunks = (n_pad + rc_max - 1) / rc_max;
  return round_up((n_pad + nchunks - 1) / nchunks, BM);
}
static long pick_row_chunk(nls_ctx* ctx, long n, const MapParams& mp, size_t fixed_bytes) {
  return pick_row_chunk_bytes(ctx, n, 16ull * ((size_t)mp.Kf + (size_t)mp.Np), fixed_bytes);
}

// ------------------------------------------------------------------------------------------------
// Context API
// ------------------------------------------------------------------------------------------------
extern "C" int nls_abi_version(void) { return NLS_ABI_VERSION; }

extern "C" int nls_ctx_create(int device, nls_ctx** out) {
  if (!out) return fail(nullptr, NLS_ERR_ARG, "ctx output pointer is NULL");
  int count = 0;
  hipError_t e = hipGetDeviceCount(&count);
  if (e != hipSuccess || count == 0)
    return fail(nullptr, NLS_ERR_HIP, "no HIP device available (%s); this library has no CPU fallback",
                e == hipSuccess ? "device count is 0" : hipGetErrorString(e));
  if (device < 0 || device >= count) return fail(nullptr, NLS_ERR_ARG, "device %d out of range [0, %d)", device, count);
  nls_ctx* ctx = new nls_ctx();
  ctx->device = device;
  auto bail = [&](const char* what, const char* why) {
    int rc = fail(nullptr, NLS_ERR_HIP, "%s failed: %s", what, why);
    delete ctx;
    return rc;
  };
  if ((e = hipSetDevice(device)) != hipSuccess) return bail("hipSetDevice", hipGetErrorString(e));
  hipDeviceProp_t prop;
  if ((e = hipGetDeviceProperties(&prop, device)) != hipSuccess) return bail("hipGetDeviceProperties", hipGetErrorString(e));
  if (std::strncmp(prop.gcnArchName, "gfx950", 6) != 0) {
    int rc = fail(nullptr, NLS_ERR_HIP, "device %d is %s; this library is built for gfx950 (MI355X) only", device,
                  prop.gcnArchName);
    delete ctx;
    return rc;
  }
  ctx->hbm_bytes = prop.totalGlobalMem;
  ctx->cus = prop.multiProcessorCount;
  if (const char* ep = std::getenv("NLS_ROT_PATCH")) ctx->rot_patch_set = std::sscanf(ep, "%dx%d", &ctx->rot_pr, &ctx->rot_pc) == 2;
  if (const char* er = std::getenv("NLS_NO_RESIDENT_PLANES")) ctx->no_resident = er[0] == '1';
  if (const char* eg = std::getenv("NLS_GRAM_ORDER")) ctx->gram_order = std::string(eg) == "patch" ? 2 : (std::string(eg) != "plain" ? 1 : 0);
  if (const char* es = std::getenv("NLS_K1_STAGGER_US")) ctx->k1_stagger_ticks = std::max(0, std::min(100000, (int)(std::atof(es) * 100.0)));
  if (const char* et = std::getenv("NLS_K1_SINCOS")) ctx->k1_table = std::string(et) == "table";
  if (const char* ek = std::getenv("NLS_ROT_KSTAGGER")) ctx->rot_kstagger = std::max(0, std::min(16, std::atoi(ek)));
  if ((e = hipStreamCreateWithFlags(&ctx->stream, hipStreamNonBlocking)) != hipSuccess)
    return bail("hipStreamCreate", hipGetErrorString(e));
  // (ROCBLAS_USE_HIPBLASLT - Q1 of the two-stage eigendecomposition 43 -> 37.5 ms at n = 1e4 - is rocBLAS's own process-wide switch and the
  // launcher's to export: the library does not touch the host's environment)
  if (rocblas_create_handle(&ctx->blas) != rocblas_status_success) return bail("rocblas_create_handle", "status != success");
  rocblas_set_stream(ctx->blas, ctx->stream);
  {  // the feature map's sincos table (8 KB), computed on the host in long double
    std::vector<double> tab((size_t)4 * SINCOS_TAB_N);
    sincos_tab_fill(tab.data());
    if ((e = hipMalloc(reinterpret_cast<void**>(&ctx->sintab), sizeof(double) * tab.size())) != hipSuccess ||
        (e = hipMemcpyAsync(ctx->sintab, tab.data(), sizeof(double) * tab.size(), hipMemcpyHostToDevice, ctx->stream)) != hipSuccess ||
        (e = hipStreamSynchronize(ctx->stream)) != hipSuccess) {  // (never the null stream: it waits for other work on the device)
      rocblas_destroy_handle(ctx->blas);
      (void)hipStreamDestroy(ctx->stream);
      return bail("sincos table upload", hipGetErrorString(e));
    }
  }
  // Opt in to > 64 KiB of dynamic LDS for the tile kernels.
  const char* lds_fail = nullptr;
  hipError_t lds_err = hipSuccess;
  auto lds_named = [&](const void* f, size_t bytes, const char* name) {
    const hipError_t r = hipFuncSetAttribute(f, hipFuncAttributeMaxDynamicSharedMemorySize, (int)bytes);
    if (r != hipSuccess && !lds_fail) {
      lds_fail = name;
      lds_err = r;
    }
  };
#define lds(f, bytes) lds_named(f, bytes, #f)
  lds(reinterpret_cast<const void*>(k_gram3), m3::SMEM3);
  lds(reinterpret_cast<const void*>(k_rotate3), m3::SMEM3);
  lds(reinterpret_cast<const void*>(k_sweep), SMEM_REAL);
  lds(reinterpret_cast<const void*>(k_featuremap<false, false>), SMEM_REAL);
  lds(reinterpret_cast<const void*>(k_featuremap<true, false>), SMEM_REAL);
  lds(reinterpret_cast<const void*>(k_featuremap_gemv<false>), SMEM_REAL);
  lds(reinterpret_cast<const void*>(k_featuremap<false, true>), SMEM_REAL);
  lds(reinterpret_cast<const void*>(k_featuremap<true, true>), SMEM_REAL);
  lds(reinterpret_cast<const void*>(k_featuremap_gemv<true>), SMEM_REAL);
  lds(reinterpret_cast<const void*>(k_gemm<EPI_STORE>), SMEM_REAL);
  lds(reinterpret_cast<const void*>(k_gemm<EPI_RBF>), SMEM_REAL);
#undef lds
  if (lds_fail) {  // without the opt-in the first tile launch would fail later with a generic launch error
    static char what[256];
    std::snprintf(what, sizeof(what), "dynamic LDS opt-in (%s)", lds_fail);
    rocblas_destroy_handle(ctx->blas);
    (void)hipStreamDestroy(ctx->stream);
    return bail(what, hipGetErrorString(lds_err));
  }
  *out = ctx;
  return NLS_OK;
}

static void factor_free(nls_factor* f) {
  for (double* p : {f->Mr, f->Mi, f->mbr, f->mbi, f->zero})
    if (p) (void)hipFree(p);
  delete f;
}

extern "C" void nls_ctx_destroy(nls_ctx* ctx) {
  if (!ctx) return;
  (void)hipSetDevice(ctx->device);
  (void)hipStreamSynchronize(ctx->stream);
  (void)nls_comm_destroy(ctx);
  if (ctx->comm_scratch) (void)hipFree(ctx->comm_scratch);
  if (ctx->sintab) (void)hipFree(ctx->sintab);
  for (nls_factor* f : ctx->factors) factor_free(f);
  for (auto& kv : ctx->ws)
    if (kv.second.p) (void)hipFree(kv.second.p);
  for (auto ev : ctx->event_pool) (void)hipEventDestroy(ev);
  for (auto ev : ctx->evd_ev)
    if (ev) (void)hipEventDestroy(ev);
  if (ctx->blas) rocblas_destroy_handle(ctx->blas);
  if (ctx->blas2) rocblas_destroy_handle(ctx->blas2);
  for (auto e : ctx->side_ev)
    if (e) (void)hipEventDestroy(e);
  for (auto e : ctx->la_ev)
    if (e) (void)hipEventDestroy(e);
  if (ctx->stream2) (void)hipStreamDestroy(ctx->stream2);
  for (auto e : ctx->blk_ev)
    if (e) (void)hipEventDestroy(e);
  if (ctx->copy_stream) (void)hipStreamDestroy(ctx->copy_stream);
  for (auto ls : ctx->copy_lane)
    if (ls) (void)hipStreamDestroy(ls);
  if (ctx->stream) (void)hipStreamDestroy(ctx->stream);
  delete ctx;
}

extern "C" const char* nls_last_error(const nls_ctx* ctx) { return ctx ? ctx->err.c_str() : g_create_error.c_str(); }

extern "C" int nls_set_allreduce(nls_ctx* ctx, nls_allreduce_fn fn, void* user, int rank, int world) {
  if (!ctx) return NLS_ERR_ARG;
  if (world < 1 || rank < 0 || rank >= world) return fail(ctx, NLS_ERR_ARG, "bad rank/world %d/%d", rank, world);
  if (ctx->comm || ctx->comm_broken) return fail(ctx, NLS_ERR_ARG, "the context already has a native communicator (nls_comm_destroy first)");
  if (fn && world > 1 && !ctx->comm_scratch) {  // the status votes' buffer (comm_vote): allocated while nothing can be waiting for this rank
    HIPCHK(ctx, hipSetDevice(ctx->device));
    HIPCHK(ctx, hipMalloc(&ctx->comm_scratch, NLS_COMM_UTIL_MAX * sizeof(double)));
  }
  ctx->allreduce = (world > 1) ? fn : nullptr;
  ctx->allreduce_user = user;
  ctx->rank = (fn && world > 1) ? rank : 0;
  ctx->world = (fn && world > 1) ? world : 1;
  return NLS_OK;
}

extern "C" int nls_set_workspace_limit(nls_ctx* ctx, size_t bytes) {
  if (!ctx) return NLS_ERR_ARG;
  ctx->ws_limit = bytes;
  return NLS_OK;
}

extern "C" int nls_host_register(nls_ctx* ctx, void* ptr, size_t bytes) {
  if (!ctx) return NLS_ERR_ARG;
  if (!ptr || bytes == 0) return fail(ctx, NLS_ERR_ARG, "nls_host_register: null pointer or zero size");
  HIPCHK(ctx, hipSetDevice(ctx->device));
  HIPCHK(ctx, hipHostRegister(ptr, bytes, hipHostRegisterDefault));
  return NLS_OK;
}
extern "C" int nls_host_unregister(nls_ctx* ctx, void* ptr) {  // (ctx may be NULL: the context that registered the buffer may be gone)
  if (!ptr) return fail(ctx, NLS_ERR_ARG, "nls_host_unregister: null pointer");
  if (ctx) HIPCHK(ctx, hipSetDevice(ctx->device));
  const hipError_t e = hipHostUnregister(ptr);
  if (e != hipSuccess) return fail(ctx, NLS_ERR_HIP, "hipHostUnregister: %s", hipGetErrorString(e));
  return NLS_OK;
}

extern "C" int nls_ws_release(nls_ctx* ctx, size_t min_bytes, size_t* still_held) {
  if (!ctx) return NLS_ERR_ARG;
  HIPCHK(ctx, hipSetDevice(ctx->device));
  HIPCHK(ctx, hipStreamSynchronize(ctx->stream));
  for (auto it = ctx->ws.begin(); it != ctx->ws.end();) {
    if (it->second.p && it->second.bytes >= min_bytes) {
      HIPCHK(ctx, hipFree(it->second.p));
      ctx->ws_bytes -= it->second.bytes;
      it = ctx->ws.erase(it);
    } else {
      ++it;
    }
  }
  if (still_held) *still_held = ctx->ws_bytes;
  return NLS_OK;
}

extern "C" int nls_device_malloc(nls_ctx* ctx, size_t bytes, void** dptr) {
  if (!ctx || !dptr) return NLS_ERR_ARG;
  HIPCHK(ctx, hipSetDevice(ctx->device));
  HIPCHK(ctx, hipMalloc(dptr, bytes));
  return NLS_OK;
}
extern "C" int nls_device_free(nls_ctx* ctx, void* dptr) {
  if (!ctx) return NLS_ERR_ARG;
  HIPCHK(ctx, hipFree(dptr));
  return NLS_OK;
}
extern "C" int nls_memcpy_h2d(nls_ctx* ctx, void* dst, const void* src, size_t bytes) {
  if (!ctx) return NLS_ERR_ARG;
  HIPCHK(ctx, hipSetDevice(ctx->device));
  HIPCHK(ctx, hipMemcpyAsync(dst, src, bytes, hipMemcpyHostToDevice, ctx->stream));  // (the library's own stream: the null stream would wait for every blocking stream of the device)
  HIPCHK(ctx, hipStreamSynchronize(ctx->stream));
  return NLS_OK;
}
extern "C" int nls_memcpy_d2h(nls_ctx* ctx, void* dst, const void* src, size_t bytes) {
  if (!ctx) return NLS_ERR_ARG;
  HIPCHK(ctx, hipSetDevice(ctx->device));
  HIPCHK(ctx, hipMemcpyAsync(dst, src, bytes, hipMemcpyDeviceToHost, ctx->stream));
  HIPCHK(ctx, hipStreamSynchronize(ctx->stream));
  return NLS_OK;
}
extern "C" int nls_synchronize(nls_ctx* ctx) {
  if (!ctx) return NLS_ERR_ARG;
  HIPCHK(ctx, hipSetDevice(ctx->device));
  // the context's own streams, not the device: other contexts (another rank of a group on this device in the tests, the host application's
  // own work) are none of this context's business - and may legitimately be waiting for it
  HIPCHK(ctx, hipStreamSynchronize(ctx->stream));
  for (hipStream_t st : {ctx->stream2, ctx->copy_stream, ctx->copy_lane[0], ctx->copy_lane[1], ctx->copy_lane[2]})
    if (st) HIPCHK(ctx, hipStreamSynchronize(st));
  return NLS_OK;
}
extern "C" int nls_device_info(nls_ctx* ctx, char* name, int name_len, int* compute_units, size_t* hbm_bytes) {
  if (!ctx) return NLS_ERR_ARG;
  hipDeviceProp_t prop;
  HIPCHK(ctx, hipGetDeviceProperties(&prop, ctx->device));
  if (name && name_len > 0) std::snprintf(name, (size_t)name_len, "%s (%s)", prop.name, prop.gcnArchName);
  if (compute_units) *compute_units = prop.multiProcessorCount;
  if (hbm_bytes) *hbm_bytes = prop.totalGlobalMem;
  return NLS_OK;
}

// ------------------------------------------------------------------------------------------------
// K1 hook
// ------------------------------------------------------------------------------------------------
extern "C" int nls_featuremap(nls_ctx* ctx, const double* X, int64_t n, int d, const double* shift, const double* scale,
                              const double* B, int D, double* phi) {
  if (!ctx) return NLS_ERR_ARG;
  if (!X || !phi || n < 0) return fail(ctx, NLS_ERR_ARG, "X/phi NULL or n < 0");
  HIPCHK(ctx, hipSetDevice(ctx->device));
  if (n == 0) return NLS_OK;
  MapParams mp;
  NLSCHK(upload_map(ctx, shift, scale, B, d, D, &mp));
  const double* dX = nullptr;
  NLSCHK(resident(ctx, "in.X", X, (size_t)n * d, &dX));
  const bool out_dev = is_device_ptr(phi);
  const size_t row_bytes = 16ull * (size_t)mp.D1;
  long rc = out_dev ? (long)round_up(n, BM) : std::max<long>(BM, (long)(((size_t)1 << 30) / row_bytes) / BM * BM);
  rc = std::min<long>(rc, round_up(n, BM));
  double* dphi = nullptr;
  if (!out_dev) NLSCHK(ws_get_t(ctx, "fm.phi", (size_t)rc * mp.D1 * 2, &dphi));
  for (long r0 = 0; r0 < n; r0 += rc) {
    const long rows = std::min<long>(rc, n - r0);
    FeatureMapParams p{};
    p.X = dX + r0 * d;
    const long rows_pad = round_up(rows, BM);
    double* Xs = nullptr;
    NLSCHK(ws_get_t(ctx, "fm.Xs", (size_t)rows_pad * mp.dk, &Xs));
    hipLaunchKernelGGL(k_shift_pad, dim3((unsigned)((rows_pad * mp.dk + 255) / 256)), dim3(256), 0, ctx->stream, p.X, mp.shift, rows, mp.d,
                       rows_pad, mp.dk, Xs);
    p.Xs = Xs;
    p.shift = mp.shift;
    p.Bs = mp.Bs;
    p.rowscale = nullptr;
    p.rows = rows;
    p.d = mp.d;
    p.dk = mp.dk;
    p.D = mp.D;
    p.Kf = mp.Kf;
    p.inv_sqrt_D = 1.0 / std::sqrt((double)D);
    p.sc = sincos_coef();
    p.tc = sincos_tab_coef();
    p.sintab = ctx->k1_table ? ctx->sintab : nullptr;
    p.Fc = p.Fs = nullptr;
    p.phi = out_dev ? phi + 2 * r0 * mp.D1 : dphi;
    dim3 grid((unsigned)(round_up(mp.D1, BN) / BN), (unsigned)(round_up(rows, BM) / BM));  // covers the bias column D
    if (p.sintab)
      hipLaunchKernelGGL((k_featuremap<true, true>), grid, dim3(Cfg4::NTHREADS), SMEM_REAL, ctx->stream, p);
    else
      hipLaunchKernelGGL((k_featuremap<true, false>), grid, dim3(Cfg4::NTHREADS), SMEM_REAL, ctx->stream, p);
    HIPCHK(ctx, hipGetLastError());
    if (!out_dev)
      HIPCHK(ctx, hipMemcpyAsync(phi + 2 * r0 * mp.D1, dphi, (size_t)rows * row_bytes, hipMemcpyDeviceToHost, ctx->stream));
    HIPCHK(ctx, hipStreamSynchronize(ctx->stream));
  }
  return NLS_OK;
}

// ------------------------------------------------------------------------------------------------
// Primal fit
// ------------------------------------------------------------------------------------------------
struct PrimalState {
  MapParams mp;
  long n = 0, n_pad = 0, rc = 0;  // local rows, padded, rows per chunk
  double n_total = 0, s_sum = 0, sy_sum = 0, hn = 0;
  double c = 0;  // 1 / (n_total * D1): the reference's normalised complexity diagonal (_neo_ls_svm.py:117-118)
  const double *dX = nullptr, *dy = nullptr, *ds = nullptr;
  double* s_norm = nullptr;  // n_pad
  double* rs = nullptr;      // n_pad  row scale of the planes: s_norm, or 2^-500 where s == 0 (0 beyond n)
  double* inv_rs = nullptr;  // n_pad  1 / rs (0 beyond n)
  bool resident = false;     // feature planes of ALL rows stay in HBM (one K1 pass per fit)
  long plane_rows = 0;       // rows of the plane buffers (rc, or nchunks * rc when resident)
  double *Fc = nullptr, *Fs = nullptr;  // plane_rows x Kf
  int nt = 0, ntri = 0;
  size_t tile_elems = 0;     // ntri * 2 * 128 * 128
  size_t gram_elems = 0;     // tile_elems + border (4 Kf + 2, padded to 8)
  double* gacc = nullptr;    // packed Gram tiles followed by the border record: the all-reduce payload
};

static inline double* planes_c(const PrimalState& st, long r0) { return st.Fc + (st.resident ? r0 * st.mp.Kf : 0); }
static inline double* planes_s(const PrimalState& st, long r0) { return st.Fs + (st.resident ? r0 * st.mp.Kf : 0); }

// Keep the feature planes of all rows resident when they fit next to everything else; then only the rotation
// outputs U, Gm are chunked.
static void plan_primal_chunks(nls_ctx* ctx, PrimalState& st, size_t fixed_bytes) {
  const size_t limit = ctx->ws_limit ? ctx->ws_limit : (size_t)(0.6 * (double)ctx->hbm_bytes);
  const size_t planes_all = 16ull * (size_t)(st.n_pad + BM) * st.mp.Kf;
  const size_t min_rot = 16ull * st.mp.Np * (size_t)std::min<long>(st.n_pad, 32768);
  if (!ctx->no_resident && fixed_bytes + planes_all + min_rot <= limit) {
    st.resident = true;
    st.rc = pick_row_chunk_bytes(ctx, st.n, 16ull * st.mp.Np, fixed_bytes + planes_all);
    st.plane_rows = ((st.n_pad + st.rc - 1) / st.rc) * st.rc;
  } else {
    st.resident = false;
    st.rc = pick_row_chunk(ctx, st.n, st.mp, fixed_bytes);
    st.plane_rows = st.rc;
  }
}

// Normalise the weights by the global sum (_neo_ls_svm.py:110) and set c.  Two halves around the first exchange of a sharded fit: the local
// half uploads the inputs and sums this rank's weights, the second all-reduces {sum s, sum s y, n} and normalises.
static int primal_prepare_local(nls_ctx* ctx, PrimalState& st, const double* X, const double* y, const double* s, long n, int d, double** sums_out) {
  st.n = n;
  st.n_pad = round_up(std::max<long>(n, 1), BM);
  {
    SpanGuard g(ctx, NLS_T_UPLOAD);
    NLSCHK(resident(ctx, "in.X", X, (size_t)n * d, &st.dX));
    NLSCHK(resident(ctx, "in.y", y, (size_t)n, &st.dy));
    NLSCHK(resident(ctx, "in.s", s, (size_t)n, &st.ds));
  }
  const long nblk = (n + 255) / 256;
  double *part = nullptr, *sums = nullptr;
  NLSCHK(ws_get_t(ctx, "pre.part", (size_t)std::max<long>(nblk, 1) * 2, &part));
  NLSCHK(ws_get_t(ctx, "pre.sums", 4, &sums));
  HIPCHK(ctx, hipMemsetAsync(sums, 0, 4 * sizeof(double), ctx->stream));
  if (n > 0) {
    hipLaunchKernelGGL(k_weight_sums, dim3((unsigned)nblk), dim3(256), 0, ctx->stream, st.ds, st.dy, n, part);
    hipLaunchKernelGGL(k_sum_partials, dim3(1), dim3(256), 0, ctx->stream, part, nblk, 2L, sums, 0);
    HIPCHK(ctx, hipGetLastError());
  }
  st.hn = (double)n;  // (a member: the copy may still be in flight when this returns)
  HIPCHK(ctx, hipMemcpyAsync(sums + 2, &st.hn, sizeof(double), hipMemcpyHostToDevice, ctx->stream));
  *sums_out = sums;
  return NLS_OK;
}
static int primal_prepare_finish(nls_ctx* ctx, PrimalState& st, double* sums) {
  const long n = st.n, nblk = (n + 255) / 256;
  {
    SpanGuard g(ctx, NLS_T_ALLREDUCE);
    NLSCHK(do_allreduce(ctx, sums, 3));
  }
  double h[3];
  HIPCHK(ctx, hipMemcpyAsync(h, sums, 3 * sizeof(double), hipMemcpyDeviceToHost, ctx->stream));
  HIPCHK(ctx, hipStreamSynchronize(ctx->stream));
  st.s_sum = h[0];
  st.sy_sum = h[1];
  st.n_total = h[2];
  if (!(st.s_sum > 0.0) || !std::isfinite(st.s_sum))
    return fail(ctx, NLS_ERR_ARG, "sample weights must have a positive finite sum (got %g)", st.s_sum);
  st.c = 1.0 / (st.n_total * (double)st.mp.D1);
  NLSCHK(ws_get_t(ctx, "pre.s_norm", (size_t)st.n_pad, &st.s_norm));
  NLSCHK(ws_get_t(ctx, "pre.rs", (size_t)st.n_pad + BM, &st.rs));
  NLSCHK(ws_get_t(ctx, "pre.inv_rs", (size_t)st.n_pad + BM, &st.inv_rs));
  HIPCHK(ctx, hipMemsetAsync(st.s_norm, 0, st.n_pad * sizeof(double), ctx->stream));
  HIPCHK(ctx, hipMemsetAsync(st.rs, 0, (st.n_pad + BM) * sizeof(double), ctx->stream));
  HIPCHK(ctx, hipMemsetAsync(st.inv_rs, 0, (st.n_pad + BM) * sizeof(double), ctx->stream));
  if (n > 0) {
    hipLaunchKernelGGL(k_scale_vec, dim3((unsigned)nblk), dim3(256), 0, ctx->stream, st.ds, 1.0 / st.s_sum, n, st.s_norm);
    hipLaunchKernelGGL(k_row_scales, dim3((unsigned)nblk), dim3(256), 0, ctx->stream, st.ds, 1.0 / st.s_sum, n, st.rs, st.inv_rs);
    HIPCHK(ctx, hipGetLastError());
  }
  return NLS_OK;
}

// Phase A: normal equations (packed Gram tiles of F = S phi + border vectors) accumulated over this rank's row chunks (the all-reduce
// across ranks follows in primal_front).
static int primal_gram_local(nls_ctx* ctx, PrimalState& st, double* timings) {
  const MapParams& mp = st.mp;
  st.nt = mp.Kf / BM;
  st.ntri = st.nt * (st.nt + 1) / 2;
  st.tile_elems = (size_t)st.ntri * 2 * BM * BN;
  const long bwidth = 4L * mp.Kf + 2;
  st.gram_elems = st.tile_elems + (size_t)round_up(bwidth, 8);
  NLSCHK(ws_get_t(ctx, "gram.acc", st.gram_elems, &st.gacc));
  HIPCHK(ctx, hipMemsetAsync(st.gacc, 0, st.gram_elems * sizeof(double), ctx->stream));
  if (st.plane_rows < st.rc) st.plane_rows = st.rc;
  NLSCHK(ws_get_t(ctx, "chunk.Fc", (size_t)st.plane_rows * mp.Kf, &st.Fc));
  NLSCHK(ws_get_t(ctx, "chunk.Fs", (size_t)st.plane_rows * mp.Kf, &st.Fs));
  // Row split so that one launch fills the chip several times over (one 256-thread workgroup per CU).
  const long half_tiles = 2L * st.ntri;
  long ns_lo = std::max<long>(1, (12L * ctx->cus + half_tiles - 1) / half_tiles), ns_hi = 3 * ns_lo;
  ns_hi = std::min<long>(ns_hi, std::max<long>(1, st.rc / (8 * BK)));
  ns_lo = std::min(ns_lo, ns_hi);
  long nsplit = ns_lo;
  double best_eff = 0.0;
  for (long ns = ns_lo; ns <= ns_hi; ++ns) {  // fill the last round of workgroups (one per CU) as full as possible
    const long blocks = half_tiles * ns, rounds = (blocks + ctx->cus - 1) / ctx->cus;
    const double eff = (double)blocks / (double)(rounds * ctx->cus);
    if (eff > best_eff + 1e-9) {
      best_eff = eff;
      nsplit = ns;
    }
  }
  double *slab = nullptr, *bpart = nullptr;
  NLSCHK(ws_get_t(ctx, "gram.slab", (size_t)nsplit * st.tile_elems, &slab));
  const long bsplit_max = 256;
  NLSCHK(ws_get_t(ctx, "gram.bpart", (size_t)bsplit_max * bwidth, &bpart));
  for (long r0 = 0; r0 < st.n; r0 += st.rc) {
    const long rows = std::min<long>(st.rc, st.n - r0);
    const long rows_pad = round_up(rows, BM);
    {
      SpanGuard g(ctx, NLS_T_FEATUREMAP);
      NLSCHK(launch_featuremap_planes(ctx, mp, st.dX + r0 * mp.d, rows, rows_pad, st.rs + r0, planes_c(st, r0), planes_s(st, r0)));
      if (timings) {
        timings[NLS_T_FEATUREMAP_LAUNCHES] += 1;
        timings[NLS_T_FEATUREMAP_FLOPS] += 2.0 * rows * mp.d * mp.D;
      }
    }
    {
      SpanGuard g(ctx, NLS_T_GRAM);
      const long rps = round_up((rows_pad + nsplit - 1) / nsplit, BK);
      const long ns = (rows_pad + rps - 1) / rps;
      const long gblocks = half_tiles * ns;
      // 0 plain (default), 1 XCD-contiguous, 2 XCD patches (k_gram3).  Rounds 4-5 took the contiguous order inside a communicator ("17 % less
      // traffic past L2 for ranks that share the fabric"); but a rank's traffic past L2 stays inside ITS package (own HBM, own Infinity Cache) and the
      // only thing ever measured is that the order costs 1.4 % of the kernel: 91.2 against 89.6-90.0 ms on rank 0's share of an 8-GPU c3 fit (round 6).
      const int order = ctx->gram_order >= 0 ? ctx->gram_order : 0;
      long ggrid = order == 1 ? round_up(gblocks, 8) : gblocks;
      if (order == 2) {
        const long pa = (st.nt + 3) / 4, npatch = pa * (pa + 1) / 2;
        ggrid = round_up(ns * npatch, 8) * 32;
      }
      hipLaunchKernelGGL(k_gram3, dim3((unsigned)ggrid), dim3(m3::NT3), m3::SMEM3, ctx->stream,
                         planes_c(st, r0), planes_s(st, r0), mp.Kf, rows_pad, st.ntri, rps, slab, gblocks, order);
      HIPCHK(ctx, hipGetLastError());
      hipLaunchKernelGGL(k_gram_reduce, dim3((unsigned)((st.tile_elems + 255) / 256)), dim3(256), 0, ctx->stream, slab, (int)ns,
                         (long)st.tile_elems, st.gacc);
      // Border: bias column and right-hand side (HBM-bound column sums of the same planes).
      const long bsplit = std::min<long>(bsplit_max, std::max<long>(1, rows / 512));
      const long brps = (rows + bsplit - 1) / bsplit;
      const long bns = (rows + brps - 1) / brps;
      hipLaunchKernelGGL(k_border, dim3((unsigned)(mp.Kf / 128), (unsigned)bns), dim3(256), 0, ctx->stream, planes_c(st, r0),
                         planes_s(st, r0), mp.Kf, st.rs + r0, st.dy + r0, rows, brps, bpart);
      hipLaunchKernelGGL(k_sum_partials, dim3((unsigned)((bwidth + 31) / 32)), dim3(256), 0, ctx->stream, bpart, bns, bwidth,
                         st.gacc + st.tile_elems, 1);
      HIPCHK(ctx, hipGetLastError());
      if (timings) {
        timings[NLS_T_GRAM_LAUNCHES] += 1;
        timings[NLS_T_GRAM_FLOPS] += 4.0 * rows * (double)mp.D1 * mp.D1;
      }
    }
  }
  return NLS_OK;
}

// Everything up to the all-reduced normal equations, shared by nls_primal_fit and nls_gram_only.  In a sharded fit every exchange is preceded
// by a status vote (comm_vote, nls_host.h): a rank whose local work failed skips to the vote, and all ranks leave together.
static int primal_front(nls_ctx* ctx, PrimalState& st, int args_rc, const double* X, const double* y, const double* s, long n, int d, const double* shift,
                        const double* scale, const double* B, int D, double* timings, const std::function<void()>& plan) {
  double* sums = nullptr;
  int rc = args_rc;
  if (rc == NLS_OK) rc = [&]() -> int {
    NLSCHK(upload_map(ctx, shift, scale, B, d, D, &st.mp));
    NLSCHK(primal_prepare_local(ctx, st, X, y, s, n, d, &sums));
    return fault_point(ctx, "prepare");
  }();
  NLSCHK(comm_vote(ctx, rc, "before the exchange of the weight sums"));
  rc = [&]() -> int {
    NLSCHK(primal_prepare_finish(ctx, st, sums));
    plan();
    NLSCHK(primal_gram_local(ctx, st, timings));
    return fault_point(ctx, "gram");
  }();
  NLSCHK(comm_vote(ctx, rc, "before the all-reduce of the normal equations"));
  {
    SpanGuard g(ctx, NLS_T_ALLREDUCE);
    NLSCHK(do_allreduce(ctx, st.gacc, st.gram_elems));
  }
  return NLS_OK;
}

static int assemble_A(nls_ctx* ctx, const PrimalState& st, double scale, double2* Acm, double2* b) {
  const int D1 = st.mp.D1;
  dim3 grid((unsigned)((D1 + 255) / 256), (unsigned)(D1 + 1));
  hipLaunchKernelGGL(k_assemble_A, grid, dim3(256), 0, ctx->stream, st.gacc, st.gacc + st.tile_elems, st.mp.D, st.mp.Kf, scale, Acm,
                     (long)D1, b);
  HIPCHK(ctx, hipGetLastError());
  return NLS_OK;
}

extern "C" int nls_gram_only(nls_ctx* ctx, const double* X, const double* y, const double* s, int64_t n, int d,
                             const double* shift, const double* scale, const double* B, int D, double* A, double* b) {
  if (!ctx) return NLS_ERR_ARG;
  ctx->voted_out = ctx->vote_victim = false;
  HIPCHK(ctx, hipSetDevice(ctx->device));
  PrimalState st;
  const int args_rc = (!X || !y || !s || n < 0) ? fail(ctx, NLS_ERR_ARG, "X/y/s NULL or n < 0") : NLS_OK;
  NLSCHK(primal_front(ctx, st, args_rc, X, y, s, n, d, shift, scale, B, D, nullptr, [&] {
    st.rc = pick_row_chunk(ctx, n, st.mp, 0);
    st.plane_rows = st.rc;
  }));
  const int D1 = st.mp.D1;
  double2 *Acm = nullptr, *Arm = nullptr, *db = nullptr;
  NLSCHK(ws_get_t(ctx, "evd.A", (size_t)D1 * D1, &Acm));
  NLSCHK(ws_get_t(ctx, "evd.Q", (size_t)D1 * D1, &Arm));
  NLSCHK(ws_get_t(ctx, "evd.b", (size_t)D1, &db));
  NLSCHK(assemble_A(ctx, st, 1.0, Acm, db));
  hipLaunchKernelGGL(k_cm_to_rm, dim3((unsigned)((D1 + 255) / 256), (unsigned)D1), dim3(256), 0, ctx->stream, Acm, (long)D1, D1,
                     false, Arm);
  HIPCHK(ctx, hipGetLastError());
  if (A) HIPCHK(ctx, hipMemcpyAsync(A, Arm, sizeof(double2) * (size_t)D1 * D1, hipMemcpyDeviceToHost, ctx->stream));
  if (b) HIPCHK(ctx, hipMemcpyAsync(b, db, sizeof(double2) * (size_t)D1, hipMemcpyDeviceToHost, ctx->stream));
  NLSCHK(spans_collect(ctx, nullptr));
  return NLS_OK;
}

// Device buffers of a rotation (B-operand planes, bias row, epilogue vector).
struct RotBuffers {
  double *Mr = nullptr, *Mi = nullptr, *mbr = nullptr, *mbi = nullptr, *vr = nullptr, *vi = nullptr;
};
static int rot_buffers(nls_ctx* ctx, const MapParams& mp, RotBuffers* rb) {
  NLSCHK(ws_get_t(ctx, "rot.Mr", (size_t)mp.Kf * mp.Np, &rb->Mr));
  NLSCHK(ws_get_t(ctx, "rot.Mi", (size_t)mp.Kf * mp.Np, &rb->Mi));
  NLSCHK(ws_get_t(ctx, "rot.mbr", (size_t)mp.Np, &rb->mbr));
  NLSCHK(ws_get_t(ctx, "rot.mbi", (size_t)mp.Np, &rb->mbi));
  NLSCHK(ws_get_t(ctx, "rot.vr", (size_t)mp.Np, &rb->vr));
  NLSCHK(ws_get_t(ctx, "rot.vi", (size_t)mp.Np, &rb->vi));
  return NLS_OK;
}
static int build_rot_planes(nls_ctx* ctx, const MapParams& mp, const double2* M, long si, long sk, bool upper, const RotBuffers& rb) {
  hipLaunchKernelGGL(k_build_rot_planes, dim3((unsigned)((mp.Np + 255) / 256), (unsigned)(mp.Kf + 1)), dim3(256), 0, ctx->stream, M, si,
                     sk, mp.D, mp.Kf, mp.Np, upper, rb.Mr, rb.Mi, rb.mbr, rb.mbi);
  HIPCHK(ctx, hipGetLastError());
  return NLS_OK;
}

extern "C" int nls_rotate_only(nls_ctx* ctx, const double* X, int64_t n, int d, const double* shift, const double* scale,
                               const double* B, int D, const double* Q, const double* v, double* Uout, double* Gmout) {
  if (!ctx) return NLS_ERR_ARG;
  if (!X || !Q || !v || n < 1) return fail(ctx, NLS_ERR_ARG, "X/Q/v NULL or n < 1");
  HIPCHK(ctx, hipSetDevice(ctx->device));
  MapParams mp;
  NLSCHK(upload_map(ctx, shift, scale, B, d, D, &mp));
  const int D1 = mp.D1, Kf = mp.Kf, Np = mp.Np;
  const double* dX = nullptr;
  NLSCHK(resident(ctx, "in.X", X, (size_t)n * d, &dX));
  const long rc = pick_row_chunk(ctx, n, mp, 4ull * D1 * D1 * 16);
  double *Fc = nullptr, *Fs = nullptr, *U = nullptr, *Gm = nullptr, *cmp = nullptr;
  double2 *dQ = nullptr, *dv = nullptr;
  RotBuffers rb;
  NLSCHK(ws_get_t(ctx, "chunk.Fc", (size_t)rc * Kf, &Fc));
  NLSCHK(ws_get_t(ctx, "chunk.Fs", (size_t)rc * Kf, &Fs));
  NLSCHK(ws_get_t(ctx, "evd.Q", (size_t)D1 * D1, &dQ));
  NLSCHK(ws_get_t(ctx, "chol.beta", (size_t)D1, &dv));
  NLSCHK(rot_buffers(ctx, mp, &rb));
  NLSCHK(ws_get_t(ctx, "chunk.U", (size_t)rc * Np, &U));
  NLSCHK(ws_get_t(ctx, "chunk.Gm", (size_t)rc * Np, &Gm));
  if (Uout || Gmout) NLSCHK(ws_get_t(ctx, "rot.compact", (size_t)rc * D1, &cmp));
  HIPCHK(ctx, hipMemcpyAsync(dQ, Q, sizeof(double2) * (size_t)D1 * D1, hipMemcpyHostToDevice, ctx->stream));
  HIPCHK(ctx, hipMemcpyAsync(dv, v, sizeof(double2) * D1, hipMemcpyHostToDevice, ctx->stream));
  NLSCHK(build_rot_planes(ctx, mp, dQ, (long)D1, 1L, false, rb));  // row-major host layout
  hipLaunchKernelGGL(k_split_vec, dim3((unsigned)((Np + 255) / 256)), dim3(256), 0, ctx->stream, dv, D1, Np, rb.vr, rb.vi);
  HIPCHK(ctx, hipGetLastError());
  for (long r0 = 0; r0 < n; r0 += rc) {
    const long rows = std::min<long>(rc, n - r0);
    const long rows_pad = round_up(rows, BM);
    NLSCHK(launch_featuremap_planes(ctx, mp, dX + r0 * d, rows, rows_pad, nullptr, Fc, Fs));
    NLSCHK(launch_rotate(ctx, mp, Fc, Fs, rb.Mr, rb.Mi, rb.mbr, rb.mbi, rb.vr, rb.vi, U, Gm, nullptr, rows_pad));
    for (int which = 0; which < 2; ++which) {
      double* dst = which ? Gmout : Uout;
      if (!dst) continue;
      hipLaunchKernelGGL(k_compact_rows, dim3((unsigned)((rows * D1 + 255) / 256)), dim3(256), 0, ctx->stream, which ? Gm : U, (long)Np,
                         rows, D1, cmp);
      HIPCHK(ctx, hipGetLastError());
      HIPCHK(ctx, hipMemcpyAsync(dst + r0 * D1, cmp, sizeof(double) * (size_t)rows * D1, hipMemcpyDeviceToHost, ctx->stream));
      HIPCHK(ctx, hipStreamSynchronize(ctx->stream));
    }
  }
  HIPCHK(ctx, hipStreamSynchronize(ctx->stream));
  return NLS_OK;
}

// ------------------------------------------------------------------------------------------------
// Compressed gamma sweep.  The sweep evaluates, for every row, the rational functions sum_j U_ij / (gamma + lam_j) on the
// whole gamma grid (_neo_ls_svm.py:146-149: two n x (D+1) x G products, G = 1024).  In x = ln gamma every term
// 1 / (e^x + lam), lam >= 0, is analytic in the strip |Im x| < pi, so on each quarter of the reference's log-spaced grid
// (length 4.2 in x) its Chebyshev interpolant through 32 nodes is exact to rounding: max relative error 2.4e-15 over
// lam in {0} u [1e-16, 1e8] and slightly negative lam (tests/test_sweep_compression.py), Lebesgue sum 3.2.  Hence
//     R (Np x G)  =  Rn (Np x 128) W (128 x G),   Rn_jq = 1 / (node_q + lam_j),  W = barycentric interpolation weights,
// and  U R = (U Rn) W : 2 n Np 128 + 2 n 128 G flops per product instead of 2 n Np G (6.4 x fewer at D = 4096, G = 1024).
// Used for strictly increasing positive grids of more than 256 points that span at most e^17.2 (the reference's spans
// 2e7 = e^16.8); anything else takes the direct product.
// ------------------------------------------------------------------------------------------------
constexpr int SWEEP_PIECES = 4, SWEEP_NODES = 32, SWEEP_GN = SWEEP_PIECES * SWEEP_NODES;  // 128 = one column tile
static bool sweep_compression(const double* gam, int G, int Gp, std::vector<double>& nodes, std::vector<double>& W) {
  if (G <= 2 * SWEEP_GN) return false;
  if (const char* e = std::getenv("NLS_SWEEP_DIRECT"))
    if (e[0] == '1') return false;
  for (int g = 0; g < G; ++g)
    if (!(gam[g] > 0.0) || !std::isfinite(gam[g]) || (g > 0 && gam[g] <= gam[g - 1])) return false;
  // Four pieces of equal length in x = ln gamma; the accuracy statement holds for pieces no longer than the reference
  // grid's (ln(2e7) / 4 = 4.2): longer grids take the direct product.
  const double x0 = std::log(gam[0]), x1 = std::log(gam[G - 1]), h = (x1 - x0) / SWEEP_PIECES;
  if (!(h > 0.0) || h > 4.3) return false;
  nodes.assign(SWEEP_GN, 1.0);
  W.assign((size_t)SWEEP_GN * Gp, 0.0);
  const double pi = 3.14159265358979323846;
  double xn[SWEEP_PIECES][SWEEP_NODES], wt[SWEEP_NODES];
  for (int j = 0; j < SWEEP_NODES; ++j) wt[j] = ((j & 1) ? -1.0 : 1.0) * std::sin((2 * j + 1) * pi / (2.0 * SWEEP_NODES));
  for (int k = 0; k < SWEEP_PIECES; ++k)
    for (int j = 0; j < SWEEP_NODES; ++j) {  // Chebyshev points of the first kind on piece k
      xn[k][j] = x0 + (k + 0.5) * h + 0.5 * h * std::cos((2 * j + 1) * pi / (2.0 * SWEEP_NODES));
      nodes[k * SWEEP_NODES + j] = std::exp(xn[k][j]);
    }
  for (int g = 0; g < G; ++g) {
    const double x = std::log(gam[g]);
    const int k = std::min(SWEEP_PIECES - 1, std::max(0, (int)((x - x0) / h)));
    double tmp[SWEEP_NODES], sum = 0.0;
    int hit = -1;
    for (int j = 0; j < SWEEP_NODES; ++j) {  // barycentric weights of the interpolant through the piece's nodes
      const double dx = x - xn[k][j];
      if (dx == 0.0) hit = j;
      tmp[j] = wt[j] / dx;
      sum += tmp[j];
    }
    for (int j = 0; j < SWEEP_NODES; ++j)
      W[(size_t)(k * SWEEP_NODES + j) * Gp + g] = hit >= 0 ? (j == hit ? 1.0 : 0.0) : tmp[j] / sum;
  }
  return true;
}

extern "C" int nls_sweep_weights(const double* gammas, int G, double* nodes, double* W, int* applies) {
  static_assert(NLS_SWEEP_NODES == SWEEP_GN, "header constant");
  if (!gammas || !nodes || !W || !applies || G < 1) return NLS_ERR_ARG;
  std::vector<double> hn, hw;
  *applies = sweep_compression(gammas, G, G, hn, hw) ? 1 : 0;
  if (*applies) {
    std::memcpy(nodes, hn.data(), sizeof(double) * SWEEP_GN);
    std::memcpy(W, hw.data(), sizeof(double) * (size_t)SWEEP_GN * G);
  }
  return NLS_OK;
}

// A = L L^H in place (lower triangle, column-major interleaved complex, leading dimension lda >= n) on `stream`: nls_zpotrf.h.
// NLS_POTRF=rocsolver takes rocSOLVER / rocBLAS in block columns of 512 instead (round 3's form; `blas` must be bound to `stream`).
// event_cols > 0 (a multiple of 256): blk_ev[b] is recorded on `stream` when block column b of that width is final and no longer read.
// info: device word, 0 or the 1-based index of the first non-positive pivot.
// rhs / ysol (device, n complex numbers each; or NULL): the forward substitution L y = rhs is carried along (k_zpotrf_panel) - the first half of
// beta = cho_solve(L, rhs); zpotrf_solve_conj_tail below is the second.  Not available with NLS_POTRF=rocsolver (returns *ysol_valid = false).
static int zpotrf_lower(nls_ctx* ctx, hipStream_t stream, rocblas_handle blas, double2* A, int n, long lda, int* dinfo, int event_cols,
                        const double2* rhs = nullptr, double2* ysol = nullptr, bool* ysol_valid = nullptr) {
  using namespace zpotrf;
  if (ysol_valid) *ysol_valid = false;
  const char* mode = std::getenv("NLS_POTRF");
  if (mode && std::string(mode) == "rocsolver") {
    constexpr int NBK = 512;
    const int nblk = (n + NBK - 1) / NBK;
    int* binfo = nullptr;
    NLSCHK(ws_get_t(ctx, "chol.binfo", (size_t)nblk, &binfo));
    HIPCHK(ctx, hipMemsetAsync(binfo, 0, sizeof(int) * nblk, stream));
    const rocblas_double_complex z_one(1.0, 0.0);
    const double h_minus = -1.0, h_one = 1.0;
    for (int b = 0; b < nblk; ++b) {
      const int k0 = b * NBK, w = std::min(NBK, n - k0), mrows = n - k0 - w;
      rocblas_double_complex* A11 = reinterpret_cast<rocblas_double_complex*>(A) + k0 + (long)k0 * lda;
      BLASCHK(ctx, rocsolver_zpotrf(blas, rocblas_fill_lower, w, A11, (rocblas_int)lda, reinterpret_cast<rocblas_int*>(binfo + b)));
      if (mrows > 0) {
        BLASCHK(ctx, rocblas_ztrsm(blas, rocblas_side_right, rocblas_fill_lower, rocblas_operation_conjugate_transpose, rocblas_diagonal_non_unit, mrows, w,
                                   &z_one, A11, (rocblas_int)lda, A11 + w, (rocblas_int)lda));
        BLASCHK(ctx, rocblas_zherk(blas, rocblas_fill_lower, rocblas_operation_none, mrows, w, &h_minus, A11 + w, (rocblas_int)lda, &h_one,
                                   A11 + w + (long)w * lda, (rocblas_int)lda));
      }
      if (event_cols > 0) {
        if (event_cols != NBK) return fail(ctx, NLS_ERR_ARG, "zpotrf_lower(rocsolver): block-column events come in columns of %d", NBK);
        HIPCHK(ctx, hipEventRecord(ctx->blk_ev[b], stream));
      }
    }
    hipLaunchKernelGGL(k_merge_block_info, dim3(1), dim3(64), 0, stream, binfo, nblk, NBK, dinfo);
    HIPCHK(ctx, hipGetLastError());
    return NLS_OK;
  }
  const long ldp = round_up(n, BM);
  double *planes = nullptr, *oplanes = nullptr;
  double2* L11w = nullptr;
  const size_t ssz = (size_t)2 * NBZ * ldp, osz = (size_t)2 * NBO * ldp;  // one stacked pair of planes: inner (32 columns), outer (256)
  NLSCHK(ws_get_t(ctx, "zpotrf.planes", 3 * ssz, &planes));
  NLSCHK(ws_get_t(ctx, "zpotrf.oplanes", 3 * osz, &oplanes));
  NLSCHK(ws_get_t(ctx, "zpotrf.L11", (size_t)NBZ * NBZ, &L11w));
  double *S1 = planes, *S2 = planes + ssz, *S3 = planes + 2 * ssz;
  double *O1 = oplanes, *O2 = oplanes + osz, *O3 = oplanes + 2 * osz;
  HIPCHK(ctx, hipMemsetAsync(dinfo, 0, sizeof(int), stream));
  if (hipFuncSetAttribute(reinterpret_cast<const void*>(k_zpotrf_panel), hipFuncAttributeMaxDynamicSharedMemorySize, (int)ZP_LDS) != hipSuccess ||
      hipFuncSetAttribute(reinterpret_cast<const void*>(k_zpotrf_herk), hipFuncAttributeMaxDynamicSharedMemorySize, (int)SMEM_REAL) != hipSuccess)
    return fail(ctx, NLS_ERR_HIP, "complex Cholesky kernels: %zu / %zu bytes of LDS refused", ZP_LDS, SMEM_REAL);
  if (event_cols > 0 && event_cols % NBO != 0) return fail(ctx, NLS_ERR_ARG, "zpotrf_lower: block-column events come in multiples of %d columns", NBO);
  // NLS_ZPOTRF_STAMP=1 (diagnostic): the in-kernel time line of the panel that starts at column 1024 (workgroup 0), printed by the caller's next sync
  static const bool want_stamps = [] { const char* m = std::getenv("NLS_ZPOTRF_STAMP"); return m && m[0] == '1'; }();
  const int stamp_k0 = std::min(1024, ((n - 1) / NBZ / 2) * NBZ);
  long long* dstamps = nullptr;
  if (want_stamps) {
    NLSCHK(ws_get_t(ctx, "zpotrf.stamps", (size_t)8, &dstamps));
    HIPCHK(ctx, hipMemsetAsync(dstamps, 0, 8 * sizeof(long long), stream));
  }
  const bool carry = rhs != nullptr && ysol != nullptr;
  double2* rhs_run = nullptr;  // the running right-hand side of the carried forward substitution
  if (carry) {
    NLSCHK(ws_get_t(ctx, "zpotrf.rhs", (size_t)n, &rhs_run));
    HIPCHK(ctx, hipMemcpyAsync(rhs_run, rhs, sizeof(double2) * n, hipMemcpyDeviceToDevice, stream));
  }
  for (int K0 = 0; K0 < n; K0 += NBO) {
    const int W = std::min(NBO, n - K0), below = n - K0 - W;  // outer block column, rows below it
    if (below > 0) HIPCHK(ctx, hipMemsetAsync(oplanes, 0, sizeof(double) * 3 * osz, stream));
    for (int k0 = K0; k0 < K0 + W; k0 += NBZ) {
      const int w = std::min(NBZ, K0 + W - k0), mrows = n - k0 - w;  // panel, rows below its diagonal block
      double2* D = A + (long)k0 + (long)k0 * lda;
      const int m_pad = (int)round_up(mrows, BM);
      const int pgrid = std::max(1, (m_pad + ZP_ROWS - 1) / ZP_ROWS);
      const long ko = (long)(k0 - K0) * ldp;  // this panel's 32 k-rows inside the halves of the outer stacks
      long long* stamps = (want_stamps && k0 == stamp_k0) ? dstamps : nullptr;
      hipLaunchKernelGGL(k_zpotrf_panel, dim3((unsigned)pgrid), dim3(ZP_THREADS), ZP_LDS, stream, D, lda, w, k0, mrows, m_pad, S1, S2, S3, ldp, O1 + ko, O2 + ko,
                         O3 + ko, (long)NBO, K0 + W - (k0 + w), L11w, rhs_run, ysol, dinfo, stamps);
      const int icols = K0 + W - (k0 + w);  // columns of the outer block right of the panel: the panel's own (tall) update
      if (icols > 0) {
        const int nt = m_pad / BM, nct = std::min(nt, (icols + BM - 1) / BM);
        const int tiles = nct * (nct + 1) / 2 + (nt - nct) * nct;
        hipLaunchKernelGGL(k_zpotrf_herk, dim3((unsigned)tiles, 2), dim3(Cfg4::NTHREADS), SMEM_REAL, stream, D + w + (long)w * lda, lda, mrows, icols, nct,
                           2 * NBZ / BK, S1, S2, S3, ldp, pgrid > 1 ? L11w : (const double2*)nullptr, D, w);
      } else if (pgrid > 1) {
        hipLaunchKernelGGL(k_zpotrf_putback, dim3(1), dim3(256), 0, stream, L11w, D, lda, w);
      }
      HIPCHK(ctx, hipGetLastError());
    }
    if (below > 0) {  // the trailing matrix beyond the outer block column, once, with K = 2 x 256 (stacked planes)
      const int nt = (int)(round_up(below, BM) / BM);
      hipLaunchKernelGGL(k_zpotrf_herk, dim3((unsigned)(nt * (nt + 1) / 2), 2), dim3(Cfg4::NTHREADS), SMEM_REAL, stream, A + (long)(K0 + W) + (long)(K0 + W) * lda,
                         lda, below, below, nt, 2 * NBO / BK, O1, O2, O3, ldp, (const double2*)nullptr, (double2*)nullptr, 0);
      HIPCHK(ctx, hipGetLastError());
    }
    if (event_cols > 0 && ((K0 + NBO) % event_cols == 0 || K0 + NBO >= n)) HIPCHK(ctx, hipEventRecord(ctx->blk_ev[K0 / event_cols], stream));
  }
  if (want_stamps) {  // diagnostic: synchronises
    long long h[8];
    HIPCHK(ctx, hipMemcpyAsync(h, dstamps, sizeof(h), hipMemcpyDeviceToHost, stream));
    HIPCHK(ctx, hipStreamSynchronize(stream));
    std::fprintf(stderr, "[zpotrf stamps] n %d panel at column %d, workgroup 0 (us): block in LDS %.2f | factored +%.2f | rhs solved +%.2f | rows solved at %.2f | "
                         "barrier at %.2f | stores drained at %.2f\n", n, stamp_k0, (h[1] - h[0]) * 0.01, (h[2] - h[1]) * 0.01, (h[3] - h[2]) * 0.01,
                 (h[4] - h[0]) * 0.01, (h[5] - h[0]) * 0.01, (h[6] - h[0]) * 0.01);
  }
  if (ysol_valid) *ysol_valid = carry;
  return NLS_OK;
}

// Second half of beta = cho_solve(L, b): L^H beta = y in place in `y`, where `Lc` holds conj(L) (the factor after the download's conjugation).
static int zpotrf_solve_conj_tail(nls_ctx* ctx, hipStream_t stream, const double2* Lc, int n, long lda, double2* y) {
  using namespace zpotrf;
  double2* sums = nullptr;
  NLSCHK(ws_get_t(ctx, "zpotrf.sums", (size_t)NBO, &sums));
  const int nblk = (n + NBO - 1) / NBO;
  for (int b = nblk - 1; b >= 0; --b) {
    const int K0 = b * NBO, W = std::min(NBO, n - K0);
    const bool tail = K0 + W < n;
    if (tail) hipLaunchKernelGGL(k_ztrsv_outer_sum, dim3((unsigned)W), dim3(256), 0, stream, Lc, lda, n, K0, W, y, sums);
    hipLaunchKernelGGL(k_ztrsv_block, dim3(1), dim3(256), 0, stream, Lc, lda, K0, W, y, tail ? sums : (const double2*)nullptr);
  }
  HIPCHK(ctx, hipGetLastError());
  return NLS_OK;
}

// Hook (tests / profiling): the primal path's own complex Cholesky factorisation on host data.  A: n x n column-major interleaved complex, lower
// triangle in, L out (the strict upper triangle is returned as it came); *info = 0 or the 1-based index of the first non-positive pivot.
extern "C" int nls_zcholesky_only(nls_ctx* ctx, double* A, int n, int* info) {
  if (!ctx) return NLS_ERR_ARG;
  if (!A || !info || n < 1) return fail(ctx, NLS_ERR_ARG, "nls_zcholesky_only: null pointer or n < 1");
  HIPCHK(ctx, hipSetDevice(ctx->device));
  double2* dA = nullptr;
  int* dinfo = nullptr;
  NLSCHK(ws_get_t(ctx, "hook.zchol", (size_t)n * n, &dA));
  NLSCHK(ws_get_t(ctx, "chol.info2", 4, &dinfo));
  HIPCHK(ctx, hipMemcpyAsync(dA, A, sizeof(double2) * (size_t)n * n, hipMemcpyHostToDevice, ctx->stream));
  BLASCHK(ctx, rocblas_set_stream(ctx->blas, ctx->stream));
  NLSCHK(zpotrf_lower(ctx, ctx->stream, ctx->blas, dA, n, n, dinfo, 0));
  int hinfo = 0;
  HIPCHK(ctx, hipMemcpyAsync(A, dA, sizeof(double2) * (size_t)n * n, hipMemcpyDeviceToHost, ctx->stream));
  HIPCHK(ctx, hipMemcpyAsync(&hinfo, dinfo, sizeof(hinfo), hipMemcpyDeviceToHost, ctx->stream));
  HIPCHK(ctx, hipStreamSynchronize(ctx->stream));
  *info = hinfo;
  return NLS_OK;
}

extern "C" int nls_primal_fit(nls_ctx* ctx, const nls_primal_fit_args* a) {
  if (!ctx) return NLS_ERR_ARG;
  if (!a) return fail(ctx, NLS_ERR_ARG, "args is NULL");
  ctx->voted_out = ctx->vote_victim = false;
  // In a sharded fit a rank never leaves between two exchanges on its own (its peers would wait for it): the status of each stretch of local
  // work - these argument checks included - goes into the vote before the next exchange (comm_vote, nls_host.h).  One rank: plain early returns.
  const int args_rc = [&]() -> int {
    if (!a->X || !a->y || !a->s || !a->gammas) return fail(ctx, NLS_ERR_ARG, "X, y, s and gammas must not be NULL");
    if (a->n < 1 || a->G < 1) return fail(ctx, NLS_ERR_ARG, "n and G must be >= 1 (n=%ld, G=%d)", (long)a->n, a->G);
    if (a->gamma_index_in >= a->G) return fail(ctx, NLS_ERR_ARG, "gamma_index_in out of range");
    if (a->flags & ~(NLS_FIT_SWEEP_ONLY | NLS_FIT_FINISH_IF_BELOW | NLS_FIT_RESIDUALS_FROM_SWEEP)) return fail(ctx, NLS_ERR_ARG, "unknown flag bits 0x%x", (unsigned)a->flags);
    return NLS_OK;
  }();
  HIPCHK(ctx, hipSetDevice(ctx->device));
  const double t_start = wall();
  double tm[NLS_NUM_TIMINGS];
  std::memset(tm, 0, sizeof(tm));
  ctx->spans.clear();
  ctx->events_used = 0;

  PrimalState st;
  const long n = a->n;
  const int G = a->G, is_clf = a->is_classifier ? 1 : 0;
  Prefault prefault;  // a fresh pageable L_ buffer: its pages are touched by helper threads behind the Gram / eigendecomposition (joined before the download)
  if (a->L && args_rc == NLS_OK) prefault.start(a->L, sizeof(double2) * (size_t)(a->D + 1) * (size_t)(a->D + 1));
  static const bool host_marks = [] { const char* m = std::getenv("NLS_HOST_MARKS"); return m && m[0] == '1'; }();
  if (host_marks) std::fprintf(stderr, "[nls host] nls_primal_fit entered at %.3f ms\n", 1e3 * std::fmod(wall(), 1000.0));
  const MapParams& mp = st.mp;
  const int Gp = (int)round_up(std::max(G, 1), BN);

  // ---- phase A: map, weights, normal equations (two exchanges: the weight sums, the packed Gram block) ----------------------------
  NLSCHK(primal_front(ctx, st, args_rc, a->X, a->y, a->s, n, a->d, a->shift, a->scale, a->B, a->D, tm, [&] {
    const size_t fixed = 2ull * st.n_pad * Gp * 8 + 6ull * st.mp.D1 * st.mp.D1 * 16 + 2ull * st.mp.Kf * st.mp.Np * 8 + (size_t)st.mp.Np * Gp * 8;
    plan_primal_chunks(ctx, st, fixed);
    tm[NLS_T_ROW_CHUNK] = (double)st.rc;
  }));
  const int D1 = mp.D1, Kf = mp.Kf, Np = mp.Np;

  // ---- phase B: EVD of A / c (P4) --------------------------------------------------------------
  double2 *Acm = nullptr, *Qcm = nullptr, *db = nullptr;
  double *lam = nullptr, *evd_e = nullptr, *dgam = nullptr, *R = nullptr;
  rocblas_int* dinfo = nullptr;
  RotBuffers rb;
  HostPin pinL;  // the Gram kernels are in flight: page-lock the L_ output now, behind them
  std::vector<double> hnodes, hW;
  bool compressed = false;  // final once the smallest eigenvalue is known (below)
  double* Wd = nullptr;
  // Identity complexity matrix (the only one the reference reaches): A / c = Q Lam Q^H with c = 1 / (n (D+1)), Q unitary,
  // leverage s^2 |phi Q|^2 / c.  General C (8(f) #4): C <- C / mean|diag C| / (n (D+1)) (_neo_ls_svm.py:117), C = Lc Lc^H,
  // Lc^-1 A Lc^-H = W Lam W^H, Q = Lc^-H W, which is what eigh(A, b=C) returns (Q^H C Q = I, so lu_solve(C Q, x) = Q^H x):
  // the same rotation with the 1 / c factors replaced by 1.
  const bool general_C = a->Cmat != nullptr;
  int Gr = Gp;  // columns of the matrix the U / Gm products run against: the 128 nodes when the sweep is compressed
  const double inv_c = general_C ? 1.0 : 1.0 / st.c;
  double2 *Cn = nullptr, *Lc = nullptr;
  double2* Qev_keep = nullptr;  // the eigenvectors (column-major), also needed for the re-solve at gamma*
  int rc = [&]() -> int {
    NLSCHK(ws_get_t(ctx, "evd.A", (size_t)D1 * D1, &Acm));
    NLSCHK(ws_get_t(ctx, "evd.Q", (size_t)D1 * D1, &Qcm));
    NLSCHK(ws_get_t(ctx, "evd.b", (size_t)D1, &db));
    NLSCHK(ws_get_t(ctx, "evd.lam", (size_t)D1, &lam));
    NLSCHK(ws_get_t(ctx, "evd.e", (size_t)D1, &evd_e));
    NLSCHK(ws_get_t(ctx, "evd.info", 4, &dinfo));
    NLSCHK(rot_buffers(ctx, mp, &rb));
    if (a->L) pinL.pin(a->L, sizeof(double2) * (size_t)D1 * D1);
    compressed = sweep_compression(a->gammas, G, Gp, hnodes, hW);
    NLSCHK(ws_get_t(ctx, "sweep.gammas", (size_t)std::max(G, SWEEP_GN), &dgam));
    NLSCHK(ws_get_t(ctx, "sweep.R", (size_t)Np * Gp, &R));
    if (compressed) {
      NLSCHK(ws_get_t(ctx, "sweep.W", (size_t)SWEEP_GN * Gp, &Wd));
      HIPCHK(ctx, hipMemcpyAsync(Wd, hW.data(), sizeof(double) * hW.size(), hipMemcpyHostToDevice, ctx->stream));
      HIPCHK(ctx, hipStreamSynchronize(ctx->stream));  // hW is a local of this call
    }
    return NLS_OK;
  }();
  {
    SpanGuard g(ctx, NLS_T_EVD);
    if (rc == NLS_OK) rc = [&]() -> int {
      NLSCHK(assemble_A(ctx, st, 1.0, Acm, db));
      NLSCHK(assemble_A(ctx, st, inv_c, Qcm, nullptr));
      if (general_C) {
        NLSCHK(ws_get_t(ctx, "gevd.C", (size_t)D1 * D1, &Cn));
        NLSCHK(ws_get_t(ctx, "gevd.L", (size_t)D1 * D1, &Lc));
        double dsum = 0.0;
        for (int k = 0; k < D1; ++k) dsum += std::fabs(a->Cmat[(size_t)k * D1 + k]);
        if (!(dsum > 0.0) || !std::isfinite(dsum)) return fail(ctx, NLS_ERR_ARG, "complexity matrix has a zero or non-finite diagonal");
        const double cscale = 1.0 / (dsum / D1) / (st.n_total * (double)D1);
        std::vector<double2> hC((size_t)D1 * D1);
        for (size_t k = 0; k < hC.size(); ++k) hC[k] = make_double2(a->Cmat[k] * cscale, 0.0);  // symmetric: row-major == column-major
        HIPCHK(ctx, hipMemcpyAsync(Cn, hC.data(), sizeof(double2) * hC.size(), hipMemcpyHostToDevice, ctx->stream));
        HIPCHK(ctx, hipStreamSynchronize(ctx->stream));
        HIPCHK(ctx, hipMemcpyAsync(Lc, Cn, sizeof(double2) * (size_t)D1 * D1, hipMemcpyDeviceToDevice, ctx->stream));
        BLASCHK(ctx, rocblas_set_stream(ctx->blas, ctx->stream));
        // (the library's own factorisation: rocsolver_zpotrf is not safe when two contexts fit at the same time, profiles/r04_two_contexts.md)
        NLSCHK(zpotrf_lower(ctx, ctx->stream, ctx->blas, Lc, D1, D1, reinterpret_cast<int*>(dinfo), 0));
        NLSCHK(check_info(ctx, dinfo, "Cholesky factorisation of the complexity matrix"));
        const rocblas_double_complex one(1.0, 0.0);
        auto* zL = reinterpret_cast<const rocblas_double_complex*>(Lc);
        auto* zA = reinterpret_cast<rocblas_double_complex*>(Qcm);
        BLASCHK(ctx, rocblas_ztrsm(ctx->blas, rocblas_side_left, rocblas_fill_lower, rocblas_operation_none, rocblas_diagonal_non_unit, D1, D1,
                                   &one, zL, D1, zA, D1));
        BLASCHK(ctx, rocblas_ztrsm(ctx->blas, rocblas_side_right, rocblas_fill_lower, rocblas_operation_conjugate_transpose,
                                   rocblas_diagonal_non_unit, D1, D1, &one, zL, D1, zA, D1));
      }
      return fault_point(ctx, "evd");
    }();
    double2* Qev = nullptr;  // eigenvectors: in Qcm (rocSOLVER path) or in the EVD's own workspace
    // (sharded: the status of the stretch above travels into the eigendecomposition's first vote; its exchanges are guarded inside)
    rc = evd_hermitian(ctx, Qcm, D1, lam, evd_e, dinfo, &Qev, true, rc);
    if (rc == NLS_OK) rc = [&]() -> int {
      if (general_C) {
        const rocblas_double_complex one(1.0, 0.0);
        BLASCHK(ctx, rocblas_ztrsm(ctx->blas, rocblas_side_left, rocblas_fill_lower, rocblas_operation_conjugate_transpose,
                                   rocblas_diagonal_non_unit, D1, D1, &one, reinterpret_cast<const rocblas_double_complex*>(Lc), D1,
                                   reinterpret_cast<rocblas_double_complex*>(Qev), D1));
      }
      NLSCHK(build_rot_planes(ctx, mp, Qev, 1L, (long)D1, false, rb));  // column-major Q
      Qev_keep = Qev;
      hipLaunchKernelGGL(k_compute_v, dim3((unsigned)Np), dim3(256), 0, ctx->stream, Qev, (long)D1, db, D1, inv_c, rb.vr, rb.vi);
      if (compressed) {
        // The interpolation identity needs every pole -lam_j well to the left of the grid: measured through nls_sweep_weights
        // (tests/test_sweep_compression.py), the interpolant of 1 / (gamma + lam) is exact to rounding (1.8e-15) for lam >= -gamma_min / 8,
        // 6e-14 at -gamma_min / 4 and 5e-10 at -gamma_min / 2 (the pole is then 0.69 from the first Chebyshev piece).  A / c is positive
        // semi-definite, so lam_min >= -eps lam_max ~ -1e-12 for the path's matrices (lam_max ~ D + 1); an eigenvalue below
        // -gamma_min / 8 (degenerate weights: lam_max up to 2 n (D+1)) takes the reference's own formula evaluated directly.
        double lam0 = 0.0;
        HIPCHK(ctx, hipMemcpyAsync(&lam0, lam, sizeof(double), hipMemcpyDeviceToHost, ctx->stream));
        HIPCHK(ctx, hipStreamSynchronize(ctx->stream));
        if (!(lam0 > -0.125 * a->gammas[0])) compressed = false;
      }
      Gr = compressed ? SWEEP_GN : Gp;
      HIPCHK(ctx, hipMemcpyAsync(dgam, compressed ? hnodes.data() : a->gammas, sizeof(double) * (compressed ? SWEEP_GN : G), hipMemcpyHostToDevice,
                                 ctx->stream));
      const long tot = (long)Np * Gr;
      hipLaunchKernelGGL(k_rgrid, dim3((unsigned)((tot + 255) / 256)), dim3(256), 0, ctx->stream, lam, dgam, D1, compressed ? SWEEP_GN : G, Np,
                         Gr, R);
      HIPCHK(ctx, hipGetLastError());
      return NLS_OK;
    }();
  }

  // ---- phase C: rotation + sweep per row chunk (P5, P6) ---------------------------------------
  double *U = nullptr, *Gm = nullptr, *num = nullptr, *hs = nullptr;
  double *T1 = nullptr, *T2 = nullptr;  // compressed sweep: U Rn and (Gm Rn) / c of one row chunk
  double *part = nullptr, *errs = nullptr;
  if (rc == NLS_OK) rc = [&]() -> int {
    NLSCHK(ws_get_t(ctx, "chunk.U", (size_t)st.rc * Np, &U));
    NLSCHK(ws_get_t(ctx, "chunk.Gm", (size_t)st.rc * Np, &Gm));
    NLSCHK(ws_get_t(ctx, "sweep.num", (size_t)st.n_pad * Gp, &num));
    NLSCHK(ws_get_t(ctx, "sweep.hs", (size_t)st.n_pad * Gp, &hs));
    if (compressed) {
      NLSCHK(ws_get_t(ctx, "sweep.T1", (size_t)st.rc * SWEEP_GN, &T1));
      NLSCHK(ws_get_t(ctx, "sweep.T2", (size_t)st.rc * SWEEP_GN, &T2));
    }
    NLSCHK(fault_point(ctx, "sweep"));
    const char* ess = std::getenv("NLS_SWEEP_SMALL");  // NLS_SWEEP_SMALL=0: the 128-wide tile for every G (read per call, like the other knobs)
    const bool small_sweep = !(ess && ess[0] == '0');
    for (long r0 = 0; r0 < n; r0 += st.rc) {
      const long rows = std::min<long>(st.rc, n - r0);
      const long rows_pad = round_up(rows, BM);
      if (!st.resident) {
        SpanGuard g(ctx, NLS_T_FEATUREMAP);
        NLSCHK(launch_featuremap_planes(ctx, mp, st.dX + r0 * mp.d, rows, rows_pad, st.rs + r0, st.Fc, st.Fs));
        tm[NLS_T_FEATUREMAP_LAUNCHES] += 1;
        tm[NLS_T_FEATUREMAP_FLOPS] += 2.0 * rows * mp.d * mp.D;
      }
      {
        SpanGuard g(ctx, NLS_T_ROTATE);
        NLSCHK(launch_rotate(ctx, mp, planes_c(st, r0), planes_s(st, r0), rb.Mr, rb.Mi, rb.mbr, rb.mbi, rb.vr, rb.vi, U, Gm,
                             st.inv_rs + r0, rows_pad));
        tm[NLS_T_ROTATE_LAUNCHES] += 1;
        tm[NLS_T_ROTATE_FLOPS] += 8.0 * rows * (double)D1 * D1;
      }
      {
        SpanGuard g(ctx, NLS_T_SWEEP);
        if (compressed) {
          hipLaunchKernelGGL(k_sweep, dim3(1, (unsigned)(rows_pad / BM), 2), dim3(Cfg4::NTHREADS), SMEM_REAL, ctx->stream, U, Gm, Np, R,
                             SWEEP_GN, inv_c, T1, T2, 0L);
          hipLaunchKernelGGL(k_sweep, dim3((unsigned)(Gp / BN), (unsigned)(rows_pad / BM), 2), dim3(Cfg4::NTHREADS), SMEM_REAL, ctx->stream,
                             T1, T2, SWEEP_GN, Wd, Gp, 1.0, num, hs, r0);
        } else if (small_sweep && G <= 32) {  // short grids (the 32-point grid of the gamma x sigma sweep): the streaming form, 32 columns
          hipLaunchKernelGGL(k_sweep_small<2>, dim3(1, (unsigned)((rows_pad + 255) / 256), 2), dim3(256), 0, ctx->stream, U, Gm, Np, R, Gp, inv_c, num, hs, Gp, r0, rows_pad);
        } else if (small_sweep && G <= 64) {
          hipLaunchKernelGGL(k_sweep_small<4>, dim3(1, (unsigned)((rows_pad + 255) / 256), 2), dim3(256), 0, ctx->stream, U, Gm, Np, R, Gp, inv_c, num, hs, Gp, r0, rows_pad);
        } else {
          hipLaunchKernelGGL(k_sweep, dim3((unsigned)(Gp / BN), (unsigned)(rows_pad / BM), 2), dim3(Cfg4::NTHREADS), SMEM_REAL, ctx->stream, U,
                             Gm, Np, R, Gp, inv_c, num, hs, r0);
        }
        HIPCHK(ctx, hipGetLastError());
        tm[NLS_T_SWEEP_LAUNCHES] += 1;
        tm[NLS_T_SWEEP_FLOPS] += 4.0 * rows * (double)D1 * G;
      }
    }

    // ---- P7: per-gamma errors ------------------------------------------------------------------
    // rows per block of the error reduction: at most 256, fewer for small n so that there are ~4 blocks per CU
    const int loo_rows = (int)std::max<long>(16, std::min<long>(LOO_ROWS_PER_BLOCK, n / (4L * ctx->cus)));
    const long nblk = (n + loo_rows - 1) / loo_rows;
    NLSCHK(ws_get_t(ctx, "loo.part", (size_t)nblk * 3 * Gp, &part));
    NLSCHK(ws_get_t(ctx, "loo.errs", (size_t)3 * Gp, &errs));
    {
      SpanGuard g(ctx, NLS_T_LOO);
      hipLaunchKernelGGL(k_loo_errors, dim3((unsigned)nblk), dim3(256), 0, ctx->stream, num, hs, st.dy, st.s_norm, n, G, Gp, is_clf,
                         loo_rows, part);
      hipLaunchKernelGGL(k_sum_partials, dim3((unsigned)((3 * Gp + 31) / 32)), dim3(256), 0, ctx->stream, part, nblk, 3L * Gp, errs, 0);
      HIPCHK(ctx, hipGetLastError());
    }
    return NLS_OK;
  }();
  NLSCHK(comm_vote(ctx, rc, "before the all-reduce of the per-gamma errors"));
  {
    SpanGuard g(ctx, NLS_T_ALLREDUCE);
    NLSCHK(do_allreduce(ctx, errs, (size_t)3 * Gp));
  }

  // ---- P7: selection (the same arithmetic on the same all-reduced numbers on every rank) -----------
  std::vector<double> herrs((size_t)3 * Gp), hobj((size_t)G);
  double lam_min = 0.0;  // smallest eigenvalue of A / c (of C^-1 A for a general C): the positive-definiteness test of gamma* C + A below
  int opt = a->gamma_index_in;
  double gamma_opt = 0.0;
  bool finish = false;
  double *loo_res = nullptr, *loo_lev = nullptr, *loo_std = nullptr, *res = nullptr, *cpart = nullptr, *csum = nullptr;
  rc = [&]() -> int {
    HIPCHK(ctx, hipMemcpyAsync(herrs.data(), errs, sizeof(double) * 3 * Gp, hipMemcpyDeviceToHost, ctx->stream));
    HIPCHK(ctx, hipMemcpyAsync(&lam_min, lam, sizeof(double), hipMemcpyDeviceToHost, ctx->stream));
    HIPCHK(ctx, hipStreamSynchronize(ctx->stream));
    for (int g = 0; g < G; ++g)  // _neo_ls_svm.py:159-165 (same summation order as the reference)
      hobj[g] = is_clf ? (herrs[Gp + g] + herrs[2 * Gp + g]) + herrs[g] : herrs[g];
    if (opt < 0) {
      // numpy.argmin semantics: first minimum, a NaN wins (first NaN is returned).
      opt = 0;
      for (int g = 0; g < G; ++g) {
        if (std::isnan(hobj[g])) {
          opt = g;
          break;
        }
        if (hobj[g] < hobj[opt]) opt = g;
      }
    }
    gamma_opt = a->gammas[opt];
    finish = !(a->flags & NLS_FIT_SWEEP_ONLY) && !((a->flags & NLS_FIT_FINISH_IF_BELOW) && !(hobj[opt] < a->finish_below));
    return NLS_OK;
  }();
  if (rc == NLS_OK && !finish) {  // a non-winning sigma of a gamma x sigma grid: the error curve is all that is needed (no further exchange: every rank
    if (a->finished) *a->finished = 0;  // decides the same from the same numbers)
    NLSCHK(spans_collect(ctx, tm));
    if (a->lam) {
      HIPCHK(ctx, hipMemcpyAsync(a->lam, lam, sizeof(double) * D1, hipMemcpyDeviceToHost, ctx->stream));
      HIPCHK(ctx, hipStreamSynchronize(ctx->stream));
    }
    if (a->loo_errors) std::memcpy(a->loo_errors, herrs.data(), sizeof(double) * G);
    if (a->objective) std::memcpy(a->objective, hobj.data(), sizeof(double) * G);
    if (a->gamma_index) *a->gamma_index = opt;
    tm[NLS_T_TOTAL] = wall() - t_start;
    if (a->timings) std::memcpy(a->timings, tm, sizeof(tm));
    return NLS_OK;
  }
  if (rc == NLS_OK) rc = [&]() -> int {
    if (a->finished) *a->finished = 1;
    // gamma* C + A = c Q^-H (gamma* + Lam) Q^-1 is positive definite iff gamma* + lam_min > 0: the test the reference's cho_factor makes
    // (_neo_ls_svm.py:177 raises LinAlgError otherwise).  It is made here, from the eigenvalues, because the factorisation itself runs only
    // when the caller asks for L_ (and then beside everything else, on a side stream).
    if (!(gamma_opt + lam_min > 0.0))
      return fail(ctx, NLS_ERR_LINALG, "gamma* C + A is not positive definite at gamma* = %g (smallest eigenvalue of A / c: %g)", gamma_opt, lam_min);

    // ---- column of the selected gamma (P7 outputs, P9 sigma) -----------------------------------
    NLSCHK(fault_point(ctx, "select"));
    NLSCHK(ws_get_t(ctx, "out.loo_res", (size_t)st.n_pad, &loo_res));
    NLSCHK(ws_get_t(ctx, "out.loo_lev", (size_t)st.n_pad, &loo_lev));
    NLSCHK(ws_get_t(ctx, "out.loo_std", (size_t)st.n_pad, &loo_std));
    NLSCHK(ws_get_t(ctx, "out.res", (size_t)st.n_pad, &res));
    const long cblk = (n + 255) / 256;
    NLSCHK(ws_get_t(ctx, "loo.cpart", (size_t)cblk * 2, &cpart));
    NLSCHK(ws_get_t(ctx, "loo.csum", 4, &csum));
    const double ybar = st.sy_sum / st.s_sum;
    {
      SpanGuard g(ctx, NLS_T_LOO);
      hipLaunchKernelGGL(k_loo_column, dim3((unsigned)cblk), dim3(256), 0, ctx->stream, num, hs, st.dy, st.s_norm, n, Gp, opt, is_clf,
                         ybar, loo_res, loo_lev, loo_std, res, cpart);
      hipLaunchKernelGGL(k_sum_partials, dim3(1), dim3(256), 0, ctx->stream, cpart, cblk, 2L, csum, 0);
      HIPCHK(ctx, hipGetLastError());
    }
    return NLS_OK;
  }();
  NLSCHK(comm_vote(ctx, rc, "before the all-reduce of the score sums"));
  {
    SpanGuard g(ctx, NLS_T_ALLREDUCE);
    NLSCHK(do_allreduce(ctx, csum, 2));
  }
  double hsum[2] = {0.0, 0.0};

  // ---- P8: re-solve at gamma* -------------------------------------------------------------------
  // The reference re-solves with the Cholesky factor at gamma* "for better accuracy" (_neo_ls_svm.py:176-178).  When the caller asks for L_
  // this does the same: zpotrf and the 16 (D+1)^2-byte download run on a side stream beside the outputs of the main stream, and
  // beta = cho_solve(L_, b) follows on that stream (two triangular solves against the factor that was just downloaded), so that the returned
  // pair satisfies beta == cho_solve(L_, b) to rounding.  With a->L == NULL (ranks > 0 of a sharded fit, sigma-grid probes) there is no
  // factorisation: beta = Q (v / (gamma* + lam)) from the eigendecomposition - the same (gamma* C + A)^-1 b, A + gamma c I = c Q (Lam + gamma) Q^H -
  // and positive definiteness was checked on the eigenvalues above.  In a collective fit rank 0's beta is broadcast: identical everywhere.
  double2* dbeta = nullptr;
  rocblas_int* dinfo2 = nullptr;
  bool side = false;
  struct SideJoin {  // an early (error) return must not leave the side streams writing into the caller's L
    hipStream_t s = nullptr, s2 = nullptr;
    ~SideJoin() {
      if (s) (void)hipStreamSynchronize(s);
      if (s2) (void)hipStreamSynchronize(s2);
    }
  } side_join;
  bool side_copy = false, y_carried = false;
  double2* ysolve = nullptr;  // beta = cho_solve(L_, b): L y = b is carried through the factorisation, L^H beta = y follows the download
  // residuals_ = Re(phi beta) - y is defined on the RETURNED beta (_neo_ls_svm.py:178-182).  When that is the Cholesky re-solve (the caller asked
  // for L_; a sharded fit: rank 0's beta is everybody's) one more pass over the feature planes computes it from that beta, below; otherwise the
  // returned beta IS the eigendecomposition's and the sweep's column (k_loo_column) is its residual vector already.
  const bool res_from_beta = a->residuals && (a->L || multi_rank(ctx)) && !(a->flags & NLS_FIT_RESIDUALS_FROM_SWEEP);
  auto mark = [&](const char* what) {  // NLS_COMM_TRACE=1: where the host is in the last stretch (rank 0 works while the others wait at the vote)
    if (comm_trace() && multi_rank(ctx)) std::fprintf(stderr, "[nls mark] rank %d: %s\n", ctx->rank, what);
  };
  rc = [&]() -> int {
    HIPCHK(ctx, hipMemcpyAsync(hsum, csum, 2 * sizeof(double), hipMemcpyDeviceToHost, ctx->stream));
    NLSCHK(ws_get_t(ctx, "chol.beta", (size_t)D1, &dbeta));
    mark("score sums requested");
    if (a->L) {
      NLSCHK(fault_point(ctx, "cholesky"));
      if (!ctx->stream2) HIPCHK(ctx, hipStreamCreateWithFlags(&ctx->stream2, hipStreamNonBlocking));  // (the dual path's look-ahead may have made it)
      if (!ctx->blas2) {
        if (rocblas_create_handle(&ctx->blas2) != rocblas_status_success) return fail(ctx, NLS_ERR_HIP, "rocblas_create_handle (side stream) failed");
        BLASCHK(ctx, rocblas_set_stream(ctx->blas2, ctx->stream2));
      }
      mark("side stream and rocBLAS handle exist");
      for (auto& e : ctx->side_ev)
        if (!e) HIPCHK(ctx, hipEventCreate(&e));
      NLSCHK(ws_get_t(ctx, "chol.info2", 4, &dinfo2));
      side = true;
      hipStream_t s2 = ctx->stream2;
      side_join.s = s2;
      HIPCHK(ctx, hipEventRecord(ctx->side_ev[0], ctx->stream));  // everything that produced Acm / gamma* is behind this point
      HIPCHK(ctx, hipStreamWaitEvent(s2, ctx->side_ev[0], 0));
      if (general_C) {  // gamma* C + A (_neo_ls_svm.py:177)
        const long tot = (long)D1 * D1;
        hipLaunchKernelGGL(k_axpy_z, dim3((unsigned)((tot + 255) / 256)), dim3(256), 0, s2, Cn, gamma_opt, tot, Acm);
      } else {
        hipLaunchKernelGGL(k_add_diag, dim3((unsigned)((D1 + 255) / 256)), dim3(256), 0, s2, Acm, (long)D1, D1, gamma_opt * st.c);
      }
      HIPCHK(ctx, hipGetLastError());
      // Own blocked right-looking factorisation (nls_zpotrf.h: panels of 64, three launches per panel).  A block column of 512 is final once
      // its last panel's trailing update has run, and travels to the host on the copy stream - conjugated there: the column-major lower factor L
      // (A = L L^H) is, byte for byte, the conjugate of the row-major upper factor U = L^H that scipy's cho_factor(lower=False) returns - while
      // the following block columns are factored.  (Round 3: rocsolver_zpotrf on 512-wide diagonal blocks + rocBLAS ztrsm / zherk, 18 ms at
      // D + 1 = 4097; a monolithic rocsolver_zpotrf 16.5 ms + 12.7 ms of download in sequence.)
      constexpr int NBK = 512;
      const int nblk = (D1 + NBK - 1) / NBK;
      NLSCHK(ensure_copy_stream(ctx, nblk));
      side_join.s2 = ctx->copy_stream;
      NLSCHK(ws_get_t(ctx, "chol.solve", (size_t)D1, &ysolve));
      mark("copy streams exist");
      NLSCHK(zpotrf_lower(ctx, s2, ctx->blas2, Acm, D1, (long)D1, reinterpret_cast<int*>(dinfo2), NBK, db, ysolve, &y_carried));
      HIPCHK(ctx, hipEventRecord(ctx->side_ev[1], s2));
      side_copy = true;
      mark("factorisation enqueued");
    }
    {
      SpanGuard g(ctx, NLS_T_RESIDUALS);
      const int nchunks = (D1 + 255) / 256;
      double2* bpart = nullptr;
      NLSCHK(ws_get_t(ctx, "chol.bpart", (size_t)nchunks * D1, &bpart));
      hipLaunchKernelGGL(k_beta_evd_partial, dim3((unsigned)((D1 + 63) / 64), (unsigned)nchunks), dim3(256), 0, ctx->stream, Qev_keep, (long)D1, D1, rb.vr, rb.vi,
                         lam, gamma_opt, bpart);
      hipLaunchKernelGGL(k_beta_evd_finish, dim3((unsigned)((D1 + 255) / 256)), dim3(256), 0, ctx->stream, bpart, nchunks, D1, dbeta);
      HIPCHK(ctx, hipGetLastError());
    }

    // (k_loo_column has left Re(phi beta_evd(gamma*)) - y in `res`: the sweep's table holds Re(phi beta(gamma)) on the whole grid)

    // ---- outputs -------------------------------------------------------------------------------
    {
      SpanGuard g(ctx, NLS_T_DOWNLOAD);
      auto d2h = [&](void* dst, const void* src, size_t bytes) -> int {
        if (dst) HIPCHK(ctx, hipMemcpyAsync(dst, src, bytes, hipMemcpyDeviceToHost, ctx->stream));
        return NLS_OK;
      };
      NLSCHK(d2h(a->lam, lam, sizeof(double) * D1));
      NLSCHK(d2h(a->loo_residuals, loo_res, sizeof(double) * n));
      NLSCHK(d2h(a->loo_leverage, loo_lev, sizeof(double) * n));
      NLSCHK(d2h(a->loo_std, loo_std, sizeof(double) * n));
      if (!res_from_beta) NLSCHK(d2h(a->residuals, res, sizeof(double) * n));
    }
    mark("row outputs requested");
    if (side_copy) {  // everything else is enqueued: the block columns of L_ now follow the factorisation (pageable memory: each copy blocks this thread)
      prefault.join();
      NLSCHK(download_block_columns(ctx, a->L, Acm, D1, (long)D1, sizeof(double2), 512, true));
      HIPCHK(ctx, hipEventRecord(ctx->side_ev[2], ctx->copy_stream));
      mark("factor downloaded");
    }
    if (side_copy) {
      // beta = cho_solve(L_, b) (_neo_ls_svm.py:178).  The copy stream has conjugated the factor in place on its way out (Acm now holds
      // Lc = conj(L), i.e. scipy's upper factor read column-major), so: L x = b <=> Lc conj(x) = conj(b);  L^H beta = x <=> Lc^T beta = x.
      hipStream_t s2 = ctx->stream2;
      double2* tmp = ysolve;
      HIPCHK(ctx, hipStreamWaitEvent(s2, ctx->side_ev[2], 0));
      if (y_carried) {  // y = L^-1 b came out of the factorisation: L^H beta = y against the conjugated factor
        NLSCHK(zpotrf_solve_conj_tail(ctx, s2, Acm, D1, (long)D1, tmp));
      } else {  // (NLS_POTRF=rocsolver)
        const auto* zL = reinterpret_cast<const rocblas_double_complex*>(Acm);
        hipLaunchKernelGGL(k_conj_vec, dim3((unsigned)((D1 + 255) / 256)), dim3(256), 0, s2, db, D1, tmp);
        HIPCHK(ctx, hipGetLastError());
        BLASCHK(ctx, rocblas_ztrsv(ctx->blas2, rocblas_fill_lower, rocblas_operation_none, rocblas_diagonal_non_unit, D1, zL, D1,
                                   reinterpret_cast<rocblas_double_complex*>(tmp), 1));
        hipLaunchKernelGGL(k_conj_vec, dim3((unsigned)((D1 + 255) / 256)), dim3(256), 0, s2, tmp, D1, tmp);
        HIPCHK(ctx, hipGetLastError());
        BLASCHK(ctx, rocblas_ztrsv(ctx->blas2, rocblas_fill_lower, rocblas_operation_transpose, rocblas_diagonal_non_unit, D1, zL, D1,
                                   reinterpret_cast<rocblas_double_complex*>(tmp), 1));
      }
      HIPCHK(ctx, hipEventRecord(ctx->side_ev[3], s2));
      HIPCHK(ctx, hipStreamWaitEvent(ctx->stream, ctx->side_ev[3], 0));
      HIPCHK(ctx, hipMemcpyAsync(dbeta, tmp, sizeof(double2) * D1, hipMemcpyDeviceToDevice, ctx->stream));
    }
    if (side && multi_rank(ctx)) {
      // sharded: rank 0's factorisation is judged BEFORE its beta is broadcast - a non-positive pivot goes into the vote below and every rank
      // raises the same LinAlgError (one rank: the check after the outputs, as before, so that the side streams keep running beside them)
      rocblas_int info2 = 0;
      HIPCHK(ctx, hipMemcpyAsync(&info2, dinfo2, sizeof(info2), hipMemcpyDeviceToHost, ctx->stream2));
      HIPCHK(ctx, hipStreamSynchronize(ctx->stream2));
      if (info2 != 0) return fail(ctx, NLS_ERR_LINALG, "Cholesky factorisation of gamma* C + A: pivot %d is not positive (matrix not positive definite)", (int)info2);
      mark("factorisation judged");
    }
    return NLS_OK;
  }();
  if (multi_rank(ctx)) {
    NLSCHK(comm_vote(ctx, rc, "before the broadcast of rank 0's beta"));
    SpanGuard g(ctx, NLS_T_ALLREDUCE);
    NLSCHK(do_broadcast(ctx, reinterpret_cast<double*>(dbeta), (size_t)2 * D1, 0));
  } else {
    NLSCHK(rc);
  }
  if (a->beta) HIPCHK(ctx, hipMemcpyAsync(a->beta, dbeta, sizeof(double2) * D1, hipMemcpyDeviceToHost, ctx->stream));
  if (res_from_beta) {  // (no exchange follows: a failure here simply returns)
    SpanGuard g(ctx, NLS_T_RESIDUALS);
    double *br = nullptr, *bi = nullptr;
    NLSCHK(ws_get_t(ctx, "chol.br", (size_t)Kf, &br));
    NLSCHK(ws_get_t(ctx, "chol.bi", (size_t)Kf, &bi));
    hipLaunchKernelGGL(k_split_vec, dim3((unsigned)((Kf + 255) / 256)), dim3(256), 0, ctx->stream, dbeta, mp.D, Kf, br, bi);
    HIPCHK(ctx, hipGetLastError());
    // resident planes: one launch over all rows (the planes carry the row scale rs: undone by inv_rs); else the feature map again, chunk by chunk
    for (long r0 = 0; r0 < n; r0 += st.resident ? n : st.rc) {
      const long rows = st.resident ? n : std::min<long>(st.rc, n - r0);
      if (!st.resident)  // (its time stays in the residuals stage: spans do not nest)
        NLSCHK(launch_featuremap_planes(ctx, mp, st.dX + r0 * mp.d, rows, round_up(rows, BM), st.rs + r0, st.Fc, st.Fs));
      hipLaunchKernelGGL(k_plane_gemv, dim3((unsigned)((rows + 3) / 4)), dim3(256), 0, ctx->stream, planes_c(st, r0), planes_s(st, r0), Kf, br, bi, dbeta, mp.D,
                         rows, st.dy + r0, is_clf, res + r0, st.inv_rs + r0);
      HIPCHK(ctx, hipGetLastError());
    }
    HIPCHK(ctx, hipMemcpyAsync(a->residuals, res, sizeof(double) * n, hipMemcpyDeviceToHost, ctx->stream));
  }
  if (side) {  // join the side streams; their stage times go into the cholesky / download slots
    rocblas_int info2 = 0;
    HIPCHK(ctx, hipMemcpyAsync(&info2, dinfo2, sizeof(info2), hipMemcpyDeviceToHost, ctx->stream2));
    HIPCHK(ctx, hipStreamSynchronize(ctx->stream2));
    HIPCHK(ctx, hipStreamSynchronize(ctx->copy_stream));
    if (info2 != 0) return fail(ctx, NLS_ERR_LINALG, "Cholesky factorisation of gamma* C + A: pivot %d is not positive (matrix not positive definite)", (int)info2);
    float ms = 0.f;
    if (hipEventElapsedTime(&ms, ctx->side_ev[0], ctx->side_ev[1]) == hipSuccess) tm[NLS_T_CHOLESKY] += ms * 1e-3;
    if (hipEventElapsedTime(&ms, ctx->side_ev[1], ctx->side_ev[2]) == hipSuccess) tm[NLS_T_DOWNLOAD] += ms * 1e-3;
    if (hipEventElapsedTime(&ms, ctx->side_ev[2], ctx->side_ev[3]) == hipSuccess) tm[NLS_T_CHOLESKY] += ms * 1e-3;
  }
  NLSCHK(spans_collect(ctx, tm));
  if (a->loo_errors) std::memcpy(a->loo_errors, herrs.data(), sizeof(double) * G);
  if (a->objective) std::memcpy(a->objective, hobj.data(), sizeof(double) * G);
  if (a->gamma_index) *a->gamma_index = opt;
  if (a->loo_score) {
    // accuracy_score / r2_score with sample_weight = s (_neo_ls_svm.py:171-174); s_norm sums to 1.
    *a->loo_score = is_clf ? hsum[0] : 1.0 - hsum[0] / hsum[1];
  }
  tm[NLS_T_TOTAL] = wall() - t_start;
  if (host_marks) std::fprintf(stderr, "[nls host] nls_primal_fit leaves at %.3f ms\n", 1e3 * std::fmod(wall(), 1000.0));
  if (a->timings) std::memcpy(a->timings, tm, sizeof(tm));
  return NLS_OK;
}

// ------------------------------------------------------------------------------------------------
// Primal inference (P10, P11)
// ------------------------------------------------------------------------------------------------
// W = phi U^-1, sigma^2 = sum_j |W_ij|^2.  The row-major upper U read as column-major is the lower triangular U^T;
// ztrtri(lower) inverts it in place, which read back row-major is U^-1 (upper): the B operand of the rotation kernel.
static int build_inverse_factor_planes(nls_ctx* ctx, const MapParams& mp, const double* L, const RotBuffers& rb) {
  const int D1 = mp.D1;
  double2* Urm = nullptr;
  rocblas_int* dinfo = nullptr;
  NLSCHK(ws_get_t(ctx, "evd.A", (size_t)D1 * D1, &Urm));
  NLSCHK(ws_get_t(ctx, "evd.info", 4, &dinfo));
  HIPCHK(ctx, hipMemcpyAsync(Urm, L, sizeof(double2) * (size_t)D1 * D1, is_device_ptr(L) ? hipMemcpyDeviceToDevice : hipMemcpyHostToDevice,
                             ctx->stream));
  BLASCHK(ctx, rocblas_set_stream(ctx->blas, ctx->stream));
  BLASCHK(ctx, rocsolver_ztrtri(ctx->blas, rocblas_fill_lower, rocblas_diagonal_non_unit, D1, reinterpret_cast<rocblas_double_complex*>(Urm),
                                D1, dinfo));
  NLSCHK(check_info(ctx, dinfo, "rocsolver_ztrtri"));
  NLSCHK(build_rot_planes(ctx, mp, Urm, (long)D1, 1L, true, rb));
  return NLS_OK;
}

static void map_sizes(int D, MapParams* mp) {
  mp->D = D;
  mp->D1 = D + 1;
  mp->Kf = (int)round_up(D, BN);
  mp->Np = (int)round_up(D + 1, m3::BN3);
}

extern "C" int nls_factor_create(nls_ctx* ctx, const double* L, int D, nls_factor** out) {
  if (!ctx) return NLS_ERR_ARG;
  if (!L || !out || D < 1) return fail(ctx, NLS_ERR_ARG, "nls_factor_create: L / factor NULL or D < 1");
  HIPCHK(ctx, hipSetDevice(ctx->device));
  MapParams mp;
  map_sizes(D, &mp);
  nls_factor* f = new nls_factor();
  f->owner = ctx;
  f->D = D;
  auto alloc = [&](double** p, size_t count) { return hipMalloc(reinterpret_cast<void**>(p), count * sizeof(double)); };
  hipError_t e = alloc(&f->Mr, (size_t)mp.Kf * mp.Np);
  if (e == hipSuccess) e = alloc(&f->Mi, (size_t)mp.Kf * mp.Np);
  if (e == hipSuccess) e = alloc(&f->mbr, (size_t)mp.Np);
  if (e == hipSuccess) e = alloc(&f->mbi, (size_t)mp.Np);
  if (e == hipSuccess) e = alloc(&f->zero, (size_t)mp.Np);
  if (e == hipSuccess) e = hipMemsetAsync(f->zero, 0, sizeof(double) * mp.Np, ctx->stream);
  int rc = NLS_OK;
  if (e != hipSuccess) {
    rc = fail(ctx, NLS_ERR_HIP, "nls_factor_create: device allocation failed: %s", hipGetErrorString(e));
  } else {
    RotBuffers rb;
    rb.Mr = f->Mr;
    rb.Mi = f->Mi;
    rb.mbr = f->mbr;
    rb.mbi = f->mbi;
    rc = build_inverse_factor_planes(ctx, mp, L, rb);
    if (rc == NLS_OK && hipStreamSynchronize(ctx->stream) != hipSuccess) rc = fail(ctx, NLS_ERR_HIP, "nls_factor_create: stream failed");
  }
  if (rc != NLS_OK) {
    factor_free(f);
    return rc;
  }
  ctx->factors.push_back(f);
  *out = f;
  return NLS_OK;
}

extern "C" int nls_factor_destroy(nls_ctx* ctx, nls_factor* f) {
  if (!ctx) return NLS_ERR_ARG;
  if (!f) return NLS_OK;
  auto it = std::find(ctx->factors.begin(), ctx->factors.end(), f);
  if (it == ctx->factors.end()) return fail(ctx, NLS_ERR_ARG, "nls_factor_destroy: not a live factor of this context");
  HIPCHK(ctx, hipSetDevice(ctx->device));
  HIPCHK(ctx, hipStreamSynchronize(ctx->stream));
  ctx->factors.erase(it);
  factor_free(f);
  return NLS_OK;
}

extern "C" int nls_primal_predict(nls_ctx* ctx, const double* X, int64_t m, int d, const double* shift, const double* scale,
                                  const double* B, int D, const double* beta, const double* L, const nls_factor* factor, double* yhat,
                                  double* sigma) {
  if (!ctx) return NLS_ERR_ARG;
  if (!X || m < 0) return fail(ctx, NLS_ERR_ARG, "X NULL or m < 0");
  if (yhat && !beta) return fail(ctx, NLS_ERR_ARG, "beta is required for yhat");
  if (sigma && !L && !factor) return fail(ctx, NLS_ERR_ARG, "L or a factor handle is required for sigma");
  if (sigma && factor) {
    if (std::find(ctx->factors.begin(), ctx->factors.end(), factor) == ctx->factors.end())
      return fail(ctx, NLS_ERR_ARG, "factor is not a live handle of this context");
    if (factor->D != D) return fail(ctx, NLS_ERR_ARG, "factor was created for D = %d, called with D = %d", factor->D, D);
  }
  HIPCHK(ctx, hipSetDevice(ctx->device));
  if (m == 0 || (!yhat && !sigma)) return NLS_OK;
  MapParams mp;
  NLSCHK(upload_map(ctx, shift, scale, B, d, D, &mp));
  const int D1 = mp.D1, Kf = mp.Kf, Np = mp.Np;
  const double* dX = nullptr;
  NLSCHK(resident(ctx, "in.X", X, (size_t)m * d, &dX));
  if (yhat && !sigma) {
    // decision_function alone: feature map and weight product fused (k_featuremap_gemv), phi never reaches HBM.
    const long m_pad_all = round_up(m, BM);
    const long rcf = std::min<long>(m_pad_all, 1L << 20);
    const int nparts = 2 * (Kf / BN);
    double *dy = nullptr, *part = nullptr, *wr = nullptr, *wi = nullptr, *Xs = nullptr;
    double2* dbeta = nullptr;
    NLSCHK(ws_get_t(ctx, "out.res", (size_t)m_pad_all, &dy));
    NLSCHK(ws_get_t(ctx, "pred.part", (size_t)nparts * rcf, &part));
    NLSCHK(ws_get_t(ctx, "chol.beta", (size_t)D1, &dbeta));
    NLSCHK(ws_get_t(ctx, "chol.br", (size_t)Kf, &wr));
    NLSCHK(ws_get_t(ctx, "chol.bi", (size_t)Kf, &wi));
    NLSCHK(ws_get_t(ctx, "fm.Xs", (size_t)rcf * mp.dk, &Xs));
    HIPCHK(ctx, hipMemcpyAsync(dbeta, beta, sizeof(double2) * D1, hipMemcpyHostToDevice, ctx->stream));
    hipLaunchKernelGGL(k_gemv_weights, dim3((unsigned)((Kf + 255) / 256)), dim3(256), 0, ctx->stream, dbeta, D, Kf, 1.0 / std::sqrt((double)D),
                       wr, wi);
    HIPCHK(ctx, hipGetLastError());
    const double bias = beta[2 * (size_t)D];  // Re beta[D]: the bias feature phi[:, D] = 1
    for (long r0 = 0; r0 < m; r0 += rcf) {
      const long rows = std::min<long>(rcf, m - r0), rows_pad = round_up(rows, BM);
      hipLaunchKernelGGL(k_shift_pad, dim3((unsigned)((rows_pad * mp.dk + 255) / 256)), dim3(256), 0, ctx->stream, dX + r0 * d, mp.shift,
                         rows, mp.d, rows_pad, mp.dk, Xs);
      FeatureMapParams p{};
      p.X = dX + r0 * d;
      p.Xs = Xs;
      p.shift = mp.shift;
      p.Bs = mp.Bs;
      p.rows = rows;
      p.d = mp.d;
      p.dk = mp.dk;
      p.D = mp.D;
      p.Kf = mp.Kf;
      p.inv_sqrt_D = 1.0 / std::sqrt((double)D);
      p.sc = sincos_coef();
      p.tc = sincos_tab_coef();
      p.sintab = ctx->k1_table ? ctx->sintab : nullptr;
      if (p.sintab)
        hipLaunchKernelGGL(k_featuremap_gemv<true>, dim3((unsigned)(Kf / BN), (unsigned)(rows_pad / BM)), dim3(Cfg4::NTHREADS), SMEM_REAL,
                           ctx->stream, p, wr, wi, rows_pad, part);
      else
        hipLaunchKernelGGL(k_featuremap_gemv<false>, dim3((unsigned)(Kf / BN), (unsigned)(rows_pad / BM)), dim3(Cfg4::NTHREADS), SMEM_REAL,
                           ctx->stream, p, wr, wi, rows_pad, part);
      hipLaunchKernelGGL(k_gemv_finish, dim3((unsigned)((rows + 255) / 256)), dim3(256), 0, ctx->stream, part, nparts, rows_pad, rows, bias,
                         dy + r0);
      HIPCHK(ctx, hipGetLastError());
    }
    HIPCHK(ctx, hipMemcpyAsync(yhat, dy, sizeof(double) * m, hipMemcpyDeviceToHost, ctx->stream));
    HIPCHK(ctx, hipStreamSynchronize(ctx->stream));
    return NLS_OK;
  }
  const long rc = pick_row_chunk(ctx, m, mp, 4ull * D1 * D1 * 16);
  double *Fc = nullptr, *Fs = nullptr, *br = nullptr, *bi = nullptr, *dy = nullptr, *dsig = nullptr;
  NLSCHK(ws_get_t(ctx, "chunk.Fc", (size_t)rc * Kf, &Fc));
  NLSCHK(ws_get_t(ctx, "chunk.Fs", (size_t)rc * Kf, &Fs));
  const long m_pad = round_up(m, BM);
  NLSCHK(ws_get_t(ctx, "out.res", (size_t)m_pad, &dy));
  NLSCHK(ws_get_t(ctx, "out.loo_std", (size_t)m_pad, &dsig));
  double2* dbeta = nullptr;
  if (yhat) {
    NLSCHK(ws_get_t(ctx, "chol.beta", (size_t)D1, &dbeta));
    NLSCHK(ws_get_t(ctx, "chol.br", (size_t)Kf, &br));
    NLSCHK(ws_get_t(ctx, "chol.bi", (size_t)Kf, &bi));
    HIPCHK(ctx, hipMemcpyAsync(dbeta, beta, sizeof(double2) * D1, hipMemcpyHostToDevice, ctx->stream));
    hipLaunchKernelGGL(k_split_vec, dim3((unsigned)((Kf + 255) / 256)), dim3(256), 0, ctx->stream, dbeta, D, Kf, br, bi);
    HIPCHK(ctx, hipGetLastError());
  }
  RotBuffers rb;
  double *U = nullptr, *Gm = nullptr;
  if (sigma) {
    NLSCHK(ws_get_t(ctx, "chunk.U", (size_t)rc * Np, &U));
    NLSCHK(ws_get_t(ctx, "chunk.Gm", (size_t)rc * Np, &Gm));
    if (factor) {
      rb.Mr = factor->Mr;
      rb.Mi = factor->Mi;
      rb.mbr = factor->mbr;
      rb.mbi = factor->mbi;
      rb.vr = rb.vi = factor->zero;
    } else {
      NLSCHK(rot_buffers(ctx, mp, &rb));
      NLSCHK(build_inverse_factor_planes(ctx, mp, L, rb));
      HIPCHK(ctx, hipMemsetAsync(rb.vr, 0, sizeof(double) * Np, ctx->stream));
      HIPCHK(ctx, hipMemsetAsync(rb.vi, 0, sizeof(double) * Np, ctx->stream));
    }
  }
  for (long r0 = 0; r0 < m; r0 += rc) {
    const long rows = std::min<long>(rc, m - r0);
    const long rows_pad = round_up(rows, BM);
    NLSCHK(launch_featuremap_planes(ctx, mp, dX + r0 * d, rows, rows_pad, nullptr, Fc, Fs));
    if (yhat) {
      hipLaunchKernelGGL(k_plane_gemv, dim3((unsigned)((rows + 3) / 4)), dim3(256), 0, ctx->stream, Fc, Fs, Kf, br, bi, dbeta, D, rows,
                         (const double*)nullptr, 0, dy + r0, (const double*)nullptr);
      HIPCHK(ctx, hipGetLastError());
    }
    if (sigma) {
      NLSCHK(launch_rotate(ctx, mp, Fc, Fs, rb.Mr, rb.Mi, rb.mbr, rb.mbi, rb.vr, rb.vi, U, Gm, nullptr, rows_pad));
      hipLaunchKernelGGL(k_rowsum_sqrt, dim3((unsigned)((rows + 3) / 4)), dim3(256), 0, ctx->stream, Gm, Np, rows, dsig + r0);
      HIPCHK(ctx, hipGetLastError());
    }
  }
  if (yhat) HIPCHK(ctx, hipMemcpyAsync(yhat, dy, sizeof(double) * m, hipMemcpyDeviceToHost, ctx->stream));
  if (sigma) HIPCHK(ctx, hipMemcpyAsync(sigma, dsig, sizeof(double) * m, hipMemcpyDeviceToHost, ctx->stream));
  HIPCHK(ctx, hipStreamSynchronize(ctx->stream));
  return NLS_OK;
}
