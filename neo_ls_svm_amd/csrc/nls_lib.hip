// Host side of libneolssvm_hip.so: context, workspace, stage orchestration and the C ABI of
// include/neolssvm_hip.h.  Device code lives in nls_gemm.h / nls_kernels.h / nls_dual.h.
#include "nls_host.h"
#include "nls_kernels.h"
#include "nls_dual_kernels.h"

using namespace nls;

std::string g_create_error;

// ------------------------------------------------------------------------------------------------
// Shared stage helpers
// ------------------------------------------------------------------------------------------------
struct MapParams {  // device-resident parameters of the affine + ORF map
  int d = 0, dk = 0, D = 0, D1 = 0, Kp = 0, Np = 0;
  double* shift = nullptr;  // d
  double* Bs = nullptr;     // dk x Kp
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
  mp->Kp = (int)round_up(D + 2, BN);
  mp->Np = (int)round_up(D + 1, BN);
  double *dB = nullptr, *dscale = nullptr;
  NLSCHK(ws_get_t(ctx, "map.shift", (size_t)d, &mp->shift));
  NLSCHK(ws_get_t(ctx, "map.scale", (size_t)d, &dscale));
  NLSCHK(ws_get_t(ctx, "map.B", (size_t)d * D, &dB));
  NLSCHK(ws_get_t(ctx, "map.Bs", (size_t)mp->dk * mp->Kp, &mp->Bs));
  HIPCHK(ctx, hipMemcpyAsync(mp->shift, shift, sizeof(double) * d, hipMemcpyHostToDevice, ctx->stream));
  HIPCHK(ctx, hipMemcpyAsync(dscale, scale, sizeof(double) * d, hipMemcpyHostToDevice, ctx->stream));
  HIPCHK(ctx, hipMemcpyAsync(dB, B, sizeof(double) * (size_t)d * D, hipMemcpyHostToDevice, ctx->stream));
  const long tot = (long)mp->dk * mp->Kp;
  hipLaunchKernelGGL(k_build_Bs, dim3((unsigned)((tot + 255) / 256)), dim3(256), 0, ctx->stream, dB, dscale, d, D,
                     mp->dk, mp->Kp, mp->Bs);
  HIPCHK(ctx, hipGetLastError());
  return NLS_OK;
}

static long rot_grid(nls_ctx* ctx, long tiles_r, long tiles_c) {
  return ctx->rot_pr > 0 ? xcd_patch_grid(tiles_r, tiles_c, ctx->rot_pr, ctx->rot_pc) : tiles_r * tiles_c;
}

constexpr size_t SMEM_REAL = 2 * 2 * TILE_DOUBLES * sizeof(double);  // double-buffered A, B
constexpr size_t SMEM_CPLX = 2 * 4 * TILE_DOUBLES * sizeof(double);  // double-buffered Ac, As, Br, Bi (144 KiB)

// K1 into split planes for `rows` rows starting at Xchunk.
static int launch_featuremap_planes(nls_ctx* ctx, const MapParams& mp, const double* Xchunk, long rows, long rows_pad,
                                    const double* rowscale, const double* target, double* Fc, double* Fs) {
  FeatureMapParams p;
  p.X = Xchunk;
  p.shift = mp.shift;
  p.Bs = mp.Bs;
  p.rowscale = rowscale;
  p.target = target;
  p.rows = rows;
  p.d = mp.d;
  p.dk = mp.dk;
  p.D = mp.D;
  p.Kp = mp.Kp;
  p.inv_sqrt_D = 1.0 / std::sqrt((double)mp.D);
  p.Fc = Fc;
  p.Fs = Fs;
  p.phi = nullptr;
  dim3 grid((unsigned)(mp.Kp / BN), (unsigned)(rows_pad / BM));
  hipLaunchKernelGGL(k_featuremap<false>, grid, dim3(Cfg4::NTHREADS), SMEM_REAL, ctx->stream, p);
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
  return pick_row_chunk_bytes(ctx, n, 16ull * ((size_t)mp.Kp + (size_t)mp.Np), fixed_bytes);
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
  if (const char* e4 = std::getenv("NLS_COMPLEX_4M")) ctx->use_4m = e4[0] == '1';
  if (const char* ep = std::getenv("NLS_ROT_PATCH")) std::sscanf(ep, "%dx%d", &ctx->rot_pr, &ctx->rot_pc);
  if (const char* er = std::getenv("NLS_NO_RESIDENT_PLANES")) ctx->no_resident = er[0] == '1';
  if (const char* eg = std::getenv("NLS_GRAM_PATCH")) ctx->gram_patches = eg[0] == '1';
  ctx->cus = prop.multiProcessorCount;
  if ((e = hipStreamCreateWithFlags(&ctx->stream, hipStreamNonBlocking)) != hipSuccess)
    return bail("hipStreamCreate", hipGetErrorString(e));
  if (rocblas_create_handle(&ctx->blas) != rocblas_status_success) return bail("rocblas_create_handle", "status != success");
  rocblas_set_stream(ctx->blas, ctx->stream);
  // Opt in to > 64 KiB of dynamic LDS for the complex tile kernels.
  (void)hipFuncSetAttribute(reinterpret_cast<const void*>(k_gram), hipFuncAttributeMaxDynamicSharedMemorySize, (int)SMEM_CPLX);
  (void)hipFuncSetAttribute(reinterpret_cast<const void*>(k_rotate), hipFuncAttributeMaxDynamicSharedMemorySize, (int)SMEM_CPLX);
  (void)hipFuncSetAttribute(reinterpret_cast<const void*>(k_gram3), hipFuncAttributeMaxDynamicSharedMemorySize, (int)m3::SMEM3);
  (void)hipFuncSetAttribute(reinterpret_cast<const void*>(k_rotate3), hipFuncAttributeMaxDynamicSharedMemorySize, (int)m3::SMEM3);
  (void)hipFuncSetAttribute(reinterpret_cast<const void*>(k_sweep), hipFuncAttributeMaxDynamicSharedMemorySize, (int)SMEM_REAL);
  (void)hipFuncSetAttribute(reinterpret_cast<const void*>(k_featuremap<false>), hipFuncAttributeMaxDynamicSharedMemorySize, (int)SMEM_REAL);
  (void)hipFuncSetAttribute(reinterpret_cast<const void*>(k_featuremap<true>), hipFuncAttributeMaxDynamicSharedMemorySize, (int)SMEM_REAL);
  (void)hipFuncSetAttribute(reinterpret_cast<const void*>(k_gemm<EPI_STORE>), hipFuncAttributeMaxDynamicSharedMemorySize, (int)SMEM_REAL);
  (void)hipFuncSetAttribute(reinterpret_cast<const void*>(k_gemm<EPI_RBF>), hipFuncAttributeMaxDynamicSharedMemorySize, (int)SMEM_REAL);
  *out = ctx;
  return NLS_OK;
}

extern "C" void nls_ctx_destroy(nls_ctx* ctx) {
  if (!ctx) return;
  (void)hipSetDevice(ctx->device);
  (void)hipStreamSynchronize(ctx->stream);
  for (auto& kv : ctx->ws)
    if (kv.second.p) (void)hipFree(kv.second.p);
  for (auto ev : ctx->event_pool) (void)hipEventDestroy(ev);
  if (ctx->blas) rocblas_destroy_handle(ctx->blas);
  if (ctx->stream) (void)hipStreamDestroy(ctx->stream);
  delete ctx;
}

extern "C" const char* nls_last_error(const nls_ctx* ctx) { return ctx ? ctx->err.c_str() : g_create_error.c_str(); }

extern "C" int nls_set_allreduce(nls_ctx* ctx, nls_allreduce_fn fn, void* user, int rank, int world) {
  if (!ctx) return NLS_ERR_ARG;
  if (world < 1 || rank < 0 || rank >= world) return fail(ctx, NLS_ERR_ARG, "bad rank/world %d/%d", rank, world);
  ctx->allreduce = (world > 1) ? fn : nullptr;
  ctx->allreduce_user = user;
  ctx->rank = rank;
  ctx->world = (fn && world > 1) ? world : 1;
  return NLS_OK;
}

extern "C" int nls_set_workspace_limit(nls_ctx* ctx, size_t bytes) {
  if (!ctx) return NLS_ERR_ARG;
  ctx->ws_limit = bytes;
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
  HIPCHK(ctx, hipMemcpy(dst, src, bytes, hipMemcpyHostToDevice));
  return NLS_OK;
}
extern "C" int nls_memcpy_d2h(nls_ctx* ctx, void* dst, const void* src, size_t bytes) {
  if (!ctx) return NLS_ERR_ARG;
  HIPCHK(ctx, hipMemcpy(dst, src, bytes, hipMemcpyDeviceToHost));
  return NLS_OK;
}
extern "C" int nls_synchronize(nls_ctx* ctx) {
  if (!ctx) return NLS_ERR_ARG;
  HIPCHK(ctx, hipSetDevice(ctx->device));
  HIPCHK(ctx, hipDeviceSynchronize());
  return NLS_OK;
}
extern "C" int nls_device_info(nls_ctx* ctx, char* name, int name_len, int* compute_units, size_t* hbm_bytes) {
  if (!ctx) return NLS_ERR_ARG;
  hipDeviceProp_t prop;
  HIPCHK(ctx, hipGetDeviceProperties(&prop, ctx->device));
  if (name && name_len > 0) {
    std::snprintf(name, (size_t)name_len, "%s (%s)", prop.name, prop.gcnArchName);
  }
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
    FeatureMapParams p;
    p.X = dX + r0 * d;
    p.shift = mp.shift;
    p.Bs = mp.Bs;
    p.rowscale = nullptr;
    p.target = nullptr;
    p.rows = rows;
    p.d = mp.d;
    p.dk = mp.dk;
    p.D = mp.D;
    p.Kp = mp.Kp;
    p.inv_sqrt_D = 1.0 / std::sqrt((double)D);
    p.Fc = p.Fs = nullptr;
    p.phi = out_dev ? phi + 2 * r0 * mp.D1 : dphi;
    dim3 grid((unsigned)(mp.Np / BN), (unsigned)(round_up(rows, BM) / BM));
    hipLaunchKernelGGL(k_featuremap<true>, grid, dim3(Cfg4::NTHREADS), SMEM_REAL, ctx->stream, p);
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
  double n_total = 0, s_sum = 0, sy_sum = 0;
  double c = 0;  // 1 / (n_total * D1): the reference's normalised complexity diagonal (_neo_ls_svm.py:117-118)
  const double *dX = nullptr, *dy = nullptr, *ds = nullptr;
  double* s_norm = nullptr;  // n_pad
  double* rs = nullptr;      // n_pad  row scale of the planes: s_norm, or 2^-500 where s == 0 (0 beyond n)
  double* inv_rs = nullptr;  // n_pad  1 / rs (0 beyond n)
  bool resident = false;     // feature planes of ALL rows stay in HBM (one K1 pass per fit)
  long plane_rows = 0;       // rows of the plane buffers (rc, or nchunks * rc when resident)
  double* sy = nullptr;      // n_pad  s_norm * y
  double *Fc = nullptr, *Fs = nullptr;  // rc x Kp
  int nt = 0, ntri = 0;
  double* gacc = nullptr;  // ntri x 2 x 128 x 128 tile-packed extended Gram
};

static inline double* planes_c(const PrimalState& st, long r0) { return st.Fc + (st.resident ? r0 * st.mp.Kp : 0); }
static inline double* planes_s(const PrimalState& st, long r0) { return st.Fs + (st.resident ? r0 * st.mp.Kp : 0); }

// Primal fit: keep the feature planes of all rows resident when they fit next to everything else; then only the
// rotation outputs U, Gm are chunked.
static void plan_primal_chunks(nls_ctx* ctx, PrimalState& st, size_t fixed_bytes) {
  const size_t limit = ctx->ws_limit ? ctx->ws_limit : (size_t)(0.6 * (double)ctx->hbm_bytes);
  const size_t planes_all = 16ull * (size_t)(st.n_pad + BM) * st.mp.Kp;
  const size_t min_rot = 16ull * st.mp.Np * (size_t)std::min<long>(st.n_pad, 32768);
  if (!ctx->no_resident && fixed_bytes + planes_all + min_rot <= limit) {
    st.resident = true;
    st.rc = pick_row_chunk_bytes(ctx, st.n, 16ull * st.mp.Np, fixed_bytes + planes_all);
    const long nchunks = (st.n_pad + st.rc - 1) / st.rc;
    st.plane_rows = nchunks * st.rc;
  } else {
    st.resident = false;
    st.rc = pick_row_chunk(ctx, st.n, st.mp, fixed_bytes);
    st.plane_rows = st.rc;
  }
}

// Normalise the weights by the global sum (_neo_ls_svm.py:110) and set c.
static int primal_prepare(nls_ctx* ctx, PrimalState& st, const double* X, const double* y, const double* s, long n, int d,
                          double* timings) {
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
    hipLaunchKernelGGL(k_sum_partials, dim3(1), dim3(64), 0, ctx->stream, part, nblk, 2L, sums);
    HIPCHK(ctx, hipGetLastError());
  }
  double hn = (double)n;
  HIPCHK(ctx, hipMemcpyAsync(sums + 2, &hn, sizeof(double), hipMemcpyHostToDevice, ctx->stream));
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
  NLSCHK(ws_get_t(ctx, "pre.sy", (size_t)st.n_pad, &st.sy));
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
  (void)timings;
  return NLS_OK;
}

// Phase A: extended Gram of F = s o [phi, y] accumulated over row chunks, all-reduced across ranks.
static int primal_gram_phase(nls_ctx* ctx, PrimalState& st, double* timings) {
  const MapParams& mp = st.mp;
  st.nt = mp.Kp / BM;
  st.ntri = st.nt * (st.nt + 1) / 2;
  const size_t tile_elems_total = (size_t)st.ntri * 2 * BM * BN;
  NLSCHK(ws_get_t(ctx, "gram.acc", tile_elems_total, &st.gacc));
  HIPCHK(ctx, hipMemsetAsync(st.gacc, 0, tile_elems_total * sizeof(double), ctx->stream));
  if (st.plane_rows < st.rc) st.plane_rows = st.rc;
  NLSCHK(ws_get_t(ctx, "chunk.Fc", (size_t)st.plane_rows * mp.Kp, &st.Fc));
  NLSCHK(ws_get_t(ctx, "chunk.Fs", (size_t)st.plane_rows * mp.Kp, &st.Fs));
  // Row split so that one launch fills the chip several times over.
  const long target_blocks = 8L * ctx->cus;
  long nsplit = std::max<long>(1, (target_blocks + st.ntri - 1) / st.ntri);
  nsplit = std::min<long>(nsplit, std::max<long>(1, st.rc / (4 * BK)));
  const long rows_per_split = round_up((st.rc + nsplit - 1) / nsplit, BK);
  nsplit = (st.rc + rows_per_split - 1) / rows_per_split;
  double* slab = nullptr;
  NLSCHK(ws_get_t(ctx, "gram.slab", (size_t)nsplit * tile_elems_total, &slab));
  for (long r0 = 0; r0 < st.n; r0 += st.rc) {
    const long rows = std::min<long>(st.rc, st.n - r0);
    const long rows_pad = round_up(rows, BM);
    {
      SpanGuard g(ctx, NLS_T_FEATUREMAP);
      NLSCHK(launch_featuremap_planes(ctx, mp, st.dX + r0 * mp.d, rows, rows_pad, st.rs + r0, st.dy + r0, planes_c(st, r0), planes_s(st, r0)));
      if (timings) {
        timings[NLS_T_FEATUREMAP_LAUNCHES] += 1;
        timings[NLS_T_FEATUREMAP_FLOPS] += 2.0 * rows * mp.d * mp.D;
      }
    }
    {
      SpanGuard g(ctx, NLS_T_GRAM);
      const long rps = round_up((rows_pad + nsplit - 1) / nsplit, BK);
      const long ns = (rows_pad + rps - 1) / rps;
      if (ctx->use_4m)
        hipLaunchKernelGGL(k_gram, dim3((unsigned)(st.ntri * ns)), dim3(Cfg8::NTHREADS), SMEM_CPLX, ctx->stream, planes_c(st, r0),
                           planes_s(st, r0), mp.Kp, rows_pad, st.ntri, rps, slab);
      else {
        const long bps = ctx->gram_patches ? xcd_patch_grid(st.nt, 2L * st.nt, 4, 8) : 2L * st.ntri;
        hipLaunchKernelGGL(k_gram3, dim3((unsigned)(bps * ns)), dim3(m3::NT3), m3::SMEM3, ctx->stream, planes_c(st, r0), planes_s(st, r0),
                           mp.Kp, rows_pad, st.ntri, rps, slab, st.nt, bps);
      }
      HIPCHK(ctx, hipGetLastError());
      hipLaunchKernelGGL(k_gram_reduce, dim3((unsigned)((tile_elems_total + 255) / 256)), dim3(256), 0, ctx->stream, slab,
                         (int)ns, (long)tile_elems_total, st.gacc);
      HIPCHK(ctx, hipGetLastError());
      if (timings) {
        timings[NLS_T_GRAM_LAUNCHES] += 1;
        timings[NLS_T_GRAM_FLOPS] += 4.0 * rows * (double)mp.D1 * mp.D1;
      }
    }
  }
  {
    SpanGuard g(ctx, NLS_T_ALLREDUCE);
    NLSCHK(do_allreduce(ctx, st.gacc, tile_elems_total));
  }
  return NLS_OK;
}

extern "C" int nls_gram_only(nls_ctx* ctx, const double* X, const double* y, const double* s, int64_t n, int d,
                             const double* shift, const double* scale, const double* B, int D, double* A, double* b) {
  if (!ctx) return NLS_ERR_ARG;
  if (!X || !y || !s || n < 0) return fail(ctx, NLS_ERR_ARG, "X/y/s NULL or n < 0");
  HIPCHK(ctx, hipSetDevice(ctx->device));
  PrimalState st;
  NLSCHK(upload_map(ctx, shift, scale, B, d, D, &st.mp));
  NLSCHK(primal_prepare(ctx, st, X, y, s, n, d, nullptr));
  st.rc = pick_row_chunk(ctx, n, st.mp, 0);
  st.plane_rows = st.rc;
  NLSCHK(primal_gram_phase(ctx, st, nullptr));
  const int D1 = st.mp.D1;
  double2 *Acm = nullptr, *Arm = nullptr, *db = nullptr;
  NLSCHK(ws_get_t(ctx, "evd.A", (size_t)D1 * D1, &Acm));
  NLSCHK(ws_get_t(ctx, "evd.Q", (size_t)D1 * D1, &Arm));
  NLSCHK(ws_get_t(ctx, "evd.b", (size_t)D1, &db));
  dim3 grid((unsigned)((D1 + 255) / 256), (unsigned)(D1 + 1));
  hipLaunchKernelGGL(k_assemble_A, grid, dim3(256), 0, ctx->stream, st.gacc, D1, 1.0, Acm, (long)D1, db);
  hipLaunchKernelGGL(k_cm_to_rm, dim3((unsigned)((D1 + 255) / 256), (unsigned)D1), dim3(256), 0, ctx->stream, Acm, (long)D1, D1,
                     false, Arm);
  HIPCHK(ctx, hipGetLastError());
  if (A) HIPCHK(ctx, hipMemcpyAsync(A, Arm, sizeof(double2) * (size_t)D1 * D1, hipMemcpyDeviceToHost, ctx->stream));
  if (b) HIPCHK(ctx, hipMemcpyAsync(b, db, sizeof(double2) * (size_t)D1, hipMemcpyDeviceToHost, ctx->stream));
  NLSCHK(spans_collect(ctx, nullptr));
  return NLS_OK;
}


// Row-major complex (D1 x D1, host layout) -> B-operand planes [Kp x Np], zero padded.
__global__ void k_build_planes_rm(const double2* Qrm, int D1, int Kp, int Np, double* Qr, double* Qi) {
  const int k = blockIdx.x * blockDim.x + threadIdx.x;
  const int i = blockIdx.y;
  if (k >= Np) return;
  double2 v = make_double2(0.0, 0.0);
  if (i < D1 && k < D1) v = Qrm[(long)i * D1 + k];
  Qr[(long)i * Np + k] = v.x;
  Qi[(long)i * Np + k] = v.y;
}
// out[i][j] = in[i][j] for j < cols (ld_in -> cols contiguous)
__global__ void k_compact_rows(const double* in, long ld_in, long rows, int cols, double* out) {
  const int j = blockIdx.x * blockDim.x + threadIdx.x;
  const long i = blockIdx.y;
  if (j < cols && i < rows) out[i * cols + j] = in[i * ld_in + j];
}

extern "C" int nls_rotate_only(nls_ctx* ctx, const double* X, int64_t n, int d, const double* shift, const double* scale,
                               const double* B, int D, const double* Q, const double* v, double* Uout, double* Gmout) {
  if (!ctx) return NLS_ERR_ARG;
  if (!X || !Q || !v || n < 1) return fail(ctx, NLS_ERR_ARG, "X/Q/v NULL or n < 1");
  HIPCHK(ctx, hipSetDevice(ctx->device));
  MapParams mp;
  NLSCHK(upload_map(ctx, shift, scale, B, d, D, &mp));
  const int D1 = mp.D1, Kp = mp.Kp, Np = mp.Np;
  const double* dX = nullptr;
  NLSCHK(resident(ctx, "in.X", X, (size_t)n * d, &dX));
  const long rc = pick_row_chunk(ctx, n, mp, 4ull * D1 * D1 * 16);
  double *Fc = nullptr, *Fs = nullptr, *Qr = nullptr, *Qi = nullptr, *vr = nullptr, *vi = nullptr, *U = nullptr, *Gm = nullptr, *cmp = nullptr;
  double2 *dQ = nullptr, *dv = nullptr;
  NLSCHK(ws_get_t(ctx, "chunk.Fc", (size_t)rc * Kp, &Fc));
  NLSCHK(ws_get_t(ctx, "chunk.Fs", (size_t)rc * Kp, &Fs));
  NLSCHK(ws_get_t(ctx, "evd.Q", (size_t)D1 * D1, &dQ));
  NLSCHK(ws_get_t(ctx, "chol.beta", (size_t)D1, &dv));
  NLSCHK(ws_get_t(ctx, "rot.Qr", (size_t)Kp * Np, &Qr));
  NLSCHK(ws_get_t(ctx, "rot.Qi", (size_t)Kp * Np, &Qi));
  NLSCHK(ws_get_t(ctx, "rot.vr", (size_t)Np, &vr));
  NLSCHK(ws_get_t(ctx, "rot.vi", (size_t)Np, &vi));
  NLSCHK(ws_get_t(ctx, "chunk.U", (size_t)rc * Np, &U));
  NLSCHK(ws_get_t(ctx, "chunk.Gm", (size_t)rc * Np, &Gm));
  if (Uout || Gmout) NLSCHK(ws_get_t(ctx, "rot.compact", (size_t)rc * D1, &cmp));
  HIPCHK(ctx, hipMemcpyAsync(dQ, Q, sizeof(double2) * (size_t)D1 * D1, hipMemcpyHostToDevice, ctx->stream));
  HIPCHK(ctx, hipMemcpyAsync(dv, v, sizeof(double2) * D1, hipMemcpyHostToDevice, ctx->stream));
  hipLaunchKernelGGL(k_build_planes_rm, dim3((unsigned)((Np + 255) / 256), (unsigned)Kp), dim3(256), 0, ctx->stream, dQ, D1, Kp, Np, Qr, Qi);
  hipLaunchKernelGGL(k_split_vec, dim3((unsigned)((Np + 255) / 256)), dim3(256), 0, ctx->stream, dv, D1, Np, vr, vi);
  HIPCHK(ctx, hipGetLastError());
  for (long r0 = 0; r0 < n; r0 += rc) {
    const long rows = std::min<long>(rc, n - r0);
    const long rows_pad = round_up(rows, BM);
    NLSCHK(launch_featuremap_planes(ctx, mp, dX + r0 * d, rows, rows_pad, nullptr, nullptr, Fc, Fs));
    if (ctx->use_4m)
      hipLaunchKernelGGL(k_rotate, dim3((unsigned)(Np / BN), (unsigned)(rows_pad / BM)), dim3(Cfg8::NTHREADS), SMEM_CPLX, ctx->stream, Fc, Fs,
                         Kp, Qr, Qi, Np, vr, vi, U, Gm, (const double*)nullptr);
    else
      hipLaunchKernelGGL(k_rotate3, dim3((unsigned)rot_grid(ctx, rows_pad / BM, Np / m3::BN3)), dim3(m3::NT3), m3::SMEM3, ctx->stream, Fc,
                         Fs, Kp, Qr, Qi, Np, vr, vi, U, Gm, (const double*)nullptr, rows_pad / BM, ctx->rot_pr, ctx->rot_pc);
    HIPCHK(ctx, hipGetLastError());
    for (int which = 0; which < 2; ++which) {
      double* dst = which ? Gmout : Uout;
      if (!dst) continue;
      hipLaunchKernelGGL(k_compact_rows, dim3((unsigned)((D1 + 255) / 256), (unsigned)rows), dim3(256), 0, ctx->stream, which ? Gm : U, (long)Np,
                         rows, D1, cmp);
      HIPCHK(ctx, hipGetLastError());
      HIPCHK(ctx, hipMemcpyAsync(dst + r0 * D1, cmp, sizeof(double) * (size_t)rows * D1, hipMemcpyDeviceToHost, ctx->stream));
      HIPCHK(ctx, hipStreamSynchronize(ctx->stream));
    }
  }
  HIPCHK(ctx, hipStreamSynchronize(ctx->stream));
  return NLS_OK;
}

extern "C" int nls_primal_fit(nls_ctx* ctx, const nls_primal_fit_args* a) {
  if (!ctx) return NLS_ERR_ARG;
  if (!a) return fail(ctx, NLS_ERR_ARG, "args is NULL");
  if (!a->X || !a->y || !a->s || !a->gammas) return fail(ctx, NLS_ERR_ARG, "X, y, s and gammas must not be NULL");
  if (a->n < 1 || a->G < 1) return fail(ctx, NLS_ERR_ARG, "n and G must be >= 1 (n=%ld, G=%d)", (long)a->n, a->G);
  if (a->gamma_index_in >= a->G) return fail(ctx, NLS_ERR_ARG, "gamma_index_in out of range");
  HIPCHK(ctx, hipSetDevice(ctx->device));
  const double t_start = wall();
  double tm[NLS_NUM_TIMINGS];
  std::memset(tm, 0, sizeof(tm));
  ctx->spans.clear();
  ctx->events_used = 0;

  PrimalState st;
  const long n = a->n;
  const int G = a->G, is_clf = a->is_classifier ? 1 : 0;
  NLSCHK(upload_map(ctx, a->shift, a->scale, a->B, a->d, a->D, &st.mp));
  const MapParams& mp = st.mp;
  const int D1 = mp.D1, Kp = mp.Kp, Np = mp.Np;
  const int Gp = (int)round_up(G, BN);
  NLSCHK(primal_prepare(ctx, st, a->X, a->y, a->s, n, a->d, tm));
  const size_t fixed = 2ull * st.n_pad * Gp * 8 + 6ull * D1 * D1 * 16 + 2ull * Kp * Np * 8 + (size_t)Np * Gp * 8;
  plan_primal_chunks(ctx, st, fixed);
  tm[NLS_T_ROW_CHUNK] = (double)st.rc;

  // ---- phase A: Gram -------------------------------------------------------------------------
  NLSCHK(primal_gram_phase(ctx, st, tm));

  // ---- phase B: EVD of A / c (P4) --------------------------------------------------------------
  double2 *Acm = nullptr, *Qcm = nullptr, *db = nullptr;
  double *lam = nullptr, *evd_e = nullptr, *Qr = nullptr, *Qi = nullptr, *vr = nullptr, *vi = nullptr, *dgam = nullptr, *R = nullptr;
  rocblas_int* dinfo = nullptr;
  NLSCHK(ws_get_t(ctx, "evd.A", (size_t)D1 * D1, &Acm));
  NLSCHK(ws_get_t(ctx, "evd.Q", (size_t)D1 * D1, &Qcm));
  NLSCHK(ws_get_t(ctx, "evd.b", (size_t)D1, &db));
  NLSCHK(ws_get_t(ctx, "evd.lam", (size_t)D1, &lam));
  NLSCHK(ws_get_t(ctx, "evd.e", (size_t)D1, &evd_e));
  NLSCHK(ws_get_t(ctx, "evd.info", 4, &dinfo));
  NLSCHK(ws_get_t(ctx, "rot.Qr", (size_t)Kp * Np, &Qr));
  NLSCHK(ws_get_t(ctx, "rot.Qi", (size_t)Kp * Np, &Qi));
  NLSCHK(ws_get_t(ctx, "rot.vr", (size_t)Np, &vr));
  NLSCHK(ws_get_t(ctx, "rot.vi", (size_t)Np, &vi));
  NLSCHK(ws_get_t(ctx, "sweep.gammas", (size_t)G, &dgam));
  NLSCHK(ws_get_t(ctx, "sweep.R", (size_t)Np * Gp, &R));
  HIPCHK(ctx, hipMemcpyAsync(dgam, a->gammas, sizeof(double) * G, hipMemcpyHostToDevice, ctx->stream));
  {
    SpanGuard g(ctx, NLS_T_EVD);
    dim3 grid((unsigned)((D1 + 255) / 256), (unsigned)(D1 + 1));
    hipLaunchKernelGGL(k_assemble_A, grid, dim3(256), 0, ctx->stream, st.gacc, D1, 1.0, Acm, (long)D1, db);
    hipLaunchKernelGGL(k_assemble_A, grid, dim3(256), 0, ctx->stream, st.gacc, D1, 1.0 / st.c, Qcm, (long)D1, (double2*)nullptr);
    HIPCHK(ctx, hipGetLastError());
    BLASCHK(ctx, rocsolver_zheevd(ctx->blas, rocblas_evect_original, rocblas_fill_lower, D1,
                                  reinterpret_cast<rocblas_double_complex*>(Qcm), D1, lam, evd_e, dinfo));
    NLSCHK(check_info(ctx, dinfo, "rocsolver_zheevd"));
    hipLaunchKernelGGL(k_build_Q_planes, dim3((unsigned)((Np + 255) / 256), (unsigned)Kp), dim3(256), 0, ctx->stream, Qcm, (long)D1,
                       D1, Kp, Np, Qr, Qi);
    hipLaunchKernelGGL(k_compute_v, dim3((unsigned)Np), dim3(256), 0, ctx->stream, Qcm, (long)D1, db, D1, 1.0 / st.c, Np, vr, vi);
    const long tot = (long)Np * Gp;
    hipLaunchKernelGGL(k_rgrid, dim3((unsigned)((tot + 255) / 256)), dim3(256), 0, ctx->stream, lam, dgam, D1, G, Np, Gp, R);
    HIPCHK(ctx, hipGetLastError());
  }

  // ---- phase C: rotation + sweep per row chunk (P5, P6) ---------------------------------------
  double *U = nullptr, *Gm = nullptr, *num = nullptr, *hs = nullptr;
  NLSCHK(ws_get_t(ctx, "chunk.U", (size_t)st.rc * Np, &U));
  NLSCHK(ws_get_t(ctx, "chunk.Gm", (size_t)st.rc * Np, &Gm));
  NLSCHK(ws_get_t(ctx, "sweep.num", (size_t)st.n_pad * Gp, &num));
  NLSCHK(ws_get_t(ctx, "sweep.hs", (size_t)st.n_pad * Gp, &hs));
  for (long r0 = 0; r0 < n; r0 += st.rc) {
    const long rows = std::min<long>(st.rc, n - r0);
    const long rows_pad = round_up(rows, BM);
    if (!st.resident) {
      SpanGuard g(ctx, NLS_T_FEATUREMAP);
      NLSCHK(launch_featuremap_planes(ctx, mp, st.dX + r0 * mp.d, rows, rows_pad, st.rs + r0, st.dy + r0, st.Fc, st.Fs));
      tm[NLS_T_FEATUREMAP_LAUNCHES] += 1;
      tm[NLS_T_FEATUREMAP_FLOPS] += 2.0 * rows * mp.d * mp.D;
    }
    {
      SpanGuard g(ctx, NLS_T_ROTATE);
      if (ctx->use_4m)
        hipLaunchKernelGGL(k_rotate, dim3((unsigned)(Np / BN), (unsigned)(rows_pad / BM)), dim3(Cfg8::NTHREADS), SMEM_CPLX, ctx->stream,
                           planes_c(st, r0), planes_s(st, r0), Kp, Qr, Qi, Np, vr, vi, U, Gm, st.inv_rs + r0);
      else
        hipLaunchKernelGGL(k_rotate3, dim3((unsigned)rot_grid(ctx, rows_pad / BM, Np / m3::BN3)), dim3(m3::NT3), m3::SMEM3,
                           ctx->stream, planes_c(st, r0), planes_s(st, r0), Kp, Qr, Qi, Np, vr, vi, U, Gm, st.inv_rs + r0, rows_pad / BM, ctx->rot_pr, ctx->rot_pc);
      HIPCHK(ctx, hipGetLastError());
      tm[NLS_T_ROTATE_LAUNCHES] += 1;
      tm[NLS_T_ROTATE_FLOPS] += 8.0 * rows * (double)D1 * D1;
    }
    {
      SpanGuard g(ctx, NLS_T_SWEEP);
      hipLaunchKernelGGL(k_sweep, dim3((unsigned)(Gp / BN), (unsigned)(rows_pad / BM), 2), dim3(Cfg4::NTHREADS), SMEM_REAL, ctx->stream, U, Gm,
                         Np, R, Gp, 1.0 / st.c, num, hs, r0);
      HIPCHK(ctx, hipGetLastError());
      tm[NLS_T_SWEEP_LAUNCHES] += 1;
      tm[NLS_T_SWEEP_FLOPS] += 4.0 * rows * (double)D1 * G;
    }
  }

  // ---- P7: per-gamma errors, selection ---------------------------------------------------------
  const long nblk = (n + LOO_ROWS_PER_BLOCK - 1) / LOO_ROWS_PER_BLOCK;
  double *part = nullptr, *errs = nullptr;
  NLSCHK(ws_get_t(ctx, "loo.part", (size_t)nblk * 3 * Gp, &part));
  NLSCHK(ws_get_t(ctx, "loo.errs", (size_t)3 * Gp, &errs));
  {
    SpanGuard g(ctx, NLS_T_LOO);
    hipLaunchKernelGGL(k_loo_errors, dim3((unsigned)nblk), dim3(256), 0, ctx->stream, num, hs, st.dy, st.s_norm, n, G, Gp, is_clf,
                       part);
    hipLaunchKernelGGL(k_sum_partials, dim3((unsigned)((3 * Gp + 255) / 256)), dim3(256), 0, ctx->stream, part, nblk, 3L * Gp, errs);
    HIPCHK(ctx, hipGetLastError());
  }
  {
    SpanGuard g(ctx, NLS_T_ALLREDUCE);
    NLSCHK(do_allreduce(ctx, errs, (size_t)3 * Gp));
  }
  std::vector<double> herrs((size_t)3 * Gp), hobj((size_t)G);
  HIPCHK(ctx, hipMemcpyAsync(herrs.data(), errs, sizeof(double) * 3 * Gp, hipMemcpyDeviceToHost, ctx->stream));
  HIPCHK(ctx, hipStreamSynchronize(ctx->stream));
  for (int g = 0; g < G; ++g)  // _neo_ls_svm.py:159-165 (same summation order as the reference)
    hobj[g] = is_clf ? (herrs[Gp + g] + herrs[2 * Gp + g]) + herrs[g] : herrs[g];
  int opt = a->gamma_index_in;
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
  const double gamma_opt = a->gammas[opt];

  // ---- column of the selected gamma (P7 outputs, P9 sigma) -------------------------------------
  double *loo_res = nullptr, *loo_lev = nullptr, *loo_std = nullptr, *res = nullptr, *cpart = nullptr, *csum = nullptr;
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
                       ybar, loo_res, loo_lev, loo_std, cpart);
    hipLaunchKernelGGL(k_sum_partials, dim3(1), dim3(64), 0, ctx->stream, cpart, cblk, 2L, csum);
    HIPCHK(ctx, hipGetLastError());
  }
  {
    SpanGuard g(ctx, NLS_T_ALLREDUCE);
    NLSCHK(do_allreduce(ctx, csum, 2));
  }
  double hsum[2];
  HIPCHK(ctx, hipMemcpyAsync(hsum, csum, 2 * sizeof(double), hipMemcpyDeviceToHost, ctx->stream));

  // ---- P8: Cholesky re-solve at gamma* ---------------------------------------------------------
  double2* dbeta = nullptr;
  double *br = nullptr, *bi = nullptr;
  NLSCHK(ws_get_t(ctx, "chol.beta", (size_t)D1, &dbeta));
  NLSCHK(ws_get_t(ctx, "chol.br", (size_t)Kp, &br));
  NLSCHK(ws_get_t(ctx, "chol.bi", (size_t)Kp, &bi));
  {
    SpanGuard g(ctx, NLS_T_CHOLESKY);
    hipLaunchKernelGGL(k_add_diag, dim3((unsigned)((D1 + 255) / 256)), dim3(256), 0, ctx->stream, Acm, (long)D1, D1, gamma_opt * st.c);
    HIPCHK(ctx, hipGetLastError());
    BLASCHK(ctx, rocsolver_zpotrf(ctx->blas, rocblas_fill_lower, D1, reinterpret_cast<rocblas_double_complex*>(Acm), D1, dinfo));
    NLSCHK(check_info(ctx, dinfo, "rocsolver_zpotrf"));
    HIPCHK(ctx, hipMemcpyAsync(dbeta, db, sizeof(double2) * D1, hipMemcpyDeviceToDevice, ctx->stream));
    BLASCHK(ctx, rocsolver_zpotrs(ctx->blas, rocblas_fill_lower, D1, 1, reinterpret_cast<rocblas_double_complex*>(Acm), D1,
                                  reinterpret_cast<rocblas_double_complex*>(dbeta), D1));
    hipLaunchKernelGGL(k_split_vec, dim3((unsigned)((Kp + 255) / 256)), dim3(256), 0, ctx->stream, dbeta, D1, Kp, br, bi);
    HIPCHK(ctx, hipGetLastError());
  }

  // ---- residuals_ = Re(phi beta) - y (P8) -------------------------------------------------------
  for (long r0 = 0; r0 < n; r0 += st.rc) {
    const long rows = std::min<long>(st.rc, n - r0);
    const long rows_pad = round_up(rows, BM);
    if (!st.resident) {
      SpanGuard g(ctx, NLS_T_FEATUREMAP);
      NLSCHK(launch_featuremap_planes(ctx, mp, st.dX + r0 * mp.d, rows, rows_pad, st.rs + r0, st.dy + r0, st.Fc, st.Fs));
      tm[NLS_T_FEATUREMAP_LAUNCHES] += 1;
      tm[NLS_T_FEATUREMAP_FLOPS] += 2.0 * rows * mp.d * mp.D;
    }
    {
      SpanGuard g(ctx, NLS_T_RESIDUALS);
      hipLaunchKernelGGL(k_plane_gemv, dim3((unsigned)((rows + 3) / 4)), dim3(256), 0, ctx->stream, planes_c(st, r0), planes_s(st, r0), Kp, br, bi, rows,
                         st.dy + r0, is_clf, res + r0, st.inv_rs + r0);
      HIPCHK(ctx, hipGetLastError());
    }
  }

  // ---- outputs ---------------------------------------------------------------------------------
  {
    SpanGuard g(ctx, NLS_T_DOWNLOAD);
    auto d2h = [&](void* dst, const void* src, size_t bytes) -> int {
      if (dst) HIPCHK(ctx, hipMemcpyAsync(dst, src, bytes, hipMemcpyDeviceToHost, ctx->stream));
      return NLS_OK;
    };
    NLSCHK(d2h(a->beta, dbeta, sizeof(double2) * D1));
    NLSCHK(d2h(a->lam, lam, sizeof(double) * D1));
    NLSCHK(d2h(a->loo_residuals, loo_res, sizeof(double) * n));
    NLSCHK(d2h(a->loo_leverage, loo_lev, sizeof(double) * n));
    NLSCHK(d2h(a->loo_std, loo_std, sizeof(double) * n));
    NLSCHK(d2h(a->residuals, res, sizeof(double) * n));
    if (a->L) {
      // Column-major lower factor L (A = L L^H) is, byte for byte, the conjugate of the row-major
      // upper factor U = L^H that scipy's cho_factor(lower=False) returns.
      const long tot = (long)D1 * D1;
      hipLaunchKernelGGL(k_conj_inplace, dim3((unsigned)((tot + 255) / 256)), dim3(256), 0, ctx->stream, Acm, tot);
      HIPCHK(ctx, hipGetLastError());
      NLSCHK(d2h(a->L, Acm, sizeof(double2) * (size_t)D1 * D1));
    }
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
  if (a->timings) std::memcpy(a->timings, tm, sizeof(tm));
  return NLS_OK;
}

// ------------------------------------------------------------------------------------------------
// Primal inference (P10, P11)
// ------------------------------------------------------------------------------------------------
// Row-major upper-triangular complex matrix (other triangle ignored) -> planes [Kp x Np].
__global__ void k_build_upper_planes(const double2* Urm, int D1, int Kp, int Np, double* Qr, double* Qi) {
  const int k = blockIdx.x * blockDim.x + threadIdx.x;
  const int i = blockIdx.y;
  if (k >= Np) return;
  double2 v = make_double2(0.0, 0.0);
  if (i < D1 && k < D1 && i <= k) v = Urm[(long)i * D1 + k];
  Qr[(long)i * Np + k] = v.x;
  Qi[(long)i * Np + k] = v.y;
}

extern "C" int nls_primal_predict(nls_ctx* ctx, const double* X, int64_t m, int d, const double* shift, const double* scale,
                                  const double* B, int D, const double* beta, const double* L, double* yhat, double* sigma) {
  if (!ctx) return NLS_ERR_ARG;
  if (!X || m < 0) return fail(ctx, NLS_ERR_ARG, "X NULL or m < 0");
  if (yhat && !beta) return fail(ctx, NLS_ERR_ARG, "beta is required for yhat");
  if (sigma && !L) return fail(ctx, NLS_ERR_ARG, "L is required for sigma");
  HIPCHK(ctx, hipSetDevice(ctx->device));
  if (m == 0 || (!yhat && !sigma)) return NLS_OK;
  MapParams mp;
  NLSCHK(upload_map(ctx, shift, scale, B, d, D, &mp));
  const int D1 = mp.D1, Kp = mp.Kp, Np = mp.Np;
  const double* dX = nullptr;
  NLSCHK(resident(ctx, "in.X", X, (size_t)m * d, &dX));
  const long rc = pick_row_chunk(ctx, m, mp, 4ull * D1 * D1 * 16);
  double *Fc = nullptr, *Fs = nullptr, *br = nullptr, *bi = nullptr, *dy = nullptr, *dsig = nullptr;
  NLSCHK(ws_get_t(ctx, "chunk.Fc", (size_t)rc * Kp, &Fc));
  NLSCHK(ws_get_t(ctx, "chunk.Fs", (size_t)rc * Kp, &Fs));
  const long m_pad = round_up(m, BM);
  NLSCHK(ws_get_t(ctx, "out.res", (size_t)m_pad, &dy));
  NLSCHK(ws_get_t(ctx, "out.loo_std", (size_t)m_pad, &dsig));
  double2* dbeta = nullptr;
  if (yhat) {
    NLSCHK(ws_get_t(ctx, "chol.beta", (size_t)D1, &dbeta));
    NLSCHK(ws_get_t(ctx, "chol.br", (size_t)Kp, &br));
    NLSCHK(ws_get_t(ctx, "chol.bi", (size_t)Kp, &bi));
    HIPCHK(ctx, hipMemcpyAsync(dbeta, beta, sizeof(double2) * D1, hipMemcpyHostToDevice, ctx->stream));
    hipLaunchKernelGGL(k_split_vec, dim3((unsigned)((Kp + 255) / 256)), dim3(256), 0, ctx->stream, dbeta, D1, Kp, br, bi);
    HIPCHK(ctx, hipGetLastError());
  }
  double *Qr = nullptr, *Qi = nullptr, *vr = nullptr, *vi = nullptr, *U = nullptr, *Gm = nullptr;
  if (sigma) {
    // W = phi U^-1, sigma^2 = sum_j |W_ij|^2.  The row-major upper U read as column-major is the lower
    // triangular U^T; ztrtri(lower) inverts it in place, which read back row-major is U^-1 (upper).
    double2* Urm = nullptr;
    rocblas_int* dinfo = nullptr;
    NLSCHK(ws_get_t(ctx, "evd.A", (size_t)D1 * D1, &Urm));
    NLSCHK(ws_get_t(ctx, "evd.info", 4, &dinfo));
    NLSCHK(ws_get_t(ctx, "rot.Qr", (size_t)Kp * Np, &Qr));
    NLSCHK(ws_get_t(ctx, "rot.Qi", (size_t)Kp * Np, &Qi));
    NLSCHK(ws_get_t(ctx, "rot.vr", (size_t)Np, &vr));
    NLSCHK(ws_get_t(ctx, "rot.vi", (size_t)Np, &vi));
    NLSCHK(ws_get_t(ctx, "chunk.U", (size_t)rc * Np, &U));
    NLSCHK(ws_get_t(ctx, "chunk.Gm", (size_t)rc * Np, &Gm));
    HIPCHK(ctx, hipMemcpyAsync(Urm, L, sizeof(double2) * (size_t)D1 * D1, hipMemcpyHostToDevice, ctx->stream));
    BLASCHK(ctx, rocsolver_ztrtri(ctx->blas, rocblas_fill_lower, rocblas_diagonal_non_unit, D1,
                                  reinterpret_cast<rocblas_double_complex*>(Urm), D1, dinfo));
    NLSCHK(check_info(ctx, dinfo, "rocsolver_ztrtri"));
    hipLaunchKernelGGL(k_build_upper_planes, dim3((unsigned)((Np + 255) / 256), (unsigned)Kp), dim3(256), 0, ctx->stream, Urm, D1, Kp,
                       Np, Qr, Qi);
    HIPCHK(ctx, hipMemsetAsync(vr, 0, sizeof(double) * Np, ctx->stream));
    HIPCHK(ctx, hipMemsetAsync(vi, 0, sizeof(double) * Np, ctx->stream));
    HIPCHK(ctx, hipGetLastError());
  }
  for (long r0 = 0; r0 < m; r0 += rc) {
    const long rows = std::min<long>(rc, m - r0);
    const long rows_pad = round_up(rows, BM);
    NLSCHK(launch_featuremap_planes(ctx, mp, dX + r0 * d, rows, rows_pad, nullptr, nullptr, Fc, Fs));
    if (yhat) {
      hipLaunchKernelGGL(k_plane_gemv, dim3((unsigned)((rows + 3) / 4)), dim3(256), 0, ctx->stream, Fc, Fs, Kp, br, bi, rows,
                         (const double*)nullptr, 0, dy + r0, (const double*)nullptr);
      HIPCHK(ctx, hipGetLastError());
    }
    if (sigma) {
      if (ctx->use_4m)
        hipLaunchKernelGGL(k_rotate, dim3((unsigned)(Np / BN), (unsigned)(rows_pad / BM)), dim3(Cfg8::NTHREADS), SMEM_CPLX, ctx->stream, Fc, Fs,
                           Kp, Qr, Qi, Np, vr, vi, U, Gm, (const double*)nullptr);
      else
        hipLaunchKernelGGL(k_rotate3, dim3((unsigned)rot_grid(ctx, rows_pad / BM, Np / m3::BN3)), dim3(m3::NT3), m3::SMEM3, ctx->stream,
                           Fc, Fs, Kp, Qr, Qi, Np, vr, vi, U, Gm, (const double*)nullptr, rows_pad / BM, ctx->rot_pr, ctx->rot_pc);
      hipLaunchKernelGGL(k_rowsum_sqrt, dim3((unsigned)((rows + 3) / 4)), dim3(256), 0, ctx->stream, Gm, Np, rows, dsig + r0);
      HIPCHK(ctx, hipGetLastError());
    }
  }
  if (yhat) HIPCHK(ctx, hipMemcpyAsync(yhat, dy, sizeof(double) * m, hipMemcpyDeviceToHost, ctx->stream));
  if (sigma) HIPCHK(ctx, hipMemcpyAsync(sigma, dsig, sizeof(double) * m, hipMemcpyDeviceToHost, ctx->stream));
  HIPCHK(ctx, hipStreamSynchronize(ctx->stream));
  return NLS_OK;
}

// ------------------------------------------------------------------------------------------------
// Dual path: see nls_dual.hip (same shared library).
// ------------------------------------------------------------------------------------------------
