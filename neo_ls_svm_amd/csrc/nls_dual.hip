// Dual path of the hot path: n x n RBF kernel, real symmetric EVD, reduced LOO
// gamma sweep (2 n^3 + O(n^2 G) instead of the reference's n x G x n tensor), Cholesky re-solve, sigma.
// Reference: NeoLSSVM._optimize_alpha_gamma, _neo_ls_svm.py:191-325; inference :470-477, :666-671.
#include "nls_dual_kernels.h"
#include "nls_host.h"
#include "nls_trsv.h"
#include "nls_kernels.h"
#include "nls_potrf.h"

using namespace nls;

namespace {

constexpr size_t SMEM_REAL_D = 2 * 2 * TILE_DOUBLES * sizeof(double);

// A = L L^T in place (lower, column-major): nls_potrf.h; NLS_POTRF=rocsolver takes rocsolver_dpotrf instead (diagnostic).  info: device word,
// 0 or the 1-based index of the first non-positive pivot.
// rhs_run (n doubles, device; may be NULL): on entry the right-hand side y of alpha = cho_solve(L, y); on return L^-1 y - the forward substitution
// carried through the factorisation (*carried = true; false on the rocSOLVER path, where the caller solves both halves afterwards).
static int potrf_lower_real(nls_ctx* ctx, double* A, int n, long lda, rocblas_int* dinfo, int event_cols = 0, double* rhs_run = nullptr,
                            bool* carried = nullptr) {
  if (carried) *carried = false;
  using namespace potrf;
  const char* m = std::getenv("NLS_POTRF");
  if (m && std::string(m) == "rocsolver") {
    BLASCHK(ctx, rocsolver_dpotrf(ctx->blas, rocblas_fill_lower, (rocblas_int)n, A, (rocblas_int)lda, dinfo));
    if (event_cols > 0)
      for (int b = 0; b < (n + event_cols - 1) / event_cols; ++b) HIPCHK(ctx, hipEventRecord(ctx->blk_ev[b], ctx->stream));
    return NLS_OK;
  }
  double* Sinv = nullptr;
  NLSCHK(ws_get_t(ctx, "potrf.Sinv", (size_t)NB * SBK, &Sinv));
  HIPCHK(ctx, hipMemsetAsync(dinfo, 0, sizeof(rocblas_int), ctx->stream));
  static_assert(sizeof(rocblas_int) == sizeof(int), "info word");
  if (hipFuncSetAttribute(reinterpret_cast<const void*>(k_potrf_leaf), hipFuncAttributeMaxDynamicSharedMemorySize, (int)LEAF_LDS) != hipSuccess ||
      hipFuncSetAttribute(reinterpret_cast<const void*>(k_potrf_syrk), hipFuncAttributeMaxDynamicSharedMemorySize, (int)SMEM_REAL_D) != hipSuccess)
    return fail(ctx, NLS_ERR_HIP, "Cholesky kernels: %zu / %zu bytes of LDS refused", LEAF_LDS, SMEM_REAL_D);
  if (lda % 2 != 0 || lda < (long)((n + BM - 1) / BM) * BM)
    return fail(ctx, NLS_ERR_ARG, "potrf_lower_real: the leading dimension (%ld) must be n rounded up to %d", lda, BM);
  // Outer blocks of two panels (256 columns): panel a updates only the strip of panel b's columns, and the trailing matrix beyond the outer block
  // is updated once with both panels (K = 256): one pass over the trailing triangle per 256 columns instead of per 128.
  static_assert(NB == BM, "a panel is one tile column");
  double* xsol = nullptr;  // the solved unknowns; the running right-hand side stays in rhs_run until its rows are solved
  if (rhs_run) NLSCHK(ws_get_t(ctx, "potrf.xsol", (size_t)n, &xsol));
  auto panel = [&](int k0) -> int {  // leaf + rows below
    const int w = std::min(NB, n - k0), mrows = n - k0 - w;
    double* D = A + (long)k0 + (long)k0 * lda;
    hipLaunchKernelGGL(k_potrf_leaf, dim3(1), dim3(256), LEAF_LDS, ctx->stream, D, lda, w, k0, Sinv, reinterpret_cast<int*>(dinfo), (const double*)rhs_run, xsol);
    if (mrows > 0)
      hipLaunchKernelGGL(k_potrf_panel, dim3((unsigned)((mrows + 63) / 64)), dim3(256), 0, ctx->stream, D + w, lda, mrows, w, D, Sinv,
                         rhs_run ? (const double*)(xsol + k0) : (const double*)nullptr, rhs_run ? rhs_run + k0 + w : (double*)nullptr);
    HIPCHK(ctx, hipGetLastError());
    return NLS_OK;
  };
  const int NBO = 2 * NB;
  // Look-ahead (round 5): the update of the trailing matrix by an outer block is split - the next panel's 128 columns on the main stream, the rest
  // on a side stream - so that the next diagonal block (ONE workgroup, 110 us) and its panel are factored beside the big update instead of after
  // it.  The regions are disjoint (tile column 0 against tile columns >= 1 of the trailing matrix); the strip update of the panel after that
  // waits for the side stream.  NLS_POTRF_LOOKAHEAD=0: everything on the main stream.
  const char* la_env = std::getenv("NLS_POTRF_LOOKAHEAD");  // (read per call: tests switch it)
  const bool lookahead = !(la_env && la_env[0] == '0');
  hipStream_t side = nullptr;
  if (lookahead && n > 4 * NBO) {
    if (!ctx->stream2) HIPCHK(ctx, hipStreamCreateWithFlags(&ctx->stream2, hipStreamNonBlocking));
    side = ctx->stream2;
    for (auto& e : ctx->la_ev)
      if (!e) HIPCHK(ctx, hipEventCreateWithFlags(&e, hipEventDisableTiming));
  }
  bool side_busy = false;  // the side stream holds an update the main stream has not waited for yet
  auto join_side = [&]() -> int {
    if (side_busy) {
      HIPCHK(ctx, hipEventRecord(ctx->la_ev[1], side));
      HIPCHK(ctx, hipStreamWaitEvent(ctx->stream, ctx->la_ev[1], 0));
      side_busy = false;
    }
    return NLS_OK;
  };
  for (int K0 = 0; K0 < n; K0 += NBO) {
    NLSCHK(panel(K0));  // (tile column 0 of the previous outer block's trailing update: main stream)
    const int ma = n - K0 - NB;  // rows (and columns) below panel a
    bool block_event_on_side = false;
    if (ma > 0) {
      double* Pa = A + (long)(K0 + NB) + (long)K0 * lda;  // panel a's rows below its diagonal block
      const int nta = (ma + BM - 1) / BM;
      NLSCHK(join_side());  // the strip below lies in the side stream's region of the previous outer block
      // the strip of the next 128 columns: tiles (R, 0)
      hipLaunchKernelGGL(k_potrf_syrk, dim3((unsigned)nta), dim3(Cfg4::NTHREADS), SMEM_REAL_D, ctx->stream, A + (long)(K0 + NB) + (long)(K0 + NB) * lda, lda, ma, Pa,
                         lda, NB / BK, 1);
      NLSCHK(panel(K0 + NB));
      const int mb = n - K0 - NBO;  // rows below the outer block
      if (mb > 0) {
        const int ntb = (mb + BM - 1) / BM;
        double* A22 = A + (long)(K0 + NBO) + (long)(K0 + NBO) * lda;
        const double* Lp = A + (long)(K0 + NBO) + (long)K0 * lda;
        if (side && ntb > 2) {
          // tile column 0 (what the next diagonal block and panel need) here, tile columns >= 1 - the lower triangle of the sub-matrix one tile down
          // and right, with the panel rows one tile down - on the side stream
          hipLaunchKernelGGL(k_potrf_syrk, dim3((unsigned)ntb), dim3(Cfg4::NTHREADS), SMEM_REAL_D, ctx->stream, A22, lda, mb, Lp, lda, NBO / BK, 1);
          HIPCHK(ctx, hipEventRecord(ctx->la_ev[0], ctx->stream));  // both panels of this outer block are final
          HIPCHK(ctx, hipStreamWaitEvent(side, ctx->la_ev[0], 0));
          const int nr = ntb - 1;
          hipLaunchKernelGGL(k_potrf_syrk, dim3((unsigned)(nr * (nr + 1) / 2)), dim3(Cfg4::NTHREADS), SMEM_REAL_D, side, A22 + BM + (long)BM * lda, lda, mb - BM,
                             Lp + BM, lda, NBO / BK, nr);
          side_busy = true;
          block_event_on_side = true;
        } else {
          hipLaunchKernelGGL(k_potrf_syrk, dim3((unsigned)(ntb * (ntb + 1) / 2)), dim3(Cfg4::NTHREADS), SMEM_REAL_D, ctx->stream, A22, lda, mb, Lp, lda,
                             NBO / BK, ntb);
        }
      }
      HIPCHK(ctx, hipGetLastError());
    }
    // event_cols > 0 (a multiple of 256): block column b of that width is final and no longer read once its last outer block's update has run
    // (on whichever stream reads its panels last)
    if (event_cols > 0 && ((K0 + NBO) % event_cols == 0 || K0 + NBO >= n)) {
      if (!block_event_on_side) NLSCHK(join_side());
      HIPCHK(ctx, hipEventRecord(ctx->blk_ev[K0 / event_cols], block_event_on_side ? side : ctx->stream));
    }
  }
  NLSCHK(join_side());  // (what follows on the main stream reads the whole factor)
  if (rhs_run) {
    HIPCHK(ctx, hipMemcpyAsync(rhs_run, xsol, sizeof(double) * (size_t)n, hipMemcpyDeviceToDevice, ctx->stream));
    if (carried) *carried = true;
  }
  return NLS_OK;
}

// x <- L^-T L^-1 x in place (nls_trsv.h): forwards then backwards in outer blocks of 256 unknowns.  NLS_TRSV=rocblas: two rocblas_dtrsv calls.
static int cho_solve_real(nls_ctx* ctx, const double* L, int n, long ldl, double* x, bool forward_done = false) {
  using namespace trsv;
  const char* mode = std::getenv("NLS_TRSV");
  if (mode && std::string(mode) == "rocblas") {
    BLASCHK(ctx, rocblas_set_stream(ctx->blas, ctx->stream));
    if (!forward_done)
      BLASCHK(ctx, rocblas_dtrsv(ctx->blas, rocblas_fill_lower, rocblas_operation_none, rocblas_diagonal_non_unit, (rocblas_int)n, L, (rocblas_int)ldl, x, 1));
    BLASCHK(ctx, rocblas_dtrsv(ctx->blas, rocblas_fill_lower, rocblas_operation_transpose, rocblas_diagonal_non_unit, (rocblas_int)n, L, (rocblas_int)ldl, x, 1));
    return NLS_OK;
  }
  double* sums = nullptr;
  NLSCHK(ws_get_t(ctx, "trsv.sums", (size_t)OB, &sums));
  const int nblk = (n + OB - 1) / OB;
  for (int b = 0; b < nblk && !forward_done; ++b) {
    const int K0 = b * OB, W = std::min(OB, n - K0), below = n - K0 - W;
    hipLaunchKernelGGL(k_trsv_fwd_block, dim3(1), dim3(256), 0, ctx->stream, L, ldl, K0, W, x);
    if (below > 0) hipLaunchKernelGGL(k_trsv_fwd_update, dim3((unsigned)((below + 255) / 256)), dim3(256), 0, ctx->stream, L, ldl, n, K0, W, x);
  }
  for (int b = nblk - 1; b >= 0; --b) {
    const int K0 = b * OB, W = std::min(OB, n - K0);
    const bool tail = K0 + W < n;
    if (tail) hipLaunchKernelGGL(k_trsv_bwd_outer_sum, dim3((unsigned)W), dim3(256), 0, ctx->stream, L, ldl, n, K0, W, x, sums);
    hipLaunchKernelGGL(k_trsv_bwd_block, dim3(1), dim3(256), 0, ctx->stream, L, ldl, K0, W, x, tail ? sums : (const double*)nullptr);
  }
  HIPCHK(ctx, hipGetLastError());
  return NLS_OK;
}

template <int EPI>
int launch_gemm(nls_ctx* ctx, const GemmParams& p, long M, long N) {
  dim3 grid((unsigned)(N / BN), (unsigned)(M / BM));
  hipLaunchKernelGGL(k_gemm<EPI>, grid, dim3(Cfg4::NTHREADS), SMEM_REAL_D, ctx->stream, p);
  HIPCHK(ctx, hipGetLastError());
  return NLS_OK;
}

int gemm_store(nls_ctx* ctx, const double* A, long lda, const double* B, long ldb, double* C, long ldc, long M, long N, long K) {
  GemmParams p{};
  p.A = A;
  p.B = B;
  p.C = C;
  p.lda = lda;
  p.ldb = ldb;
  p.ldc = ldc;
  p.K = (int)K;
  return launch_gemm<EPI_STORE>(ctx, p, M, N);
}

dim3 grid2(long cols, long rows) { return dim3((unsigned)((cols + 255) / 256), (unsigned)rows); }

// K(Xa, Xb) = exp(-||xa - xb||^2 / 2) + add on padded operands; Xb given transposed ([r_pad][nb_pad]).
int rbf_block(nls_ctx* ctx, const double* Xa_pad, const double* XbT_pad, const double* aa, const double* bb, long ma, long nb,
              long ma_pad, long nb_pad, long r_pad, int same, double add, double* out) {
  GemmParams p{};
  p.A = Xa_pad;
  p.B = XbT_pad;
  p.C = out;
  p.lda = r_pad;
  p.ldb = nb_pad;
  p.ldc = nb_pad;
  p.K = (int)r_pad;
  p.xx = aa;
  p.yy = bb;
  p.m_valid = ma;
  p.n_valid = nb;
  p.same = same;
  p.add = add;
  return launch_gemm<EPI_RBF>(ctx, p, ma_pad, nb_pad);
}

}  // namespace

// Hook (tests / profiling): the dual path's own Cholesky factorisation on host data.  A: n x n column-major, lower triangle in, L out (the strict
// upper triangle is returned as it came); *info = 0 or the 1-based index of the first non-positive pivot (the factor is then garbage).
extern "C" int nls_cholesky_only(nls_ctx* ctx, double* A, int n, int* info) {
  if (!ctx) return NLS_ERR_ARG;
  if (!A || !info || n < 1) return fail(ctx, NLS_ERR_ARG, "nls_cholesky_only: null pointer or n < 1");
  HIPCHK(ctx, hipSetDevice(ctx->device));
  const long ld = round_up(n, BM);
  double* dA = nullptr;
  rocblas_int* dinfo = nullptr;
  NLSCHK(ws_get_t(ctx, "hook.chol", (size_t)ld * ld, &dA));
  NLSCHK(ws_get_t(ctx, "chol.info", 4, &dinfo));
  HIPCHK(ctx, hipMemsetAsync(dA, 0, sizeof(double) * (size_t)ld * ld, ctx->stream));
  HIPCHK(ctx, hipMemcpy2DAsync(dA, sizeof(double) * ld, A, sizeof(double) * n, sizeof(double) * n, (size_t)n, hipMemcpyHostToDevice, ctx->stream));
  BLASCHK(ctx, rocblas_set_stream(ctx->blas, ctx->stream));
  NLSCHK(potrf_lower_real(ctx, dA, n, ld, dinfo));
  rocblas_int hinfo = 0;
  HIPCHK(ctx, hipMemcpy2DAsync(A, sizeof(double) * n, dA, sizeof(double) * ld, sizeof(double) * n, (size_t)n, hipMemcpyDeviceToHost, ctx->stream));
  HIPCHK(ctx, hipMemcpyAsync(&hinfo, dinfo, sizeof(hinfo), hipMemcpyDeviceToHost, ctx->stream));
  HIPCHK(ctx, hipStreamSynchronize(ctx->stream));
  *info = (int)hinfo;
  return NLS_OK;
}

extern "C" int nls_dual_fit(nls_ctx* ctx, const nls_dual_fit_args* a) {
  if (!ctx) return NLS_ERR_ARG;
  if (!a) return fail(ctx, NLS_ERR_ARG, "args is NULL");
  if (!a->Xt || !a->y || !a->s || !a->gammas) return fail(ctx, NLS_ERR_ARG, "Xt, y, s and gammas must not be NULL");
  if (a->n < 2 || a->r < 1 || a->G < 1) return fail(ctx, NLS_ERR_ARG, "need n >= 2, r >= 1, G >= 1");
  if (a->n > 65535 - BM) return fail(ctx, NLS_ERR_ARG, "dual path: n = %ld exceeds 65407 rows (n x n kernel matrices); use the primal path", (long)a->n);
  if (a->gamma_index_in >= a->G) return fail(ctx, NLS_ERR_ARG, "gamma_index_in out of range");
  HIPCHK(ctx, hipSetDevice(ctx->device));
  const double t_start = wall();
  double tm[NLS_NUM_TIMINGS];
  std::memset(tm, 0, sizeof(tm));
  ctx->spans.clear();
  ctx->events_used = 0;

  const long n = a->n, r = a->r;
  const int G = a->G, is_clf = a->is_classifier ? 1 : 0;
  const long n_pad = round_up(n, BM), r_pad = round_up(r, BK);
  const int Gp = (int)round_up(G, BN);
  Prefault prefault;  // the pages of the L_ output are faulted in behind the eigendecomposition (joined before the download starts)
  if (a->L) prefault.start(a->L, sizeof(double) * (size_t)n * n);
  const double *dX = nullptr, *dy = nullptr, *ds_in = nullptr;
  {
    SpanGuard g(ctx, NLS_T_UPLOAD);
    NLSCHK(resident(ctx, "in.X", a->Xt, (size_t)n * r, &dX));
    NLSCHK(resident(ctx, "in.y", a->y, (size_t)n, &dy));
    NLSCHK(resident(ctx, "in.s", a->s, (size_t)n, &ds_in));
  }
  // Weights: s1 = s / sum(s); sn = s1 / median|s1| (_neo_ls_svm.py:252-253).  The median is an O(n) host
  // selection on n doubles; everything else stays on the device.
  std::vector<double> hs((size_t)n), hy((size_t)n);
  HIPCHK(ctx, hipMemcpyAsync(hs.data(), ds_in, sizeof(double) * n, hipMemcpyDeviceToHost, ctx->stream));
  HIPCHK(ctx, hipMemcpyAsync(hy.data(), dy, sizeof(double) * n, hipMemcpyDeviceToHost, ctx->stream));
  HIPCHK(ctx, hipStreamSynchronize(ctx->stream));
  double ssum = 0.0, sysum = 0.0;
  for (long i = 0; i < n; ++i) {
    if (!(hs[i] > 0.0) || !std::isfinite(hs[i]))
      return fail(ctx, NLS_ERR_ARG, "dual path needs strictly positive finite weights (s[%ld] = %g); drop zero-weight rows first", i, hs[i]);
    ssum += hs[i];
  }
  std::vector<double> s1((size_t)n), sn((size_t)n), tmp((size_t)n);
  for (long i = 0; i < n; ++i) {
    s1[i] = hs[i] / ssum;
    tmp[i] = std::fabs(s1[i]);
    sysum += s1[i] * hy[i];
  }
  std::nth_element(tmp.begin(), tmp.begin() + n / 2, tmp.end());
  double med = tmp[n / 2];
  if (n % 2 == 0) med = 0.5 * (med + *std::max_element(tmp.begin(), tmp.begin() + n / 2));
  for (long i = 0; i < n; ++i) sn[i] = s1[i] / med;
  const double ybar = sysum;  // weights s1 sum to one
  double *d_s1 = nullptr, *d_sn = nullptr;
  NLSCHK(ws_get_t(ctx, "dual.s1", (size_t)n_pad, &d_s1));
  NLSCHK(ws_get_t(ctx, "dual.sn", (size_t)n_pad, &d_sn));
  HIPCHK(ctx, hipMemcpyAsync(d_s1, s1.data(), sizeof(double) * n, hipMemcpyHostToDevice, ctx->stream));
  HIPCHK(ctx, hipMemcpyAsync(d_sn, sn.data(), sizeof(double) * n, hipMemcpyHostToDevice, ctx->stream));

  const size_t NN = (size_t)n_pad * n_pad;
  double *Xp = nullptr, *XpT = nullptr, *xx = nullptr, *F = nullptr, *F0 = nullptr, *Q = nullptr, *W = nullptr, *M = nullptr,
         *WW = nullptr, *WQ = nullptr, *lam = nullptr, *evd_e = nullptr, *qy = nullptr, *dgam = nullptr, *R = nullptr;
  rocblas_int* dinfo = nullptr;
  NLSCHK(ws_get_t(ctx, "dual.Xp", (size_t)n_pad * r_pad, &Xp));
  NLSCHK(ws_get_t(ctx, "dual.XpT", (size_t)r_pad * n_pad, &XpT));
  NLSCHK(ws_get_t(ctx, "dual.xx", (size_t)n_pad, &xx));
  NLSCHK(ws_get_t(ctx, "dual.F", NN, &F));
  NLSCHK(ws_get_t(ctx, "dual.F0", NN, &F0));
  NLSCHK(ws_get_t(ctx, "dual.Q", NN, &Q));
  NLSCHK(ws_get_t(ctx, "dual.W", NN, &W));
  NLSCHK(ws_get_t(ctx, "dual.M", NN, &M));
  NLSCHK(ws_get_t(ctx, "dual.WW", NN, &WW));
  NLSCHK(ws_get_t(ctx, "dual.WQ", NN, &WQ));
  NLSCHK(ws_get_t(ctx, "dual.lam", (size_t)n_pad, &lam));
  NLSCHK(ws_get_t(ctx, "dual.e", (size_t)n_pad, &evd_e));
  NLSCHK(ws_get_t(ctx, "dual.qy", (size_t)n_pad, &qy));
  NLSCHK(ws_get_t(ctx, "evd.info", 4, &dinfo));
  NLSCHK(ws_get_t(ctx, "sweep.gammas", (size_t)G, &dgam));
  NLSCHK(ws_get_t(ctx, "dual.R", (size_t)n_pad * Gp, &R));
  HIPCHK(ctx, hipMemcpyAsync(dgam, a->gammas, sizeof(double) * G, hipMemcpyHostToDevice, ctx->stream));

  // ---- D1: F = rbf(Xt, 1/2) + 1 -----------------------------------------------------------------
  {
    SpanGuard g(ctx, NLS_T_GRAM);
    hipLaunchKernelGGL(k_copy_pad, grid2(r_pad, n_pad), dim3(256), 0, ctx->stream, dX, n, r, r, Xp, n_pad, r_pad);
    hipLaunchKernelGGL(k_transpose_pad, grid2(n_pad, r_pad), dim3(256), 0, ctx->stream, dX, n, r, r, XpT, r_pad, n_pad);
    hipLaunchKernelGGL(k_row_sqnorm, dim3((unsigned)((n + 3) / 4)), dim3(256), 0, ctx->stream, dX, n, r, r, xx);
    HIPCHK(ctx, hipGetLastError());
    NLSCHK(rbf_block(ctx, Xp, XpT, xx, xx, n, n, n_pad, n_pad, r_pad, 1, 1.0, F));
    tm[NLS_T_GRAM_LAUNCHES] += 1;
    tm[NLS_T_GRAM_FLOPS] += 2.0 * n * n * r;
  }
  HostPin pinL;
  // ---- D2: EVD of sn K sn ------------------------------------------------------------------------
  double* Qev = nullptr;  // eigenvectors: in Q (rocSOLVER path) or in the EVD's own workspace
  {
    SpanGuard g(ctx, NLS_T_EVD);
    hipLaunchKernelGGL(k_dual_scale_sym, grid2(n, n), dim3(256), 0, ctx->stream, F, n_pad, d_sn, n, Q, n);
    HIPCHK(ctx, hipGetLastError());
    NLSCHK(evd_symmetric(ctx, Q, (int)n, lam, evd_e, dinfo, &Qev));
  }
  // ---- D3: reduced sweep -------------------------------------------------------------------------
  double *T = nullptr, *HD = nullptr, *AG = nullptr, *FA = nullptr, *SG = nullptr;
  // (Kt W)^2 lives in the buffer the EVD has consumed (its eigenvectors are in the EVD's own workspace, or - rocSOLVER
  // path - in Q itself, in which case a separate buffer is taken); the Cholesky stage reuses Q only after the sweep.
  double* KW2 = Q;
  if (Qev == Q) NLSCHK(ws_get_t(ctx, "dual.KW2", NN, &KW2));
  NLSCHK(ws_get_t(ctx, "dual.SG", (size_t)n_pad * Gp, &SG));
  NLSCHK(ws_get_t(ctx, "dual.T", (size_t)n_pad * Gp, &T));
  NLSCHK(ws_get_t(ctx, "dual.HD", (size_t)n_pad * Gp, &HD));
  NLSCHK(ws_get_t(ctx, "dual.AG", (size_t)n_pad * Gp, &AG));
  NLSCHK(ws_get_t(ctx, "dual.FA", (size_t)n_pad * Gp, &FA));
  {
    SpanGuard g(ctx, NLS_T_ROTATE);
    hipLaunchKernelGGL(k_dual_build_W, grid2(n_pad, n_pad), dim3(256), 0, ctx->stream, Qev, n, d_sn, W, n_pad);
    hipLaunchKernelGGL(k_dual_qty, dim3((unsigned)n), dim3(256), 0, ctx->stream, Qev, n, d_sn, dy, qy);
    hipLaunchKernelGGL(k_zero_diag_copy, grid2(n_pad, n_pad), dim3(256), 0, ctx->stream, F, n_pad, n_pad, n, F0);
    HIPCHK(ctx, hipGetLastError());
    NLSCHK(gemm_store(ctx, F0, n_pad, W, n_pad, M, n_pad, n_pad, n_pad, n_pad));
    // column sums of W for Kt W = M + 2 W - 1 cs (see k_dual_hadamards)
    const long cchunks = std::min<long>(64, (n + 127) / 128), crows = (n + cchunks - 1) / cchunks;
    double *cpart2 = nullptr, *cs = nullptr;
    NLSCHK(ws_get_t(ctx, "dual.cpart", (size_t)cchunks * n_pad, &cpart2));
    NLSCHK(ws_get_t(ctx, "dual.cs", (size_t)n_pad, &cs));
    hipLaunchKernelGGL(k_col_partial_sums, dim3((unsigned)((n_pad + 255) / 256), (unsigned)cchunks), dim3(256), 0, ctx->stream, W, n, n_pad,
                       crows, cpart2);
    hipLaunchKernelGGL(k_sum_partials, dim3((unsigned)((n_pad + 31) / 32)), dim3(256), 0, ctx->stream, cpart2, cchunks, n_pad, cs, 0);
    hipLaunchKernelGGL(k_dual_hadamards, grid2(n_pad, n_pad), dim3(256), 0, ctx->stream, W, M, qy, cs, n, n_pad, M, WW, WQ, KW2);
    HIPCHK(ctx, hipGetLastError());
    tm[NLS_T_ROTATE_LAUNCHES] += 1;
    tm[NLS_T_ROTATE_FLOPS] += 2.0 * n * n * n;
  }
  {
    SpanGuard g(ctx, NLS_T_SWEEP);
    const long tot = n_pad * (long)Gp;
    hipLaunchKernelGGL(k_rgrid, dim3((unsigned)((tot + 255) / 256)), dim3(256), 0, ctx->stream, lam, dgam, (int)n, G, (int)n_pad, Gp, R);
    HIPCHK(ctx, hipGetLastError());
    NLSCHK(gemm_store(ctx, M, n_pad, R, Gp, T, Gp, n_pad, Gp, n_pad));
    NLSCHK(gemm_store(ctx, WW, n_pad, R, Gp, HD, Gp, n_pad, Gp, n_pad));
    NLSCHK(gemm_store(ctx, WQ, n_pad, R, Gp, AG, Gp, n_pad, Gp, n_pad));
    NLSCHK(gemm_store(ctx, F0, n_pad, AG, Gp, FA, Gp, n_pad, Gp, n_pad));
    NLSCHK(gemm_store(ctx, KW2, n_pad, R, Gp, SG, Gp, n_pad, Gp, n_pad));  // sum_j (Kt W)_ij^2 / (gamma_g + lam_j): 1 - sigma^2 on the grid
    hipLaunchKernelGGL(k_dual_yloo, dim3((unsigned)((tot + 255) / 256)), dim3(256), 0, ctx->stream, T, HD, AG, FA, tot);
    HIPCHK(ctx, hipGetLastError());
    tm[NLS_T_SWEEP_LAUNCHES] += 5;
    tm[NLS_T_SWEEP_FLOPS] += 10.0 * n * n * G;
  }
  // ---- D4: selection -----------------------------------------------------------------------------
  const long nblk = (n + 63) / 64;
  double *part = nullptr, *errs = nullptr;
  NLSCHK(ws_get_t(ctx, "loo.part", (size_t)nblk * 3 * Gp, &part));
  NLSCHK(ws_get_t(ctx, "loo.errs", (size_t)3 * Gp, &errs));
  {
    SpanGuard g(ctx, NLS_T_LOO);
    hipLaunchKernelGGL(k_dual_errors, dim3((unsigned)nblk), dim3(256), 0, ctx->stream, T, dy, d_s1, n, G, Gp, is_clf, part);
    hipLaunchKernelGGL(k_sum_partials, dim3((unsigned)((3 * Gp + 31) / 32)), dim3(256), 0, ctx->stream, part, nblk, 3L * Gp, errs);
    HIPCHK(ctx, hipGetLastError());
  }
  std::vector<double> herrs((size_t)3 * Gp), hobj((size_t)G);
  HIPCHK(ctx, hipMemcpyAsync(herrs.data(), errs, sizeof(double) * 3 * Gp, hipMemcpyDeviceToHost, ctx->stream));
  // NLS_PIN_OUTPUT=1: page-lock the L_ output here (the host would wait anyway) and send the finished block columns with asynchronous copies
  // (round 3).  Default since round 4: no registration - a helper thread sends them into the pageable buffer while the factorisation runs
  // (registering and releasing 800 MB cost ~ 30 ms of the call that nothing hid; profiles/r04_dual_L_download.log).
  static const bool pin_on = [] { const char* m = std::getenv("NLS_PIN_OUTPUT"); return m && m[0] == '1'; }();
  if (a->L && pin_on) pinL.pin(a->L, sizeof(double) * (size_t)n * n);
  HIPCHK(ctx, hipStreamSynchronize(ctx->stream));
  for (int g = 0; g < G; ++g) hobj[g] = is_clf ? (herrs[Gp + g] + herrs[2 * Gp + g]) + herrs[g] : herrs[g];  // :296-302
  int opt = a->gamma_index_in;
  if (opt < 0) {
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
  double *loo_res = nullptr, *res = nullptr, *sig = nullptr, *cpart = nullptr, *csum = nullptr, *alpha = nullptr;
  NLSCHK(ws_get_t(ctx, "out.loo_res", (size_t)n_pad, &loo_res));
  NLSCHK(ws_get_t(ctx, "out.res", (size_t)n_pad, &res));
  NLSCHK(ws_get_t(ctx, "out.loo_std", (size_t)n_pad, &sig));
  NLSCHK(ws_get_t(ctx, "dual.alpha", (size_t)n_pad, &alpha));
  const long cblk = (n + 255) / 256;
  NLSCHK(ws_get_t(ctx, "loo.cpart", (size_t)cblk * 2, &cpart));
  NLSCHK(ws_get_t(ctx, "loo.csum", 4, &csum));
  {
    SpanGuard g(ctx, NLS_T_LOO);
    hipLaunchKernelGGL(k_dual_column, dim3((unsigned)cblk), dim3(256), 0, ctx->stream, T, dy, d_s1, n, Gp, opt, is_clf, ybar, loo_res,
                       cpart);
    hipLaunchKernelGGL(k_sum_partials, dim3(1), dim3(256), 0, ctx->stream, cpart, cblk, 2L, csum);
    HIPCHK(ctx, hipGetLastError());
  }
  double hsum[2];
  HIPCHK(ctx, hipMemcpyAsync(hsum, csum, 2 * sizeof(double), hipMemcpyDeviceToHost, ctx->stream));

  // ---- D5: Cholesky re-solve, residuals, sigma -----------------------------------------------------
  double* M2 = Q;  // n x n with leading dimension n_pad; Q (n_pad x n_pad) is dead
  bool pipelined_L = false;
  struct CopyJoin {  // an early (error) return must not leave the copy stream writing into the caller's L
    hipStream_t s = nullptr;
    ~CopyJoin() {
      if (s) (void)hipStreamSynchronize(s);
    }
  } copy_join;
  struct HelperJoin {  // the helper threads that download L_ into pageable memory: never outlive the call
    enum { LANES = 2 };  // (four lanes: no further gain)
    std::thread t[LANES];
    int rc[LANES] = {NLS_OK, NLS_OK};
    std::string msg[LANES];  // each lane's own failure message (tls_err_sink): ctx->err belongs to the calling thread
    void join() {
      for (auto& th : t)
        if (th.joinable()) th.join();
    }
    ~HelperJoin() { join(); }
  } dl;
  // alpha(gamma*) = M^-1 y = sn W (gamma* + Lam)^-1 W^T sn y is the selected column of AG, which the sweep has already formed for the whole
  // grid (_neo_ls_svm.py:313-316 solve it with the Cholesky factor: two n x n triangular solves, 10.6 ms at n = 10^4).  The factorisation
  // itself only produces the L_ output and is skipped when the caller does not ask for it.
  hipLaunchKernelGGL(k_dual_take_column, dim3((unsigned)((n + 255) / 256)), dim3(256), 0, ctx->stream, AG, (long)Gp, opt, n, alpha);
  HIPCHK(ctx, hipGetLastError());
  if (a->L) {
    SpanGuard g(ctx, NLS_T_CHOLESKY);
    hipLaunchKernelGGL(k_dual_chol_inputs, grid2(n, n), dim3(256), 0, ctx->stream, F, n_pad, d_sn, n, gamma_opt, M2, n_pad, (double*)nullptr);
    HIPCHK(ctx, hipGetLastError());
    // own factorisation (nls_potrf.h): 24.7 ms at n = 10^4 against rocsolver_dpotrf's 42.5 (a blocked variant on rocBLAS trsm / syrk with
    // rocSOLVER leaves had given 49.7 against 52.7 with the triangular solves still inside)
    // (leading dimension n_pad: the rank-128 update reads whole 128-row blocks.)  Into page-locked memory the finished block columns of 512
    // travel on the copy stream while the following ones are factored; pageable outputs are downloaded afterwards in one piece.
    pipelined_L = pinL.p != nullptr || is_pinned_host(a->L);  // page-locked by this call (NLS_PIN_OUTPUT=1) or by the caller (nls_host_register)
    NLSCHK(ensure_copy_stream(ctx, (int)((n + 511) / 512)));
    copy_join.s = ctx->copy_stream;
    // alpha = cho_solve(L_, y): the forward substitution travels with the factorisation (alpha holds the running right-hand side)
    HIPCHK(ctx, hipMemcpyAsync(alpha, dy, sizeof(double) * n, hipMemcpyDeviceToDevice, ctx->stream));
    bool carried = false;
    NLSCHK(potrf_lower_real(ctx, M2, (int)n, n_pad, dinfo, 512, alpha, &carried));
    if (pipelined_L) {
      NLSCHK(download_block_columns(ctx, a->L, M2, (int)n, n_pad, sizeof(double), 512, false));
    } else {
      prefault.join();
      // pageable output: every copy blocks its caller until the block column has arrived (and touches the pages of a fresh buffer for the
      // first time) - in two threads of their own (even and odd block columns, a stream each), beside the factorisation:
      // 514 -> 496 ms per c4 fit against the page-locked path (profiles/r04_dual_L_download.log)
      void* hostL = a->L;
      for (int lane = 0; lane < HelperJoin::LANES; ++lane)
        dl.t[lane] = std::thread([ctx, hostL, M2, n, n_pad, lane, &dl] {
          tls_err_sink = &dl.msg[lane];  // (this thread's failures go to its own slot, never to ctx->err: the caller reports them after the join)
          hipStream_t cs = lane == 0 ? ctx->copy_stream : ctx->copy_lane[lane - 1];
          hipError_t e = hipSetDevice(ctx->device);
          if (e != hipSuccess) {
            dl.rc[lane] = fail(ctx, NLS_ERR_HIP, "hipSetDevice in download lane %d: %s", lane, hipGetErrorString(e));
            return;
          }
          dl.rc[lane] = download_block_columns(ctx, hostL, M2, (int)n, n_pad, sizeof(double), 512, false, lane, HelperJoin::LANES, cs);
          if (dl.rc[lane] == NLS_OK && (e = hipStreamSynchronize(cs)) != hipSuccess)
            dl.rc[lane] = fail(ctx, NLS_ERR_HIP, "hipStreamSynchronize in download lane %d: %s", lane, hipGetErrorString(e));
        });
    }
    // alpha = cho_solve(L_, y) (_neo_ls_svm.py:314: "resolve the linear system for better accuracy"): two triangular solves with one right-hand
    // side against the factor just formed, so that the returned pair satisfies alpha == cho_solve(L_, y) to rounding.  (The selected column of
    // the sweep's table above is the same vector from the eigendecomposition; it stays the answer when no factor is asked for.)
    NLSCHK(cho_solve_real(ctx, M2, (int)n, n_pad, alpha, carried));
    NLSCHK(check_info(ctx, dinfo, "Cholesky factorisation (potrf)"));
  } else {
    // no factorisation, hence no pivot test: gamma* diag(sn^-2) + K is positive definite iff gamma* + lam_min(sn K sn) > 0
    double lam_min = 0.0;
    HIPCHK(ctx, hipMemcpyAsync(&lam_min, lam, sizeof(double), hipMemcpyDeviceToHost, ctx->stream));
    HIPCHK(ctx, hipStreamSynchronize(ctx->stream));
    if (!(gamma_opt + lam_min > 0.0))
      return fail(ctx, NLS_ERR_LINALG, "gamma* diag(sn^-2) + K is not positive definite at gamma* = %g (smallest eigenvalue of sn K sn: %g)", gamma_opt, lam_min);
  }
  {
    SpanGuard g(ctx, NLS_T_RESIDUALS);
    hipLaunchKernelGGL(k_dual_gemv, dim3((unsigned)((n + 3) / 4)), dim3(256), 0, ctx->stream, F, n_pad, n, n, alpha, 0.0, dy, is_clf, res);
    // sigma^2 = 1 - diag(Kt M^-1 Kt^T) (_neo_ls_svm.py:321-322) with M^-1 = W (gamma + Lam)^-1 W^T from the eigendecomposition the
    // sweep already has: the selected column of SG.  The reference's n x n cho_solve (one 10^12-flop triangular solve here in round 1)
    // is not needed for it; L is still factorised for alpha and returned as L_.
    hipLaunchKernelGGL(k_dual_sigma_col, dim3((unsigned)cblk), dim3(256), 0, ctx->stream, SG, Gp, opt, n, sig);
    HIPCHK(ctx, hipGetLastError());
  }
  {
    SpanGuard g(ctx, NLS_T_DOWNLOAD);
    auto d2h = [&](void* dst, const void* src, size_t bytes) -> int {
      if (dst) HIPCHK(ctx, hipMemcpyAsync(dst, src, bytes, hipMemcpyDeviceToHost, ctx->stream));
      return NLS_OK;
    };
    NLSCHK(d2h(a->alpha, alpha, sizeof(double) * n));
    NLSCHK(d2h(a->lam, lam, sizeof(double) * n));
    NLSCHK(d2h(a->loo_residuals, loo_res, sizeof(double) * n));
    NLSCHK(d2h(a->loo_std, sig, sizeof(double) * n));
    NLSCHK(d2h(a->residuals, res, sizeof(double) * n));
    // Column-major lower Cholesky factor == row-major upper factor U (M2 = U^T U): scipy's lower=False layout.
    dl.join();
    for (int lane = 0; lane < HelperJoin::LANES; ++lane)
      if (dl.rc[lane] != NLS_OK) return fail(ctx, dl.rc[lane], "download of the Cholesky factor (lane %d): %s", lane, dl.msg[lane].c_str());
    if (pipelined_L) HIPCHK(ctx, hipStreamSynchronize(ctx->copy_stream));
  }
  NLSCHK(spans_collect(ctx, tm));
  if (a->loo_errors) std::memcpy(a->loo_errors, herrs.data(), sizeof(double) * G);
  if (a->objective) std::memcpy(a->objective, hobj.data(), sizeof(double) * G);
  if (a->gamma_index) *a->gamma_index = opt;
  if (a->loo_score) *a->loo_score = is_clf ? hsum[0] : 1.0 - hsum[0] / hsum[1];
  tm[NLS_T_TOTAL] = wall() - t_start;
  if (a->timings) std::memcpy(a->timings, tm, sizeof(tm));
  return NLS_OK;
}

extern "C" int nls_dual_predict(nls_ctx* ctx, const double* Xq, int64_t m, const double* Xt, int64_t n, int r, const double* alpha,
                                const double* L, double* yhat, double* sigma) {
  if (!ctx) return NLS_ERR_ARG;
  if (!Xq || !Xt || m < 0 || n < 1 || r < 1) return fail(ctx, NLS_ERR_ARG, "Xq/Xt NULL or bad sizes");
  if (yhat && !alpha) return fail(ctx, NLS_ERR_ARG, "alpha is required for yhat");
  if (sigma && !L) return fail(ctx, NLS_ERR_ARG, "L is required for sigma");
  HIPCHK(ctx, hipSetDevice(ctx->device));
  if (m == 0 || (!yhat && !sigma)) return NLS_OK;
  const long n_pad = round_up(n, BM), r_pad = round_up(r, BK);
  const double *dXt = nullptr, *dXq = nullptr;
  NLSCHK(resident(ctx, "in.X", Xt, (size_t)n * r, &dXt));
  NLSCHK(resident(ctx, "in.Xq", Xq, (size_t)m * r, &dXq));
  // Query rows per chunk: K block of at most ~2 GiB.
  long mc = std::max<long>(BM, (long)(((size_t)2 << 30) / ((size_t)n_pad * 8)) / BM * BM);
  mc = std::min<long>(mc, 65535 / BM * BM);  // the row index of the padding kernels is gridDim.y (HIP limit 65535)
  mc = std::min<long>(mc, round_up(m, BM));
  double *XtT = nullptr, *xx = nullptr, *Qp = nullptr, *qq = nullptr, *K = nullptr, *dalpha = nullptr, *dL = nullptr, *dy = nullptr,
         *dsig = nullptr, *asum = nullptr;
  NLSCHK(ws_get_t(ctx, "dual.XpT", (size_t)r_pad * n_pad, &XtT));
  NLSCHK(ws_get_t(ctx, "dual.xx", (size_t)n_pad, &xx));
  NLSCHK(ws_get_t(ctx, "dual.Qp", (size_t)mc * r_pad, &Qp));
  NLSCHK(ws_get_t(ctx, "dual.qq", (size_t)mc, &qq));
  NLSCHK(ws_get_t(ctx, "dual.Kq", (size_t)mc * n_pad, &K));
  NLSCHK(ws_get_t(ctx, "out.res", (size_t)round_up(m, BM), &dy));
  NLSCHK(ws_get_t(ctx, "out.loo_std", (size_t)round_up(m, BM), &dsig));
  NLSCHK(ws_get_t(ctx, "loo.csum", 4, &asum));
  hipLaunchKernelGGL(k_transpose_pad, grid2(n_pad, r_pad), dim3(256), 0, ctx->stream, dXt, (long)n, (long)r, (long)r, XtT, r_pad, n_pad);
  hipLaunchKernelGGL(k_row_sqnorm, dim3((unsigned)((n + 3) / 4)), dim3(256), 0, ctx->stream, dXt, (long)n, (long)r, (long)r, xx);
  HIPCHK(ctx, hipGetLastError());
  double halpha_sum = 0.0;
  if (yhat) {
    NLSCHK(ws_get_t(ctx, "dual.alpha", (size_t)n_pad, &dalpha));
    HIPCHK(ctx, hipMemcpyAsync(dalpha, alpha, sizeof(double) * n, hipMemcpyHostToDevice, ctx->stream));
    for (long i = 0; i < n; ++i) halpha_sum += alpha[i];  // b = sum(alpha), _neo_ls_svm.py:670
  }
  if (sigma) {
    NLSCHK(ws_get_t(ctx, "dual.Q", (size_t)n_pad * n_pad, &dL));
    HIPCHK(ctx, hipMemcpyAsync(dL, L, sizeof(double) * n * n, hipMemcpyHostToDevice, ctx->stream));
  }
  const double one = 1.0;
  for (long q0 = 0; q0 < m; q0 += mc) {
    const long rows = std::min<long>(mc, m - q0);
    const long rows_pad = round_up(rows, BM);
    hipLaunchKernelGGL(k_copy_pad, grid2(r_pad, rows_pad), dim3(256), 0, ctx->stream, dXq + q0 * r, rows, (long)r, (long)r, Qp, rows_pad, r_pad);
    hipLaunchKernelGGL(k_row_sqnorm, dim3((unsigned)((rows + 3) / 4)), dim3(256), 0, ctx->stream, dXq + q0 * r, rows, (long)r, (long)r, qq);
    HIPCHK(ctx, hipGetLastError());
    NLSCHK(rbf_block(ctx, Qp, XtT, qq, xx, rows, n, rows_pad, n_pad, r_pad, 0, 0.0, K));
    if (yhat) {
      hipLaunchKernelGGL(k_dual_gemv, dim3((unsigned)((rows + 3) / 4)), dim3(256), 0, ctx->stream, K, n_pad, rows, (long)n, dalpha, halpha_sum,
                         (const double*)nullptr, 0, dy + q0);
      HIPCHK(ctx, hipGetLastError());
    }
    if (sigma) {
      // Z = Lc^-1 K^T with Lc the column-major lower factor (== the row-major upper U); K (rows x n_pad row-major) is
      // K^T column-major with leading dimension n_pad.  sigma2_i = 1 - ||Z[:, i]||^2.
      BLASCHK(ctx, rocblas_dtrsm(ctx->blas, rocblas_side_left, rocblas_fill_lower, rocblas_operation_none, rocblas_diagonal_non_unit,
                                 (rocblas_int)n, (rocblas_int)rows, &one, dL, (rocblas_int)n, K, (rocblas_int)n_pad));
      hipLaunchKernelGGL(k_dual_sigma, dim3((unsigned)((rows + 3) / 4)), dim3(256), 0, ctx->stream, K, K, n_pad, rows, (long)n, dsig + q0);
      HIPCHK(ctx, hipGetLastError());
    }
  }
  if (yhat) HIPCHK(ctx, hipMemcpyAsync(yhat, dy, sizeof(double) * m, hipMemcpyDeviceToHost, ctx->stream));
  if (sigma) HIPCHK(ctx, hipMemcpyAsync(sigma, dsig, sizeof(double) * m, hipMemcpyDeviceToHost, ctx->stream));
  HIPCHK(ctx, hipStreamSynchronize(ctx->stream));
  return NLS_OK;
}
