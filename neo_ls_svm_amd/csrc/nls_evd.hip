// Eigendecomposition of the path's Hermitian / real symmetric matrices (P4: _neo_ls_svm.py:120, D2: :265):
// tridiagonalisation by the panel of nls_trd.h (two or three kernels per column), rocSOLVER's divide-and-conquer
// (stedc) on the tridiagonal matrix, and a blocked back-transformation built from rocBLAS GEMMs (apply_q_blocked).
// NLS_EVD=rocsolver selects the all-rocSOLVER zheevd / dsyevd instead.
#include "nls_host.h"
#include "nls_trd.h"
#include "nls_sb.h"
#include "nls_chase.h"
#include "nls_q2.h"
#include "nls_stedc.h"

namespace nls {

static inline rocblas_status trd_rank2k(rocblas_handle h, int n2, int k, const trd::Z* V, long ldv, const trd::Z* W, long ldw, trd::Z* C, long ldc) {
  const rocblas_double_complex minus_one(-1.0, 0.0);
  const double one = 1.0;
  return rocblas_zher2k(h, rocblas_fill_lower, rocblas_operation_none, n2, k, &minus_one, reinterpret_cast<const rocblas_double_complex*>(V),
                        (rocblas_int)ldv, reinterpret_cast<const rocblas_double_complex*>(W), (rocblas_int)ldw, &one,
                        reinterpret_cast<rocblas_double_complex*>(C), (rocblas_int)ldc);
}
static inline rocblas_status trd_rank2k(rocblas_handle h, int n2, int k, const double* V, long ldv, const double* W, long ldw, double* C, long ldc) {
  const double minus_one = -1.0, one = 1.0;
  return rocblas_dsyr2k(h, rocblas_fill_lower, rocblas_operation_none, n2, k, &minus_one, V, (rocblas_int)ldv, W, (rocblas_int)ldw, &one, C,
                        (rocblas_int)ldc);
}

// Two or three kernels per column (nls_trd.h).  Folding the column update into the neighbouring kernels saves a launch
// (~5 us) per column but every matrix-vector tile then forms x on the fly (two loads + the reduced scalars): a gain
// while the columns are latency bound (n = 4097: 181 -> 176 ms, n = 1025: 24.2 -> 22.4 ms), a loss once the tiles are
// bandwidth bound (real n = 10^4: 778 -> 800 ms).  NLS_TRD_KERNELS=2 / 3 overrides the size rule.
static bool trd_three_kernels(int n) {
  const char* m = std::getenv("NLS_TRD_KERNELS");
  if (m && m[0] == '3') return true;
  if (m && m[0] == '2') return false;
  return n > 6144;
}
static int trd_dotgroups(int n) {  // 64-row groups per dot block (NLS_TRD_DOTGROUPS overrides the size rule; test hook)
  if (const char* m = std::getenv("NLS_TRD_DOTGROUPS")) {
    const int g = std::atoi(m);
    if (g >= 1 && g <= 16) return g;
  }
  return n > 6144 ? 4 : 1;
}
static bool trd_use_rocblas_rank2k() {  // NLS_TRD_RANK2K=rocblas: trailing updates through zher2k / dsyr2k (diagnostic)
  const char* m = std::getenv("NLS_TRD_RANK2K");
  return m && std::string(m) == "rocblas";
}

// A: n x n column-major (lda), lower triangle in, reflectors + (d, e on the diagonals) out; d[n], e[n-1], tau[n-1].
template <class T>
static int trd_fused(nls_ctx* ctx, T* A, int n, long lda, double* d, double* e, T* tau) {
  using namespace trd;
  if (n < 1) return NLS_OK;
  const int NSC = (n + TS - 1) / TS, NSR = (n + RT - 1) / RT;  // column / row strips of the matrix-vector tiles
  const int nrb = (n + ROWT - 1) / ROWT;
  const int ndot_max = (n + RD - 1) / RD;
  Args<T> a{};
  a.A = A;
  a.lda = lda;
  a.n = n;
  a.d = d;
  a.e = e;
  a.tau = tau;
  a.nrowblocks = nrb;
  {
    const char* m = std::getenv("NLS_TRD_BOUSTROPHEDON");
    a.boustrophedon = m ? (m[0] == '1') : ((size_t)n * n * sizeof(T) / 2 > ((size_t)200 << 20));
  }
  NLSCHK(ws_get_t(ctx, "trd.W", (size_t)n * NB, &a.W));
  NLSCHK(ws_get_t(ctx, "trd.wtmp", (size_t)n, &a.wtmp));
  NLSCHK(ws_get_t(ctx, "trd.xvec", (size_t)n, &a.xvec));
  NLSCHK(ws_get_t(ctx, "trd.ylow", (size_t)NSC * n, &a.ylow));
  NLSCHK(ws_get_t(ctx, "trd.yup", (size_t)NSR * n, &a.yup));
  NLSCHK(ws_get_t(ctx, "trd.zpart", (size_t)ndot_max * 2 * NB, &a.zpart));
  NLSCHK(ws_get_t(ctx, "trd.spart", (size_t)nrb, &a.spart));
  NLSCHK(ws_get_t(ctx, "trd.pnorm", (size_t)nrb, &a.pnorm));
  HIPCHK(ctx, hipMemsetAsync(a.W, 0, sizeof(T) * (size_t)n * NB, ctx->stream));
  BLASCHK(ctx, rocblas_set_stream(ctx->blas, ctx->stream));
  const bool two_kernels = !trd_three_kernels(n);
  T *wt[2] = {a.wtmp, nullptr}, *sp[2] = {a.spart, nullptr};
  if (two_kernels) {
    NLSCHK(ws_get_t(ctx, "trd.wtmp2", (size_t)n, &wt[1]));
    NLSCHK(ws_get_t(ctx, "trd.spart2", (size_t)nrb, &sp[1]));
    NLSCHK(ws_get_t(ctx, "trd.bvec", (size_t)n, &a.bvec));
  }
  // 2-3 kernels per column, ~8200 launches at n = 4097.  (A hipGraph replay of this sequence and a persistent one-launch
  // panel were built and measured slower in round 2 - profiles/r02_trd_graph.log, profiles/r02_trd_persistent.log - and so was, in round 5,
  // a ONE-launch column whose finish blocks ride behind the matrix-vector blocks on a flag hand-off (write-through stores + polled flags, the
  // bulge chase's protocol): bit-identical, 110.6 against 100.6 ms at n = 4097 - profiles/r05_evd_one_launch.md.  The kernel times of a column add
  // up to its wall time: there is no launch gap to remove, only the kernels' own dependent round trips.  None of the three is in the library.)
  auto enqueue = [&]() -> int {
  int cur = 0;  // buffer that the column being finished writes (two-kernel variant)
  for (int j0 = 0; j0 < n; j0 += NB) {
    const int jend = std::min(j0 + NB, n);
    a.j0 = j0;
    for (int j = j0; j < jend; ++j) {
      a.j = j;
      const int i = j - j0;
      const int S0 = (j + 1) / TS, K = NSC - S0, ntiles = K * (K + 1) / 2;
      if (two_kernels) {
        a.wtmp = wt[cur];
        a.spart = sp[cur];
        a.wtmp_prev = wt[cur ^ 1];
        a.spart_prev = sp[cur ^ 1];
        a.ndot = (n - j + RD - 1) / RD;  // dot blocks start at row j
        a.make_base = j + 1 < jend;
        hipLaunchKernelGGL(k_trd_hemv2<T>, dim3((j < n - 1 ? ntiles : 0) + a.ndot), dim3(256), 0, ctx->stream, a, S0, j < n - 1 ? ntiles : 0);
        if (j < n - 1) {
          hipLaunchKernelGGL(k_trd_finish2<T>, dim3(nrb), dim3(ROWT * TPR), 0, ctx->stream, a, S0, NSR);
          cur ^= 1;
        }
        continue;
      }
      a.dotgroups = trd_dotgroups(n);
      a.ndot = i > 0 ? (n - j - 1 + RD * a.dotgroups - 1) / (RD * a.dotgroups) : 0;
      hipLaunchKernelGGL(k_trd_column<T>, dim3(nrb), dim3(ROWT * TPR), 0, ctx->stream, a);
      if (j < n - 1) {
        hipLaunchKernelGGL(k_trd_hemv<T>, dim3(ntiles + a.ndot), dim3(256), 0, ctx->stream, a, S0, ntiles);
        hipLaunchKernelGGL(k_trd_finish<T>, dim3(nrb), dim3(ROWT * TPR), 0, ctx->stream, a, S0, NSR);
      }
    }
    HIPCHK(ctx, hipGetLastError());
    const int jl = std::min(jend - 1, n - 2);  // last column of the panel that has a reflector
    const int n2 = n - jend;
    if (n2 > 0) {
      Args<T> ae = a;
      if (two_kernels) {  // the last finished column wrote the buffers that are "previous" now
        ae.wtmp = wt[cur ^ 1];
        ae.spart = sp[cur ^ 1];
      }
      hipLaunchKernelGGL(k_trd_panel_end<T>, dim3((n + 255) / 256), dim3(256), 0, ctx->stream, ae, jl);
      if (trd_use_rocblas_rank2k()) {
        BLASCHK(ctx, trd_rank2k(ctx->blas, n2, jend - j0, A + jend + (long)j0 * lda, lda, a.W + jend, n, A + jend + (long)jend * lda, lda));
      } else {
        const int ut = (n2 + UT - 1) / UT;
        hipLaunchKernelGGL(k_trd_rank2k<T>, dim3(ut * (ut + 1) / 2), dim3(256), 0, ctx->stream, A, lda, a.W, (long)n, n, j0, jend - j0, jend);
      }
    }
    if (jl >= j0)
      hipLaunchKernelGGL(k_trd_restore_subdiag<T>, dim3(1), dim3(64), 0, ctx->stream, A, lda, e, j0, jl - j0 + 1);
    HIPCHK(ctx, hipGetLastError());
  }
  return NLS_OK;
  };  // enqueue

  return enqueue();
}

// rocBLAS shims for the blocked back-transformation
static inline rocblas_status bt_gemm(rocblas_handle h, rocblas_operation ta, rocblas_operation tb, int m, int n, int k, double alpha,
                                     const trd::Z* A, long lda, const trd::Z* B, long ldb, double beta, trd::Z* C, long ldc) {
  const rocblas_double_complex al(alpha, 0.0), be(beta, 0.0);
  return rocblas_zgemm(h, ta, tb, m, n, k, &al, reinterpret_cast<const rocblas_double_complex*>(A), (rocblas_int)lda,
                       reinterpret_cast<const rocblas_double_complex*>(B), (rocblas_int)ldb, &be, reinterpret_cast<rocblas_double_complex*>(C),
                       (rocblas_int)ldc);
}
static inline rocblas_status bt_gemm(rocblas_handle h, rocblas_operation ta, rocblas_operation tb, int m, int n, int k, double alpha,
                                     const double* A, long lda, const double* B, long ldb, double beta, double* C, long ldc) {
  return rocblas_dgemm(h, ta, tb, m, n, k, &alpha, A, (rocblas_int)lda, B, (rocblas_int)ldb, &beta, C, (rocblas_int)ldc);
}
static inline rocblas_status bt_trsm(rocblas_handle h, int m, int n, const trd::Z* A, long lda, trd::Z* B, long ldb) {
  const rocblas_double_complex one(1.0, 0.0);
  return rocblas_ztrsm(h, rocblas_side_left, rocblas_fill_upper, rocblas_operation_none, rocblas_diagonal_non_unit, m, n, &one,
                       reinterpret_cast<const rocblas_double_complex*>(A), (rocblas_int)lda, reinterpret_cast<rocblas_double_complex*>(B),
                       (rocblas_int)ldb);
}
static inline rocblas_status bt_trsm(rocblas_handle h, int m, int n, const double* A, long lda, double* B, long ldb) {
  const double one = 1.0;
  return rocblas_dtrsm(h, rocblas_side_left, rocblas_fill_upper, rocblas_operation_none, rocblas_diagonal_non_unit, m, n, &one, A,
                       (rocblas_int)lda, B, (rocblas_int)ldb);
}
static inline rocblas_operation bt_op_h(const trd::Z*) { return rocblas_operation_conjugate_transpose; }
static inline rocblas_operation bt_op_h(const double*) { return rocblas_operation_transpose; }

// C (n x ncols, column-major, ldc) <- Q C with Q = H_0 H_1 ... H_{nrefl-1}, reflector j stored in column j of A below its unit entry at
// row j + off (off = 1: the reflectors of trd_fused; off = B: the panels of the band reduction, nls_sb.h) - "back-transformation".
template <class T>
static int apply_q_blocked(nls_ctx* ctx, const T* A, long lda, int n, const T* tau, T* C, long ldc, int ncols, int off = 1, int nrefl = -1) {
  using namespace trd;
  if (nrefl < 0) nrefl = n - off;
  if (nrefl <= 0 || ncols <= 0) return NLS_OK;
  // reflectors per block: 512; 1024 from n = 6000 (round 5, real n = 10^4: 256: 57.9, 512: 45.9, 1024: 42.8, 2048: 48.6 ms; complex n = 4097: 18.2 / 15.2 / 14.9).
  // NLS_BT_KB = 128 .. 2048 overrides (diagnostic).
  int kbq = n >= 6000 ? 2 * KBQ : KBQ;
  if (const char* e = std::getenv("NLS_BT_KB")) kbq = std::max(128, std::min(2048, std::atoi(e) / 32 * 32));
  T *Vw = nullptr, *S = nullptr, *W = nullptr;
  NLSCHK(ws_get_t(ctx, "bt.V", (size_t)n * kbq, &Vw));
  NLSCHK(ws_get_t(ctx, "bt.S", (size_t)kbq * kbq, &S));
  NLSCHK(ws_get_t(ctx, "bt.W", (size_t)kbq * ncols, &W));
  // V^H stored explicitly (k_trd_transpose_v) makes the two products with it no-transpose GEMMs.  Measured (round 5, tools/gpu_r05_v.sh): complex
  // n = 4097: 15.15 -> 13.0 ms per back-transformation; real n = 10^4: 42.6-43.9 -> 45.7-46.0 ms (the real transposed-operand kernel is the
  // better one there).  Default: complex only; NLS_BT_VT=0 / 1 overrides.
  T* Vt = nullptr;
  bool want_vt = sizeof(T) == 16;
  if (const char* e = std::getenv("NLS_BT_VT")) want_vt = e[0] == '1';
  if (want_vt) NLSCHK(ws_get_t(ctx, "bt.Vt", (size_t)n * kbq, &Vt));
  BLASCHK(ctx, rocblas_set_stream(ctx->blas, ctx->stream));
  for (int j0 = ((nrefl - 1) / kbq) * kbq; j0 >= 0; j0 -= kbq) {
    const int kb = std::min(kbq, nrefl - j0), r0 = j0 + off, m = n - r0;
    hipLaunchKernelGGL(k_trd_copy_v<T>, dim3((unsigned)(((long)m * kb + 255) / 256)), dim3(256), 0, ctx->stream, A, lda, n, j0, kb, Vw, off);
    if (Vt) {  // V^H explicitly: both products with it are no-transpose GEMMs
      hipLaunchKernelGGL(k_trd_transpose_v<T>, dim3((unsigned)((m + 31) / 32), (unsigned)((kb + 31) / 32)), dim3(32, 8), 0, ctx->stream, Vw, (long)m, kb, Vt);
      BLASCHK(ctx, bt_gemm(ctx->blas, rocblas_operation_none, rocblas_operation_none, kb, kb, m, 1.0, Vt, kb, Vw, m, 0.0, S, kb));
    } else {
      BLASCHK(ctx, bt_gemm(ctx->blas, bt_op_h(A), rocblas_operation_none, kb, kb, m, 1.0, Vw, m, Vw, m, 0.0, S, kb));
    }
    hipLaunchKernelGGL(k_trd_tinv<T>, dim3((unsigned)((kb * kb + 255) / 256)), dim3(256), 0, ctx->stream, S, kb, tau, j0);
    HIPCHK(ctx, hipGetLastError());
    if (Vt)
      BLASCHK(ctx, bt_gemm(ctx->blas, rocblas_operation_none, rocblas_operation_none, kb, ncols, m, 1.0, Vt, kb, C + r0, ldc, 0.0, W, kb));
    else
      BLASCHK(ctx, bt_gemm(ctx->blas, bt_op_h(A), rocblas_operation_none, kb, ncols, m, 1.0, Vw, m, C + r0, ldc, 0.0, W, kb));
    BLASCHK(ctx, bt_trsm(ctx->blas, kb, ncols, S, kb, W, kb));
    BLASCHK(ctx, bt_gemm(ctx->blas, rocblas_operation_none, rocblas_operation_none, m, ncols, kb, -1.0, Vw, m, W, kb, 1.0, C + r0, ldc));
  }
  return NLS_OK;
}

// ================================================================================================================
// Two-stage reduction: dense -> band (nls_sb.h) -> tridiagonal (nls_chase.h), two back-transformations (nls_q2.h and
// apply_q_blocked with off = B).
// ================================================================================================================
template <class T>
struct TwoStageBw;  // default band width per arithmetic
template <>
struct TwoStageBw<double> {
  static constexpr int B = 32;  // n = 10^4: 478 ms against 590 with 64 (the chase's critical path is 2 n stages whose time grows with the band width)
};
template <>
struct TwoStageBw<trd::Z> {
  static constexpr int B = 32;
};

template <class F>
static int sb_lds_optin(nls_ctx* ctx, F f, size_t bytes, const char* name) {
  if (bytes > 65536) {
    const hipError_t r = hipFuncSetAttribute(reinterpret_cast<const void*>(f), hipFuncAttributeMaxDynamicSharedMemorySize, (int)bytes);
    if (r != hipSuccess) return fail(ctx, NLS_ERR_HIP, "dynamic LDS opt-in (%zu bytes) for %s failed: %s", bytes, name, hipGetErrorString(r));
  }
  return NLS_OK;
}

// Stage 1.  A: n x n column-major (lda), lower triangle in; out: band in the B sub-diagonals, block reflectors below it, tau1[n].
// flag (device int): raised when a panel could not be orthogonalised (see nls_sb.h); the caller falls back.
template <class T, int B>
static int sy2sb(nls_ctx* ctx, T* A, int n, long lda, T* tau1, int* dflag, int* ncols_reduced, bool perturb = false) {
  using namespace sb;
  *ncols_reduced = 0;
  HIPCHK(ctx, hipMemsetAsync(tau1, 0, sizeof(T) * (size_t)n, ctx->stream));
  if (n - B < 2) return NLS_OK;
  const int nch_max = (n + RW - 1) / RW, nrb_max = (n + RC - 1) / RC;
  const int parts_max = 32;  // partial results of W = A22 Z per row block (one per wave)
  T *Yb = nullptr, *Zb = nullptr, *Zr = nullptr, *Wb = nullptr, *Gp = nullptr, *Wp = nullptr, *Mp = nullptr;
  PanelSmall<T, B>* ps = nullptr;
  NLSCHK(ws_get_t(ctx, "sb.Y", (size_t)n * B, &Yb));
  NLSCHK(ws_get_t(ctx, "sb.Z", (size_t)n * B, &Zb));
  NLSCHK(ws_get_t(ctx, "sb.Zr", (size_t)n * B, &Zr));
  NLSCHK(ws_get_t(ctx, "sb.W", (size_t)n * B, &Wb));
  NLSCHK(ws_get_t(ctx, "sb.Gp", (size_t)nch_max * B * B, &Gp));
  NLSCHK(ws_get_t(ctx, "sb.Mp", (size_t)nrb_max * B * B, &Mp));
  NLSCHK(ws_get_t(ctx, "sb.Wp", (size_t)parts_max * n * B, &Wp));
  NLSCHK(ws_get_t(ctx, "sb.ps", (size_t)1, &ps));
  const size_t mat = sb_mat_bytes<T, B>(B), tile = sb_mat_bytes<T, B>(64);
  const size_t lds_chol = 2 * mat, lds_recon = 4 * mat, lds_apply = mat + tile, lds_finish = 2 * mat + tile, lds_reduce = 2 * tile, lds_x = tile + mat;
  NLSCHK(sb_lds_optin(ctx, k_sb_small_chol<T, B>, lds_chol, "k_sb_small_chol"));
  NLSCHK(sb_lds_optin(ctx, k_sb_small_recon<T, B>, lds_recon, "k_sb_small_recon"));
  NLSCHK(sb_lds_optin(ctx, k_sb_apply<T, B>, lds_apply, "k_sb_apply"));
  NLSCHK(sb_lds_optin(ctx, k_sb_finish<T, B>, lds_finish, "k_sb_finish"));
  NLSCHK(sb_lds_optin(ctx, k_sb_hemm_reduce<T, B>, lds_reduce, "k_sb_hemm_reduce"));
  NLSCHK(sb_lds_optin(ctx, k_sb_x<T, B>, lds_x, "k_sb_x"));
  const size_t lds_her2k = her2k_lds_bytes<T, B>();
  NLSCHK(sb_lds_optin(ctx, k_sb_her2k<T, B>, lds_her2k, "k_sb_her2k"));
  hipStream_t st = ctx->stream;
  const dim3 red_grid((4 * B * B + 255) / 256);  // (a quad per element)
  // well-conditioned panels take two passes instead of three (k_sb_small_chol); not in the rescue attempt; NLS_SB_ADAPTIVE=0: never
  const char* adaptive_env = std::getenv("NLS_SB_ADAPTIVE");
  const int adaptive = !(adaptive_env && adaptive_env[0] == '0') && !perturb ? 1 : 0;
  // NLS_SB_STAMP=1 (diagnostic): the in-kernel time line of the first panel's k_sb_small_recon, printed when the reduction has been queued
  static const bool want_stamps = [] { const char* m = std::getenv("NLS_SB_STAMP"); return m && m[0] == '1'; }();
  long long* dstamps = nullptr;
  if (want_stamps) {
    NLSCHK(ws_get_t(ctx, "sb.stamps", (size_t)16, &dstamps));
    HIPCHK(ctx, hipMemsetAsync(dstamps, 0, 16 * sizeof(long long), st));
  }
  const char* series_env = std::getenv("NLS_SB_SERIES");  // 0: the third pass's Cholesky factor always by elimination (test of that branch)
  const int series = !(series_env && series_env[0] == '0') ? 1 : 0;
  int j = 0;
  for (;;) {
    const int m = n - j - B;            // rows below the band in column j
    const int kb = std::min(B, m - 1);  // columns with something to annihilate (the last panel may be narrower)
    if (kb <= 0) break;
    const int zh = B - kb, mh = m + zh;  // zero rows on top of Y for the two-sided update of A[j+kb:, j+kb:]
    if (zh > 0) {
      HIPCHK(ctx, hipMemsetAsync(Yb, 0, sizeof(T) * (size_t)n * B, st));
      HIPCHK(ctx, hipMemsetAsync(Zb, 0, sizeof(T) * (size_t)n * B, st));
      HIPCHK(ctx, hipMemsetAsync(Zr, 0, sizeof(T) * (size_t)n * B, st));
    }
    T* P = A + (long)(j + B) + (long)j * lda;
    const int nch = (m + RW - 1) / RW;
    hipLaunchKernelGGL((k_sb_gram<T, B>), dim3(nch), dim3(256), 0, st, P, lda, m, kb, Gp);
    hipLaunchKernelGGL((k_sb_reduce<T>), red_grid, dim3(256), 0, st, Gp, nch, B * B, ps->G, nullptr);
    if (perturb) {  // second attempt after a degenerate panel (nls_sb.h, k_sb_perturb)
      hipLaunchKernelGGL((k_sb_perturb<T>), dim3((unsigned)(((long)m * kb + 255) / 256)), dim3(256), 0, st, P, lda, m, kb, ps->G, B, (unsigned)j);
      hipLaunchKernelGGL((k_sb_gram<T, B>), dim3(nch), dim3(256), 0, st, P, lda, m, kb, Gp);
      hipLaunchKernelGGL((k_sb_reduce<T>), red_grid, dim3(256), 0, st, Gp, nch, B * B, ps->G, nullptr);
    }
    hipLaunchKernelGGL((k_sb_small_chol<T, B>), dim3(1), dim3(256), lds_chol, st, kb, m, 0, ps, dflag, adaptive);
    hipLaunchKernelGGL((k_sb_apply<T, B>), dim3(nch), dim3(256), lds_apply, st, P, lda, m, kb, ps, Yb + zh, (long)n, Gp, 0);
    hipLaunchKernelGGL((k_sb_reduce<T>), red_grid, dim3(256), 0, st, Gp, nch, B * B, ps->G, nullptr);
    // (second pass: its three kernels return at once for a well-conditioned panel - ps->G still holds the first pass's Gram matrix)
    hipLaunchKernelGGL((k_sb_small_chol<T, B>), dim3(1), dim3(256), lds_chol, st, kb, m, 1, ps, dflag, adaptive);
    hipLaunchKernelGGL((k_sb_apply<T, B>), dim3(nch), dim3(256), lds_apply, st, Yb + zh, (long)n, m, kb, ps, Yb + zh, (long)n, Gp, 1);
    hipLaunchKernelGGL((k_sb_reduce<T>), red_grid, dim3(256), 0, st, Gp, nch, B * B, ps->G, &ps->skip2);
    hipLaunchKernelGGL((k_sb_small_recon<T, B>), dim3(1), dim3(256), lds_recon, st, kb, Yb + zh, (long)n, ps, P, lda, tau1 + j, dflag, j == 0 ? dstamps : nullptr, series);
    hipLaunchKernelGGL((k_sb_finish<T, B>), dim3(nch), dim3(256), lds_finish, st, Yb, (long)n, m, kb, ps, Zb, Zr, P, lda);
    HIPCHK(ctx, hipGetLastError());
    T* A22 = A + (long)(j + kb) + (long)(j + kb) * lda;
    const int NT = (mh + HT - 1) / HT, nrb = (mh + RC - 1) / RC;
    // one wave per (row block, part): ~8 waves per CU, at least two tiles per wave
    const int parts = std::max(1, std::min({parts_max, (NT + 1) / 2, (8 * ctx->cus + NT - 1) / NT}));
    hipLaunchKernelGGL((k_sb_hemm<T, B>), dim3(NT, (parts + HemmCfg<T, B>::PPW - 1) / HemmCfg<T, B>::PPW), dim3(256), 0, st, A22, lda, mh, Zr, kb, parts, Wp);
    hipLaunchKernelGGL((k_sb_hemm_reduce<T, B>), dim3(nrb), dim3(256), lds_reduce, st, Wp, parts, mh, kb, Zb, Wb, (long)n, Mp);
    hipLaunchKernelGGL((k_sb_reduce<T>), red_grid, dim3(256), 0, st, Mp, nrb, B * B, ps->G, nullptr);
    hipLaunchKernelGGL((k_sb_x<T, B>), dim3(nrb), dim3(256), lds_x, st, Wb, Yb, (long)n, mh, kb, ps);
    hipLaunchKernelGGL((k_sb_her2k<T, B>), dim3(NT * (NT + 1) / 2), dim3(256), lds_her2k, st, A22, lda, mh, Wb, Yb, (long)n, kb);
    HIPCHK(ctx, hipGetLastError());
    j += kb;
  }
  if (want_stamps) {
    long long h[16];
    HIPCHK(ctx, hipMemcpyAsync(h, dstamps, sizeof(h), hipMemcpyDeviceToHost, st));
    HIPCHK(ctx, hipStreamSynchronize(st));
    std::fprintf(stderr, "[nls] k_sb_small_recon time line (us, 100 MHz clock; n = %d):", n);
    for (int i = 1; i <= 9; ++i) std::fprintf(stderr, " %.2f", h[i] ? (double)(h[i] - h[0]) * 0.01 : 0.0);
    std::fprintf(stderr, "\n");
  }
  *ncols_reduced = j;
  return NLS_OK;
}

// The chase reflectors live in a zero-padded array: v2_ld(n, B) rows (B extra: a reflector's B rows never run past the column) and
// v2_cols(n, B) columns (a whole number of sweep groups), so the second back-transformation loads whole blocks without bounds tests.
static inline long v2_ld(int n, int B) { return (long)n + B; }
static inline long v2_cols(int n, int B) { return (long)((n - 1 + B - 1) / B) * B + B; }

// Stage 2.  Band part of A (lower, bandwidth B) -> d[n], e[n-1], chase reflectors V2 (v2_ld x v2_cols, zeroed here).  ctl[1] != 0 afterwards: a workgroup timed out.
template <class T, int B>
static int sb2st(nls_ctx* ctx, const T* A, int n, long lda, double* d, double* e, T* V2, unsigned** ctl_out) {
  using namespace chase;
  const int ldab = 2 * B + 1;
  const long ldv = v2_ld(n, B);
  HIPCHK(ctx, hipMemsetAsync(V2, 0, sizeof(T) * (size_t)ldv * v2_cols(n, B), ctx->stream));
  T* AB = nullptr;
  unsigned* ctl = nullptr;
  NLSCHK(ws_get_t(ctx, "chase.AB", (size_t)n * ldab, &AB));
  NLSCHK(ws_get_t(ctx, "chase.ctl", (size_t)n + 16, &ctl));  // [0] next sweep, [1] error, [16 ..] done[s]
  HIPCHK(ctx, hipMemsetAsync(ctl, 0, sizeof(unsigned) * ((size_t)n + 16), ctx->stream));
  hipLaunchKernelGGL((sb::k_sb_to_band<T>), dim3((unsigned)(((long)n * ldab + 255) / 256)), dim3(256), 0, ctx->stream, A, lda, n, B, AB, ldab);
  const size_t lds = std::max(sizeof(ChaseLds<T, B>), (size_t)84 << 10);  // > 80 KiB: one workgroup per CU (hand-off protocol)
  NLSCHK(sb_lds_optin(ctx, k_chase<T, B>, lds, "k_chase"));
  int W = n / (2 * B) + 4;
  if (const char* ew = std::getenv("NLS_CHASE_WG")) W = std::atoi(ew);
  W = std::max(1, std::min(W, ctx->cus));
  long long* stamps = nullptr;
  static const bool want_stamps = [] { const char* m = std::getenv("NLS_CHASE_STAMP"); return m && m[0] == '1'; }();
  if (want_stamps) {
    NLSCHK(ws_get_t(ctx, "chase.stamps", (size_t)64, &stamps));
    HIPCHK(ctx, hipMemsetAsync(stamps, 0, 64 * sizeof(long long), ctx->stream));
  }
  if (n >= 2) hipLaunchKernelGGL((k_chase<T, B>), dim3(W), dim3(256), lds, ctx->stream, AB, ldab, n, V2, ldv, d, e, ctl, ctl + 16, stamps);
  HIPCHK(ctx, hipGetLastError());
  if (want_stamps) {  // diagnostic: time line of sweep 64 (100 MHz clock: 10 ns units)
    long long h[64];
    HIPCHK(ctx, hipMemcpyAsync(h, stamps, sizeof(h), hipMemcpyDeviceToHost, ctx->stream));
    HIPCHK(ctx, hipStreamSynchronize(ctx->stream));
    for (int q = 0; q < 8; ++q)
      std::fprintf(stderr, "[chase stamps] stage %d: wait %.2f us, load %.2f, P1-P3 %.2f, P4+store issue %.2f, drain %.2f, publish %.2f | next stage starts %.2f us later\n",
                   q + 1, (h[q * 8 + 1] - h[q * 8]) * 0.01, (h[q * 8 + 2] - h[q * 8 + 1]) * 0.01, (h[q * 8 + 3] - h[q * 8 + 2]) * 0.01,
                   (h[q * 8 + 4] - h[q * 8 + 3]) * 0.01, (h[q * 8 + 5] - h[q * 8 + 4]) * 0.01, (h[q * 8 + 6] - h[q * 8 + 5]) * 0.01,
                   q < 7 ? (h[(q + 1) * 8] - h[q * 8]) * 0.01 : 0.0);
  }
  *ctl_out = ctl;
  return NLS_OK;
}

// C (n x ncols, column-major ldc) <- Q2 C with the chase reflectors in V2.
template <class T, int B>
static int apply_q2(nls_ctx* ctx, const T* V2, int n, T* C, long ldc, int ncols) {
  using namespace q2;
  constexpr int NC = 16;
  const long ldv = v2_ld(n, B);
  if (n < 2 || ncols <= 0) return NLS_OK;
  const int ngroups = (n - 1 + B - 1) / B;
  std::vector<int> off((size_t)ngroups + 1, 0);
  for (int S = 0; S < ngroups; ++S) off[S + 1] = off[S] + q2_nblocks(n, B, S);
  const int nblocks = off[ngroups];
  int* doff = nullptr;
  T *Tb = nullptr, *Pk = nullptr;
  using L = Q2P<T, B>;
  static const bool valu_env = [] { const char* m = std::getenv("NLS_Q2_VALU"); return m && m[0] == '1'; }();  // the VALU form (reference; tests)
  const bool valu = valu_env || !L::AVAILABLE;
  NLSCHK(ws_get_t(ctx, "q2.off", (size_t)ngroups + 1, &doff));
  if (valu)
    NLSCHK(ws_get_t(ctx, "q2.T", (size_t)nblocks * B * B, &Tb));
  else
    NLSCHK(ws_get_t(ctx, "q2.P", (size_t)nblocks * L::PER_BLOCK, &Pk));
  HIPCHK(ctx, hipMemcpyAsync(doff, off.data(), sizeof(int) * off.size(), hipMemcpyHostToDevice, ctx->stream));
  HIPCHK(ctx, hipStreamSynchronize(ctx->stream));  // off is a local
  const size_t mat = (sizeof(T) * B * (B + 1) + 15) & ~(size_t)15, wmat = (sizeof(T) * B * (NC + 1) + 15) & ~(size_t)15;
  const size_t lds_t = (L::AVAILABLE ? 3 : 2) * sizeof(T) * B * (B + 1);
  NLSCHK(sb_lds_optin(ctx, k_q2_tfactor<T, B>, lds_t, "k_q2_tfactor"));
  hipLaunchKernelGGL((k_q2_tfactor<T, B>), dim3(nblocks), dim3(256), lds_t, ctx->stream, V2, ldv, n, doff, ngroups, Tb, Pk);
  // groups per pass: as many as the LDS ring allows (2 G block rows of the slab), at most 8
  int G = 8;
  if (const char* eg = std::getenv("NLS_Q2_GROUPS")) G = std::max(1, std::atoi(eg));
  if (!valu) {
    if constexpr (L::AVAILABLE) {
      // more slabs than CUs: two workgroups per CU (half the LDS each) cover each other's barriers and operand latencies
      // (three for real blocks of 32, whose kernel is light enough in registers: 84 VGPRs)
      const unsigned nwg = (unsigned)((ncols + 15) / 16);
      if constexpr (sizeof(T) == 8 && B == 32) {
        // real blocks of 32, NLS_Q2_FORM=wave: one wave per block + a mover wave (k_q2_apply_wave).  Built and measured in round 4: bit-identical
        // results, 117 against 108 ms at n = 10^4 (profiles/r04_q2_forms.md) - the two-waves-per-block form below stays the default.
        static const bool wave_form = [] { const char* m = std::getenv("NLS_Q2_FORM"); return m && std::string(m) == "wave"; }();
        if (wave_form) {
          const size_t lds_wave = Q2Wave::lds_bytes();
          NLSCHK(sb_lds_optin(ctx, k_q2_apply_wave, lds_wave, "k_q2_apply_wave"));
          long long* stamps = nullptr;
          static const bool want_stamps = [] { const char* m = std::getenv("NLS_Q2_STAMP"); return m && m[0] == '1'; }();
          if (want_stamps) {
            NLSCHK(ws_get_t(ctx, "q2.stamps", (size_t)384, &stamps));
            HIPCHK(ctx, hipMemsetAsync(stamps, 0, 384 * sizeof(long long), ctx->stream));
          }
          hipLaunchKernelGGL(k_q2_apply_wave, dim3(nwg), dim3(256), lds_wave, ctx->stream, Pk, doff, ngroups, n, C, ldc, ncols, stamps);
          HIPCHK(ctx, hipGetLastError());
          if (want_stamps) {
            long long h[384];
            HIPCHK(ctx, hipMemcpyAsync(h, stamps, sizeof(h), hipMemcpyDeviceToHost, ctx->stream));
            HIPCHK(ctx, hipStreamSynchronize(ctx->stream));
            for (int wsel = 0; wsel < 2; ++wsel)
              for (int pi = 0; pi + 1 < 120 && h[128 + 128 * wsel + pi + 1] != 0; pi += 8)
                std::fprintf(stderr, "[q2 wave stamps] workgroup %d pass %d: started %.1f us after pass 0, this pass %.1f us (%d steps)\n", wsel ? 300 : 0, pi,
                             (h[128 + 128 * wsel + pi] - h[128 + 128 * wsel]) * 0.01, (h[128 + 128 * wsel + pi + 1] - h[128 + 128 * wsel + pi]) * 0.01, 3 * pi + 5);
          }
          return NLS_OK;
        }
      }
      const unsigned per_cu = std::min<unsigned>((nwg + ctx->cus - 1) / ctx->cus, (sizeof(T) == 8 && B == 32) ? 3u : 2u);
      const size_t budget = per_cu >= 3 ? ((size_t)53 << 10) : per_cu == 2 ? ((size_t)80 << 10) : ((size_t)158 << 10);
      while (G > 1 && L::lds_bytes(G) > budget) --G;
      G = std::min(G, ngroups);
      NLSCHK(sb_lds_optin(ctx, k_q2_apply_packed<T, B>, L::lds_bytes(G), "k_q2_apply_packed"));
      long long* stamps = nullptr;
      static const bool want_stamps = [] { const char* m = std::getenv("NLS_Q2_STAMP"); return m && m[0] == '1'; }();
      if (want_stamps) {
        NLSCHK(ws_get_t(ctx, "q2.stamps", (size_t)64, &stamps));
        HIPCHK(ctx, hipMemsetAsync(stamps, 0, 64 * sizeof(long long), ctx->stream));
      }
      hipLaunchKernelGGL((k_q2_apply_packed<T, B>), dim3(nwg), dim3(256), L::lds_bytes(G), ctx->stream, Pk, doff, ngroups, n, C, ldc, ncols, G, stamps);
      HIPCHK(ctx, hipGetLastError());
      if (want_stamps) {
        long long h[64];
        HIPCHK(ctx, hipMemcpyAsync(h, stamps, sizeof(h), hipMemcpyDeviceToHost, ctx->stream));
        HIPCHK(ctx, hipStreamSynchronize(ctx->stream));
        for (int q = 0; q < 8; ++q)
          std::fprintf(stderr, "[q2 stamps G=%d] block %d: barrier + ring %.2f us, W1 %.2f, barrier %.2f, update %.2f | next block %.2f us later\n", G, 200 + q,
                       (h[q * 8 + 1] - h[q * 8]) * 0.01, (h[q * 8 + 2] - h[q * 8 + 1]) * 0.01, (h[q * 8 + 3] - h[q * 8 + 2]) * 0.01,
                       (h[q * 8 + 4] - h[q * 8 + 3]) * 0.01, q < 7 ? (h[(q + 1) * 8] - h[q * 8]) * 0.01 : 0.0);
      }
    }
    return NLS_OK;
  }
  auto lds_apply = [&](int g) { return ((sizeof(T) * (size_t)(2 * g) * B * (NC + 1) + 15) & ~(size_t)15) + 2 * mat + 2 * wmat; };
  while (G > 1 && lds_apply(G) > ((size_t)158 << 10)) --G;
  if (lds_apply(G) > ((size_t)158 << 10)) return fail(ctx, NLS_ERR_ARG, "second back-transformation: complex blocks of %d do not fit the LDS", B);
  G = std::min(G, ngroups);
  NLSCHK(sb_lds_optin(ctx, k_q2_apply<T, B, NC>, lds_apply(G), "k_q2_apply"));
  hipLaunchKernelGGL((k_q2_apply<T, B, NC>), dim3((unsigned)((ncols + NC - 1) / NC)), dim3(256), lds_apply(G), ctx->stream, V2, ldv, n, doff, ngroups, Tb, C,
                     ldc, ncols, G);
  HIPCHK(ctx, hipGetLastError());
  return NLS_OK;
}

// Invariants of an orthogonal / unitary similarity A -> T: trace and squared Frobenius norm.  The two-stage reduction is checked with them
// (evd_two_stage): the chase hands data between workgroups inside one launch, and a hand-off that went wrong must not become silently wrong
// eigenpairs.  One block per column of the stored lower triangle, per-column partials, then one block adds them in a fixed order.
template <class T>
__global__ void __launch_bounds__(256) k_herm_invariants(const T* A, long lda, int n, double* part) {
  using namespace trd;
  __shared__ double sh[4];
  const int c = blockIdx.x;
  double f = 0.0;
  for (long r = c + 1 + threadIdx.x; r < n; r += 256) f += 2.0 * abs2_(A[r + (long)c * lda]);
  f = wave_sum(f);
  if ((threadIdx.x & 63) == 0) sh[threadIdx.x >> 6] = f;
  __syncthreads();
  if (threadIdx.x == 0) {
    const double dg = real_(A[c + (long)c * lda]);
    part[2 * c] = dg;
    part[2 * c + 1] = ((sh[0] + sh[1]) + (sh[2] + sh[3])) + dg * dg;
  }
}
// out[0..1] = (trace, |.|_F^2) of the matrix whose column partials are `part` (skipped when part == nullptr), out[2..3] = the same of the
// symmetric tridiagonal matrix (d, e).
__global__ void __launch_bounds__(256) k_tridiag_invariants(const double* part, const double* d, const double* e, int n, double* out) {
  using namespace trd;
  __shared__ double sh[4][4];
  double v[4] = {0.0, 0.0, 0.0, 0.0};
  for (int i = threadIdx.x; i < n; i += 256) {
    if (part) {
      v[0] += part[2 * i];
      v[1] += part[2 * i + 1];
    }
    v[2] += d[i];
    v[3] += d[i] * d[i] + (i + 1 < n ? 2.0 * e[i] * e[i] : 0.0);
  }
  for (int k = 0; k < 4; ++k) {
    v[k] = wave_sum(v[k]);
    if ((threadIdx.x & 63) == 0) sh[threadIdx.x >> 6][k] = v[k];
  }
  __syncthreads();
  if (threadIdx.x < 4 && (part || threadIdx.x >= 2)) out[threadIdx.x] = (sh[0][threadIdx.x] + sh[1][threadIdx.x]) + (sh[2][threadIdx.x] + sh[3][threadIdx.x]);
}

__global__ void k_real_to_complex(const double* src, long n_elems, double2* dst) {
  const long i = blockIdx.x * (long)blockDim.x + threadIdx.x;
  if (i < n_elems) dst[i] = make_double2(src[i], 0.0);
}

// Band width of the two-stage reduction: NLS_SB_BW = 32 / 64 overrides the default of the arithmetic (TwoStageBw).
static int evd_bw(bool cplx) {
  if (cplx) return TwoStageBw<trd::Z>::B;  // complex blocks of 64 do not fit the LDS of the second back-transformation
  if (const char* e = std::getenv("NLS_SB_BW")) {
    const int b = std::atoi(e);
    if (b == 32 || b == 64) return b;
  }
  return TwoStageBw<double>::B;
}

// When to take the two-stage reduction.  Measured on MI355X in round 3 (profiles/r03_evd_stages.log):
//   real:    n = 4000: 139 against 128 ms one-stage, n = 5000: 135 against ~150, n = 6500: ~200 against ~250, n = 8000: 281 against 330, n = 10^4: 478 against 587
//            -> two-stage from n = 6000 (the one-stage panel streams the trailing matrix from HBM once per column as soon as its lower
//            triangle outgrows the 256 MB Infinity Cache);
//            round 5 (band reduction 117 -> 75 ms at 10^4; eigh incl. copies, tools/gpu_r05_cross.sh): n = 3000: 111 against 70 one-stage,
//            4000: 111 against 116, 5000: 163 against 178, 6000: 222 against 251 -> two-stage from n = 4500;
//   complex: n = 4097: 171 against 143, n = 1025: 27 against 18.5 -> one-stage.
// NLS_EVD=twostage forces it (n >= 4), NLS_EVD=onestage forbids it, NLS_TWOSTAGE_MIN = n moves the size rule (both arithmetics).
static bool evd_use_two_stage(int n, bool cplx) {
  const char* m = std::getenv("NLS_EVD");
  if (m && std::string(m) == "twostage") return n >= 4;
  if (m && std::string(m) == "onestage") return false;
  if (const char* e = std::getenv("NLS_TWOSTAGE_MIN")) return n >= std::max(4, std::atoi(e));
  return !cplx && n >= 4500;
}

static bool evd_rocsolver_backtransform() {  // NLS_EVD_UNMTR=rocsolver: zunmtr / dormtr instead of apply_q_blocked (diagnostic)
  const char* m = std::getenv("NLS_EVD_UNMTR");
  return m && std::string(m) == "rocsolver";
}

static bool evd_use_rocsolver() {
  const char* m = std::getenv("NLS_EVD");
  return m && std::string(m) == "rocsolver";
}

// LAPACK's zheev / dsyev scale the matrix when its largest entry leaves [sqrt(safmin / eps), sqrt(eps / safmin)]; the panel's
// larfg (nls_trd.h) has no safmin rescaling loop of its own, so a general-purpose nls_eigh_only relies on a scaling at
// the driver level too: then sums of squares of entries neither overflow nor lose more than entries already negligible
// against the matrix norm.  Returns the factor applied (1 = none); eigenvalues are divided by it afterwards.
__global__ void k_absmax_lower(const double* A, long lda_d, int n, int comps, unsigned long long* out) {
  // A viewed as doubles; comps = 2 for complex.  One block per column.
  const int c = blockIdx.x;
  double m = 0.0;
  for (long r = c + threadIdx.x; r < n; r += blockDim.x)
    for (int k = 0; k < comps; ++k) {
      const double v = fabs(A[(long)c * lda_d + r * comps + k]);
      m = (v > m || v != v) ? v : m;  // NaN propagates
    }
  for (int o = 32; o > 0; o >>= 1) {
    const double t = __shfl_xor(m, o, 64);
    m = (t > m || t != t) ? t : m;
  }
  if ((threadIdx.x & 63) == 0) atomicMax(out, (unsigned long long)__double_as_longlong(m));  // non-negative doubles order like integers
}
__global__ void k_scale_lower(double* A, long lda_d, int n, int comps, double f) {
  const int c = blockIdx.x;
  for (long r = c + threadIdx.x; r < n; r += blockDim.x)
    for (int k = 0; k < comps; ++k) A[(long)c * lda_d + r * comps + k] *= f;
}
static int evd_prescale(nls_ctx* ctx, void* A, int n, int comps, double* factor) {
  unsigned long long* slot = nullptr;
  NLSCHK(ws_get_t(ctx, "evd.absmax", 1, &slot));
  HIPCHK(ctx, hipMemsetAsync(slot, 0, sizeof(*slot), ctx->stream));
  hipLaunchKernelGGL(k_absmax_lower, dim3((unsigned)n), dim3(256), 0, ctx->stream, static_cast<const double*>(A), (long)n * comps, n, comps, slot);
  HIPCHK(ctx, hipGetLastError());
  unsigned long long bits = 0;
  HIPCHK(ctx, hipMemcpyAsync(&bits, slot, sizeof(bits), hipMemcpyDeviceToHost, ctx->stream));
  HIPCHK(ctx, hipStreamSynchronize(ctx->stream));
  double anrm;
  std::memcpy(&anrm, &bits, sizeof(anrm));
  if (!std::isfinite(anrm)) return fail(ctx, NLS_ERR_LINALG, "eigendecomposition: the matrix contains NaN or Inf");
  // Outside [1e-100, 1e100] scale by the power of two that brings the largest entry to [1, 2): exact, and everything
  // downstream (sums of squares in larfg, rocSOLVER's stedc) sees an O(1) matrix.
  *factor = 1.0;
  if (anrm > 0.0 && (anrm < 1e-100 || anrm > 1e100)) *factor = std::ldexp(1.0, -std::ilogb(anrm));
  if (*factor != 1.0) {
    hipLaunchKernelGGL(k_scale_lower, dim3((unsigned)n), dim3(256), 0, ctx->stream, static_cast<double*>(A), (long)n * comps, n, comps, *factor);
    HIPCHK(ctx, hipGetLastError());
  }
  return NLS_OK;
}
__global__ void k_vec_scale(double* v, int n, double f) {
  const int i = blockIdx.x * blockDim.x + threadIdx.x;
  if (i < n) v[i] *= f;
}
static int evd_unscale(nls_ctx* ctx, double* lam, int n, double factor) {
  if (factor == 1.0) return NLS_OK;
  hipLaunchKernelGGL(k_vec_scale, dim3((unsigned)((n + 255) / 256)), dim3(256), 0, ctx->stream, lam, n, 1.0 / factor);
  HIPCHK(ctx, hipGetLastError());
  return NLS_OK;
}

// The library's own divide and conquer (nls_stedc.h): d (n: diagonal in, eigenvalues ascending out), e (n - 1: off-diagonal, read only),
// C (n x n column-major, leading dimension n: eigenvectors out).  A fixed launch sequence, no host synchronisation.  dinfo: raised (leaf index + 1)
// when a leaf's QL iteration does not converge.
static int stedc_dc(nls_ctx* ctx, int n, double* d, const double* e, double* C, rocblas_int* dinfo) {
  using namespace dc;
  hipStream_t st = ctx->stream;
  HIPCHK(ctx, hipMemsetAsync(dinfo, 0, sizeof(rocblas_int), st));
  if (n == 1) {
    const double one = 1.0;
    HIPCHK(ctx, hipMemcpyAsync(C, &one, sizeof(double), hipMemcpyHostToDevice, st));
    HIPCHK(ctx, hipStreamSynchronize(st));
    return NLS_OK;
  }
  int nlev = 0;
  size_t slot = 0;
  for (long S = LEAF; S < n; S *= 2) {
    ++nlev;
    const long nm = (n + 2 * S - 1) / (2 * S), P = round_up(std::min<long>(2 * S, n), BM);
    slot = std::max(slot, (size_t)(nm * P * P));
  }
  const int nm_max = (n + 2 * LEAF - 1) / (2 * LEAF);
  double *lam2 = nullptr, *Q2 = nullptr, *ds = nullptr, *zs = nullptr, *dl = nullptr, *w = nullptr, *dorg = nullptr, *mu = nullptr, *lamnew = nullptr, *zhat = nullptr;
  double *U = nullptr, *G = nullptr, *R = nullptr;
  int *src = nullptr, *kidx = nullptr, *didx = nullptr, *pos = nullptr, *gpos = nullptr;
  Rot* rots = nullptr;
  MergeInfo* info = nullptr;
  NLSCHK(ws_get_t(ctx, "dc.lam2", (size_t)n, &lam2));
  NLSCHK(ws_get_t(ctx, "dc.ds", (size_t)n, &ds));
  NLSCHK(ws_get_t(ctx, "dc.zs", (size_t)n, &zs));
  NLSCHK(ws_get_t(ctx, "dc.dl", (size_t)n, &dl));
  NLSCHK(ws_get_t(ctx, "dc.w", (size_t)n, &w));
  NLSCHK(ws_get_t(ctx, "dc.dorg", (size_t)n, &dorg));
  NLSCHK(ws_get_t(ctx, "dc.mu", (size_t)n, &mu));
  NLSCHK(ws_get_t(ctx, "dc.lamnew", (size_t)n, &lamnew));
  NLSCHK(ws_get_t(ctx, "dc.zhat", (size_t)n, &zhat));
  NLSCHK(ws_get_t(ctx, "dc.src", (size_t)n, &src));
  NLSCHK(ws_get_t(ctx, "dc.kidx", (size_t)n, &kidx));
  NLSCHK(ws_get_t(ctx, "dc.didx", (size_t)n, &didx));
  NLSCHK(ws_get_t(ctx, "dc.pos", (size_t)n, &pos));
  NLSCHK(ws_get_t(ctx, "dc.gpos", (size_t)n, &gpos));
  NLSCHK(ws_get_t(ctx, "dc.rots", (size_t)n, &rots));
  NLSCHK(ws_get_t(ctx, "dc.info", (size_t)nm_max, &info));
  if (nlev > 0) {
    NLSCHK(ws_get_t(ctx, "dc.Q2", (size_t)n * n, &Q2));
    NLSCHK(ws_get_t(ctx, "dc.U", slot, &U));
    NLSCHK(ws_get_t(ctx, "dc.G", slot, &G));
    NLSCHK(ws_get_t(ctx, "dc.R", slot, &R));
    if (hipFuncSetAttribute(reinterpret_cast<const void*>(k_dc_gemm), hipFuncAttributeMaxDynamicSharedMemorySize, (int)(2 * 2 * TILE_DOUBLES * sizeof(double))) != hipSuccess)
      return fail(ctx, NLS_ERR_HIP, "k_dc_gemm: dynamic LDS refused");
  }
  // Ping-pong between C and Q2 so that the last level writes C.  Both start as zero: a level reads the whole square of a merge, whose
  // off-diagonal blocks nobody has written.
  double* Qbuf[2] = {(nlev % 2 == 0) ? C : Q2, (nlev % 2 == 0) ? Q2 : C};
  double* Lbuf[2] = {(nlev % 2 == 0) ? d : lam2, (nlev % 2 == 0) ? lam2 : d};
  HIPCHK(ctx, hipMemsetAsync(C, 0, sizeof(double) * (size_t)n * n, st));
  if (nlev > 0) HIPCHK(ctx, hipMemsetAsync(Q2, 0, sizeof(double) * (size_t)n * n, st));
  // (the leaves read d: when the eigenvalues start in d itself, read a copy)
  double* dsrc = d;
  if (Lbuf[0] == d) {
    HIPCHK(ctx, hipMemcpyAsync(lam2, d, sizeof(double) * n, hipMemcpyDeviceToDevice, st));
    dsrc = lam2;
  }
  hipLaunchKernelGGL(k_dc_leaf, dim3((unsigned)((n + LEAF - 1) / LEAF)), dim3(64), 0, st, n, dsrc, e, Lbuf[0], Qbuf[0], (long)n, reinterpret_cast<int*>(dinfo));
  HIPCHK(ctx, hipGetLastError());
  int cur = 0;
  for (long S = LEAF; S < n; S *= 2) {
    Level L{n, (int)S, (long)round_up(std::min<long>(2 * S, n), BM)};
    const unsigned nm = (unsigned)((n + 2 * S - 1) / (2 * S));
    const unsigned mmax = (unsigned)std::min<long>(2 * S, n);
    const double* Qc = Qbuf[cur];
    double* Qn = Qbuf[cur ^ 1];
    hipLaunchKernelGGL(k_dc_setup, dim3(nm), dim3(256), 0, st, L, Lbuf[cur], Qc, (long)n, e, ds, zs, src, dl, w, kidx, didx, gpos, rots, info);
    hipLaunchKernelGGL(k_dc_rotate, dim3((mmax + 255) / 256, nm), dim3(256), 0, st, L, Qbuf[cur], (long)n, rots, info);
    hipLaunchKernelGGL(k_dc_gather, dim3((unsigned)((L.P + 255) / 256), (mmax + 15) / 16, nm), dim3(256), 0, st, L, Qc, (long)n, src, kidx, gpos, info, G);
    hipLaunchKernelGGL(k_dc_secular, dim3((mmax + 4 * RW - 1) / (4 * RW), nm), dim3(256), 0, st, L, dl, w, info, dorg, mu, lamnew);
    hipLaunchKernelGGL(k_dc_zhat, dim3((mmax + 3) / 4, nm), dim3(256), 0, st, L, dl, w, dorg, mu, info, zhat);
    hipLaunchKernelGGL(k_dc_vectors, dim3(mmax, nm), dim3(256), 0, st, L, dl, zhat, dorg, mu, gpos, info, U);
    hipLaunchKernelGGL(k_dc_gemm, dim3((unsigned)(L.P / BN), (unsigned)(L.P / BM), nm), dim3(Cfg4::NTHREADS), 2 * 2 * TILE_DOUBLES * sizeof(double), st, L, U, G, R, info);
    hipLaunchKernelGGL(k_dc_place, dim3((mmax + 255) / 256, nm), dim3(256), 0, st, L, lamnew, ds, didx, info, pos, Lbuf[cur ^ 1]);
    hipLaunchKernelGGL(k_dc_scatter, dim3((mmax + 255) / 256, (mmax + 7) / 8, nm), dim3(256), 0, st, L, R, Qc, Qn, (long)n, src, didx, pos, info);
    HIPCHK(ctx, hipGetLastError());
    cur ^= 1;
  }
  return NLS_OK;
}

static bool stedc_use_rocsolver() {  // NLS_STEDC=rocsolver: rocsolver_dstedc instead of the library's own divide and conquer (diagnostic)
  const char* m = std::getenv("NLS_STEDC");
  return m && std::string(m) == "rocsolver";
}
static int stedc_any(nls_ctx* ctx, int n, double* lam, double* e_work, double* Cr, rocblas_int* dinfo) {
  if (stedc_use_rocsolver()) {
    BLASCHK(ctx, rocsolver_dstedc(ctx->blas, rocblas_evect_tridiagonal, n, lam, e_work, Cr, n, dinfo));
    return NLS_OK;
  }
  return stedc_dc(ctx, n, lam, e_work, Cr, dinfo);
}

// Measurement hook (nls_comm_set_virtual_rank): the eigenvector blocks of the OTHER virtual ranks.  A real peer would have sent them; here
// they come from the copy a complete call of the same problem left in "virt.Q" (the timed steps of a bench repeat one problem) - a device
// copy standing in for the arrival of the all-gather.  Without such a copy (virt_n != n) this returns NLS_ERR_ARG: the caller runs one
// complete fit first (nls_comm_set_virtual_rank(ctx, 0, 1, capture = 1)).
static int virtual_other_blocks(nls_ctx* ctx, double2* C, int n, long c0, long c1) {
  if (ctx->virt_capture) {
    double2* Qfull = nullptr;
    NLSCHK(ws_get_t(ctx, "virt.Q", (size_t)n * n, &Qfull));
    HIPCHK(ctx, hipMemcpyAsync(Qfull, C, sizeof(double2) * (size_t)n * n, hipMemcpyDeviceToDevice, ctx->stream));
    ctx->virt_n = n;
    return NLS_OK;
  }
  if (ctx->virt_world <= 1) return NLS_OK;
  if (ctx->virt_n != n) return fail(ctx, NLS_ERR_ARG, "virtual rank: no captured eigenvectors of order %d (run one complete fit with capture first)", n);
  double2* Qfull = nullptr;
  NLSCHK(ws_get_t(ctx, "virt.Q", (size_t)n * n, &Qfull));
  if (c0 > 0) HIPCHK(ctx, hipMemcpyAsync(C, Qfull, sizeof(double2) * (size_t)c0 * n, hipMemcpyDeviceToDevice, ctx->stream));
  if (c1 < n) HIPCHK(ctx, hipMemcpyAsync(C + c1 * n, Qfull + c1 * n, sizeof(double2) * (size_t)(n - c1) * n, hipMemcpyDeviceToDevice, ctx->stream));
  return NLS_OK;
}

// stedc on the (real) tridiagonal matrix.  collective: rank 0 computes, everybody receives (lam, Cr) - the ranks then pair the same
// eigenvalues with the same basis whatever stedc does on clustered spectra.  rc_in: the status of this rank's work since the previous
// exchange (the tridiagonalisation); together with rank 0's solver status it goes into the vote that guards the two broadcasts, so a
// failure anywhere - an API error or info != 0 on rank 0 included - is an error on every rank instead of a rank missing from a collective.
static int stedc_real(nls_ctx* ctx, int n, double* lam, double* e_work, double* Cr, rocblas_int* dinfo, bool collective, int rc_in = NLS_OK) {
  if (!(collective && multi_rank(ctx))) {
    NLSCHK(rc_in);
    NLSCHK(stedc_any(ctx, n, lam, e_work, Cr, dinfo));
    return check_info(ctx, dinfo, "tridiagonal eigensolver (stedc)");
  }
  int rc = rc_in;
  if (rc == NLS_OK && work_rank(ctx) == 0) rc = [&]() -> int {
    NLSCHK(stedc_any(ctx, n, lam, e_work, Cr, dinfo));
    return check_info(ctx, dinfo, "tridiagonal eigensolver (stedc) on rank 0");
  }();
  NLSCHK(comm_vote(ctx, rc, "at the tridiagonal eigensolver (rank 0 solves, every rank receives)"));
  NLSCHK(do_broadcast(ctx, lam, (size_t)n, 0));
  NLSCHK(do_broadcast(ctx, Cr, (size_t)n * n, 0));
  if (ctx->virt_capture || ctx->virt_world > 1) {  // measurement hook (nls_comm_set_virtual_rank): what rank 0 would have sent
    double *vlam = nullptr, *vCr = nullptr;
    NLSCHK(ws_get_t(ctx, "virt.lam", (size_t)n, &vlam));
    NLSCHK(ws_get_t(ctx, "virt.Cr", (size_t)n * n, &vCr));
    if (ctx->virt_capture) {
      HIPCHK(ctx, hipMemcpyAsync(vlam, lam, sizeof(double) * n, hipMemcpyDeviceToDevice, ctx->stream));
      HIPCHK(ctx, hipMemcpyAsync(vCr, Cr, sizeof(double) * (size_t)n * n, hipMemcpyDeviceToDevice, ctx->stream));
    } else if (work_rank(ctx) != 0) {
      if (ctx->virt_n != n) return fail(ctx, NLS_ERR_ARG, "virtual rank: no captured eigenpairs of order %d (run one complete fit with capture first)", n);
      HIPCHK(ctx, hipMemcpyAsync(lam, vlam, sizeof(double) * n, hipMemcpyDeviceToDevice, ctx->stream));
      HIPCHK(ctx, hipMemcpyAsync(Cr, vCr, sizeof(double) * (size_t)n * n, hipMemcpyDeviceToDevice, ctx->stream));
    }
  }
  return NLS_OK;
}

// Eigendecomposition through the two-stage reduction.  A (n x n, lower) is destroyed; eigenvectors in C (n x n, distinct from A).
// *used = false: the band reduction met a panel it could not orthogonalise (exactly dependent / zero columns); A has been restored
// and the caller takes the one-stage path.  collective: see evd_hermitian_core.
template <class T, int B>
static int evd_two_stage(nls_ctx* ctx, T* A, int n, double* lam, double* e_work, rocblas_int* dinfo, T* C, bool collective, bool* used, int rc_in = NLS_OK,
                         int* carry = nullptr) {
  constexpr bool CPLX = sizeof(T) == 16;
  *used = false;
  if (carry) *carry = NLS_OK;  // sharded, *used == false: the status this rank takes into the one-stage path's first vote
  const bool shared = collective && multi_rank(ctx);  // sharded fit: no early return between exchanges - the status goes into the next vote (comm_vote)
  if (!shared) NLSCHK(rc_in);
  T *Acopy = nullptr, *tau1 = nullptr, *V2 = nullptr;
  int* dflag = nullptr;
  int nred = 0;
  bool fell_back = false, bad = false;
  static const bool prof = [] { const char* m = std::getenv("NLS_EVD_PROFILE"); return m && m[0] == '1'; }();
  auto mark = [&](int i) { evd_mark(ctx, i); };
  int rc = rc_in;
  if (rc == NLS_OK) rc = [&]() -> int {
  NLSCHK(ws_get_t(ctx, CPLX ? "evd2.Acopy" : "evd2.Acopy_r", (size_t)n * n, &Acopy));
  NLSCHK(ws_get_t(ctx, CPLX ? "evd2.tau1" : "evd2.tau1_r", (size_t)n, &tau1));
  NLSCHK(ws_get_t(ctx, CPLX ? "evd2.V2" : "evd2.V2_r", (size_t)v2_ld(n, B) * v2_cols(n, B), &V2));
  NLSCHK(ws_get_t(ctx, "evd2.flag", 4, &dflag));
  HIPCHK(ctx, hipMemcpyAsync(Acopy, A, sizeof(T) * (size_t)n * n, hipMemcpyDeviceToDevice, ctx->stream));
  HIPCHK(ctx, hipMemsetAsync(dflag, 0, 4 * sizeof(int), ctx->stream));
  unsigned* ctl = nullptr;
  // stage times: nls_evd_stage_ms; NLS_EVD_PROFILE=1 also prints one line on stderr per eigendecomposition
  ctx->evd_kind = 0;
  mark(0);
  // A panel whose columns are dependent to working precision raises the flag (nls_sb.h).  Second attempt: the saved copy again, every
  // panel perturbed by 1e-13 of its norm (k_sb_perturb); if that fails too (zero panels: a diagonal matrix) the one-stage panel takes over.
  unsigned hctl[2] = {0, 0};
  double *inv_part = nullptr, *inv = nullptr;  // invariants of the input and of the tridiagonal matrix (k_herm_invariants)
  NLSCHK(ws_get_t(ctx, "evd2.invpart", (size_t)2 * n, &inv_part));
  NLSCHK(ws_get_t(ctx, "evd2.inv", 4, &inv));
  hipLaunchKernelGGL(k_herm_invariants<T>, dim3((unsigned)n), dim3(256), 0, ctx->stream, Acopy, (long)n, n, inv_part);
  HIPCHK(ctx, hipGetLastError());
  double hinv[4] = {0.0, 0.0, 0.0, 0.0};
  for (int attempt = 0;; ++attempt) {
    NLSCHK((sy2sb<T, B>(ctx, A, n, n, tau1, dflag, &nred, attempt == 1)));
    if (attempt == 0) mark(1);
    NLSCHK((sb2st<T, B>(ctx, A, n, n, lam, e_work, V2, &ctl)));
    if (attempt == 0) mark(2);
    hipLaunchKernelGGL(k_tridiag_invariants, dim3(1), dim3(256), 0, ctx->stream, attempt == 0 ? inv_part : (const double*)nullptr, lam, e_work, n, inv);
    HIPCHK(ctx, hipGetLastError());
    int hflag = 0;
    HIPCHK(ctx, hipMemcpyAsync(&hflag, dflag, sizeof(int), hipMemcpyDeviceToHost, ctx->stream));
    HIPCHK(ctx, hipMemcpyAsync(hctl, ctl, sizeof(hctl), hipMemcpyDeviceToHost, ctx->stream));
    HIPCHK(ctx, hipMemcpyAsync(hinv, inv, sizeof(hinv), hipMemcpyDeviceToHost, ctx->stream));
    HIPCHK(ctx, hipStreamSynchronize(ctx->stream));
    if (hflag == 0) {
      if (attempt == 1) ctx->twostage_rescues++;
      break;
    }
    HIPCHK(ctx, hipMemcpyAsync(A, Acopy, sizeof(T) * (size_t)n * n, hipMemcpyDeviceToDevice, ctx->stream));
    if (attempt == 1) {
      ctx->twostage_fallbacks++;
      fell_back = true;  // (the same matrix on every rank of a sharded fit: the same decision)
      return NLS_OK;
    }
    HIPCHK(ctx, hipMemsetAsync(dflag, 0, 4 * sizeof(int), ctx->stream));
  }
  // The chase hands band data between workgroups of one launch (sc1 stores / loads and a progress word, nls_chase.h).  Two things can go wrong
  // there and neither may reach the caller as eigenpairs: a workgroup that gave up waiting (hctl[1]: the GPU was shared or pre-empted) and - not
  // observed, but outside what the compiler's memory model promises - a stale read.  A unitary similarity keeps trace and Frobenius norm:
  // |sum d - tr A| and | |T|_F^2 - |A|_F^2 | beyond 1e-10 relative (rounding: ~sqrt(n) eps) reject the reduction.  Either way A is restored from
  // the saved copy and the one-stage panel takes over (counted in nls_twostage_fallbacks).  In a collective fit the ranks decide together.
  {
    const double scale = std::max(std::sqrt(std::fabs(hinv[1])) * std::sqrt((double)n), std::fabs(hinv[0]));
    bad = hctl[1] != 0 || !std::isfinite(hinv[2]) || !std::isfinite(hinv[3]) || std::fabs(hinv[2] - hinv[0]) > 1e-10 * scale ||
          std::fabs(hinv[3] - hinv[1]) > 1e-10 * std::fabs(hinv[1]);
    if (const char* inj = std::getenv("NLS_CHASE_INJECT_FAILURE"))  // test hook: pretend the check failed
      if (inj[0] == '1') bad = true;
  }
  return NLS_OK;
  }();
  if (shared) {
    NLSCHK(comm_vote(ctx, rc, "after the reduction to tridiagonal form"));
    if (!fell_back) {  // the ranks decide together about the chase's invariants (the vote above guards this exchange)
      double* vote = nullptr;
      NLSCHK(ws_get_t(ctx, "evd2.vote", 2, &vote));
      double hv = bad ? 1.0 : 0.0;
      HIPCHK(ctx, hipMemcpyAsync(vote, &hv, sizeof(double), hipMemcpyHostToDevice, ctx->stream));
      HIPCHK(ctx, hipStreamSynchronize(ctx->stream));
      NLSCHK(do_allreduce(ctx, vote, 1));
      HIPCHK(ctx, hipMemcpyAsync(&hv, vote, sizeof(double), hipMemcpyDeviceToHost, ctx->stream));
      HIPCHK(ctx, hipStreamSynchronize(ctx->stream));
      bad = hv != 0.0;
    }
  } else {
    NLSCHK(rc);
  }
  if (fell_back) return NLS_OK;
  double* Cr = nullptr;
  long c0 = 0, c1 = n;
  const bool split = shared;
  rc = NLS_OK;
  if (bad) {
    rc = hipMemcpyAsync(A, Acopy, sizeof(T) * (size_t)n * n, hipMemcpyDeviceToDevice, ctx->stream) == hipSuccess
             ? NLS_OK : fail(ctx, NLS_ERR_HIP, "restoring the matrix after a rejected two-stage reduction failed");
    ctx->twostage_fallbacks++;
    if (!shared || !carry) return rc;
    *carry = rc;
    return NLS_OK;
  }
  rc = [&]() -> int {
    if (CPLX)
      NLSCHK(ws_get_t(ctx, "evd2.Cr", (size_t)n * n, &Cr));
    else
      Cr = reinterpret_cast<double*>(C);
    return NLS_OK;
  }();
  NLSCHK(stedc_real(ctx, n, lam, e_work, Cr, dinfo, collective, rc));
  mark(3);
  if (split) {
    c0 = (long)n * work_rank(ctx) / work_world(ctx);
    c1 = (long)n * (work_rank(ctx) + 1) / work_world(ctx);
  }
  rc = [&]() -> int {
    if (CPLX && c1 > c0) {
      const long cnt = (c1 - c0) * n;
      hipLaunchKernelGGL(k_real_to_complex, dim3((unsigned)((cnt + 255) / 256)), dim3(256), 0, ctx->stream, Cr + c0 * n, cnt,
                         reinterpret_cast<double2*>(C) + c0 * n);
      HIPCHK(ctx, hipGetLastError());
    }
    NLSCHK((apply_q2<T, B>(ctx, V2, n, C + c0 * n, n, (int)(c1 - c0))));
    mark(4);
    NLSCHK(apply_q_blocked<T>(ctx, A, n, n, tau1, C + c0 * n, n, (int)(c1 - c0), B, nred));
    mark(5);
    return fault_point(ctx, "backtransform");
  }();
  if (!split) NLSCHK(rc);
  ctx->evd_kind = CPLX ? 4 : 3;
  ctx->evd_n = n;
  if (prof) {
    (void)hipStreamSynchronize(ctx->stream);
    float t[5] = {0, 0, 0, 0, 0};
    for (int i = 0; i < 5; ++i) (void)hipEventElapsedTime(&t[i], ctx->evd_ev[i], ctx->evd_ev[i + 1]);
    std::fprintf(stderr, "[nls evd two-stage] n=%d %s bw=%d cols=%ld: band %.3f ms, chase %.3f, stedc %.3f, Q2 %.3f, Q1 %.3f, total %.3f\n", n,
                 CPLX ? "complex" : "real", B, c1 - c0, t[0], t[1], t[2], t[3], t[4], t[0] + t[1] + t[2] + t[3] + t[4]);
  }
  if (split) {
    NLSCHK(comm_vote(ctx, rc, "before the all-gather of the eigenvector blocks"));
    const int W = work_world(ctx);
    std::vector<size_t> offs((size_t)W + 1);
    const size_t comps = CPLX ? 2 : 1;
    for (int r = 0; r <= W; ++r) offs[r] = comps * n * (size_t)((long)n * r / W);
    NLSCHK(do_allgather_blocks(ctx, reinterpret_cast<double*>(C), offs));
    if (CPLX) NLSCHK(virtual_other_blocks(ctx, reinterpret_cast<double2*>(C), n, c0, c1));
  }
  *used = true;
  return NLS_OK;
}

// Hermitian: A (n x n complex column-major, lower) is destroyed; eigenvalues ascending in lam, eigenvectors
// (columns) in *Q, which is either A itself (rocSOLVER path) or the workspace "evd.C".
// collective = true (nls_primal_fit, where every rank holds the same all-reduced matrix): stedc on rank 0 + broadcast,
// back-transformation split by columns over the ranks, blocks all-gathered (Q bit-identical everywhere).
static int evd_hermitian_core(nls_ctx* ctx, double2* A, int n, double* lam, double* e_work, rocblas_int* dinfo, double2** Q, bool collective, int rc_in) {
  const bool shared = collective && multi_rank(ctx);  // sharded fit: local failures travel to the next vote instead of returning (comm_vote)
  if (!shared) NLSCHK(rc_in);
  BLASCHK(ctx, rocblas_set_stream(ctx->blas, ctx->stream));
  if (evd_use_rocsolver() || n < 3) {
    if (shared) NLSCHK(comm_vote(ctx, rc_in, "before the eigendecomposition"));  // (replicated on every rank: no exchange inside)
    for (int i = 0; i < 5; ++i) evd_mark(ctx, i);
    BLASCHK(ctx, rocsolver_zheevd(ctx->blas, rocblas_evect_original, rocblas_fill_lower, n, reinterpret_cast<rocblas_double_complex*>(A), n, lam,
                                  e_work, dinfo));
    evd_mark(ctx, 5);
    ctx->evd_kind = 5;
    ctx->evd_n = n;
    NLSCHK(check_info(ctx, dinfo, "rocsolver_zheevd"));
    *Q = A;
    return NLS_OK;
  }
  trd::Z* tau = nullptr;
  double2* C = nullptr;
  int rc = rc_in;
  if (rc == NLS_OK) rc = [&]() -> int {
    NLSCHK(ws_get_t(ctx, "evd.tau", (size_t)n, &tau));
    NLSCHK(ws_get_t(ctx, "evd.C", (size_t)n * n, &C));
    return NLS_OK;
  }();
  if (!shared) NLSCHK(rc);
  if (evd_use_two_stage(n, true)) {
    bool used = false;
    int carry = NLS_OK;
    NLSCHK((evd_two_stage<trd::Z, 32>(ctx, reinterpret_cast<trd::Z*>(A), n, lam, e_work, dinfo, reinterpret_cast<trd::Z*>(C), collective, &used, rc, &carry)));
    if (used) {
      *Q = C;
      return NLS_OK;
    }
    rc = carry;  // (sharded: everybody passed the two-stage path's vote, so only a failure after it is still travelling)
  }
  double* Cr = nullptr;
  if (rc == NLS_OK) rc = [&]() -> int {
    ctx->evd_kind = 0;
    evd_mark(ctx, 0);
    NLSCHK(trd_fused<trd::Z>(ctx, reinterpret_cast<trd::Z*>(A), n, n, lam, e_work, tau));
    evd_mark(ctx, 1);
    evd_mark(ctx, 2);
    return ws_get_t(ctx, "evd2.Cr", (size_t)n * n, &Cr);
  }();
  // The tridiagonal matrix of a Hermitian matrix is REAL (LAPACK convention: the phases live in the reflectors), so its eigenvectors
  // come from dstedc (15 instead of zstedc's 22 ms at n = 4097) and are widened to complex only for the back-transformation.
  // collective: every rank holds the same all-reduced matrix and the tridiagonalisation is bit-reproducible, so the reflectors are
  // replicated.  The tridiagonal eigensolver runs on rank 0 ONLY and (lam, the real eigenvectors: 134 instead of 268 MB) are broadcast:
  // all ranks then pair the same eigenvalues with the same basis whatever rocSOLVER's stedc does on clustered spectra (a failure on
  // rank 0 travels through the broadcast flag: stedc_real).  The back-transformation is split by columns over the ranks and the blocks
  // are all-gathered.
  const bool split = shared && n >= 64;
  if (shared && !split) {  // tiny matrices: replicated, no exchange - but the status still has to meet the other ranks'
    NLSCHK(comm_vote(ctx, rc, "before the tridiagonal eigensolver"));
    rc = NLS_OK;
  }
  NLSCHK(stedc_real(ctx, n, lam, e_work, Cr, dinfo, split, rc));  // (sharded: the vote inside takes rc; else rc is NLS_OK here)
  evd_mark(ctx, 3);
  evd_mark(ctx, 4);
  long c0 = 0, c1 = n;
  if (split) {
    c0 = (long)n * work_rank(ctx) / work_world(ctx);
    c1 = (long)n * (work_rank(ctx) + 1) / work_world(ctx);
  }
  rc = [&]() -> int {
    if (c1 > c0) {
      const long cnt = (c1 - c0) * n;
      hipLaunchKernelGGL(k_real_to_complex, dim3((unsigned)((cnt + 255) / 256)), dim3(256), 0, ctx->stream, Cr + c0 * n, cnt, C + c0 * n);
      HIPCHK(ctx, hipGetLastError());
    }
    if (evd_rocsolver_backtransform()) {
      BLASCHK(ctx, rocsolver_zunmtr(ctx->blas, rocblas_side_left, rocblas_fill_lower, rocblas_operation_none, n, (rocblas_int)(c1 - c0),
                                    reinterpret_cast<rocblas_double_complex*>(A), n, reinterpret_cast<rocblas_double_complex*>(tau),
                                    reinterpret_cast<rocblas_double_complex*>(C + c0 * n), n));
    } else {
      NLSCHK(apply_q_blocked<trd::Z>(ctx, reinterpret_cast<trd::Z*>(A), n, n, tau, reinterpret_cast<trd::Z*>(C + c0 * n), n, (int)(c1 - c0)));
    }
    evd_mark(ctx, 5);
    ctx->evd_kind = 2;
    ctx->evd_n = n;
    return fault_point(ctx, "backtransform");
  }();
  if (split) {
    NLSCHK(comm_vote(ctx, rc, "before the all-gather of the eigenvector blocks"));
    const int W = work_world(ctx);
    std::vector<size_t> offs((size_t)W + 1);
    for (int r = 0; r <= W; ++r) offs[r] = (size_t)2 * n * (size_t)((long)n * r / W);
    NLSCHK(do_allgather_blocks(ctx, reinterpret_cast<double*>(C), offs));
    NLSCHK(virtual_other_blocks(ctx, C, n, c0, c1));
  } else {
    NLSCHK(rc);
  }
  *Q = C;
  return NLS_OK;
}

static int evd_symmetric_core(nls_ctx* ctx, double* A, int n, double* lam, double* e_work, rocblas_int* dinfo, double** Q) {
  BLASCHK(ctx, rocblas_set_stream(ctx->blas, ctx->stream));
  if (evd_use_rocsolver() || n < 3) {
    for (int i = 0; i < 5; ++i) evd_mark(ctx, i);
    BLASCHK(ctx, rocsolver_dsyevd(ctx->blas, rocblas_evect_original, rocblas_fill_lower, n, A, n, lam, e_work, dinfo));
    evd_mark(ctx, 5);
    ctx->evd_kind = 5;
    ctx->evd_n = n;
    NLSCHK(check_info(ctx, dinfo, "rocsolver_dsyevd"));
    *Q = A;
    return NLS_OK;
  }
  double *tau = nullptr, *C = nullptr;
  NLSCHK(ws_get_t(ctx, "evd.taur", (size_t)n, &tau));
  NLSCHK(ws_get_t(ctx, "evd.Cr", (size_t)n * n, &C));
  if (evd_use_two_stage(n, false)) {
    bool used = false;
    if (evd_bw(false) == 32)
      NLSCHK((evd_two_stage<double, 32>(ctx, A, n, lam, e_work, dinfo, C, false, &used)));
    else
      NLSCHK((evd_two_stage<double, 64>(ctx, A, n, lam, e_work, dinfo, C, false, &used)));
    if (used) {
      *Q = C;
      return NLS_OK;
    }
  }
  ctx->evd_kind = 0;
  evd_mark(ctx, 0);
  NLSCHK(trd_fused<double>(ctx, A, n, n, lam, e_work, tau));
  evd_mark(ctx, 1);
  evd_mark(ctx, 2);
  NLSCHK(stedc_any(ctx, n, lam, e_work, C, dinfo));
  NLSCHK(check_info(ctx, dinfo, "tridiagonal eigensolver (stedc)"));
  evd_mark(ctx, 3);
  evd_mark(ctx, 4);
  if (evd_rocsolver_backtransform())
    BLASCHK(ctx, rocsolver_dormtr(ctx->blas, rocblas_side_left, rocblas_fill_lower, rocblas_operation_none, n, n, A, n, tau, C, n));
  else
    NLSCHK(apply_q_blocked<double>(ctx, A, n, n, tau, C, n, n));
  evd_mark(ctx, 5);
  ctx->evd_kind = 1;
  ctx->evd_n = n;
  *Q = C;
  return NLS_OK;
}

// rc_in (sharded fits): the status of the caller's local work since the previous exchange; it travels into the first vote inside.  The return
// value is then either a voted failure (every rank has it) or the status of the local tail (evd_unscale), which the caller takes to ITS next vote.
static int evd_hermitian(nls_ctx* ctx, double2* A, int n, double* lam, double* e_work, rocblas_int* dinfo, double2** Q, bool collective = false,
                         int rc_in = NLS_OK) {
  double f = 1.0;
  int rc = rc_in;
  if (rc == NLS_OK) rc = evd_prescale(ctx, A, n, 2, &f);
  NLSCHK(evd_hermitian_core(ctx, A, n, lam, e_work, dinfo, Q, collective, rc));
  return evd_unscale(ctx, lam, n, f);
}
static int evd_symmetric(nls_ctx* ctx, double* A, int n, double* lam, double* e_work, rocblas_int* dinfo, double** Q) {
  double f = 1.0;
  NLSCHK(evd_prescale(ctx, A, n, 1, &f));
  NLSCHK(evd_symmetric_core(ctx, A, n, lam, e_work, dinfo, Q));
  return evd_unscale(ctx, lam, n, f);
}

}  // namespace nls

using namespace nls;

extern "C" int nls_tridiag_only(nls_ctx* ctx, void* A, int n, int is_complex, double* d, double* e, void* tau) {
  if (!ctx) return NLS_ERR_ARG;
  if (!A || !d || !e || !tau || n < 1) return fail(ctx, NLS_ERR_ARG, "nls_tridiag_only: null pointer or n < 1");
  HIPCHK(ctx, hipSetDevice(ctx->device));
  const size_t esz = is_complex ? 16 : 8;
  void *dA = nullptr, *dtau = nullptr;
  double *dd = nullptr, *de = nullptr;
  NLSCHK(ws_get(ctx, "hook.A", esz * (size_t)n * n, &dA));
  NLSCHK(ws_get(ctx, "hook.tau", esz * (size_t)n, &dtau));
  NLSCHK(ws_get_t(ctx, "hook.d", (size_t)n, &dd));
  NLSCHK(ws_get_t(ctx, "hook.e", (size_t)n, &de));
  HIPCHK(ctx, hipMemcpyAsync(dA, A, esz * (size_t)n * n, hipMemcpyHostToDevice, ctx->stream));
  HIPCHK(ctx, hipMemsetAsync(dtau, 0, esz * (size_t)n, ctx->stream));
  HIPCHK(ctx, hipMemsetAsync(de, 0, sizeof(double) * n, ctx->stream));
  if (is_complex)
    NLSCHK(trd_fused<trd::Z>(ctx, static_cast<trd::Z*>(dA), n, n, dd, de, static_cast<trd::Z*>(dtau)));
  else
    NLSCHK(trd_fused<double>(ctx, static_cast<double*>(dA), n, n, dd, de, static_cast<double*>(dtau)));
  HIPCHK(ctx, hipMemcpyAsync(A, dA, esz * (size_t)n * n, hipMemcpyDeviceToHost, ctx->stream));
  HIPCHK(ctx, hipMemcpyAsync(d, dd, sizeof(double) * n, hipMemcpyDeviceToHost, ctx->stream));
  if (n > 1) {
    HIPCHK(ctx, hipMemcpyAsync(e, de, sizeof(double) * (n - 1), hipMemcpyDeviceToHost, ctx->stream));
    HIPCHK(ctx, hipMemcpyAsync(tau, dtau, esz * (size_t)(n - 1), hipMemcpyDeviceToHost, ctx->stream));
  }
  HIPCHK(ctx, hipStreamSynchronize(ctx->stream));
  return NLS_OK;
}

extern "C" int nls_eigh_only(nls_ctx* ctx, void* A, int n, int is_complex, double* lam) {
  if (!ctx) return NLS_ERR_ARG;
  if (!A || !lam || n < 1) return fail(ctx, NLS_ERR_ARG, "nls_eigh_only: null pointer or n < 1");
  HIPCHK(ctx, hipSetDevice(ctx->device));
  const size_t esz = is_complex ? 16 : 8;
  void* dA = nullptr;
  double *dlam = nullptr, *de = nullptr;
  rocblas_int* dinfo = nullptr;
  NLSCHK(ws_get(ctx, "hook.A", esz * (size_t)n * n, &dA));
  NLSCHK(ws_get_t(ctx, "hook.d", (size_t)n, &dlam));
  NLSCHK(ws_get_t(ctx, "hook.e", (size_t)n, &de));
  NLSCHK(ws_get_t(ctx, "evd.info", 4, &dinfo));
  HIPCHK(ctx, hipMemcpyAsync(dA, A, esz * (size_t)n * n, hipMemcpyHostToDevice, ctx->stream));
  void* Q = nullptr;
  if (is_complex) {
    double2* q = nullptr;
    NLSCHK(evd_hermitian(ctx, static_cast<double2*>(dA), n, dlam, de, dinfo, &q));
    Q = q;
  } else {
    double* q = nullptr;
    NLSCHK(evd_symmetric(ctx, static_cast<double*>(dA), n, dlam, de, dinfo, &q));
    Q = q;
  }
  HIPCHK(ctx, hipMemcpyAsync(A, Q, esz * (size_t)n * n, hipMemcpyDeviceToHost, ctx->stream));
  HIPCHK(ctx, hipMemcpyAsync(lam, dlam, sizeof(double) * n, hipMemcpyDeviceToHost, ctx->stream));
  HIPCHK(ctx, hipStreamSynchronize(ctx->stream));
  return NLS_OK;
}

// Hook (tests / profiling): the tridiagonal eigensolver alone on host data.  d[n] in: diagonal, out: eigenvalues ascending; e[n - 1]; Q: n x n
// column-major eigenvectors out.
extern "C" int nls_stedc_only(nls_ctx* ctx, double* d, const double* e, int n, double* Q) {
  if (!ctx) return NLS_ERR_ARG;
  if (!d || !Q || n < 1 || (n > 1 && !e)) return fail(ctx, NLS_ERR_ARG, "nls_stedc_only: null pointer or n < 1");
  HIPCHK(ctx, hipSetDevice(ctx->device));
  double *dd = nullptr, *de = nullptr, *dQ = nullptr;
  rocblas_int* dinfo = nullptr;
  NLSCHK(ws_get_t(ctx, "hook.d", (size_t)n, &dd));
  NLSCHK(ws_get_t(ctx, "hook.e", (size_t)n, &de));
  NLSCHK(ws_get_t(ctx, "hook.A", (size_t)n * n, &dQ));
  NLSCHK(ws_get_t(ctx, "evd.info", 4, &dinfo));
  HIPCHK(ctx, hipMemcpyAsync(dd, d, sizeof(double) * n, hipMemcpyHostToDevice, ctx->stream));
  HIPCHK(ctx, hipMemsetAsync(de, 0, sizeof(double) * n, ctx->stream));
  if (n > 1) HIPCHK(ctx, hipMemcpyAsync(de, e, sizeof(double) * (n - 1), hipMemcpyHostToDevice, ctx->stream));
  BLASCHK(ctx, rocblas_set_stream(ctx->blas, ctx->stream));
  NLSCHK(stedc_any(ctx, n, dd, de, dQ, dinfo));
  NLSCHK(check_info(ctx, dinfo, "tridiagonal eigensolver (stedc)"));
  HIPCHK(ctx, hipMemcpyAsync(d, dd, sizeof(double) * n, hipMemcpyDeviceToHost, ctx->stream));
  HIPCHK(ctx, hipMemcpyAsync(Q, dQ, sizeof(double) * (size_t)n * n, hipMemcpyDeviceToHost, ctx->stream));
  HIPCHK(ctx, hipStreamSynchronize(ctx->stream));
  return NLS_OK;
}

// ---- stage hooks of the two-stage reduction (tests / profiling) ------------------------------------------------------------------
template <class T>
static int twostage_stage_impl(nls_ctx* ctx, int stage, void* A, int n, int bw, void* aux, double* d, double* e, int ncols, int* info) {
  const size_t esz = sizeof(T);
  T *dA = nullptr, *daux = nullptr;
  double *dd = nullptr, *de = nullptr;
  int* dflag = nullptr;
  NLSCHK(ws_get_t(ctx, "hook.A", (size_t)n * n * (esz / 8), reinterpret_cast<double**>(&dA)));
  NLSCHK(ws_get_t(ctx, "hook2.aux", (size_t)n * std::max(n, ncols) * (esz / 8), reinterpret_cast<double**>(&daux)));
  T* dV2 = nullptr;  // the chase reflectors in the library's padded layout (stages 2 and 3)
  const long ldv = v2_ld(n, bw), v2c = v2_cols(n, bw);
  if (stage != 1) NLSCHK(ws_get_t(ctx, "hook2.V2", (size_t)ldv * v2c * (esz / 8), reinterpret_cast<double**>(&dV2)));
  NLSCHK(ws_get_t(ctx, "hook.d", (size_t)n, &dd));
  NLSCHK(ws_get_t(ctx, "hook.e", (size_t)n, &de));
  NLSCHK(ws_get_t(ctx, "evd2.flag", 4, &dflag));
  HIPCHK(ctx, hipMemcpyAsync(dA, A, esz * (size_t)n * n, hipMemcpyHostToDevice, ctx->stream));
  HIPCHK(ctx, hipMemsetAsync(dflag, 0, 4 * sizeof(int), ctx->stream));
  constexpr bool CPLX = sizeof(T) == 16;
  if (CPLX && bw != 32) return fail(ctx, NLS_ERR_ARG, "nls_twostage_stage: complex matrices take band width 32 only");
  if (stage == 1) {  // dense -> band: A in/out (band + block reflectors), aux = tau1[n] out, info = {flag, columns reduced}
    int nred = 0;
    if (bw == 32)
      NLSCHK((sy2sb<T, 32>(ctx, dA, n, n, daux, dflag, &nred)));
    else if constexpr (!CPLX)
      NLSCHK((sy2sb<T, 64>(ctx, dA, n, n, daux, dflag, &nred)));
    HIPCHK(ctx, hipMemcpyAsync(A, dA, esz * (size_t)n * n, hipMemcpyDeviceToHost, ctx->stream));
    HIPCHK(ctx, hipMemcpyAsync(aux, daux, esz * (size_t)n, hipMemcpyDeviceToHost, ctx->stream));
    HIPCHK(ctx, hipMemcpyAsync(info, dflag, sizeof(int), hipMemcpyDeviceToHost, ctx->stream));
    HIPCHK(ctx, hipStreamSynchronize(ctx->stream));
    info[1] = nred;
  } else if (stage == 2) {  // band (the lower bw sub-diagonals of A) -> d, e; aux = V2 (n x n) out; info[0] = chase error word
    unsigned* ctl = nullptr;
    if (bw == 32)
      NLSCHK((sb2st<T, 32>(ctx, dA, n, n, dd, de, dV2, &ctl)));
    else if constexpr (!CPLX)
      NLSCHK((sb2st<T, 64>(ctx, dA, n, n, dd, de, dV2, &ctl)));
    unsigned hctl[2] = {0, 0};
    HIPCHK(ctx, hipMemcpyAsync(hctl, ctl, sizeof(hctl), hipMemcpyDeviceToHost, ctx->stream));
    HIPCHK(ctx, hipMemcpy2DAsync(aux, esz * (size_t)n, dV2, esz * (size_t)ldv, esz * (size_t)n, (size_t)n, hipMemcpyDeviceToHost, ctx->stream));
    HIPCHK(ctx, hipMemcpyAsync(d, dd, sizeof(double) * n, hipMemcpyDeviceToHost, ctx->stream));
    if (n > 1) HIPCHK(ctx, hipMemcpyAsync(e, de, sizeof(double) * (n - 1), hipMemcpyDeviceToHost, ctx->stream));
    HIPCHK(ctx, hipStreamSynchronize(ctx->stream));
    info[0] = (int)hctl[1];
    info[1] = (int)hctl[0];
  } else if (stage == 3) {  // aux (n x ncols) <- Q2 aux with the chase reflectors V2 = A
    HIPCHK(ctx, hipMemcpyAsync(daux, aux, esz * (size_t)n * ncols, hipMemcpyHostToDevice, ctx->stream));
    HIPCHK(ctx, hipMemsetAsync(dV2, 0, esz * (size_t)ldv * v2c, ctx->stream));
    HIPCHK(ctx, hipMemcpy2DAsync(dV2, esz * (size_t)ldv, A, esz * (size_t)n, esz * (size_t)n, (size_t)n, hipMemcpyHostToDevice, ctx->stream));
    if (bw == 32)
      NLSCHK((apply_q2<T, 32>(ctx, dV2, n, daux, n, ncols)));
    else if constexpr (!CPLX)
      NLSCHK((apply_q2<T, 64>(ctx, dV2, n, daux, n, ncols)));
    HIPCHK(ctx, hipMemcpyAsync(aux, daux, esz * (size_t)n * ncols, hipMemcpyDeviceToHost, ctx->stream));
    HIPCHK(ctx, hipStreamSynchronize(ctx->stream));
  } else {
    return fail(ctx, NLS_ERR_ARG, "nls_twostage_stage: stage must be 1, 2 or 3");
  }
  return NLS_OK;
}

extern "C" int nls_twostage_stage(nls_ctx* ctx, int stage, void* A, int n, int is_complex, int bw, void* aux, double* d, double* e, int ncols,
                                  int* info) {
  if (!ctx) return NLS_ERR_ARG;
  if (!A || !aux || !info || n < 1 || (bw != 32 && bw != 64)) return fail(ctx, NLS_ERR_ARG, "nls_twostage_stage: null pointer, n < 1 or bw not in {32, 64}");
  if (stage == 2 && (!d || !e)) return fail(ctx, NLS_ERR_ARG, "nls_twostage_stage: stage 2 needs d and e");
  if (stage == 3 && ncols < 1) return fail(ctx, NLS_ERR_ARG, "nls_twostage_stage: stage 3 needs ncols >= 1");
  HIPCHK(ctx, hipSetDevice(ctx->device));
  return is_complex ? twostage_stage_impl<trd::Z>(ctx, stage, A, n, bw, aux, d, e, ncols, info)
                    : twostage_stage_impl<double>(ctx, stage, A, n, bw, aux, d, e, ncols, info);
}

extern "C" int nls_evd_stage_ms(nls_ctx* ctx, double* out8) {
  if (!ctx) return NLS_ERR_ARG;
  if (!out8) return fail(ctx, NLS_ERR_ARG, "nls_evd_stage_ms: out8 is NULL");
  if (ctx->evd_kind == 0) return fail(ctx, NLS_ERR_ARG, "nls_evd_stage_ms: no eigendecomposition has completed on this context");
  HIPCHK(ctx, hipSetDevice(ctx->device));
  for (int i = 0; i < 6; ++i)
    if (!ctx->evd_ev[i]) return fail(ctx, NLS_ERR_HIP, "nls_evd_stage_ms: stage events missing");
  HIPCHK(ctx, hipEventSynchronize(ctx->evd_ev[5]));
  double sum = 0.0;
  for (int i = 0; i < 5; ++i) {
    float ms = 0.f;
    HIPCHK(ctx, hipEventElapsedTime(&ms, ctx->evd_ev[i], ctx->evd_ev[i + 1]));
    out8[i] = ms;
    sum += ms;
  }
  out8[5] = sum;
  out8[6] = (double)ctx->evd_n;
  out8[7] = (double)ctx->evd_kind;
  return NLS_OK;
}

extern "C" long nls_twostage_fallbacks(const nls_ctx* ctx) { return ctx ? ctx->twostage_fallbacks : -1; }
extern "C" long nls_twostage_rescues(const nls_ctx* ctx) { return ctx ? ctx->twostage_rescues : -1; }
