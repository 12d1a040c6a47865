// Eigendecomposition of the path's Hermitian / real symmetric matrices (P4: _neo_ls_svm.py:120, D2: :265):
// tridiagonalisation by the panel of nls_trd.h (two or three kernels per column), rocSOLVER's divide-and-conquer
// (stedc) on the tridiagonal matrix, and a blocked back-transformation built from rocBLAS GEMMs (apply_q_blocked).
// NLS_EVD=rocsolver selects the all-rocSOLVER zheevd / dsyevd instead.
#include "nls_host.h"
#include "nls_trd.h"

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
  // panel were built and measured slower in round 2 - the cost is the GPU's own dependent-dispatch latency: profiles/r02_trd_graph.log,
  // profiles/r02_trd_persistent.log; they are in the history, not in the library.)
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

// C (n x ncols, column-major, ldc) <- Q C with the reflectors of trd_fused in A / tau (nls_trd.h, "back-transformation").
template <class T>
static int apply_q_blocked(nls_ctx* ctx, const T* A, long lda, int n, const T* tau, T* C, long ldc, int ncols) {
  using namespace trd;
  const int nrefl = n - 1;
  if (nrefl <= 0 || ncols <= 0) return NLS_OK;
  T *Vw = nullptr, *S = nullptr, *W = nullptr;
  NLSCHK(ws_get_t(ctx, "bt.V", (size_t)n * KBQ, &Vw));
  NLSCHK(ws_get_t(ctx, "bt.S", (size_t)KBQ * KBQ, &S));
  NLSCHK(ws_get_t(ctx, "bt.W", (size_t)KBQ * ncols, &W));
  BLASCHK(ctx, rocblas_set_stream(ctx->blas, ctx->stream));
  for (int j0 = ((nrefl - 1) / KBQ) * KBQ; j0 >= 0; j0 -= KBQ) {
    const int kb = std::min(KBQ, nrefl - j0), r0 = j0 + 1, m = n - r0;
    hipLaunchKernelGGL(k_trd_copy_v<T>, dim3((unsigned)(((long)m * kb + 255) / 256)), dim3(256), 0, ctx->stream, A, lda, n, j0, kb, Vw);
    BLASCHK(ctx, bt_gemm(ctx->blas, bt_op_h(A), rocblas_operation_none, kb, kb, m, 1.0, Vw, m, Vw, m, 0.0, S, kb));
    hipLaunchKernelGGL(k_trd_tinv<T>, dim3((unsigned)((kb * kb + 255) / 256)), dim3(256), 0, ctx->stream, S, kb, tau, j0);
    HIPCHK(ctx, hipGetLastError());
    BLASCHK(ctx, bt_gemm(ctx->blas, bt_op_h(A), rocblas_operation_none, kb, ncols, m, 1.0, Vw, m, C + r0, ldc, 0.0, W, kb));
    BLASCHK(ctx, bt_trsm(ctx->blas, kb, ncols, S, kb, W, kb));
    BLASCHK(ctx, bt_gemm(ctx->blas, rocblas_operation_none, rocblas_operation_none, m, ncols, kb, -1.0, Vw, m, W, kb, 1.0, C + r0, ldc));
  }
  return NLS_OK;
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

// Hermitian: A (n x n complex column-major, lower) is destroyed; eigenvalues ascending in lam, eigenvectors
// (columns) in *Q, which is either A itself (rocSOLVER path) or the workspace "evd.C".
// collective = true (nls_primal_fit, where every rank holds the same all-reduced matrix): stedc on rank 0 + broadcast,
// back-transformation split by columns over the ranks, blocks all-gathered (Q bit-identical everywhere).
static int evd_hermitian_core(nls_ctx* ctx, double2* A, int n, double* lam, double* e_work, rocblas_int* dinfo, double2** Q, bool collective) {
  BLASCHK(ctx, rocblas_set_stream(ctx->blas, ctx->stream));
  if (evd_use_rocsolver() || n < 3) {
    BLASCHK(ctx, rocsolver_zheevd(ctx->blas, rocblas_evect_original, rocblas_fill_lower, n, reinterpret_cast<rocblas_double_complex*>(A), n, lam,
                                  e_work, dinfo));
    NLSCHK(check_info(ctx, dinfo, "rocsolver_zheevd"));
    *Q = A;
    return NLS_OK;
  }
  trd::Z* tau = nullptr;
  double2* C = nullptr;
  NLSCHK(ws_get_t(ctx, "evd.tau", (size_t)n, &tau));
  NLSCHK(ws_get_t(ctx, "evd.C", (size_t)n * n, &C));
  NLSCHK(trd_fused<trd::Z>(ctx, reinterpret_cast<trd::Z*>(A), n, n, lam, e_work, tau));
  if (collective && multi_rank(ctx) && n >= 64) {
    // Every rank holds the same all-reduced matrix and the tridiagonalisation is bit-reproducible, so the reflectors are
    // replicated.  The tridiagonal eigensolver runs on rank 0 ONLY and (lam, C) are broadcast: all ranks then pair the
    // same eigenvalues with the same basis whatever rocSOLVER's stedc does on clustered spectra.  The back-transformation
    // is split by columns over the ranks and the blocks are all-gathered.
    double* flag = e_work;  // e is dead once stedc has run
    double hflag = 0.0;
    if (ctx->rank == 0) {
      BLASCHK(ctx, rocsolver_zstedc(ctx->blas, rocblas_evect_tridiagonal, n, lam, e_work, reinterpret_cast<rocblas_double_complex*>(C), n, dinfo));
      rocblas_int info = 0;
      HIPCHK(ctx, hipMemcpyAsync(&info, dinfo, sizeof(info), hipMemcpyDeviceToHost, ctx->stream));
      HIPCHK(ctx, hipStreamSynchronize(ctx->stream));
      hflag = (double)info;
    }
    HIPCHK(ctx, hipMemcpyAsync(flag, &hflag, sizeof(double), hipMemcpyHostToDevice, ctx->stream));
    NLSCHK(do_broadcast(ctx, flag, 1, 0));
    HIPCHK(ctx, hipMemcpyAsync(&hflag, flag, sizeof(double), hipMemcpyDeviceToHost, ctx->stream));
    HIPCHK(ctx, hipStreamSynchronize(ctx->stream));
    if (hflag != 0.0) return fail(ctx, NLS_ERR_LINALG, "rocsolver_zstedc (rank 0): info = %d (no convergence)", (int)hflag);
    NLSCHK(do_broadcast(ctx, lam, (size_t)n, 0));
    NLSCHK(do_broadcast(ctx, reinterpret_cast<double*>(C), (size_t)2 * n * n, 0));
    std::vector<size_t> offs((size_t)ctx->world + 1);
    for (int r = 0; r <= ctx->world; ++r) offs[r] = (size_t)2 * n * (size_t)((long)n * r / ctx->world);
    const long c0 = (long)n * ctx->rank / ctx->world, c1 = (long)n * (ctx->rank + 1) / ctx->world;
    if (evd_rocsolver_backtransform()) {
      BLASCHK(ctx, rocsolver_zunmtr(ctx->blas, rocblas_side_left, rocblas_fill_lower, rocblas_operation_none, n, (rocblas_int)(c1 - c0),
                                    reinterpret_cast<rocblas_double_complex*>(A), n, reinterpret_cast<rocblas_double_complex*>(tau),
                                    reinterpret_cast<rocblas_double_complex*>(C + c0 * n), n));
    } else {
      NLSCHK(apply_q_blocked<trd::Z>(ctx, reinterpret_cast<trd::Z*>(A), n, n, tau, reinterpret_cast<trd::Z*>(C + c0 * n), n, (int)(c1 - c0)));
    }
    NLSCHK(do_allgather_blocks(ctx, reinterpret_cast<double*>(C), offs));
    *Q = C;
    return NLS_OK;
  }
  BLASCHK(ctx, rocsolver_zstedc(ctx->blas, rocblas_evect_tridiagonal, n, lam, e_work, reinterpret_cast<rocblas_double_complex*>(C), n, dinfo));
  NLSCHK(check_info(ctx, dinfo, "rocsolver_zstedc"));
  if (evd_rocsolver_backtransform()) {
    BLASCHK(ctx, rocsolver_zunmtr(ctx->blas, rocblas_side_left, rocblas_fill_lower, rocblas_operation_none, n, n,
                                  reinterpret_cast<rocblas_double_complex*>(A), n, reinterpret_cast<rocblas_double_complex*>(tau),
                                  reinterpret_cast<rocblas_double_complex*>(C), n));
  } else {
    NLSCHK(apply_q_blocked<trd::Z>(ctx, reinterpret_cast<trd::Z*>(A), n, n, tau, reinterpret_cast<trd::Z*>(C), n, n));
  }
  *Q = C;
  return NLS_OK;
}

static int evd_symmetric_core(nls_ctx* ctx, double* A, int n, double* lam, double* e_work, rocblas_int* dinfo, double** Q) {
  BLASCHK(ctx, rocblas_set_stream(ctx->blas, ctx->stream));
  if (evd_use_rocsolver() || n < 3) {
    BLASCHK(ctx, rocsolver_dsyevd(ctx->blas, rocblas_evect_original, rocblas_fill_lower, n, A, n, lam, e_work, dinfo));
    NLSCHK(check_info(ctx, dinfo, "rocsolver_dsyevd"));
    *Q = A;
    return NLS_OK;
  }
  double *tau = nullptr, *C = nullptr;
  NLSCHK(ws_get_t(ctx, "evd.taur", (size_t)n, &tau));
  NLSCHK(ws_get_t(ctx, "evd.Cr", (size_t)n * n, &C));
  NLSCHK(trd_fused<double>(ctx, A, n, n, lam, e_work, tau));
  BLASCHK(ctx, rocsolver_dstedc(ctx->blas, rocblas_evect_tridiagonal, n, lam, e_work, C, n, dinfo));
  NLSCHK(check_info(ctx, dinfo, "rocsolver_dstedc"));
  if (evd_rocsolver_backtransform())
    BLASCHK(ctx, rocsolver_dormtr(ctx->blas, rocblas_side_left, rocblas_fill_lower, rocblas_operation_none, n, n, A, n, tau, C, n));
  else
    NLSCHK(apply_q_blocked<double>(ctx, A, n, n, tau, C, n, n));
  *Q = C;
  return NLS_OK;
}

static int evd_hermitian(nls_ctx* ctx, double2* A, int n, double* lam, double* e_work, rocblas_int* dinfo, double2** Q, bool collective = false) {
  double f = 1.0;
  NLSCHK(evd_prescale(ctx, A, n, 2, &f));
  NLSCHK(evd_hermitian_core(ctx, A, n, lam, e_work, dinfo, Q, collective));
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
