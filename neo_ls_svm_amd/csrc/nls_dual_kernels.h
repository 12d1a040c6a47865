// Kernels of the dual path.  All matrices are padded to multiples of 128 (rows
// and columns) / 16 (contraction) and zero filled, so the tile engine runs without guards.
#pragma once
#include "nls_gemm.h"

namespace nls {

// C = A B on the fp64 matrix pipe.  A: M x K row-major (lda), B: K x N row-major (ldb), C: M x N (ldc).
// grid = (N / 128, M / 128).  Epilogues:
//   EPI_STORE : C = acc
//   EPI_RBF   : C = exp(-0.5 max(0, xx_i + yy_j - 2 acc)) + add, exact-zero distance on the diagonal when
//               `same` (sklearn euclidean_distances + rbf_kernel, called at _neo_ls_svm.py:261,321,474,669);
//               entries outside the valid m x n block are written as zero.
enum { EPI_STORE = 0, EPI_RBF = 1 };
struct GemmParams {
  const double* A;
  const double* B;
  double* C;
  long lda, ldb, ldc;
  int K;
  // RBF epilogue
  const double* xx;
  const double* yy;
  long m_valid, n_valid;
  int same;
  double add;
};

template <int EPI>
__global__ void __launch_bounds__(Cfg4::NTHREADS, 2) k_gemm(GemmParams p) {
  using C = Cfg4;
  extern __shared__ double smem[];
  const long row0 = (long)blockIdx.y * BM;
  const long col0 = (long)blockIdx.x * BN;
  v4d acc[C::MT][C::NTL];
  zero_acc(acc);
  MMajorLoader<C::NTHREADS, BM> la{p.A, p.lda, row0};
  KMajorLoader<C::NTHREADS, BN> lb{p.B, p.ldb, col0};
  mainloop_real<C, false>(acc, la, lb, 0, p.K / BK, smem);
#pragma unroll
  for (int mt = 0; mt < C::MT; ++mt)
#pragma unroll
    for (int r = 0; r < 4; ++r) {
      const long row = row0 + C::acc_row(mt, r);
#pragma unroll
      for (int nt = 0; nt < C::NTL; ++nt) {
        const long col = col0 + C::acc_col(nt);
        double v = acc[mt][nt][r];
        if constexpr (EPI == EPI_RBF) {
          if (row < p.m_valid && col < p.n_valid) {
            double d2 = p.xx[row] + p.yy[col] - 2.0 * v;
            d2 = d2 > 0.0 ? d2 : 0.0;
            if (p.same && row == col) d2 = 0.0;
            v = exp(-0.5 * d2) + p.add;
          } else {
            v = 0.0;
          }
        }
        p.C[row * p.ldc + col] = v;
      }
    }
}

// out[r][c] = in[c][r]; in: rows x cols (ld_in), out: cols_pad x rows_pad (ld_out), zero padded.
__global__ void k_transpose_pad(const double* in, long rows, long cols, long ld_in, double* out, long out_rows, long ld_out) {
  const long c = blockIdx.x * (long)blockDim.x + threadIdx.x;  // column of out == row of in
  const long r = blockIdx.y;                                   // row of out == column of in
  if (c >= ld_out || r >= out_rows) return;
  out[r * ld_out + c] = (r < cols && c < rows) ? in[c * ld_in + r] : 0.0;
}

// out (rows_pad x ld_out) = zero-padded copy of in (rows x cols, ld_in).
__global__ void k_copy_pad(const double* in, long rows, long cols, long ld_in, double* out, long out_rows, long ld_out) {
  const long c = blockIdx.x * (long)blockDim.x + threadIdx.x;
  const long r = blockIdx.y;
  if (c >= ld_out || r >= out_rows) return;
  out[r * ld_out + c] = (r < rows && c < cols) ? in[r * ld_in + c] : 0.0;
}

// xx[i] = sum_k X[i][k]^2 ; one wave per row.
__global__ void k_row_sqnorm(const double* X, long rows, long cols, long ld, double* out) {
  const long row = blockIdx.x * (long)(blockDim.x >> 6) + (threadIdx.x >> 6);
  const int lane = threadIdx.x & 63;
  if (row >= rows) return;
  double acc = 0.0;
  for (long j = lane; j < cols; j += 64) {
    const double v = X[row * ld + j];
    acc += v * v;
  }
  for (int o = 32; o > 0; o >>= 1) acc += __shfl_down(acc, o);
  if (lane == 0) out[row] = acc;
}

// S[i][j] = sn_i F[i][j] sn_j (n x n, lds = n, for dsyevd); F0 = F with a zero diagonal (in place option).
__global__ void k_dual_scale_sym(const double* F, long ldf, const double* sn, long n, double* S, long lds) {
  const long j = blockIdx.x * (long)blockDim.x + threadIdx.x;
  const long i = blockIdx.y;
  if (j >= n) return;
  S[i * lds + j] = sn[i] * F[i * ldf + j] * sn[j];
}
__global__ void k_zero_diag_copy(const double* F, long ld, long n_pad, long n, double* F0) {
  const long j = blockIdx.x * (long)blockDim.x + threadIdx.x;
  const long i = blockIdx.y;
  if (j >= ld || i >= n_pad) return;
  F0[i * ld + j] = (i == j || i >= n || j >= n) ? 0.0 : F[i * ld + j];
}

// W[i][k] = sn_i Q[i][k] from column-major Q (ldq = n): W row-major, zero padded to n_pad x n_pad.
__global__ void k_dual_build_W(const double* Qcm, long n, const double* sn, double* W, long n_pad) {
  const long k = blockIdx.x * (long)blockDim.x + threadIdx.x;
  const long i = blockIdx.y;
  if (k >= n_pad || i >= n_pad) return;
  W[i * n_pad + k] = (i < n && k < n) ? sn[i] * Qcm[k * n + i] : 0.0;
}

// qy_k = sum_i Q[i][k] sn_i y_i : one block per k (column k contiguous).
__global__ void k_dual_qty(const double* Qcm, long n, const double* sn, const double* y, double* qy) {
  const long k = blockIdx.x;
  __shared__ double sh[256];
  double a = 0.0;
  for (long i = threadIdx.x; i < n; i += blockDim.x) a += Qcm[k * n + i] * sn[i] * y[i];
  sh[threadIdx.x] = a;
  __syncthreads();
  for (int s = blockDim.x / 2; s > 0; s >>= 1) {
    if (threadIdx.x < s) sh[threadIdx.x] += sh[threadIdx.x + s];
    __syncthreads();
  }
  if (threadIdx.x == 0) qy[k] = sh[0];
}

// The left operands of the reduced sweep (D3): WM = W o M, WW = W o W, WQ = W o qy^T (in place on M ok), and for the
// predictive variance (D5) KW2 = (Kt W)^2 elementwise with Kt = rbf(X_, 1/2) = F - 1:  Kt W = F0 W + 2 W - 1 colsum(W)
// (F = F0 + 2 I: the kernel's unit diagonal plus the + 1), i.e. M + 2 W - cs, everything the sweep has at hand anyway.
__global__ void k_dual_hadamards(const double* W, const double* M, const double* qy, const double* cs, long n, long n_pad, double* WM,
                                 double* WW, double* WQ, double* KW2) {
  const long k = blockIdx.x * (long)blockDim.x + threadIdx.x;
  const long i = blockIdx.y;
  if (k >= n_pad || i >= n_pad) return;
  const long o = i * n_pad + k;
  const double w = W[o], m = M[o];
  const double kw = (i < n && k < n) ? m + 2.0 * w - cs[k] : 0.0;
  WM[o] = w * m;
  WW[o] = w * w;
  WQ[o] = (k < n) ? w * qy[k] : 0.0;
  KW2[o] = kw * kw;
}

// part[chunk][k] = sum of W[i][k] over the rows of the chunk (column sums in two steps: k_sum_partials finishes).
__global__ void k_col_partial_sums(const double* W, long n, long ld, long rows_per_chunk, double* part) {
  const long k = blockIdx.x * (long)blockDim.x + threadIdx.x;
  if (k >= ld) return;
  const long r0 = (long)blockIdx.y * rows_per_chunk;
  long r1 = r0 + rows_per_chunk;
  if (r1 > n) r1 = n;
  double a = 0.0;
  for (long i = r0; i < r1; ++i) a += W[i * ld + k];
  part[(long)blockIdx.y * ld + k] = a;
}

// sigma_i = sqrt(1 - SG[i][g]) : the predictive standard deviation at the selected gamma from the sweep's table.
__global__ void k_dual_sigma_col(const double* SG, int Gp, int g, long n, double* out) {
  const long i = blockIdx.x * (long)blockDim.x + threadIdx.x;
  if (i < n) out[i] = sqrt(1.0 - SG[i * Gp + g]);
}

// yloo = -(t / hd) a + Fa with hd == 0 -> eps (_neo_ls_svm.py:279-286), in place into t.
__global__ void k_dual_yloo(double* t, const double* hd, const double* a, const double* Fa, long total) {
  const long idx = blockIdx.x * (long)blockDim.x + threadIdx.x;
  if (idx >= total) return;
  double h = hd[idx];
  if (h == 0.0) h = 2.220446049250313e-16;
  t[idx] = -(t[idx] / h) * a[idx] + Fa[idx];
}

// Per-gamma weighted error sums of e = yloo - y (classifier clipping as the primal): part[blk][3][Gp].
__global__ void k_dual_errors(const double* yloo, const double* y, const double* s, long n, int G, int Gp, int is_clf,
                              double* part) {
  const long r0 = (long)blockIdx.x * 64;
  long r1 = r0 + 64;
  if (r1 > n) r1 = n;
  for (int g = threadIdx.x; g < G; g += blockDim.x) {
    double e0 = 0.0, e1 = 0.0, e2 = 0.0;
    for (long i = r0; i < r1; ++i) {
      const double yi = y[i], si = s[i];
      double e = yloo[i * Gp + g] - yi;
      if (is_clf && ((yi > 0 && e > 0) || (yi < 0 && e < 0))) e = 0.0;
      const double ae = fabs(e);
      e0 += si * ae;
      if (is_clf) {
        e1 += (ae >= 1.0) ? si : 0.0;
        e2 += si * fmax(0.0, ae - 1.0);
      }
    }
    double* o = part + (long)blockIdx.x * 3 * Gp;
    o[g] = e0;
    o[Gp + g] = e1;
    o[2 * Gp + g] = e2;
  }
}

// Column of the selected gamma: residuals and score sums (same reductions as the primal k_loo_column).
__global__ void k_dual_column(const double* yloo, const double* y, const double* s, long n, int Gp, int g, int is_clf,
                              double ybar, double* loo_res, double* part) {
  __shared__ double s0[256], s1[256];
  const long i = blockIdx.x * (long)blockDim.x + threadIdx.x;
  double a0 = 0.0, a1 = 0.0;
  if (i < n) {
    const double yi = y[i], si = s[i], yl = yloo[i * Gp + g];
    double e = yl - yi;
    if (is_clf && ((yi > 0 && e > 0) || (yi < 0 && e < 0))) e = 0.0;
    loo_res[i] = e;
    if (is_clf) {
      const double sg = (yl > 0.0) ? 1.0 : ((yl < 0.0) ? -1.0 : 0.0);
      a0 = (sg == yi) ? si : 0.0;
    } else {
      a0 = si * (yi - yl) * (yi - yl);
      a1 = si * (yi - ybar) * (yi - ybar);
    }
  }
  s0[threadIdx.x] = a0;
  s1[threadIdx.x] = a1;
  __syncthreads();
  for (int st = blockDim.x / 2; st > 0; st >>= 1) {
    if (threadIdx.x < st) {
      s0[threadIdx.x] += s0[threadIdx.x + st];
      s1[threadIdx.x] += s1[threadIdx.x + st];
    }
    __syncthreads();
  }
  if (threadIdx.x == 0) {
    part[2L * blockIdx.x] = s0[0];
    part[2L * blockIdx.x + 1] = s1[0];
  }
}

// M2[i][j] = F[i][j] + (i == j) gamma / sn_i^2 (n x n, ld = n, for dpotrf); Kp = F - 1 alongside.
__global__ void k_dual_chol_inputs(const double* F, long ldf, const double* sn, long n, double gamma, double* M2, long ldm, double* Kp) {
  const long j = blockIdx.x * (long)blockDim.x + threadIdx.x;
  const long i = blockIdx.y;
  if (j >= n) return;
  const double f = F[i * ldf + j];
  M2[i * ldm + j] = f + ((i == j) ? gamma / (sn[i] * sn[i]) : 0.0);
  if (Kp) Kp[i * n + j] = f - 1.0;
}

// out[i] = T[i][g] (T row-major with leading dimension ld): the selected column of a grid table
__global__ void k_dual_take_column(const double* T, long ld, int g, long n, double* out) {
  const long i = blockIdx.x * (long)blockDim.x + threadIdx.x;
  if (i < n) out[i] = T[i * ld + g];
}

// out[i] = f(sum_j A[i][j] x[j]) ; one wave per row.  mode 0: dot + bias ; mode 1: dot - y[i] with clipping.
__global__ void k_dual_gemv(const double* A, long ld, long rows, long cols, const double* x, double bias, const double* y,
                            int is_clf, double* out) {
  const long row = blockIdx.x * (long)(blockDim.x >> 6) + (threadIdx.x >> 6);
  const int lane = threadIdx.x & 63;
  if (row >= rows) return;
  double acc = 0.0;
  for (long j = lane; j < cols; j += 64) acc += A[row * ld + j] * x[j];
  for (int o = 32; o > 0; o >>= 1) acc += __shfl_down(acc, o);
  if (lane == 0) {
    if (y) {
      double e = acc - y[row];
      if (is_clf && ((y[row] > 0 && e > 0) || (y[row] < 0 && e < 0))) e = 0.0;
      out[row] = e;
    } else {
      out[row] = acc + bias;
    }
  }
}

// out[i] = sqrt(1 - sum_j A[i][j] B[i][j]) ; one wave per row (sigma of the dual model).
__global__ void k_dual_sigma(const double* A, const double* B, long ld, long rows, long cols, double* out) {
  const long row = blockIdx.x * (long)(blockDim.x >> 6) + (threadIdx.x >> 6);
  const int lane = threadIdx.x & 63;
  if (row >= rows) return;
  double acc = 0.0;
  for (long j = lane; j < cols; j += 64) acc += A[row * ld + j] * B[row * ld + j];
  for (int o = 32; o > 0; o >>= 1) acc += __shfl_down(acc, o);
  if (lane == 0) out[row] = sqrt(1.0 - acc);
}

__global__ void k_sum_vec(const double* x, long n, double* out) {
  __shared__ double sh[256];
  double a = 0.0;
  for (long i = threadIdx.x; i < n; i += blockDim.x) a += x[i];
  sh[threadIdx.x] = a;
  __syncthreads();
  for (int s = blockDim.x / 2; s > 0; s >>= 1) {
    if (threadIdx.x < s) sh[threadIdx.x] += sh[threadIdx.x + s];
    __syncthreads();
  }
  if (threadIdx.x == 0) out[0] = sh[0];
}

}  // namespace nls
