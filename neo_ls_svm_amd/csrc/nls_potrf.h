// Cholesky factorisation A = L L^T of a real symmetric positive definite matrix (lower triangle, column-major, in place) - the L_ output of
// the dual fit (cho_factor(gamma* diag(sn^-2) + K), _neo_ls_svm.py:313-314).
//
// Why not rocsolver_dpotrf: at n = 10^4 it takes 42 ms = 8 TFLOP/s - 79 diagonal blocks through a chain of small kernels (potf2_kernel_small,
// 165 us each) with trsm / syrk calls of 128 columns in between.  Here (24.7 ms: k_potrf_syrk 12.7, k_potrf_leaf 8.7, k_potrf_panel 3.0), right-looking
// in panels of NB = 128 columns, three launches per panel:
//   k_potrf_leaf : ONE workgroup factors the 128 x 128 diagonal block in LDS (four 32 x 32 sub-blocks: a wave holds a sub-block in registers, one
//                  row per lane, pivot and column entries travel by v_readlane - no barrier inside a sub-block; the rows below it are solved one
//                  row per thread, the rest of the block updated by all threads) and writes L11 and the inverses of its four diagonal sub-blocks;
//   k_potrf_panel: L21 = A21 L11^-T by blocked forward substitution on fp64 MFMA, a wave = 16 rows held as B-operand fragments: block s is
//                  A_s - sum_{t<s} X_t L_st^T (products with the stored L11) times the inverse of L_ss.  The product is formed transposed,
//                  D'[c][r]: its accumulator layout (c = 4 reg + lane / 16) IS the B-operand layout (k = 4 ks + lane / 16), so a finished block
//                  feeds the next products straight from its registers, and the store runs down stored columns (16 lanes = 16 consecutive
//                  rows); every wave reads and writes its own 16 rows only: in place;
//   k_potrf_syrk : A22 -= L21 L21^T on the lower 128 x 128 tiles through the real tile engine (nls_gemm.h).
// A pivot <= 0 (or NaN) raises info = its 1-based index, as LAPACK does; the factorisation carries on with garbage (finite control flow).
#pragma once
#include "nls_gemm.h"
#include "nls_sb.h"

namespace nls {
namespace potrf {
using sb::hv4d;

constexpr int NB = 128;    // panel width = leaf size
constexpr int SBK = 32;    // sub-block of the leaf
constexpr int LDL = NB + 1;
constexpr size_t LEAF_LDS = (size_t)NB * LDL * sizeof(double);

__device__ __forceinline__ double readlane_f64(double v, int src) {
  return __hiloint2double(__builtin_amdgcn_readlane(__double2hiint(v), src), __builtin_amdgcn_readlane(__double2loint(v), src));
}

// A: the w x w diagonal block (lower, leading dimension lda), w <= NB.  Out: L11 in place (strict upper part untouched),
// Sinv (four 32 x 32 column-major inverses of the diagonal sub-blocks; the block is identity-padded beyond w), info (0 or the global 1-based
// index of the first bad pivot; only ever raised).
// With `rhs_run` (round 4) the forward half of alpha = cho_solve(L, y) is carried along: rhs_run = y - L[:, :k0] x[:k0] on entry; wave 0 solves the
// block's unknowns x[k0 .. k0 + w) against the finished L11 (two rows per lane, the solved value travels by v_readlane) and stores them in xsol;
// k_potrf_panel then takes L21 x off the rows below.
__global__ void __launch_bounds__(256) k_potrf_leaf(double* A, long lda, int w, int k0, double* Sinv, int* info, const double* rhs_run, double* xsol) {
  extern __shared__ __attribute__((aligned(16))) unsigned char potrf_smem[];
  double(*Ls)[LDL] = reinterpret_cast<double(*)[LDL]>(potrf_smem);  // Ls[r][c], lower part valid
  __shared__ double dinv[NB];                                       // 1 / L[k][k]: the solves multiply
  const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
  // (sixteen entries per thread requested together: a load and an LDS store per loop trip is one memory round trip per trip, 64 in a row)
  for (int i0 = 0; i0 < NB * NB / 256; i0 += 16) {
    double bv[16];
#pragma unroll
    for (int it = 0; it < 16; ++it) {
      const int idx = tid + 256 * (i0 + it), r = idx % NB, c = idx / NB;
      bv[it] = (r < w && c < w && r >= c) ? A[r + (long)c * lda] : (r == c ? 1.0 : 0.0);
    }
#pragma unroll
    for (int it = 0; it < 16; ++it) {
      const int idx = tid + 256 * (i0 + it);
      Ls[idx % NB][idx / NB] = bv[it];
    }
  }
  __syncthreads();
  for (int kb = 0; kb < NB; kb += SBK) {
    // (a) the 32 x 32 diagonal sub-block: wave 0, lane = row, the row in registers
    if (wave == 0) {
      double a[SBK];
      const int r = lane & (SBK - 1);
#pragma unroll
      for (int c = 0; c < SBK; ++c) a[c] = Ls[kb + r][kb + c];
      int bad = 0;
      // L[c][k] reaches every lane as a scalar (v_readlane)
#pragma unroll
      for (int k = 0; k < SBK; ++k) {
        double d = readlane_f64(a[k], k);
        if (!(d > 0.0) || !isfinite(d)) {  // uniform
          if (bad == 0) bad = kb + k + 1;
          d = 1.0;
        }
        const double inv = sb::fast_rsqrt(d), sq = d * inv;  // (v_rsq_f64 + a Newton step instead of sqrt and a division on the critical path)
        if (lane == 0) dinv[kb + k] = inv;
        const double l = r == k ? sq : a[k] * inv;
        a[k] = l;
#pragma unroll
        for (int c = k + 1; c < SBK; ++c) a[c] -= l * readlane_f64(l, c);  // (meaningful for r >= c)
      }
      if (lane < SBK) {
#pragma unroll
        for (int c = 0; c < SBK; ++c)
          if (c <= r) Ls[kb + r][kb + c] = a[c];
      }
      if (bad != 0 && bad <= w && lane == 0) atomicCAS(info, 0, k0 + bad);  // (pivots of the identity padding beyond w cannot fail)
    }
    __syncthreads();
    const int below = NB - kb - SBK;  // rows under the sub-block
    if (below > 0) {
      // (b) X = A[rows below][kb .. kb+32) L11^-T: one row per thread (forward substitution along the row)
      if (tid < below) {
        const int r = kb + SBK + tid;
        double x[SBK];
#pragma unroll
        for (int c = 0; c < SBK; ++c) x[c] = Ls[r][kb + c];
#pragma unroll
        for (int c = 0; c < SBK; ++c) {
          double s = x[c];
#pragma unroll
          for (int t = 0; t < c; ++t) s -= x[t] * Ls[kb + c][kb + t];
          x[c] = s * dinv[kb + c];
        }
#pragma unroll
        for (int c = 0; c < SBK; ++c) Ls[r][kb + c] = x[c];
      }
      __syncthreads();
      // (c) the rest of the block: A[r][c] -= sum_t X[r][t] X[c][t], r >= c > kb + 31; a thread takes four rows (a quarter of the rows apart:
      // consecutive threads read consecutive rows - conflict-free) of one column, so X[c][t] is read once for four products
      const int q4 = below / 4;
      for (int idx = tid; idx < q4 * below; idx += 256) {
        const int rq = idx % q4, cc = idx / q4;
        if (rq + 3 * q4 >= cc) {
          const int c = kb + SBK + cc;
          double s4[4] = {0.0, 0.0, 0.0, 0.0};
#pragma unroll 8
          for (int t = 0; t < SBK; ++t) {
            const double lc = Ls[c][kb + t];
#pragma unroll
            for (int i = 0; i < 4; ++i) s4[i] += Ls[kb + SBK + rq + i * q4][kb + t] * lc;
          }
#pragma unroll
          for (int i = 0; i < 4; ++i) {
            const int rr = rq + i * q4;
            if (rr >= cc) Ls[kb + SBK + rr][c] -= s4[i];
          }
        }
      }
      __syncthreads();
    }
  }
  if (rhs_run != nullptr && wave == 0) {  // L11 x = rhs: rows lane and lane + 64
    double a0 = lane < w ? rhs_run[k0 + lane] : 0.0, a1 = lane + 64 < w ? rhs_run[k0 + 64 + lane] : 0.0;
    for (int t = 0; t < NB; ++t) {  // (identity rows beyond w: their unknowns are 0)
      const double xt = (t < 64 ? readlane_f64(a0, t) : readlane_f64(a1, t - 64)) * dinv[t];
      if (lane == (t & 63)) {
        if (t < 64)
          a0 = xt;
        else
          a1 = xt;
      }
      if (lane > t) a0 -= Ls[lane][t] * xt;
      if (lane + 64 > t) a1 -= Ls[lane + 64][t] * xt;
    }
    if (lane < w) xsol[k0 + lane] = a0;
    if (lane + 64 < w) xsol[k0 + 64 + lane] = a1;
  }
  // L11 back, and the inverses of its diagonal sub-blocks: thread (s, c) solves L_ss x = e_c (32 x 32, the solution in registers)
  for (int idx = tid; idx < NB * NB; idx += 256) {
    const int r = idx % NB, c = idx / NB;
    if (r < w && c < w && r >= c) A[r + (long)c * lda] = Ls[r][c];
  }
  if (tid < NB) {
    const int sblk = tid / SBK, c = tid % SBK, kb = sblk * SBK;
    double xs[SBK];
#pragma unroll
    for (int i = 0; i < SBK; ++i) {
      double v = i == c ? 1.0 : 0.0;
#pragma unroll
      for (int t = 0; t < i; ++t) v -= Ls[kb + i][kb + t] * xs[t];  // (xs[t] = 0 for t < c: the products vanish)
      xs[i] = v * dinv[kb + i];
    }
    double* out = Sinv + (long)sblk * SBK * SBK + (long)c * SBK;  // Sinv[s][i + 32 c]
#pragma unroll
    for (int i = 0; i < SBK; ++i) out[i] = i < c ? 0.0 : xs[i];
  }
}

// L21 = A21 L11^-T, in place.  A21: m x w (leading dimension lda), L11: the factored diagonal block (w x w, same leading dimension), Sinv: the
// inverses of its diagonal sub-blocks.  Workgroup = 64 rows, wave = 16 rows.
// xsol_p / rhs_rows (round 4, both or neither): the block's solved unknowns (k_potrf_leaf) and the running right-hand side of the panel's rows:
// rhs_rows[r] -= L21[r, :] . xsol_p.
__global__ void __launch_bounds__(256) k_potrf_panel(double* A21, long lda, int m, int w, const double* L11, const double* Sinv, const double* xsol_p,
                                                     double* rhs_rows) {
  const int lane = threadIdx.x & 63, x = lane & 15, kk = lane >> 4;
  const int wv = __builtin_amdgcn_readfirstlane((int)(threadIdx.x >> 6));
  const long r = (long)blockIdx.x * 64 + 16 * wv + x;
  constexpr int KS = NB / 4, KSB = SBK / 4, NSB = NB / SBK;  // k-steps of the panel / of a sub-block, sub-blocks
  double b[KS];  // B-operand fragments of this wave's rows: b[ks] = A21[r][4 ks + kk]
#pragma unroll
  for (int ks = 0; ks < KS; ++ks) {
    const int k = 4 * ks + kk;
    b[ks] = (r < m && k < w) ? A21[r + (long)k * lda] : 0.0;
  }
#pragma unroll
  for (int sb_ = 0; sb_ < NSB; ++sb_) {
    if (SBK * sb_ >= w) break;  // uniform
    // A_s - sum_{t < s} X_t L_st^T:  D'[c][r] = sum_k L[c][k] X[r][k], c in block s, k in the blocks before it
    hv4d acc[2];
#pragma unroll
    for (int ct = 0; ct < 2; ++ct) {
      acc[ct] = hv4d{0.0, 0.0, 0.0, 0.0};
      const int c = SBK * sb_ + 16 * ct + x;  // A operand: i = x
#pragma unroll
      for (int ks = 0; ks < KSB * sb_; ++ks) {
        const int k = 4 * ks + kk;
        const double l = (c < w) ? L11[c + (long)k * lda] : 0.0;
        acc[ct] = __builtin_amdgcn_mfma_f64_16x16x4f64(l, b[ks], acc[ct], 0, 0, 0);
      }
#pragma unroll
      for (int reg = 0; reg < 4; ++reg) b[KSB * sb_ + 4 * ct + reg] -= acc[ct][reg];  // accumulator layout == B-operand layout
    }
    // X_s = (that) L_ss^-T:  D'[c][r] = sum_k Sinv_s[c][k] A_s[r][k]  (Sinv_s lower triangular: k <= c)
    hv4d xs[2];
#pragma unroll
    for (int ct = 0; ct < 2; ++ct) {
      xs[ct] = hv4d{0.0, 0.0, 0.0, 0.0};
      const double* sp = Sinv + (long)sb_ * SBK * SBK + 16 * ct + x;
#pragma unroll
      for (int ks = 0; ks < KSB; ++ks) {
        if (4 * ks <= 16 * ct + 15) xs[ct] = __builtin_amdgcn_mfma_f64_16x16x4f64(sp[(4 * ks + kk) * SBK], b[KSB * sb_ + ks], xs[ct], 0, 0, 0);
      }
    }
#pragma unroll
    for (int ct = 0; ct < 2; ++ct)
#pragma unroll
      for (int reg = 0; reg < 4; ++reg) b[KSB * sb_ + 4 * ct + reg] = xs[ct][reg];
  }
#pragma unroll
  for (int ks = 0; ks < KS; ++ks) {
    const int k = 4 * ks + kk;
    if (r < m && k < w) A21[r + (long)k * lda] = b[ks];
  }
  if (xsol_p != nullptr) {  // this row's share of the forward substitution: the lanes x, x + 16, x + 32, x + 48 hold the row's k = kk mod 4
    double sdot = 0.0;
#pragma unroll
    for (int ks = 0; ks < KS; ++ks) {
      const int k = 4 * ks + kk;
      sdot += b[ks] * (k < w ? xsol_p[k] : 0.0);
    }
    sdot += __shfl_xor(sdot, 16, 64);
    sdot += __shfl_xor(sdot, 32, 64);
    if (kk == 0 && r < m) rhs_rows[r] -= sdot;
  }
}

// A22 -= L21 L21^T on the lower triangle: 128 x 128 tiles R >= C of the m x m matrix A22 on the real tile engine of nls_gemm.h (double-buffered
// LDS slices of 16 k, one barrier per slice, 64 x 64 per wave).  L21: m x NB, leading dimension ldl - as a [k][row] plane it is the engine's
// k-major operand for BOTH sides.  The tile is formed transposed (A operand: the tile's columns, B operand: its rows), so the 16 lanes of an
// accumulator register are 16 consecutive rows of one stored column.  The loaders have no bounds tests: the panel must be readable up to a
// multiple of 128 rows (the caller's leading dimension is padded; rows beyond m only feed results that are not stored).
// Tile list: rows R = 0 .. nt - 1, for each the columns C = 0 .. min(R, ncol_tiles - 1) (ncol_tiles = nt: the whole lower triangle; 1: the strip of
// the next panel's columns only).  K = 16 ktiles columns of Lp (round 4: two panels of 128 share ONE pass over the trailing matrix, K = 256 -
// the update is bound by the traffic of the trailing triangle, which halves).
__global__ void __launch_bounds__(Cfg4::NTHREADS, 2) k_potrf_syrk(double* A, long lda, int m, const double* Lp, long ldl, int ktiles, int ncol_tiles) {
  using C4 = Cfg4;
  extern __shared__ double smem[];
  int t = blockIdx.x, R, C;
  const int tri = ncol_tiles * (ncol_tiles + 1) / 2;
  if (t < tri) {
    R = (int)((sqrt(8.0 * t + 1.0) - 1.0) * 0.5);
    while ((R + 1) * (R + 2) / 2 <= t) ++R;
    while (R * (R + 1) / 2 > t) --R;
    C = t - R * (R + 1) / 2;
  } else {
    R = ncol_tiles + (t - tri) / ncol_tiles;
    C = (t - tri) % ncol_tiles;
  }
  v4d acc[C4::MT][C4::NTL];
  zero_acc(acc);
  KMajorLoader<C4::NTHREADS, BM> la{Lp, ldl, (long)C * BM};
  KMajorLoader<C4::NTHREADS, BN> lb{Lp, ldl, (long)R * BN};
  mainloop_real<C4, true>(acc, la, lb, 0, ktiles, smem);
  // read-modify-write of the tile: the 16 entries of an accumulator row block are LOADED together, then stored (written as A[..] -= acc the
  // compiler must assume a store may feed the next load - the same array - and every entry pays its own memory round trip: 64 in a row).
  // A tile inside the matrix and off the diagonal - nearly all of them - takes the copy of the loop without per-entry tests (each test is a
  // scalar branch in front of its load / store).
  auto rmw = [&](auto plainc) {
    constexpr bool PLAIN = decltype(plainc)::value;
#pragma unroll
    for (int mt = 0; mt < C4::MT; ++mt) {
      double old[4][C4::NTL];
#pragma unroll
      for (int reg = 0; reg < 4; ++reg) {
        const long cc = (long)C * BM + C4::acc_row(mt, reg);
#pragma unroll
        for (int nt = 0; nt < C4::NTL; ++nt) {
          const long r = (long)R * BN + C4::acc_col(nt);
          old[reg][nt] = (PLAIN || (r < m && cc < m && r >= cc)) ? A[r + cc * lda] : 0.0;
        }
      }
#pragma unroll
      for (int reg = 0; reg < 4; ++reg) {
        const long cc = (long)C * BM + C4::acc_row(mt, reg);
#pragma unroll
        for (int nt = 0; nt < C4::NTL; ++nt) {
          const long r = (long)R * BN + C4::acc_col(nt);
          if (PLAIN || (r < m && cc < m && r >= cc)) A[r + cc * lda] = old[reg][nt] - acc[mt][nt][reg];
        }
      }
    }
  };
  if (R != C && ((long)R + 1) * BN <= m)  // (C < R: the columns are inside too)
    rmw(std::true_type{});
  else
    rmw(std::false_type{});
}

}  // namespace potrf
}  // namespace nls
