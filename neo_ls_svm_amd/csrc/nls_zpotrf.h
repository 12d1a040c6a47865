// Cholesky factorisation A = L L^H of a complex Hermitian positive definite matrix (lower triangle, column-major interleaved complex,
// in place) - the L_ output of the primal fit (cho_factor(gamma* C + A), _neo_ls_svm.py:176-177).
//
// Why not rocsolver_zpotrf: D + 1 = 4097 takes 16.5 ms (18 ms in block columns of 512 on rocBLAS ztrsm / zherk) and 1025 takes 4 ms
// (245 potf2_kernel_small launches) for 1.2 ms / 0.02 ms of arithmetic: a chain of small dependent kernels.  Here, right-looking in panels of
// NBZ = 32 columns (everything here is latency bound: narrow panels of light kernels), TWO launches per panel, and two levels of blocking so that
// the trailing matrix is not streamed once per narrow panel (n^3 / 6 bytes at NBZ = 32: 11.5 GB at n = 4097): a panel's update reaches only to the
// end of its OUTER block column of NBO = 256 columns; the rest of the trailing matrix is updated once per outer block with K = 256
// (the real counterpart is nls_potrf.h):
//   k_zpotrf_panel: every workgroup factors the 32 x 32 diagonal block in LDS (redundantly: cheaper than a kernel boundary) and then solves its
//                   256 rows of L21 = A21 L11^-H by forward substitution along the row, one row per thread, the row in LDS.  Besides L21 (in
//                   place) the kernel writes three real planes of it - Re, Im, -Im as [k][row] - which are the k-major operands of
//   k_zpotrf_herk : A22 -= L21 L21^H on the lower 128 x 128 tiles through the real tile engine (nls_gemm.h), one workgroup per (tile, part):
//                   Re = Lr Lr^T + Li Li^T, Im = Li Lr^T + Lr (-Li)^T - two K = 32 passes of mainloop_real into one accumulator set each.
// A pivot <= 0 (or NaN) raises info = its 1-based index, as LAPACK does; the factorisation carries on with garbage (finite control flow).
#pragma once
#include "nls_gemm.h"
#include "nls_potrf.h"

namespace nls {
namespace zpotrf {

constexpr int NBZ = 32;   // panel width = leaf size
constexpr int NBO = 256;  // outer block column: the trailing matrix beyond it is updated once per NBO columns

// One launch per panel for the diagonal block AND the rows below it.  Every workgroup (256 threads = 256 rows of the panel) first factors the
// 32 x 32 diagonal block itself, in LDS - redundantly: a few microseconds of work against a kernel boundary (the panel is a chain of dependent
// launches, and a launch costs more than the block) - then solves its rows: L21 = A21 L11^-H by forward substitution along the row,
// x[c] = (a[c] - sum_{t < c} x[t] conj(L[c][t])) / L[c][c], one row per thread, the row in LDS ([t][thread]: conflict-free), the factor's
// entries the same for every lane (broadcast reads).  Plain loops, a dozen registers (a register-resident form with v_readlane broadcasts
// - the real leaf's design - spills ~1400 scalar registers here: every lane predicate r == k, c <= r is a 64-bit mask).
// Workgroup 0 hands L11 over (strict upper part untouched; see L11out below) and raises info (0 or the global 1-based index of the first bad pivot).
// Out besides L21 (in place): Pr, Pi, Pn: [NBZ][ldp] planes (Re, Im, -Im of L21, row index fastest), zero for k >= w and for rows m .. m_pad - 1.
constexpr int ZP_ROWS = 256;
constexpr size_t ZP_LDS = (size_t)2 * NBZ * (NBZ + 1) * sizeof(double) + (size_t)NBZ * ZP_ROWS * sizeof(double2) + NBZ * sizeof(double);
__global__ void __launch_bounds__(ZP_ROWS) k_zpotrf_panel(double2* __restrict__ A, long lda, int w, int k0, int m, int m_pad, double* __restrict__ Pr,
                                                          double* __restrict__ Pi, double* __restrict__ Pn, long ldp, double* __restrict__ Or,
                                                          double* __restrict__ Oi, double* __restrict__ On, int orow0, double2* __restrict__ L11out,
                                                          int* info) {
  extern __shared__ __attribute__((aligned(16))) unsigned char zp_smem[];
  double(*Lr)[NBZ + 1] = reinterpret_cast<double(*)[NBZ + 1]>(zp_smem);
  double(*Li)[NBZ + 1] = reinterpret_cast<double(*)[NBZ + 1]>(zp_smem + (size_t)NBZ * (NBZ + 1) * sizeof(double));
  double2(*xs)[ZP_ROWS] = reinterpret_cast<double2(*)[ZP_ROWS]>(zp_smem + (size_t)2 * NBZ * (NBZ + 1) * sizeof(double));
  double* dv = reinterpret_cast<double*>(zp_smem + (size_t)2 * NBZ * (NBZ + 1) * sizeof(double) + (size_t)NBZ * ZP_ROWS * sizeof(double2));
  const int tid = threadIdx.x;
  // ---- the diagonal block (rows / columns k0 .. k0 + w of A), identity-padded to 32 x 32 ----
  for (int idx = tid; idx < NBZ * NBZ; idx += ZP_ROWS) {
    const int r = idx % NBZ, c = idx / NBZ;
    double2 v = make_double2(r == c ? 1.0 : 0.0, 0.0);
    if (r < w && c < w && r >= c) v = A[r + (long)c * lda];
    Lr[r][c] = v.x;
    Li[r][c] = r == c ? 0.0 : v.y;  // the diagonal of a Hermitian matrix is real (LAPACK ignores its imaginary part too)
  }
  __syncthreads();
  int bad = 0;
  for (int k = 0; k < NBZ; ++k) {
    double d = Lr[k][k];
    if (!(d > 0.0) || !isfinite(d)) {  // uniform
      if (bad == 0) bad = k + 1;
      d = 1.0;
    }
    const double sq = sqrt(d), inv = 1.0 / sq;
    __syncthreads();  // everybody has read the pivot
    if (tid < NBZ && tid >= k) {
      Lr[tid][k] = tid == k ? sq : Lr[tid][k] * inv;
      Li[tid][k] = tid == k ? 0.0 : Li[tid][k] * inv;
    }
    if (tid == 0) dv[k] = inv;
    __syncthreads();
    for (int idx = tid; idx < NBZ * NBZ; idx += ZP_ROWS) {  // a[r][c] -= l[r] conj(l[c]), r >= c > k
      const int r = idx % NBZ, c = idx / NBZ;
      if (c > k && r >= c) {
        const double lr = Lr[r][k], li = Li[r][k], cr = Lr[c][k], ci = Li[c][k];
        Lr[r][c] -= lr * cr + li * ci;
        Li[r][c] = c == r ? 0.0 : Li[r][c] - (li * cr - lr * ci);
      }
    }
    __syncthreads();
  }
  if (blockIdx.x == 0) {
    // L11 goes back into A directly only when no other workgroup exists that may still be reading the un-factored block; otherwise into
    // L11out, from where the update kernel (the next launch) puts it in place.
    double2* dst = gridDim.x == 1 ? A : L11out;
    const long ldd = gridDim.x == 1 ? lda : NBZ;
    for (int idx = tid; idx < NBZ * NBZ; idx += ZP_ROWS) {
      const int r = idx % NBZ, c = idx / NBZ;
      if (r < w && c <= r) dst[r + (long)c * ldd] = make_double2(Lr[r][c], Li[r][c]);
    }
    if (bad != 0 && bad <= w && tid == 0) atomicCAS(info, 0, k0 + bad);  // (pivots of the identity padding beyond w cannot fail)
  }
  // ---- this workgroup's rows of the panel below the block ----
  const long r = (long)blockIdx.x * ZP_ROWS + tid;
  if (r >= m_pad) return;
  const bool live = r < m;
  double2* A21 = A + w;
  for (int c = 0; c < NBZ; ++c) xs[c][tid] = (live && c < w) ? A21[r + (long)c * lda] : make_double2(0.0, 0.0);  // (32 independent loads in flight)
  for (int c = 0; c < NBZ; ++c) {
    double2 x = xs[c][tid];
    double sr = x.x, si = x.y;
#pragma unroll 4
    for (int t = 0; t < c; ++t) {  // x[c] -= x[t] conj(L[c][t])
      const double lr = Lr[c][t], li = Li[c][t];
      x = xs[t][tid];
      sr = fma(-x.x, lr, sr);
      sr = fma(-x.y, li, sr);
      si = fma(-x.y, lr, si);
      si = fma(x.x, li, si);
    }
    xs[c][tid] = make_double2(sr * dv[c], si * dv[c]);  // (read back by this thread only: no barrier)
  }
  for (int c = 0; c < NBZ; ++c) {
    const bool keep = live && c < w;
    const double2 x = xs[c][tid];
    if (keep) A21[r + (long)c * lda] = x;
    Pr[(long)c * ldp + r] = keep ? x.x : 0.0;
    Pi[(long)c * ldp + r] = keep ? x.y : 0.0;
    Pn[(long)c * ldp + r] = keep ? -x.y : 0.0;
    // the same entries as rows of the OUTER block column's planes (rows below the outer block only: index r - orow0; the caller has
    // zeroed those planes, padding included; Or .. On already point at this panel's 32 k-rows)
    if (keep && r >= orow0) {
      Or[(long)c * ldp + (r - orow0)] = x.x;
      Oi[(long)c * ldp + (r - orow0)] = x.y;
      On[(long)c * ldp + (r - orow0)] = -x.y;
    }
  }
}

// The factored diagonal block put in place when no update kernel follows the panel kernel (last panel of an outer block column).
__global__ void __launch_bounds__(256) k_zpotrf_putback(const double2* L11src, double2* L11dst, long lda, int w) {
  for (int idx = threadIdx.x; idx < NBZ * NBZ; idx += 256) {
    const int r = idx % NBZ, c = idx / NBZ;
    if (r < w && c <= r) L11dst[r + (long)c * lda] = L11src[r + c * NBZ];
  }
}

// A22 -= L21 L21^H on the lower triangle: 128 x 128 tiles (R, C), R >= C, C < ncol_tiles, of the m x m matrix A22 (interleaved complex, leading
// dimension lda; only columns < ncols are touched) on the real tile engine.  blockIdx.y = 0: real part, 1: imaginary part.  The tile is formed
// transposed (A operand: the tile's columns, B operand: its rows), as in k_potrf_syrk.  Planes: [ktiles * 16][ldp], readable (zero) up to a multiple
// of 128 rows and of 16 k.  ncol_tiles < the number of row tiles gives the tall update inside an outer block column.
__global__ void __launch_bounds__(Cfg4::NTHREADS, 2) k_zpotrf_herk(double2* A, long lda, int m, int ncols, int ncol_tiles, int ktiles, const double* Pr,
                                                                   const double* Pi, const double* Pn, long ldp, const double2* L11src, double2* L11dst, int w) {
  using C4 = Cfg4;
  extern __shared__ double smem[];
  if (L11src && blockIdx.x == 0 && blockIdx.y == 0) {  // the panel kernel's factored diagonal block, put in place (k_zpotrf_panel)
    for (int idx = threadIdx.x; idx < NBZ * NBZ; idx += C4::NTHREADS) {
      const int r = idx % NBZ, c = idx / NBZ;
      if (r < w && c <= r) L11dst[r + (long)c * lda] = L11src[r + c * NBZ];
    }
  }
  // tile list: rows R = 0 .. nt - 1, for each the columns C = 0 .. min(R, ncol_tiles - 1): the first ncol_tiles rows form a triangle
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
  const int part = blockIdx.y;
  v4d acc[C4::MT][C4::NTL];
  zero_acc(acc);
  // (L L^H)[r][c] = sum_k (ar br + ai bi) + i (ai br - ar bi) with a = L[r][k], b = L[c][k]; A operand <-> c, B operand <-> r
  {
    KMajorLoader<C4::NTHREADS, BM> la{Pr, ldp, (long)C * BM};
    KMajorLoader<C4::NTHREADS, BN> lb{part == 0 ? Pr : Pi, ldp, (long)R * BN};
    mainloop_real<C4, true>(acc, la, lb, 0, ktiles, smem);
  }
  __syncthreads();
  {
    KMajorLoader<C4::NTHREADS, BM> la{part == 0 ? Pi : Pn, ldp, (long)C * BM};
    KMajorLoader<C4::NTHREADS, BN> lb{part == 0 ? Pi : Pr, ldp, (long)R * BN};
    mainloop_real<C4, true>(acc, la, lb, 0, ktiles, smem);
  }
  double* Ad = reinterpret_cast<double*>(A) + part;
#pragma unroll
  for (int mt = 0; mt < C4::MT; ++mt)
#pragma unroll
    for (int reg = 0; reg < 4; ++reg) {
      const long cc = (long)C * BM + C4::acc_row(mt, reg);
#pragma unroll
      for (int nt = 0; nt < C4::NTL; ++nt) {
        const long r = (long)R * BN + C4::acc_col(nt);
        if (r < m && cc < ncols && (part == 0 ? r >= cc : r > cc)) Ad[2 * (r + cc * lda)] -= acc[mt][nt][reg];
      }
    }
}

}  // namespace zpotrf
}  // namespace nls
