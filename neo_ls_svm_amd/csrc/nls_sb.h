// Two-stage tridiagonalisation, stage 1: Hermitian / real symmetric dense -> band (lower bandwidth B), in place, by panels
// of B columns (P4: eigh(A / c), _neo_ls_svm.py:120; D2: eigh(sn K sn), :265 - the reference calls LAPACK there; the
// algorithm below is this library's own).
//
// Why: the one-stage panel (nls_trd.h) pays one matrix-vector product per COLUMN - two dependent kernel launches per column
// at n <= 6144 (latency bound: 23 us x 4097 columns at c3) and one sweep over the whole lower triangle per column beyond
// (HBM bound: 1.33 TB at real n = 1e4).  Here a PANEL of B columns costs a dozen launches and the trailing matrix is
// read three times per panel by matrix-matrix kernels.
//
// One panel (columns j .. j+kb-1, P = A[j+B:, j:j+kb], m x kb, m = n - j - B, kb = min(B, m - 1)):
//   1. orthogonal factor of P by shifted CholeskyQR3 (three Gram / Cholesky / triangular-multiply passes; the shift of the
//      first pass, 11 (m kb + kb (kb+1)) u trace(G), keeps its Cholesky alive up to kappa(P) ~ 1/u; a well-conditioned panel -
//      unshifted first factorisation with pivots within a factor 100 - takes two passes: CholeskyQR2): k_sb_gram, k_sb_reduce,
//      k_sb_small_chol, k_sb_apply;
//   2. Householder reconstruction (Ballard et al., "Reconstructing Householder vectors from TSQR", 2014): a modified LU
//      of the orthonormal factor gives Y (unit lower trapezoidal) and T (upper triangular) with Q = I - Y T Y^H and
//      P = Q[:, :kb] R: k_sb_small_recon, k_sb_finish;
//   3. two-sided update of A22 = A[j+kb:, j+kb:] (the block of Y zero-padded by B - kb rows on top, so that the narrower LAST
//      panel also transforms the panel columns it does not factor): Z = Y T, W = A22 Z (k_sb_hemm + k_sb_hemm_reduce,
//      both on fp64 MFMA with operands straight from global memory), M = Z^H W (partials from k_sb_hemm_reduce, k_sb_reduce),
//      X = W - Y M / 2 (k_sb_x), A22 -= X Y^H + Y X^H (k_sb_her2k, fp64 MFMA).
// Y stays below the band in A (LAPACK layout: unit diagonal implied at row j + B + c of column j + c), tau1[j + c] = T[c][c]
// (the block is a product of kb elementary reflectors, so the blocked back-transformation of nls_evd.hip rebuilds its
// T^-1 = striu(Y^H Y) + diag(1 / tau) from Y and tau1 alone).
//
// How the panel's kernels are laid out (round 5: a panel is a chain of a dozen dependent launches whose cost is latency, not work):
//   * the row kernels (k_sb_gram / k_sb_apply / k_sb_finish) take 64 rows per workgroup with FOUR threads per row - a quad shares a row's
//     triangular solve (quad_row_solve_upper: the solved entry travels on DPP) - so a panel of 10^4 rows is 157 workgroups, not 40;
//   * the 32 x 32 factorisations of the real path run on ONE wave with the matrix in registers, lane = column (wave_cholesky,
//     wave_modified_lu: v_readlane broadcasts, v_rsq / v_rcp + Newton steps, no LDS, no barrier); the third pass's Cholesky factor comes from
//     its series (G3 = I + E, |E| < 1e-8), the elimination only otherwise (NLS_SB_SERIES=0 forces it);
//   * every cooperative copy requests all of a thread's items before its first store (block_copy), every read-modify-write loads its entries
//     in a batch, and the tile kind / bounds case of the big products is decided per tile, not per load (tools/isa_scan.py).
// Failure: a Cholesky pivot <= 0 / NaN or a second-pass Gram matrix further than 1e-6 from I (kappa(P) beyond ~1e15:
// exactly dependent or zero panel columns, e.g. a diagonal matrix) raises flag[0]; the driver then repeats the reduction on the saved
// copy with every panel perturbed by 1e-13 of its norm (k_sb_perturb) and, if that fails too, hands the copy to the one-stage panel.
// All reductions run over per-block partials in a fixed order: bit-reproducible.
#pragma once
#include "nls_gemm.h"
#include "nls_trd.h"

namespace nls {
namespace sb {
using namespace trd;

constexpr int RC = 64;  // rows per workgroup of the panel kernels

// Dynamic LDS (several kernels need more than the 64 KiB a static allocation may take; the driver opts in with hipFuncSetAttribute).
extern __shared__ __attribute__((aligned(16))) unsigned char sb_smem[];
template <class T, int COLS>
__device__ __forceinline__ T (*sb_carve(int rows, size_t& off))[COLS] {
  T(*p)[COLS] = reinterpret_cast<T(*)[COLS]>(sb_smem + off);
  off += (((size_t)rows * COLS * sizeof(T)) + 15) & ~(size_t)15;
  return p;
}
template <class T, int B>
constexpr size_t sb_mat_bytes(int rows) {
  return (((size_t)rows * (B + 1) * sizeof(T)) + 15) & ~(size_t)15;
}

template <class T>
__device__ __forceinline__ T zero_() {
  return make_<T>(0.0, 0.0);
}
template <class T>
__device__ __forceinline__ T one_() {
  return make_<T>(1.0, 0.0);
}
__device__ __forceinline__ Z neg_(Z a) { return {-a.re, -a.im}; }
__device__ __forceinline__ double neg_(double a) { return -a; }
__device__ __forceinline__ bool finite_(Z a) { return isfinite(a.re) && isfinite(a.im); }
__device__ __forceinline__ bool finite_(double a) { return isfinite(a); }

// ================================================================================================================
// Panel kernels.  Rows are handled 64 per workgroup (a quad of threads per row for the triangular solves), Gram matrices are
// accumulated over one 64-row LDS tile per workgroup, partial Gram matrices are summed by an element-parallel kernel
// (k_sb_reduce) in a fixed order.  The B x B factorisations: one wave, lane = column, for the real B = 32 path (wave_cholesky,
// wave_modified_lu); the generic form (complex, other band widths) keeps the matrix in registers across the workgroup (thread (r, cg)
// owns row r, columns cg + TPR q) with ONE barrier per elimination step (the pivot column / row travels through a double-buffered
// LDS vector).  No explicit inverses: Q = P R^-1 is a row-wise back substitution with R in LDS.
// ================================================================================================================
// Cooperative copy of COUNT items by the 256 threads of a workgroup (item e = threadIdx.x + 256 it): ALL of a thread's loads are issued
// before its first store.  Written as `for (e = threadIdx.x; e < COUNT; e += 256) dst(e) = src(e)` the loop is not unrolled (its trip count
// hangs on threadIdx.x) and every iteration pays its own memory round trip - 4 to 8 in a row at the head of every small kernel of a panel.
template <int COUNT, class LOAD, class STORE>
__device__ __forceinline__ void block_copy(LOAD&& load, STORE&& store) {
  constexpr int NIT = (COUNT + 255) / 256;
  decltype(load(0)) v[NIT];
#pragma unroll
  for (int it = 0; it < NIT; ++it) {
    const int e = (int)threadIdx.x + 256 * it;
    if (COUNT % 256 == 0 || e < COUNT) v[it] = load(e);
  }
#pragma unroll
  for (int it = 0; it < NIT; ++it) {
    const int e = (int)threadIdx.x + 256 * it;
    if (COUNT % 256 == 0 || e < COUNT) store(e, v[it]);
  }
}

constexpr int RW = 64;  // rows per workgroup of the panel kernels: FOUR threads per row (thread t: row t / 4, columns t % 4 + 4 i)

template <int B>
struct Small {
  static constexpr int TPR = 256 / B;  // threads per row
  static constexpr int CPT = B / TPR;  // columns per thread
};

// Per-panel small state in global memory
template <class T, int B>
struct PanelSmall {
  T G[B * B];      // reduced Gram matrix / M = Z^H W (input of the small kernels), i + B j
  T Rs[B * B];     // upper triangular matrix of the next row solve (R1, R2, then M = U R3), identity outside the leading block
  T Racc[B * B];   // R3 R2 R1 so far (upper triangular)
  T Tm[B * B];     // T of the block reflector (upper triangular)
  T Y1[B * B];     // top block of Y (unit lower triangular), explicit
  int skip2;       // the panel is well conditioned: its first pass ran unshifted and the second pass is skipped (CholeskyQR2)
};

// Value of lane SRC of every quad, in all four lanes (DPP quad_perm [SRC, SRC, SRC, SRC]; whole quads must be active)
template <int SRC>
__device__ __forceinline__ double quad_bcast(double v) {
  const int lo = __builtin_amdgcn_update_dpp(0, __double2loint(v), SRC * 0x55, 0xf, 0xf, false);
  const int hi = __builtin_amdgcn_update_dpp(0, __double2hiint(v), SRC * 0x55, 0xf, 0xf, false);
  return __hiloint2double(hi, lo);
}
template <int SRC>
__device__ __forceinline__ Z quad_bcast(Z v) {
  return {quad_bcast<SRC>(v.re), quad_bcast<SRC>(v.im)};
}
template <int CTRL>
__device__ __forceinline__ double quad_perm(double v) {
  const int lo = __builtin_amdgcn_update_dpp(0, __double2loint(v), CTRL, 0xf, 0xf, false);
  const int hi = __builtin_amdgcn_update_dpp(0, __double2hiint(v), CTRL, 0xf, 0xf, false);
  return __hiloint2double(hi, lo);
}
template <int CTRL>
__device__ __forceinline__ Z quad_perm(Z v) {
  return {quad_perm<CTRL>(v.re), quad_perm<CTRL>(v.im)};
}

// out[e] = sum_p part[p * count + e] in a fixed order (bit-reproducible): the four lanes of a quad take the parts p = g, g + 4, ... of one
// element (8 loads in flight each) and their sums are added as (s0 + s1) + (s2 + s3).  Grid: ceil(4 count / 256).  skip: optional device word.
template <class T>
__global__ void __launch_bounds__(256) k_sb_reduce(const T* part, int nparts, int count, T* out, const int* skip) {
  if (skip && *skip) return;  // (the partials of a skipped pass are the previous pass's: out already holds their sum)
  const int e = (blockIdx.x * 256 + threadIdx.x) / 4, g = threadIdx.x % 4;
  T s = zero_<T>();
  if (e < count) {  // uniform over a quad
    int p = g;
    for (; p + 28 < nparts; p += 32) {
      T v[8];
#pragma unroll
      for (int q = 0; q < 8; ++q) v[q] = part[(long)(p + 4 * q) * count + e];
#pragma unroll
      for (int q = 0; q < 8; ++q) s = s + v[q];
    }
    for (; p < nparts; p += 4) s = s + part[(long)p * count + e];
  }
  s = s + quad_perm<0xB1>(s);  // lane ^ 1
  s = s + quad_perm<0x4E>(s);  // lane ^ 2
  if (g == 0 && e < count) out[e] = s;
}

// Gram of the rows held in a 64 x B LDS tile, accumulated into acc (thread (ti, tj): entries (ti + 16 x, tj + 16 y))
template <class T, int B>
__device__ __forceinline__ void gram_tile_accumulate(T (*Ps)[B + 1], T (&acc)[B / 16][B / 16]) {
  constexpr int TI = B / 16;
  const int ti = threadIdx.x % 16, tj = threadIdx.x / 16;
#pragma unroll 4
  for (int r = 0; r < 64; ++r) {
    T a[TI], b[TI];
#pragma unroll
    for (int x = 0; x < TI; ++x) {
      a[x] = conj_(Ps[r][ti + 16 * x]);
      b[x] = Ps[r][tj + 16 * x];
    }
#pragma unroll
    for (int x = 0; x < TI; ++x)
#pragma unroll
      for (int y = 0; y < TI; ++y) acc[x][y] = acc[x][y] + a[x] * b[y];
  }
}
template <class T, int B>
__device__ __forceinline__ void gram_store(const T (&acc)[B / 16][B / 16], T* out) {
  constexpr int TI = B / 16;
  const int ti = threadIdx.x % 16, tj = threadIdx.x / 16;
#pragma unroll
  for (int x = 0; x < TI; ++x)
#pragma unroll
    for (int y = 0; y < TI; ++y) out[(ti + 16 * x) + B * (tj + 16 * y)] = acc[x][y];
}

// ---- panel Gram partials: Gp[block][i + B j] = sum over the block's RW rows of conj(P[r][i]) P[r][j] ------------------
template <class T, int B>
__global__ void __launch_bounds__(256) k_sb_gram(const T* P, long ldp, int m, int kb, T* Gp) {
  constexpr int TI = B / 16;
  __shared__ T Ps[64][B + 1];
  T acc[TI][TI];
#pragma unroll
  for (int x = 0; x < TI; ++x)
#pragma unroll
    for (int y = 0; y < TI; ++y) acc[x][y] = zero_<T>();
  for (int sub = 0; sub < RW / 64; ++sub) {
    const int r0 = blockIdx.x * RW + sub * 64;
    if (r0 >= m) break;  // uniform
    __syncthreads();
    block_copy<64 * B>([&](int idx) { return (r0 + idx % 64 < m && idx / 64 < kb) ? P[(long)(r0 + idx % 64) + (long)(idx / 64) * ldp] : zero_<T>(); },
                       [&](int idx, T v) { Ps[idx % 64][idx / 64] = v; });
    __syncthreads();
    gram_tile_accumulate<T, B>(Ps, acc);
  }
  gram_store<T, B>(acc, Gp + (long)blockIdx.x * B * B);
}

// In-register Cholesky of the leading k x k block: thread (r, cg) holds a[q] = A[r][cg + TPR q] (lower triangle meaningful) and leaves
// L there.  col: double-buffered LDS vector [2][B].  Returns false (uniformly) when a pivot is <= 0 or not finite (the factor is then
// garbage but finite control flow is kept).
template <class T, int B>
__device__ __forceinline__ bool reg_cholesky(T (&a)[Small<B>::CPT], int k, T (*col)[B]) {
  using S = Small<B>;
  const int r = threadIdx.x / S::TPR, cg = threadIdx.x % S::TPR;
  bool ok = true;
#pragma unroll
  for (int p = 0; p < B; ++p) {
    if (p < k) {  // uniform
      if (cg == p % S::TPR) col[p & 1][r] = a[p / S::TPR];
      __syncthreads();
      double d = real_(col[p & 1][p]);
      if (!(d > 0.0) || !isfinite(d)) {
        ok = false;
        d = 1.0;
      }
      const double sq = sqrt(d), inv = 1.0 / sq;
      const T lr = inv * col[p & 1][r];
#pragma unroll
      for (int q = 0; q < S::CPT; ++q) {
        const int c = cg + S::TPR * q;
        if (c > p && c <= r && r < k) {
          T v = a[q] - lr * conj_(inv * col[p & 1][c]);
          if (c == r) v = make_<T>(real_(v), 0.0);
          a[q] = v;
        }
      }
      if (cg == p % S::TPR && r >= p && r < k) a[p / S::TPR] = r == p ? make_<T>(sq, 0.0) : lr;
    }
  }
  return ok;
}

// q R = p for one row per thread: R (upper triangular, identity outside the leading block) in LDS, dinv[j] = 1 / R[j][j].  In place.
// Row j + 1 of R is fetched into registers while row j is being applied (uniform, i.e. broadcast, LDS reads: without the explicit
// double buffer every multiply-add waits for its own read).
template <class T, int B, bool UNIT>
__device__ __forceinline__ void row_solve_upper(T (&x)[B], T (*R)[B + 1], const T* dinv) {
  T rn[B];
#pragma unroll
  for (int c = 1; c < B; ++c) rn[c] = R[0][c];
#pragma unroll
  for (int j = 0; j < B; ++j) {
    T rj[B];
#pragma unroll
    for (int c = j + 1; c < B; ++c) rj[c] = rn[c];
    if (j + 1 < B) {
#pragma unroll
      for (int c = j + 2; c < B; ++c) rn[c] = R[j + 1][c];
    }
    if (!UNIT) x[j] = x[j] * dinv[j];
    const T xj = x[j];
#pragma unroll
    for (int c = j + 1; c < B; ++c) x[c] = x[c] - xj * rj[c];
  }
}

// Row i of X (LDS, B x B) solved in place against the upper triangular R (LDS): x R = x_old.  Own function (not inlined) so that its B
// registers do not add to the caller's live set.
template <class T, int B, bool UNIT>
__device__ __attribute__((noinline)) void lds_row_solve_upper(T (*X)[B + 1], int i, T (*R)[B + 1], const T* dinv) {
  T x[B];
#pragma unroll
  for (int c = 0; c < B; ++c) x[c] = X[i][c];
  row_solve_upper<T, B, UNIT>(x, R, dinv);
#pragma unroll
  for (int c = 0; c < B; ++c) X[i][c] = x[c];
}

// q R = p for one row per QUAD: thread q4 of the quad holds the columns q4 + 4 i of its row (x[i], i < B / 4).  Rq: the upper triangular R
// (identity outside the leading block, exact zeros below the diagonal) in LDS with the columns permuted to the owners' order,
// Rq[j][(B / 4) q4 + i] = R[j][q4 + 4 i] - a thread's B / 4 entries of a row are one contiguous run.  Step j: the owner's x[j] / R[j][j] travels
// through the quad (DPP), every thread updates its columns to the right of j.  A quarter of the thread-per-row form's multiply-adds and
// LDS reads per thread, four times the threads per row: the row blocks of the panel kernels are 64 rows instead of 256.
template <class T, int B>
__device__ __forceinline__ void load_rq(T (*Rq)[B], const T* R /* R[r][c] at r + B c */) {
  constexpr int CQ = B / 4;
  block_copy<B * B>([&](int e) { return R[e]; }, [&](int e, T v) { Rq[e % B][CQ * ((e / B) % 4) + (e / B) / 4] = v; });
}
template <class T, int B, bool UNIT>
__device__ __forceinline__ void quad_row_solve_upper(T (&x)[B / 4], T (*Rq)[B], const T* dinv, int q4) {
  constexpr int CQ = B / 4;
  // Row j + 1 of R is fetched while row j is applied (the scheduling barriers keep the LDS reads ahead of the step's arithmetic: left alone,
  // the compiler issues them where the previous step's registers come free and every step waits out an LDS round trip).
  T rn[CQ];
#pragma unroll
  for (int i = 0; i < CQ; ++i) rn[i] = Rq[0][CQ * q4 + i];
  T dn = UNIT ? T{} : dinv[0];
  static_for<B>([&](auto jc) {
    constexpr int j = decltype(jc)::value, io = j / 4, sl = j % 4;
    T r[CQ];
#pragma unroll
    for (int i = io; i < CQ; ++i) r[i] = rn[i];
    if constexpr (j + 1 < B) {
#pragma unroll
      for (int i = (j + 1) / 4; i < CQ; ++i) rn[i] = Rq[j + 1][CQ * q4 + i];
    }
    const T dj = dn;
    if constexpr (!UNIT && j + 1 < B) dn = dinv[j + 1];
    __builtin_amdgcn_sched_barrier(0);
    T t = x[io];
    if (!UNIT) t = t * dj;
    const T xj = quad_bcast<sl>(t);
    const T u = x[io] - xj * r[io];
    x[io] = q4 == sl ? xj : (q4 > sl ? u : x[io]);  // column j itself: the solved value; its right neighbours in this group of four
#pragma unroll
    for (int i = io + 1; i < CQ; ++i) x[i] = x[i] - xj * r[i];
    __builtin_amdgcn_sched_barrier(0);
  });
}

// ---- one-wave factorisations of a real B x B matrix, lane = column ---------------------------------------------------------------
// The 256-thread forms above pay a workgroup barrier and an LDS round trip per elimination step (0.42 us per step of reg_cholesky, 0.54 per
// step of the modified LU: in-kernel time line, NLS_SB_STAMP).  Here ONE wave keeps the matrix in registers - lane c holds column c, a[r] =
// A[r][c]; lanes >= B mirror lanes < B and write nothing - and a step costs two v_readlane (the pivot column's entry r, a uniform value in
// scalar registers) and one multiply-add per row: no LDS, no barrier, no division (v_rsq / v_rcp + Newton steps).  The pivot column is
// scaled once at the end (its factor is folded into the other operand of the update), so the rounding differs from the classical order in
// the last bits.
__device__ __forceinline__ double rdlane(double v, int l) {
  return __hiloint2double(__builtin_amdgcn_readlane(__double2hiint(v), l), __builtin_amdgcn_readlane(__double2loint(v), l));
}
__device__ __forceinline__ double fast_rsqrt(double d) {  // d > 0, finite
  const double y = __builtin_amdgcn_rsq(d);
  const double e = fma(-(d * y), y, 1.0);
  return fma(y * e, fma(0.375, e, 0.5), y);
}
__device__ __forceinline__ double fast_rcp(double d) {  // |d| >= 1 here
  double y = __builtin_amdgcn_rcp(d);
  double e = fma(-d, y, 1.0);
  y = fma(y, e, y);
  e = fma(-d, y, 1.0);
  return fma(y, e, y);
}
// Cholesky of the leading k x k block of a symmetric matrix held IN FULL (both triangles are updated: the entry L[c][p] a lane needs is its
// own a[p]).  Out: a[r] = L[r][c] for r >= c (r, c < k).  dmin / dmax: extreme diagonal entries of L.  false: a pivot <= 0 or not finite.
template <int B>
__device__ __forceinline__ bool wave_cholesky(double (&a)[B], int k, int lane, double& dmin, double& dmax) {
  bool ok = true;
  double myinv = 1.0, mysq = 1.0;
  dmin = 1e300;
  dmax = 0.0;
  const int c = lane & (B - 1);
  static_for<B>([&](auto pc) {
    constexpr int p = decltype(pc)::value;
    if (p < k) {  // uniform
      double d = rdlane(a[p], p);
      if (!(d > 0.0) || !isfinite(d)) {
        ok = false;
        d = 1.0;
      }
      const double inv = fast_rsqrt(d), sq = d * inv;
      dmin = fmin(dmin, sq);
      dmax = fmax(dmax, sq);
      if (c == p) {
        myinv = inv;
        mysq = sq;
      }
      const double lc2 = a[p] * inv * inv;  // L[c][p] / sqrt(d)
      if (c > p) {
#pragma unroll
        for (int r = p + 1; r < B; ++r) a[r] = fma(-rdlane(a[r], p), lc2, a[r]);
      }
    }
  });
#pragma unroll
  for (int r = 0; r < B; ++r) a[r] = r == c ? mysq : a[r] * myinv;
  return ok;
}
// Modified LU (Householder reconstruction) of Q - S in place: step p takes s = -sign(d) (d = the diagonal entry; -1 for d = 0), pivot d - s
// (|pivot| = |d| + 1: no pivoting needed).  Out: a[r] = L[r][c] (r > c, unit diagonal implied), U[r][c] (r <= c, the pivot on the diagonal);
// lane p holds its step's s in sd.
template <int B>
__device__ __forceinline__ void wave_modified_lu(double (&a)[B], int k, int lane, double& sd) {
  double myipiv = 1.0;
  sd = -1.0;
  const int c = lane & (B - 1);
  static_for<B>([&](auto pc) {
    constexpr int p = decltype(pc)::value;
    if (p < k) {  // uniform
      const double d = rdlane(a[p], p);
      const double s = d > 0.0 ? -1.0 : (d < 0.0 ? 1.0 : -1.0);
      const double piv = d - s, ipiv = fast_rcp(piv);
      if (c == p) {
        myipiv = ipiv;
        sd = s;
        a[p] = piv;
      }
      const double rp = a[p] * ipiv;  // U[p][c] / pivot
      if (c > p) {
#pragma unroll
        for (int r = p + 1; r < B; ++r) a[r] = fma(-rdlane(a[r], p), rp, a[r]);
      }
    }
  });
#pragma unroll
  for (int r = 0; r < B; ++r) a[r] = r > c ? a[r] * myipiv : a[r];
}

// ---- small kernel of passes 1 and 2:  G (+ shift) = R^H R;  ps->Rs = R;  Racc = R Racc -----------------------------------------
// adaptive: pass 0 first factors the UNSHIFTED Gram matrix; if that succeeds with pivots within a factor 100 of each other (kappa(P) of the
// order of 10^2 .. 10^3: one unshifted pass leaves an orthogonality error kappa^2 u <= 1e-10, which the last pass - k_sb_small_recon - removes:
// CholeskyQR2) the second pass is skipped (ps->skip2; its kernels return at once).  Otherwise the shifted three-pass scheme runs as before.
template <class T, int B>
__device__ __forceinline__ void small_chol_generic(int kb, int m, int pass, PanelSmall<T, B>* ps, int* flag, int adaptive) {
  using S = Small<B>;
  size_t off = 0;
  T(*R)[B + 1] = sb_carve<T, B + 1>(B, off);
  T(*Ra)[B + 1] = sb_carve<T, B + 1>(B, off);
  __shared__ T col[2][B];
  __shared__ double sc[1];
  __shared__ int skip_s;
  const int r = threadIdx.x / S::TPR, cg = threadIdx.x % S::TPR;
  T a[S::CPT];
#pragma unroll
  for (int q = 0; q < S::CPT; ++q) a[q] = ps->G[r + B * (cg + S::TPR * q)];
  bool done = false;
  if (pass == 0 && adaptive) {
    const bool ok0 = reg_cholesky<T, B>(a, kb, col);
    __syncthreads();
#pragma unroll
    for (int q = 0; q < S::CPT; ++q)
      if (cg + S::TPR * q == r) col[0][r] = a[q];
    __syncthreads();
    if (threadIdx.x < 64) {
      double mn = 1e300, mx = 0.0;
      for (int i = threadIdx.x; i < kb; i += 64) {
        const double v = real_(col[0][i]);
        mn = fmin(mn, v);
        mx = fmax(mx, v);
      }
#pragma unroll
      for (int msk = 32; msk > 0; msk >>= 1) {
        mn = fmin(mn, __shfl_xor(mn, msk, 64));
        mx = fmax(mx, __shfl_xor(mx, msk, 64));
      }
      if (threadIdx.x == 0) skip_s = (ok0 && mn > 1e-2 * mx) ? 1 : 0;
    }
    __syncthreads();
    done = skip_s != 0;
    if (!done) {
#pragma unroll
      for (int q = 0; q < S::CPT; ++q) a[q] = ps->G[r + B * (cg + S::TPR * q)];
    }
  }
  if (pass == 0 && threadIdx.x == 0) ps->skip2 = done ? 1 : 0;
  if (!done) {
    if (pass == 0) {
      // trace through LDS: the diagonal entry of row r is held by the thread with cg == r % TPR (static register index: compare per q)
      __syncthreads();
#pragma unroll
      for (int q = 0; q < S::CPT; ++q)
        if (cg + S::TPR * q == r) col[0][r] = r < kb ? a[q] : zero_<T>();
      __syncthreads();
      if (threadIdx.x < 64) {
        double tr = 0.0;
        for (int i = threadIdx.x; i < B; i += 64) tr += real_(col[0][i]);
        tr = wave_sum(tr);
        if (threadIdx.x == 0) sc[0] = 11.0 * ((double)m * kb + (double)kb * (kb + 1)) * 1.1102230246251565e-16 * tr + 1e-300;
      }
      __syncthreads();
#pragma unroll
      for (int q = 0; q < S::CPT; ++q)
        if (cg + S::TPR * q == r && r < kb) a[q] = a[q] + make_<T>(sc[0], 0.0);
      __syncthreads();
    }
    const bool ok = reg_cholesky<T, B>(a, kb, col);
    if (!ok && threadIdx.x == 0) flag[0] = 1;
  }
  // R = L^H (upper; identity outside the leading block)
#pragma unroll
  for (int q = 0; q < S::CPT; ++q) {
    const int c = cg + S::TPR * q;
    const bool in = r < kb && c < kb;
    if (in && c <= r) R[c][r] = conj_(a[q]);
    if (in && c < r) R[r][c] = zero_<T>();
    if (!in) R[r][c] = r == c ? one_<T>() : zero_<T>();
    Ra[r][c] = pass == 0 ? (r == c ? one_<T>() : zero_<T>()) : ps->Racc[r + B * c];
  }
  __syncthreads();
  T acc[S::CPT];
#pragma unroll
  for (int q = 0; q < S::CPT; ++q) acc[q] = zero_<T>();
#pragma unroll 2
  for (int t = 0; t < B; ++t) {
    const T x = R[r][t];
#pragma unroll
    for (int q = 0; q < S::CPT; ++q) acc[q] = acc[q] + x * Ra[t][cg + S::TPR * q];
  }
#pragma unroll
  for (int q = 0; q < S::CPT; ++q) {
    const int c = cg + S::TPR * q;
    ps->Racc[r + B * c] = acc[q];
    ps->Rs[r + B * c] = R[r][c];
  }
}

// Real B = 32: the factorisation on one wave (wave_cholesky); the other waves stage Racc meanwhile.
template <int B>
__device__ __forceinline__ void small_chol_wave(int kb, int m, int pass, PanelSmall<double, B>* ps, int* flag, int adaptive) {
  using S = Small<B>;
  size_t off = 0;
  double(*R)[B + 1] = sb_carve<double, B + 1>(B, off);
  double(*Ra)[B + 1] = sb_carve<double, B + 1>(B, off);
  const int wave = threadIdx.x >> 6, lane = threadIdx.x & 63;
  if (wave == 0) {
    const int c = lane & (B - 1);
    double a[B];
    auto load = [&]() {
#pragma unroll
      for (int r = 0; r < B; ++r) a[r] = ps->G[r + B * c];
    };
    load();
    bool done = false;
    double dmin, dmax;
    if (pass == 0 && adaptive) {
      const bool ok0 = wave_cholesky<B>(a, kb, lane, dmin, dmax);
      done = ok0 && dmin > 1e-2 * dmax;
      if (!done) load();
    }
    if (pass == 0 && lane == 0) ps->skip2 = done ? 1 : 0;
    if (!done) {
      if (pass == 0) {
        double tr = (lane < B && c < kb) ? ps->G[c + B * c] : 0.0;
        tr = wave_sum(tr);
        const double shift = 11.0 * ((double)m * kb + (double)kb * (kb + 1)) * 1.1102230246251565e-16 * tr + 1e-300;
#pragma unroll
        for (int r = 0; r < B; ++r) a[r] += (r == c && c < kb) ? shift : 0.0;
      }
      const bool ok = wave_cholesky<B>(a, kb, lane, dmin, dmax);
      if (!ok && lane == 0) flag[0] = 1;
    }
    // R = L^T (upper; identity outside the leading block): lane c writes row c
    if (lane < B) {
#pragma unroll
      for (int r = 0; r < B; ++r) R[c][r] = (c < kb && r < kb) ? (r >= c ? a[r] : 0.0) : (r == c ? 1.0 : 0.0);
    }
  } else if (pass != 0) {
    for (int e = threadIdx.x - 64; e < B * B; e += 192) Ra[e % B][e / B] = ps->Racc[e];
  }
  __syncthreads();
  const int r = threadIdx.x / S::TPR, cg = threadIdx.x % S::TPR;
  if (pass == 0) {  // Racc = R
#pragma unroll
    for (int q = 0; q < S::CPT; ++q) {
      const int c = cg + S::TPR * q;
      ps->Racc[r + B * c] = R[r][c];
      ps->Rs[r + B * c] = R[r][c];
    }
    return;
  }
  double acc[S::CPT];
#pragma unroll
  for (int q = 0; q < S::CPT; ++q) acc[q] = 0.0;
#pragma unroll 2
  for (int t = 0; t < B; ++t) {
    const double x = R[r][t];
#pragma unroll
    for (int q = 0; q < S::CPT; ++q) acc[q] = acc[q] + x * Ra[t][cg + S::TPR * q];
  }
#pragma unroll
  for (int q = 0; q < S::CPT; ++q) {
    const int c = cg + S::TPR * q;
    ps->Racc[r + B * c] = acc[q];
    ps->Rs[r + B * c] = R[r][c];
  }
}

template <class T, int B>
__global__ void __launch_bounds__(256) k_sb_small_chol(int kb, int m, int pass, PanelSmall<T, B>* ps, int* flag, int adaptive) {
  if (pass == 1 && ps->skip2) return;  // uniform
  if constexpr (sizeof(T) == 8 && B == 32)
    small_chol_wave<B>(kb, m, pass, ps, flag, adaptive);
  else
    small_chol_generic<T, B>(kb, m, pass, ps, flag, adaptive);
}

// ---- apply pass: rows of Q = rows of P solved against R (ps->Rs), written to dst; Gram partial of the block's 64 result rows ----------
template <class T, int B>
__global__ void __launch_bounds__(256) k_sb_apply(const T* src, long lds_, int m, int kb, const PanelSmall<T, B>* ps, T* dst, long ldd, T* Gp, int second) {
  constexpr int TI = B / 16, CQ = B / 4;
  if (second && ps->skip2) return;  // the second pass of a well-conditioned panel (k_sb_small_chol)
  size_t off = 0;
  T(*Rq)[B] = sb_carve<T, B>(B, off);
  T(*Ps)[B + 1] = sb_carve<T, B + 1>(64, off);
  __shared__ T dinv[B];
  // (the rows and the diagonal are requested BEFORE the triangular matrix is staged: three groups of loads, one round trip)
  const int rr = threadIdx.x / 4, q4 = threadIdx.x % 4;
  const long row = (long)blockIdx.x * RW + rr;
  T x[CQ];
#pragma unroll
  for (int i = 0; i < CQ; ++i) {
    const int c = q4 + 4 * i;
    x[i] = (row < m && c < kb) ? src[row + (long)c * lds_] : zero_<T>();
  }
  const T dg = threadIdx.x < B ? ps->Rs[threadIdx.x + B * threadIdx.x] : one_<T>();
  load_rq<T, B>(Rq, ps->Rs);
  if (threadIdx.x < B) dinv[threadIdx.x] = inv_(dg);
  __syncthreads();
  quad_row_solve_upper<T, B, false>(x, Rq, dinv, q4);
#pragma unroll
  for (int i = 0; i < CQ; ++i) {
    const int c = q4 + 4 * i;
    if (row < m && c < kb) dst[row + (long)c * ldd] = x[i];
    Ps[rr][c] = x[i];
  }
  __syncthreads();
  T acc[TI][TI];
#pragma unroll
  for (int a = 0; a < TI; ++a)
#pragma unroll
    for (int b = 0; b < TI; ++b) acc[a][b] = zero_<T>();
  gram_tile_accumulate<T, B>(Ps, acc);
  gram_store<T, B>(acc, Gp + (long)blockIdx.x * B * B);
}

// ---- small kernel of pass 3 + Householder reconstruction ----------------------------------------------------------------
// In: ps->G = Q2^H Q2 (Q2 = the twice-orthogonalised panel in Yb), Yb's top kb x kb block.  Out: ps->Rs = M = U R3 (rows of Q2 below the top
// block become rows of Y by one solve against M), ps->Y1, ps->Tm, the kb x kb R factor S R3 R2 R1 written into the band block of A
// (upper triangle; zeros below), tau1.
template <class T, int B>
__global__ void __launch_bounds__(256, 1) k_sb_small_recon(int kb, const T* Ytop, long ldy, PanelSmall<T, B>* ps, T* Aband, long lda, T* tau1, int* flag, long long* stamps, int series) {
  using S = Small<B>;
  constexpr int CQ = B / 4;
  size_t off = 0;
  T(*R3)[B + 1] = sb_carve<T, B + 1>(B, off);   // R3, later U
  T(*X)[B + 1] = sb_carve<T, B + 1>(B, off);    // Racc, later Y1^H
  T(*Q)[B + 1] = sb_carve<T, B + 1>(B, off);    // U1 of the series, Qtop rows (after the solve), later -U S^-1 rows -> T
  T(*Rq)[B] = sb_carve<T, B>(B, off);           // the triangular matrix of a quad row solve, columns in the owners' order
  __shared__ T col[2][B], Sd[B], dinv[B];
  __shared__ int bad, coarse;
  const int r = threadIdx.x / S::TPR, cg = threadIdx.x % S::TPR;
  if (threadIdx.x == 0) bad = 0, coarse = 0;
  if (stamps && threadIdx.x == 0) stamps[0] = wall_clock64();
  T a[S::CPT];
#pragma unroll
  for (int q = 0; q < S::CPT; ++q) a[q] = ps->G[r + B * (cg + S::TPR * q)];
  // (the kernel's other two inputs - the accumulated R factor and the top block of Q2 - are requested now, not where they are used)
  T racc[S::CPT], ytop[CQ];
#pragma unroll
  for (int q = 0; q < S::CPT; ++q) racc[q] = ps->Racc[r + B * (cg + S::TPR * q)];
#pragma unroll
  for (int i = 0; i < CQ; ++i) {
    const int rr = threadIdx.x / 4, c = threadIdx.x % 4 + 4 * i;
    ytop[i] = (threadIdx.x < 4 * B && rr < kb && c < kb) ? Ytop[(long)rr + (long)c * ldy] : zero_<T>();
  }
  __syncthreads();
  {  // orthogonality after the second pass: |G3 - I| < 1e-6, else the third pass cannot finish the job
    double dev = 0.0;
#pragma unroll
    for (int q = 0; q < S::CPT; ++q) {
      const int c = cg + S::TPR * q;
      if (r < kb && c < kb) {
        const double x = abs2_(a[q] - (r == c ? one_<T>() : zero_<T>()));
        dev = (x > dev || !(x == x)) ? x : dev;
      }
    }
    if (!(dev < 1e-12)) bad = 1;
    if (!series || !(dev < 1e-16)) coarse = 1;  // (series = 0, NLS_SB_SERIES=0: always the factorisation proper - the test of that branch)
  }
  __syncthreads();
  if (stamps && threadIdx.x == 0) stamps[1] = wall_clock64();
  bool ok = bad == 0;
  if (coarse) {  // uniform.  |G3 - I| >= 1e-8: the factorisation proper
    ok = reg_cholesky<T, B>(a, kb, col) && ok;
#pragma unroll
    for (int q = 0; q < S::CPT; ++q) {
      const int c = cg + S::TPR * q;
      const bool in = r < kb && c < kb;
      if (in && c <= r) R3[c][r] = conj_(a[q]);
      if (in && c < r) R3[r][c] = zero_<T>();
      if (!in) R3[r][c] = r == c ? one_<T>() : zero_<T>();
    }
  } else {
    // G3 = I + E with |E| < 1e-8 (the usual case: E is the rounding of the second pass): the Cholesky factor by its series, R3 = I + U1 + U2
    // with U1 = striu(E) + diag(E) / 2 and U2 = -(striu(P) + diag(P) / 2), P = U1^H U1 - no elimination steps; the next term is of the
    // order of |E|^3 < 1e-24.
    T u1[S::CPT];
#pragma unroll
    for (int q = 0; q < S::CPT; ++q) {
      const int c = cg + S::TPR * q;
      const bool in = r < kb && c < kb;
      u1[q] = !in ? zero_<T>() : (r < c ? a[q] : (r == c ? make_<T>(0.5 * (real_(a[q]) - 1.0), 0.0) : zero_<T>()));
      Q[r][c] = u1[q];
    }
    __syncthreads();
    T pr[S::CPT];
#pragma unroll
    for (int q = 0; q < S::CPT; ++q) pr[q] = zero_<T>();
#pragma unroll 2
    for (int t = 0; t < B; ++t) {
      const T x = conj_(Q[t][r]);
#pragma unroll
      for (int q = 0; q < S::CPT; ++q) pr[q] = pr[q] + x * Q[t][cg + S::TPR * q];
    }
#pragma unroll
    for (int q = 0; q < S::CPT; ++q) {
      const int c = cg + S::TPR * q;
      const T u2 = r < c ? neg_(pr[q]) : (r == c ? make_<T>(-0.5 * real_(pr[q]), 0.0) : zero_<T>());
      R3[r][c] = (r == c ? one_<T>() : zero_<T>()) + u1[q] + u2;
    }
    __syncthreads();  // (Q is rewritten below)
  }
  if (!ok && threadIdx.x == 0) flag[0] = 1;
  if (stamps && threadIdx.x == 0) stamps[2] = wall_clock64();
#pragma unroll
  for (int q = 0; q < S::CPT; ++q) X[r][cg + S::TPR * q] = racc[q];
  __syncthreads();
  if (threadIdx.x < B) dinv[threadIdx.x] = inv_(R3[threadIdx.x][threadIdx.x]);
  for (int e = threadIdx.x; e < B * B; e += 256) Rq[e / B][CQ * ((e % B) % 4) + (e % B) / 4] = R3[e / B][e % B];
  // Rtot = R3 Racc (kept in registers: rtot[q] = Rtot[r][c])
  T rtot[S::CPT];
#pragma unroll
  for (int q = 0; q < S::CPT; ++q) rtot[q] = zero_<T>();
#pragma unroll 2
  for (int t = 0; t < B; ++t) {
    const T x = R3[r][t];
#pragma unroll
    for (int q = 0; q < S::CPT; ++q) rtot[q] = rtot[q] + x * X[t][cg + S::TPR * q];
  }
  __syncthreads();
  if (stamps && threadIdx.x == 0) stamps[3] = wall_clock64();
  // Qtop = Q2top R3^-1: a quad per row (threads < 4 B)
  if (threadIdx.x < 4 * B) {
    const int rr = threadIdx.x / 4, q4 = threadIdx.x % 4;
    T x[CQ];
#pragma unroll
    for (int i = 0; i < CQ; ++i) x[i] = ytop[i];
    if (stamps) {
      asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
      if (threadIdx.x == 0) stamps[7] = wall_clock64();
    }
    quad_row_solve_upper<T, B, false>(x, Rq, dinv, q4);
    if (stamps && threadIdx.x == 0) stamps[9] = wall_clock64();
#pragma unroll
    for (int i = 0; i < CQ; ++i) Q[rr][q4 + 4 * i] = x[i];
  }
  __syncthreads();
  if (stamps && threadIdx.x == 0) stamps[4] = wall_clock64();
  if constexpr (sizeof(T) == 8 && B == 32) {
    // Modified LU of Qtop - S on one wave, the matrix in registers (wave_modified_lu)
    if (threadIdx.x < 64) {
      const int lane = threadIdx.x, c = lane & (B - 1);
      double w[B], sd;
#pragma unroll
      for (int i = 0; i < B; ++i) w[i] = Q[i][c];
      wave_modified_lu<B>(w, kb, lane, sd);
      if (lane < B) {
#pragma unroll
        for (int i = 0; i < B; ++i) Q[i][c] = w[i];
        Sd[c] = sd;
      }
    }
    __syncthreads();
  } else {
    // Modified LU of Qtop - S, in place in the LDS matrix Q (a plain loop: two barriers per step; unrolled in registers it spills)
    for (int p = 0; p < kb; ++p) {
      const T d = Q[p][p];
      const double ad = sqrt(abs2_(d));
      const T s = ad > 0.0 ? (-1.0 / ad) * d : make_<T>(-1.0, 0.0);
      const T piv = d - s;  // |piv| = |d| + 1: no pivoting needed
      const T mult = r > p && r < kb ? Q[r][p] * inv_(piv) : zero_<T>();
      T upd[S::CPT];
#pragma unroll
      for (int q = 0; q < S::CPT; ++q) {
        const int c = cg + S::TPR * q;
        upd[q] = (c > p && c < kb) ? mult * Q[p][c] : zero_<T>();
      }
      __syncthreads();
      if (r > p && r < kb) {
#pragma unroll
        for (int q = 0; q < S::CPT; ++q) {
          const int c = cg + S::TPR * q;
          if (c > p && c < kb) Q[r][c] = Q[r][c] - upd[q];
        }
        if (cg == p % S::TPR) Q[r][p] = mult;
      }
      if (threadIdx.x == 0) {
        Q[p][p] = piv;
        Sd[p] = s;
      }
      __syncthreads();
    }
  }
#pragma unroll
  for (int q = 0; q < S::CPT; ++q) a[q] = Q[r][cg + S::TPR * q];
  __syncthreads();
  if (stamps && threadIdx.x == 0) stamps[5] = wall_clock64();
  // a now holds L (strict lower, unit diagonal implied) and U (upper incl. diagonal).
  // R factor of the Householder QR: diag(S) Rtot into the band block (upper triangle; zeros below: k_sb_finish puts Y there)
#pragma unroll
  for (int q = 0; q < S::CPT; ++q) {
    const int c = cg + S::TPR * q;
    if (r < kb && c < kb) Aband[(long)r + (long)c * lda] = r <= c ? Sd[r] * rtot[q] : zero_<T>();
  }
  // Y1 (unit lower) -> ps->Y1 and X := Y1^H (unit upper, identity outside; Rq: its image for the quad solve);  Q := U -> later M = U R3
#pragma unroll
  for (int q = 0; q < S::CPT; ++q) {
    const int c = cg + S::TPR * q;
    const bool in = r < kb && c < kb;
    const T y = in ? (r > c ? a[q] : (r == c ? one_<T>() : zero_<T>())) : (r == c ? one_<T>() : zero_<T>());
    ps->Y1[r + B * c] = y;
    X[c][r] = conj_(y);
    Rq[c][CQ * (r % 4) + r / 4] = conj_(y);
    Q[r][c] = in ? (r <= c ? a[q] : zero_<T>()) : (r == c ? one_<T>() : zero_<T>());
  }
  __syncthreads();
  // M = U R3 (upper triangular) -> ps->Rs
  {
    T acc[S::CPT];
#pragma unroll
    for (int q = 0; q < S::CPT; ++q) acc[q] = zero_<T>();
#pragma unroll 2
    for (int t = 0; t < B; ++t) {
      const T x = Q[r][t];
#pragma unroll
      for (int q = 0; q < S::CPT; ++q) acc[q] = acc[q] + x * R3[t][cg + S::TPR * q];
    }
#pragma unroll
    for (int q = 0; q < S::CPT; ++q) ps->Rs[r + B * (cg + S::TPR * q)] = acc[q];
  }
  if (stamps && threadIdx.x == 0) stamps[6] = wall_clock64();
  // T Y1^H = -U S^-1: a quad per row of T solves against the unit upper triangular Y1^H (Rq)
  __syncthreads();
  if (threadIdx.x < 4 * B) {
    const int rr = threadIdx.x / 4, q4 = threadIdx.x % 4;
    T x[CQ];
#pragma unroll
    for (int i = 0; i < CQ; ++i) {
      const int c = q4 + 4 * i;
      x[i] = (rr < kb && c < kb && c >= rr) ? neg_(Q[rr][c] * conj_(Sd[c])) : zero_<T>();
    }
    quad_row_solve_upper<T, B, true>(x, Rq, dinv, q4);
#pragma unroll
    for (int i = 0; i < CQ; ++i) {
      const int c = q4 + 4 * i;
      const T t = (rr < kb && c < kb && c >= rr) ? x[i] : zero_<T>();
      ps->Tm[rr + B * c] = t;
      if (c == rr && rr < kb) tau1[rr] = t;
    }
  }
  if (stamps && threadIdx.x == 0) stamps[8] = wall_clock64();
}

// ---- finish: Y (explicit, m x kb) and Z = Y T into the panel buffers, Y's strictly-lower part into A below the band -----------------
// Row r < kb of Y comes from Y1, rows >= kb are the rows of Q2 solved against M = U R3 (ps->Rs).  Yb / Zb have zh = B - kb zero rows on top;
// Zr is Z once more, row-major with B columns.  64 rows per workgroup, four threads per row.
template <class T, int B>
__global__ void __launch_bounds__(256) k_sb_finish(T* Yb, long ldy, int m, int kb, const PanelSmall<T, B>* ps, T* Zb, T* Zr, T* Apanel, long lda) {
  constexpr int CQ = B / 4;
  size_t off = 0;
  T(*Mq)[B] = sb_carve<T, B>(B, off);
  T(*Ts)[B + 1] = sb_carve<T, B + 1>(B, off);
  T(*Ps)[B + 1] = sb_carve<T, B + 1>(64, off);
  __shared__ T dinv[B];
  const int zh = B - kb;
  // (rows and diagonal requested before the two matrices are staged: one round trip instead of four)
  const int rr = threadIdx.x / 4, q4 = threadIdx.x % 4;
  const long row = (long)blockIdx.x * RW + rr;
  T x[CQ];
#pragma unroll
  for (int i = 0; i < CQ; ++i) {
    const int c = q4 + 4 * i;
    if (row < kb)
      x[i] = ps->Y1[row + B * c];
    else
      x[i] = (row < m && c < kb) ? Yb[(row + zh) + (long)c * ldy] : zero_<T>();
  }
  const T dg = threadIdx.x < B ? ps->Rs[threadIdx.x + B * threadIdx.x] : one_<T>();
  {
    T mv[B * B / 256], tv[B * B / 256];
#pragma unroll
    for (int it = 0; it < B * B / 256; ++it) {
      mv[it] = ps->Rs[threadIdx.x + 256 * it];
      tv[it] = ps->Tm[threadIdx.x + 256 * it];
    }
#pragma unroll
    for (int it = 0; it < B * B / 256; ++it) {
      const int e = threadIdx.x + 256 * it;
      Mq[e % B][CQ * ((e / B) % 4) + (e / B) / 4] = mv[it];
      Ts[e % B][e / B] = tv[it];
    }
  }
  if (threadIdx.x < B) dinv[threadIdx.x] = inv_(dg);
  __syncthreads();
  if (row >= kb) quad_row_solve_upper<T, B, false>(x, Mq, dinv, q4);  // (uniform over a quad)
#pragma unroll
  for (int i = 0; i < CQ; ++i) {
    const int c = q4 + 4 * i;
    if (row < m && c < kb) {
      Yb[(row + zh) + (long)c * ldy] = x[i];
      if (row > c) Apanel[row + (long)c * lda] = x[i];  // below the unit diagonal of Y: below the band of A
    }
    Ps[rr][c] = c < kb ? x[i] : zero_<T>();
  }
  __syncthreads();
  // Z = Y T
  T z[CQ];
#pragma unroll
  for (int c = 0; c < CQ; ++c) z[c] = zero_<T>();
#pragma unroll 2
  for (int t = 0; t < B; ++t) {
    const T y = Ps[rr][t];
#pragma unroll
    for (int c = 0; c < CQ; ++c) z[c] = z[c] + y * Ts[t][q4 + 4 * c];
  }
  if (row < m) {
#pragma unroll
    for (int c = 0; c < CQ; ++c)
      if (q4 + 4 * c < kb) {
        Zb[(row + zh) + (long)(q4 + 4 * c) * ldy] = z[c];
        Zr[(row + zh) * B + q4 + 4 * c] = z[c];  // row-major copy: the operand layout of k_sb_hemm
      }
  }
}

// ---- rescue of a degenerate panel: P += E, |E|_F = 1e-13 |P|_F ----------------------------------------------------------------
// CholeskyQR cannot orthogonalise a panel whose columns are dependent to working precision (a matrix "identity + low rank" runs out of
// rank in the middle of a panel).  The second attempt of the band reduction (driver: evd_two_stage) adds a fixed pseudo-random
// perturbation of relative size 1e-13 to every panel before it is factored: its condition number is then <= ~1e13, which the shifted
// CholeskyQR3 handles, and what is not annihilated below the band - and dropped - is of the size of the perturbation, a backward error
// of 1e-13 |A| per panel.  G = the panel's Gram matrix (its trace gives |P|_F); the hash depends on (row, column, j) only: reproducible.
__device__ __forceinline__ double sb_hash_unit(unsigned a, unsigned b, unsigned c) {  // in [-1, 1)
  unsigned h = a * 0x9E3779B1u ^ (b + 0x7F4A7C15u) * 0x85EBCA77u ^ (c + 0x165667B1u) * 0xC2B2AE3Du;
  h ^= h >> 15;
  h *= 0x2C1B3C6Du;
  h ^= h >> 12;
  h *= 0x297A2D39u;
  h ^= h >> 15;
  return (double)(int)h * (1.0 / 2147483648.0);
}
template <class T>
__global__ void __launch_bounds__(256) k_sb_perturb(T* P, long lda, int m, int kb, const T* G, int ldg, unsigned j) {
  const long idx = blockIdx.x * 256L + threadIdx.x;
  if (idx >= (long)m * kb) return;
  double tr = 0.0;
  for (int c = 0; c < kb; ++c) tr += real_(G[c + (long)ldg * c]);
  const double delta = 1e-13 * sqrt(tr / ((double)m * kb));
  const int r = (int)(idx % m), c = (int)(idx / m);
  P[r + (long)c * lda] = P[r + (long)c * lda] + make_<T>(delta * sb_hash_unit((unsigned)r, (unsigned)c, j), sizeof(T) == 16 ? delta * sb_hash_unit((unsigned)r, (unsigned)c, ~j) : 0.0);
}

// ---- W = A22 Z for the Hermitian A22 stored in the lower triangle ----------------------------------------------------------
// Row block I (64 rows) of the result needs the tiles A[I][J] (J < I), the Hermitian diagonal tile and A[K][I]^H (K > I): NT = mh / 64
// tiles in all, every row block the same number - the lower triangle is read twice per product, all row blocks do equal work.  The
// tiles of a row block are dealt round-robin to PARTS waves whose partial results k_sb_hemm_reduce adds in a fixed order.
// On v_mfma_f64_16x16x4_f64 without LDS and without barriers: a WAVE owns (row block, part): all 64 rows x B columns of its partial
// result (4 x B/16 accumulator tiles) and whole 64 x 64 tiles of A - no operand is fetched by two waves.  Both operands go from global
// memory straight into the fragment registers.  The k index of an MFMA step is free as long as both operands agree, so a lane takes FOUR
// consecutive k (k = 16 q + 4 (lane / 16) + e, e = 0 .. 3): the conjugate-transposed tiles (k runs down a stored column) are then one
// 32-byte (complex: 64-byte) run per lane, whole 128-byte lines per 16 lanes; the tiles left of the diagonal (k runs along a stored row)
// and Z - read from its ROW-major copy Zr[k][c] - are 8-byte loads contiguous over the 16 lanes.  One (tile, q) slice is in flight
// while the previous one multiplies.
constexpr int HT = 64;  // tile edge
typedef double hv4d __attribute__((ext_vector_type(4)));

// Row tiles per wave: 4 (the whole row block) where the accumulators and two slices fit 256 registers (real, B = 32), else 2 - the two
// halves of a row block then go to two waves, which both fetch the slice of Z.
template <class T, int B>
struct HemmCfg {
  static constexpr int RT = (sizeof(T) == 8 && B == 32) ? 4 : 2;
  static constexpr int SLABS = 4 / RT;      // waves that share a row block
  static constexpr int PPW = 4 / SLABS;     // parts per workgroup (blockIdx.y)
};
// fragments of one (tile, q) slice: A operand a[rt][e] (row tile rt), Z operand z[jt][e] (column tile jt)
template <class T, int B>
struct HemmSlice {
  T a[HemmCfg<T, B>::RT][4], z[B / 16][4];
};
// FULL: the tile and the row block lie inside the matrix (no bounds tests).  r0: first row of this wave's slab.
// KIND: 0 tile left of the diagonal (stored as is), 1 below it (read transposed), 2 the diagonal tile (stored half, mirrored).  The kind is
// uniform over the slice and decided ONCE (hemm_load_any): tested per element it put a ladder of scalar branches in front of every load.
template <class T, int B, bool FULL, int KIND>
__device__ __forceinline__ void hemm_load(HemmSlice<T, B>& f, const T* A, long lda, int mh, const T* Zr, long r0, int t, int q, int lane) {
  constexpr int RT = HemmCfg<T, B>::RT;
  const int x = lane & 15, kk = lane >> 4;
  const long k0 = (long)t * HT + 16 * q + 4 * kk;  // first of this lane's four k
#pragma unroll
  for (int rt = 0; rt < RT; ++rt) {
    const long r = r0 + 16 * rt + x;  // row of the result this lane feeds (A operand: i = x)
#pragma unroll
    for (int e = 0; e < 4; ++e) {
      const long k = k0 + e;
      T v = zero_<T>();
      if (FULL || (r < mh && k < mh)) {
        if constexpr (KIND == 0) {
          v = A[r + k * lda];
        } else if constexpr (KIND == 1) {
          v = conj_(A[k + r * lda]);
        } else {  // diagonal tile: the stored half, mirrored
          v = r >= k ? A[r + k * lda] : conj_(A[k + r * lda]);
          if (r == k) v = make_<T>(real_(v), 0.0);
        }
      }
      f.a[rt][e] = v;
    }
  }
#pragma unroll
  for (int jt = 0; jt < B / 16; ++jt)
#pragma unroll
    for (int e = 0; e < 4; ++e) f.z[jt][e] = (FULL || k0 + e < mh) ? Zr[(k0 + e) * B + 16 * jt + x] : zero_<T>();
}
// bounds tests only for the tiles that touch the end of the matrix (uniform choice)
template <class T, int B>
__device__ __forceinline__ void hemm_load_any(HemmSlice<T, B>& f, const T* A, long lda, int mh, const T* Zr, int I, long r0, int t, int q, int lane) {
  const bool full = (long)(max(I, t) + 1) * HT <= mh;
  if (t < I) {
    if (full)
      hemm_load<T, B, true, 0>(f, A, lda, mh, Zr, r0, t, q, lane);
    else
      hemm_load<T, B, false, 0>(f, A, lda, mh, Zr, r0, t, q, lane);
  } else if (t > I) {
    if (full)
      hemm_load<T, B, true, 1>(f, A, lda, mh, Zr, r0, t, q, lane);
    else
      hemm_load<T, B, false, 1>(f, A, lda, mh, Zr, r0, t, q, lane);
  } else {
    if (full)
      hemm_load<T, B, true, 2>(f, A, lda, mh, Zr, r0, t, q, lane);
    else
      hemm_load<T, B, false, 2>(f, A, lda, mh, Zr, r0, t, q, lane);
  }
}
template <int B, int RT, int JT>
__device__ __forceinline__ void hemm_mac(hv4d (&acc)[2][RT][JT], const HemmSlice<double, B>& f) {
#pragma unroll
  for (int e = 0; e < 4; ++e)
#pragma unroll
    for (int rt = 0; rt < RT; ++rt)
#pragma unroll
      for (int jt = 0; jt < JT; ++jt) acc[0][rt][jt] = __builtin_amdgcn_mfma_f64_16x16x4f64(f.a[rt][e], f.z[jt][e], acc[0][rt][jt], 0, 0, 0);
}
template <int B, int RT, int JT>
__device__ __forceinline__ void hemm_mac(hv4d (&acc)[2][RT][JT], const HemmSlice<Z, B>& f) {
#pragma unroll
  for (int e = 0; e < 4; ++e)
#pragma unroll
    for (int rt = 0; rt < RT; ++rt) {
      const double ar = f.a[rt][e].re, ai = f.a[rt][e].im, nai = -ai;
#pragma unroll
      for (int jt = 0; jt < JT; ++jt) {
        acc[0][rt][jt] = __builtin_amdgcn_mfma_f64_16x16x4f64(ar, f.z[jt][e].re, acc[0][rt][jt], 0, 0, 0);
        acc[0][rt][jt] = __builtin_amdgcn_mfma_f64_16x16x4f64(nai, f.z[jt][e].im, acc[0][rt][jt], 0, 0, 0);
        acc[1][rt][jt] = __builtin_amdgcn_mfma_f64_16x16x4f64(ar, f.z[jt][e].im, acc[1][rt][jt], 0, 0, 0);
        acc[1][rt][jt] = __builtin_amdgcn_mfma_f64_16x16x4f64(ai, f.z[jt][e].re, acc[1][rt][jt], 0, 0, 0);
      }
    }
}
// grid (NT, ceil(parts / PPW)), 256 threads: wave w of workgroup (I, y) is slab w % SLABS of row block I, part PPW y + w / SLABS
template <class T, int B>
__global__ void __launch_bounds__(256) k_sb_hemm(const T* A, long lda, int mh, const T* Zr, int kb, int parts, T* Wp) {
  using Cf = HemmCfg<T, B>;
  constexpr bool CX = sizeof(T) == 16;
  constexpr int RT = Cf::RT;
  const int I = blockIdx.x;
  const int lane = threadIdx.x & 63;
  const int w = __builtin_amdgcn_readfirstlane((int)(threadIdx.x >> 6));
  const int part = Cf::PPW * blockIdx.y + w / Cf::SLABS;
  const long r0 = (long)I * HT + 16 * RT * (w % Cf::SLABS);
  if (part >= parts) return;
  const int NT = (mh + HT - 1) / HT;
  hv4d acc[2][RT][B / 16];
#pragma unroll
  for (int p = 0; p < 2; ++p)
#pragma unroll
    for (int rt = 0; rt < RT; ++rt)
#pragma unroll
      for (int jt = 0; jt < B / 16; ++jt) acc[p][rt][jt] = hv4d{0.0, 0.0, 0.0, 0.0};
  const int nmine = part < NT ? (NT - 1 - part) / parts + 1 : 0;  // tiles part, part + parts, ...
  const int items = 4 * nmine;                                     // (tile, q) slices: an even number
  if (items > 0) {
    HemmSlice<T, B> f0, f1;
    hemm_load_any<T, B>(f0, A, lda, mh, Zr, I, r0, part, 0, lane);
    for (int it = 0; it < items; it += 2) {
      hemm_load_any<T, B>(f1, A, lda, mh, Zr, I, r0, part + parts * ((it + 1) >> 2), (it + 1) & 3, lane);
      hemm_mac(acc, f0);
      if (it + 2 < items) hemm_load_any<T, B>(f0, A, lda, mh, Zr, I, r0, part + parts * ((it + 2) >> 2), (it + 2) & 3, lane);
      hemm_mac(acc, f1);
    }
  }
  // D[i][j]: lane = 16 (i % 4) + j, reg = i / 4
  T* out = Wp + ((long)part * mh) * B;  // partial p: [mh][B] row-major
#pragma unroll
  for (int rt = 0; rt < RT; ++rt)
#pragma unroll
    for (int reg = 0; reg < 4; ++reg) {
      const long gr = r0 + 16 * rt + 4 * reg + (lane >> 4);
      if (gr < mh) {
#pragma unroll
        for (int jt = 0; jt < B / 16; ++jt) out[gr * B + 16 * jt + (lane & 15)] = make_<T>(acc[0][rt][jt][reg], CX ? acc[1][rt][jt][reg] : 0.0);
      }
    }
}

// W = sum of the partials (fixed order) -> Wb (column-major, ld);  per row block the partial of M = Z^H W
template <class T, int B>
__global__ void __launch_bounds__(256) k_sb_hemm_reduce(const T* Wp, int split, int mh, int kb, const T* Zb, T* Wb, long ld, T* Mp) {
  constexpr int TI = B / 16;
  size_t off = 0;
  T(*Ws)[B + 1] = sb_carve<T, B + 1>(RC, off);
  T(*Zs)[B + 1] = sb_carve<T, B + 1>(RC, off);
  const int r0 = blockIdx.x * RC;
  // (the block's rows of Z are requested before the partial sums are walked - that loop waits trip by trip)
  T zv[RC * B / 256];
#pragma unroll
  for (int it = 0; it < RC * B / 256; ++it) {
    const int idx = threadIdx.x + 256 * it;
    zv[it] = (r0 + idx % RC < mh && idx / RC < kb) ? Zb[(long)(r0 + idx % RC) + (long)(idx / RC) * ld] : zero_<T>();
  }
  {
    // partials are row-major: thread t adds up the elements (row t / B + (256 / B) u, column t % B), u < RC B / 256, of every partial in the fixed
    // order p = 0, 1, ...; four partials x all its elements are in flight at a time
    constexpr int NU = RC * B / 256, RSTEP = 256 / B;
    const int c = threadIdx.x % B, rb = threadIdx.x / B;
    const long pstride = (long)mh * B;
    const T* wp = Wp + ((long)r0 + rb) * B + c;
    T s[NU];
#pragma unroll
    for (int u = 0; u < NU; ++u) s[u] = zero_<T>();
    auto add_up = [&](auto plainc) {  // PLAIN: the whole row block lies inside the matrix and the panel is full - no test in front of the loads
      constexpr bool PLAIN = decltype(plainc)::value;
      bool valid[NU];
#pragma unroll
      for (int u = 0; u < NU; ++u) valid[u] = PLAIN || (r0 + rb + RSTEP * u < mh && c < kb);
      int p = 0;
      for (; p + 4 <= split; p += 4) {
        T v[4][NU];
#pragma unroll
        for (int q = 0; q < 4; ++q)
#pragma unroll
          for (int u = 0; u < NU; ++u) v[q][u] = valid[u] ? wp[(p + q) * pstride + (long)u * RSTEP * B] : zero_<T>();
#pragma unroll
        for (int q = 0; q < 4; ++q)
#pragma unroll
          for (int u = 0; u < NU; ++u) s[u] = s[u] + v[q][u];
      }
      for (; p < split; ++p) {
        T v[NU];
#pragma unroll
        for (int u = 0; u < NU; ++u) v[u] = valid[u] ? wp[p * pstride + (long)u * RSTEP * B] : zero_<T>();
#pragma unroll
        for (int u = 0; u < NU; ++u) s[u] = s[u] + v[u];
      }
    };
    if (r0 + RC <= mh && kb == B)
      add_up(std::true_type{});
    else
      add_up(std::false_type{});
#pragma unroll
    for (int u = 0; u < NU; ++u) Ws[rb + RSTEP * u][c] = s[u];
  }
#pragma unroll
  for (int it = 0; it < RC * B / 256; ++it) {
    const int idx = threadIdx.x + 256 * it;
    Zs[idx % RC][idx / RC] = zv[it];
  }
  __syncthreads();
  for (int idx = threadIdx.x; idx < RC * B; idx += 256) {
    const int r = idx % RC, c = idx / RC;
    if (r0 + r < mh && c < kb) Wb[(long)(r0 + r) + (long)c * ld] = Ws[r][c];
  }
  const int ti = threadIdx.x % 16, tj = threadIdx.x / 16;
  T acc[TI][TI];
#pragma unroll
  for (int x = 0; x < TI; ++x)
#pragma unroll
    for (int y = 0; y < TI; ++y) acc[x][y] = zero_<T>();
#pragma unroll 4
  for (int r = 0; r < RC; ++r) {
    T a[TI], b[TI];
#pragma unroll
    for (int x = 0; x < TI; ++x) {
      a[x] = conj_(Zs[r][ti + 16 * x]);
      b[x] = Ws[r][tj + 16 * x];
    }
#pragma unroll
    for (int x = 0; x < TI; ++x)
#pragma unroll
      for (int y = 0; y < TI; ++y) acc[x][y] = acc[x][y] + a[x] * b[y];
  }
  T* out = Mp + (long)blockIdx.x * B * B;
#pragma unroll
  for (int x = 0; x < TI; ++x)
#pragma unroll
    for (int y = 0; y < TI; ++y) out[(ti + 16 * x) + B * (tj + 16 * y)] = acc[x][y];
}

// X = W - Y M / 2  (in place over Wb), M = the symmetrised reduced sum of the Z^H W partials
template <class T, int B>
__global__ void __launch_bounds__(256) k_sb_x(T* Wb, const T* Yb, long ld, int mh, int kb, const PanelSmall<T, B>* ps) {
  size_t off = 0;
  T(*Ys)[B + 1] = sb_carve<T, B + 1>(RC, off);
  T(*Ms)[B + 1] = sb_carve<T, B + 1>(B, off);
  const int r0 = blockIdx.x * RC;
  constexpr int CQ = B / 4;
  const int r = threadIdx.x / 4, q4 = threadIdx.x % 4;
  // Everything the block reads - its rows of W (updated in place at the end), its rows of Y, M - is requested up front: one round trip.
  T wold[CQ];
#pragma unroll
  for (int c = 0; c < CQ; ++c) wold[c] = (r0 + r < mh && q4 + 4 * c < kb) ? Wb[(long)(r0 + r) + (long)(q4 + 4 * c) * ld] : zero_<T>();
  constexpr int NY = RC * B / 256, NM = B * B / 256;
  T yv[NY], g1[NM], g2[NM];
#pragma unroll
  for (int it = 0; it < NY; ++it) {
    const int idx = threadIdx.x + 256 * it;
    yv[it] = (r0 + idx % RC < mh && idx / RC < kb) ? Yb[(long)(r0 + idx % RC) + (long)(idx / RC) * ld] : zero_<T>();
  }
#pragma unroll
  for (int it = 0; it < NM; ++it) {
    const int e = threadIdx.x + 256 * it, i = e % B, j = e / B;
    g1[it] = ps->G[i + B * j];
    g2[it] = ps->G[j + B * i];
  }
#pragma unroll
  for (int it = 0; it < NY; ++it) {
    const int idx = threadIdx.x + 256 * it;
    Ys[idx % RC][idx / RC] = yv[it];
  }
  // M = Z^H A22 Z is Hermitian; the reduced sum (ps->G) is symmetrised here (rounding makes its two triangles differ in the last bit)
#pragma unroll
  for (int it = 0; it < NM; ++it) {
    const int e = threadIdx.x + 256 * it, i = e % B, j = e / B;
    T v = 0.5 * (g1[it] + conj_(g2[it]));
    if (i == j) v = make_<T>(real_(v), 0.0);
    Ms[i][j] = v;
  }
  __syncthreads();
  T acc[CQ];
#pragma unroll
  for (int c = 0; c < CQ; ++c) acc[c] = zero_<T>();
#pragma unroll 2
  for (int t = 0; t < B; ++t) {
    const T a = Ys[r][t];
#pragma unroll
    for (int c = 0; c < CQ; ++c) acc[c] = acc[c] + a * Ms[t][q4 + 4 * c];
  }
#pragma unroll
  for (int c = 0; c < CQ; ++c)
    if (r0 + r < mh && q4 + 4 * c < kb) Wb[(long)(r0 + r) + (long)(q4 + 4 * c) * ld] = wold[c] - 0.5 * acc[c];
}

// A22 -= X Y^H + Y X^H on the lower triangle, 64 x 64 tiles R >= C, on v_mfma_f64_16x16x4_f64.  The product is formed TRANSPOSED,
// D[i][j] = update of A[r0 + 16 jt + j][c0 + 16 w + i] (wave w = 16 columns of the tile, jt = 0 .. 3), so that the 16 lanes of an accumulator
// register are 16 consecutive rows of one stored column: the read-modify-write of A moves whole 128-byte lines.
// A operand (i = column of the tile): conj(Y[c][k]), then conj(X[c][k]), from global memory in the fragment layout (16 lanes = 16
// consecutive rows of one panel column; every wave has its own 16 columns).  B operand (j = row): X[r][k], then Y[r][k] - the same 64
// rows for all four waves, staged once in LDS as [k][row] (leading dimension 72 = 8 mod 32 doubles: conflict-free fragment reads).
// The panel columns beyond kb (last, narrow panel) are masked.
constexpr int H2LD = 72;
template <class T, int B>
constexpr size_t her2k_lds_bytes() {
  return 2 * (size_t)B * H2LD * sizeof(T);
}
// FULL: the tile lies inside the matrix; FULLK: the panel has all its B columns (kb == B) - no predicates on the loads then
template <class T, int B, bool FULL, bool FULLK, bool DIAG>
__device__ __forceinline__ void her2k_tile(T* A, long lda, int mh, const T* Xb, const T* Yb, long ld, int kb, long r0, long c0, int w, int lane) {
  constexpr bool CX = sizeof(T) == 16;
  constexpr int KS = B / 4;
  T(*Xs)[H2LD] = reinterpret_cast<T(*)[H2LD]>(sb_smem);
  T(*Ys)[H2LD] = Xs + B;
  block_copy<B * 64>([&](int idx) { return ((FULLK || idx / 64 < kb) && (FULL || r0 + idx % 64 < mh)) ? Xb[r0 + idx % 64 + (long)(idx / 64) * ld] : zero_<T>(); },
                     [&](int idx, T v) { Xs[idx / 64][idx % 64] = v; });
  block_copy<B * 64>([&](int idx) { return ((FULLK || idx / 64 < kb) && (FULL || r0 + idx % 64 < mh)) ? Yb[r0 + idx % 64 + (long)(idx / 64) * ld] : zero_<T>(); },
                     [&](int idx, T v) { Ys[idx / 64][idx % 64] = v; });
  const int x = lane & 15, kk = lane >> 4;
  const long c = c0 + 16 * w + x;  // A operand: column c of the tile
  T a[2][KS];
#pragma unroll
  for (int half = 0; half < 2; ++half) {
    const T* Pa = half == 0 ? Yb : Xb;
#pragma unroll
    for (int ks = 0; ks < KS; ++ks) {
      const int k = 4 * ks + kk;
      a[half][ks] = ((FULLK || k < kb) && (FULL || c < mh)) ? conj_(Pa[c + (long)k * ld]) : zero_<T>();
    }
  }
  __syncthreads();
  hv4d acc[2][4];
#pragma unroll
  for (int p = 0; p < 2; ++p)
#pragma unroll
    for (int jt = 0; jt < 4; ++jt) acc[p][jt] = hv4d{0.0, 0.0, 0.0, 0.0};
#pragma unroll
  for (int half = 0; half < 2; ++half) {
    T(*Pb)[H2LD] = half == 0 ? Xs : Ys;  // row side
#pragma unroll
    for (int jt = 0; jt < 4; ++jt) {
      T b[KS];
#pragma unroll
      for (int ks = 0; ks < KS; ++ks) b[ks] = Pb[4 * ks + kk][16 * jt + x];
#pragma unroll
      for (int ks = 0; ks < KS; ++ks) {
        if constexpr (!CX) {
          acc[0][jt] = __builtin_amdgcn_mfma_f64_16x16x4f64(a[half][ks], b[ks], acc[0][jt], 0, 0, 0);
        } else {
          const double ar = a[half][ks].re, ai = a[half][ks].im, nai = -ai;
          acc[0][jt] = __builtin_amdgcn_mfma_f64_16x16x4f64(ar, b[ks].re, acc[0][jt], 0, 0, 0);
          acc[0][jt] = __builtin_amdgcn_mfma_f64_16x16x4f64(nai, b[ks].im, acc[0][jt], 0, 0, 0);
          acc[1][jt] = __builtin_amdgcn_mfma_f64_16x16x4f64(ar, b[ks].im, acc[1][jt], 0, 0, 0);
          acc[1][jt] = __builtin_amdgcn_mfma_f64_16x16x4f64(ai, b[ks].re, acc[1][jt], 0, 0, 0);
        }
      }
    }
  }
  // D[i][j]: lane = 16 (i % 4) + j, reg = i / 4:  column c0 + 16 w + 4 reg + lane / 16, row r0 + 16 jt + lane % 16
  // (the tile's 16 entries per lane are loaded TOGETHER and then stored: entry by entry, every read-modify-write is its own memory round
  // trip - the compiler keeps a store to A ahead of the next load from A)
  T old[4][4];
#pragma unroll
  for (int reg = 0; reg < 4; ++reg) {
    const long cc = c0 + 16 * w + 4 * reg + kk;
#pragma unroll
    for (int jt = 0; jt < 4; ++jt) {
      const long r = r0 + 16 * jt + x;
      old[reg][jt] = ((FULL || (r < mh && cc < mh)) && (!DIAG || r >= cc)) ? A[r + cc * lda] : zero_<T>();
    }
  }
#pragma unroll
  for (int reg = 0; reg < 4; ++reg) {
    const long cc = c0 + 16 * w + 4 * reg + kk;
#pragma unroll
    for (int jt = 0; jt < 4; ++jt) {
      const long r = r0 + 16 * jt + x;
      if ((FULL || (r < mh && cc < mh)) && (!DIAG || r >= cc)) {
        // D holds sum_k conj(P[cc][k]) Q[r][k] = (X Y^H + Y X^H)[r][cc]
        T v = old[reg][jt] - make_<T>(acc[0][jt][reg], CX ? acc[1][jt][reg] : 0.0);
        if (r == cc) v = make_<T>(real_(v), 0.0);
        A[r + cc * lda] = v;
      }
    }
  }
}
template <class T, int B>
__global__ void __launch_bounds__(256) k_sb_her2k(T* A, long lda, int mh, const T* Xb, const T* Yb, long ld, int kb) {
  constexpr int UT = 64;
  int t = blockIdx.x, R = (int)((sqrt(8.0 * t + 1.0) - 1.0) * 0.5);
  while ((R + 1) * (R + 2) / 2 <= t) ++R;
  while (R * (R + 1) / 2 > t) --R;
  const int C = t - R * (R + 1) / 2;
  const int lane = threadIdx.x & 63;
  const int w = __builtin_amdgcn_readfirstlane((int)(threadIdx.x >> 6));
  const long r0 = (long)R * UT, c0 = (long)C * UT;
  if (r0 + UT <= mh && kb == B && R != C)  // c0 <= r0: the common tile - inside the matrix, off the diagonal, a full panel: no predicates at all
    her2k_tile<T, B, true, true, false>(A, lda, mh, Xb, Yb, ld, kb, r0, c0, w, lane);
  else if (r0 + UT <= mh && R != C)
    her2k_tile<T, B, true, false, false>(A, lda, mh, Xb, Yb, ld, kb, r0, c0, w, lane);
  else if (r0 + UT <= mh)
    her2k_tile<T, B, true, false, true>(A, lda, mh, Xb, Yb, ld, kb, r0, c0, w, lane);
  else if (R != C)
    her2k_tile<T, B, false, false, false>(A, lda, mh, Xb, Yb, ld, kb, r0, c0, w, lane);
  else
    her2k_tile<T, B, false, false, true>(A, lda, mh, Xb, Yb, ld, kb, r0, c0, w, lane);
}

// ---- dense (lower, bandwidth B) -> band storage AB[(i - j) + j * ldab], rows 0 .. 2B (the rows beyond B: room for the bulges, zero) ----
template <class T>
__global__ void k_sb_to_band(const T* A, long lda, int n, int B, T* AB, int ldab) {
  const long idx = blockIdx.x * (long)blockDim.x + threadIdx.x;
  if (idx >= (long)n * ldab) return;
  const int o = (int)(idx % ldab);
  const long j = idx / ldab;
  const long i = j + o;
  T v = make_<T>(0.0, 0.0);
  if (o <= B && i < n) {
    v = A[i + j * lda];
    if (o == 0) v = make_<T>(real_(v), 0.0);
  }
  AB[idx] = v;
}

}  // namespace sb
}  // namespace nls
