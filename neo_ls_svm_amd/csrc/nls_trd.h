// Householder tridiagonalisation of a Hermitian / real symmetric matrix (lower triangle, column-major), LAPACK
// zhetrd / dsytrd conventions for d, e, tau and the reflectors left below the sub-diagonal - the serial section of
// the path (P4: eigh(A / c), _neo_ls_svm.py:120; D2: eigh(sn K sn), :265).
//
// Why not rocsolver_zhetrd: its panel runs FIVE kernels per column (a 16 us matrix-vector product and four 4-5 us
// vector kernels, profiles/r01b_c3_kernel_stats_summary.md) - 140 of the 219 ms of zheevd(4097) and 25 of the 28 ms
// of zheevd(1025).  Here a column is THREE kernels - or TWO, see k_trd_hemv2 / k_trd_finish2 below, which fold the first
// one into its neighbours - and the matrix-vector product reads the lower triangle once (it sits in the 256 MB
// Infinity Cache: 134 MB at n = 4097):
//   k_trd_column : x = A[j:, j] - V W[j, :]^H - W V[j, :]^H   (blocked zlatrd update of the column), d[j], |x|^2 partials
//   k_trd_hemv   : y = A22 x on 64 x 64 tiles of the lower triangle - every tile adds its A x contribution to its
//                  rows and its A^H x contribution to its columns - plus the short products W^H x, V^H x
//   k_trd_finish : larfg scalars (beta, tau), v, w' = tau (A22 v - V (W^H v) - W (V^H v)), partials of w'^H v
// The reflector is never formed before the product: with alpha = x[j+1], v = (x - beta e1) / (alpha - beta) and
// A22 v = (A22 x - beta A22[:, 0]) / (alpha - beta), so the product runs on the un-normalised column while the norm
// is still being reduced.  The last step of zlatrd, w = w' - tau/2 (w'^H v) v, needs a global scalar and is applied
// by the NEXT column's k_trd_column.  All reductions go through per-block partials summed in a fixed order: results
// are bit-reproducible.  After nb = 32 columns the trailing matrix gets the rank-2nb update (rocBLAS her2k / syr2k).
// LAPACK's rescaling loop for |beta| < safmin is replaced by the driver-level scaling of the whole matrix into
// an O(1) range (evd_prescale in nls_evd.hip; zheev / dsyev scale at the driver level too).
#pragma once
#include <hip/hip_runtime.h>

namespace nls {
namespace trd {

struct Z {
  double re, im;
};
__device__ __forceinline__ Z operator+(Z a, Z b) { return {a.re + b.re, a.im + b.im}; }
__device__ __forceinline__ Z operator-(Z a, Z b) { return {a.re - b.re, a.im - b.im}; }
__device__ __forceinline__ Z operator*(Z a, Z b) { return {a.re * b.re - a.im * b.im, a.re * b.im + a.im * b.re}; }
__device__ __forceinline__ Z operator*(double s, Z a) { return {s * a.re, s * a.im}; }
__device__ __forceinline__ Z conj_(Z a) { return {a.re, -a.im}; }
__device__ __forceinline__ double conj_(double a) { return a; }
__device__ __forceinline__ double real_(Z a) { return a.re; }
__device__ __forceinline__ double real_(double a) { return a; }
__device__ __forceinline__ double imag_(Z a) { return a.im; }
__device__ __forceinline__ double imag_(double) { return 0.0; }
__device__ __forceinline__ double abs2_(Z a) { return a.re * a.re + a.im * a.im; }
__device__ __forceinline__ double abs2_(double a) { return a * a; }
template <class T>
__device__ __forceinline__ T make_(double re, double im);
template <>
__device__ __forceinline__ Z make_<Z>(double re, double im) { return {re, im}; }
template <>
__device__ __forceinline__ double make_<double>(double re, double) { return re; }
__device__ __forceinline__ Z inv_(Z a) {  // 1 / a
  const double s = 1.0 / (a.re * a.re + a.im * a.im);
  return {a.re * s, -a.im * s};
}
__device__ __forceinline__ double inv_(double a) { return 1.0 / a; }
// component-wise select: a ternary on the struct itself is lowered through memory and keeps the arrays in scratch
__device__ __forceinline__ Z sel_(bool c, Z a, Z b) { return {c ? a.re : b.re, c ? a.im : b.im}; }
__device__ __forceinline__ double sel_(bool c, double a, double b) { return c ? a : b; }
__device__ __forceinline__ Z shfl_xor_(Z a, int m) { return {__shfl_xor(a.re, m, 64), __shfl_xor(a.im, m, 64)}; }
__device__ __forceinline__ double shfl_xor_(double a, int m) { return __shfl_xor(a, m, 64); }
// value of lane `src` (a compile-time or wave-uniform index) as a scalar broadcast: no vector register, no memory
__device__ __forceinline__ double lane_bcast(double v, int src) {
  return __hiloint2double(__builtin_amdgcn_readlane(__double2hiint(v), src), __builtin_amdgcn_readlane(__double2loint(v), src));
}
__device__ __forceinline__ Z lane_bcast(Z v, int src) { return {lane_bcast(v.re, src), lane_bcast(v.im, src)}; }
// Cross-lane sums on the DPP path (quad_perm / row_half_mirror / row_mirror inside a row of 16 lanes, v_readlane across rows): a
// __shfl_xor goes through ds_bpermute (an LDS round trip per step and 32-bit half).  Whole rows of 16 lanes must be active.
template <int CTRL>
__device__ __forceinline__ double dpp_mov(double v) {
  const int lo = __builtin_amdgcn_update_dpp(0, __double2loint(v), CTRL, 0xf, 0xf, false);
  const int hi = __builtin_amdgcn_update_dpp(0, __double2hiint(v), CTRL, 0xf, 0xf, false);
  return __hiloint2double(hi, lo);
}
template <int CTRL>
__device__ __forceinline__ Z dpp_mov(Z v) {
  return {dpp_mov<CTRL>(v.re), dpp_mov<CTRL>(v.im)};
}
// sum over the 16 lanes of a DPP row, the same value in all of them
template <class T>
__device__ __forceinline__ T row16_sum(T v) {
  v = v + dpp_mov<0xB1>(v);   // quad_perm [1,0,3,2]: lane ^ 1
  v = v + dpp_mov<0x4E>(v);   // quad_perm [2,3,0,1]: lane ^ 2
  v = v + dpp_mov<0x141>(v);  // row_half_mirror: the other quad of the 8
  v = v + dpp_mov<0x140>(v);  // row_mirror: the other half of the 16
  return v;
}
__device__ __forceinline__ double wave_sum_dpp(double v) {
  v = row16_sum(v);
  return lane_bcast(v, 0) + lane_bcast(v, 16) + lane_bcast(v, 32) + lane_bcast(v, 48);
}
__device__ __forceinline__ Z wave_sum_dpp(Z v) { return {wave_sum_dpp(v.re), wave_sum_dpp(v.im)}; }
// sum over the wave, the same value in every lane (round 5: on DPP; the whole wave must be active)
template <class T>
__device__ __forceinline__ T wave_sum(T v) {
  return wave_sum_dpp(v);
}

constexpr int NB = 32;    // panel width
constexpr int TS = 64;    // hemv tile edge
constexpr int TPR = 16;   // threads per row in the row kernels (they split the panel columns / partial strips)
constexpr int ROWT = 16;  // rows per block of the row kernels (256 threads)
template <class T>
__device__ __forceinline__ T row_sum(T v) {  // sum over the TPR consecutive lanes of one row, same value in all of them
  static_assert(TPR == 16, "a row of the row kernels is a DPP row");
  return row16_sum(v);
}

// larfg scalars from alpha = x[j+1] and xnorm^2 = |x[j+2:]|^2  (zlarfg / dlarfg)
template <class T>
struct Larfg {
  double beta;
  T tau, scale;  // scale = 1 / (alpha - beta)
  bool identity;  // H = I (tau = 0)
};
template <class T>
__device__ __forceinline__ Larfg<T> larfg(T alpha, double xnorm2) {
  Larfg<T> h;
  const double ar = real_(alpha), ai = imag_(alpha);
  if (xnorm2 == 0.0 && ai == 0.0) {
    h.identity = true;
    h.beta = ar;
    h.tau = make_<T>(0.0, 0.0);
    h.scale = make_<T>(0.0, 0.0);
    return h;
  }
  h.identity = false;
  const double s2 = ar * ar + ai * ai + xnorm2;
  if (s2 > 1e-280 && s2 < 1e280) {
    // The reflector's scalars sit on the critical path of every column (one-stage) / stage (bulge chase): v_rsq_f64 / v_rcp_f64 with Newton
    // steps (results within an ulp or two) instead of the IEEE sqrt and two divisions - a third of the dependent instructions.
    const double y0 = __builtin_amdgcn_rsq(s2);
    const double e0 = fma(-(s2 * y0), y0, 1.0);
    const double y = fma(y0 * e0, fma(0.375, e0, 0.5), y0);  // 1 / sqrt(s2)
    const double nrm = s2 * y;
    h.beta = ar >= 0.0 ? -nrm : nrm;
    const double ib = ar >= 0.0 ? -y : y;  // 1 / beta
    h.tau = make_<T>((h.beta - ar) * ib, -ai * ib);
    // 1 / (alpha - beta): |alpha - beta| >= |beta|, no cancellation (beta has the opposite sign of Re alpha)
    const double dr = ar - h.beta, d2 = dr * dr + ai * ai;
    double z = __builtin_amdgcn_rcp(d2);
    double ez = fma(-d2, z, 1.0);
    z = fma(z, ez, z);
    ez = fma(-d2, z, 1.0);
    z = fma(z, ez, z);
    h.scale = make_<T>(dr * z, -ai * z);
    return h;
  }
  const double nrm = sqrt(s2);
  h.beta = ar >= 0.0 ? -nrm : nrm;
  h.tau = make_<T>((h.beta - ar) / h.beta, -ai / h.beta);
  h.scale = inv_(alpha - make_<T>(h.beta, 0.0));
  return h;
}

template <class T>
struct Args {
  T* A;      // n x n column-major, lower triangle; columns < j hold the reflectors (explicit 1 on the sub-diagonal inside the panel)
  long lda;
  int n;
  T* W;       // n x NB
  T* wtmp;    // n: w' of the previous column (before the - tau/2 (w'^H v) v step)
  T* xvec;    // n: the current column's x for rows > j, 0 above
  T* ylow;    // [strips][n]  A x contributions per column strip
  T* yup;     // [strips][n]  A^H x contributions per row strip
  T* zpart;   // [dot blocks][2 NB]  partials of W^H x (first NB) and V^H x
  T* spart;   // [row blocks]  partials of w'^H v of the previous column
  double* pnorm;  // [row blocks]  partials of |x[j+2:]|^2
  double* d;
  double* e;
  T* tau;
  int j0, j;  // panel start, current column
  int nrowblocks, ndot;
  int dotgroups;  // 64-row groups per dot block of k_trd_hemv
  // two-kernels-per-column variant (k_trd_hemv2 / k_trd_finish2)
  T* bvec;          // n: column j without the terms that need the previous column's global scalars (see k_trd_finish2)
  T* wtmp_prev;     // w' of column j - 1 (wtmp holds the one being written)
  T* spart_prev;    // partials of w'^H v of column j - 1 (spart holds the ones being written)
  int make_base;    // k_trd_finish2: also prepare bvec for column j + 1 (same panel)
  int boustrophedon;  // reverse the tile order of the matrix-vector product on odd columns (large n)
};

// Sum of cnt per-block partials, computed by the first wave of the block in a fixed order (lane-strided, then a
// shuffle tree) and broadcast through shared memory.  Call with all threads of the block.
template <class T>
__device__ __forceinline__ T sum_partials(const T* p, int cnt, T* slot) {
  if (threadIdx.x < 64) {
    // (eight partials per lane requested together - at n = 4097 there are 257: one load per loop trip was five dependent round trips at the
    // head of EVERY block of the matrix-vector kernel; the order of the additions is still fixed)
    T s = make_<T>(0.0, 0.0);
    for (int b0 = 0; b0 < cnt; b0 += 512) {
      T v[8];
#pragma unroll
      for (int u = 0; u < 8; ++u) {
        const int b = b0 + 64 * u + (int)threadIdx.x;
        v[u] = b < cnt ? p[b] : make_<T>(0.0, 0.0);
      }
#pragma unroll
      for (int u = 0; u < 8; ++u) s = s + v[u];
    }
    s = wave_sum(s);
    if (threadIdx.x == 0) *slot = s;
  }
  __syncthreads();
  return *slot;
}

// ---- k1: column update -------------------------------------------------------------------------
// 256 threads = 32 rows x TPR sub-lanes; sub-lane q takes the panel columns p = q, q + TPR, ...
// Every global load that does not depend on the reduced scalar is issued before the first barrier.
template <class T>
__global__ void __launch_bounds__(ROWT * TPR) k_trd_column(Args<T> a) {
  __shared__ double red[ROWT];
  __shared__ T wj[NB], vj[NB], slot;
  constexpr int PPT = NB / TPR;  // panel columns per sub-lane
  const int j = a.j, i = a.j - a.j0, n = a.n;
  const int q = threadIdx.x % TPR, rl = threadIdx.x / TPR;
  const long r = (long)blockIdx.x * ROWT + rl;
  const bool live = r < n && r >= j;
  // ---- loads
  T sp = make_<T>(0.0, 0.0);
  if (i > 0 && threadIdx.x < 64)
    for (int b = threadIdx.x; b < a.nrowblocks; b += 64) sp = sp + a.spart[b];
  const T tau_prev = i > 0 ? a.tau[j - 1] : make_<T>(0.0, 0.0);
  T wrow = make_<T>(0.0, 0.0), vrow = make_<T>(0.0, 0.0);  // row j of W', V for p = threadIdx.x
  if (threadIdx.x < i) {
    wrow = threadIdx.x == i - 1 ? a.wtmp[j] : a.W[(long)j + (long)threadIdx.x * n];
    vrow = a.A[(long)j + (long)(a.j0 + threadIdx.x) * a.lda];
  }
  T xa = make_<T>(0.0, 0.0), wt = make_<T>(0.0, 0.0), vprev = make_<T>(0.0, 0.0), vv[PPT], ww[PPT];
#pragma unroll
  for (int k = 0; k < PPT; ++k) vv[k] = ww[k] = make_<T>(0.0, 0.0);
  if (live) {
    xa = a.A[r + (long)j * a.lda];
    if (i > 0) {
      wt = a.wtmp[r];
      vprev = a.A[r + (long)(j - 1) * a.lda];
    }
#pragma unroll
    for (int k = 0; k < PPT; ++k) {
      const int p = q + TPR * k;
      if (p < i) {
        vv[k] = a.A[r + (long)(a.j0 + p) * a.lda];
        if (p < i - 1) ww[k] = a.W[r + (long)p * n];
      }
    }
  }
  // ---- alpha2 = -tau/2 (w'^H v) of the previous column; row j of W (final) and V, conjugated
  T alpha2 = make_<T>(0.0, 0.0);
  if (i > 0) {
    if (threadIdx.x < 64) {
      sp = wave_sum(sp);
      if (threadIdx.x == 0) slot = sp;
    }
    __syncthreads();
    alpha2 = (-0.5) * (tau_prev * slot);
  }
  if (threadIdx.x < i) {
    wj[threadIdx.x] = conj_(threadIdx.x == i - 1 ? wrow + alpha2 * vrow : wrow);
    vj[threadIdx.x] = conj_(vrow);
  }
  __syncthreads();
  // ---- x = A[:, j] - V W[j, :]^H - W V[j, :]^H
  double nrm = 0.0;
  T corr = make_<T>(0.0, 0.0);
  if (live) {
    const T wlast = wt + alpha2 * vprev;  // final W[r][i - 1]
    if (i > 0 && q == (i - 1) % TPR) a.W[r + (long)(i - 1) * n] = wlast;
#pragma unroll
    for (int k = 0; k < PPT; ++k) {
      const int p = q + TPR * k;
      if (p < i) corr = corr + vv[k] * wj[p] + (p == i - 1 ? wlast : ww[k]) * vj[p];
    }
  }
  corr = row_sum(corr);
  if (r < n && q == 0) {
    if (!live) {
      a.xvec[r] = make_<T>(0.0, 0.0);
    } else {
      T x = xa - corr;
      if (r == j) {
        x = make_<T>(real_(x), 0.0);
        a.d[j] = real_(x);
      }
      a.A[r + (long)j * a.lda] = x;
      a.xvec[r] = r > j ? x : make_<T>(0.0, 0.0);
      if (r >= j + 2) nrm = abs2_(x);
    }
  }
  if (q == 0) red[rl] = nrm;
  __syncthreads();
  if (threadIdx.x < 64) {
    const double t = wave_sum(threadIdx.x < ROWT ? red[threadIdx.x] : 0.0);
    if (threadIdx.x == 0) a.pnorm[blockIdx.x] = t;
  }
}

// ---- k2: y = A22 x on lower-triangle tiles, and the short products W^H x, V^H x -------------------------------
// Tile blocks: 64 x 64; lane = row, wave = 16 columns whose loads are all in flight at once.  Tile (R, C) adds
// sum_c A[r][c] x[c]  to ylow[C][r]  (one slot per column strip) and  sum_r conj(A[r][c]) x[r]  (r > c) to yup[R][c]
// (one slot per row strip); k_trd_finish sums the slots.  Enumeration: R >= C over the active strips, row by row.
// Dot blocks (after the tile blocks): RD = 64 rows each; wave w takes the panel columns p = w, w + 4, ... of W and V.
constexpr int RT = 64;   // rows per matrix-vector tile (= TS)
constexpr int RD = 64;   // rows per dot block
constexpr int GW = 8;         // columns per butterfly group (8 keeps the kernel at ~100 registers, no scratch)

// Column sums of v[0..8) over the 64 lanes by halving: after the step with mask m a lane keeps half of its values
// (which half: its bit m) summed with its partner's - 7 exchanges, then 3 plain steps, instead of 8 x 6.
// Result for column 4 b5 + 2 b4 + b3 (bits of the lane index) in v[0] of the lanes with (lane & 7) == 0.
// The exchanges stay on the vector ALU (round 5; a __shfl_xor is two ds_bpermute round trips per double): v_permlane32_swap / v_permlane16_swap
// (gfx950) across the rows of 16 lanes, DPP inside a row.  xchg<M>(v): the value of the lane's partner - lane ^ M for M = 32, 16, 8, 2, 1; for
// M = 4 the partner is lane ^ 7 (row_half_mirror), which serves a plain sum over groups of 8 just as well.
template <int M>
__device__ __forceinline__ unsigned xchg32(unsigned x, int lane) {
  if constexpr (M == 32) {
    const auto r = __builtin_amdgcn_permlane32_swap(x, x, false, false);  // r[0]: the lower half's values in both halves, r[1]: the upper's
    return (lane & 32) ? r[0] : r[1];
  } else if constexpr (M == 16) {
    const auto r = __builtin_amdgcn_permlane16_swap(x, x, false, false);  // r[0]: the even rows' values in both rows of a pair, r[1]: the odd rows'
    return (lane & 16) ? r[0] : r[1];
  } else {
    constexpr int CTRL = M == 8 ? 0x128 /* row_ror:8 */ : (M == 4 ? 0x141 /* row_half_mirror */ : (M == 2 ? 0x4E : 0xB1));
    return (unsigned)__builtin_amdgcn_update_dpp(0, (int)x, CTRL, 0xf, 0xf, false);
  }
}
template <int M>
__device__ __forceinline__ double xchg(double v, int lane) {
  return __hiloint2double((int)xchg32<M>((unsigned)__double2hiint(v), lane), (int)xchg32<M>((unsigned)__double2loint(v), lane));
}
template <int M>
__device__ __forceinline__ Z xchg(Z v, int lane) {
  return {xchg<M>(v.re, lane), xchg<M>(v.im, lane)};
}
template <int H, int M, class T>
__device__ __forceinline__ void butterfly_level(T (&v)[GW], int lane) {
  const bool hi = lane & M;
#pragma unroll
  for (int k = 0; k < H; ++k) {
    const T send = sel_(hi, v[k], v[k + H]);
    const T keep = sel_(hi, v[k + H], v[k]);
    v[k] = keep + xchg<M>(send, lane);
  }
}
template <class T>
__device__ __forceinline__ void butterfly8(T (&v)[GW], int lane) {
  butterfly_level<4, 32>(v, lane);
  butterfly_level<2, 16>(v, lane);
  butterfly_level<1, 8>(v, lane);
  v[0] = v[0] + xchg<4>(v[0], lane);
  v[0] = v[0] + xchg<2>(v[0], lane);
  v[0] = v[0] + xchg<1>(v[0], lane);
}
__device__ __forceinline__ int butterfly_col(int lane) { return ((lane >> 5) & 1) * 4 + ((lane >> 4) & 1) * 2 + ((lane >> 3) & 1); }

template <class T>
__global__ void __launch_bounds__(256, 2) k_trd_hemv(Args<T> a, int S0, int ntiles) {
  __shared__ T sh[4][TS];
  const int n = a.n, j = a.j, i = a.j - a.j0;
  const int lane = threadIdx.x & 63;
  const int w = __builtin_amdgcn_readfirstlane(threadIdx.x >> 6);  // wave index as a scalar: x[c] below becomes scalar loads
  if ((int)blockIdx.x < ntiles) {
    // tile (R, C) of RT x TS = 64 x 64; lane = row, wave w = columns 16 w .. 16 w + 15 (all 16 loads in flight at once)
    // Odd columns walk the tiles backwards: a trailing matrix larger than the 256 MB Infinity Cache (real n = 10^4: 400 MB of
    // lower triangle) then starts each column on the tiles the previous column touched last, which are still resident.
    int t = (a.boustrophedon && (a.j & 1)) ? ntiles - 1 - (int)blockIdx.x : (int)blockIdx.x, Rr = (int)((sqrt(8.0 * t + 1.0) - 1.0) * 0.5);
    while ((Rr + 1) * (Rr + 2) / 2 <= t) ++Rr;
    while (Rr * (Rr + 1) / 2 > t) --Rr;
    const int R = S0 + Rr, C = S0 + (t - Rr * (Rr + 1) / 2);
    const long r = (long)R * RT + lane;
    const T xr = r < n ? a.xvec[r] : make_<T>(0.0, 0.0);
    T av[2 * GW];
#pragma unroll
    for (int cc = 0; cc < 2 * GW; ++cc) {
      const long c = (long)C * TS + 2 * GW * w + cc;
      av[cc] = make_<T>(0.0, 0.0);
      if (r < n && c < n && r >= c) av[cc] = a.A[r + c * a.lda];
    }
    // x of the wave's 16 columns: lane l < 16 loads x[c0 + l], every lane then reads it as a scalar broadcast
    const long cl = (long)C * TS + 2 * GW * w + (lane & 15);
    const T xcol = cl < n ? a.xvec[cl] : make_<T>(0.0, 0.0);
    T low = make_<T>(0.0, 0.0);
#pragma unroll
    for (int g = 0; g < 2; ++g) {
      T up[GW];  // this row's contribution to the columns of the group: conj(A[r][c]) x[r]
#pragma unroll
      for (int cc = 0; cc < GW; ++cc) {
        const long c = (long)C * TS + 2 * GW * w + GW * g + cc;
        T v = av[GW * g + cc];
        if (r == c) v = make_<T>(real_(v), 0.0);
        const T xc = lane_bcast(xcol, GW * g + cc);
        low = low + v * xc;
        up[cc] = sel_(r > c, conj_(v) * xr, make_<T>(0.0, 0.0));
      }
      butterfly8(up, lane);
      if ((lane & 7) == 0) {
        const long c = (long)C * TS + 2 * GW * w + GW * g + butterfly_col(lane);
        if (c < n) a.yup[(long)R * n + c] = up[0];
      }
    }
    sh[w][lane] = low;
    __syncthreads();
    if (w == 0 && r < n) a.ylow[(long)C * n + r] = ((sh[0][lane] + sh[1][lane]) + sh[2][lane]) + sh[3][lane];
  } else {
    // dot block: a.dotgroups groups of RD = 64 rows (lane = row; one group when the columns are latency bound, four for
    // large n so that k_trd_finish has few partials to add); wave w takes the panel columns p = w, w + 4, ... of W and V
    const int b = blockIdx.x - ntiles;
    T mw[GW], mv[GW];
#pragma unroll
    for (int k = 0; k < GW; ++k) mw[k] = mv[k] = make_<T>(0.0, 0.0);
    for (int gq = 0; gq < a.dotgroups; ++gq) {
      const long r = (long)j + 1 + ((long)b * a.dotgroups + gq) * RD + lane;
      const bool live = r < n;
      const T xr = live ? a.xvec[r] : make_<T>(0.0, 0.0);
      T lw[GW], lv[GW];
#pragma unroll
      for (int k = 0; k < GW; ++k) {
        const int p = w + 4 * k;
        lw[k] = lv[k] = make_<T>(0.0, 0.0);
        if (live && p < i) {
          lw[k] = a.W[r + (long)p * n];
          lv[k] = a.A[r + (long)(a.j0 + p) * a.lda];
        }
      }
#pragma unroll
      for (int k = 0; k < GW; ++k) {
        mw[k] = mw[k] + conj_(lw[k]) * xr;
        mv[k] = mv[k] + conj_(lv[k]) * xr;
      }
    }
    butterfly8(mw, lane);
    butterfly8(mv, lane);
    if ((lane & 7) == 0) {
      const int p = w + 4 * butterfly_col(lane);
      a.zpart[(long)b * 2 * NB + p] = sel_(p < i, mw[0], make_<T>(0.0, 0.0));
      a.zpart[(long)b * 2 * NB + NB + p] = sel_(p < i, mv[0], make_<T>(0.0, 0.0));
    }
  }
}

// ---- k3: reflector, w', partials of w'^H v -------------------------------------------------------------------
// Same layout as k_trd_column; all loads that do not depend on the reduced scalars come first.
template <class T>
__global__ void __launch_bounds__(ROWT * TPR) k_trd_finish(Args<T> a, int S0, int NS) {
  __shared__ T zsh[4][2 * NB], zw[NB], zv[NB], red[ROWT];
  __shared__ double dslot;
  constexpr int PPT = NB / TPR;
  const int n = a.n, j = a.j, i = a.j - a.j0;
  const int q = threadIdx.x % TPR, rl = threadIdx.x / TPR;
  const long r = (long)blockIdx.x * ROWT + rl;
  const bool live = r < n && r >= j + 1;
  // ---- loads
  double pn = 0.0;
  if (threadIdx.x < 64)
    for (int b = threadIdx.x; b < a.nrowblocks; b += 64) pn += a.pnorm[b];
  const T alpha = a.xvec[j + 1];
  T zp = make_<T>(0.0, 0.0);  // thread (slot = t % 64, part = t / 64) sums the dot partials b = part, part + 4, ...
  {
    const int slot = threadIdx.x % (2 * NB), part = threadIdx.x / (2 * NB);
    if (slot % NB < i) {  // (four partials requested together: one load per loop trip is one round trip per trip)
      int b = part;
      for (; b + 12 < a.ndot; b += 16) {
        const T z0 = a.zpart[(long)b * 2 * NB + slot], z1 = a.zpart[(long)(b + 4) * 2 * NB + slot];
        const T z2 = a.zpart[(long)(b + 8) * 2 * NB + slot], z3 = a.zpart[(long)(b + 12) * 2 * NB + slot];
        zp = (((zp + z0) + z1) + z2) + z3;
      }
      for (; b < a.ndot; b += 4) zp = zp + a.zpart[(long)b * 2 * NB + slot];
    }
  }
  T zrow = make_<T>(0.0, 0.0);  // row j + 1 of W (slots < NB) and of V
  if (threadIdx.x < 2 * NB && threadIdx.x % NB < i) {
    const int p = threadIdx.x % NB;
    zrow = threadIdx.x >= NB ? a.A[(long)(j + 1) + (long)(a.j0 + p) * a.lda] : a.W[(long)(j + 1) + (long)p * n];
  }
  T y = make_<T>(0.0, 0.0), xr = make_<T>(0.0, 0.0), t0 = make_<T>(0.0, 0.0), vv[PPT], ww[PPT];
#pragma unroll
  for (int k = 0; k < PPT; ++k) vv[k] = ww[k] = make_<T>(0.0, 0.0);
  if (live) {
    // strip partials of the matrix-vector product, four independent chains so that the loads overlap
    T y1 = make_<T>(0.0, 0.0), y2 = y1, y3 = y1;
    const int Clast = (int)(r / TS);
    for (int C = S0 + q; C <= Clast; C += 4 * TPR) {
      y = y + a.ylow[(long)C * n + r];
      if (C + TPR <= Clast) y1 = y1 + a.ylow[(long)(C + TPR) * n + r];
      if (C + 2 * TPR <= Clast) y2 = y2 + a.ylow[(long)(C + 2 * TPR) * n + r];
      if (C + 3 * TPR <= Clast) y3 = y3 + a.ylow[(long)(C + 3 * TPR) * n + r];
    }
    for (int R = (int)(r / RT) + q; R < NS; R += 4 * TPR) {  // NS: number of RT-row strips
      y = y + a.yup[(long)R * n + r];
      if (R + TPR < NS) y1 = y1 + a.yup[(long)(R + TPR) * n + r];
      if (R + 2 * TPR < NS) y2 = y2 + a.yup[(long)(R + 2 * TPR) * n + r];
      if (R + 3 * TPR < NS) y3 = y3 + a.yup[(long)(R + 3 * TPR) * n + r];
    }
    y = (y + y1) + (y2 + y3);
    xr = a.xvec[r];
    t0 = a.A[r + (long)(j + 1) * a.lda];
#pragma unroll
    for (int k = 0; k < PPT; ++k) {
      const int p = q + TPR * k;
      if (p < i) {
        vv[k] = a.A[r + (long)(a.j0 + p) * a.lda];
        ww[k] = a.W[r + (long)p * n];
      }
    }
  }
  // ---- reductions of the scalars
  if (threadIdx.x < 64) {
    pn = wave_sum(pn);
    if (threadIdx.x == 0) dslot = pn;
  }
  zsh[threadIdx.x / (2 * NB)][threadIdx.x % (2 * NB)] = zp;
  __syncthreads();
  const Larfg<T> h = larfg<T>(alpha, dslot);
  const T beta = make_<T>(h.beta, 0.0);
  if (threadIdx.x < 2 * NB) {  // W^H v and V^H v from the products with x
    const int p = threadIdx.x % NB;
    T z = make_<T>(0.0, 0.0);
    if (p < i && !h.identity)
      z = ((((zsh[0][threadIdx.x] + zsh[1][threadIdx.x]) + zsh[2][threadIdx.x]) + zsh[3][threadIdx.x]) - beta * conj_(zrow)) * h.scale;
    if (threadIdx.x >= NB) zv[p] = z; else zw[p] = z;
  }
  __syncthreads();
  // ---- v, w'
  T corr = make_<T>(0.0, 0.0);
#pragma unroll
  for (int k = 0; k < PPT; ++k) {
    const int p = q + TPR * k;
    if (live && p < i) corr = corr + vv[k] * zw[p] + ww[k] * zv[p];
  }
  y = row_sum(y);
  corr = row_sum(corr);
  T sv = make_<T>(0.0, 0.0);
  if (live && q == 0) {
    T v, wp = make_<T>(0.0, 0.0);
    if (h.identity) {
      v = make_<T>(r == j + 1 ? 1.0 : 0.0, 0.0);
    } else {
      v = r == j + 1 ? make_<T>(1.0, 0.0) : xr * h.scale;
      if (r == j + 1) t0 = make_<T>(real_(t0), 0.0);
      wp = h.tau * ((y - beta * t0) * h.scale - corr);
    }
    a.wtmp[r] = wp;
    a.A[r + (long)j * a.lda] = v;
    sv = conj_(wp) * v;
  }
  if (r == 0 && q == 0) {
    a.e[j] = h.beta;
    a.tau[j] = h.tau;
  }
  if (q == 0) red[rl] = sv;
  __syncthreads();
  if (threadIdx.x < 64) {
    const T t = wave_sum(sel_(threadIdx.x < ROWT, red[threadIdx.x % ROWT], make_<T>(0.0, 0.0)));
    if (threadIdx.x == 0) a.spart[blockIdx.x] = t;
  }
}

// ================================================================================================================
// Two kernels per column.  The column update of column j + 1 is folded into the finish of column j, and what it lacks -
// the global scalar s = w'^H v of column j - is applied by the matrix-vector kernel of column j + 1 on the fly:
//   zlatrd's update of column j + 1 subtracts, for the newest reflector, v conj(w[j+1]) + w conj(v[j+1]) with
//   w = w' + alpha2 v, alpha2 = -tau s / 2, v[j+1] = 1, i.e.  w' + v (conj(w'[j+1]) + 2 Re alpha2).  So
//       x_{j+1} = base + mu v ,   base = A[:, j+1] - sum_{p < i} (...) - w' ,   mu = Re(tau s) - conj(w'[j+1]) ,
//   base needs only quantities local to a row (k_trd_finish2 has them in registers already), mu two scalars.
//   k_trd_hemv2(j)  : x = bvec + mu A[:, j-1] formed on the fly (first column of a panel: x = A[:, j]); tiles as in
//                     k_trd_hemv; the dot blocks start at row j: they store x, d[j], the |x|^2 partials, finish column
//                     i - 1 of W (W = w' + alpha2 v) and form W^H x, V^H x.
//   k_trd_finish2(j): k_trd_finish + bvec for column j + 1.
// ================================================================================================================
template <class T>
struct PrevScalars {
  T mu, alpha2;
};
// mu and alpha2 of the previous column (i > 0), every thread of the block gets them; slot: one T of shared memory
template <class T>
__device__ __forceinline__ PrevScalars<T> prev_scalars(const Args<T>& a, T* slot) {
  const T tauv = a.tau[a.j - 1], wj = a.wtmp_prev[a.j];  // (requested before the reduction's barrier, not after it)
  const T s = sum_partials(a.spart_prev, a.nrowblocks, slot);
  const T ts = tauv * s;
  PrevScalars<T> ps;
  ps.alpha2 = (-0.5) * ts;
  ps.mu = make_<T>(real_(ts), 0.0) - conj_(wj);
  return ps;
}

template <class T>
__device__ __forceinline__ void trd_hemv2_body(const Args<T>& a, int S0, int ntiles, int bid) {
  __shared__ T sh[4][TS];
  __shared__ T slot;
  const int n = a.n, j = a.j, i = a.j - a.j0;
  const int lane = threadIdx.x & 63;
  const int w = __builtin_amdgcn_readfirstlane(threadIdx.x >> 6);
  // x[q] of column j for q > j (0 above) is base + mu v_{j-1} (first column of a panel: A[q][j] itself).  Its two parts are REQUESTED before
  // mu is known - and so are the block's matrix entries: mu comes out of a reduction over the previous column's partials, a barrier and two
  // more loads, three dependent round trips that used to stand in front of every block's own loads (round 5).
  struct XRaw {
    T base, v;
  };
  auto xraw = [&](long q) -> XRaw {
    XRaw x{make_<T>(0.0, 0.0), make_<T>(0.0, 0.0)};
    if (q <= j || q >= n) return x;
    if (i > 0) {
      x.base = a.bvec[q];
      x.v = a.A[q + (long)(j - 1) * a.lda];
    } else {
      x.base = a.A[q + (long)j * a.lda];
    }
    return x;
  };
  const bool tile = bid < ntiles;
  int R = 0, C = 0;
  long r = 0;
  T av[2 * GW];
  XRaw xr_raw{make_<T>(0.0, 0.0), make_<T>(0.0, 0.0)}, xc_raw = xr_raw;
  // dot block: rows j + 64 b + lane (row j itself included: it carries d[j])
  const int b = bid - ntiles;
  bool inrange = false;
  T xbase = make_<T>(0.0, 0.0), wt = make_<T>(0.0, 0.0), vprev = make_<T>(0.0, 0.0);
  T mw[GW], mv[GW];  // (dot block) this wave's panel columns of W and V at row r
  if (tile) {
    // Odd columns walk the tiles backwards: a trailing matrix larger than the 256 MB Infinity Cache (real n = 10^4: 400 MB of
    // lower triangle) then starts each column on the tiles the previous column touched last, which are still resident.
    int t = (a.boustrophedon && (a.j & 1)) ? ntiles - 1 - bid : bid, Rr = (int)((sqrt(8.0 * t + 1.0) - 1.0) * 0.5);
    while ((Rr + 1) * (Rr + 2) / 2 <= t) ++Rr;
    while (Rr * (Rr + 1) / 2 > t) --Rr;
    R = S0 + Rr;
    C = S0 + (t - Rr * (Rr + 1) / 2);
    r = (long)R * RT + lane;
#pragma unroll
    for (int cc = 0; cc < 2 * GW; ++cc) {
      const long c = (long)C * TS + 2 * GW * w + cc;
      av[cc] = make_<T>(0.0, 0.0);
      if (r < n && c < n && r >= c) av[cc] = a.A[r + c * a.lda];
    }
    xr_raw = xraw(r);
    xc_raw = xraw((long)C * TS + 2 * GW * w + (lane & 15));
  } else {
    r = (long)j + (long)b * RD + lane;
    inrange = r < n;
    if (inrange) {
      if (i > 0) {
        vprev = a.A[r + (long)(j - 1) * a.lda];
        wt = a.wtmp_prev[r];
        xbase = a.bvec[r];
      } else {
        xbase = a.A[r + (long)j * a.lda];
      }
    }
#pragma unroll
    for (int k = 0; k < GW; ++k) {
      const int p = w + 4 * k;
      mw[k] = mv[k] = make_<T>(0.0, 0.0);
      if (inrange && r > j && p < i) {
        if (p != i - 1) mw[k] = a.W[r + (long)p * n];  // (column i - 1 of W is finished below, from wtmp and alpha2)
        mv[k] = a.A[r + (long)(a.j0 + p) * a.lda];
      }
    }
  }
  PrevScalars<T> ps;
  ps.mu = ps.alpha2 = make_<T>(0.0, 0.0);
  if (i > 0) ps = prev_scalars(a, &slot);
  if (tile) {
    // tile (R, C) of RT x TS = 64 x 64; lane = row, wave w = columns 16 w .. 16 w + 15 (all 16 loads in flight at once)
    const T xr = xr_raw.base + ps.mu * xr_raw.v;
    const T xcol = xc_raw.base + ps.mu * xc_raw.v;
    T low = make_<T>(0.0, 0.0);
#pragma unroll
    for (int g = 0; g < 2; ++g) {
      T up[GW];
#pragma unroll
      for (int cc = 0; cc < GW; ++cc) {
        const long c = (long)C * TS + 2 * GW * w + GW * g + cc;
        T v = av[GW * g + cc];
        if (r == c) v = make_<T>(real_(v), 0.0);
        const T xc = lane_bcast(xcol, GW * g + cc);
        low = low + v * xc;
        up[cc] = sel_(r > c, conj_(v) * xr, make_<T>(0.0, 0.0));
      }
      butterfly8(up, lane);
      if ((lane & 7) == 0) {
        const long c = (long)C * TS + 2 * GW * w + GW * g + butterfly_col(lane);
        if (c < n) a.yup[(long)R * n + c] = up[0];
      }
    }
    sh[w][lane] = low;
    __syncthreads();
    if (w == 0 && r < n) a.ylow[(long)C * n + r] = ((sh[0][lane] + sh[1][lane]) + sh[2][lane]) + sh[3][lane];
  } else {
    // dot block b; wave w: panel columns p = w, w + 4, ...
    const T xfull = inrange ? (i > 0 ? xbase + ps.mu * vprev : xbase) : make_<T>(0.0, 0.0);
    const bool live = inrange && r > j;
    const T xr = live ? xfull : make_<T>(0.0, 0.0);
    const T wlast = wt + ps.alpha2 * vprev;  // final W[r][i - 1]
#pragma unroll
    for (int k = 0; k < GW; ++k)
      if (live && w + 4 * k == i - 1) mw[k] = wlast;
    if (w == 0 && inrange) {
      if (r == j) {
        const double dj = real_(xfull);
        a.d[j] = dj;
        a.A[r + (long)j * a.lda] = make_<T>(dj, 0.0);
      } else {
        a.xvec[r] = xfull;
      }
      if (i > 0) a.W[r + (long)(i - 1) * n] = wlast;
    }
    if (w == 0) {
      const double nrm = wave_sum(inrange && r >= j + 2 ? abs2_(xfull) : 0.0);
      if (lane == 0) a.pnorm[b] = nrm;
    }
#pragma unroll
    for (int k = 0; k < GW; ++k) {
      mw[k] = conj_(mw[k]) * xr;
      mv[k] = conj_(mv[k]) * xr;
    }
    butterfly8(mw, lane);
    butterfly8(mv, lane);
    if ((lane & 7) == 0) {
      const int p = w + 4 * butterfly_col(lane);
      a.zpart[(long)b * 2 * NB + p] = sel_(p < i, mw[0], make_<T>(0.0, 0.0));
      a.zpart[(long)b * 2 * NB + NB + p] = sel_(p < i, mv[0], make_<T>(0.0, 0.0));
    }
  }
}
template <class T>
__global__ void __launch_bounds__(256, 2) k_trd_hemv2(Args<T> a, int S0, int ntiles) {
  trd_hemv2_body<T>(a, S0, ntiles, (int)blockIdx.x);
}

template <class T>
__device__ __forceinline__ void trd_finish2_body(const Args<T>& a, int S0, int NS, int bid) {
  __shared__ T zsh[4][2 * NB], zw[NB], zv[NB], wj1[NB], vj1[NB], red[ROWT];
  __shared__ double dslot;
  constexpr int PPT = NB / TPR;
  const int n = a.n, j = a.j, i = a.j - a.j0;
  const int q = threadIdx.x % TPR, rl = threadIdx.x / TPR;
  const long r = (long)bid * ROWT + rl;
  const bool live = r < n && r >= j + 1;
  // ---- loads
  const T alpha = a.xvec[j + 1];
  T zrow = make_<T>(0.0, 0.0);  // row j + 1 of W (slots < NB) and of V
  if (threadIdx.x < 2 * NB && threadIdx.x % NB < i) {
    const int p = threadIdx.x % NB;
    zrow = threadIdx.x >= NB ? a.A[(long)(j + 1) + (long)(a.j0 + p) * a.lda] : a.W[(long)(j + 1) + (long)p * n];
  }
  T y = make_<T>(0.0, 0.0), xr = make_<T>(0.0, 0.0), t0 = make_<T>(0.0, 0.0), vv[PPT], ww[PPT];
#pragma unroll
  for (int k = 0; k < PPT; ++k) vv[k] = ww[k] = make_<T>(0.0, 0.0);
  if (live) {
    xr = a.xvec[r];
    t0 = a.A[r + (long)(j + 1) * a.lda];
#pragma unroll
    for (int k = 0; k < PPT; ++k) {
      const int p = q + TPR * k;
      if (p < i) {
        vv[k] = a.A[r + (long)(a.j0 + p) * a.lda];
        ww[k] = a.W[r + (long)p * n];
      }
    }
    T y1 = make_<T>(0.0, 0.0), y2 = y1, y3 = y1;
    const int Clast = (int)(r / TS);
    for (int C = S0 + q; C <= Clast; C += 4 * TPR) {
      y = y + a.ylow[(long)C * n + r];
      if (C + TPR <= Clast) y1 = y1 + a.ylow[(long)(C + TPR) * n + r];
      if (C + 2 * TPR <= Clast) y2 = y2 + a.ylow[(long)(C + 2 * TPR) * n + r];
      if (C + 3 * TPR <= Clast) y3 = y3 + a.ylow[(long)(C + 3 * TPR) * n + r];
    }
    for (int R = (int)(r / RT) + q; R < NS; R += 4 * TPR) {
      y = y + a.yup[(long)R * n + r];
      if (R + TPR < NS) y1 = y1 + a.yup[(long)(R + TPR) * n + r];
      if (R + 2 * TPR < NS) y2 = y2 + a.yup[(long)(R + 2 * TPR) * n + r];
      if (R + 3 * TPR < NS) y3 = y3 + a.yup[(long)(R + 3 * TPR) * n + r];
    }
    y = (y + y1) + (y2 + y3);
  }
  // (the loops below wait for their loads trip by trip: they come LAST, with everything else already in flight)
  double pn = 0.0;
  if (threadIdx.x < 64)
    for (int b = threadIdx.x; b < a.ndot; b += 64) pn += a.pnorm[b];  // |x[j+2:]|^2 partials of the dot blocks
  T zp = make_<T>(0.0, 0.0);
  {
    const int slot = threadIdx.x % (2 * NB), part = threadIdx.x / (2 * NB);
    if (slot % NB < i) {  // (eight partials requested together: one load per loop trip is one round trip per trip)
      for (int b0 = part; b0 < a.ndot; b0 += 32) {
        T z[8];
#pragma unroll
        for (int u = 0; u < 8; ++u) z[u] = b0 + 4 * u < a.ndot ? a.zpart[(long)(b0 + 4 * u) * 2 * NB + slot] : make_<T>(0.0, 0.0);
#pragma unroll
        for (int u = 0; u < 8; ++u) zp = zp + z[u];
      }
    }
  }
  // ---- reductions of the scalars
  if (threadIdx.x < 64) {
    pn = wave_sum(pn);
    if (threadIdx.x == 0) dslot = pn;
  }
  zsh[threadIdx.x / (2 * NB)][threadIdx.x % (2 * NB)] = zp;
  __syncthreads();
  const Larfg<T> h = larfg<T>(alpha, dslot);
  const T beta = make_<T>(h.beta, 0.0);
  if (threadIdx.x < 2 * NB) {
    const int p = threadIdx.x % NB;
    T z = make_<T>(0.0, 0.0);
    if (p < i && !h.identity)
      z = ((((zsh[0][threadIdx.x] + zsh[1][threadIdx.x]) + zsh[2][threadIdx.x]) + zsh[3][threadIdx.x]) - beta * conj_(zrow)) * h.scale;
    if (threadIdx.x >= NB) {
      zv[p] = z;
      vj1[p] = conj_(zrow);
    } else {
      zw[p] = z;
      wj1[p] = conj_(zrow);
    }
  }
  __syncthreads();
  // ---- v, w', and the local part of column j + 1
  T corr = make_<T>(0.0, 0.0), upd = make_<T>(0.0, 0.0);
#pragma unroll
  for (int k = 0; k < PPT; ++k) {
    const int p = q + TPR * k;
    if (live && p < i) {
      corr = corr + vv[k] * zw[p] + ww[k] * zv[p];
      upd = upd + vv[k] * wj1[p] + ww[k] * vj1[p];
    }
  }
  y = row_sum(y);
  corr = row_sum(corr);
  upd = row_sum(upd);
  T sv = make_<T>(0.0, 0.0);
  if (live && q == 0) {
    T v, wp = make_<T>(0.0, 0.0);
    T t0d = t0;
    if (r == j + 1) t0d = make_<T>(real_(t0), 0.0);
    if (h.identity) {
      v = make_<T>(r == j + 1 ? 1.0 : 0.0, 0.0);
    } else {
      v = r == j + 1 ? make_<T>(1.0, 0.0) : xr * h.scale;
      wp = h.tau * ((y - beta * t0d) * h.scale - corr);
    }
    a.wtmp[r] = wp;
    a.A[r + (long)j * a.lda] = v;
    if (a.make_base) a.bvec[r] = t0d - upd - wp;
    sv = conj_(wp) * v;
  }
  if (r == 0 && q == 0) {
    a.e[j] = h.beta;
    a.tau[j] = h.tau;
  }
  if (q == 0) red[rl] = sv;
  __syncthreads();
  if (threadIdx.x < 64) {
    const T t = wave_sum(sel_(threadIdx.x < ROWT, red[threadIdx.x % ROWT], make_<T>(0.0, 0.0)));
    if (threadIdx.x == 0) a.spart[bid] = t;
  }
}
template <class T>
__global__ void __launch_bounds__(ROWT * TPR) k_trd_finish2(Args<T> a, int S0, int NS) {
  trd_finish2_body<T>(a, S0, NS, (int)blockIdx.x);
}

// End of a panel (last column jl): finish column jl - j0 of W for the rows the rank-2nb update reads.
template <class T>
__device__ __forceinline__ void trd_panel_end_body(const Args<T>& a, int jl, int bid) {
  __shared__ T slot;
  const long r = (long)bid * blockDim.x + threadIdx.x;
  const T alpha2 = (-0.5) * (a.tau[jl] * sum_partials(a.spart, a.nrowblocks, &slot));
  if (r >= a.n || r <= jl) return;
  a.W[r + (long)(jl - a.j0) * a.n] = a.wtmp[r] + alpha2 * a.A[r + (long)jl * a.lda];
}
template <class T>
__global__ void k_trd_panel_end(Args<T> a, int jl) {
  trd_panel_end_body<T>(a, jl, (int)blockIdx.x);
}
// Trailing update of a panel:  C -= V2 W2^H + W2 V2^H  on the lower triangle of C = A[jend:, jend:], with
// V2 = A[jend:, j0 : j0 + nb] and W2 = W[jend:, 0 : nb]  (zher2k / dsyr2k of zhetrd).  rocBLAS runs this rank-2nb
// update as ~21 small GEMMs per panel (25 ms per EVD at n = 4097 for 2.5 ms of memory traffic); here one launch per
// panel: a workgroup owns a 64 x 64 tile of C (tiles R >= C), stages the four 64 x 16 operand panels in LDS and
// each thread accumulates a 4 x 4 block in registers.
constexpr int UT = 64, UK = 16;
template <class T>
__device__ __forceinline__ void trd_rank2k_body(T* A, long lda, const T* W, long ldw, int n, int j0, int nb, int jend, int bid) {
  __shared__ T Vr[UK][UT], Wr[UK][UT], Vc[UK][UT], Wc[UK][UT];
  int t = bid, R = (int)((sqrt(8.0 * t + 1.0) - 1.0) * 0.5);
  while ((R + 1) * (R + 2) / 2 <= t) ++R;
  while (R * (R + 1) / 2 > t) --R;
  const int C = t - R * (R + 1) / 2;
  const int tx = threadIdx.x % 16, ty = threadIdx.x / 16;
  const long r0 = (long)jend + (long)R * UT, c0 = (long)jend + (long)C * UT;
  // A workgroup lives for a handful of memory round trips: the tile's own entries - needed last - are requested first, and the operand panels
  // of slice k0 + UK while slice k0 is multiplied (round 5: every slice and the read-modify-write used to wait for their own loads in turn).
  constexpr bool EARLY = sizeof(T) == 8;  // (complex: the 64 extra registers would halve the occupancy - its entries are requested after the loop)
  T old[4][4];
  auto fetch_old = [&]() {
#pragma unroll
    for (int y = 0; y < 4; ++y)
#pragma unroll
      for (int x = 0; x < 4; ++x) {
        const long r = r0 + tx + 16 * x, c = c0 + ty + 16 * y;
        old[y][x] = (r < n && c < n && r >= c) ? A[r + c * lda] : make_<T>(0.0, 0.0);
      }
  };
  if constexpr (EARLY) fetch_old();
  constexpr int NIT = UK * UT / 256;
  T sv[NIT][4];  // staged operands of one slice: V / W at the tile's rows, conj V / conj W at its columns
  auto fetch = [&](int k0) {
#pragma unroll
    for (int it = 0; it < NIT; ++it) {
      const int row = threadIdx.x % UT, k = threadIdx.x / UT + (256 / UT) * it;
      const bool kin = k0 + k < nb;
      const long rr = r0 + row, cc = c0 + row;
      sv[it][0] = kin && rr < n ? A[rr + (long)(j0 + k0 + k) * lda] : make_<T>(0.0, 0.0);
      sv[it][1] = kin && rr < n ? W[rr + (long)(k0 + k) * ldw] : make_<T>(0.0, 0.0);
      sv[it][2] = kin && cc < n ? conj_(A[cc + (long)(j0 + k0 + k) * lda]) : make_<T>(0.0, 0.0);
      sv[it][3] = kin && cc < n ? conj_(W[cc + (long)(k0 + k) * ldw]) : make_<T>(0.0, 0.0);
    }
  };
  T acc[4][4];
#pragma unroll
  for (int x = 0; x < 4; ++x)
#pragma unroll
    for (int y = 0; y < 4; ++y) acc[x][y] = make_<T>(0.0, 0.0);
  fetch(0);
  for (int k0 = 0; k0 < nb; k0 += UK) {
    __syncthreads();
#pragma unroll
    for (int it = 0; it < NIT; ++it) {
      const int row = threadIdx.x % UT, k = threadIdx.x / UT + (256 / UT) * it;
      Vr[k][row] = sv[it][0];
      Wr[k][row] = sv[it][1];
      Vc[k][row] = sv[it][2];
      Wc[k][row] = sv[it][3];
    }
    if (k0 + UK < nb) fetch(k0 + UK);  // (uniform)
    __syncthreads();
#pragma unroll 4
    for (int k = 0; k < UK; ++k) {
      T vr[4], wr[4], vc[4], wc[4];
#pragma unroll
      for (int x = 0; x < 4; ++x) {
        vr[x] = Vr[k][tx + 16 * x];
        wr[x] = Wr[k][tx + 16 * x];
        vc[x] = Vc[k][ty + 16 * x];
        wc[x] = Wc[k][ty + 16 * x];
      }
#pragma unroll
      for (int x = 0; x < 4; ++x)
#pragma unroll
        for (int y = 0; y < 4; ++y) acc[x][y] = acc[x][y] + vr[x] * wc[y] + wr[x] * vc[y];
    }
  }
  if constexpr (!EARLY) fetch_old();
#pragma unroll
  for (int y = 0; y < 4; ++y)
#pragma unroll
    for (int x = 0; x < 4; ++x) {
      const long r = r0 + tx + 16 * x, c = c0 + ty + 16 * y;
      if (r < n && c < n && r >= c) {
        T v = old[y][x] - acc[x][y];
        if (r == c) v = make_<T>(real_(v), 0.0);
        A[r + c * lda] = v;
      }
    }
}
template <class T>
__global__ void __launch_bounds__(256) k_trd_rank2k(T* A, long lda, const T* W, long ldw, int n, int j0, int nb, int jend) {
  trd_rank2k_body<T>(A, lda, W, ldw, n, j0, nb, jend, (int)blockIdx.x);
}

// After the trailing update: put e back on the sub-diagonal of the panel's columns (LAPACK layout).
template <class T>
__global__ void k_trd_restore_subdiag(T* A, long lda, const double* e, int j0, int cnt) {
  const int p = blockIdx.x * blockDim.x + threadIdx.x;
  if (p < cnt) A[(long)(j0 + p + 1) + (long)(j0 + p) * lda] = make_<T>(e[j0 + p], 0.0);
}

// ---- back-transformation  C <- Q C,  Q = H_0 H_1 ... H_{n-2}  (zunmtr / dormtr, left, lower, no transpose) --------------
// In blocks of KBQ reflectors: H_{j0} ... H_{j0+kb-1} = I - V T V^H with T^-1 = striu(V^H V) + diag(1 / tau), so a block is
// two large GEMMs (W = V^H C, C -= V X) around one kb x kb triangular solve (T^-1 X = W) - no sequential larft, and the
// first GEMM has 512 rows instead of rocSOLVER's 64 (65 workgroups = a quarter of the CUs at n = 4097).
constexpr int KBQ = 512;  // 256: 146.8 ms, 512: 143.9 ms, 1024: 143.7 ms per EVD at n = 4097
// Vw[q][p] (column-major m x kb): reflector j0 + p restricted to rows r0 = j0 + off .. n-1: 0 above its unit entry.  off = 1: the
// one-stage reflectors (unit entry on the first sub-diagonal); off = B: the block reflectors of the band reduction (nls_sb.h: unit
// entry on the B-th sub-diagonal, the band itself lies above it and is not read).
template <class T>
__global__ void k_trd_copy_v(const T* A, long lda, int n, int j0, int kb, T* Vw, int off) {
  const long m = n - (j0 + off);
  const long idx = blockIdx.x * (long)blockDim.x + threadIdx.x;
  if (idx >= m * kb) return;
  const long q = idx % m;
  const int p = (int)(idx / m);
  Vw[idx] = q < p ? make_<T>(0.0, 0.0) : (q == p ? make_<T>(1.0, 0.0) : A[(j0 + off + q) + (long)(j0 + p) * lda]);
}
// Vt = Vw^H (kb x m, column-major, leading dimension kb): with it both products with V^H are plain (no-transpose) GEMMs - the library's
// transposed-operand kernels run W = V^H C at 0.67 of the matrix pipe, its plain ones at 0.87 (round 5).  32 x 32 tiles through LDS: both
// sides of the transposition move whole 256-byte runs.  grid (ceil(m / 32), ceil(kb / 32)), block (32, 8).
template <class T>
__global__ void __launch_bounds__(256) k_trd_transpose_v(const T* Vw, long m, int kb, T* Vt) {
  __shared__ T tile[32][33];
  const long q0 = (long)blockIdx.x * 32;
  const int p0 = blockIdx.y * 32;
#pragma unroll
  for (int i = 0; i < 4; ++i) {
    const int p = p0 + threadIdx.y + 8 * i;
    const long q = q0 + threadIdx.x;
    tile[threadIdx.y + 8 * i][threadIdx.x] = (q < m && p < kb) ? Vw[q + (long)p * m] : make_<T>(0.0, 0.0);
  }
  __syncthreads();
#pragma unroll
  for (int i = 0; i < 4; ++i) {
    const long q = q0 + threadIdx.y + 8 * i;
    const int p = p0 + threadIdx.x;
    if (q < m && p < kb) Vt[p + q * (long)kb] = conj_(tile[threadIdx.x][threadIdx.y + 8 * i]);
  }
}
// S (kb x kb, = V^H V) -> T^-1 = striu(S) + diag(1 / tau); tau = 0 (H = I) gets a huge diagonal, i.e. a zero row of X
template <class T>
__global__ void k_trd_tinv(T* S, int kb, const T* tau, int j0) {
  const int idx = blockIdx.x * blockDim.x + threadIdx.x;
  if (idx >= kb * kb) return;
  const int r = idx % kb, c = idx / kb;
  if (r > c) {
    S[idx] = make_<T>(0.0, 0.0);
  } else if (r == c) {
    const T t = tau[j0 + r];
    S[idx] = abs2_(t) == 0.0 ? make_<T>(1e300, 0.0) : inv_(t);
  }
}

}  // namespace trd
}  // namespace nls
