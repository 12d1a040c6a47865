// Two-stage tridiagonalisation, second back-transformation:  C <- Q2 C  with Q2 = prod_s prod_k H_{s,k} the reflectors of the bulge
// chase (nls_chase.h; H_{s,k} acts on rows s + 1 + k B .. s + k B + B).
//
// Reflectors of B consecutive sweeps (a "group" S: sweeps S B .. S B + B - 1) at the same step k form a parallelogram-shaped block
// V_{S,k}: column i is the reflector of sweep S B + i, non-zero in rows i .. i + B - 1 of the 2B - 1 rows the block spans, which start at
// row (S + k) B + 1.  Their product is the compact WY transform I - V T V^H with T^-1 = striu(V^H V) + diag(1 / tau) (k_q2_tfactor, one
// workgroup per block).  Order (see tools/twostage_proto.py for the derivation and the check against the reflector-by-reflector
// product): groups descending, steps ascending inside a group; block (S - 1, k) needs the blocks (S, k' <= k).
//
// k_q2_apply: one workgroup per slab of NC columns of C, no communication between workgroups.  G groups are taken together ("pass"):
// at step u of a pass group S_hi - i applies its block k = u - i; those G blocks are two block rows apart, hence independent, and all lie in
// a window of 2G block rows of the slab that is kept in LDS and slides down by ONE block row (B rows) per step - the slab streams through
// LDS once per pass instead of once per group.  The products skip the structural zeros of V (exactly B terms per output).
#pragma once
#include "nls_chase.h"
#include "nls_gemm.h"

namespace nls {
namespace q2 {
using namespace trd;
using sb::one_;
using sb::zero_;

// number of steps (blocks) of group S: that of its first sweep
__host__ __device__ inline int q2_nblocks(int n, int B, int S) { return chase::chase_nstages(n, B, S * B); }

// Vc[i][t] (t = 0 .. B-1): column i of block (S, k) from the chase's V2; tau[i].  Non-existent reflectors: zero column, tau = 0.
template <class T, int B>
__device__ __forceinline__ void q2_load_block(const T* V2, long ldv, int n, int S, int k, T (*Vc)[B + 1], T* tau) {
  for (int idx = threadIdx.x; idx < B * B; idx += 256) {
    const int t = idx % B, i = idx / B;
    const long s = (long)S * B + i;
    const long r0 = s + 1 + (long)k * B;
    T v = zero_<T>();
    if (s <= n - 2 && r0 + t < n) {
      const T x = V2[(r0 + t) + s * ldv];
      if (t == 0) {
        tau[i] = x;
        v = one_<T>();
      } else {
        v = x;
      }
    } else if (t == 0) {
      tau[i] = zero_<T>();
    }
    Vc[i][t] = v;
  }
}

// T factor of every block: Tb[block][i + B j] (upper triangular, zeros below)
template <class T, int B>
__global__ void __launch_bounds__(256) k_q2_tfactor(const T* V2, long ldv, int n, const int* blk_off, int ngroups, T* Tb) {
  extern __shared__ __attribute__((aligned(16))) unsigned char q2_smem[];
  T(*Vc)[B + 1] = reinterpret_cast<T(*)[B + 1]>(q2_smem);
  T(*Ti)[B + 1] = reinterpret_cast<T(*)[B + 1]>(q2_smem + sizeof(T) * B * (B + 1));
  T(*Tm)[B + 1] = Vc;  // the block itself is dead once Ti is formed
  __shared__ T tau[B];
  // block index -> (S, k)
  const int b = blockIdx.x;
  int S = 0;
  while (S + 1 < ngroups && blk_off[S + 1] <= b) ++S;
  const int k = b - blk_off[S];
  q2_load_block<T, B>(V2, ldv, n, S, k, Vc, tau);
  __syncthreads();
  for (int e = threadIdx.x; e < B * B; e += 256) {
    const int i = e % B, j = e / B;
    T g = zero_<T>();
    if (i < j) {  // rows j .. i + B - 1 of the block are common to columns i and j
      for (int rho = j; rho < i + B; ++rho) g = g + conj_(Vc[i][rho - i]) * Vc[j][rho - j];
    } else if (i == j) {
      const T t = tau[i];
      g = abs2_(t) == 0.0 ? make_<T>(1e300, 0.0) : inv_(t);
    }
    Ti[i][j] = g;
  }
  __syncthreads();
  // T = Ti^-1 (upper triangular): thread j solves column j by back substitution
  if (threadIdx.x < B) {
    const int j = threadIdx.x;
    for (int i = B - 1; i >= 0; --i) {
      T s = zero_<T>();
      if (i <= j) {
        s = i == j ? one_<T>() : zero_<T>();
        for (int t = i + 1; t <= j; ++t) s = s - Ti[i][t] * Tm[t][j];
        s = inv_(Ti[i][i]) * s;
      }
      Tm[i][j] = s;
    }
  }
  __syncthreads();
  T* out = Tb + (long)b * B * B;
  for (int e = threadIdx.x; e < B * B; e += 256) out[e] = Tm[e % B][e / B];
}

template <class T, int B, int NC>
struct Q2Cfg {
  static constexpr int NT = B * NC / 256;  // outputs per thread of the B x NC products
  static constexpr int LPR = NC / NT;      // lanes per row
  static_assert(NT >= 1 && NC % NT == 0 && 256 / LPR == B, "thread layout: 256 threads = B rows x (NC / NT) lanes");
};

// C: n x ncols column-major (ldc).  Workgroup x handles columns [x NC, x NC + NC).  G groups per pass; ring of R = 2 G block rows.
template <class T, int B, int NC>
__global__ void __launch_bounds__(256) k_q2_apply(const T* V2, long ldv, int n, const int* blk_off, int ngroups, const T* Tb, T* C, long ldc, int ncols,
                                                  int G) {
  using Cf = Q2Cfg<T, B, NC>;
  constexpr int NT = Cf::NT, LPR = Cf::LPR;
  extern __shared__ __attribute__((aligned(16))) unsigned char q2_smem[];
  const int R = 2 * G;
  size_t off = 0;
  T(*Zr)[NC + 1] = reinterpret_cast<T(*)[NC + 1]>(q2_smem + off);  // ring: R * B rows
  off += (sizeof(T) * (size_t)R * B * (NC + 1) + 15) & ~(size_t)15;
  T(*Vc)[B + 1] = reinterpret_cast<T(*)[B + 1]>(q2_smem + off);
  off += (sizeof(T) * B * (B + 1) + 15) & ~(size_t)15;
  T(*Ts)[B + 1] = reinterpret_cast<T(*)[B + 1]>(q2_smem + off);
  off += (sizeof(T) * B * (B + 1) + 15) & ~(size_t)15;
  T(*W1)[NC + 1] = reinterpret_cast<T(*)[NC + 1]>(q2_smem + off);
  off += (sizeof(T) * B * (NC + 1) + 15) & ~(size_t)15;
  T(*W2)[NC + 1] = reinterpret_cast<T(*)[NC + 1]>(q2_smem + off);
  __shared__ T tau[B];
  const long c0 = (long)blockIdx.x * NC;
  const int row = threadIdx.x / LPR, cl = threadIdx.x % LPR;  // products: thread = (row, column set cl + LPR q)

  // block row tau_b <-> rows [tau_b B + 1, tau_b B + B] of C; slot = tau_b mod R
  auto load_blockrow = [&](int tb) {
    const int slot = tb % R;
    for (int idx = threadIdx.x; idx < B * NC; idx += 256) {
      const int rr = idx % B, cc = idx / B;
      const long gr = (long)tb * B + 1 + rr;
      Zr[slot * B + rr][cc] = (gr < n && c0 + cc < ncols) ? C[gr + (c0 + cc) * ldc] : zero_<T>();
    }
  };
  auto store_blockrow = [&](int tb) {
    const int slot = tb % R;
    for (int idx = threadIdx.x; idx < B * NC; idx += 256) {
      const int rr = idx % B, cc = idx / B;
      const long gr = (long)tb * B + 1 + rr;
      if (gr < n && c0 + cc < ncols) C[gr + (c0 + cc) * ldc] = Zr[slot * B + rr][cc];
    }
  };
  auto zrow = [&](int tb, int rho) -> int {  // LDS row of window row rho (0 .. 2B-1) of the block whose first block row is tb
    const int t2 = tb + (rho >= B ? 1 : 0);
    return (t2 % R) * B + (rho >= B ? rho - B : rho);
  };

  for (int S_hi = ngroups - 1; S_hi >= 0; S_hi -= G) {
    const int gcount = min(G, S_hi + 1);
    int u_last = 0;
    for (int i = 0; i < gcount; ++i) u_last = max(u_last, q2_nblocks(n, B, S_hi - i) - 1 + i);
    // Block rows currently in the ring: [lo, hi].  Group S_hi - i starts (k = 0) at block row S_hi - i, so the window first grows
    // upwards to S_hi - gcount + 1 while the groups join (step u: rows S_hi + u - 2 min(u, gcount - 1) .. S_hi + u + 1) - load those
    // rows up front (gcount + 1 <= 2 G block rows) - and then slides down by one block row per step.
    int lo = S_hi - gcount + 1, hi = lo - 1;
    __syncthreads();
    for (int u = 0; u <= u_last; ++u) {
      // bring the ring to cover block rows up to S_hi + u + 1, retiring what falls out of the window of R block rows
      const int want_hi = S_hi + u + 1;
      __syncthreads();  // the previous step's updates of the ring are complete before rows are retired / slots re-used
      while (hi < want_hi) {
        if (hi + 1 - lo >= R) {
          store_blockrow(lo);
          ++lo;
          __syncthreads();
        }
        ++hi;
        load_blockrow(hi);
      }
      for (int i = 0; i < gcount; ++i) {
        const int S = S_hi - i, k = u - i;
        if (k < 0 || k >= q2_nblocks(n, B, S)) continue;  // uniform
        const int tb = S + k;  // first block row of the block
        __syncthreads();       // previous block's products are done with Vc / Ts / W1 / W2; ring loads are visible
        q2_load_block<T, B>(V2, ldv, n, S, k, Vc, tau);
        {
          const T* tsrc = Tb + ((long)blk_off[S] + k) * B * B;
          for (int e = threadIdx.x; e < B * B; e += 256) Ts[e % B][e / B] = tsrc[e];
        }
        __syncthreads();
        // W1 = V^H Z:  W1[row][c] = sum_t conj(Vc[row][t]) Z[row + t][c]
        {
          T acc[NT];
#pragma unroll
          for (int q = 0; q < NT; ++q) acc[q] = zero_<T>();
          for (int t = 0; t < B; ++t) {
            const T a = conj_(Vc[row][t]);
            const int zr = zrow(tb, row + t);
#pragma unroll
            for (int q = 0; q < NT; ++q) acc[q] = acc[q] + a * Zr[zr][cl + LPR * q];
          }
#pragma unroll
          for (int q = 0; q < NT; ++q) W1[row][cl + LPR * q] = acc[q];
        }
        __syncthreads();
        // W2 = T W1 (T upper triangular)
        {
          T acc[NT];
#pragma unroll
          for (int q = 0; q < NT; ++q) acc[q] = zero_<T>();
          for (int t = row; t < B; ++t) {
            const T a = Ts[row][t];
#pragma unroll
            for (int q = 0; q < NT; ++q) acc[q] = acc[q] + a * W1[t][cl + LPR * q];
          }
#pragma unroll
          for (int q = 0; q < NT; ++q) W2[row][cl + LPR * q] = acc[q];
        }
        __syncthreads();
        // Z -= V W2: thread (row) owns window rows rho = row (terms i <= row) and rho = row + B (terms i > row): B terms in all
        {
          T a0[NT], a1[NT];
#pragma unroll
          for (int q = 0; q < NT; ++q) {
            a0[q] = zero_<T>();
            a1[q] = zero_<T>();
          }
          for (int i2 = 0; i2 <= row; ++i2) {
            const T v = Vc[i2][row - i2];
#pragma unroll
            for (int q = 0; q < NT; ++q) a0[q] = a0[q] + v * W2[i2][cl + LPR * q];
          }
          for (int i2 = row + 1; i2 < B; ++i2) {
            const T v = Vc[i2][row + B - i2];
#pragma unroll
            for (int q = 0; q < NT; ++q) a1[q] = a1[q] + v * W2[i2][cl + LPR * q];
          }
          const int z0 = zrow(tb, row), z1 = zrow(tb, row + B);
#pragma unroll
          for (int q = 0; q < NT; ++q) {
            Zr[z0][cl + LPR * q] = Zr[z0][cl + LPR * q] - a0[q];
            if (row + B < 2 * B - 1) Zr[z1][cl + LPR * q] = Zr[z1][cl + LPR * q] - a1[q];
          }
        }
      }
    }
    __syncthreads();
    for (int tb = lo; tb <= hi; ++tb) store_blockrow(tb);
  }
}

// ================================================================================================================
// MFMA form of k_q2_apply (the one the library runs).  Same passes / steps / ring as above; the three products of a block run on
// v_mfma_f64_16x16x4_f64 (A[i][k]: lane = 16 k + i, B[k][j]: lane = 16 k + j, D[i][j]: lane = 16 (i % 4) + j, reg = i / 4), NC = 16 columns
// of C per workgroup = one MFMA tile wide.  Everything in LDS is split into real planes.  Structural zeros of the parallelogram are
// skipped in units of one k-step (4): W1 tile a (reflectors 16a .. 16a+15) sums window rows 16a .. 16a + B + 15; the update of window
// rows 16b .. 16b+15 sums the reflectors that reach them; T is upper triangular.  Work per wave and block: real B = 64: 20 + 16 + 20
// MFMAs; complex B = 32: 24 + 16 + 24 (wave = (tile set, real / imaginary part of the result)).
// V and T of the NEXT block are fetched into registers while the current block computes.
// ================================================================================================================
template <class T, int B>
struct Q2M {
  static constexpr bool CX = sizeof(T) == 16;
  static constexpr int NP = CX ? 2 : 1;
  static constexpr int NC = 16;
  static constexpr int LDV = B + 3;  // (LDV - 1) = 2 (mod 32): the A-operand reads of V^H (lane stride LDV - 1) hit 32 distinct 8-byte banks
  static constexpr int EPT = B * B / 256;  // elements of V (and of T) per thread
  static_assert(!(CX && B == 64), "complex blocks of 64 do not fit the LDS budget");
  static size_t lds_bytes(int G) {
    return sizeof(double) * ((size_t)NP * (2 * G) * B * NC + (size_t)NP * B * LDV + 2 * (size_t)NP * B * NC);
  }
};

__device__ __forceinline__ void q2_split(double x, double& re, double& im) {
  re = x;
  im = 0.0;
}
__device__ __forceinline__ void q2_split(Z x, double& re, double& im) {
  re = x.re;
  im = x.im;
}

template <class T, int B>
__global__ void __launch_bounds__(256, 1) k_q2_apply_mfma(const T* V2, long ldv, int n, const int* blk_off, int ngroups, const T* Tb, T* C, long ldc, int ncols,
                                                       int G, long long* stamps /* diagnostic (nullptr: none): workgroup 0, blocks 200 .. 207 */) {
  using M = Q2M<T, B>;
  constexpr bool CX = M::CX;
  constexpr int NP = M::NP, NC = M::NC, LDV = M::LDV, EPT = M::EPT;
  extern __shared__ __attribute__((aligned(16))) unsigned char q2_smem[];
  const int R = 2 * G;
  double* ring = reinterpret_cast<double*>(q2_smem);          // [NP][R * B][NC]
  double* Vc = ring + (size_t)NP * R * B * NC;                  // [NP][B][LDV]   Vc[i][t]: entry of reflector i at window row i + t
  double* W1 = Vc + (size_t)NP * B * LDV;                       // [NP][B][NC]
  double* W2 = W1 + (size_t)NP * B * NC;                        // [NP][B][NC]
  const size_t ringp = (size_t)R * B * NC, vp = (size_t)B * LDV, wp = (size_t)B * NC;
  const long c0 = (long)blockIdx.x * NC;
  const int lane = threadIdx.x & 63;
  const int wv = __builtin_amdgcn_readfirstlane((int)(threadIdx.x >> 6));  // wave index in a scalar register: tile choices are uniform
  const int l15 = lane & 15, l4 = lane >> 4;

  auto load_blockrow = [&](int tb) {
    const int slot = tb % R;
    for (int idx = threadIdx.x; idx < B * NC; idx += 256) {
      const int rr = idx % B, cc = idx / B;
      const long gr = (long)tb * B + 1 + rr;
      double re = 0.0, im = 0.0;
      if (gr < n && c0 + cc < ncols) q2_split(C[gr + (c0 + cc) * ldc], re, im);
      ring[(size_t)(slot * B + rr) * NC + cc] = re;
      if (CX) ring[ringp + (size_t)(slot * B + rr) * NC + cc] = im;
    }
  };
  auto store_blockrow = [&](int tb) {
    const int slot = tb % R;
    for (int idx = threadIdx.x; idx < B * NC; idx += 256) {
      const int rr = idx % B, cc = idx / B;
      const long gr = (long)tb * B + 1 + rr;
      if (gr < n && c0 + cc < ncols) {
        const double re = ring[(size_t)(slot * B + rr) * NC + cc];
        const double im = CX ? ring[ringp + (size_t)(slot * B + rr) * NC + cc] : 0.0;
        C[gr + (c0 + cc) * ldc] = make_<T>(re, im);
      }
    }
  };

  int nblk_done = 0;
  // Prefetch registers.  V: this thread's EPT elements (idx = tid + 256 q: t = idx % B, i = idx / B), staged through LDS.  T: this lane's
  // A-operand fragments of the W2 = T W1 product, straight from global memory (T[i][t] at Tb[i + B t]: the 16 lanes of a k-row read 16
  // consecutive elements): tile a = a_w + ASTEP x, k-step ks, element (i = 16 a + l15, t = 4 ks + l4); only ks >= 4 a is non-zero.
  constexpr int TT = CX ? 1 : (B / 16 + 3) / 4;  // T tiles per wave
  constexpr int TKS = B / 4;
  T pv[EPT], tfn[TT][TKS], tf[TT][TKS];
  const int a_w = CX ? (wv >> 1) : wv;
  constexpr int ASTEP_T = CX ? 2 : 4;
  // V2 is zero-padded (ldv = n + B rows, whole groups of columns), so a block is loaded without bounds tests: element (i, t) of block
  // (S, k) sits at V2[u + voff[q]] with the block-uniform u = S B (ldv + 1) + 1 + k B and the per-thread constant voff = i (ldv + 1) + t;
  // only the leading entry (t = 0: it holds tau) is replaced by 1 where the reflector exists.
  unsigned voff[EPT];
#pragma unroll
  for (int q = 0; q < EPT; ++q) {
    const int idx = threadIdx.x + 256 * q;
    voff[q] = (unsigned)((idx / B) * (ldv + 1) + idx % B);
  }
  auto fetch_block = [&](int S, int k) {
    const T* tsrc = Tb + ((long)blk_off[S] + k) * B * B;
    const T* vsrc = V2 + ((long)S * B * (ldv + 1) + 1 + (long)k * B);
#pragma unroll
    for (int q = 0; q < EPT; ++q) {
      const int idx = threadIdx.x + 256 * q;
      T v = vsrc[voff[q]];
      if (idx % B == 0) v = ((long)S * B + idx / B + 1 + (long)k * B < n) ? one_<T>() : zero_<T>();
      pv[q] = v;
    }
#pragma unroll
    for (int x = 0; x < TT; ++x) {
      const int a = a_w + ASTEP_T * x;
#pragma unroll
      for (int ks = 0; ks < TKS; ++ks)
        tfn[x][ks] = (a < B / 16 && ks >= 4 * a) ? tsrc[(16 * a + l15) + (long)B * (4 * ks + l4)] : zero_<T>();
    }
  };
  auto commit_block = [&]() {
#pragma unroll
    for (int q = 0; q < EPT; ++q) {
      const int idx = threadIdx.x + 256 * q;
      double re, im;
      q2_split(pv[q], re, im);
      Vc[(size_t)(idx / B) * LDV + idx % B] = re;
      if (CX) Vc[vp + (size_t)(idx / B) * LDV + idx % B] = im;
    }
#pragma unroll
    for (int x = 0; x < TT; ++x)
#pragma unroll
      for (int ks = 0; ks < TKS; ++ks) tf[x][ks] = tfn[x][ks];
  };

  for (int S_hi = ngroups - 1; S_hi >= 0; S_hi -= G) {
    const int gcount = min(G, S_hi + 1);
    int u_last = 0;
    for (int i = 0; i < gcount; ++i) u_last = max(u_last, q2_nblocks(n, B, S_hi - i) - 1 + i);
    // cursor over the active blocks of the pass in execution order: (u, i) with k = u - i in [0, nblocks(S_hi - i))
    auto next_active = [&](int& u, int& i) -> bool {  // advance to the next active block strictly after (u, i)
      for (;;) {
        ++i;
        if (i >= gcount) {
          i = 0;
          ++u;
        }
        if (u > u_last) return false;
        const int k = u - i;
        if (k >= 0 && k < q2_nblocks(n, B, S_hi - i)) return true;
      }
    };
    int cu = 0, ci = -1;
    bool have = next_active(cu, ci);
    if (have) fetch_block(S_hi - ci, cu - ci);
    int lo = S_hi - gcount + 1, hi = lo - 1;  // block rows in the ring
    int ring_u = -1;                         // the step the ring has been prepared for
    while (have) {
      const int u = cu, i = ci;
      const int S = S_hi - i, k = u - i, tb = S + k;
      const bool stamp = stamps != nullptr && blockIdx.x == 0 && threadIdx.x == 0 && nblk_done >= 200 && nblk_done < 208;
      long long* stp = stamps + (nblk_done - 200) * 8;
      if (stamp) stp[0] = wall_clock64();
      chase::lds_barrier();  // B0: the previous block's update of the ring and its reads of Vc / Ts / W2 are complete
      if (ring_u != u) {
        const int want_hi = S_hi + u + 1;
        while (hi < want_hi) {
          if (hi + 1 - lo >= R) {
            store_blockrow(lo);
            ++lo;
            __syncthreads();
          }
          ++hi;
          load_blockrow(hi);
        }
        ring_u = u;
      }
      commit_block();
      int nu = cu, ni = ci;
      const bool more = next_active(nu, ni);
      if (more) fetch_block(S_hi - ni, nu - ni);
      if (stamp) stp[1] = wall_clock64();
      chase::lds_barrier();  // B1 (LDS only: the prefetch loads of the next block stay in flight)
      if (stamp) stp[2] = wall_clock64();
      const int base0 = (tb % R) * B, base1 = ((tb + 1) % R) * B;
      // window row rho0 + l4 (rho0 a multiple of 4, wave-uniform) of column l15 sits at ring[wbase(rho0) * NC + lane]
      auto wbase = [&](int rho0) -> int { return rho0 < B ? base0 + rho0 : base1 + rho0 - B; };
      const int part = CX ? (wv & 1) : 0;
      // ---------------- W1 = V^H Z ----------------
      {
        constexpr int NTILE = B / 16;  // row tiles of W1
        constexpr int KS = B / 4 + 4;  // k-steps per tile
        constexpr int ASTEP = CX ? 2 : 4;
        for (int a = CX ? (wv >> 1) : wv; a < NTILE; a += ASTEP) {
          const int t0 = l4 - l15;                                    // t = 4 ks + t0
          const double* va = Vc + (size_t)(16 * a + l15) * LDV + t0;  // + 4 ks
          double fvr[KS], fvi[KS], fzr[KS], fzi[KS];
#pragma unroll
          for (int ks = 0; ks < KS; ++ks) {
            const int t = 4 * ks + t0;
            const bool ok = (ks >= 4 || t >= 0) && (ks < B / 4 || t < B);
            const int zo = wbase(16 * a + 4 * ks) * NC + lane;
            const double xr = va[4 * ks];
            fvr[ks] = ok ? xr : 0.0;
            fzr[ks] = ring[zo];
            if (CX) {
              const double xi = va[vp + 4 * ks];
              fvi[ks] = ok ? xi : 0.0;
              fzi[ks] = ring[ringp + zo];
            }
          }
          // Re(conj(v) z) = vr zr + vi zi (part 0);  Im = vr zi - vi zr (part 1): the part is chosen OUTSIDE the MFMA chains (a branch per
          // k-step makes the compiler copy the accumulators around)
          double* dst = W1 + (size_t)part * wp + 256 * a + lane;
          if (!CX) {
            v4d acc = {0.0, 0.0, 0.0, 0.0};
#pragma unroll
            for (int ks = 0; ks < KS; ++ks) acc = __builtin_amdgcn_mfma_f64_16x16x4f64(fvr[ks], fzr[ks], acc, 0, 0, 0);
#pragma unroll
            for (int j = 0; j < 4; ++j) dst[64 * j] = acc[j];
          } else if (part == 0) {
            v4d acc = {0.0, 0.0, 0.0, 0.0}, acc2 = {0.0, 0.0, 0.0, 0.0};
#pragma unroll
            for (int ks = 0; ks < KS; ++ks) {
              acc = __builtin_amdgcn_mfma_f64_16x16x4f64(fvr[ks], fzr[ks], acc, 0, 0, 0);
              acc2 = __builtin_amdgcn_mfma_f64_16x16x4f64(fvi[ks], fzi[ks], acc2, 0, 0, 0);
            }
#pragma unroll
            for (int j = 0; j < 4; ++j) dst[64 * j] = acc[j] + acc2[j];
          } else {
            v4d acc = {0.0, 0.0, 0.0, 0.0}, acc2 = {0.0, 0.0, 0.0, 0.0};
#pragma unroll
            for (int ks = 0; ks < KS; ++ks) {
              acc = __builtin_amdgcn_mfma_f64_16x16x4f64(fvr[ks], fzi[ks], acc, 0, 0, 0);
              acc2 = __builtin_amdgcn_mfma_f64_16x16x4f64(fvi[ks], fzr[ks], acc2, 0, 0, 0);
            }
#pragma unroll
            for (int j = 0; j < 4; ++j) dst[64 * j] = acc[j] - acc2[j];
          }
        }
      }
      if (stamp) stp[3] = wall_clock64();
      chase::lds_barrier();  // B2
      // ---------------- W2 = T W1 (T upper triangular; its fragments are in registers) ----------------
      {
#pragma unroll
        for (int x = 0; x < TT; ++x) {
          const int a = a_w + ASTEP_T * x;
          if (a < B / 16) {  // uniform
            const double* wa = W1 + lane;  // + 64 ks
            double fwr[TKS], fwi[TKS];
#pragma unroll
            for (int ks = 0; ks < TKS; ++ks) {
              fwr[ks] = wa[64 * ks];
              if (CX) fwi[ks] = wa[wp + 64 * ks];
            }
            // T is zero below the diagonal, so the k-steps before 4 a contribute exact zeros: running them all keeps the chains branch-free
            double ftr[TKS], fti[TKS];
#pragma unroll
            for (int ks = 0; ks < TKS; ++ks) q2_split(tf[x][ks], ftr[ks], fti[ks]);
            double* dst = W2 + (size_t)part * wp + 256 * a + lane;
            if (!CX) {
              v4d acc = {0.0, 0.0, 0.0, 0.0};
#pragma unroll
              for (int ks = 0; ks < TKS; ++ks) acc = __builtin_amdgcn_mfma_f64_16x16x4f64(ftr[ks], fwr[ks], acc, 0, 0, 0);
#pragma unroll
              for (int j = 0; j < 4; ++j) dst[64 * j] = acc[j];
            } else if (part == 0) {  // Re = tr wr - ti wi
              v4d acc = {0.0, 0.0, 0.0, 0.0}, acc2 = {0.0, 0.0, 0.0, 0.0};
#pragma unroll
              for (int ks = 0; ks < TKS; ++ks) {
                acc = __builtin_amdgcn_mfma_f64_16x16x4f64(ftr[ks], fwr[ks], acc, 0, 0, 0);
                acc2 = __builtin_amdgcn_mfma_f64_16x16x4f64(fti[ks], fwi[ks], acc2, 0, 0, 0);
              }
#pragma unroll
              for (int j = 0; j < 4; ++j) dst[64 * j] = acc[j] - acc2[j];
            } else {  // Im = tr wi + ti wr
              v4d acc = {0.0, 0.0, 0.0, 0.0}, acc2 = {0.0, 0.0, 0.0, 0.0};
#pragma unroll
              for (int ks = 0; ks < TKS; ++ks) {
                acc = __builtin_amdgcn_mfma_f64_16x16x4f64(ftr[ks], fwi[ks], acc, 0, 0, 0);
                acc2 = __builtin_amdgcn_mfma_f64_16x16x4f64(fti[ks], fwr[ks], acc2, 0, 0, 0);
              }
#pragma unroll
              for (int j = 0; j < 4; ++j) dst[64 * j] = acc[j] + acc2[j];
            }
          }
        }
      }
      // ---------------- Z -= V W2 ----------------
      // The V operands of this product only depend on Vc, so they are fetched BEFORE the barrier that publishes W2 (the reads overlap the
      // wait); tile order balancing the k-steps: real B = 64: (0,3) (1,2) (4,7) (5,6); complex B = 32: (0,1) (2,3).
      constexpr int NTILE2 = 2 * B / 16;  // row tiles of the window
      constexpr int NW2 = CX ? 2 : 4;     // waves sharing the tiles (per part)
      constexpr int TPW = NTILE2 / NW2;   // tiles per wave
      const int wsel = CX ? (wv >> 1) : wv;
      int btile[TPW];
      double gvr[TPW][TKS], gvi[TPW][TKS];
#pragma unroll
      for (int bi = 0; bi < TPW; ++bi) {
        int b;
        if (TPW == 2) {
          if (NTILE2 == 8) {
            const int h = wsel >> 1, w2 = wsel & 1;
            b = 4 * h + (bi == 0 ? w2 : 3 - w2);
          } else {
            b = 2 * wsel + bi;
          }
        } else {
          b = wsel + NW2 * bi;
        }
        btile[bi] = b;
        const int tl = 16 * b + l15 - l4;               // t = tl - 4 ks
        const double* va = Vc + (size_t)l4 * LDV + tl;  // + 4 ks (LDV - 1)
#pragma unroll
        for (int ks = 0; ks < TKS; ++ks) {
          const int t = tl - 4 * ks;
          const bool ok = t >= 0 && t < B;
          const double xr = va[4 * ks * (LDV - 1)];
          gvr[bi][ks] = ok ? xr : 0.0;
          if (CX) {
            const double xi = va[vp + 4 * ks * (LDV - 1)];
            gvi[bi][ks] = ok ? xi : 0.0;
          }
        }
      }
      if (stamp) stp[4] = wall_clock64();
      chase::lds_barrier();  // B3
      {
#pragma unroll
        for (int bi = 0; bi < TPW; ++bi) {
          const int b = btile[bi];
          const double* wa = W2 + lane;  // + 64 ks
          double fwr[TKS], fwi[TKS];
          double (&fvr)[TKS] = gvr[bi];
          double (&fvi)[TKS] = gvi[bi];
#pragma unroll
          for (int ks = 0; ks < TKS; ++ks) {
            fwr[ks] = wa[64 * ks];
            if (CX) fwi[ks] = wa[wp + 64 * ks];
          }
          const int ilo = 0, ihi = 0;
          // k-steps outside [ilo / 4, ihi / 4] multiply structural zeros of V (the "ok" mask): all B / 4 are run, branch-free
          (void)ilo;
          (void)ihi;
          v4d res;
          if (!CX) {
            v4d acc = {0.0, 0.0, 0.0, 0.0};
#pragma unroll
            for (int ks = 0; ks < TKS; ++ks) acc = __builtin_amdgcn_mfma_f64_16x16x4f64(fvr[ks], fwr[ks], acc, 0, 0, 0);
            res = acc;
          } else if (part == 0) {  // Re = vr wr - vi wi
            v4d acc = {0.0, 0.0, 0.0, 0.0}, acc2 = {0.0, 0.0, 0.0, 0.0};
#pragma unroll
            for (int ks = 0; ks < TKS; ++ks) {
              acc = __builtin_amdgcn_mfma_f64_16x16x4f64(fvr[ks], fwr[ks], acc, 0, 0, 0);
              acc2 = __builtin_amdgcn_mfma_f64_16x16x4f64(fvi[ks], fwi[ks], acc2, 0, 0, 0);
            }
            res = acc - acc2;
          } else {  // Im = vr wi + vi wr
            v4d acc = {0.0, 0.0, 0.0, 0.0}, acc2 = {0.0, 0.0, 0.0, 0.0};
#pragma unroll
            for (int ks = 0; ks < TKS; ++ks) {
              acc = __builtin_amdgcn_mfma_f64_16x16x4f64(fvr[ks], fwi[ks], acc, 0, 0, 0);
              acc2 = __builtin_amdgcn_mfma_f64_16x16x4f64(fvi[ks], fwr[ks], acc2, 0, 0, 0);
            }
            res = acc + acc2;
          }
          double* dst = ring + (size_t)part * ringp + lane;
#pragma unroll
          for (int j = 0; j < 4; ++j) {
            const size_t o = (size_t)wbase(16 * b + 4 * j) * NC;
            dst[o] = dst[o] - res[j];
          }
        }
      }
      if (stamp) stp[5] = wall_clock64();
      ++nblk_done;
      cu = nu;
      ci = ni;
      have = more;
    }
    __syncthreads();
    for (int tb = lo; tb <= hi; ++tb) store_blockrow(tb);
    __syncthreads();
  }
}

}  // namespace q2
}  // namespace nls
