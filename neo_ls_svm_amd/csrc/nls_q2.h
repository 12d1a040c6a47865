// Two-stage tridiagonalisation, second back-transformation:  C <- Q2 C  with Q2 = prod_s prod_k H_{s,k} the reflectors of the bulge
// chase (nls_chase.h; H_{s,k} acts on rows s + 1 + k B .. s + k B + B).
//
// Reflectors of B consecutive sweeps (a "group" S: sweeps S B .. S B + B - 1) at the same step k form a parallelogram-shaped block
// V_{S,k}: column i is the reflector of sweep S B + i, non-zero in rows i .. i + B - 1 of the 2B - 1 rows the block spans, which start at
// row (S + k) B + 1.  Their product is the compact WY transform I - V T V^H with T^-1 = striu(V^H V) + diag(1 / tau) (k_q2_tfactor, one
// workgroup per block).  Order (see tools/twostage_proto.py for the derivation and the check against the reflector-by-reflector
// product): groups descending, steps ascending inside a group; block (S - 1, k) needs the blocks (S, k' <= k).
//
// k_q2_apply (VALU reference form; the library runs k_q2_apply_packed below, same passes / steps / ring on fp64 MFMA): one workgroup per slab
// of NC columns of C, no communication between workgroups.  G groups are taken together ("pass"):
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

// Operands of one block in the layout the MFMA kernel (k_q2_apply_packed) consumes, written once by k_q2_tfactor: every lane of a wave
// then fetches its A-operand fragments with contiguous 16-byte loads straight into registers - no LDS staging of V, no T stage:
//   F1[a][ks][lane]  = conj(V[rho][i]),  rho = 16 a + 4 ks + lane / 16, i = 16 a + lane % 16      (W1 = V^H Z; tile a = reflectors 16a ..)
//   F2[b][ks][lane]  = (V T)[rho][j],    rho = 16 b + lane % 16,         j = 4 ks + lane / 16      (Z -= (V T) W1; tile b = window rows 16b ..)
// with V[rho][i] = Vc[i][rho - i] (0 <= rho - i < B).  V T is zero for rho >= j + B: the k-steps ks < ks2_first(b) of tile b are structural zeros
// (stored, never read).  Real operands are stored in pairs of k-steps so that one 16-byte load feeds two MFMAs.
// Leading dimension of the LDS ring (doubles per column of the slab) = rows + Q2_RING_PAD.  A lane (column c = lane % 16, row group lane / 16) reads
// and updates ring[c LDR + row]; hipcc pairs these accesses into ds_read2_b64 / ds_write2_b64, which the LDS serves in groups of 16 consecutive
// lanes over 32 four-byte banks: the 16 columns of a group must fall on 16 different bank pairs, i.e. LDR must be ODD (round 4: the + 4 of round 3
// was laid out for ds_read_b64's 64 banks and is a 4-way conflict in the paired form - SQ_LDS_BANK_CONFLICT was a quarter of the kernel time).
constexpr int Q2_RING_PAD = 1;

template <class T, int B>
struct Q2P {
  static constexpr bool CX = sizeof(T) == 16;
  static constexpr int NP = CX ? 2 : 1;
  static constexpr int NT1 = B / 16, KS1 = B / 4 + 4, NT2 = 2 * B / 16, KS2 = B / 4;
  static constexpr size_t F1 = (size_t)NT1 * KS1 * 64, F2 = (size_t)NT2 * KS2 * 64, PER_BLOCK = F1 + F2;  // elements of T per block
  static constexpr bool AVAILABLE = !(CX && B == 64);  // complex blocks of 64: k_q2_tfactor's three B x B matrices do not fit the LDS
  __host__ __device__ static constexpr int ks2_first(int b) { return 4 * b - B / 4 > 0 ? 4 * b - B / 4 : 0; }
  __host__ __device__ static size_t pos(int tile, int KS, int ks, int lane) {
    return CX ? ((size_t)tile * KS + ks) * 64 + lane : (((size_t)tile * (KS / 2) + ks / 2) * 64 + lane) * 2 + (ks & 1);
  }
  // Real blocks of 32 hold too little work for four waves (2 + 4 tiles): the waves form two teams that take two blocks of the same step
  // - they lie two block rows apart, hence independent - side by side, each with its own W1.
  static constexpr int TEAMS = (!CX && B == 32) ? 2 : 1;
  static size_t lds_bytes(int G) { return sizeof(double) * ((size_t)NP * 16 * ((size_t)2 * G * B + Q2_RING_PAD) + (size_t)TEAMS * NP * B * 16); }
};

// T factor of every block: Tb[block][i + B j] (upper triangular, zeros below; Tb == nullptr: not stored) and the packed operands P
// (nullptr: not stored).
template <class T, int B>
__global__ void __launch_bounds__(256) k_q2_tfactor(const T* V2, long ldv, int n, const int* blk_off, int ngroups, T* Tb, T* P) {
  using L = Q2P<T, B>;
  extern __shared__ __attribute__((aligned(16))) unsigned char q2_smem[];
  T(*Vc)[B + 1] = reinterpret_cast<T(*)[B + 1]>(q2_smem);
  T(*Ti)[B + 1] = reinterpret_cast<T(*)[B + 1]>(q2_smem + sizeof(T) * B * (B + 1));
  // with the packed operands the block is still needed after T is formed (V T); without them T overwrites it
  T(*Tm)[B + 1] = L::AVAILABLE ? reinterpret_cast<T(*)[B + 1]>(q2_smem + 2 * sizeof(T) * B * (B + 1)) : Vc;
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
  if (Tb != nullptr) {
    T* out = Tb + (long)b * B * B;
    for (int e = threadIdx.x; e < B * B; e += 256) out[e] = Tm[e % B][e / B];
  }
  if constexpr (L::AVAILABLE) {
    if (P == nullptr) return;
    T* out = P + (size_t)b * L::PER_BLOCK;
    for (int e = threadIdx.x; e < (int)L::F1; e += 256) {
      const int lane = e % 64, ks = (e / 64) % L::KS1, a = e / (64 * L::KS1);
      const int i = 16 * a + (lane & 15), t = 4 * ks + (lane >> 4) - (lane & 15);
      out[L::pos(a, L::KS1, ks, lane)] = (t >= 0 && t < B) ? conj_(Vc[i][t]) : zero_<T>();
    }
    out += L::F1;
    for (int e = threadIdx.x; e < (int)L::F2; e += 256) {
      const int lane = e % 64, ks = (e / 64) % L::KS2, bt = e / (64 * L::KS2);
      const int rho = 16 * bt + (lane & 15), j = 4 * ks + (lane >> 4);
      T s = zero_<T>();
      for (int i = max(0, rho - B + 1); i <= min(j, rho); ++i) s = s + Vc[i][rho - i] * Tm[i][j];
      out[L::pos(bt, L::KS2, ks, lane)] = s;
    }
  }
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
// MFMA form of k_q2_apply (the one the library runs).  Same passes / steps / ring as above, 16 columns of C per workgroup = one tile of
// v_mfma_f64_16x16x4_f64 (A[i][k]: lane = 16 k + i, B[k][j]: lane = 16 k + j, D[i][j]: lane = 16 (i % 4) + j, reg = i / 4).  A block is two
// products, W1 = V^H Z and Z -= (V T) W1, whose A operands come from the packed copy Q2P in registers: the fragments of the NEXT block are
// requested as soon as the current ones have been consumed, so they travel while the other product runs.  LDS holds only the ring (real
// planes, column-major with a leading dimension = 4 mod 32 doubles: operand reads, accumulator updates and the row transfers are all
// conflict-free) and W1 - 48 to 80 KB, two workgroups per CU, which is what hides the barriers and the LDS latency of each other.
// Waves: real: wave = tile set; complex: wave = (tile set, real / imaginary part of the result); real blocks of 32: two teams of two
// waves, two independent blocks of a step at a time.  The roles rotate with the workgroup index so that co-resident workgroups put
// their heavier waves on different SIMDs.
// ================================================================================================================
__device__ __forceinline__ void q2_split(double x, double& re, double& im) {
  re = x;
  im = 0.0;
}
__device__ __forceinline__ void q2_split(Z x, double& re, double& im) {
  re = x.re;
  im = x.im;
}
// fragment registers of one tile: real: pairs of k-steps; complex: (re, im) of one k-step
template <class T, int KS>
struct Q2Frag;
template <int KS>
struct Q2Frag<double, KS> {
  double2 v[KS / 2];
  __device__ __forceinline__ void load(const double* base, int tile, int lane) {
    const double2* p = reinterpret_cast<const double2*>(base) + (size_t)tile * (KS / 2) * 64 + lane;
#pragma unroll
    for (int q = 0; q < KS / 2; ++q) v[q] = p[64 * q];
  }
  __device__ __forceinline__ double re(int ks) const { return (ks & 1) ? v[ks / 2].y : v[ks / 2].x; }
  __device__ __forceinline__ double im(int) const { return 0.0; }
};
template <int KS>
struct Q2Frag<Z, KS> {
  double2 v[KS];
  __device__ __forceinline__ void load(const Z* base, int tile, int lane) {
    const double2* p = reinterpret_cast<const double2*>(base) + (size_t)tile * KS * 64 + lane;
#pragma unroll
    for (int q = 0; q < KS; ++q) v[q] = p[64 * q];
  }
  __device__ __forceinline__ double re(int ks) const { return v[ks].x; }
  __device__ __forceinline__ double im(int ks) const { return v[ks].y; }
};

template <class T, int B>
__global__ void __launch_bounds__(256, 2) k_q2_apply_packed(const T* __restrict__ P, const int* __restrict__ blk_off, int ngroups, int n, T* C, long ldc,
                                                             int ncols, int G, long long* stamps /* diagnostic (nullptr: none): workgroup 0, blocks 200 .. 207 */) {
  using L = Q2P<T, B>;
  constexpr bool CX = L::CX;
  constexpr int NP = L::NP, KS1 = L::KS1, KS2 = L::KS2, NT1 = L::NT1, NT2 = L::NT2;
  constexpr int TEAMS = L::TEAMS, WT = 4 / TEAMS;  // teams of WT waves, one block per team at a time
  constexpr int NW = CX ? WT / 2 : WT;             // waves per part of the result
  constexpr int TPW = NT2 / NW;                    // window tiles per wave
  static_assert(NT1 <= NW && (TPW == 1 || TPW == 2), "tile distribution");
  extern __shared__ __attribute__((aligned(16))) unsigned char q2_smem[];
  const int R = 2 * G, LDR = R * B + Q2_RING_PAD;
  double* ring = reinterpret_cast<double*>(q2_smem);  // [NP][16][LDR]: ring[c LDR + slot B + r]
  const int ringp = 16 * LDR, wp = B * 16;
  const long c0 = (long)blockIdx.x * 16;
  const int lane = threadIdx.x & 63;
  const int wv = __builtin_amdgcn_readfirstlane((int)(threadIdx.x >> 6));
  const int l15 = lane & 15, l4 = lane >> 4;
  const int team = wv / WT;
  const int role = (wv % WT + (int)(blockIdx.x >> 8)) % WT;
  const int wsel = CX ? role >> 1 : role, part = CX ? role & 1 : 0;
  double* W1 = ring + (size_t)NP * 16 * LDR + (size_t)team * NP * wp;  // [NP][B][16], one per team
  const int rl = l15 * LDR + l4;  // this lane's element of a 4-row x 16-column operand / accumulator slice: + first row
  // complex products (ar + i ai)(br + i bi): part 0 sums ar br and ai bi (result: difference), part 1 sums ar bi and ai br (result: sum);
  // which plane feeds which chain is a matter of addresses, decided once
  const int pl1 = part ? 1 : 0, pl2 = part ? 0 : 1;
  const double sgn = part ? 1.0 : -1.0;

  // ---- block rows: C <-> ring, one row in flight in registers
  constexpr int EPR = B * 16 / 256;
  T pre[EPR];
  auto fetch_row = [&](int tb) {
#pragma unroll
    for (int q = 0; q < EPR; ++q) {
      const int idx = threadIdx.x + 256 * q, rr = idx % B, cc = idx / B;
      const long gr = (long)tb * B + 1 + rr;
      pre[q] = (gr < n && c0 + cc < ncols) ? C[gr + (c0 + cc) * ldc] : zero_<T>();
    }
  };
  auto commit_row = [&](int tb) {
    const int slot = tb % R;
#pragma unroll
    for (int q = 0; q < EPR; ++q) {
      const int idx = threadIdx.x + 256 * q, rr = idx % B, cc = idx / B;
      double re, im;
      q2_split(pre[q], re, im);
      ring[cc * LDR + slot * B + rr] = re;
      if (CX) ring[ringp + cc * LDR + slot * B + rr] = im;
    }
  };
  auto store_row = [&](int tb) {
    const int slot = tb % R;
#pragma unroll
    for (int q = 0; q < EPR; ++q) {
      const int idx = threadIdx.x + 256 * q, rr = idx % B, cc = idx / B;
      const long gr = (long)tb * B + 1 + rr;
      if (gr < n && c0 + cc < ncols) {
        const double re = ring[cc * LDR + slot * B + rr];
        const double im = CX ? ring[ringp + cc * LDR + slot * B + rr] : 0.0;
        C[gr + (c0 + cc) * ldc] = make_<T>(re, im);
      }
    }
  };

  // window tiles of this wave
  int btile[TPW];
  btile[0] = wsel;
  if (TPW == 2) btile[TPW - 1] = NT2 - 1 - wsel;
  Q2Frag<T, KS1> f1;
  Q2Frag<T, KS2> f2[TPW];
  auto fetch1 = [&](int S, int k) {
    if (wsel < NT1) f1.load(P + ((size_t)blk_off[S] + k) * L::PER_BLOCK, wsel, lane);
  };
  auto fetch2 = [&](int S, int k) {
#pragma unroll
    for (int bi = 0; bi < TPW; ++bi) f2[bi].load(P + ((size_t)blk_off[S] + k) * L::PER_BLOCK + L::F1, btile[bi], lane);
  };

  int nblk_done = 0;
  for (int S_hi = ngroups - 1; S_hi >= 0; S_hi -= G) {
    const int gcount = min(G, S_hi + 1);
    int u_last = 0;
    for (int i = 0; i < gcount; ++i) u_last = max(u_last, q2_nblocks(n, B, S_hi - i) - 1 + i);
    // cursor over the active blocks of the pass in execution order: (u, i) with k = u - i in [0, nblocks(S_hi - i))
    auto next_active = [&](int& u, int& i) -> bool {
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
    // one iteration = the next active block (team 0) and, with two teams, the following active block if it belongs to the SAME step (-1: none)
    auto next_iter = [&](int& u, int& i, int& i0, int& i1) -> bool {
      if (!next_active(u, i)) return false;
      i0 = i;
      i1 = -1;
      if (TEAMS == 2) {
        int u2 = u, i2 = i;
        if (next_active(u2, i2) && u2 == u) {
          i1 = i2;
          i = i2;
        }
      }
      return true;
    };
    static_assert(TEAMS <= 2, "next_iter hands out at most two blocks");
    int cu = 0, ci = -1, ci0 = -1, ci1 = -1;
    bool have = next_iter(cu, ci, ci0, ci1);
    {
      const int mine = team == 0 ? ci0 : ci1;
      if (have && mine >= 0) {
        fetch1(S_hi - mine, cu - mine);
        fetch2(S_hi - mine, cu - mine);
      }
    }
    // the ring starts with the block rows S_hi - gcount + 1 .. S_hi + 1 (what step 0 and the joining groups need) and slides by one per step
    int lo = S_hi - gcount + 1, hi = lo - 1;
    __syncthreads();  // the previous pass has written its rows back
    while (hi < S_hi + 1) {
      ++hi;
      fetch_row(hi);
      commit_row(hi);
    }
    fetch_row(hi + 1);
    int pre_tb = hi + 1;
    while (have) {
      const int u = cu, i = team == 0 ? ci0 : ci1;
      const bool active = i >= 0;
      const int S = S_hi - i, k = u - i, tb = S + k;
      const bool stamp = stamps != nullptr && blockIdx.x == 0 && threadIdx.x == 0 && nblk_done >= 200 && nblk_done < 208;
      long long* stp = stamps + (nblk_done - 200) * 8;
      if (stamp) stp[0] = wall_clock64();
      chase::lds_barrier();  // B0: the previous blocks' updates of the ring and their reads of W1 are complete
      if (hi < S_hi + u + 1) {
        while (hi < S_hi + u + 1) {
          if (hi + 1 - lo >= R) {
            store_row(lo);
            ++lo;
            chase::lds_barrier();  // the slot has been read by everyone before it is overwritten
          }
          ++hi;
          if (pre_tb != hi) fetch_row(hi);
          commit_row(hi);
        }
        fetch_row(hi + 1);
        pre_tb = hi + 1;
        chase::lds_barrier();
      }
      int nu = cu, ni = ci, ni0 = -1, ni1 = -1;
      const bool more = next_iter(nu, ni, ni0, ni1);
      const int nmine = team == 0 ? ni0 : ni1;
      const bool more_mine = more && nmine >= 0;
      if (stamp) stp[1] = wall_clock64();
      const int base0 = (tb % R) * B, base1 = ((tb + 1) % R) * B;
      // first ring row of the 4-row slice that starts at window row rho0 (a multiple of 4, wave-uniform)
      auto wbase = [&](int rho0) -> int { return rho0 < B ? base0 + rho0 : base1 + rho0 - B; };
      // ---------------- W1 = V^H Z ----------------
      if (active && wsel < NT1) {
        const int a = wsel;
        double z1[KS1], z2[KS1];
#pragma unroll
        for (int ks = 0; ks < KS1; ++ks) {
          const int o = rl + wbase(16 * a + 4 * ks);
          z1[ks] = ring[pl1 * ringp + o];
          if (CX) z2[ks] = ring[pl2 * ringp + o];
        }
        v4d acc = {0.0, 0.0, 0.0, 0.0}, acc2 = {0.0, 0.0, 0.0, 0.0};
#pragma unroll
        for (int ks = 0; ks < KS1; ++ks) {
          acc = __builtin_amdgcn_mfma_f64_16x16x4f64(f1.re(ks), z1[ks], acc, 0, 0, 0);
          if (CX) acc2 = __builtin_amdgcn_mfma_f64_16x16x4f64(f1.im(ks), z2[ks], acc2, 0, 0, 0);
        }
        double* dst = W1 + part * wp + 256 * a + lane;
#pragma unroll
        for (int j = 0; j < 4; ++j) dst[64 * j] = CX ? acc[j] + sgn * acc2[j] : acc[j];
      }
      if (more_mine) fetch1(S_hi - nmine, nu - nmine);
      if (stamp) stp[2] = wall_clock64();
      chase::lds_barrier();  // B1: W1 is complete
      if (stamp) stp[3] = wall_clock64();
      // ---------------- Z -= (V T) W1 ----------------
      if (active) {
        double w1[KS2], w2[KS2];
#pragma unroll
        for (int ks = 0; ks < KS2; ++ks) {
          w1[ks] = W1[pl1 * wp + 64 * ks + lane];
          if (CX) w2[ks] = W1[pl2 * wp + 64 * ks + lane];
        }
#pragma unroll
        for (int bi = 0; bi < TPW; ++bi) {
          const int b = btile[bi];
          v4d acc = {0.0, 0.0, 0.0, 0.0}, acc2 = {0.0, 0.0, 0.0, 0.0};
#pragma unroll
          for (int ks = 0; ks < KS2; ++ks) {
            acc = __builtin_amdgcn_mfma_f64_16x16x4f64(f2[bi].re(ks), w1[ks], acc, 0, 0, 0);
            if (CX) acc2 = __builtin_amdgcn_mfma_f64_16x16x4f64(f2[bi].im(ks), w2[ks], acc2, 0, 0, 0);
          }
          double* dst = ring + part * ringp + rl;
#pragma unroll
          for (int j = 0; j < 4; ++j) {
            const int o = wbase(16 * b + 4 * j);
            dst[o] = dst[o] - (CX ? acc[j] + sgn * acc2[j] : acc[j]);
          }
        }
      }
      if (more_mine) fetch2(S_hi - nmine, nu - nmine);
      if (stamp) stp[4] = wall_clock64();
      ++nblk_done;
      cu = nu;
      ci = ni;
      ci0 = ni0;
      ci1 = ni1;
      have = more;
    }
    __syncthreads();
    for (int tb = lo; tb <= hi; ++tb) store_row(tb);
  }
}


// ================================================================================================================
// Real blocks of 32, one WAVE per block plus a MOVER wave (round 4; opt-in with NLS_Q2_FORM=wave: bit-identical to k_q2_apply_packed, measured
// 8 % SLOWER at n = 10^4 - 117 against 108 ms, profiles/r04_q2_forms.md - and kept as the record of the experiment).  In k_q2_apply_packed two waves share a block and meet at two
// workgroup barriers per block, and every thread takes part in sliding the ring: a block is only 56 matrix instructions, and on this
// hardware a wave's vector-memory counter is in order - the first wait for an operand fragment (an L2 hit) also waits for the ring's
// next block row, which comes from HBM.  Here the two kinds of traffic live in different waves:
//   * three COMPUTE waves: a pass takes G = 3 groups, wave w owns group S_hi - w and at step u applies block (S_hi - w, u - w) all by
//     itself: W1 is private to the wave (a wave-level fence orders its LDS write and read), the (V T) fragments of the block are requested
//     before the first product and the V^H fragments of the next block before the second; their only global accesses are these L2 hits;
//   * one MOVER wave: while the others work on the window of step u (block rows S_hi + u - 2G + 2 .. S_hi + u + 1) it retires the block row
//     that left the window after step u - 1 (LDS -> global) and puts the row that step u + 1 needs into the freed slot (requested one
//     step earlier) - the ring has 2G + 1 slots so that this touches no row of the current window.
// One workgroup barrier per step.  41.5 KB of LDS, <= 168 registers: three workgroups per CU; the roles rotate with the workgroup index so
// that co-resident workgroups put their mover on different SIMDs.  Same packed operands (Q2P) and the same order of operations per
// element as k_q2_apply_packed: bit-identical results.
// ================================================================================================================
struct Q2Wave {
  static constexpr int B = 32, G = 3, R = 2 * G + 1, LDR = R * B + Q2_RING_PAD;
  static constexpr size_t lds_bytes() { return sizeof(double) * ((size_t)16 * LDR + (size_t)G * B * 16); }
};

__global__ void __launch_bounds__(256, 3) k_q2_apply_wave(const double* __restrict__ P, const int* __restrict__ blk_off, int ngroups, int n, double* C,
                                                           long ldc, int ncols, long long* stamps /* diagnostic (nullptr: none) */) {
  constexpr int B = Q2Wave::B, G = Q2Wave::G, R = Q2Wave::R, LDR = Q2Wave::LDR;
  using L = Q2P<double, B>;
  constexpr int KS1 = L::KS1, KS2 = L::KS2, NT1 = L::NT1, NT2 = L::NT2;
  static_assert(NT1 == 2 && NT2 == 4 && KS1 == 12 && KS2 == 8, "tile counts of a real block of 32");
  extern __shared__ __attribute__((aligned(16))) unsigned char q2_smem[];
  double* ring = reinterpret_cast<double*>(q2_smem);  // [16][LDR]: ring[c LDR + slot B + r], slot = block row % R
  const long c0 = (long)blockIdx.x * 16;
  const int lane = threadIdx.x & 63;
  const int wv = __builtin_amdgcn_readfirstlane((int)(threadIdx.x >> 6));
  // 0 .. G - 1: compute wave of group S_hi - role; G: mover.  Workgroups that share a CU come from the same XCD (index = multiples of 8 apart),
  // consecutive or 32 apart in index / 8 depending on how the dispatcher walks the CUs: rotate with both
  const int role = (wv + (int)(blockIdx.x >> 3) + (int)(blockIdx.x >> 8)) & 3;
  double* W1 = ring + (size_t)16 * LDR + (size_t)(role < G ? role : 0) * B * 16;  // [B][16], this wave's

  // whole-workgroup transfers of one block row (start and end of a pass)
  constexpr int EPR = B * 16 / 256;
  auto load_row_all = [&](int tb) {
#pragma unroll
    for (int q = 0; q < EPR; ++q) {
      const int idx = threadIdx.x + 256 * q, rr = idx % B, cc = idx / B;
      const long gr = (long)tb * B + 1 + rr;
      ring[cc * LDR + (tb % R) * B + rr] = (gr < n && c0 + cc < ncols) ? C[gr + (c0 + cc) * ldc] : 0.0;
    }
  };
  auto store_row_all = [&](int tb) {
#pragma unroll
    for (int q = 0; q < EPR; ++q) {
      const int idx = threadIdx.x + 256 * q, rr = idx % B, cc = idx / B;
      const long gr = (long)tb * B + 1 + rr;
      if (gr < n && c0 + cc < ncols) C[gr + (c0 + cc) * ldc] = ring[cc * LDR + (tb % R) * B + rr];
    }
  };
  // the mover's transfers: 8 elements per lane
  constexpr int EPM = B * 16 / 64;
  auto mover_fetch = [&](double (&pre)[EPM], int tb) {
#pragma unroll
    for (int q = 0; q < EPM; ++q) {
      const int idx = lane + 64 * q, rr = idx % B, cc = idx / B;
      const long gr = (long)tb * B + 1 + rr;
      pre[q] = (gr < n && c0 + cc < ncols) ? C[gr + (c0 + cc) * ldc] : 0.0;
    }
  };
  // retire block row `old` (same slot as `tb`; old < first: never loaded, nothing to store) and put the fetched row tb in its place
  auto mover_swap = [&](const double (&pre)[EPM], int tb, int old, int first) {
    const int slot = tb % R;
#pragma unroll
    for (int q = 0; q < EPM; ++q) {
      const int idx = lane + 64 * q, rr = idx % B, cc = idx / B;
      double* cell = ring + cc * LDR + slot * B + rr;
      const long gr = (long)old * B + 1 + rr;
      if (old >= first && gr < n && c0 + cc < ncols) C[gr + (c0 + cc) * ldc] = *cell;
      *cell = pre[q];
    }
  };

  Q2Frag<double, KS1> f1[NT1];
  Q2Frag<double, KS2> f2[NT2];
  for (int S_hi = ngroups - 1; S_hi >= 0; S_hi -= G) {
    const int gcount = min(G, S_hi + 1);
    int u_last = 0;
    for (int i = 0; i < gcount; ++i) u_last = max(u_last, q2_nblocks(n, B, S_hi - i) - 1 + i);
    const bool mine = role < gcount;
    const int S = S_hi - role;
    const int nb = mine ? q2_nblocks(n, B, S) : 0;
    const double* Pg = P + (size_t)(mine ? blk_off[S] : 0) * L::PER_BLOCK;
    if (nb > 0) {  // (never true for the mover: its role is not a group)
#pragma unroll
      for (int a = 0; a < NT1; ++a) f1[a].load(Pg, a, lane);
#pragma unroll
      for (int b = 0; b < NT2; ++b) f2[b].load(Pg + L::F1, b, lane);
    }
    // the ring starts with the block rows first .. S_hi + 1 (what step 0 and the joining groups need); the mover adds one row per step
    const int first = S_hi - gcount + 1;
    __syncthreads();  // the previous pass has written its rows back
    if (stamps != nullptr && threadIdx.x == 0 && (blockIdx.x == 0 || blockIdx.x == 300)) {
      const int pi = (ngroups - 1 - S_hi) / G;
      if (pi < 120) stamps[128 + (blockIdx.x == 0 ? 0 : 128) + pi] = wall_clock64();
    }
    for (int tb = first; tb <= S_hi + 1; ++tb) load_row_all(tb);
    if (role == G) {
      // ---- mover (its own loop: the fetched row's registers are live across steps only here) ----
      // three rows in flight (a row comes from HBM: ~ 2 us under load, a step lasts ~ 1.5 us; one row per workgroup in flight would also be
      // too few bytes in flight chip-wide); the step loop is unrolled by three so that the buffers are named statically
      double pre[3][EPM];
#pragma unroll
      for (int j = 0; j < 3; ++j)
        if (j <= u_last) mover_fetch(pre[j], S_hi + 2 + j);
      for (int u = 0; u <= u_last; u += 3) {
#pragma unroll
        for (int j = 0; j < 3; ++j) {
          if (u + j <= u_last) {
            chase::lds_barrier();
            // row S_hi + u + 2 (first needed at step u + 1) into the slot of row S_hi + u + 2 - R, which left the window after step u - 1
            mover_swap(pre[j], S_hi + u + j + 2, S_hi + u + j + 2 - R, first);
            if (u + j + 3 <= u_last) mover_fetch(pre[j], S_hi + u + j + 5);
          }
        }
      }
    } else {
      for (int u = 0; u <= u_last; ++u) {
        chase::lds_barrier();  // the previous step's ring updates and the mover's row are complete
        const int k = u - role;
        if (k < 0 || k >= nb) continue;  // wave-uniform
        const int tb = S + k;
        const double* Pn = Pg + (size_t)(k + 1) * L::PER_BLOCK;  // the next block of this group
        const bool more = k + 1 < nb;
        // the lane index is formed again in every block: as a loop invariant it (and everything derived from it) would be one more set of
        // registers live across the loop, and the allocator then spills one of them - a scratch reload in the loop makes every wait a full drain
        int ln;
        asm volatile("v_mbcnt_lo_u32_b32 %0, -1, 0\n\tv_mbcnt_hi_u32_b32 %0, -1, %0" : "=v"(ln));
        // window rows 0 .. 31: ring slot tb % R, rows 32 .. 63: slot (tb + 1) % R
        double* zb = ring + (ln & 15) * LDR + (ln >> 4);
        double* z0 = zb + (tb % R) * B;
        double* z1 = zb + ((tb + 1) % R) * B - B;
        double* W1l = W1 + ln;
        // ---------------- W1 = V^H Z: tile a covers window rows 16 a .. 16 a + 47 (k-step q of the window = k-step q - 4 a of tile a) ----
        v4d acc1[NT1];
#pragma unroll
        for (int a = 0; a < NT1; ++a) acc1[a] = v4d{0.0, 0.0, 0.0, 0.0};
#pragma unroll
        for (int q = 0; q < 16; ++q) {
          const double zq = q < 8 ? z0[4 * q] : z1[4 * q];
          if (q < KS1) acc1[0] = __builtin_amdgcn_mfma_f64_16x16x4f64(f1[0].re(q), zq, acc1[0], 0, 0, 0);
          if (q >= 4) acc1[1] = __builtin_amdgcn_mfma_f64_16x16x4f64(f1[1].re(q - 4), zq, acc1[1], 0, 0, 0);
        }
        __builtin_amdgcn_sched_barrier(0);
        // operand fragments travel one block ahead, each set requested into its registers as soon as the product that read them has been
        // issued (an L2 hit takes longer than one product lasts)
        if (more) {
#pragma unroll
          for (int a = 0; a < NT1; ++a) f1[a].load(Pn, a, ln);
        }
#pragma unroll
        for (int a = 0; a < NT1; ++a)
#pragma unroll
          for (int j = 0; j < 4; ++j) W1l[256 * a + 64 * j] = acc1[a][j];
        __builtin_amdgcn_fence(__ATOMIC_RELEASE, "wavefront");
        __builtin_amdgcn_wave_barrier();
        __builtin_amdgcn_sched_barrier(0);
        // ---------------- Z -= (V T) W1, two tiles at a time ----------------
        double w1[KS2];
#pragma unroll
        for (int ks = 0; ks < KS2; ++ks) w1[ks] = W1l[64 * ks];
#pragma unroll
        for (int h = 0; h < 2; ++h) {
          v4d acc2[2];
          acc2[0] = acc2[1] = v4d{0.0, 0.0, 0.0, 0.0};
#pragma unroll
          for (int ks = 0; ks < KS2; ++ks)
#pragma unroll
            for (int bb = 0; bb < 2; ++bb) acc2[bb] = __builtin_amdgcn_mfma_f64_16x16x4f64(f2[2 * h + bb].re(ks), w1[ks], acc2[bb], 0, 0, 0);
          __builtin_amdgcn_sched_barrier(0);
          if (more) {
#pragma unroll
            for (int bb = 0; bb < 2; ++bb) f2[2 * h + bb].load(Pn + L::F1, 2 * h + bb, ln);
          }
#pragma unroll
          for (int bb = 0; bb < 2; ++bb)
#pragma unroll
            for (int j = 0; j < 4; ++j) {
              double* d = (h == 0 ? z0 : z1) + 16 * (2 * h + bb) + 4 * j;
              *d = *d - acc2[bb][j];
            }
          __builtin_amdgcn_sched_barrier(0);
        }
      }
    }
    __syncthreads();
    // what the mover has not retired: the rows of the last window and the row it brought in during the last step
    for (int tb = max(first, S_hi + u_last + 3 - R); tb <= S_hi + u_last + 2; ++tb) store_row_all(tb);
  }
}

}  // namespace q2
}  // namespace nls
