// fp64 MFMA tile engine for gfx950 (MI355X).
//
// A workgroup owns a 128 x 128 output tile and walks K in 16-deep slices.  Operands are staged
// global -> registers -> LDS; the next slice's global loads are issued before the current slice's
// MFMAs so HBM/L2 latency hides under the matrix pipe (v_mfma_f64_16x16x4_f64 occupies its SIMD for 64
// cycles - measured: 78 TFLOP/s chip-wide at 2.39 GHz, profiles/r01_probe_mfma.log - so one complex
// slice is ~16k cycles of matrix work per SIMD against a few hundred cycles of staging).
//
// Two wave layouts (template Cfg):
//   Cfg4 : 256 threads, 2 x 2 waves, 64 x 64 per wave (16 accumulator tiles)  - real products
//   Cfg8 : 512 threads, 2 x 4 waves, 64 x 32 per wave ( 8 accumulator tiles per plane) - complex
//          products: 2 x 8 x 8 = 128 accumulator registers, <= 256 registers per wave, two waves per
//          SIMD so one wave's LDS/VMEM phases overlap the other's MFMAs.
//
// Two arithmetic modes share the loop:
//   REAL : acc  += A B
//   CPLX : accR += Ac Br + As Bi ,  accI += Ac Bi - As Br        (split re/im planes)
// CPLX is both the rotation P = phi Q  (phi = Fc - i Fs, Q = Qr + i Qi) and, read with the panels of
// one matrix on both sides, the Hermitian Gram  A_jk = sum_i conj(F_ij) F_ik = accR - i accI.
//
// LDS images (doubles):
//   k-major tile  T[k][m]  row stride LDT = 128 + 16 : a wave's fragment read touches rows k, k+1 of
//                 16 consecutive doubles each; 144 = 16 (mod 32) puts the two rows on disjoint halves
//                 of the 64 x 4 B banks -> conflict-free ds_read_b64.
//   m-major tile  T[m][k]  row stride LDM = 16 + 2   : rows m..m+15 at k, k+1 land on 32 distinct
//                 8-byte bank pairs because 18 m mod 32 enumerates the even residues.
// MFMA f64 16x16x4 fragment maps (cdna_hip_programming.md section 3): A[i][k]: lane = 16 k + i,
// B[k][j]: lane = 16 k + j, D[i][j]: lane = 16 (i % 4) + j, reg = i / 4.
#pragma once
#include <hip/hip_runtime.h>

namespace nls {

typedef double v4d __attribute__((ext_vector_type(4)));
// Native 16-byte vector for staging: HIP's double2 is a struct whose copies lower to memcpy between
// address spaces, which can keep the staging arrays in scratch instead of registers.
typedef double v2d __attribute__((ext_vector_type(2)));

constexpr int BM = 128, BN = 128, BK = 16;
constexpr int LDT = BM + 16;            // k-major tile row stride
constexpr int LDM = BK + 2;             // m-major tile row stride
constexpr int TILE_DOUBLES = BK * LDT;  // == BM * LDM == 2304
static_assert(BK * LDT == BM * LDM, "both LDS images have the same footprint");

template <int NTHREADS_, int WAVES_M_, int WAVES_N_>
struct TileCfg {
  static constexpr int NTHREADS = NTHREADS_;
  static constexpr int WAVES_M = WAVES_M_, WAVES_N = WAVES_N_;
  static constexpr int WM = BM / WAVES_M, WN = BN / WAVES_N;  // wave tile
  static constexpr int MT = WM / 16, NTL = WN / 16;           // MFMA tiles per wave
  static constexpr int STAGE = (BK * BM / 2) / NTHREADS;      // v2d per thread per plane per slice
  static_assert(NTHREADS == 64 * WAVES_M * WAVES_N, "one wave per sub-tile");
  static __device__ __forceinline__ int wave_m() { return (threadIdx.x >> 6) / WAVES_N; }
  static __device__ __forceinline__ int wave_n() { return (threadIdx.x >> 6) % WAVES_N; }
  // Position of accumulator element (mt, nt, reg) of this lane inside the 128 x 128 tile.
  static __device__ __forceinline__ int acc_row(int mt, int reg) {
    return wave_m() * WM + mt * 16 + ((threadIdx.x & 63) >> 4) + 4 * reg;
  }
  static __device__ __forceinline__ int acc_col(int nt) { return wave_n() * WN + nt * 16 + (threadIdx.x & 15); }
};
using Cfg4 = TileCfg<256, 2, 2>;
using Cfg8 = TileCfg<512, 2, 4>;

// ------------------------------------------------------------------------------------------------
// Staging: each thread moves Cfg::STAGE 16-byte vectors per plane per K slice.
//   k-major source S[k][m] (m contiguous): idx = t + NTHREADS it -> k = idx >> 6, m = 2 (idx & 63)
//   m-major source S[m][k] (k contiguous): idx = t + NTHREADS it -> m = idx >> 3, k = 2 (idx & 7)
// ------------------------------------------------------------------------------------------------
template <class Cfg>
struct KMajorPlaneLoader {  // tile of a row-major [K][ld] plane, columns col0 .. col0+127
  const double* base;
  long ld;
  long col0;
  __device__ __forceinline__ void fetch(long k0, v2d (&r)[Cfg::STAGE]) const {
    const int t = threadIdx.x;
#pragma unroll
    for (int it = 0; it < Cfg::STAGE; ++it) {
      const int idx = t + Cfg::NTHREADS * it;
      const long k = k0 + (idx >> 6);
      r[it] = *reinterpret_cast<const v2d*>(base + k * ld + col0 + 2 * (idx & 63));
    }
  }
  static __device__ __forceinline__ void store(double* sm, const v2d (&r)[Cfg::STAGE]) {
    const int t = threadIdx.x;
#pragma unroll
    for (int it = 0; it < Cfg::STAGE; ++it) {
      const int idx = t + Cfg::NTHREADS * it;
      *reinterpret_cast<v2d*>(sm + (idx >> 6) * LDT + 2 * (idx & 63)) = r[it];
    }
  }
};

template <class Cfg>
struct MMajorPlaneLoader {  // tile of a row-major [M][ld] plane, rows row0 .. row0+127, k contiguous
  const double* base;
  long ld;
  long row0;
  __device__ __forceinline__ void fetch(long k0, v2d (&r)[Cfg::STAGE]) const {
    const int t = threadIdx.x;
#pragma unroll
    for (int it = 0; it < Cfg::STAGE; ++it) {
      const int idx = t + Cfg::NTHREADS * it;
      r[it] = *reinterpret_cast<const v2d*>(base + (row0 + (idx >> 3)) * ld + k0 + 2 * (idx & 7));
    }
  }
  static __device__ __forceinline__ void store(double* sm, const v2d (&r)[Cfg::STAGE]) {
    const int t = threadIdx.x;
#pragma unroll
    for (int it = 0; it < Cfg::STAGE; ++it) {
      const int idx = t + Cfg::NTHREADS * it;
      *reinterpret_cast<v2d*>(sm + (idx >> 3) * LDM + 2 * (idx & 7)) = r[it];
    }
  }
};

// Guarded m-major loader for the user's X (arbitrary n, d, any alignment) with the affine shift fused:
// element = X[row][k] - shift[k] inside the matrix, 0 outside.
template <class Cfg>
struct XShiftLoader {
  const double* X;
  const double* shift;
  long n, d;
  long row0;
  __device__ __forceinline__ void fetch(long k0, v2d (&r)[Cfg::STAGE]) const {
    const int t = threadIdx.x;
#pragma unroll
    for (int it = 0; it < Cfg::STAGE; ++it) {
      const int idx = t + Cfg::NTHREADS * it;
      const long row = row0 + (idx >> 3);
      const long k = k0 + 2 * (idx & 7);
      v2d v = {0.0, 0.0};
      if (row < n) {
        const double* p = X + row * d + k;
        if (k < d) v.x = p[0] - shift[k];
        if (k + 1 < d) v.y = p[1] - shift[k + 1];
      }
      r[it] = v;
    }
  }
  static __device__ __forceinline__ void store(double* sm, const v2d (&r)[Cfg::STAGE]) {
    MMajorPlaneLoader<Cfg>::store(sm, r);
  }
};

template <class Cfg, bool A_KMAJOR>
__device__ __forceinline__ double frag_a(const double* sm, int ks, int mt) {
  const int lane = threadIdx.x & 63;
  if constexpr (A_KMAJOR)
    return sm[(ks * 4 + (lane >> 4)) * LDT + Cfg::wave_m() * Cfg::WM + mt * 16 + (lane & 15)];
  else
    return sm[(Cfg::wave_m() * Cfg::WM + mt * 16 + (lane & 15)) * LDM + ks * 4 + (lane >> 4)];
}
template <class Cfg>
__device__ __forceinline__ double frag_b(const double* sm, int ks, int nt) {
  const int lane = threadIdx.x & 63;
  return sm[(ks * 4 + (lane >> 4)) * LDT + Cfg::wave_n() * Cfg::WN + nt * 16 + (lane & 15)];
}

// ------------------------------------------------------------------------------------------------
// Main loops.  smem must hold 2 (REAL) or 4 (CPLX) tiles of TILE_DOUBLES doubles.
// Fragments are double-buffered by hand: the ds_reads of K sub-step ks+1 are issued in front of the
// MFMAs of sub-step ks; the sched_barrier keeps the compiler from hoisting every sub-step's reads to
// the top of the slice (that costs ~100 VGPRs and pushes the staging registers out to scratch).
// ------------------------------------------------------------------------------------------------
template <class Cfg, bool A_KMAJOR, class ALoad, class BLoad>
__device__ __forceinline__ void mainloop_real(v4d (&acc)[Cfg::MT][Cfg::NTL], const ALoad& la, const BLoad& lb,
                                              long kbegin, int ktiles, double* smem) {
  double* smA = smem;
  double* smB = smem + TILE_DOUBLES;
  v2d ra[Cfg::STAGE], rb[Cfg::STAGE];
  la.fetch(kbegin, ra);
  lb.fetch(kbegin, rb);
  for (int kt = 0; kt < ktiles; ++kt) {
    __syncthreads();
    ALoad::store(smA, ra);
    BLoad::store(smB, rb);
    __syncthreads();
    if (kt + 1 < ktiles) {
      la.fetch(kbegin + (long)(kt + 1) * BK, ra);
      lb.fetch(kbegin + (long)(kt + 1) * BK, rb);
    }
    double a[2][Cfg::MT], b[2][Cfg::NTL];
#pragma unroll
    for (int i = 0; i < Cfg::MT; ++i) a[0][i] = frag_a<Cfg, A_KMAJOR>(smA, 0, i);
#pragma unroll
    for (int i = 0; i < Cfg::NTL; ++i) b[0][i] = frag_b<Cfg>(smB, 0, i);
#pragma unroll
    for (int ks = 0; ks < BK / 4; ++ks) {
      const int cur = ks & 1, nxt = cur ^ 1;
      __builtin_amdgcn_sched_barrier(0);
      if (ks + 1 < BK / 4) {
#pragma unroll
        for (int i = 0; i < Cfg::MT; ++i) a[nxt][i] = frag_a<Cfg, A_KMAJOR>(smA, ks + 1, i);
#pragma unroll
        for (int i = 0; i < Cfg::NTL; ++i) b[nxt][i] = frag_b<Cfg>(smB, ks + 1, i);
      }
#pragma unroll
      for (int mt = 0; mt < Cfg::MT; ++mt)
#pragma unroll
        for (int nt = 0; nt < Cfg::NTL; ++nt)
          acc[mt][nt] = __builtin_amdgcn_mfma_f64_16x16x4f64(a[cur][mt], b[cur][nt], acc[mt][nt], 0, 0, 0);
    }
  }
}

template <class Cfg, bool A_KMAJOR, class ALoad, class BLoad>
__device__ __forceinline__ void mainloop_cplx(v4d (&accR)[Cfg::MT][Cfg::NTL], v4d (&accI)[Cfg::MT][Cfg::NTL],
                                              const ALoad& lac, const ALoad& las, const BLoad& lbr, const BLoad& lbi,
                                              long kbegin, int ktiles, double* smem) {
  double* smAc = smem;
  double* smAs = smem + TILE_DOUBLES;
  double* smBr = smem + 2 * TILE_DOUBLES;
  double* smBi = smem + 3 * TILE_DOUBLES;
  v2d rac[Cfg::STAGE], ras[Cfg::STAGE], rbr[Cfg::STAGE], rbi[Cfg::STAGE];
  lac.fetch(kbegin, rac);
  las.fetch(kbegin, ras);
  lbr.fetch(kbegin, rbr);
  lbi.fetch(kbegin, rbi);
  for (int kt = 0; kt < ktiles; ++kt) {
    __syncthreads();
    ALoad::store(smAc, rac);
    ALoad::store(smAs, ras);
    BLoad::store(smBr, rbr);
    BLoad::store(smBi, rbi);
    __syncthreads();
    if (kt + 1 < ktiles) {
      const long k1 = kbegin + (long)(kt + 1) * BK;
      lac.fetch(k1, rac);
      las.fetch(k1, ras);
      lbr.fetch(k1, rbr);
      lbi.fetch(k1, rbi);
    }
    double ac[2][Cfg::MT], as[2][Cfg::MT], br[2][Cfg::NTL], bi[2][Cfg::NTL];
#pragma unroll
    for (int i = 0; i < Cfg::MT; ++i) {
      ac[0][i] = frag_a<Cfg, A_KMAJOR>(smAc, 0, i);
      as[0][i] = frag_a<Cfg, A_KMAJOR>(smAs, 0, i);
    }
#pragma unroll
    for (int i = 0; i < Cfg::NTL; ++i) {
      br[0][i] = frag_b<Cfg>(smBr, 0, i);
      bi[0][i] = frag_b<Cfg>(smBi, 0, i);
    }
#pragma unroll
    for (int ks = 0; ks < BK / 4; ++ks) {
      const int cur = ks & 1, nxt = cur ^ 1;
      __builtin_amdgcn_sched_barrier(0);
      if (ks + 1 < BK / 4) {
#pragma unroll
        for (int i = 0; i < Cfg::MT; ++i) {
          ac[nxt][i] = frag_a<Cfg, A_KMAJOR>(smAc, ks + 1, i);
          as[nxt][i] = frag_a<Cfg, A_KMAJOR>(smAs, ks + 1, i);
        }
#pragma unroll
        for (int i = 0; i < Cfg::NTL; ++i) {
          br[nxt][i] = frag_b<Cfg>(smBr, ks + 1, i);
          bi[nxt][i] = frag_b<Cfg>(smBi, ks + 1, i);
        }
      }
#pragma unroll
      for (int mt = 0; mt < Cfg::MT; ++mt) {
        const double an = -as[cur][mt];
#pragma unroll
        for (int nt = 0; nt < Cfg::NTL; ++nt) {
          accR[mt][nt] = __builtin_amdgcn_mfma_f64_16x16x4f64(ac[cur][mt], br[cur][nt], accR[mt][nt], 0, 0, 0);
          accI[mt][nt] = __builtin_amdgcn_mfma_f64_16x16x4f64(ac[cur][mt], bi[cur][nt], accI[mt][nt], 0, 0, 0);
          accR[mt][nt] = __builtin_amdgcn_mfma_f64_16x16x4f64(as[cur][mt], bi[cur][nt], accR[mt][nt], 0, 0, 0);
          accI[mt][nt] = __builtin_amdgcn_mfma_f64_16x16x4f64(an, br[cur][nt], accI[mt][nt], 0, 0, 0);
        }
      }
    }
  }
}

template <int MT, int NTL>
__device__ __forceinline__ void zero_acc(v4d (&acc)[MT][NTL]) {
#pragma unroll
  for (int i = 0; i < MT; ++i)
#pragma unroll
    for (int j = 0; j < NTL; ++j) acc[i][j] = v4d{0.0, 0.0, 0.0, 0.0};
}

}  // namespace nls
