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
template <class Cfg, bool WEIGHTED = false>
struct KMajorPlaneLoader {  // tile of a row-major [K][ld] plane, columns col0 .. col0+127
  const double* base;
  long ld;
  long col0;
  // Optional per-k weights (the Gram's s_i^2).  They are LOADED with the slice but only APPLIED when the slice
  // is written to LDS one iteration later: multiplying at fetch time would make the wave wait for its own
  // global loads in the middle of the MFMA stream (measured: Gram 849 -> 1462 ms).
  const double* w = nullptr;
  static constexpr int NREG = Cfg::STAGE + (WEIGHTED ? (Cfg::STAGE + 1) / 2 : 0);
  __device__ __forceinline__ void fetch(long k0, v2d (&r)[NREG]) const {
    const int t = threadIdx.x;
#pragma unroll
    for (int it = 0; it < Cfg::STAGE; ++it) {
      const int idx = t + Cfg::NTHREADS * it;
      const long k = k0 + (idx >> 6);
      r[it] = *reinterpret_cast<const v2d*>(base + k * ld + col0 + 2 * (idx & 63));
      if constexpr (WEIGHTED) r[Cfg::STAGE + it / 2][it & 1] = w[k];
    }
  }
  static __device__ __forceinline__ void store(double* sm, const v2d (&r)[NREG]) {
    const int t = threadIdx.x;
#pragma unroll
    for (int it = 0; it < Cfg::STAGE; ++it) {
      const int idx = t + Cfg::NTHREADS * it;
      v2d v = r[it];
      if constexpr (WEIGHTED) v *= r[Cfg::STAGE + it / 2][it & 1];
      *reinterpret_cast<v2d*>(sm + (idx >> 6) * LDT + 2 * (idx & 63)) = v;
    }
  }
};

template <class Cfg>
struct MMajorPlaneLoader {  // tile of a row-major [M][ld] plane, rows row0 .. row0+127, k contiguous
  const double* base;
  long ld;
  long row0;
  static constexpr int NREG = Cfg::STAGE;
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
  static constexpr int NREG = Cfg::STAGE;
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
// Main loops.  smem must hold 2 x 2 (REAL) or 2 x 4 (CPLX) tiles of TILE_DOUBLES doubles.
//
// Software pipeline of one K slice (4 sub-steps ks of 4 k each), per wave:
//     ks = 0 : MFMA(ks 0)  interleaved with  ds_read frags(ks 1)
//     ks = 1 : MFMA(ks 1)  interleaved with  ds_read frags(ks 2), ds_write slice kt+1 -> other LDS buffer,
//                                            global loads of slice kt+2
//     ks = 2 : MFMA(ks 2)  interleaved with  ds_read frags(ks 3)
//     ---- one workgroup barrier: the other buffer is complete, nobody reads this one any more ----
//     ks = 3 : MFMA(ks 3)  interleaved with  ds_read frags(next slice, ks 0) from the other buffer
// Every MFMA group has its operands in registers one sub-step ahead, there is ONE barrier per slice, and the
// loop body is branch free (the last slices re-load a clamped address and store into the idle buffer) so that
// sched_group_barrier can spread the non-MFMA instructions between the MFMAs: the two waves of a SIMD run in
// lockstep, so any burst of non-MFMA issue in one is a burst in both and leaves the matrix pipe idle
// (profiles/r01_ablation.md).  ABL is a diagnostic switch used only by tools/ablate_gemm.
// ------------------------------------------------------------------------------------------------
enum { ABL_NO_GLOAD = 1, ABL_NO_LDS_STORE = 2, ABL_NO_BARRIER = 4, ABL_NO_FRAG = 8, ABL_NO_INTERLEAVE = 16 };

// sched_group_barrier masks (LLVM AMDGPU): 0x8 MFMA, 0x20 VMEM read, 0x100 DS read, 0x200 DS write.
template <int N_MFMA, int MASK, int REPS>
__device__ __forceinline__ void interleave() {
#pragma unroll
  for (int i = 0; i < REPS; ++i) {
    __builtin_amdgcn_sched_group_barrier(0x008, N_MFMA, 0);
    __builtin_amdgcn_sched_group_barrier(MASK, 1, 0);
  }
}

template <class Cfg, bool A_KMAJOR, class ALoad, class BLoad, int ABL = 0>
__device__ __forceinline__ void mainloop_real(v4d (&acc)[Cfg::MT][Cfg::NTL], const ALoad& la, const BLoad& lb,
                                              long kbegin, int ktiles, double* smem) {
  constexpr int BUF = 2 * TILE_DOUBLES;
  constexpr int KS = BK / 4;
  constexpr int NMFMA = Cfg::MT * Cfg::NTL, NFRAG = Cfg::MT + Cfg::NTL;
  v2d ra[ALoad::NREG], rb[BLoad::NREG];
  if (ktiles <= 0) return;
  la.fetch(kbegin, ra);
  lb.fetch(kbegin, rb);
  ALoad::store(smem, ra);
  BLoad::store(smem + TILE_DOUBLES, rb);
  la.fetch(kbegin + (ktiles > 1 ? BK : 0), ra);
  lb.fetch(kbegin + (ktiles > 1 ? BK : 0), rb);
  __syncthreads();
  double a[2][Cfg::MT], b[2][Cfg::NTL];
#pragma unroll
  for (int i = 0; i < Cfg::MT; ++i) a[0][i] = frag_a<Cfg, A_KMAJOR>(smem, 0, i);
#pragma unroll
  for (int i = 0; i < Cfg::NTL; ++i) b[0][i] = frag_b<Cfg>(smem + TILE_DOUBLES, 0, i);
  for (int kt = 0; kt < ktiles; ++kt) {
    const double* cur = smem + (kt & 1) * BUF;
    double* nx = smem + ((kt + 1) & 1) * BUF;
    const int kt2 = kt + 2 < ktiles ? kt + 2 : ktiles - 1;  // clamped: keeps the body branch free
#pragma unroll
    for (int ks = 0; ks < KS; ++ks) {
      const int c = ks & 1, n = c ^ 1;
      __builtin_amdgcn_sched_barrier(0);
      if (!(ABL & ABL_NO_FRAG)) {
        const double* src = (ks + 1 < KS) ? cur : nx;
        const int kn = (ks + 1 < KS) ? ks + 1 : 0;
#pragma unroll
        for (int i = 0; i < Cfg::MT; ++i) a[n][i] = frag_a<Cfg, A_KMAJOR>(src, kn, i);
#pragma unroll
        for (int i = 0; i < Cfg::NTL; ++i) b[n][i] = frag_b<Cfg>(src + TILE_DOUBLES, kn, i);
      }
      if (ks == 1) {
        if (!(ABL & ABL_NO_LDS_STORE)) {
          ALoad::store(nx, ra);
          BLoad::store(nx + TILE_DOUBLES, rb);
        }
        if (!(ABL & ABL_NO_GLOAD)) {
          la.fetch(kbegin + (long)kt2 * BK, ra);
          lb.fetch(kbegin + (long)kt2 * BK, rb);
        }
      }
#pragma unroll
      for (int mt = 0; mt < Cfg::MT; ++mt)
#pragma unroll
        for (int nt = 0; nt < Cfg::NTL; ++nt)
          acc[mt][nt] = __builtin_amdgcn_mfma_f64_16x16x4f64(a[c][mt], b[c][nt], acc[mt][nt], 0, 0, 0);
      if (!(ABL & ABL_NO_INTERLEAVE)) {
        if (ks == 1) {
          interleave<1, 0x200, 2 * Cfg::STAGE>();
          interleave<1, 0x020, 2 * Cfg::STAGE>();
          interleave<1, 0x100, NFRAG>();
        } else {
          interleave<(NMFMA / NFRAG > 0 ? NMFMA / NFRAG : 1), 0x100, NFRAG>();
        }
      }
      if (ks == KS - 2 && !(ABL & ABL_NO_BARRIER)) {
        __builtin_amdgcn_sched_barrier(0);
        __syncthreads();
      }
    }
  }
}

template <class Cfg, bool A_KMAJOR, class ALoad, class BLoad, int ABL = 0>
__device__ __forceinline__ void mainloop_cplx(v4d (&accR)[Cfg::MT][Cfg::NTL], v4d (&accI)[Cfg::MT][Cfg::NTL],
                                              const ALoad& lac, const ALoad& las, const BLoad& lbr, const BLoad& lbi,
                                              long kbegin, int ktiles, double* smem) {
  constexpr int BUF = 4 * TILE_DOUBLES;
  constexpr int KS = BK / 4;
  constexpr int NMFMA = 4 * Cfg::MT * Cfg::NTL, NFRAG = 2 * (Cfg::MT + Cfg::NTL);
  static_assert(KS == 4, "fragment parity relies on an even number of sub-steps");
  v2d rac[ALoad::NREG], ras[ALoad::NREG], rbr[BLoad::NREG], rbi[BLoad::NREG];
  if (ktiles <= 0) return;
  lac.fetch(kbegin, rac);
  las.fetch(kbegin, ras);
  lbr.fetch(kbegin, rbr);
  lbi.fetch(kbegin, rbi);
  ALoad::store(smem, rac);
  ALoad::store(smem + TILE_DOUBLES, ras);
  BLoad::store(smem + 2 * TILE_DOUBLES, rbr);
  BLoad::store(smem + 3 * TILE_DOUBLES, rbi);
  {
    const long k1 = kbegin + (ktiles > 1 ? BK : 0);
    lac.fetch(k1, rac);
    las.fetch(k1, ras);
    lbr.fetch(k1, rbr);
    lbi.fetch(k1, rbi);
  }
  __syncthreads();
  double ac[2][Cfg::MT], as[2][Cfg::MT], br[2][Cfg::NTL], bi[2][Cfg::NTL];
#pragma unroll
  for (int i = 0; i < Cfg::MT; ++i) {
    ac[0][i] = frag_a<Cfg, A_KMAJOR>(smem, 0, i);
    as[0][i] = frag_a<Cfg, A_KMAJOR>(smem + TILE_DOUBLES, 0, i);
  }
#pragma unroll
  for (int i = 0; i < Cfg::NTL; ++i) {
    br[0][i] = frag_b<Cfg>(smem + 2 * TILE_DOUBLES, 0, i);
    bi[0][i] = frag_b<Cfg>(smem + 3 * TILE_DOUBLES, 0, i);
  }
  for (int kt = 0; kt < ktiles; ++kt) {
    const double* cur = smem + (kt & 1) * BUF;
    double* nx = smem + ((kt + 1) & 1) * BUF;
    const int kt2 = kt + 2 < ktiles ? kt + 2 : ktiles - 1;  // clamped: keeps the body branch free
#pragma unroll
    for (int ks = 0; ks < KS; ++ks) {
      const int c = ks & 1, n = c ^ 1;
      __builtin_amdgcn_sched_barrier(0);
      if (!(ABL & ABL_NO_FRAG)) {
        const double* src = (ks + 1 < KS) ? cur : nx;
        const int kn = (ks + 1 < KS) ? ks + 1 : 0;
#pragma unroll
        for (int i = 0; i < Cfg::MT; ++i) {
          ac[n][i] = frag_a<Cfg, A_KMAJOR>(src, kn, i);
          as[n][i] = frag_a<Cfg, A_KMAJOR>(src + TILE_DOUBLES, kn, i);
        }
#pragma unroll
        for (int i = 0; i < Cfg::NTL; ++i) {
          br[n][i] = frag_b<Cfg>(src + 2 * TILE_DOUBLES, kn, i);
          bi[n][i] = frag_b<Cfg>(src + 3 * TILE_DOUBLES, kn, i);
        }
      }
      if (ks == 1) {
        if (!(ABL & ABL_NO_LDS_STORE)) {
          ALoad::store(nx, rac);
          ALoad::store(nx + TILE_DOUBLES, ras);
          BLoad::store(nx + 2 * TILE_DOUBLES, rbr);
          BLoad::store(nx + 3 * TILE_DOUBLES, rbi);
        }
        if (!(ABL & ABL_NO_GLOAD)) {
          const long k2 = kbegin + (long)kt2 * BK;
          lac.fetch(k2, rac);
          las.fetch(k2, ras);
          lbr.fetch(k2, rbr);
          lbi.fetch(k2, rbi);
        }
      }
#pragma unroll
      for (int mt = 0; mt < Cfg::MT; ++mt) {
        const double an = -as[c][mt];
#pragma unroll
        for (int nt = 0; nt < Cfg::NTL; ++nt) {
          accR[mt][nt] = __builtin_amdgcn_mfma_f64_16x16x4f64(ac[c][mt], br[c][nt], accR[mt][nt], 0, 0, 0);
          accI[mt][nt] = __builtin_amdgcn_mfma_f64_16x16x4f64(ac[c][mt], bi[c][nt], accI[mt][nt], 0, 0, 0);
          accR[mt][nt] = __builtin_amdgcn_mfma_f64_16x16x4f64(as[c][mt], bi[c][nt], accR[mt][nt], 0, 0, 0);
          accI[mt][nt] = __builtin_amdgcn_mfma_f64_16x16x4f64(an, br[c][nt], accI[mt][nt], 0, 0, 0);
        }
      }
      if (!(ABL & ABL_NO_INTERLEAVE)) {
        if (ks == 1) {
          interleave<1, 0x200, 4 * Cfg::STAGE>();
          interleave<1, 0x020, 4 * Cfg::STAGE>();
          interleave<1, 0x100, NFRAG>();
        } else {
          interleave<(NMFMA / NFRAG > 0 ? NMFMA / NFRAG : 1), 0x100, NFRAG>();
        }
      }
      if (ks == KS - 2 && !(ABL & ABL_NO_BARRIER)) {
        __builtin_amdgcn_sched_barrier(0);
        __syncthreads();
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
