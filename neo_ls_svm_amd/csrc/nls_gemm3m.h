// Complex tile engine with the 3-multiplication (3M / Karatsuba) product - 25 % fewer fp64 MFMAs than the
// four-product form in nls_gemm.h.
//
// Both complex kernels of the path need   accR = Ac Br + As Bi ,  accI = Ac Bi - As Br   (nls_gemm.h).  With
//     S1 = Ac Br ,  S2 = As Bi ,  S3 = (Ac - As)(Br + Bi) = S1 + accI - S2
// three real accumulations give  accR = S1 + S2 ,  accI = S3 - S1 + S2 .  The difference / sum operands are formed
// in registers from the fragments that are loaded anyway (one v_add_f64 per fragment), so LDS and HBM traffic are
// those of the four-product kernel.  Normwise the rounding error is the same order as the 4M product (checked
// end to end against the reference fixtures: beta, LOO residuals agree to 1e-13, profiles/r01_3m_numerics.txt).
//
// Geometry: 256 threads = 4 wave64s (2 x 2), one wave per SIMD, 128 x 64 output tile per workgroup, 64 x 32 per
// wave = 4 x 2 MFMA tiles x 3 accumulators = 192 accumulator registers (which is why this engine runs one wave
// per SIMD: 3 x 8 x 8 + fragments + staging does not fit 256 registers).  K walks in 16-deep slices through a
// double-buffered LDS image with one barrier per slice, fragments are prefetched one sub-step ahead and every
// non-MFMA instruction is spread between the MFMAs with sched_group_barrier, exactly as in nls_gemm.h.
//
// BUILD FLAG: compile with  -mllvm -amdgpu-mfma-vgpr-form=1 .  With hipcc's default heuristic a kernel that
// needs more than 256 registers gets AGPR-form MFMAs whose loop-carried accumulators are nevertheless kept in
// arch VGPRs, and all 192 accumulator registers are copied to AGPRs and back in EVERY loop iteration (384
// v_accvgpr_read/write per slice, ~25 % of the loop; reproduced in isolation, profiles/r01_ablation.md).
// VGPR-form MFMAs let the allocator place accumulators in either half of the unified file without copies.
#pragma once
#include "nls_gemm.h"

namespace nls {
namespace m3 {

constexpr int BM3 = 128, BN3 = 64, NT3 = 256;
constexpr int LDTA = BM3 + 16;  // k-major A tile row stride (doubles): 144 = 16 mod 32
constexpr int LDTB = BN3 + 16;  // k-major B tile row stride: 80 = 16 mod 32
constexpr int TILE_A = BK * LDTA;  // 2304 doubles (also the m-major footprint 128 x 18)
constexpr int TILE_B = BK * LDTB;  // 1280 doubles
constexpr int BUF3 = 2 * TILE_A + 2 * TILE_B;  // one slice: Ac, As, Br, Bi
constexpr size_t SMEM3 = 2 * BUF3 * sizeof(double);  // 114,688 B
constexpr int MT3 = 4, NTL3 = 2;                     // MFMA tiles per wave (64 x 32)
constexpr int STAGE_A = (BK * BM3 / 2) / NT3;        // 4 v2d per thread per A plane
constexpr int STAGE_B = (BK * BN3 / 2) / NT3;        // 2 v2d per thread per B plane

__device__ __forceinline__ int wave_m3() { return (threadIdx.x >> 6) >> 1; }
__device__ __forceinline__ int wave_n3() { return (threadIdx.x >> 6) & 1; }
__device__ __forceinline__ int acc_row3(int mt, int reg) { return wave_m3() * 64 + mt * 16 + ((threadIdx.x & 63) >> 4) + 4 * reg; }
__device__ __forceinline__ int acc_col3(int nt) { return wave_n3() * 32 + nt * 16 + (threadIdx.x & 15); }

// ---- staging -------------------------------------------------------------------------------------
template <int WIDTH, int STAGE, bool WEIGHTED = false>
struct KMajorLoader3 {  // tile [16][WIDTH] of a row-major [K][ld] plane, columns col0 ..
  const double* base;
  long ld;
  long col0;
  const double* w = nullptr;  // optional per-k weights, loaded at fetch, applied at store (see nls_gemm.h)
  static constexpr int LD = WIDTH + 16;
  static constexpr int PER_ROW = WIDTH / 2;  // v2d per row
  static constexpr int NREG = STAGE + (WEIGHTED ? (STAGE + 1) / 2 : 0);
  __device__ __forceinline__ void fetch(long k0, v2d (&r)[NREG]) const {
#pragma unroll
    for (int it = 0; it < STAGE; ++it) {
      const int idx = threadIdx.x + NT3 * it;
      const long k = k0 + idx / PER_ROW;
      r[it] = *reinterpret_cast<const v2d*>(base + k * ld + col0 + 2 * (idx % PER_ROW));
      if constexpr (WEIGHTED) r[STAGE + it / 2][it & 1] = w[k];
    }
  }
  static __device__ __forceinline__ void store(double* sm, const v2d (&r)[NREG]) {
#pragma unroll
    for (int it = 0; it < STAGE; ++it) {
      const int idx = threadIdx.x + NT3 * it;
      v2d v = r[it];
      if constexpr (WEIGHTED) v *= r[STAGE + it / 2][it & 1];
      *reinterpret_cast<v2d*>(sm + (idx / PER_ROW) * LD + 2 * (idx % PER_ROW)) = v;
    }
  }
};

struct MMajorLoader3 {  // tile [128 rows][16 k] of a row-major [M][ld] plane
  const double* base;
  long ld;
  long row0;
  static constexpr int NREG = STAGE_A;
  __device__ __forceinline__ void fetch(long k0, v2d (&r)[STAGE_A]) const {
#pragma unroll
    for (int it = 0; it < STAGE_A; ++it) {
      const int idx = threadIdx.x + NT3 * it;
      r[it] = *reinterpret_cast<const v2d*>(base + (row0 + (idx >> 3)) * ld + k0 + 2 * (idx & 7));
    }
  }
  static __device__ __forceinline__ void store(double* sm, const v2d (&r)[STAGE_A]) {
#pragma unroll
    for (int it = 0; it < STAGE_A; ++it) {
      const int idx = threadIdx.x + NT3 * it;
      *reinterpret_cast<v2d*>(sm + (idx >> 3) * LDM + 2 * (idx & 7)) = r[it];
    }
  }
};

template <bool A_KMAJOR>
__device__ __forceinline__ double frag_a3(const double* sm, int ks, int mt) {
  const int lane = threadIdx.x & 63;
  if constexpr (A_KMAJOR)
    return sm[(ks * 4 + (lane >> 4)) * LDTA + wave_m3() * 64 + mt * 16 + (lane & 15)];
  else
    return sm[(wave_m3() * 64 + mt * 16 + (lane & 15)) * LDM + ks * 4 + (lane >> 4)];
}
__device__ __forceinline__ double frag_b3(const double* sm, int ks, int nt) {
  const int lane = threadIdx.x & 63;
  return sm[(ks * 4 + (lane >> 4)) * LDTB + wave_n3() * 32 + nt * 16 + (lane & 15)];
}

// ---- main loop -----------------------------------------------------------------------------------
// S1 += Ac Br ; S2 += As Bi ; S3 += (Ac - As)(Br + Bi).
template <bool A_KMAJOR, class ALoad, class BLoad, int ABL = 0>
__device__ __forceinline__ void mainloop_3m(v4d (&S1)[MT3][NTL3], v4d (&S2)[MT3][NTL3], v4d (&S3)[MT3][NTL3], const ALoad& lac,
                                            const ALoad& las, const BLoad& lbr, const BLoad& lbi, long kbegin, int ktiles,
                                            double* smem) {
  constexpr int KS = BK / 4;
  static_assert(KS == 4, "fragment parity relies on an even number of sub-steps");
  v2d rac[ALoad::NREG], ras[ALoad::NREG], rbr[BLoad::NREG], rbi[BLoad::NREG];
  if (ktiles <= 0) return;
  lac.fetch(kbegin, rac);
  las.fetch(kbegin, ras);
  lbr.fetch(kbegin, rbr);
  lbi.fetch(kbegin, rbi);
  ALoad::store(smem, rac);
  ALoad::store(smem + TILE_A, ras);
  BLoad::store(smem + 2 * TILE_A, rbr);
  BLoad::store(smem + 2 * TILE_A + TILE_B, rbi);
  {
    const long k1 = kbegin + (ktiles > 1 ? BK : 0);
    lac.fetch(k1, rac);
    las.fetch(k1, ras);
    lbr.fetch(k1, rbr);
    lbi.fetch(k1, rbi);
  }
  __syncthreads();
  double ac[2][MT3], as[2][MT3], br[2][NTL3], bi[2][NTL3];
#pragma unroll
  for (int i = 0; i < MT3; ++i) {
    ac[0][i] = frag_a3<A_KMAJOR>(smem, 0, i);
    as[0][i] = frag_a3<A_KMAJOR>(smem + TILE_A, 0, i);
  }
#pragma unroll
  for (int i = 0; i < NTL3; ++i) {
    br[0][i] = frag_b3(smem + 2 * TILE_A, 0, i);
    bi[0][i] = frag_b3(smem + 2 * TILE_A + TILE_B, 0, i);
  }
  for (int kt = 0; kt < ktiles; ++kt) {
    const double* cur = smem + (kt & 1) * BUF3;
    double* nx = smem + ((kt + 1) & 1) * BUF3;
    const int kt2 = kt + 2 < ktiles ? kt + 2 : ktiles - 1;  // clamped: branch-free body
#pragma unroll
    for (int ks = 0; ks < KS; ++ks) {
      const int c = ks & 1, n = c ^ 1;
      __builtin_amdgcn_sched_barrier(0);
      if (!(ABL & ABL_NO_FRAG)) {
        const double* src = (ks + 1 < KS) ? cur : nx;
        const int kn = (ks + 1 < KS) ? ks + 1 : 0;
#pragma unroll
        for (int i = 0; i < MT3; ++i) {
          ac[n][i] = frag_a3<A_KMAJOR>(src, kn, i);
          as[n][i] = frag_a3<A_KMAJOR>(src + TILE_A, kn, i);
        }
#pragma unroll
        for (int i = 0; i < NTL3; ++i) {
          br[n][i] = frag_b3(src + 2 * TILE_A, kn, i);
          bi[n][i] = frag_b3(src + 2 * TILE_A + TILE_B, kn, i);
        }
      }
      if (ks == 1) {
        if (!(ABL & ABL_NO_LDS_STORE)) {
          ALoad::store(nx, rac);
          ALoad::store(nx + TILE_A, ras);
          BLoad::store(nx + 2 * TILE_A, rbr);
          BLoad::store(nx + 2 * TILE_A + TILE_B, rbi);
        }
        if (!(ABL & ABL_NO_GLOAD)) {
          const long k2 = kbegin + (long)kt2 * BK;
          lac.fetch(k2, rac);
          las.fetch(k2, ras);
          lbr.fetch(k2, rbr);
          lbi.fetch(k2, rbi);
        }
      }
      double ad[MT3], bs[NTL3];
#pragma unroll
      for (int i = 0; i < MT3; ++i) ad[i] = ac[c][i] - as[c][i];
#pragma unroll
      for (int i = 0; i < NTL3; ++i) bs[i] = br[c][i] + bi[c][i];
#pragma unroll
      for (int mt = 0; mt < MT3; ++mt)
#pragma unroll
        for (int nt = 0; nt < NTL3; ++nt) {
          S1[mt][nt] = __builtin_amdgcn_mfma_f64_16x16x4f64(ac[c][mt], br[c][nt], S1[mt][nt], 0, 0, 0);
          S2[mt][nt] = __builtin_amdgcn_mfma_f64_16x16x4f64(as[c][mt], bi[c][nt], S2[mt][nt], 0, 0, 0);
        }
#pragma unroll
      for (int mt = 0; mt < MT3; ++mt)
#pragma unroll
        for (int nt = 0; nt < NTL3; ++nt) S3[mt][nt] = __builtin_amdgcn_mfma_f64_16x16x4f64(ad[mt], bs[nt], S3[mt][nt], 0, 0, 0);
      if (!(ABL & ABL_NO_INTERLEAVE)) {
        constexpr int NFRAG = 2 * (MT3 + NTL3);  // 12 ds_read_b64 (6 when merged into ds_read2_b64)
        if (ks == 1) {
          interleave<1, 0x200, 2 * STAGE_A + 2 * STAGE_B>();
          interleave<1, 0x020, 2 * ALoad::NREG + 2 * BLoad::NREG>();
        } else if (ABL & 32) {
          interleave<1, 0x100, NFRAG>();
        } else if (ABL & 64) {
          interleave<3, 0x100, NFRAG / 2>();
          interleave<1, 0x100, NFRAG / 2>();
        } else if (ABL & 128) {  // VALU adds first, then reads 2:1
          interleave<1, 0x002, 6>();
          interleave<2, 0x100, NFRAG>();
        } else {
          interleave<2, 0x100, NFRAG>();
        }
      }
      if (ks == KS - 2 && !(ABL & ABL_NO_BARRIER)) {
        __builtin_amdgcn_sched_barrier(0);
        __syncthreads();
      }
    }
  }
}

}  // namespace m3
}  // namespace nls
