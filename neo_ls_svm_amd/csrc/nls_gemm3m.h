// Complex tile engine with the 3-multiplication (3M / Karatsuba) product - 25 % fewer fp64 MFMAs than the
// textbook four-product form.
//
// Both complex kernels of the path need   accR = Ac Br + As Bi ,  accI = Ac Bi - As Br   .  With
//     S1 = Ac Br ,  S2 = As Bi ,  S3 = (Ac - As)(Br + Bi) = S1 + accI - S2
// three real accumulations give  accR = S1 + S2 ,  accI = S3 - S1 + S2 .  The difference / sum operands are formed
// in registers from the fragments that are loaded anyway (one v_add_f64 per fragment), so LDS and HBM traffic are
// those of the four-product kernel.  Normwise the rounding error is the same order as the 4M product (checked
// end to end against the reference fixtures: beta, LOO residuals agree to 1e-13, profiles/r01b_numerics.txt).
//
// Geometry: 256 threads = 4 wave64s (2 x 2), one wave per SIMD, 128 x 64 output tile per workgroup, 64 x 32 per
// wave = 4 x 2 MFMA tiles x 3 accumulators = 192 accumulator registers, which live in AGPRs a0 .. a191 for the
// whole kernel (see "main loop" below).  K walks in 16-deep slices through a double-buffered LDS image with one
// barrier per slice and fragments are prefetched one sub-step ahead.
//
// Two hardware facts shape the loop (tools/probe_lds_mfma.hip, profiles/r01_probe_valu_cost.log):
//   * the fp64 MFMA shares its datapath with the vector ALU - EVERY VALU instruction a wave issues between
//     MFMAs (a 32-bit address add as much as a v_add_f64 or a v_accvgpr copy) costs ~13.6 matrix-pipe cycles;
//   * LDS reads / writes, global loads and SALU issued between MFMAs are free.
// So the loop is built to contain no vector address arithmetic and no register copies at all.
//
// BUILD FLAG: -mllvm -amdgpu-mfma-vgpr-form=1 is kept for the compiler-scheduled engine in nls_gemm.h.
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

// ---- accumulators in physical AGPRs -----------------------------------------------------------------
// Accumulator tile T (0 .. 23: S1[mt][nt] = mt * 2 + nt, S2 = 8 + ..., S3 = 16 + ...) is a[8 T : 8 T + 7].
// They are addressed by NAME in inline asm rather than passed as operands: as loop-carried SSA values hipcc
// gives them arch-VGPR phis and copies all 192 registers to AGPRs and back on every trip (both with the
// default AGPR-form MFMA selection and with "+a" asm operands), and with VGPR-form MFMAs it spills the staging
// registers through v_accvgpr_write/read instead.  acc_reserve() declares the registers so that the kernel is
// allocated all of them; nothing else in these kernels needs more than ~130 arch VGPRs, so the compiler never
// touches an AGPR itself (checked in the ISA: no 'a' register outside these asm statements).
#define NLS_A8(b) "a" #b "0", "a" #b "1", "a" #b "2", "a" #b "3", "a" #b "4", "a" #b "5", "a" #b "6", "a" #b "7", "a" #b "8", "a" #b "9"
__device__ __forceinline__ void acc_reserve() {
  asm volatile("" ::: "a0", "a1", "a2", "a3", "a4", "a5", "a6", "a7", "a8", "a9", NLS_A8(1), NLS_A8(2), NLS_A8(3), NLS_A8(4), NLS_A8(5),
               NLS_A8(6), NLS_A8(7), NLS_A8(8), NLS_A8(9), NLS_A8(10), NLS_A8(11), NLS_A8(12), NLS_A8(13), NLS_A8(14), NLS_A8(15),
               NLS_A8(16), NLS_A8(17), NLS_A8(18), "a190", "a191");
}
#undef NLS_A8
template <int T>
__device__ __forceinline__ void acc_mfma(double a, double b) {
  asm volatile("v_mfma_f64_16x16x4_f64 a[%2:%3], %0, %1, a[%2:%3]" : : "v"(a), "v"(b), "n"(8 * T), "n"(8 * T + 7));
}
__device__ __forceinline__ void acc_zero() {
  static_for<192>([](auto r) { asm volatile("v_accvgpr_write_b32 a[%0], 0" : : "n"(decltype(r)::value)); });
}
// Call once after the last MFMA and before the first acc_get: inline-asm MFMAs get no hazard padding from the
// compiler, and a 16-pass DGEMM result needs 19 wait states before it may be read.
__device__ __forceinline__ void acc_settle() { asm volatile("s_nop 15\n\ts_nop 15" ::: "memory"); }
template <int T>
__device__ __forceinline__ v4d acc_get() {
  v4d out;
  static_for<4>([&](auto e) {
    constexpr int E = decltype(e)::value;
    unsigned lo, hi;
    asm volatile("v_accvgpr_read_b32 %0, a[%1]" : "=v"(lo) : "n"(8 * T + 2 * E));
    asm volatile("v_accvgpr_read_b32 %0, a[%1]" : "=v"(hi) : "n"(8 * T + 2 * E + 1));
    out[E] = __hiloint2double((int)hi, (int)lo);
  });
  return out;
}
constexpr int ACC_S1 = 0, ACC_S2 = MT3 * NTL3, ACC_S3 = 2 * MT3 * NTL3;

// ---- staging: the shared VALU-free loaders (nls_gemm.h) at this engine's geometry ----------------------------
template <int WIDTH, int STAGE>
using KMajorLoader3 = KMajorLoader<NT3, WIDTH>;
using MMajorLoader3 = MMajorLoader<NT3, BM3>;
static_assert(KMajorLoader3<BM3, STAGE_A>::NREG == STAGE_A && KMajorLoader3<BN3, STAGE_B>::NREG == STAGE_B &&
                  MMajorLoader3::NREG == STAGE_A && KMajorLoader3<BM3, 0>::LD == LDTA && KMajorLoader3<BN3, 0>::LD == LDTB,
              "staging plan");

// Fragment reads relative to a per-lane base (frag_base_*): only immediate offsets.
template <bool A_KMAJOR>
__device__ __forceinline__ int frag_base_a3() {
  const int lane = threadIdx.x & 63;
  return A_KMAJOR ? (lane >> 4) * LDTA + wave_m3() * 64 + (lane & 15) : (wave_m3() * 64 + (lane & 15)) * LDM + (lane >> 4);
}
__device__ __forceinline__ int frag_base_b3() {
  const int lane = threadIdx.x & 63;
  return (lane >> 4) * LDTB + wave_n3() * 32 + (lane & 15);
}
template <bool A_KMAJOR>
__device__ __forceinline__ double frag_a3(const lds_f64* rd, int ks, int mt) {
  return A_KMAJOR ? rd[ks * 4 * LDTA + mt * 16] : rd[mt * 16 * LDM + ks * 4];
}
__device__ __forceinline__ double frag_b3(const lds_f64* rd, int ks, int nt) { return rd[ks * 4 * LDTB + nt * 16]; }

// ---- main loop -----------------------------------------------------------------------------------
// S1 += Ac Br ; S2 += As Bi ; S3 += (Ac - As)(Br + Bi), accumulators in a0 .. a191 (acc_zero() first; acc_settle() and
// acc_get<T>() after).  Inline-asm MFMAs are invisible to sched_group_barrier, so the instruction order is written
// out in source order and pinned with sched_barrier(0): after MFMA j of a sub-step come that sub-step's side
// operations (fragment reads of the next sub-step; at ks == 1 also the LDS stores of slice kt + 1 and the global
// loads of slice kt + 2).  Two slices per trip so that the LDS buffer parity is a compile-time
// constant and every LDS address is a loop-invariant register plus an immediate.
// koff (in slices, 0 <= koff < ktiles): the walk starts at slice koff and wraps around - workgroups that share an operand
// panel in one L2 can be taken out of phase (one leads and misses, the others hit) without changing what is summed,
// only the order (the slice index is SALU arithmetic: free).
template <bool A_KMAJOR, class ALoad, class BLoad, int ABL = 0>
__device__ __forceinline__ void mainloop_3m(const ALoad& lac, const ALoad& las, const BLoad& lbr, const BLoad& lbi, long kbegin,
                                            int ktiles, double* smem, int koff = 0) {
  constexpr int KS = BK / 4;
  static_assert(KS == 4, "fragment parity relies on an even number of sub-steps");
  static_assert(ALoad::NREG == STAGE_A && BLoad::NREG == STAGE_B, "staging plan");
  constexpr int NMFMA = 3 * MT3 * NTL3;            // 24 per sub-step
  constexpr int NFRAG = 2 * (MT3 + NTL3);          // 12 fragment reads per sub-step
  constexpr int NSTAGE = 2 * (STAGE_A + STAGE_B);  // 12 staging registers (v2d)
  acc_reserve();
  v2d rac[STAGE_A], ras[STAGE_A], rbr[STAGE_B], rbi[STAGE_B];
  if (ktiles <= 0) return;
  // loop-invariant per-thread LDS bases: [buffer]
  lds_f64* wrA[2] = {lds_base(smem, ALoad::lds_off()), lds_base(smem, BUF3 + ALoad::lds_off())};
  lds_f64* wrB[2] = {lds_base(smem, 2 * TILE_A + BLoad::lds_off()), lds_base(smem, BUF3 + 2 * TILE_A + BLoad::lds_off())};
  const lds_f64* rdA[2] = {lds_base(smem, frag_base_a3<A_KMAJOR>()), lds_base(smem, BUF3 + frag_base_a3<A_KMAJOR>())};
  const lds_f64* rdB[2] = {lds_base(smem, 2 * TILE_A + frag_base_b3()), lds_base(smem, BUF3 + 2 * TILE_A + frag_base_b3())};
  auto fetch_all = [&](long k) {
#pragma unroll
    for (int it = 0; it < STAGE_A; ++it) {
      rac[it] = lac.fetch1(k, it);
      ras[it] = las.fetch1(k, it);
    }
#pragma unroll
    for (int it = 0; it < STAGE_B; ++it) {
      rbr[it] = lbr.fetch1(k, it);
      rbi[it] = lbi.fetch1(k, it);
    }
  };
  auto kpos = [&](int kt) -> long {  // global k of slice number kt of the walk (uniform -> scalar ALU)
    int w = kt + koff;
    w = w >= ktiles ? w - ktiles : w;
    return kbegin + (long)w * BK;
  };
  fetch_all(kpos(0));
#pragma unroll
  for (int it = 0; it < STAGE_A; ++it) {
    ALoad::store1(wrA[0], it, rac[it]);
    ALoad::store1(wrA[0] + TILE_A, it, ras[it]);
  }
#pragma unroll
  for (int it = 0; it < STAGE_B; ++it) {
    BLoad::store1(wrB[0], it, rbr[it]);
    BLoad::store1(wrB[0] + TILE_B, it, rbi[it]);
  }
  fetch_all(kpos(ktiles > 1 ? 1 : 0));
  __syncthreads();
  double ac[2][MT3], as[2][MT3], br[2][NTL3], bi[2][NTL3];
#pragma unroll
  for (int i = 0; i < MT3; ++i) {
    ac[0][i] = frag_a3<A_KMAJOR>(rdA[0], 0, i);
    as[0][i] = frag_a3<A_KMAJOR>(rdA[0] + TILE_A, 0, i);
  }
#pragma unroll
  for (int i = 0; i < NTL3; ++i) {
    br[0][i] = frag_b3(rdB[0], 0, i);
    bi[0][i] = frag_b3(rdB[0] + TILE_B, 0, i);
  }
  // one 16-deep slice; P = parity of the LDS buffer that holds it
  auto slice = [&](auto parity, int kt) {
    constexpr int P = decltype(parity)::value;
    const int kt2 = kt + 2 < ktiles ? kt + 2 : ktiles - 1;  // clamped: branch-free body
    const long k2 = kpos(kt2);
    static_for<KS>([&](auto ksc) {
      constexpr int ks = decltype(ksc)::value;
      constexpr int c = ks & 1, n = c ^ 1;
      const lds_f64* sa = (ks + 1 < KS) ? rdA[P] : rdA[P ^ 1];
      const lds_f64* sb = (ks + 1 < KS) ? rdB[P] : rdB[P ^ 1];
      constexpr int kn = (ks + 1 < KS) ? ks + 1 : 0;
      auto side = [&](int idx) {  // side operation number idx of this sub-step (idx is a constant after unrolling)
        if (idx < NFRAG) {
          if (ABL & ABL_NO_FRAG) return;
          if (idx < MT3) ac[n][idx] = frag_a3<A_KMAJOR>(sa, kn, idx);
          else if (idx < 2 * MT3) as[n][idx - MT3] = frag_a3<A_KMAJOR>(sa + TILE_A, kn, idx - MT3);
          else if (idx < 2 * MT3 + NTL3) br[n][idx - 2 * MT3] = frag_b3(sb, kn, idx - 2 * MT3);
          else bi[n][idx - 2 * MT3 - NTL3] = frag_b3(sb + TILE_B, kn, idx - 2 * MT3 - NTL3);
          return;
        }
        const int r = (idx - NFRAG) >> 1;     // staging register: rac[0..3], ras[0..3], rbr[0..1], rbi[0..1]
        const bool load = (idx - NFRAG) & 1;  // store it to LDS first, then refill it from global memory
        if (!load && !(ABL & ABL_NO_LDS_STORE)) {
          if (r < STAGE_A) ALoad::store1(wrA[P ^ 1], r, rac[r]);
          else if (r < 2 * STAGE_A) ALoad::store1(wrA[P ^ 1] + TILE_A, r - STAGE_A, ras[r - STAGE_A]);
          else if (r < 2 * STAGE_A + STAGE_B) BLoad::store1(wrB[P ^ 1], r - 2 * STAGE_A, rbr[r - 2 * STAGE_A]);
          else BLoad::store1(wrB[P ^ 1] + TILE_B, r - 2 * STAGE_A - STAGE_B, rbi[r - 2 * STAGE_A - STAGE_B]);
        }
        if (load && !(ABL & ABL_NO_GLOAD)) {
          if (r < STAGE_A) rac[r] = lac.fetch1(k2, r);
          else if (r < 2 * STAGE_A) ras[r - STAGE_A] = las.fetch1(k2, r - STAGE_A);
          else if (r < 2 * STAGE_A + STAGE_B) rbr[r - 2 * STAGE_A] = lbr.fetch1(k2, r - 2 * STAGE_A);
          else rbi[r - 2 * STAGE_A - STAGE_B] = lbi.fetch1(k2, r - 2 * STAGE_A - STAGE_B);
        }
      };
      constexpr int nside = NFRAG + (ks == 1 ? 2 * NSTAGE : 0);
      __builtin_amdgcn_sched_barrier(0);
      double ad[MT3], bs[NTL3];
#pragma unroll
      for (int i = 0; i < MT3; ++i) ad[i] = ac[c][i] - as[c][i];
#pragma unroll
      for (int i = 0; i < NTL3; ++i) bs[i] = br[c][i] + bi[c][i];
      __builtin_amdgcn_sched_barrier(0);
      static_for<NMFMA>([&](auto jc) {
        constexpr int j = decltype(jc)::value;
        if constexpr (j < 2 * MT3 * NTL3) {
          constexpr int t = j >> 1, mt = t / NTL3, nt = t % NTL3;
          if constexpr (j & 1) acc_mfma<ACC_S2 + t>(as[c][mt], bi[c][nt]);
          else acc_mfma<ACC_S1 + t>(ac[c][mt], br[c][nt]);
        } else {
          constexpr int t = j - 2 * MT3 * NTL3, mt = t / NTL3, nt = t % NTL3;
          acc_mfma<ACC_S3 + t>(ad[mt], bs[nt]);
        }
        if (!(ABL & ABL_NO_INTERLEAVE)) {
          // fragment reads one per MFMA at the front of the sub-step (the last one then has 12 MFMAs to land
          // before the next sub-step needs it); staging traffic, two per MFMA, behind them.
          if (j < NFRAG) side(j);
          if (ks == 1 && j >= NMFMA - NSTAGE) {
            side(NFRAG + 2 * (j - (NMFMA - NSTAGE)));
            side(NFRAG + 2 * (j - (NMFMA - NSTAGE)) + 1);
          }
          __builtin_amdgcn_sched_barrier(0);
        }
      });
      if (ABL & ABL_NO_INTERLEAVE) {
#pragma unroll
        for (int idx = 0; idx < nside; ++idx) side(idx);
      }
      if (ks == KS - 2 && !(ABL & ABL_NO_BARRIER)) {
        __builtin_amdgcn_sched_barrier(0);
        __syncthreads();
      }
    });
  };
  int kt = 0;
  for (; kt + 1 < ktiles; kt += 2) {
    slice(std::integral_constant<int, 0>{}, kt);
    slice(std::integral_constant<int, 1>{}, kt + 1);
  }
  if (kt < ktiles) slice(std::integral_constant<int, 0>{}, kt);
}

}  // namespace m3
}  // namespace nls
