// fp64 MFMA tile engine for gfx950 (MI355X): shared pieces and the REAL product  acc += A B.
// (The complex 3M product of the Gram and the rotation is nls_gemm3m.h.)
//
// A workgroup owns a 128 x 128 output tile and walks K in 16-deep slices.  Operands are staged
// global -> registers -> LDS; the next slice's global loads are issued before the current slice's
// MFMAs so HBM/L2 latency hides under the matrix pipe (v_mfma_f64_16x16x4_f64 occupies its SIMD for 64
// cycles - measured: 78 TFLOP/s chip-wide at 2.39 GHz, profiles/r01_probe_mfma.log).
//
// Wave layout Cfg4: 256 threads, 2 x 2 waves, 64 x 64 per wave (16 accumulator tiles = 128 registers), two
// workgroups per CU.
//
// What the main loops are built around (tools/probe_lds_mfma.hip, profiles/r01_probe_valu_cost.log): between
// fp64 MFMAs of one wave, LDS reads / writes, global loads and SALU cost (almost) nothing, but EVERY VALU
// instruction - a 32-bit address add as much as a v_add_f64 or a v_accvgpr copy - costs ~13.6 matrix-pipe
// cycles.  So the loops carry no vector address arithmetic:
//   global:  address = uniform 64-bit pointer (SGPR pair, advanced with SALU) + one loop-invariant 32-bit
//            per-thread byte offset           -> global_load_dwordx4 v, v_off, s[base:base+1]
//   LDS:     address = one loop-invariant per-thread base per buffer + immediate offsets (two slices per
//            loop trip make the buffer parity a compile-time constant).
//
// LDS images (doubles):
//   k-major tile  T[k][m]  row stride = width + 16 : a wave's fragment read touches rows k, k+1 of
//                 16 consecutive doubles each; a stride = 16 (mod 32) puts the two rows on disjoint halves
//                 of the 64 x 4 B banks -> conflict-free ds_read_b64.
//   m-major tile  T[m][k]  row stride LDM = 16 + 2   : rows m..m+15 at k, k+1 land on 32 distinct
//                 8-byte bank pairs because 18 m mod 32 enumerates the even residues.
// MFMA f64 16x16x4 fragment maps (cdna_hip_programming.md section 3): A[i][k]: lane = 16 k + i,
// B[k][j]: lane = 16 k + j, D[i][j]: lane = 16 (i % 4) + j, reg = i / 4.
#pragma once
#include <hip/hip_runtime.h>
#include <type_traits>
#include <utility>

namespace nls {

typedef double v4d __attribute__((ext_vector_type(4)));
// Native 16-byte vector for staging: HIP's double2 is a struct whose copies lower to memcpy between
// address spaces, which can keep the staging arrays in scratch instead of registers.
typedef double v2d __attribute__((ext_vector_type(2)));
typedef __attribute__((address_space(3))) double lds_f64;
typedef __attribute__((address_space(3))) v2d lds_v2d;

constexpr int BM = 128, BN = 128, BK = 16;
constexpr int LDT = BM + 16;            // k-major tile row stride (128 wide)
constexpr int LDM = BK + 2;             // m-major tile row stride
constexpr int TILE_DOUBLES = BK * LDT;  // == BM * LDM == 2304
static_assert(BK * LDT == BM * LDM, "both LDS images have the same footprint");

template <int... I, class F>
__device__ __forceinline__ void static_for_impl(std::integer_sequence<int, I...>, F&& f) {
  (f(std::integral_constant<int, I>{}), ...);
}
template <int N, class F>
__device__ __forceinline__ void static_for(F&& f) {  // f(integral_constant<int, 0>) ... f(integral_constant<int, N - 1>)
  static_for_impl(std::make_integer_sequence<int, N>{}, static_cast<F&&>(f));
}

// LDS pointer (32-bit) to smem + off doubles, made opaque so that the compiler keeps it in one register instead of
// re-deriving it with VALU adds inside the loop.
__device__ __forceinline__ lds_f64* lds_base(double* smem, int off) {
  lds_f64* p = (lds_f64*)(smem) + off;
  asm volatile("" : "+v"(p));
  return p;
}

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

// ------------------------------------------------------------------------------------------------
// Staging loaders: NT threads move one [16 k] x WIDTH (k-major source) or ROWS x [16 k] (m-major source) tile per
// slice, 16 bytes per thread per pass.  Interface: NREG passes; fetch1(k0, it) -> the 16 bytes of pass it of the
// slice starting at k0; lds_off() -> this thread's offset (doubles) inside the tile's LDS image;
// store1(wr, it, v) with wr = image + lds_off().
// ------------------------------------------------------------------------------------------------
template <int NT, int WIDTH>
struct KMajorLoader {      // tile [16][WIDTH] of a row-major [K][ld] plane, columns col0 ..
  const char* base;        // uniform
  long ldb;                // uniform row pitch in bytes
  mutable unsigned goff;   // per thread
  static constexpr int LD = WIDTH + 16;
  static constexpr int PER_ROW = WIDTH / 2;     // v2d per row
  static constexpr int ROWS_IT = NT / PER_ROW;  // rows covered by one pass of the NT threads
  static constexpr int NREG = BK / ROWS_IT;
  __device__ __forceinline__ KMajorLoader(const double* plane, long ld, long col0)
      : base(reinterpret_cast<const char*>(plane + col0)), ldb(ld * 8),
        goff((unsigned)((threadIdx.x / PER_ROW) * ld * 8 + (threadIdx.x % PER_ROW) * 16)) {}
  __device__ __forceinline__ v2d fetch1(long k0, int it) const {
    asm volatile("" : "+v"(goff));  // keeps the zero-extension next to the load: saddr + 32-bit voffset addressing
    return *reinterpret_cast<const v2d*>(base + (k0 + it * ROWS_IT) * ldb + goff);
  }
  static __device__ __forceinline__ int lds_off() { return (threadIdx.x / PER_ROW) * LD + 2 * (threadIdx.x % PER_ROW); }
  static __device__ __forceinline__ void store1(lds_f64* wr, int it, v2d v) { *(lds_v2d*)(wr + it * ROWS_IT * LD) = v; }
};

template <int NT, int ROWS>
struct MMajorLoader {  // tile [ROWS][16 k] of a row-major [M][ld] plane, rows row0 ..
  const char* base;
  long ldb;
  mutable unsigned goff;
  static constexpr int ROWS_IT = NT / 8;
  static constexpr int NREG = ROWS / ROWS_IT;
  __device__ __forceinline__ MMajorLoader(const double* plane, long ld, long row0)
      : base(reinterpret_cast<const char*>(plane + row0 * ld)), ldb(ld * 8),
        goff((unsigned)((threadIdx.x >> 3) * ld * 8 + (threadIdx.x & 7) * 16)) {}
  __device__ __forceinline__ v2d fetch1(long k0, int it) const {
    asm volatile("" : "+v"(goff));
    return *reinterpret_cast<const v2d*>(base + k0 * 8 + it * ROWS_IT * ldb + goff);
  }
  static __device__ __forceinline__ int lds_off() { return (threadIdx.x >> 3) * LDM + 2 * (threadIdx.x & 7); }
  static __device__ __forceinline__ void store1(lds_f64* wr, int it, v2d v) { *(lds_v2d*)(wr + it * ROWS_IT * LDM) = v; }
};

// ------------------------------------------------------------------------------------------------
// REAL main loop.  smem must hold 2 x 2 tiles of TILE_DOUBLES doubles.
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
// sched_group_barrier can place the memory instructions between the MFMAs (left alone, hipcc sinks the
// ds_reads below the MFMAs and waits for all of them).  ABL is a diagnostic switch used only by tools/ablate_gemm.
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

template <class Cfg, bool A_KMAJOR>
__device__ __forceinline__ int frag_base_a() {
  const int lane = threadIdx.x & 63;
  return A_KMAJOR ? (lane >> 4) * LDT + Cfg::wave_m() * Cfg::WM + (lane & 15)
                  : (Cfg::wave_m() * Cfg::WM + (lane & 15)) * LDM + (lane >> 4);
}
template <class Cfg>
__device__ __forceinline__ int frag_base_b() {
  const int lane = threadIdx.x & 63;
  return (lane >> 4) * LDT + Cfg::wave_n() * Cfg::WN + (lane & 15);
}
template <bool A_KMAJOR>
__device__ __forceinline__ double frag_a(const lds_f64* rd, int ks, int mt) {
  return A_KMAJOR ? rd[ks * 4 * LDT + mt * 16] : rd[mt * 16 * LDM + ks * 4];
}
__device__ __forceinline__ double frag_b(const lds_f64* rd, int ks, int nt) { return rd[ks * 4 * LDT + nt * 16]; }

template <class Cfg, bool A_KMAJOR, class ALoad, class BLoad, int ABL = 0>
__device__ __forceinline__ void mainloop_real(v4d (&acc)[Cfg::MT][Cfg::NTL], const ALoad& la, const BLoad& lb,
                                              long kbegin, int ktiles, double* smem) {
  constexpr int BUF = 2 * TILE_DOUBLES;
  constexpr int KS = BK / 4;
  constexpr int NFRAG = Cfg::MT + Cfg::NTL;
  constexpr int SA = ALoad::NREG, SB = BLoad::NREG;
  static_assert(KS == 4, "fragment parity relies on an even number of sub-steps");
  v2d ra[SA], rb[SB];
  if (ktiles <= 0) return;
  lds_f64* wrA[2] = {lds_base(smem, ALoad::lds_off()), lds_base(smem, BUF + ALoad::lds_off())};
  lds_f64* wrB[2] = {lds_base(smem, TILE_DOUBLES + BLoad::lds_off()), lds_base(smem, BUF + TILE_DOUBLES + BLoad::lds_off())};
  const lds_f64* rdA[2] = {lds_base(smem, frag_base_a<Cfg, A_KMAJOR>()), lds_base(smem, BUF + frag_base_a<Cfg, A_KMAJOR>())};
  const lds_f64* rdB[2] = {lds_base(smem, TILE_DOUBLES + frag_base_b<Cfg>()), lds_base(smem, BUF + TILE_DOUBLES + frag_base_b<Cfg>())};
  auto fetch_all = [&](long k) {
#pragma unroll
    for (int it = 0; it < SA; ++it) ra[it] = la.fetch1(k, it);
#pragma unroll
    for (int it = 0; it < SB; ++it) rb[it] = lb.fetch1(k, it);
  };
  auto store_all = [&](int buf) {
#pragma unroll
    for (int it = 0; it < SA; ++it) ALoad::store1(wrA[buf], it, ra[it]);
#pragma unroll
    for (int it = 0; it < SB; ++it) BLoad::store1(wrB[buf], it, rb[it]);
  };
  fetch_all(kbegin);
  store_all(0);
  fetch_all(kbegin + (ktiles > 1 ? BK : 0));
  __syncthreads();
  double a[2][Cfg::MT], b[2][Cfg::NTL];
#pragma unroll
  for (int i = 0; i < Cfg::MT; ++i) a[0][i] = frag_a<A_KMAJOR>(rdA[0], 0, i);
#pragma unroll
  for (int i = 0; i < Cfg::NTL; ++i) b[0][i] = frag_b(rdB[0], 0, i);
  auto slice = [&](auto parity, int kt) {
    constexpr int P = decltype(parity)::value;
    const int kt2 = kt + 2 < ktiles ? kt + 2 : ktiles - 1;  // clamped: keeps the body branch free
#pragma unroll
    for (int ks = 0; ks < KS; ++ks) {
      const int c = ks & 1, n = c ^ 1;
      __builtin_amdgcn_sched_barrier(0);
      if (!(ABL & ABL_NO_FRAG)) {
        const lds_f64* sa = (ks + 1 < KS) ? rdA[P] : rdA[P ^ 1];
        const lds_f64* sb = (ks + 1 < KS) ? rdB[P] : rdB[P ^ 1];
        const int kn = (ks + 1 < KS) ? ks + 1 : 0;
#pragma unroll
        for (int i = 0; i < Cfg::MT; ++i) a[n][i] = frag_a<A_KMAJOR>(sa, kn, i);
#pragma unroll
        for (int i = 0; i < Cfg::NTL; ++i) b[n][i] = frag_b(sb, kn, i);
      }
      if (ks == 1) {
        if (!(ABL & ABL_NO_LDS_STORE)) store_all(P ^ 1);
        if (!(ABL & ABL_NO_GLOAD)) fetch_all(kbegin + (long)kt2 * BK);
      }
#pragma unroll
      for (int mt = 0; mt < Cfg::MT; ++mt)
#pragma unroll
        for (int nt = 0; nt < Cfg::NTL; ++nt)
          acc[mt][nt] = __builtin_amdgcn_mfma_f64_16x16x4f64(a[c][mt], b[c][nt], acc[mt][nt], 0, 0, 0);
      if (!(ABL & ABL_NO_INTERLEAVE)) {
        // fragment reads first, one per MFMA (the last then has time to land before the next sub-step)
        interleave<1, 0x100, NFRAG>();
        if (ks == 1) {
          interleave<1, 0x200, (SA + SB) / 2>();  // remaining MFMAs: two stores / loads each
          interleave<0, 0x200, (SA + SB) - (SA + SB) / 2>();
          interleave<0, 0x020, SA + SB>();
        }
      }
      if (ks == KS - 2 && !(ABL & ABL_NO_BARRIER)) {
        __builtin_amdgcn_sched_barrier(0);
        __syncthreads();
      }
    }
  };
  int kt = 0;
  for (; kt + 1 < ktiles; kt += 2) {
    slice(std::integral_constant<int, 0>{}, kt);
    slice(std::integral_constant<int, 1>{}, kt + 1);
  }
  if (kt < ktiles) slice(std::integral_constant<int, 0>{}, kt);
}

template <int MT, int NTL>
__device__ __forceinline__ void zero_acc(v4d (&acc)[MT][NTL]) {
#pragma unroll
  for (int i = 0; i < MT; ++i)
#pragma unroll
    for (int j = 0; j < NTL; ++j) acc[i][j] = v4d{0.0, 0.0, 0.0, 0.0};
}

}  // namespace nls
