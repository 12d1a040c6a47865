// Kernels of the primal path.  Stage names follow SURVEY.md section 8(a): P1/P2 feature map (K1),
// P3 Gram (K2), P5 rotation (K4), P6 sweep (K5), P7 selection, P8/P9 re-solve and LOO sigma.
//
// Layout convention: the feature planes hold ONLY the D random features (Kf = ceil(D / 128) * 128 columns).  The
// bias feature phi[:, D] = 1 (_feature_maps.py:202) never enters a matrix tile - D + 1 = 4097 would cost a whole
// extra 128-wide tile row/column (6 % of the Gram and of the rotation at D = 4096).  It is carried as
//   * border vectors of the Gram  (k_border: sum_i conj(F_ij) rs_i, sum_i conj(F_ij) rs_i y_i, 2 scalars),
//   * a rank-1 epilogue term of the rotation  (P_ij += M[D][j]),
//   * a scalar in the prediction GEMV  (+ Re beta[D]).
#pragma once
#include "nls_gemm.h"
#include "nls_gemm3m.h"
#include "nls_sincos.h"
#include "nls_trd.h"

namespace nls {

constexpr int LOO_ROWS_PER_BLOCK = 256;  // upper bound; small problems take fewer rows per block so that the chip is filled

// ------------------------------------------------------------------------------------------------
// K1: feature map.  T = (X - shift) Bs on the matrix pipe (K = d), sincos on the vector pipe in the
// epilogue.  Output either as split planes for the solver (phi = Fc - i Fs, scaled per row):
//     Fc[i][j] = rs_i cos(t_ij)/sqrt(D), Fs[i][j] = rs_i sin(t_ij)/sqrt(D)   for j < D, 0 for D <= j < Kf
// or as interleaved complex128 phi (n x (D+1), column D = 1) for nls_featuremap.
// The solver uses rs_i = s_i / sum(s): the Gram needs S phi, the rotation and the residuals undo the row scale in
// their epilogues (P_i = (F_i Q) / rs_i), so one K1 pass serves all three when the planes of all rows fit in HBM.
// grid = (ceil(cols / 128), rows_pad / 128) with cols = Kf (planes) or D + 1 (complex).
// ------------------------------------------------------------------------------------------------
struct FeatureMapParams {
  const double* X;         // rows x d (pointer to the first row of this chunk)
  const double* Xs;        // rows_pad x dk: (X - shift), zero padded (k_shift_pad) - what the tile loop reads
  const double* shift;     // d
  const double* Bs;        // dk x Kf, B / scale^T zero padded
  const double* rowscale;  // rows (chunk-local) or nullptr (= 1)
  long rows;               // valid rows in this chunk
  int d, dk, D, Kf;
  double inv_sqrt_D;
  double* Fc;  // rows_pad x Kf
  double* Fs;
  double* phi;  // rows x (D+1) complex interleaved (complex variant only)
  int stagger_ticks;  // > 0: the workgroups of the first round start out of phase (see k1_stagger)
  SinCosCoef sc;      // constants of the short sincos (kernel arguments -> SGPR operands, nls_sincos.h)
  SinCosTabCoef tc;   // constants of the table form (what the epilogues run)
  const double* sintab;  // 256 x {S_hi, C_hi, S_lo, C_lo} in global memory; copied into the (then idle) tile LDS before the epilogue
};

// The epilogues' table of (sin, cos)(2 pi n / 256): every thread fetches its 4 entries' worth of doubles BEFORE the main loop (the
// loads fly under it) and drops them over the tile engine's LDS image once the main loop is done with it.
struct SinCosTabRegs {
  double v[4 * SINCOS_TAB_N / Cfg4::NTHREADS];
};
static_assert(4 * SINCOS_TAB_N % Cfg4::NTHREADS == 0, "the table is dealt evenly to the threads");
__device__ __forceinline__ SinCosTabRegs fetch_sincos_table(const FeatureMapParams& p) {
  SinCosTabRegs r;
#pragma unroll
  for (int q = 0; q < 4 * SINCOS_TAB_N / Cfg4::NTHREADS; ++q) r.v[q] = p.sintab ? p.sintab[threadIdx.x + Cfg4::NTHREADS * q] : 0.0;
  return r;
}
__device__ __forceinline__ const double* stage_sincos_table(const FeatureMapParams& p, const SinCosTabRegs& r, double* smem) {
  if (!p.sintab) return nullptr;  // uniform: the polynomial form needs no table
  __syncthreads();  // every wave is out of the main loop: the LDS image is free
#pragma unroll
  for (int q = 0; q < 4 * SINCOS_TAB_N / Cfg4::NTHREADS; ++q) smem[threadIdx.x + Cfg4::NTHREADS * q] = r.v[q];
  __syncthreads();
  return smem;
}

// sin / cos of one accumulator: the polynomial form by default; the table form when the launch handed a table (NLS_K1_SINCOS=table).
// Measured on MI355X (profiles/r03_k1_sincos_table.log): the table form has 19 instead of ~35 vector instructions and a third of the
// error, but its two 16-byte LDS reads at per-lane random indices collide on the banks: K1 26.6 -> 29.1 ms per c3 fit, decision_function
// 49 -> 39 M rows/s.  The polynomial form stays the default.
template <bool TAB>
__device__ __forceinline__ void k1_sincos(double t, double& s, double& c, const FeatureMapParams& p, const double* tab) {
  if constexpr (TAB) sincos_table(t, s, c, p.tc, tab);
  else sincos_reduced_full(t, s, c, p.sc);
}

// The tile kernels of K1 have two phases - matrix pipe (K = d is short), then sincos + 2 x 128 stores per lane - and all
// workgroups take the same time, so the 512 workgroups that start together (2 per CU) stay in lock-step: the whole chip
// stores at once, then not at all.  Delaying the first-round workgroups by different fractions of one tile time spreads
// the store bursts evenly (equal durations keep the phases apart for the rest of the launch).  Ticks of the 100 MHz
// wall clock.
__device__ __forceinline__ void k1_stagger(int ticks) {
  if (ticks <= 0) return;
  const long b = (long)blockIdx.y * gridDim.x + blockIdx.x;
  if (b >= 512) return;
  const int phase = (int)(((b >> 8) * 4 + (b & 3) * 2 + ((b >> 2) & 1)) & 7);  // the two residents of a CU differ by half a period
  const long long wait = (long long)ticks * phase / 8;
  const long long t0 = wall_clock64();
  while (wall_clock64() - t0 < wait) __builtin_amdgcn_s_sleep(8);
}

// One range test per thread for the short sincos: largest |t| over the thread's accumulators <= 2^30, taken on the high words
// as integers (two 32-bit instructions per element; a NaN or Inf compares large and takes the library routine, which handles them).
template <int MT, int NTL>
__device__ __forceinline__ bool all_args_small(const v4d (&acc)[MT][NTL]) {
  unsigned m = 0;
#pragma unroll
  for (int mt = 0; mt < MT; ++mt)
#pragma unroll
    for (int nt = 0; nt < NTL; ++nt)
#pragma unroll
      for (int r = 0; r < 4; ++r) {
        const unsigned h = (unsigned)__double2hiint(acc[mt][nt][r]) & 0x7fffffffu;
        m = h > m ? h : m;
      }
  return m < 0x41d00000u;  // high word of 2^30
}

template <bool COMPLEX_OUT, bool TAB = false>
__global__ void __launch_bounds__(Cfg4::NTHREADS, 1) k_featuremap(FeatureMapParams p) {
  using C = Cfg4;
  extern __shared__ double smem[];
  const long row0 = (long)blockIdx.y * BM;
  const long col0 = (long)blockIdx.x * BN;
  k1_stagger(p.stagger_ticks);
  SinCosTabRegs tabr;
  if constexpr (TAB) tabr = fetch_sincos_table(p);
  v4d acc[C::MT][C::NTL];
  zero_acc(acc);
  if (col0 < p.D) {
    MMajorLoader<C::NTHREADS, BM> la{p.Xs, p.dk, row0};
    KMajorLoader<C::NTHREADS, BN> lb{p.Bs, p.Kf, col0};
    mainloop_real<C, false>(acc, la, lb, 0, p.dk / BK, smem);
  }
  const double* tab = nullptr;
  if constexpr (TAB) tab = stage_sincos_table(p, tabr, smem);
  // One range test per thread: |t| <= 2^30 everywhere (always, in practice) -> the short sincos, else the library's.
  const bool small = all_args_small(acc);
  if constexpr (!COMPLEX_OUT) {
    if (small && col0 + BN <= p.D) {
      // Plane tile without padded columns (every tile when D is a multiple of 128): no per-element column test, one address
      // per accumulator row and plane, the four column groups at immediate offsets.
      const long cbase = col0 + C::acc_col(0);
      // The row scales are fetched for all 16 accumulator rows BEFORE the first store: a load between the stores makes the wave wait for
      // vmcnt(0) - on this chip the counter the stores use too - i.e. drain its eight stores sixteen times per tile (round 5).
      double fs[C::MT][4];
#pragma unroll
      for (int mt = 0; mt < C::MT; ++mt)
#pragma unroll
        for (int r = 0; r < 4; ++r) {
          const long row = row0 + C::acc_row(mt, r);
          fs[mt][r] = row < p.rows ? p.inv_sqrt_D * (p.rowscale ? p.rowscale[row] : 1.0) : 0.0;
        }
#pragma unroll
      for (int mt = 0; mt < C::MT; ++mt)
#pragma unroll
        for (int r = 0; r < 4; ++r) {
          const long row = row0 + C::acc_row(mt, r);
          const double f = fs[mt][r];
          double* pc = p.Fc + row * p.Kf + cbase;
          double* ps = p.Fs + row * p.Kf + cbase;
#pragma unroll
          for (int nt = 0; nt < C::NTL; ++nt) {
            double sv, cv;
            k1_sincos<TAB>(acc[mt][nt][r], sv, cv, p, tab);
            pc[nt * 16] = cv * f;
            ps[nt * 16] = sv * f;
          }
        }
      return;
    }
  }
  double rsv[C::MT][4];  // (fetched before the first store: see above)
#pragma unroll
  for (int mt = 0; mt < C::MT; ++mt)
#pragma unroll
    for (int r = 0; r < 4; ++r) {
      const long row = row0 + C::acc_row(mt, r);
      rsv[mt][r] = row < p.rows ? (p.rowscale ? p.rowscale[row] : 1.0) : 0.0;
    }
  auto epilogue = [&](auto fastc) {
    constexpr bool FAST = decltype(fastc)::value;
#pragma unroll
    for (int mt = 0; mt < C::MT; ++mt)
#pragma unroll
      for (int r = 0; r < 4; ++r) {
        const long row = row0 + C::acc_row(mt, r);
        const bool live = row < p.rows;
        const double rs = rsv[mt][r];
        const double f = p.inv_sqrt_D * rs;
#pragma unroll
        for (int nt = 0; nt < C::NTL; ++nt) {
          const long col = col0 + C::acc_col(nt);
          double c = 0.0, s = 0.0;
          if (col < p.D) {
            double sv, cv;
            if constexpr (FAST) k1_sincos<TAB>(acc[mt][nt][r], sv, cv, p, tab);
            else sincos(acc[mt][nt][r], &sv, &cv);
            c = cv * f;
            s = sv * f;
          } else if (COMPLEX_OUT && col == p.D) {
            c = rs;
          }
          if constexpr (COMPLEX_OUT) {
            if (live && col <= p.D)
              *reinterpret_cast<double2*>(p.phi + 2 * (row * (p.D + 1) + col)) = make_double2(c, -s);
          } else {
            p.Fc[row * p.Kf + col] = c;
            p.Fs[row * p.Kf + col] = s;
          }
        }
      }
  };
  if (small) epilogue(std::true_type{});
  else epilogue(std::false_type{});
}

// ------------------------------------------------------------------------------------------------
// K1 + K8 fused (decision_function, P10): yhat_i = Re(phi_i . beta) without ever writing phi.  Same tile loop as
// k_featuremap; the epilogue multiplies cos / sin by the (1 / sqrt(D))-scaled weight planes and reduces over the
// tile's columns: per accumulator row the 16 lanes of a row group are summed with shuffles, and every wave writes ONE
// partial per row for its 64 columns: part[(2 blockIdx.x + wave_n) * rows_pad + row].  k_gemv_finish adds the
// Kf / 64 partials in a fixed order (+ Re beta[D]): bit-reproducible, and 512 B per row of traffic instead of the
// 2 x 16 D bytes of writing the planes and reading them back (decision_function: 24 -> ~50 M rows/s at D = 4096).
// ------------------------------------------------------------------------------------------------
template <bool TAB = false>
__global__ void __launch_bounds__(Cfg4::NTHREADS, 1) k_featuremap_gemv(FeatureMapParams p, const double* wr, const double* wi, long rows_pad,
                                                                     double* part) {
  using C = Cfg4;
  extern __shared__ double smem[];
  const long row0 = (long)blockIdx.y * BM;
  const long col0 = (long)blockIdx.x * BN;
  SinCosTabRegs tabr;
  if constexpr (TAB) tabr = fetch_sincos_table(p);
  v4d acc[C::MT][C::NTL];
  zero_acc(acc);
  MMajorLoader<C::NTHREADS, BM> la{p.Xs, p.dk, row0};
  KMajorLoader<C::NTHREADS, BN> lb{p.Bs, p.Kf, col0};
  mainloop_real<C, false>(acc, la, lb, 0, p.dk / BK, smem);
  const double* tab = nullptr;
  if constexpr (TAB) tab = stage_sincos_table(p, tabr, smem);
  double br[C::NTL], bi[C::NTL];  // weights of this lane's columns (zero beyond D: padded columns drop out)
#pragma unroll
  for (int nt = 0; nt < C::NTL; ++nt) {
    const long col = col0 + C::acc_col(nt);
    br[nt] = wr[col];
    bi[nt] = wi[col];
  }
  double* out = part + ((long)blockIdx.x * C::WAVES_N + C::wave_n()) * rows_pad;
  // Wave-uniform choice: both branches hold the 16-lane shuffle reduction, and the 16 lanes of a row group own different
  // columns, so a per-thread choice would let a row whose arguments straddle 2^30 split the group across the branches
  // (shuffles would then read inactive lanes).
  const bool small = __all(all_args_small(acc));
  auto epilogue = [&](auto fastc) {
    constexpr bool FAST = decltype(fastc)::value;
#pragma unroll
    for (int mt = 0; mt < C::MT; ++mt)
#pragma unroll
      for (int r = 0; r < 4; ++r) {
        double sum = 0.0;
#pragma unroll
        for (int nt = 0; nt < C::NTL; ++nt) {
          double sv, cv;
          if constexpr (FAST) k1_sincos<TAB>(acc[mt][nt][r], sv, cv, p, tab);
          else sincos(acc[mt][nt][r], &sv, &cv);
          sum += cv * br[nt] + sv * bi[nt];
        }
        sum = trd::row16_sum(sum);  // (the 16 lanes of a row group are a DPP row: four vector instructions instead of four LDS round trips)
        if ((threadIdx.x & 15) == 0) out[row0 + C::acc_row(mt, r)] = sum;
      }
  };
  if (small) epilogue(std::true_type{});
  else epilogue(std::false_type{});
}

// yhat[i] = sum_b part[b][i] + bias  (b in block order)
__global__ void k_gemv_finish(const double* part, int nparts, long rows_pad, long rows, double bias, double* out) {
  const long i = blockIdx.x * (long)blockDim.x + threadIdx.x;
  if (i >= rows) return;
  double v = 0.0;
  for (int b = 0; b < nparts; ++b) v += part[(long)b * rows_pad + i];
  out[i] = v + bias;
}

// wr[j] = Re beta[j] / sqrt(D), wi[j] = Im beta[j] / sqrt(D) for j < D, zero up to Kf:  Re(phi . beta) = cos t . wr + sin t . wi
__global__ void k_gemv_weights(const double2* beta, int D, int Kf, double inv_sqrt_D, double* wr, double* wi) {
  const int j = blockIdx.x * blockDim.x + threadIdx.x;
  if (j >= Kf) return;
  wr[j] = j < D ? beta[j].x * inv_sqrt_D : 0.0;
  wi[j] = j < D ? beta[j].y * inv_sqrt_D : 0.0;
}

// Xs[r][k] = X[r][k] - shift[k] for r < rows, k < d; zero elsewhere (rows_pad x dk).  One cheap pass (16 n d bytes against the
// 16 n D the feature map writes) that lets the K1 tile loop use the plain aligned loader: the guarded, shifting loader
// carried ~30 VALU instructions per slice, and VALU instructions are what the fp64 MFMA pipe pays for (nls_gemm.h).
__global__ void k_shift_pad(const double* X, const double* shift, long rows, int d, long rows_pad, int dk, double* Xs) {
  const long idx = blockIdx.x * (long)blockDim.x + threadIdx.x;
  if (idx >= rows_pad * dk) return;
  const long r = idx / dk;
  const int k = (int)(idx % dk);
  Xs[idx] = (r < rows && k < d) ? X[r * d + k] - shift[k] : 0.0;
}

// Bs[k][j] = B[k][j] / scale[k] for k < d, j < D; zero elsewhere (dk x Kf).
__global__ void k_build_Bs(const double* B, const double* scale, int d, int D, int dk, int Kf, double* Bs) {
  const long idx = blockIdx.x * (long)blockDim.x + threadIdx.x;
  if (idx >= (long)dk * Kf) return;
  const int k = idx / Kf, j = idx % Kf;
  Bs[idx] = (k < d && j < D) ? B[(long)k * D + j] / scale[k] : 0.0;
}

// ------------------------------------------------------------------------------------------------
// XCD-aware tile order (optional, off by default).  Workgroups are dealt round-robin over the 8 XCDs (block b
// runs on XCD b % 8, each with its own 4 MiB L2), and 32 of them are resident per XCD.  Block b is mapped to patch
// (b / 8) / 32 of XCD b % 8 and to position (b / 8) % 32 inside that patch of PR x PC tiles, so the workgroups that
// share an L2 start together on one compact patch and walk K in phase.  Measured on k_rotate3: L2 hit rate
// 0.57 -> 0.78, fabric traffic halved, but 1-4 % slower (profiles/r01_pmc_summary.md): speed only, never
// correctness.  Returns false for the padding blocks of partial patches.
// ------------------------------------------------------------------------------------------------
__device__ __forceinline__ bool xcd_patch_tile(int PR, int PC, long b, long tiles_r, long tiles_c, long& tr, long& tc) {
  const long xcd = b & 7, slot = b >> 3;
  const long patch = (slot / (PR * PC)) * 8 + xcd, within = slot % (PR * PC);
  const long patches_c = (tiles_c + PC - 1) / PC;
  tr = (patch / patches_c) * PR + within / PC;
  tc = (patch % patches_c) * PC + within % PC;
  return tr < tiles_r && tc < tiles_c;
}
static inline long xcd_patch_grid(long tiles_r, long tiles_c, int PR, int PC) {
  const long patches = ((tiles_r + PR - 1) / PR) * ((tiles_c + PC - 1) / PC);
  return ((patches + 7) / 8) * 8 * PR * PC;
}

// ------------------------------------------------------------------------------------------------
// K2: Hermitian Gram A_jk = sum_i conj(F_ij) F_ik of the feature planes, lower block triangle, split over rows,
// 3M complex product (nls_gemm3m.h).  Packed layout: 128 x 128 tile (tj, tk), tk <= tj, at index tj (tj + 1) / 2 + tk,
// each {Re, -Im} x 128 x 128.  A workgroup computes a 128 x 64 half tile: half h = tj (tj + 1) + tk64
// (0 <= tk64 <= 2 tj + 1) is columns (tk64 & 1) * 64 .. +63 of tile (tj, tk64 >> 1).
// grid.x = nt (nt + 1) * nsplit;  slab[(split * ntri + tile)][{R, I}][128][128].
// ------------------------------------------------------------------------------------------------
__global__ void __launch_bounds__(m3::NT3, 1)
    k_gram3(const double* Fc, const double* Fs, int Kf, long rows_pad, int ntri, long rows_per_split, double* slab, long nblocks, int xcd_contig) {
  using namespace m3;
  extern __shared__ double smem[];
  const int nhalf = 2 * ntri;
  // Workgroups are dealt round-robin to the 8 XCDs (each with its own L2).  xcd_contig: XCD x takes the CONTIGUOUS run
  // [x per, (x + 1) per) of the (split, half tile) list, so the ~32 workgroups an XCD runs at a time are neighbours in that list - same
  // row split, same or adjacent tj - and share their A panel (and most B panels) in that L2, instead of every XCD streaming every panel.
  // The grid is padded to a multiple of 8; the order never enters the arithmetic (one slab slot per (split, tile)).
  long lin = blockIdx.x;
  int tj, tk64, split;
  if (xcd_contig == 2) {
    // XCD PATCH order (round 5): the 32 workgroups an XCD runs at a time form a patch of 4 tile rows x 8 half-tile columns of ONE row split:
    // 4 A panels (128 columns each) + 8 B panels (64 each) = 16 panel-units of 64 columns through that L2 for 32 tiles, against 34 in the
    // contiguous order (one tile row, 32 different B panels) and up to 96 in the plain one.  Patches tile the block triangle (patch (a, b), b <= a;
    // the diagonal patches are 20 / 32 full: their padding blocks return at once).  nblocks here = patches x 32.
    // Patches are dealt round-robin to the XCDs, so the list is ordered for equal WORK per XCD: first the diagonal patches of all splits
    // (20 tiles each), then the full ones (32 each) - in (split, a, b) order the diagonal patches pile up on some XCDs (+1.3 % kernel time).
    const int nt = (int)((sqrt(4.0 * nhalf + 1.0) - 1.0) * 0.5 + 0.5);  // nhalf = nt (nt + 1)
    const int pa = (nt + 3) / 4, nfull = pa * (pa - 1) / 2;
    const int ns = (int)((rows_pad + rows_per_split - 1) / rows_per_split);
    const long xcd = blockIdx.x & 7, slot = blockIdx.x >> 3;
    const long patch = (slot / 32) * 8 + xcd;
    const int within = (int)(slot % 32);
    int a, b;
    if (patch < (long)ns * pa) {
      split = (int)(patch / pa);
      a = b = (int)(patch % pa);
    } else {
      const long q = patch - (long)ns * pa;
      if (nfull == 0 || q >= (long)ns * nfull) return;  // padding of the grid
      split = (int)(q / nfull);
      const int f = (int)(q % nfull);  // (a, b), b < a, in row order: f = a (a - 1) / 2 + b
      a = (int)((sqrt(8.0 * f + 1.0) + 1.0) * 0.5);
      while (a * (a + 1) / 2 <= f) ++a;
      while (a * (a - 1) / 2 > f) --a;
      b = f - a * (a - 1) / 2;
    }
    tj = 4 * a + within / 8;
    tk64 = 8 * b + within % 8;
    if (tj >= nt || tk64 > 2 * tj + 1) return;
    lin = 0;
  } else {
    if (xcd_contig) {
      const long per = (nblocks + 7) / 8;
      lin = (long)(blockIdx.x & 7) * per + (blockIdx.x >> 3);
      if (lin >= nblocks) return;
    }
    const int half = (int)(lin % nhalf);
    split = (int)(lin / nhalf);
    tj = (int)((sqrt(4.0 * half + 1.0) - 1.0) * 0.5);
    while ((tj + 1) * (tj + 2) <= half) ++tj;
    while (tj * (tj + 1) > half) --tj;
    tk64 = half - tj * (tj + 1);
  }
  const long r0 = (long)split * rows_per_split;
  long r1 = r0 + rows_per_split;
  if (r1 > rows_pad) r1 = rows_pad;
  acc_zero();
  if (r1 > r0) {
    KMajorLoader3<BM3, STAGE_A> lac{Fc, Kf, (long)tj * BM3}, las{Fs, Kf, (long)tj * BM3};
    KMajorLoader3<BN3, STAGE_B> lbr{Fc, Kf, (long)tk64 * BN3}, lbi{Fs, Kf, (long)tk64 * BN3};
    mainloop_3m<true>(lac, las, lbr, lbi, r0, (int)((r1 - r0) / BK), smem);
  }
  acc_settle();
  const long tile128 = (long)tj * (tj + 1) / 2 + (tk64 >> 1);
  double* outR = slab + ((long)split * ntri + tile128) * (2L * BM * BN) + (tk64 & 1) * BN3;
  double* outI = outR + BM * BN;
  static_for<MT3 * NTL3>([&](auto tc) {
    constexpr int t = decltype(tc)::value, mt = t / NTL3, nt = t % NTL3;
    const v4d S1 = acc_get<ACC_S1 + t>(), S2 = acc_get<ACC_S2 + t>(), S3 = acc_get<ACC_S3 + t>();
#pragma unroll
    for (int r = 0; r < 4; ++r) {
      const int o = acc_row3(mt, r) * BN + acc_col3(nt);
      outR[o] = S1[r] + S2[r];            // Fc_j Fc_k + Fs_j Fs_k =  Re A_jk
      outI[o] = (S3[r] - S1[r]) + S2[r];  // Fc_j Fs_k - Fs_j Fc_k = -Im A_jk
    }
  });
}

// acc[i] += sum_split slab[split][i], fixed order (bit-reproducible).
__global__ void k_gram_reduce(const double* slab, int nsplit, long elems, double* acc) {
  const long idx = blockIdx.x * (long)blockDim.x + threadIdx.x;
  if (idx >= elems) return;
  double v = acc[idx];
  for (int s = 0; s < nsplit; ++s) v += slab[(long)s * elems + idx];
  acc[idx] = v;
}

// Border of the normal equations (bias feature and right-hand side), HBM-bound column sums of the planes:
//   part[split][0][j] = sum_i Fc_ij rs_i        part[split][1][j] = sum_i Fs_ij rs_i          (A[j][D] = [0] + i [1])
//   part[split][2][j] = sum_i Fc_ij rs_i y_i    part[split][3][j] = sum_i Fs_ij rs_i y_i      (b[j]    = [2] + i [3])
//   part[split][4][0] = sum_i rs_i^2 (A[D][D]), part[split][4][1] = sum_i rs_i^2 y_i (b[D])
// width of one split record = 4 Kf + 2.  grid = (Kf / 128, nsplit), 256 threads = 4 row lanes x 64 column pairs.
__global__ void k_border(const double* Fc, const double* Fs, int Kf, const double* rs, const double* y, long rows,
                         long rows_per_split, double* part) {
  __shared__ v2d sh[4][4][64];
  __shared__ double shq[4][2];
  const int cp = threadIdx.x & 63, rl = threadIdx.x >> 6;
  const long col = (long)blockIdx.x * 128 + 2 * cp;
  const long r0 = (long)blockIdx.y * rows_per_split;
  long r1 = r0 + rows_per_split;
  if (r1 > rows) r1 = rows;
  v2d a0 = {0.0, 0.0}, a1 = a0, a2 = a0, a3 = a0;
  double q0 = 0.0, q1 = 0.0;
  for (long i = r0 + rl; i < r1; i += 4) {
    const double w = rs[i], wy = w * y[i];
    const v2d c = *reinterpret_cast<const v2d*>(Fc + i * Kf + col), s = *reinterpret_cast<const v2d*>(Fs + i * Kf + col);
    a0 += c * w;
    a1 += s * w;
    a2 += c * wy;
    a3 += s * wy;
    q0 += w * w;
    q1 += w * wy;
  }
  sh[rl][0][cp] = a0;
  sh[rl][1][cp] = a1;
  sh[rl][2][cp] = a2;
  sh[rl][3][cp] = a3;
  if (cp == 0) {
    shq[rl][0] = q0;
    shq[rl][1] = q1;
  }
  __syncthreads();
  const long width = 4L * Kf + 2;
  double* o = part + (long)blockIdx.y * width;
  if (rl == 0) {
#pragma unroll
    for (int v = 0; v < 4; ++v) {
      const v2d t = (sh[0][v][cp] + sh[1][v][cp]) + (sh[2][v][cp] + sh[3][v][cp]);
      *reinterpret_cast<v2d*>(o + (long)v * Kf + col) = t;
    }
    if (blockIdx.x == 0 && cp == 0) {
      o[4L * Kf] = (shq[0][0] + shq[1][0]) + (shq[2][0] + shq[3][0]);
      o[4L * Kf + 1] = (shq[0][1] + shq[1][1]) + (shq[2][1] + shq[3][1]);
    }
  }
}

// Packed Gram tiles + border -> column-major complex A (D1 x D1, full Hermitian), optionally scaled, and b.
// Element (j, k), j >= k, both < D, of the packed Gram is (R, -I) of tile (j / 128, k / 128).
__device__ __forceinline__ double2 gram_elem(const double* g, int j, int k) {
  const int tj = j >> 7, tk = k >> 7;
  const double* t = g + ((long)tj * (tj + 1) / 2 + tk) * (2L * BM * BN);
  const int o = (j & 127) * BN + (k & 127);
  return make_double2(t[o], -t[BM * BN + o]);
}
__global__ void k_assemble_A(const double* g, const double* border, int D, int Kf, double scale, double2* Acm, long lda,
                             double2* b) {
  const int D1 = D + 1;
  const int j = blockIdx.x * blockDim.x + threadIdx.x;  // row
  const int k = blockIdx.y;                             // column (k == D1: the right-hand side)
  if (j >= D1) return;
  if (k < D1) {
    double2 v;
    if (j == D && k == D) {
      v = make_double2(border[4L * Kf], 0.0);
    } else if (j == D) {  // A[D][k] = conj(A[k][D])
      v = make_double2(border[k], -border[Kf + k]);
    } else if (k == D) {  // A[j][D] = sum_i conj(F_ij) rs_i
      v = make_double2(border[j], border[Kf + j]);
    } else if (j > k) {
      v = gram_elem(g, j, k);
    } else if (j < k) {
      v = gram_elem(g, k, j);
      v.y = -v.y;
    } else {
      v = gram_elem(g, j, j);
      v.y = 0.0;
    }
    Acm[(long)k * lda + j] = make_double2(v.x * scale, v.y * scale);
  } else if (k == D1 && b != nullptr) {
    b[j] = (j == D) ? make_double2(border[4L * Kf + 1], 0.0) : make_double2(border[2L * Kf + j], border[3L * Kf + j]);
  }
}

// Column-major complex (lda) -> row-major complex (D1 x D1), optional conjugate.
__global__ void k_cm_to_rm(const double2* Acm, long lda, int D1, bool conj, double2* out) {
  const int j = blockIdx.x * blockDim.x + threadIdx.x;  // column of out (contiguous)
  const int i = blockIdx.y;
  if (j >= D1) return;
  double2 v = Acm[(long)j * lda + i];
  if (conj) v.y = -v.y;
  out[(long)i * D1 + j] = v;
}

// Rotation matrix M (D1 x D1 complex) -> B-operand planes Mr, Mi [Kf x Np] for the feature rows 0 .. D-1 and the
// bias row D as vectors mbr, mbi [Np]; zero padded.  Element (i, k) is read at M[i * si + k * sk], which covers
// column-major Q (si = 1, sk = ld) and row-major matrices (si = ld, sk = 1); `upper` keeps only i <= k.
// grid = (ceil(Np / 256), Kf + 1): blockIdx.y == Kf is the bias row.
__global__ void k_build_rot_planes(const double2* M, long si, long sk, int D, int Kf, int Np, bool upper, double* Mr, double* Mi,
                                   double* mbr, double* mbi) {
  const int k = blockIdx.x * blockDim.x + threadIdx.x;  // output column
  const int i = blockIdx.y;
  if (k >= Np) return;
  const bool bias = (i == Kf);
  const int row = bias ? D : i;
  double2 v = make_double2(0.0, 0.0);
  if ((bias || i < D) && k <= D && (!upper || row <= k)) v = M[(long)row * si + (long)k * sk];
  if (bias) {
    mbr[k] = v.x;
    mbi[k] = v.y;
  } else {
    Mr[(long)i * Np + k] = v.x;
    Mi[(long)i * Np + k] = v.y;
  }
}

// v_k = (Q^H b)_k * inv_c, one block per k (column k of the column-major Q is contiguous); zero for k >= D1.
__global__ void k_compute_v(const double2* Qcm, long ldq, const double2* b, int D1, double inv_c, double* vr, double* vi) {
  const int k = blockIdx.x;
  __shared__ double sr[256], si[256];
  double ar = 0.0, ai = 0.0;
  if (k < D1) {
    for (int i = threadIdx.x; i < D1; i += blockDim.x) {
      const double2 q = Qcm[(long)k * ldq + i], bb = b[i];
      ar += q.x * bb.x + q.y * bb.y;  // conj(q) * b
      ai += q.x * bb.y - q.y * bb.x;
    }
  }
  sr[threadIdx.x] = ar;
  si[threadIdx.x] = ai;
  __syncthreads();
  for (int s = blockDim.x / 2; s > 0; s >>= 1) {
    if (threadIdx.x < s) {
      sr[threadIdx.x] += sr[threadIdx.x + s];
      si[threadIdx.x] += si[threadIdx.x + s];
    }
    __syncthreads();
  }
  if (threadIdx.x == 0) {
    vr[k] = sr[0] * inv_c;
    vi[k] = si[0] * inv_c;
  }
}

// ------------------------------------------------------------------------------------------------
// K4: rotation P = phi M with the P5 epilogue fused (3M complex product, 128-row x 64-column tiles):
//     P_ij = inv_rs_i sum_{k < D} F_ik M_kj + M_Dj ;   U = Re(P v) ,  Gm = |P|^2 .
// grid.x = tiles_r * (Np / 64) (column tile fastest) or xcd_patch_grid(...) when PR > 0.
// ------------------------------------------------------------------------------------------------
__global__ void __launch_bounds__(m3::NT3, 1)
    k_rotate3(const double* Fc, const double* Fs, int Kf, const double* Mr, const double* Mi, const double* mbr, const double* mbi,
              int Np, const double* vr, const double* vi, double* U, double* Gm, const double* inv_rs, long tiles_r, int PR, int PC,
              int kstagger) {
  using namespace m3;
  extern __shared__ double smem[];
  long tr, tc;
  if (PR > 0) {
    if (!xcd_patch_tile(PR, PC, blockIdx.x, tiles_r, Np / BN3, tr, tc)) return;
  } else {
    tc = blockIdx.x % (Np / BN3);
    tr = blockIdx.x / (Np / BN3);
    if (tr >= tiles_r) return;
  }
  const long row0 = tr * BM3;
  const long col0 = tc * BN3;
  acc_zero();
  MMajorLoader3 lac{Fc, Kf, row0}, las{Fs, Kf, row0};
  KMajorLoader3<BN3, STAGE_B> lbr{Mr, Np, col0}, lbi{Mi, Np, col0};
  // K-walk phase of this workgroup: neighbours in the tile grid (which share an A or a B panel) differ by one slice
  const int koff = kstagger > 0 ? (int)((tr + tc) % kstagger) : 0;
  mainloop_3m<false>(lac, las, lbr, lbi, 0, Kf / BK, smem, koff);
  acc_settle();
  static_for<NTL3>([&](auto ntc) {
    constexpr int nt = decltype(ntc)::value;
    const long col = col0 + acc_col3(nt);
    const double wr = vr[col], wi = vi[col], br = mbr[col], bi = mbi[col];
    static_for<MT3>([&](auto mtc) {
      constexpr int mt = decltype(mtc)::value, t = mt * NTL3 + nt;
      const v4d S1 = acc_get<ACC_S1 + t>(), S2 = acc_get<ACC_S2 + t>(), S3 = acc_get<ACC_S3 + t>();
#pragma unroll
      for (int r = 0; r < 4; ++r) {
        const long row = row0 + acc_row3(mt, r);
        const double f = inv_rs ? inv_rs[row] : 1.0;  // planes hold rs_i phi_i: undo the row scale
        const double pr = (S1[r] + S2[r]) * f + br, pi = ((S3[r] - S1[r]) + S2[r]) * f + bi;
        U[row * Np + col] = pr * wr - pi * wi;
        Gm[row * Np + col] = pr * pr + pi * pi;
      }
    });
  });
}

// R[j][g] = 1 / (gamma_g + lam_j) (zero padded to Np x Gp).
__global__ void k_rgrid(const double* lam, const double* gammas, int D1, int G, int Np, int Gp, double* R) {
  const long idx = blockIdx.x * (long)blockDim.x + threadIdx.x;
  if (idx >= (long)Np * Gp) return;
  const int j = idx / Gp, g = idx % Gp;
  R[idx] = (j < D1 && g < G) ? 1.0 / (gammas[g] + lam[j]) : 0.0;
}

// ------------------------------------------------------------------------------------------------
// K5: sweep GEMMs  num = U R,  hs = (Gm R) / c.   grid = (Gp / 128, rows_pad / 128, 2).
// ------------------------------------------------------------------------------------------------
__global__ void __launch_bounds__(Cfg4::NTHREADS, 2)
    k_sweep(const double* U, const double* Gm, int Np, const double* R, int Gp, double inv_c, double* num, double* hs,
            long out_row0) {
  using C = Cfg4;
  extern __shared__ double smem[];
  const long row0 = (long)blockIdx.y * BM;
  const long col0 = (long)blockIdx.x * BN;
  const bool second = blockIdx.z == 1;
  v4d acc[C::MT][C::NTL];
  zero_acc(acc);
  MMajorLoader<C::NTHREADS, BM> la{second ? Gm : U, Np, row0};
  KMajorLoader<C::NTHREADS, BN> lb{R, Gp, col0};
  mainloop_real<C, false>(acc, la, lb, 0, Np / BK, smem);
  double* out = second ? hs : num;
  const double f = second ? inv_c : 1.0;
#pragma unroll
  for (int mt = 0; mt < C::MT; ++mt)
#pragma unroll
    for (int r = 0; r < 4; ++r) {
      const long row = out_row0 + row0 + C::acc_row(mt, r);
#pragma unroll
      for (int nt = 0; nt < C::NTL; ++nt) out[row * Gp + col0 + C::acc_col(nt)] = acc[mt][nt][r] * f;
    }
}

// ------------------------------------------------------------------------------------------------
// K5 for SHORT gamma grids (G <= 16 NT: 32 or 64 columns; the 32-point grid of the gamma x sigma sweep, BASELINE config 5):
//   num = U R,  hs = (Gm R) / c  with R [Np x ldr] (only its first 16 NT columns are read).
// The 128-wide N tile of the engine above would multiply 96 of its 128 columns by padding.  Here the product is what it is at this shape -
// a stream of U / Gm from HBM (rows x Np x 8 B per product; 2 x 33 GB per 10^6 rows at D = 4096) against a 1.3 MB operand that lives in L2:
//   * no LDS: a wave owns 64 rows x 16 NT columns and walks K alone; no barriers;
//   * A fragments straight from global memory in 32-byte runs: lane (i, kq) loads U[row i][k0 + 4 kq .. + 3] - the four k of one lane feed
//     four successive MFMAs (the contraction does not care in which order k is visited), so a 16-row tile reads whole 128-byte lines;
//   * the matching B fragment of MFMA t is R[k0 + 4 kq + t][col j] - 16 lanes x 8 B contiguous, L1 / L2 hits shared by the waves of a CU;
//   * the next 16-k step's operands are loaded while the current one multiplies.
// grid = (1, ceil(rows_pad / 256), 2): 4 waves per workgroup, blockIdx.z = 0: U -> num, 1: Gm -> hs.
// ------------------------------------------------------------------------------------------------
template <int NT>
__global__ void __launch_bounds__(256) k_sweep_small(const double* U, const double* Gm, int Np, const double* R, int ldr, double inv_c, double* num,
                                                     double* hs, int ldo, long out_row0, long rows_pad) {
  constexpr int MT = 4;
  const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6;
  const int li = lane & 15, kq = lane >> 4;
  const bool second = blockIdx.z == 1;
  const long row0 = ((long)blockIdx.y * 4 + wave) * 64;
  if (row0 >= rows_pad) return;  // (rows_pad is a multiple of 128: the last workgroup may hold two waves only; no barriers in this kernel)
  const double* A = (second ? Gm : U) + (row0 + li) * (long)Np + 4 * kq;  // + mt * 16 * Np + k0
  const double* Bp = R + (long)(4 * kq) * ldr + li;                        // + (k0 + t) * ldr + nt * 16
  v4d acc[MT][NT];
#pragma unroll
  for (int mt = 0; mt < MT; ++mt)
#pragma unroll
    for (int nt = 0; nt < NT; ++nt) acc[mt][nt] = v4d{0.0, 0.0, 0.0, 0.0};
  v2d a0[MT][2], a1[MT][2];
  double b0[4][NT], b1[4][NT];
  auto load = [&](v2d (&a)[MT][2], double (&b)[4][NT], int k0) {
#pragma unroll
    for (int mt = 0; mt < MT; ++mt) {
      const v2d* p = reinterpret_cast<const v2d*>(A + (long)mt * 16 * Np + k0);
      a[mt][0] = p[0];
      a[mt][1] = p[1];
    }
#pragma unroll
    for (int t = 0; t < 4; ++t)
#pragma unroll
      for (int nt = 0; nt < NT; ++nt) b[t][nt] = Bp[(long)(k0 + t) * ldr + nt * 16];
  };
  auto mult = [&](const v2d (&a)[MT][2], const double (&b)[4][NT]) {
#pragma unroll
    for (int t = 0; t < 4; ++t)
#pragma unroll
      for (int mt = 0; mt < MT; ++mt)
#pragma unroll
        for (int nt = 0; nt < NT; ++nt) acc[mt][nt] = __builtin_amdgcn_mfma_f64_16x16x4f64(a[mt][t >> 1][t & 1], b[t][nt], acc[mt][nt], 0, 0, 0);
  };
  load(a0, b0, 0);
  int k0 = 0;
  for (; k0 + 32 <= Np; k0 += 32) {  // two steps per trip: the register sets alternate at compile time
    load(a1, b1, k0 + 16);
    mult(a0, b0);
    if (k0 + 32 < Np) load(a0, b0, k0 + 32);
    mult(a1, b1);
  }
  if (k0 < Np) mult(a0, b0);  // (Np is a multiple of 16: an odd number of steps leaves one)
  double* out = second ? hs : num;
  const double f = second ? inv_c : 1.0;
#pragma unroll
  for (int mt = 0; mt < MT; ++mt)
#pragma unroll
    for (int r = 0; r < 4; ++r) {
      const long row = out_row0 + row0 + mt * 16 + 4 * r + kq;
#pragma unroll
      for (int nt = 0; nt < NT; ++nt) out[row * ldo + nt * 16 + li] = acc[mt][nt][r] * f;
    }
}

// ------------------------------------------------------------------------------------------------
// P6/P7: LOO residuals for every (row, gamma) and their weighted column sums.
//   e = (num - y) / (1 - s^2 hs); classifier: zero on the correct side (_neo_ls_svm.py:153-155)
//   part[blk][0][g] = sum s |e|, [1] = sum s [|e| >= 1], [2] = sum s max(0, |e| - 1)
// grid.x = ceil(n / rows_per_block), block = 256 threads striding over g.
// ------------------------------------------------------------------------------------------------
__global__ void k_loo_errors(const double* num, const double* hs, const double* y, const double* s, long n, int G,
                             int Gp, int is_clf, int rows_per_block, double* part) {
  const long r0 = (long)blockIdx.x * rows_per_block;
  long r1 = r0 + rows_per_block;
  if (r1 > n) r1 = n;
  for (int g = threadIdx.x; g < G; g += blockDim.x) {
    double e0 = 0.0, e1 = 0.0, e2 = 0.0;
    for (long i = r0; i < r1; ++i) {
      const double yi = y[i], si = s[i];
      double e = (num[i * Gp + g] - yi) / (1.0 - si * si * hs[i * Gp + g]);
      if (is_clf) {
        if ((yi > 0 && e > 0) || (yi < 0 && e < 0)) e = 0.0;
      }
      const double ae = fabs(e);
      e0 += si * ae;
      if (is_clf) {
        e1 += (ae >= 1.0) ? si : 0.0;
        e2 += si * fmax(0.0, ae - 1.0);
      }
    }
    double* o = part + (long)blockIdx.x * 3 * Gp;
    o[g] = e0;
    o[Gp + g] = e1;
    o[2 * Gp + g] = e2;
  }
}

// out[idx] = (accumulate ? out[idx] : 0) + sum_blk part[blk][idx].  256 threads = 32 outputs x 8 slices of the block range
// (slice q takes blocks q, q + 8, ...), combined in a fixed order: bit-reproducible, and 8 loads in flight per output
// instead of one dependent chain over all blocks.  Launch with 256 threads and grid.x = ceil(width / 32).
__global__ void __launch_bounds__(256) k_sum_partials(const double* part, long nblk, long width, double* out, int accumulate = 0) {
  __shared__ double sh[8][32];
  const int o = threadIdx.x & 31, q = threadIdx.x >> 5;
  const long idx = (long)blockIdx.x * 32 + o;
  double v = 0.0;
  if (idx < width)
    for (long b = q; b < nblk; b += 8) v += part[b * width + idx];
  sh[q][o] = v;
  __syncthreads();
  if (q == 0 && idx < width) {
    double t = accumulate ? out[idx] : 0.0;
#pragma unroll
    for (int k = 0; k < 8; ++k) t += sh[k][o];
    out[idx] = t;
  }
}

// Column of the selected gamma: P7 / P9 outputs and the score sums.
//   part[blk][0] = clf: sum s [sign(yloo) == y]   reg: sum s (y - yloo)^2
//   part[blk][1] = reg: sum s (y - ybar)^2
// res: residuals_ = Re(phi beta(gamma*)) - y (clipped for classifiers; _neo_ls_svm.py:184-187): num IS Re(phi beta) on the grid, so the selected
// column gives them without another pass over the feature planes.
__global__ void k_loo_column(const double* num, const double* hs, const double* y, const double* s, long n, int Gp,
                             int g, int is_clf, double ybar, double* loo_res, double* loo_lev, double* loo_std,
                             double* res, double* part) {
  __shared__ double s0[256], s1[256];
  const long i = blockIdx.x * (long)blockDim.x + threadIdx.x;
  double a0 = 0.0, a1 = 0.0;
  if (i < n) {
    const double yi = y[i], si = s[i], h = hs[i * Gp + g];
    const double lev = si * si * h;
    const double e_raw = (num[i * Gp + g] - yi) / (1.0 - lev);
    double e = e_raw;
    if (is_clf && ((yi > 0 && e > 0) || (yi < 0 && e < 0))) e = 0.0;
    loo_res[i] = e;
    loo_lev[i] = lev;
    {
      double r = num[i * Gp + g] - yi;
      if (is_clf && ((yi > 0 && r > 0) || (yi < 0 && r < 0))) r = 0.0;
      res[i] = r;
    }
    const double sh = si * h;
    loo_std[i] = sqrt(h + sh * sh / (1.0 - lev));
    const double yl = yi + e_raw;
    if (is_clf) {
      const double sg = (yl > 0.0) ? 1.0 : ((yl < 0.0) ? -1.0 : 0.0);
      a0 = (sg == yi) ? si : 0.0;
    } else {
      a0 = si * e_raw * e_raw;
      a1 = si * (yi - ybar) * (yi - ybar);
    }
  }
  s0[threadIdx.x] = a0;
  s1[threadIdx.x] = a1;
  __syncthreads();
  for (int st = blockDim.x / 2; st > 0; st >>= 1) {
    if (threadIdx.x < st) {
      s0[threadIdx.x] += s0[threadIdx.x + st];
      s1[threadIdx.x] += s1[threadIdx.x + st];
    }
    __syncthreads();
  }
  if (threadIdx.x == 0) {
    part[2L * blockIdx.x] = s0[0];
    part[2L * blockIdx.x + 1] = s1[0];
  }
}

// ------------------------------------------------------------------------------------------------
// K8: yhat_i = Re(phi_i . beta) = inv_rs_i (Fc_i . beta_r + Fs_i . beta_i) + Re beta[D] ; one wave per row.
// out = yhat - y (clipped for classifiers) when y != nullptr, else yhat.
// ------------------------------------------------------------------------------------------------
__global__ void k_plane_gemv(const double* Fc, const double* Fs, int Kf, const double* br, const double* bi, const double2* beta,
                             int D, long rows, const double* y, int is_clf, double* out, const double* inv_rs) {
  const long row = blockIdx.x * (long)(blockDim.x >> 6) + (threadIdx.x >> 6);
  const int lane = threadIdx.x & 63;
  if (row >= rows) return;
  const double* c = Fc + row * Kf;
  const double* s = Fs + row * Kf;
  double acc = 0.0;
  for (int j = 2 * lane; j < Kf; j += 128) {
    const double2 cv = *reinterpret_cast<const double2*>(c + j), sv = *reinterpret_cast<const double2*>(s + j);
    const double2 rv = *reinterpret_cast<const double2*>(br + j), iv = *reinterpret_cast<const double2*>(bi + j);
    acc += cv.x * rv.x + sv.x * iv.x + cv.y * rv.y + sv.y * iv.y;
  }
  for (int o = 32; o > 0; o >>= 1) acc += __shfl_down(acc, o);
  if (lane == 0) {
    if (inv_rs) acc *= inv_rs[row];
    acc += beta[D].x;
    if (y) {
      double e = acc - y[row];
      if (is_clf && ((y[row] > 0 && e > 0) || (y[row] < 0 && e < 0))) e = 0.0;
      out[row] = e;
    } else {
      out[row] = acc;
    }
  }
}

// sigma_i = sqrt(sum_j Gm[i][j]) (predict_std after rotating by U^-1).
__global__ void k_rowsum_sqrt(const double* Gm, int Np, long rows, double* out) {
  const long row = blockIdx.x * (long)(blockDim.x >> 6) + (threadIdx.x >> 6);
  const int lane = threadIdx.x & 63;
  if (row >= rows) return;
  const double* g = Gm + row * Np;
  double acc = 0.0;
  for (int j = lane; j < Np; j += 64) acc += g[j];
  for (int o = 32; o > 0; o >>= 1) acc += __shfl_down(acc, o);
  if (lane == 0) out[row] = sqrt(acc);
}

// Small vector helpers -------------------------------------------------------------------------
// part[blk] = {sum s, sum s*y} over a block of rows.
__global__ void k_weight_sums(const double* s, const double* y, long n, double* part) {
  __shared__ double s0[256], s1[256];
  const long i = blockIdx.x * (long)blockDim.x + threadIdx.x;
  double a0 = 0.0, a1 = 0.0;
  if (i < n) {
    a0 = s[i];
    a1 = s[i] * y[i];
  }
  s0[threadIdx.x] = a0;
  s1[threadIdx.x] = a1;
  __syncthreads();
  for (int st = blockDim.x / 2; st > 0; st >>= 1) {
    if (threadIdx.x < st) {
      s0[threadIdx.x] += s0[threadIdx.x + st];
      s1[threadIdx.x] += s1[threadIdx.x + st];
    }
    __syncthreads();
  }
  if (threadIdx.x == 0) {
    part[2L * blockIdx.x] = s0[0];
    part[2L * blockIdx.x + 1] = s1[0];
  }
}

// Row scale of the feature planes and its inverse.  rs_i = s_i / sum(s) (_neo_ls_svm.py:110-112); a zero weight is
// replaced by 2^-500 so that the row's features survive in the planes (its Gram contribution, ~2^-1000 relative,
// vanishes in rounding exactly as a zero would) and P_i = (F_i Q) / rs_i is recovered exactly (power of two).
__global__ void k_row_scales(const double* s, double inv_sum, long n, double* rs, double* inv_rs) {
  const long i = blockIdx.x * (long)blockDim.x + threadIdx.x;
  if (i < n) {
    double v = s[i] * inv_sum;
    if (!(v > 0.0)) v = 0x1p-500;
    rs[i] = v;
    inv_rs[i] = 1.0 / v;
  }
}

__global__ void k_scale_vec(const double* in, double f, long n, double* out) {
  const long i = blockIdx.x * (long)blockDim.x + threadIdx.x;
  if (i < n) out[i] = in[i] * f;
}

__global__ void k_add_diag(double2* Acm, long lda, int D1, double v) {
  const int i = blockIdx.x * blockDim.x + threadIdx.x;
  if (i < D1) Acm[(long)i * lda + i].x += v;
}

// dst = conj(src) (complex vectors; dst may be src)
__global__ void k_conj_vec(const double2* src, int n, double2* dst) {
  const int i = blockIdx.x * blockDim.x + threadIdx.x;
  if (i < n) dst[i] = make_double2(src[i].x, -src[i].y);
}

// y += alpha x (complex arrays, real alpha)
__global__ void k_axpy_z(const double2* x, double alpha, long n, double2* y) {
  const long i = blockIdx.x * (long)blockDim.x + threadIdx.x;
  if (i < n) {
    y[i].x += alpha * x[i].x;
    y[i].y += alpha * x[i].y;
  }
}

__global__ void k_conj_inplace(double2* a, long n) {
  const long i = blockIdx.x * (long)blockDim.x + threadIdx.x;
  if (i < n) a[i].y = -a[i].y;
}

// v (complex, `len` valid entries) -> planes vr, vi [n_out] zero padded.
// beta = Q (v / (gamma + lam)) from the eigendecomposition the sweep already has (A + gamma c I = c Q (Lam + gamma) Q^H, v = Q^H b / c):
// the re-solve at gamma* needs no factorisation on the critical path (the Cholesky factor is still computed, for L_, beside the residual pass).
// Q column-major (ldq); block (x: 64 rows, y: 256 columns) -> part[y][row]; k_beta_evd_finish adds the column chunks in order.
__global__ void __launch_bounds__(256) k_beta_evd_partial(const double2* Q, long ldq, int D1, const double* vr, const double* vi, const double* lam,
                                                          double gamma, double2* part) {
  __shared__ double2 red[4][64];
  const int row = blockIdx.x * 64 + (threadIdx.x & 63), cgp = threadIdx.x >> 6;
  const int c0 = blockIdx.y * 256 + cgp * 64;
  double ar = 0.0, ai = 0.0;
  if (row < D1) {
    for (int c = c0; c < min(c0 + 64, D1); ++c) {
      const double w = 1.0 / (gamma + lam[c]);
      const double xr = vr[c] * w, xi = vi[c] * w;
      const double2 q = Q[row + (long)c * ldq];
      ar += q.x * xr - q.y * xi;
      ai += q.x * xi + q.y * xr;
    }
  }
  red[cgp][threadIdx.x & 63] = make_double2(ar, ai);
  __syncthreads();
  if (cgp == 0 && row < D1) {
    double2 s = red[0][threadIdx.x];
    for (int g = 1; g < 4; ++g) {
      s.x += red[g][threadIdx.x].x;
      s.y += red[g][threadIdx.x].y;
    }
    part[(long)blockIdx.y * D1 + row] = s;
  }
}
__global__ void k_beta_evd_finish(const double2* part, int nchunks, int D1, double2* beta) {
  const int i = blockIdx.x * blockDim.x + threadIdx.x;
  if (i >= D1) return;
  double2 s = make_double2(0.0, 0.0);
  for (int c = 0; c < nchunks; ++c) {
    s.x += part[(long)c * D1 + i].x;
    s.y += part[(long)c * D1 + i].y;
  }
  beta[i] = s;
}

__global__ void k_split_vec(const double2* v, int len, int n_out, double* vr, double* vi) {
  const int i = blockIdx.x * blockDim.x + threadIdx.x;
  if (i >= n_out) return;
  vr[i] = i < len ? v[i].x : 0.0;
  vi[i] = i < len ? v[i].y : 0.0;
}

// out[i][j] = in[i][j] for j < cols (ld_in -> cols contiguous)
__global__ void k_compact_rows(const double* in, long ld_in, long rows, int cols, double* out) {
  const long idx = blockIdx.x * (long)blockDim.x + threadIdx.x;
  if (idx >= rows * cols) return;
  const long i = idx / cols;
  out[idx] = in[i * ld_in + (idx - i * cols)];
}

}  // namespace nls
