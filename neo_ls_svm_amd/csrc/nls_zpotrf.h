// Cholesky factorisation A = L L^H of a complex Hermitian positive definite matrix (lower triangle, column-major interleaved complex,
// in place) - the L_ output of the primal fit (cho_factor(gamma* C + A), _neo_ls_svm.py:176-177).
//
// Why not rocsolver_zpotrf: D + 1 = 4097 takes 16.5 ms (18 ms in block columns of 512 on rocBLAS ztrsm / zherk) and 1025 takes 4 ms
// (245 potf2_kernel_small launches) for 1.2 ms / 0.02 ms of arithmetic: a chain of small dependent kernels.  Here, right-looking in panels of
// NBZ = 32 columns (everything here is latency bound: narrow panels of light kernels), TWO launches per panel, and two levels of blocking so that
// the trailing matrix is not streamed once per narrow panel (n^3 / 6 bytes at NBZ = 32: 11.5 GB at n = 4097): a panel's update reaches only to the
// end of its OUTER block column of NBO = 256 columns; the rest of the trailing matrix is updated once per outer block with K = 256
// (the real counterpart is nls_potrf.h):
//   k_zpotrf_panel: every workgroup factors the 32 x 32 diagonal block in LDS (redundantly: cheaper than a kernel boundary) and then solves its
//                   256 rows of L21 = A21 L11^-H by forward substitution along the row, one row per thread, the row in LDS.  Besides L21 (in
//                   place) the kernel writes three real planes of it - Re, Im, -Im as [k][row] - which are the k-major operands of
//   k_zpotrf_herk : A22 -= L21 L21^H on the lower 128 x 128 tiles through the real tile engine (nls_gemm.h), one workgroup per (tile, part):
//                   Re = Lr Lr^T + Li Li^T, Im = Li Lr^T + Lr (-Li)^T - two K = 32 passes of mainloop_real into one accumulator set each.
// A pivot <= 0 (or NaN) raises info = its 1-based index, as LAPACK does; the factorisation carries on with garbage (finite control flow).
#pragma once
#include "nls_gemm.h"
#include "nls_potrf.h"

namespace nls {
namespace zpotrf {

constexpr int NBZ = 32;   // panel width = leaf size
constexpr int NBO = 256;  // outer block column: the trailing matrix beyond it is updated once per NBO columns

// One launch per panel for the diagonal block AND the rows below it.  Every workgroup (256 threads = 256 rows of the panel) first factors the
// 32 x 32 diagonal block itself, in LDS - redundantly: a few microseconds of work against a kernel boundary (the panel is a chain of dependent
// launches, and a launch costs more than the block) - then solves its rows: L21 = A21 L11^-H by forward substitution along the row,
// x[c] = (a[c] - sum_{t < c} x[t] conj(L[c][t])) / L[c][c], one row per thread, the row in LDS ([t][thread]: conflict-free), the factor's
// entries the same for every lane (broadcast reads).  The block is factored by WAVE 0 alone (lane = (row, column parity): each half of the wave
// owns the columns of one parity; LDS operations of one wave complete in order, so its 32 steps need no workgroup barrier) while all waves'
// row loads are in flight.
// Workgroup 0 hands L11 over (strict upper part untouched; see L11out below) and raises info (0 or the global 1-based index of the first bad pivot).
// With `rhs_run` the right-hand side of beta = cho_solve(L, b) is carried along (the forward substitution of _neo_ls_svm.py:178 without a
// separate pass over L, and without reading finished columns of L, which the download may already be conjugating): rhs_run holds the running
// right-hand side b - L[:, :k0] y[:k0]; wave 0 of every workgroup solves the panel's 32 unknowns y[k0 .. k0 + w) against the diagonal block
// (workgroup 0 stores them in ysol), and every row thread takes its row's share L21[r, :] y off rhs_run[r].
// Out besides L21 (in place): the planes of L21 as the update kernel reads them, [k][row] with the row index fastest, zero for k >= w and for rows
// m .. m_pad - 1, stacked in pairs along k (each half padded to NBZ rows): S1 = [Re; Im], S2 = [Re; -Im], S3 = [Im; Re]; and the same entries in
// the outer block column's stacks O1 .. O3 (halves of ohalf rows; rows below the outer block only, index r - orow0; zeroed by the caller).
constexpr int ZP_ROWS = 256;                // rows of the panel per workgroup
constexpr int ZP_THREADS = ZP_ROWS + 64;    // + the wave that factors the diagonal block
constexpr size_t ZP_LDS = (size_t)2 * NBZ * (NBZ + 1) * sizeof(double) + (size_t)NBZ * ZP_ROWS * sizeof(double2) + NBZ * sizeof(double);
__global__ void __launch_bounds__(ZP_THREADS) k_zpotrf_panel(double2* __restrict__ A, long lda, int w, int k0, int m, int m_pad, double* __restrict__ S1,
                                                          double* __restrict__ S2, double* __restrict__ S3, long ldp, double* __restrict__ O1,
                                                          double* __restrict__ O2, double* __restrict__ O3, long ohalf, int orow0,
                                                          double2* __restrict__ L11out, double2* __restrict__ rhs_run, double2* __restrict__ ysol, int* info,
                                                          long long* stamps /* diagnostic (NLS_ZPOTRF_STAMP=1; nullptr: none): workgroup 0's time line, 100 MHz ticks */) {
  extern __shared__ __attribute__((aligned(16))) unsigned char zp_smem[];
  double(*Lr)[NBZ + 1] = reinterpret_cast<double(*)[NBZ + 1]>(zp_smem);
  double(*Li)[NBZ + 1] = reinterpret_cast<double(*)[NBZ + 1]>(zp_smem + (size_t)NBZ * (NBZ + 1) * sizeof(double));
  double2(*xs)[ZP_ROWS] = reinterpret_cast<double2(*)[ZP_ROWS]>(zp_smem + (size_t)2 * NBZ * (NBZ + 1) * sizeof(double));
  double* dv = reinterpret_cast<double*>(zp_smem + (size_t)2 * NBZ * (NBZ + 1) * sizeof(double) + (size_t)NBZ * ZP_ROWS * sizeof(double2));
  __shared__ double2 ysh[NBZ];
  __shared__ int cols_done;  // columns of the diagonal block's factor that are final in Lr / Li / dv (written by wave 0, polled by the row waves)
  // Round 4: wave 0 only factors the diagonal block and PUBLISHES each finished column; the four row waves (thread = row) trail it by one
  // column - step c of a row needs L[c][0 .. c], final after the block's step c - so the two chains of 32 dependent steps overlap instead of
  // following each other (80 -> ~ 50 us per panel).
  const int ftid = threadIdx.x;          // 0 .. 63: the factoring wave
  const int tid = (int)threadIdx.x - 64;  // >= 0: row thread
  const bool rowthr = tid >= 0;
  const long r = (long)blockIdx.x * ZP_ROWS + tid;
  const bool live = rowthr && r < m;
  double2* A21 = A + w;
  if (ftid == 0) cols_done = 0;
  const bool st = stamps != nullptr && blockIdx.x == 0;
  if (st && ftid == 0) stamps[0] = wall_clock64();
  // ---- this thread's row of the panel: loads in flight while the block is factored ----
  // ---- the diagonal block (rows / columns k0 .. k0 + w of A), identity-padded to 32 x 32: requested first, stored last - its round trip
  // runs beside the row panel's ----
  constexpr int NIT = (NBZ * NBZ + ZP_THREADS - 1) / ZP_THREADS;
  double2 dv4[NIT];
#pragma unroll
  for (int it = 0; it < NIT; ++it) {
    const int idx = ftid + it * ZP_THREADS, rr = idx % NBZ, c = idx / NBZ;
    dv4[it] = make_double2(rr == c ? 1.0 : 0.0, 0.0);
    if (idx < NBZ * NBZ && rr < w && c < w && rr >= c) dv4[it] = A[rr + (long)c * lda];
  }
  // (sixteen columns are requested together and then stored: written as one predicated load and LDS store per column the loop is compiled into
  // load - wait - store, 32 dependent memory round trips: the 10 us this prologue took in the round-5 time line)
  if (rowthr) {
#pragma unroll
    for (int c0 = 0; c0 < NBZ; c0 += 16) {
      double2 xr[16];
#pragma unroll
      for (int c = 0; c < 16; ++c) xr[c] = (live && c0 + c < w) ? A21[r + (long)(c0 + c) * lda] : make_double2(0.0, 0.0);
#pragma unroll
      for (int c = 0; c < 16; ++c) xs[c0 + c][tid] = xr[c];
    }
  }
#pragma unroll
  for (int it = 0; it < NIT; ++it) {
    const int idx = ftid + it * ZP_THREADS, rr = idx % NBZ, c = idx / NBZ;
    if (idx < NBZ * NBZ) {
      Lr[rr][c] = dv4[it].x;
      Li[rr][c] = rr == c ? 0.0 : dv4[it].y;  // the diagonal of a Hermitian matrix is real (LAPACK ignores its imaginary part too)
    }
  }
  __syncthreads();
  if (st && ftid == 0) stamps[1] = wall_clock64();  // diagonal block in LDS
  int bad = 0;
  if (!rowthr) {
    // Wave 0: lane = (row rr, half h).  Round 5: the two halves SHARE the row's columns - half h keeps and updates the columns of parity h
    // (ao_r / ao_i[q] = entry (rr, 2 q + h)) - instead of mirroring each other: 16 instead of 31 column updates per step and lane, and the
    // pivot column's entries L[c][k] come from the LDS copy the step has just published (two broadcast reads per column, off the VALU) instead
    // of four v_readlane per column.  The pivot column's own entries travel from its half to both with v_permlane32_swap.  Arithmetic per
    // element is unchanged (same products, same order): the factor is bit-identical to the mirrored form's.
    const int rr = ftid & (NBZ - 1), h = ftid >> 5;
    // No lane predicates inside the factorisation: entries above the diagonal are computed as garbage and never read; half h's update of a
    // column <= k (the pivot column itself, once per even step) lands in a register nobody reads again.
    double ao_r[NBZ / 2], ao_i[NBZ / 2];
#pragma unroll
    for (int q = 0; q < NBZ / 2; ++q) {
      ao_r[q] = Lr[rr][2 * q + h];
      ao_i[q] = Li[rr][2 * q + h];
    }
    // v_permlane32_swap exchanges the upper 32 lanes of one operand with the lower 32 of the other; fed the same register twice, one element
    // of the result carries the lower half's values in all 64 lanes and the other the upper half's.  Which is which is read off once (a
    // uniform value), so the code does not depend on the operand order of the instruction.
    const bool e0_upper = __builtin_amdgcn_readfirstlane((int)__builtin_amdgcn_permlane32_swap(h, h, false, false)[0]) != 0;
    auto from_half = [e0_upper](double v, int hk) -> double {  // the value lane (rr, hk) holds, in both halves
      const auto lo = __builtin_amdgcn_permlane32_swap(__double2loint(v), __double2loint(v), false, false);
      const auto hi = __builtin_amdgcn_permlane32_swap(__double2hiint(v), __double2hiint(v), false, false);
      const bool first = e0_upper == (hk != 0);  // uniform
      return __hiloint2double(first ? (int)hi[0] : (int)hi[1], first ? (int)lo[0] : (int)lo[1]);
    };
#pragma unroll
    for (int k = 0; k < NBZ; ++k) {
      const int hk = k & 1, qk = k >> 1;
      double d = potrf::readlane_f64(ao_r[qk], k + 32 * hk);
      if (!(d > 0.0) || !isfinite(d)) {  // uniform
        if (bad == 0) bad = k + 1;
        d = 1.0;
      }
      const double inv = sb::fast_rsqrt(d);  // (v_rsq_f64 + one Newton step of third order: within an ulp or two of 1 / sqrt(d), a third of the dependent instructions)
      const double lr = from_half(ao_r[qk] * inv, hk), li = from_half(ao_i[qk] * inv, hk);
      // column k is final: publish it (rows above the diagonal hold garbage that nobody reads), then the count
      dv[k] = inv;
      Lr[rr][k] = lr;
      Li[rr][k] = li;
      __builtin_amdgcn_fence(__ATOMIC_RELEASE, "wavefront");
      __builtin_amdgcn_wave_barrier();
      __hip_atomic_store(&cols_done, k + 1, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_WORKGROUP);
#pragma unroll
      for (int q = (k + 1) / 2; q < NBZ / 2; ++q) {  // a[rr][c] -= l[rr] conj(l[c]),  c = 2 q + h   (meaningful for rr >= c > k)
        const double cr = Lr[2 * q + h][k], ci = Li[2 * q + h][k];  // (this wave's own stores: LDS operations of one wave complete in order)
        ao_r[q] -= lr * cr + li * ci;
        // two ROUNDED products, not a fused pair: on the diagonal (rr == c) they are the same product and must cancel exactly - the diagonal
        // of the factor is real (HIP's __dmul_rn is a plain multiplication and gets contracted; the empty asm pins the rounding)
        double p1 = li * cr, p2 = lr * ci;
        asm volatile("" : "+v"(p1), "+v"(p2));
        ao_i[q] -= p1 - p2;
      }
      __builtin_amdgcn_sched_barrier(0);
    }
    // (every column went to Lr / Li as it was finished; the diagonal's imaginary part stays exactly zero: see above)
    if (st && ftid == 0) stamps[2] = wall_clock64();  // block factored
    if (rhs_run) {  // y[k0 .. k0 + w): L11 y = (running right-hand side), forward substitution across the lanes
      double2 acc = (h == 0 && rr < w) ? rhs_run[k0 + rr] : make_double2(0.0, 0.0);
      for (int t = 0; t < NBZ; ++t) {
        if (ftid == t) ysh[t] = make_double2(acc.x * dv[t], acc.y * dv[t]);
        __builtin_amdgcn_fence(__ATOMIC_RELEASE, "wavefront");
        __builtin_amdgcn_wave_barrier();
        if (h == 0 && rr > t) {
          const double2 yt = ysh[t];
          const double lr = Lr[rr][t], li = Li[rr][t];
          acc.x -= lr * yt.x - li * yt.y;
          acc.y -= lr * yt.y + li * yt.x;
        }
      }
      if (blockIdx.x == 0 && h == 0 && rr < w) ysol[k0 + rr] = ysh[rr];
    }
    if (st && ftid == 0) stamps[3] = wall_clock64();  // the panel's unknowns of the carried substitution solved
  }
  if (rowthr) {
    // ---- this workgroup's rows of the panel below the block: column c as soon as the block's column c is final ----
    for (int c = 0; c < NBZ; ++c) {
      while (__hip_atomic_load(&cols_done, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_WORKGROUP) <= c) __builtin_amdgcn_s_sleep(1);
      __builtin_amdgcn_fence(__ATOMIC_ACQUIRE, "workgroup");
      double2 x = xs[c][tid];
      double sr = x.x, si = x.y;
#pragma unroll 4
      for (int t = 0; t < c; ++t) {  // x[c] -= x[t] conj(L[c][t])
        const double lr = Lr[c][t], li = Li[c][t];
        x = xs[t][tid];
        sr = fma(-x.x, lr, sr);
        sr = fma(-x.y, li, sr);
        si = fma(-x.y, lr, si);
        si = fma(x.x, li, si);
      }
      xs[c][tid] = make_double2(sr * dv[c], si * dv[c]);  // (read back by this thread only: no barrier)
    }
    if (st && tid == 0) stamps[4] = wall_clock64();  // this workgroup's rows solved
  }
  __syncthreads();  // the factor, ysh and every row are complete
  if (st && ftid == 0) stamps[5] = wall_clock64();
  if (blockIdx.x == 0) {
    // L11 goes back into A directly only when no other workgroup exists that may still be reading the un-factored block; otherwise into
    // L11out, from where the update kernel (the next launch) puts it in place.
    double2* dst = gridDim.x == 1 ? A : L11out;
    const long ldd = gridDim.x == 1 ? lda : NBZ;
    for (int idx = ftid; idx < NBZ * NBZ; idx += ZP_THREADS) {
      const int rr = idx % NBZ, c = idx / NBZ;
      if (rr < w && c <= rr) dst[rr + (long)c * ldd] = make_double2(Lr[rr][c], Li[rr][c]);
    }
    if (bad != 0 && bad <= w && ftid == 0) atomicCAS(info, 0, k0 + bad);  // (pivots of the identity padding beyond w cannot fail)
  }
  if (!rowthr || r >= m_pad) return;
  if (rhs_run && live) {  // this row's share of the forward substitution: b[r] -= L21[r, :] y
    double2 acc = rhs_run[k0 + w + r];
    for (int c = 0; c < w; ++c) {
      const double2 x = xs[c][tid], yv = ysh[c];
      acc.x -= x.x * yv.x - x.y * yv.y;
      acc.y -= x.x * yv.y + x.y * yv.x;
    }
    rhs_run[k0 + w + r] = acc;
  }
  const long half = (long)NBZ * ldp;
  for (int c = 0; c < NBZ; ++c) {
    const bool keep = live && c < w;
    const double2 x = keep ? xs[c][tid] : make_double2(0.0, 0.0);
    if (keep) A21[r + (long)c * lda] = x;
    const long o = (long)c * ldp + r;
    S1[o] = x.x;
    S1[half + o] = x.y;
    S2[o] = x.x;
    S2[half + o] = -x.y;
    S3[o] = x.y;
    S3[half + o] = x.x;
    if (keep && r >= orow0) {  // (O1 .. O3 already point at this panel's 32 k-rows of the first halves)
      const long oo = (long)c * ldp + (r - orow0), oh = ohalf * ldp;
      O1[oo] = x.x;
      O1[oh + oo] = x.y;
      O2[oo] = x.x;
      O2[oh + oo] = -x.y;
      O3[oo] = x.y;
      O3[oh + oo] = x.x;
    }
  }
  if (st && tid == 0) {
    asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
    stamps[6] = wall_clock64();  // stores of the panel and its planes drained
  }
}

// The factored diagonal block put in place when no update kernel follows the panel kernel (last panel of an outer block column).
__global__ void __launch_bounds__(256) k_zpotrf_putback(const double2* L11src, double2* L11dst, long lda, int w) {
  for (int idx = threadIdx.x; idx < NBZ * NBZ; idx += 256) {
    const int r = idx % NBZ, c = idx / NBZ;
    if (r < w && c <= r) L11dst[r + (long)c * lda] = L11src[r + c * NBZ];
  }
}

// A22 -= L21 L21^H on the lower triangle: 128 x 128 tiles (R, C), R >= C, C < ncol_tiles, of the m x m matrix A22 (interleaved complex, leading
// dimension lda; only columns < ncols are touched) on the real tile engine.  blockIdx.y = 0: real part, 1: imaginary part.  The tile is formed
// transposed (A operand: the tile's columns, B operand: its rows), as in k_potrf_syrk.  With a = L[r][k], b = L[c][k]:
//   Re (L L^H)[r][c] = sum_k ar br + ai bi  = [Re; Im] . [Re; Im]      Im = sum_k ai br - ar bi = [Re; -Im](c) . [Im; Re](r)
// i.e. ONE product over the stacked planes S1 x S1 or S2 x S3 (k_zpotrf_panel), K = 2 x the half height: one main loop, one accumulator set
// (two passes over separate planes kept both loaders' state alive and spilled 145 registers).  Planes: [2 ktiles_half * 16][ldp], readable
// (zero) up to a multiple of 128 rows.  ncol_tiles < the number of row tiles gives the tall update inside an outer block column.
__global__ void __launch_bounds__(Cfg4::NTHREADS, 2) k_zpotrf_herk(double2* A, long lda, int m, int ncols, int ncol_tiles, int ktiles, const double* S1,
                                                                   const double* S2, const double* S3, long ldp, const double2* L11src, double2* L11dst, int w) {
  using C4 = Cfg4;
  extern __shared__ double smem[];
  if (L11src && blockIdx.x == 0 && blockIdx.y == 0) {  // the panel kernel's factored diagonal block, put in place (k_zpotrf_panel)
    for (int idx = threadIdx.x; idx < NBZ * NBZ; idx += C4::NTHREADS) {
      const int r = idx % NBZ, c = idx / NBZ;
      if (r < w && c <= r) L11dst[r + (long)c * lda] = L11src[r + c * NBZ];
    }
  }
  // tile list: rows R = 0 .. nt - 1, for each the columns C = 0 .. min(R, ncol_tiles - 1): the first ncol_tiles rows form a triangle
  int t = blockIdx.x, R, C;
  const int tri = ncol_tiles * (ncol_tiles + 1) / 2;
  if (t < tri) {
    R = (int)((sqrt(8.0 * t + 1.0) - 1.0) * 0.5);
    while ((R + 1) * (R + 2) / 2 <= t) ++R;
    while (R * (R + 1) / 2 > t) --R;
    C = t - R * (R + 1) / 2;
  } else {
    R = ncol_tiles + (t - tri) / ncol_tiles;
    C = (t - tri) % ncol_tiles;
  }
  const int part = blockIdx.y;
  v4d acc[C4::MT][C4::NTL];
  zero_acc(acc);
  KMajorLoader<C4::NTHREADS, BM> la{part == 0 ? S1 : S2, ldp, (long)C * BM};
  KMajorLoader<C4::NTHREADS, BN> lb{part == 0 ? S1 : S3, ldp, (long)R * BN};
  mainloop_real<C4, true>(acc, la, lb, 0, ktiles, smem);
  double* Ad = reinterpret_cast<double*>(A) + part;
  // read-modify-write: 16 entries loaded together, then stored (A[..] -= acc entry by entry is one memory round trip per entry: nls_potrf.h);
  // tiles inside the matrix and off the diagonal without per-entry tests
  auto rmw = [&](auto plainc) {
    constexpr bool PLAIN = decltype(plainc)::value;
#pragma unroll
    for (int mt = 0; mt < C4::MT; ++mt) {
      double old[4][C4::NTL];
#pragma unroll
      for (int reg = 0; reg < 4; ++reg) {
        const long cc = (long)C * BM + C4::acc_row(mt, reg);
#pragma unroll
        for (int nt = 0; nt < C4::NTL; ++nt) {
          const long r = (long)R * BN + C4::acc_col(nt);
          old[reg][nt] = (PLAIN || (r < m && cc < ncols && (part == 0 ? r >= cc : r > cc))) ? Ad[2 * (r + cc * lda)] : 0.0;
        }
      }
#pragma unroll
      for (int reg = 0; reg < 4; ++reg) {
        const long cc = (long)C * BM + C4::acc_row(mt, reg);
#pragma unroll
        for (int nt = 0; nt < C4::NTL; ++nt) {
          const long r = (long)R * BN + C4::acc_col(nt);
          if (PLAIN || (r < m && cc < ncols && (part == 0 ? r >= cc : r > cc))) Ad[2 * (r + cc * lda)] = old[reg][nt] - acc[mt][nt][reg];
        }
      }
    }
  };
  if (R != C && ((long)R + 1) * BN <= m && ((long)C + 1) * BM <= ncols)
    rmw(std::true_type{});
  else
    rmw(std::false_type{});
}

// beta of L^H beta = y, in place in y, where the array holds Lc = conj(L) (column-major lower: what the download's conjugation leaves behind,
// i.e. scipy's upper factor read column-major): Lc^T beta = y.  Backwards in outer blocks of NBO columns, two launches per block:
//   k_ztrsv_outer_sum: s[c] = sum_{r >= K0 + W} Lc[r][c] beta[r] for the block's columns - one workgroup per column (a column is contiguous);
//   k_ztrsv_block    : ONE workgroup finishes the block: panels of 32 columns backwards, per panel the 256 threads form the sums over the
//                      block's later rows and thread 0 solves the 32 x 32 triangle.
// (rocblas_ztrsv: 2.2 ms per solve at n = 4097 - 134 MB of matrix at 60 GB/s.)
__global__ void __launch_bounds__(256) k_ztrsv_outer_sum(const double2* __restrict__ Lc, long lda, int n, int K0, int W, const double2* __restrict__ y,
                                                         double2* __restrict__ sums) {
  __shared__ double2 red[4];
  const int c = K0 + blockIdx.x;
  double sr = 0.0, si = 0.0;
  {
    const double2* col = Lc + (long)c * lda;
    double sr1 = 0.0, si1 = 0.0;
    int r = K0 + W + threadIdx.x;
    for (; r + 768 < n; r += 1024) {  // four products in flight per thread
      const double2 l0 = col[r], l1 = col[r + 256], l2 = col[r + 512], l3 = col[r + 768];
      const double2 b0 = y[r], b1 = y[r + 256], b2 = y[r + 512], b3 = y[r + 768];
      sr += l0.x * b0.x - l0.y * b0.y;
      si += l0.x * b0.y + l0.y * b0.x;
      sr1 += l1.x * b1.x - l1.y * b1.y;
      si1 += l1.x * b1.y + l1.y * b1.x;
      sr += l2.x * b2.x - l2.y * b2.y;
      si += l2.x * b2.y + l2.y * b2.x;
      sr1 += l3.x * b3.x - l3.y * b3.y;
      si1 += l3.x * b3.y + l3.y * b3.x;
    }
    for (; r < n; r += 256) {
      const double2 l = col[r], b = y[r];
      sr += l.x * b.x - l.y * b.y;
      si += l.x * b.y + l.y * b.x;
    }
    sr += sr1;
    si += si1;
  }
#pragma unroll
  for (int o = 32; o > 0; o >>= 1) {
    sr += __shfl_xor(sr, o, 64);
    si += __shfl_xor(si, o, 64);
  }
  if ((threadIdx.x & 63) == 0) red[threadIdx.x >> 6] = make_double2(sr, si);
  __syncthreads();
  if (threadIdx.x == 0) sums[blockIdx.x] = make_double2((red[0].x + red[1].x) + (red[2].x + red[3].x), (red[0].y + red[1].y) + (red[2].y + red[3].y));
}
__global__ void __launch_bounds__(256) k_ztrsv_block(const double2* __restrict__ Lc, long lda, int K0, int W, double2* __restrict__ y,
                                                     const double2* __restrict__ sums) {
  __shared__ double2 part[8][NBZ];
  __shared__ double2 blk[NBZ][NBZ + 1];
  __shared__ double2 sol[NBZ];
  const int tid = threadIdx.x, c = tid & (NBZ - 1), g = tid >> 5;
  const int bend = K0 + W, np = (W + NBZ - 1) / NBZ;
  for (int p = np - 1; p >= 0; --p) {
    const int k0 = K0 + p * NBZ, w = min(NBZ, bend - k0), rend = k0 + w;
    // (four products in flight per thread and the block's entries requested together: one load per loop trip is one memory round trip per trip)
    double sr = 0.0, si = 0.0;
    if (c < w) {
      const double2* col = Lc + (long)(k0 + c) * lda;
      double sr1 = 0.0, si1 = 0.0;
      int r = rend + g;
      for (; r + 24 < bend; r += 32) {
        const double2 l0 = col[r], l1 = col[r + 8], l2 = col[r + 16], l3 = col[r + 24];
        const double2 b0 = y[r], b1 = y[r + 8], b2 = y[r + 16], b3 = y[r + 24];
        sr += l0.x * b0.x - l0.y * b0.y;
        si += l0.x * b0.y + l0.y * b0.x;
        sr1 += l1.x * b1.x - l1.y * b1.y;
        si1 += l1.x * b1.y + l1.y * b1.x;
        sr += l2.x * b2.x - l2.y * b2.y;
        si += l2.x * b2.y + l2.y * b2.x;
        sr1 += l3.x * b3.x - l3.y * b3.y;
        si1 += l3.x * b3.y + l3.y * b3.x;
      }
      for (; r < bend; r += 8) {
        const double2 l = col[r], b = y[r];
        sr += l.x * b.x - l.y * b.y;
        si += l.x * b.y + l.y * b.x;
      }
      sr += sr1;
      si += si1;
    }
    part[g][c] = make_double2(sr, si);
    {
      double2 bv[NBZ * NBZ / 256];
#pragma unroll
      for (int it = 0; it < NBZ * NBZ / 256; ++it) {
        const int idx = tid + 256 * it, rr = idx % NBZ, cc = idx / NBZ;
        bv[it] = (rr < w && cc < w && rr >= cc) ? Lc[(long)(k0 + rr) + (long)(k0 + cc) * lda] : make_double2(rr == cc ? 1.0 : 0.0, 0.0);
      }
#pragma unroll
      for (int it = 0; it < NBZ * NBZ / 256; ++it) {
        const int idx = tid + 256 * it;
        blk[idx % NBZ][idx / NBZ] = bv[it];
      }
    }
    __syncthreads();
    if (tid < 64) {  // wave 0: lane cc holds its unknown's right-hand side; 32 steps of backward substitution across the lanes
      const int cc = tid & (NBZ - 1);
      double2 acc = make_double2(0.0, 0.0);
      if (tid < w) {
        acc = y[k0 + cc];
        if (sums) {
          acc.x -= sums[k0 - K0 + cc].x;
          acc.y -= sums[k0 - K0 + cc].y;
        }
        for (int q = 0; q < 8; ++q) {
          acc.x -= part[q][cc].x;
          acc.y -= part[q][cc].y;
        }
      }
      const double dinv = 1.0 / blk[cc][cc].x;  // the diagonal of a Cholesky factor is real; (one division per unknown, all at once, not one per step)
      for (int t = NBZ - 1; t >= 0; --t) {
        if (tid == t) sol[t] = make_double2(acc.x * dinv, acc.y * dinv);
        __builtin_amdgcn_fence(__ATOMIC_RELEASE, "wavefront");
        __builtin_amdgcn_wave_barrier();
        if (tid < t) {  // unknown cc < t: take Lc[t][cc] beta[t] off
          const double2 l = blk[t][cc], b = sol[t];
          acc.x -= l.x * b.x - l.y * b.y;
          acc.y -= l.x * b.y + l.y * b.x;
        }
      }
      if (tid < w) y[k0 + cc] = sol[cc];
    }
    __threadfence_block();
    __syncthreads();
  }
}

}  // namespace zpotrf
}  // namespace nls
