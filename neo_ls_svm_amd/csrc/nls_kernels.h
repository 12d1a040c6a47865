// Kernels of the primal path.  Stage names follow SURVEY.md section 8(a): P1/P2 feature map (K1),
// P3 Gram (K2), P5 rotation (K4), P6 sweep (K5), P7 selection, P8/P9 re-solve and LOO sigma.
#pragma once
#include "nls_gemm.h"
#include "nls_gemm3m.h"

namespace nls {

constexpr int LOO_ROWS_PER_BLOCK = 256;

// ------------------------------------------------------------------------------------------------
// K1: feature map.  T = (X - shift) Bs on the matrix pipe (K = d), sincos on the vector pipe in the
// epilogue.  Output either as split planes for the solver:
//     Fc[i][j] = rs_i cos(t_ij)/sqrt(D), Fs[i][j] = rs_i sin(t_ij)/sqrt(D)      (phi = Fc - i Fs)
//     column D  : Fc = rs_i (bias feature, _feature_maps.py:202), Fs = 0
//     column D+1: Fc = rs_i y_i when `target` is given (so the Gram's last row is b), else 0.  The solver uses
//     rs_i = s_i / sum(s) and target = y for every phase: the Gram needs S phi, the rotation and the residuals undo
//     the row scale in their epilogues (P_i = (F_i Q) / rs_i), so one K1 pass serves all three when the planes of
//     all rows fit in HBM.
// or as interleaved complex128 phi for nls_featuremap.
// grid = (Kp / 128, rows_pad / 128).
// ------------------------------------------------------------------------------------------------
struct FeatureMapParams {
  const double* X;      // n_total_rows x d (pointer to the first row of this chunk)
  const double* shift;  // d
  const double* Bs;     // dk x Kp, B / scale^T zero padded
  const double* rowscale;  // rows (chunk-local) or nullptr (= 1)
  const double* target;    // rows (chunk-local) or nullptr
  long rows;               // valid rows in this chunk
  int d, dk, D, Kp;
  double inv_sqrt_D;
  double* Fc;  // rows_pad x Kp
  double* Fs;
  double* phi;  // rows x (D+1) complex interleaved (complex variant only)
};

template <bool COMPLEX_OUT>
__global__ void __launch_bounds__(Cfg4::NTHREADS, 1) k_featuremap(FeatureMapParams p) {
  using C = Cfg4;
  extern __shared__ double smem[];
  const long row0 = (long)blockIdx.y * BM;
  const long col0 = (long)blockIdx.x * BN;
  v4d acc[C::MT][C::NTL];
  zero_acc(acc);
  if (col0 < p.D) {
    XShiftLoader<C> la{p.X, p.shift, p.rows, p.d, row0};
    KMajorPlaneLoader<C> lb{p.Bs, p.Kp, col0};
    mainloop_real<C, false>(acc, la, lb, 0, p.dk / BK, smem);
  }
#pragma unroll
  for (int mt = 0; mt < C::MT; ++mt)
#pragma unroll
    for (int r = 0; r < 4; ++r) {
      const long row = row0 + C::acc_row(mt, r);
      const bool live = row < p.rows;
      const double rs = live ? (p.rowscale ? p.rowscale[row] : 1.0) : 0.0;
      const double tg = (live && p.target) ? p.target[row] : 0.0;
#pragma unroll
      for (int nt = 0; nt < C::NTL; ++nt) {
        const long col = col0 + C::acc_col(nt);
        double c = 0.0, s = 0.0;
        if (col < p.D) {
          double sv, cv;
          sincos(acc[mt][nt][r], &sv, &cv);
          c = cv * p.inv_sqrt_D * rs;
          s = sv * p.inv_sqrt_D * rs;
        } else if (col == p.D) {
          c = rs;
        } else if (col == p.D + 1) {
          c = rs * tg;
        }
        if constexpr (COMPLEX_OUT) {
          if (live && col <= p.D)
            *reinterpret_cast<double2*>(p.phi + 2 * (row * (p.D + 1) + col)) = make_double2(c, -s);
        } else {
          p.Fc[row * p.Kp + col] = c;
          p.Fs[row * p.Kp + col] = s;
        }
      }
    }
}

// Bs[k][j] = B[k][j] / scale[k] for k < d, j < D; zero elsewhere (dk x Kp).
__global__ void k_build_Bs(const double* B, const double* scale, int d, int D, int dk, int Kp, double* Bs) {
  const long idx = blockIdx.x * (long)blockDim.x + threadIdx.x;
  if (idx >= (long)dk * Kp) return;
  const int k = idx / Kp, j = idx % Kp;
  Bs[idx] = (k < d && j < D) ? B[(long)k * D + j] / scale[k] : 0.0;
}

// ------------------------------------------------------------------------------------------------
// K2: Hermitian Gram of the extended feature planes, lower block triangle, split over rows.
// grid.x = ntri * nsplit; slab[(split * ntri + tile)][{R, I}][128][128].
// ------------------------------------------------------------------------------------------------
__device__ __forceinline__ void tri_decode(int t, int& tj, int& tk) {
  // t = tj (tj + 1) / 2 + tk, 0 <= tk <= tj
  int j = (int)((sqrt(8.0 * t + 1.0) - 1.0) * 0.5);
  while ((j + 1) * (j + 2) / 2 <= t) ++j;
  while (j * (j + 1) / 2 > t) --j;
  tj = j;
  tk = t - j * (j + 1) / 2;
}

__global__ void __launch_bounds__(Cfg8::NTHREADS, 2)
    k_gram(const double* Fc, const double* Fs, int Kp, long rows_pad, int ntri, long rows_per_split, double* slab) {
  extern __shared__ double smem[];
  const int tile = blockIdx.x % ntri, split = blockIdx.x / ntri;
  int tj, tk;
  tri_decode(tile, tj, tk);
  const long r0 = (long)split * rows_per_split;
  long r1 = r0 + rows_per_split;
  if (r1 > rows_pad) r1 = rows_pad;
  using C = Cfg8;
  v4d accR[C::MT][C::NTL], accI[C::MT][C::NTL];
  zero_acc(accR);
  zero_acc(accI);
  if (r1 > r0) {
    KMajorPlaneLoader<C> lac{Fc, Kp, (long)tj * BM}, las{Fs, Kp, (long)tj * BM};
    KMajorPlaneLoader<C> lbr{Fc, Kp, (long)tk * BN}, lbi{Fs, Kp, (long)tk * BN};
    mainloop_cplx<C, true>(accR, accI, lac, las, lbr, lbi, r0, (int)((r1 - r0) / BK), smem);
  }
  double* outR = slab + ((long)split * ntri + tile) * (2L * BM * BN);
  double* outI = outR + BM * BN;
#pragma unroll
  for (int mt = 0; mt < C::MT; ++mt)
#pragma unroll
    for (int nt = 0; nt < C::NTL; ++nt)
#pragma unroll
      for (int r = 0; r < 4; ++r) {
        const int o = C::acc_row(mt, r) * BN + C::acc_col(nt);
        outR[o] = accR[mt][nt][r];
        outI[o] = accI[mt][nt][r];
      }
}

// XCD-aware tile order shared by the 3M kernels.  Workgroups are dealt round-robin over the 8 XCDs (block b runs
// on XCD b % 8, each with its own 4 MiB L2), and 32 of them are resident per XCD.  Block b is therefore mapped to
// patch (b / 8) / 32 of XCD b % 8 and to position (b / 8) % 32 inside that patch of PR x PC tiles: the 32 workgroups
// that share an L2 start together on one compact patch and walk K in phase, so every A panel slice is fetched once
// for PC tiles and every B panel slice once for PR tiles (measured on k_rotate3: L2 hit rate 0.57 -> see
// profiles/r01_pmc_summary.md; speed only, never correctness - a different placement just hits less).
// Returns false for the padding blocks of partial patches.
__device__ __forceinline__ bool xcd_patch_tile_rt(int PR, int PC, long b, long tiles_r, long tiles_c, long& tr, long& tc) {
  const long xcd = b & 7, slot = b >> 3;
  const long patch = (slot / (PR * PC)) * 8 + xcd, within = slot % (PR * PC);
  const long patches_c = (tiles_c + PC - 1) / PC;
  tr = (patch / patches_c) * PR + within / PC;
  tc = (patch % patches_c) * PC + within % PC;
  return tr < tiles_r && tc < tiles_c;
}
template <int PR, int PC>
__device__ __forceinline__ bool xcd_patch_tile(long b, long tiles_r, long tiles_c, long& tr, long& tc) {
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

// 3M variant of K2 (nls_gemm3m.h) writing the same packed layout in 128 x 64 half tiles: half tile (tj, tk64),
// 0 <= tk64 <= 2 tj + 1, is columns (tk64 & 1) * 64 .. +63 of the 128 x 128 tile (tj, tk64 >> 1).  The nt x 2 nt
// grid.x = blocks_per_split * nsplit with blocks_per_split = nt (nt + 1) (plain order, default) or
// xcd_patch_grid(nt, 2 nt, 4, 8) (XCD patches: L2 hit rate 0.58 -> 0.72 but unbalanced XCDs, +11 % time; kept as a knob).
__global__ void __launch_bounds__(m3::NT3, 1)
    k_gram3(const double* Fc, const double* Fs, int Kp, long rows_pad, int ntri, long rows_per_split, double* slab, int nt,
            long blocks_per_split) {
  using namespace m3;
  extern __shared__ double smem[];
  const long split = blockIdx.x / blocks_per_split;
  int tj, tk64;
  if (blocks_per_split == 2L * ntri) {  // plain order: half tile h = tj (tj + 1) + tk64
    const int half = (int)(blockIdx.x % blocks_per_split);
    tj = (int)((sqrt(4.0 * half + 1.0) - 1.0) * 0.5);
    while ((tj + 1) * (tj + 2) <= half) ++tj;
    while (tj * (tj + 1) > half) --tj;
    tk64 = half - tj * (tj + 1);
  } else {  // XCD patches of 4 x 8 over the nt x 2 nt rectangle; blocks above the diagonal exit at once
    long tjl, tkl;
    if (!xcd_patch_tile<4, 8>(blockIdx.x % blocks_per_split, nt, 2L * nt, tjl, tkl)) return;
    if (tkl > 2 * tjl + 1) return;
    tj = (int)tjl;
    tk64 = (int)tkl;
  }
  const long r0 = (long)split * rows_per_split;
  long r1 = r0 + rows_per_split;
  if (r1 > rows_pad) r1 = rows_pad;
  v4d S1[MT3][NTL3], S2[MT3][NTL3], S3[MT3][NTL3];
  zero_acc(S1);
  zero_acc(S2);
  zero_acc(S3);
  if (r1 > r0) {
    KMajorLoader3<BM3, STAGE_A> lac{Fc, Kp, (long)tj * BM3}, las{Fs, Kp, (long)tj * BM3};
    KMajorLoader3<BN3, STAGE_B> lbr{Fc, Kp, (long)tk64 * BN3}, lbi{Fs, Kp, (long)tk64 * BN3};
    mainloop_3m<true>(S1, S2, S3, lac, las, lbr, lbi, r0, (int)((r1 - r0) / BK), smem);
  }
  const long tile128 = (long)tj * (tj + 1) / 2 + (tk64 >> 1);
  double* outR = slab + ((long)split * ntri + tile128) * (2L * BM * BN) + (tk64 & 1) * BN3;
  double* outI = outR + BM * BN;
#pragma unroll
  for (int mt = 0; mt < MT3; ++mt)
#pragma unroll
    for (int nt = 0; nt < NTL3; ++nt)
#pragma unroll
      for (int r = 0; r < 4; ++r) {
        const int o = acc_row3(mt, r) * BN + acc_col3(nt);
        const double s1 = S1[mt][nt][r], s2 = S2[mt][nt][r];
        outR[o] = s1 + s2;                     // Ac Br + As Bi
        outI[o] = (S3[mt][nt][r] - s1) + s2;   // Ac Bi - As Br
      }
}

// acc[tile] += sum_split slab[split][tile], fixed order (bit-reproducible).
__global__ void k_gram_reduce(const double* slab, int nsplit, long tile_elems_total, double* acc) {
  const long idx = blockIdx.x * (long)blockDim.x + threadIdx.x;
  if (idx >= tile_elems_total) return;
  double v = acc[idx];
  for (int s = 0; s < nsplit; ++s) v += slab[(long)s * tile_elems_total + idx];
  acc[idx] = v;
}

// Tile-packed extended Gram -> column-major complex A (D1 x D1, full Hermitian), optionally scaled,
// and b.  Element (j, k), j >= k of the packed Gram is (R, -I) of tile (j / 128, k / 128).
__device__ __forceinline__ double2 gram_elem(const double* g, int j, int k) {
  const int tj = j >> 7, tk = k >> 7;
  const double* t = g + ((long)tj * (tj + 1) / 2 + tk) * (2L * BM * BN);
  const int o = (j & 127) * BN + (k & 127);
  return make_double2(t[o], -t[BM * BN + o]);
}
__global__ void k_assemble_A(const double* g, int D1, double scale, double2* Acm, long lda, double2* b) {
  const int j = blockIdx.x * blockDim.x + threadIdx.x;  // row
  const int k = blockIdx.y;                             // column
  if (j >= D1) return;
  if (k < D1) {
    double2 v;
    if (j > k) {
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
    // b_j = conj(G[D1][j]) (row D+1 of the extended Gram is the target pseudo-feature)
    double2 v = gram_elem(g, D1, j);
    b[j] = make_double2(v.x, -v.y);
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

// Q (column-major complex, eigenvector k in column k) -> B-operand planes Qr, Qi [Kp x Np], zero padded.
__global__ void k_build_Q_planes(const double2* Qcm, long ldq, int D1, int Kp, int Np, double* Qr, double* Qi) {
  const int k = blockIdx.x * blockDim.x + threadIdx.x;  // eigen index (plane column)
  const int i = blockIdx.y;                             // feature index (plane row)
  if (k >= Np) return;
  double2 v = make_double2(0.0, 0.0);
  if (i < D1 && k < D1) v = Qcm[(long)k * ldq + i];
  Qr[(long)i * Np + k] = v.x;
  Qi[(long)i * Np + k] = v.y;
}

// v_k = (Q^H b)_k * inv_c, one block per k (column k of Q is contiguous).
__global__ void k_compute_v(const double2* Qcm, long ldq, const double2* b, int D1, double inv_c, int Np, double* vr,
                            double* vi) {
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
// K4: rotation P = phi Q with the P5 epilogue fused: U = Re(P v), Gm = |P|^2.
// grid = (Np / 128, rows_pad / 128).
// ------------------------------------------------------------------------------------------------
__global__ void __launch_bounds__(Cfg8::NTHREADS, 2)
    k_rotate(const double* Fc, const double* Fs, int Kp, const double* Qr, const double* Qi, int Np, const double* vr,
             const double* vi, double* U, double* Gm, const double* inv_rs) {
  extern __shared__ double smem[];
  const long row0 = (long)blockIdx.y * BM;
  const long col0 = (long)blockIdx.x * BN;
  using C = Cfg8;
  v4d accR[C::MT][C::NTL], accI[C::MT][C::NTL];
  zero_acc(accR);
  zero_acc(accI);
  MMajorPlaneLoader<C> lac{Fc, Kp, row0}, las{Fs, Kp, row0};
  KMajorPlaneLoader<C> lbr{Qr, Np, col0}, lbi{Qi, Np, col0};
  mainloop_cplx<C, false>(accR, accI, lac, las, lbr, lbi, 0, Kp / BK, smem);
#pragma unroll
  for (int nt = 0; nt < C::NTL; ++nt) {
    const long col = col0 + C::acc_col(nt);
    const double wr = vr[col], wi = vi[col];
#pragma unroll
    for (int mt = 0; mt < C::MT; ++mt)
#pragma unroll
      for (int r = 0; r < 4; ++r) {
        const long row = row0 + C::acc_row(mt, r);
        const double f = inv_rs ? inv_rs[row] : 1.0;  // planes hold rs_i phi_i: undo the row scale
        const double pr = accR[mt][nt][r] * f, pi = accI[mt][nt][r] * f;
        U[row * Np + col] = pr * wr - pi * wi;
        Gm[row * Np + col] = pr * pr + pi * pi;
      }
  }
}

// 3M variant of K4 (nls_gemm3m.h): 128-row x 64-column tiles in 4 x 8 patches per XCD.
// grid.x = xcd_patch_grid(rows_pad / 128, Np / 64, 4, 8).
__global__ void __launch_bounds__(m3::NT3, 1)
    k_rotate3(const double* Fc, const double* Fs, int Kp, const double* Qr, const double* Qi, int Np, const double* vr,
              const double* vi, double* U, double* Gm, const double* inv_rs, long tiles_r, int PR, int PC) {
  using namespace m3;
  extern __shared__ double smem[];
  long tr, tc;
  if (PR > 0) {
    if (!xcd_patch_tile_rt(PR, PC, blockIdx.x, tiles_r, Np / BN3, tr, tc)) return;
  } else {  // plain order: column tile fastest
    tc = blockIdx.x % (Np / BN3);
    tr = blockIdx.x / (Np / BN3);
    if (tr >= tiles_r) return;
  }
  const long row0 = tr * BM3;
  const long col0 = tc * BN3;
  v4d S1[MT3][NTL3], S2[MT3][NTL3], S3[MT3][NTL3];
  zero_acc(S1);
  zero_acc(S2);
  zero_acc(S3);
  MMajorLoader3 lac{Fc, Kp, row0}, las{Fs, Kp, row0};
  KMajorLoader3<BN3, STAGE_B> lbr{Qr, Np, col0}, lbi{Qi, Np, col0};
  mainloop_3m<false>(S1, S2, S3, lac, las, lbr, lbi, 0, Kp / BK, smem);
#pragma unroll
  for (int nt = 0; nt < NTL3; ++nt) {
    const long col = col0 + acc_col3(nt);
    const double wr = vr[col], wi = vi[col];
#pragma unroll
    for (int mt = 0; mt < MT3; ++mt)
#pragma unroll
      for (int r = 0; r < 4; ++r) {
        const long row = row0 + acc_row3(mt, r);
        const double f = inv_rs ? inv_rs[row] : 1.0;  // planes hold rs_i phi_i: undo the row scale
        const double s1 = S1[mt][nt][r], s2 = S2[mt][nt][r];
        const double pr = (s1 + s2) * f, pi = ((S3[mt][nt][r] - s1) + s2) * f;
        U[row * Np + col] = pr * wr - pi * wi;
        Gm[row * Np + col] = pr * pr + pi * pi;
      }
  }
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
  extern __shared__ double smem[];
  const long row0 = (long)blockIdx.y * BM;
  const long col0 = (long)blockIdx.x * BN;
  const bool second = blockIdx.z == 1;
  using C = Cfg4;
  v4d acc[C::MT][C::NTL];
  zero_acc(acc);
  MMajorPlaneLoader<C> la{second ? Gm : U, Np, row0};
  KMajorPlaneLoader<C> lb{R, Gp, col0};
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
// P6/P7: LOO residuals for every (row, gamma) and their weighted column sums.
//   e = (num - y) / (1 - s^2 hs); classifier: zero on the correct side (_neo_ls_svm.py:153-155)
//   part[blk][0][g] = sum s |e|, [1] = sum s [|e| >= 1], [2] = sum s max(0, |e| - 1)
// grid.x = ceil(n / LOO_ROWS_PER_BLOCK), block = 256 threads striding over g.
// ------------------------------------------------------------------------------------------------
__global__ void k_loo_errors(const double* num, const double* hs, const double* y, const double* s, long n, int G,
                             int Gp, int is_clf, double* part) {
  const long r0 = (long)blockIdx.x * LOO_ROWS_PER_BLOCK;
  long r1 = r0 + LOO_ROWS_PER_BLOCK;
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

// out[c][g] = sum_blk part[blk][c][g] in block order.
__global__ void k_sum_partials(const double* part, long nblk, long width, double* out) {
  const long idx = blockIdx.x * (long)blockDim.x + threadIdx.x;
  if (idx >= width) return;
  double v = 0.0;
  for (long b = 0; b < nblk; ++b) v += part[b * width + idx];
  out[idx] = v;
}

// Column of the selected gamma: P7 / P9 outputs and the score sums.
//   part[blk][0] = clf: sum s [sign(yloo) == y]   reg: sum s (y - yloo)^2
//   part[blk][1] = reg: sum s (y - ybar)^2
__global__ void k_loo_column(const double* num, const double* hs, const double* y, const double* s, long n, int Gp,
                             int g, int is_clf, double ybar, double* loo_res, double* loo_lev, double* loo_std,
                             double* part) {
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
// K8: yhat_i = Re(phi_i . beta) = Fc_i . beta_r + Fs_i . beta_i ; one wave per row.
// out = yhat - y (clipped for classifiers) when y != nullptr, else yhat.
// ------------------------------------------------------------------------------------------------
__global__ void k_plane_gemv(const double* Fc, const double* Fs, int Kp, const double* br, const double* bi, long rows,
                             const double* y, int is_clf, double* out, const double* inv_rs) {
  const long row = blockIdx.x * (long)(blockDim.x >> 6) + (threadIdx.x >> 6);
  const int lane = threadIdx.x & 63;
  if (row >= rows) return;
  const double* c = Fc + row * Kp;
  const double* s = Fs + row * Kp;
  double acc = 0.0;
  for (int j = 2 * lane; j < Kp; j += 128) {
    const double2 cv = *reinterpret_cast<const double2*>(c + j), sv = *reinterpret_cast<const double2*>(s + j);
    const double2 rv = *reinterpret_cast<const double2*>(br + j), iv = *reinterpret_cast<const double2*>(bi + j);
    acc += cv.x * rv.x + sv.x * iv.x + cv.y * rv.y + sv.y * iv.y;
  }
  for (int o = 32; o > 0; o >>= 1) acc += __shfl_down(acc, o);
  if (lane == 0) {
    if (inv_rs) acc *= inv_rs[row];
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

__global__ void k_conj_inplace(double2* a, long n) {
  const long i = blockIdx.x * (long)blockDim.x + threadIdx.x;
  if (i < n) a[i].y = -a[i].y;
}

// beta (complex, D1) -> planes br, bi [Kp] zero padded.
__global__ void k_split_vec(const double2* v, int D1, int Kp, double* vr, double* vi) {
  const int i = blockIdx.x * blockDim.x + threadIdx.x;
  if (i >= Kp) return;
  vr[i] = i < D1 ? v[i].x : 0.0;
  vi[i] = i < D1 ? v[i].y : 0.0;
}

}  // namespace nls
