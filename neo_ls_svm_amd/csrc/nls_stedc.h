// Divide-and-conquer eigensolver for real symmetric TRIDIAGONAL matrices, T = diag(d) + offdiag(e)  ->  lam ascending, T Q = Q diag(lam):
// the tridiagonal stage of both eigendecompositions of the path (P4: eigh(A / c), _neo_ls_svm.py:120; D2: eigh(sn K sn), :265).
//
// Why not rocsolver_dstedc: 119 ms of the dual c4 fit (n = 10^4; its secular-equation kernel alone 61 ms: one root per THREAD, every thread
// walking all n poles), 25 ms at n = 4097, 4.3 ms at n = 1025, with host synchronisations inside - a quarter of the dual eigendecomposition,
// and the last library call on its critical path.
//
// Here (tools/stedc_proto.py is the NumPy prototype the conventions were fixed with): Cuppen's divide and conquer organised as LAPACK's
// dstedc / dlaed0 .. dlaed4, level-synchronous - leaves of 32 rows (implicit QL, one wave per leaf), then log2(n / 32) levels in which all
// merges of neighbouring blocks run side by side in the same launches:
//   k_dc_setup   : z = (last row of Q1, first row of Q2) / sqrt 2, the merged order of the two sorted spectra (binary-search ranks), dlaed2's
//                  deflation scan (negligible z_i; close poles rotated into one) by one thread over LDS-staged chunks;
//   k_dc_rotate  : the deflation rotations on the eigenvector columns, rows in parallel, the chain's running column carried in a register;
//   k_dc_gather  : the k non-deflated columns, contiguous;
//   k_dc_secular : the roots of 1 + rho sum w_i^2 / (d_i - lam) = 0 as (nearest pole, offset): one wave per 4 roots, the poles spread over its
//                  lanes; dlaed4's iteration (a two-pole rational model of the function, "the middle way") with its stopping criterion, every
//                  step that leaves the bracket replaced by a bisection step in the logarithm of the offset: 3 - 6 sweeps over the poles;
//   k_dc_zhat    : Gu / Eisenstat (dlaed3): z is RECOMPUTED from the computed roots - they are then the exact eigenvalues of
//                  D + rho zhat zhat^T - so orthogonality does not rest on the accuracy of the roots; every difference d_i - lam_j is formed as
//                  (d_i - d_origin(j)) - mu_j, never from a rounded lam_j;
//   k_dc_vectors : U_ij = zhat_i / (d_i - lam_j), columns normalised;
//   k_dc_gemm    : Q_new = Q[:, kept] U on the fp64 MFMA tile engine (nls_gemm.h), all merges of the level in one launch;
//   k_dc_place / k_dc_scatter : the merged ascending order of new and deflated eigenvalues, columns copied to their places.
// The number of non-deflated poles k of a merge lives in device memory only: every launch is sized for k = m and returns early - no
// host synchronisation anywhere, a fixed launch sequence per n.
#pragma once
#include "nls_gemm.h"

namespace nls {
namespace dc {

constexpr int LEAF = 32;
constexpr int SCAN_CHUNK = 2048;  // poles staged in LDS per step of the deflation scan
constexpr int RW = 4;             // roots per wave in the secular kernel

struct MergeInfo {
  int k;       // non-deflated poles
  int ndefl;   // deflated (their eigenpairs are final)
  int nrot;    // deflation rotations
  int k1, k2;  // of the k kept columns: k1 live in the upper block only, k2 are dense (rotated across the blocks), the rest in the lower block only
  int pad;
  double rho;  // 2 |e|: the rank-one coefficient after normalising z
};
struct Rot {
  int a, b;  // columns (block-relative): a is deflated by the rotation, b carries on
  double c, s;
};

struct Level {  // geometry of one level: blocks of S rows are merged in pairs
  int n, S;
  long P;  // slot edge of the per-merge matrices (U, G, R): min(2 S, n) rounded up to 128
  __host__ __device__ int b0(int mi) const { return mi * 2 * S; }
  __host__ __device__ int mid(int mi) const { return mi * 2 * S + S; }
  __host__ __device__ int b1(int mi) const { return (mi * 2 * S + 2 * S < n) ? mi * 2 * S + 2 * S : n; }
};

__device__ __forceinline__ double wave_sum_d(double v) {
#pragma unroll
  for (int o = 32; o > 0; o >>= 1) v += __shfl_xor(v, o, 64);
  return v;
}
__device__ __forceinline__ double wave_max_d(double v) {
#pragma unroll
  for (int o = 32; o > 0; o >>= 1) v = fmax(v, __shfl_xor(v, o, 64));
  return v;
}

// ---- leaves ---------------------------------------------------------------------------------------------------------------------------
// Tears T at every multiple of LEAF (d[p - 1] -= |e[p - 1]|, d[p] -= |e[p - 1]|: T = diag(T1', T2') + |rho| v v^T) and solves each leaf by the
// implicit QL algorithm (EISPACK tql2): one wave per leaf, lane = row of the eigenvector block (in LDS), every lane runs the scalar recurrence.
// Out: lam[b0 .. b1) ascending, Q[b0 .. b1, b0 .. b1) (the rest of the block's columns zeroed by the caller); info raised on no convergence.
__global__ void __launch_bounds__(64) k_dc_leaf(int n, const double* d_in, const double* e_in, double* lam, double* Q, long ldq, int* info) {
  __shared__ double d[LEAF], e[LEAF], V[LEAF][LEAF + 1];
  const int b0 = blockIdx.x * LEAF, nb = min(LEAF, n - b0), lane = threadIdx.x;
  if (lane < nb) {
    double v = d_in[b0 + lane];
    if (lane == 0 && b0 > 0) v -= fabs(e_in[b0 - 1]);
    if (lane == nb - 1 && b0 + nb < n) v -= fabs(e_in[b0 + nb - 1]);
    d[lane] = v;
    e[lane] = lane < nb - 1 ? e_in[b0 + lane] : 0.0;  // e[i] couples rows i, i + 1 of the leaf
    for (int c = 0; c < nb; ++c) V[lane][c] = lane == c ? 1.0 : 0.0;
  }
  __syncthreads();
  const double eps = 2.220446049250313e-16;
  double f = 0.0, tst1 = 0.0;
  int bad = 0;
  for (int l = 0; l < nb; ++l) {
    tst1 = fmax(tst1, fabs(d[l]) + fabs(e[l]));
    int m = l;
    while (m < nb - 1 && fabs(e[m]) > eps * tst1) ++m;
    if (m > l) {
      int iter = 0;
      do {
        if (++iter > 60) {
          bad = 1;
          break;
        }
        double g = d[l], p = (d[l + 1] - g) / (2.0 * e[l]), r = hypot(p, 1.0);
        if (p < 0.0) r = -r;
        __syncthreads();  // (all lanes have read d, e of this sweep's start)
        const double dl1 = e[l] * (p + r), h = g - e[l] / (p + r), el1 = e[l + 1];
        if (lane == 0) {
          d[l] = e[l] / (p + r);
          d[l + 1] = dl1;
          for (int i = l + 2; i < nb; ++i) d[i] -= h;
        }
        __syncthreads();
        f += h;
        p = d[m];
        double c = 1.0, c2 = 1.0, c3 = 1.0, s = 0.0, s2 = 0.0;
        for (int i = m - 1; i >= l; --i) {
          c3 = c2;
          c2 = c;
          s2 = s;
          const double ei = e[i], di = d[i];
          g = c * ei;
          const double hh = c * p;
          r = hypot(p, ei);
          const double enew = s * r;
          s = ei / r;
          c = p / r;
          p = c * di - s * g;
          const double dnew = hh + s * (c * g + s * di);
          if (lane < nb) {  // accumulate the rotation in this lane's row of V
            const double vh = V[lane][i + 1];
            V[lane][i + 1] = s * V[lane][i] + c * vh;
            V[lane][i] = c * V[lane][i] - s * vh;
          }
          __syncthreads();  // (everybody has read e[i], d[i] before lane 0 overwrites their neighbours)
          if (lane == 0) {
            e[i + 1] = enew;
            d[i + 1] = dnew;
          }
          __syncthreads();
        }
        p = -s * s2 * c3 * el1 * e[l] / dl1;
        __syncthreads();
        if (lane == 0) {
          e[l] = s * p;
          d[l] = c * p;
        }
        __syncthreads();
      } while (fabs(e[l]) > eps * tst1);
    }
    __syncthreads();
    if (lane == 0) {
      d[l] += f;
      e[l] = 0.0;
    }
    __syncthreads();
  }
  // ascending order: rank of every eigenvalue (ties by index), columns written to their places
  if (lane < nb) {
    const double v = d[lane];
    int rank = 0;
    for (int c = 0; c < nb; ++c) rank += (d[c] < v || (d[c] == v && c < lane)) ? 1 : 0;
    lam[b0 + rank] = v;
    for (int r = 0; r < nb; ++r) Q[(long)(b0 + r) + (long)(b0 + rank) * ldq] = V[r][lane];
  }
  if (bad && lane == 0) atomicCAS(info, 0, b0 + 1);
}

// ---- merge: setup -----------------------------------------------------------------------------------------------------------------------
// One workgroup per merge.  Arrays of length n are indexed b0 + (position inside the merge):
//   ds, zs, src : eigenvalues, z components and block-relative source columns in merged ascending order (ds is updated by the rotations);
//   dl, w, kidx : the k kept poles (ascending), their z components, their sorted positions;   didx: sorted positions of the deflated ones.
__global__ void __launch_bounds__(256) k_dc_setup(Level L, const double* lam, const double* Q, long ldq, const double* e, double* ds, double* zs, int* src,
                                                  double* dl, double* w, int* kidx, int* didx, int* gpos, Rot* rots, MergeInfo* info) {
  __shared__ double sd[SCAN_CHUNK], sz[SCAN_CHUNK];
  __shared__ int cnt[4];
  __shared__ double red[2][4];
  const int mi = blockIdx.x, tid = threadIdx.x;
  const int b0 = L.b0(mi), mid = L.mid(mi), b1 = L.b1(mi);
  if (b0 >= L.n) return;
  const int m = b1 - b0;
  if (mid >= L.n) {  // a block without partner: passed through as "everything deflated"
    for (int t = tid; t < m; t += 256) {
      ds[b0 + t] = lam[b0 + t];
      src[b0 + t] = t;
      didx[b0 + t] = t;
    }
    if (tid == 0) info[mi] = MergeInfo{0, m, 0, 0, 0, 0, 0.0};
    return;
  }
  const int n1 = mid - b0;
  const double rho_raw = e[mid - 1], sgn = rho_raw < 0.0 ? -1.0 : 1.0, rho = 2.0 * fabs(rho_raw);
  double dmax = 0.0, zmax = 0.0;
  for (int t = tid; t < m; t += 256) {
    const double val = lam[b0 + t];
    double zt;
    int lo, hi, rank;
    if (t < n1) {  // rank = t + #(second half < val)
      zt = Q[(long)(mid - 1) + (long)(b0 + t) * ldq];
      lo = 0;
      hi = m - n1;
      while (lo < hi) {
        const int c = (lo + hi) >> 1;
        if (lam[mid + c] < val) lo = c + 1; else hi = c;
      }
      rank = t + lo;
    } else {  // rank = (t - n1) + #(first half <= val)
      zt = sgn * Q[(long)mid + (long)(b0 + t) * ldq];
      lo = 0;
      hi = n1;
      while (lo < hi) {
        const int c = (lo + hi) >> 1;
        if (lam[b0 + c] <= val) lo = c + 1; else hi = c;
      }
      rank = (t - n1) + lo;
    }
    zt *= 0.70710678118654752440;
    ds[b0 + rank] = val;
    zs[b0 + rank] = zt;
    src[b0 + rank] = t;
    dmax = fmax(dmax, fabs(val));
    zmax = fmax(zmax, fabs(zt));
  }
  dmax = wave_max_d(dmax);
  zmax = wave_max_d(zmax);
  if ((tid & 63) == 0) {
    red[0][tid >> 6] = dmax;
    red[1][tid >> 6] = zmax;
  }
  __syncthreads();
  dmax = fmax(fmax(red[0][0], red[0][1]), fmax(red[0][2], red[0][3]));
  zmax = fmax(fmax(red[1][0], red[1][1]), fmax(red[1][2], red[1][3]));
  const double tol = 8.0 * 2.220446049250313e-16 * fmax(dmax, zmax);
  if (rho * zmax <= tol) {  // the coupling is negligible: the merged spectrum is the union
    for (int t = tid; t < m; t += 256) didx[b0 + t] = t;
    if (tid == 0) info[mi] = MergeInfo{0, m, 0, 0, 0, 0, rho};
    return;
  }
  // dlaed2's scan, sequential in the merged order: thread 0, the poles staged through LDS
  // Column types as in dlaed2 (the product Q[:, kept] U is block structured): 1 = lives in the upper block only, 3 = in the lower block only,
  // 2 = dense (a rotation mixed an upper with a lower column).  gpos[i] starts as (type << 28) | ordinal within the type.
  //
  // Fast path, all threads: the scan below is sequential only through its rotations.  Every thread takes a contiguous segment of the merged
  // order; a first pass counts the segment's candidates (components that are not negligible) and remembers its last one, a prefix over the 256
  // segments gives every thread its offsets and the candidate before its segment, and a second pass writes the kept / deflated lists and
  // tests every consecutive pair of candidates for the rotation criterion.  No pair passing it means the sequential scan would never rotate -
  // the lists are then exactly its result; otherwise thread 0 runs the scan (and overwrites them).
  __shared__ int seg_cnt[256][4];  // candidates of type 1, of type 3, deflated, last candidate (-1: none)
  __shared__ int any_rot;
  {
    const int len = (m + 255) / 256, j0 = min(tid * len, m), j1 = min(j0 + len, m);
    int c1 = 0, c3 = 0, cd = 0, last = -1;
    for (int j = j0; j < j1; ++j) {
      if (rho * fabs(zs[b0 + j]) <= tol) ++cd;
      else {
        if (src[b0 + j] < n1) ++c1; else ++c3;
        last = j;
      }
    }
    seg_cnt[tid][0] = c1;
    seg_cnt[tid][1] = c3;
    seg_cnt[tid][2] = cd;
    seg_cnt[tid][3] = last;
    if (tid == 0) any_rot = 0;
    __syncthreads();
    int o1 = 0, o3 = 0, od = 0, prev = -1;  // exclusive prefix (256 short reads per thread)
    for (int t = 0; t < tid; ++t) {
      o1 += seg_cnt[t][0];
      o3 += seg_cnt[t][1];
      od += seg_cnt[t][2];
      if (seg_cnt[t][3] >= 0) prev = seg_cnt[t][3];
    }
    int t1 = 0, t3 = 0, td = 0;
    if (tid == 255) {
      t1 = o1 + c1;
      t3 = o3 + c3;
      td = od + cd;
    }
    int ko = o1 + o3;
    bool rot = false;
    for (int j = j0; j < j1; ++j) {
      const double dj = ds[b0 + j], zj = zs[b0 + j];
      if (rho * fabs(zj) <= tol) {
        didx[b0 + od++] = j;
        continue;
      }
      if (prev >= 0) {
        const double zp = zs[b0 + prev], tau = hypot(zj, zp), c = zj / tau, sn = -zp / tau;
        rot = rot || fabs((dj - ds[b0 + prev]) * c * sn) <= tol;
      }
      const int ty = src[b0 + j] < n1 ? 1 : 3;
      dl[b0 + ko] = dj;
      w[b0 + ko] = zj;
      kidx[b0 + ko] = j;
      gpos[b0 + ko] = (ty << 28) | (ty == 1 ? o1++ : o3++);
      ++ko;
      prev = j;
    }
    if (rot) any_rot = 1;
    if (tid == 255) {
      info[mi] = MergeInfo{t1 + t3, td, 0, t1, 0, 0, rho};
      cnt[0] = t1 + t3;
      cnt[1] = t1;
      cnt[2] = 0;
    }
    __syncthreads();
    if (!any_rot) {
      for (int i = tid; i < cnt[0]; i += 256) {
        const int g = gpos[b0 + i], ty = g >> 28, ord = g & 0x0fffffff;
        gpos[b0 + i] = ord + (ty == 1 ? 0 : cnt[1]);
      }
      return;
    }
    __syncthreads();
  }
  int k = 0, nd = 0, nrot = 0, pj = -1, tpj = 0, nty[4] = {0, 0, 0, 0};
  double dpj = 0.0, zpj = 0.0;
  for (int c0 = 0; c0 < m; c0 += SCAN_CHUNK) {
    const int cn = min(SCAN_CHUNK, m - c0);
    __syncthreads();
    for (int t = tid; t < cn; t += 256) {
      sd[t] = ds[b0 + c0 + t];
      sz[t] = zs[b0 + c0 + t];
    }
    __syncthreads();
    if (tid == 0) {
      for (int jj = 0; jj < cn; ++jj) {
        const int j = c0 + jj;
        const double dj = sd[jj], zj = sz[jj];
        if (rho * fabs(zj) <= tol) {  // negligible component: the eigenpair is final
          didx[b0 + nd++] = j;
          continue;
        }
        const int tj = src[b0 + j] < n1 ? 1 : 3;
        if (pj < 0) {
          pj = j;
          dpj = dj;
          zpj = zj;
          tpj = tj;
          continue;
        }
        const double tau = hypot(zj, zpj), t = dj - dpj, c = zj / tau, s = -zpj / tau;
        if (fabs(t * c * s) <= tol) {  // close poles: rotate z[pj] away, pj is final
          rots[b0 + nrot++] = Rot{src[b0 + pj], src[b0 + j], c, s};
          ds[b0 + pj] = dpj * c * c + dj * s * s;
          didx[b0 + nd++] = pj;
          dpj = dpj * s * s + dj * c * c;
          zpj = tau;
          pj = j;
          tpj = tpj == tj ? tj : 2;
        } else {
          dl[b0 + k] = dpj;
          w[b0 + k] = zpj;
          gpos[b0 + k] = (tpj << 28) | nty[tpj]++;
          kidx[b0 + k++] = pj;
          pj = j;
          dpj = dj;
          zpj = zj;
          tpj = tj;
        }
      }
    }
  }
  if (tid == 0) {
    if (pj >= 0) {
      dl[b0 + k] = dpj;
      w[b0 + k] = zpj;
      gpos[b0 + k] = (tpj << 28) | nty[tpj]++;
      kidx[b0 + k++] = pj;
    }
    info[mi] = MergeInfo{k, nd, nrot, nty[1], nty[2], 0, rho};
    cnt[0] = k;
    cnt[1] = nty[1];
    cnt[2] = nty[2];
  }
  __syncthreads();
  for (int i = tid; i < cnt[0]; i += 256) {  // position in the gathered order: type 1, then 2, then 3
    const int g = gpos[b0 + i], ty = g >> 28, ord = g & 0x0fffffff;
    gpos[b0 + i] = ord + (ty == 1 ? 0 : ty == 2 ? cnt[1] : cnt[1] + cnt[2]);
  }
}

// ---- merge: the deflation rotations on the eigenvector columns --------------------------------------------------------------------------
// (q_a, q_b) <- (c q_a + s q_b, c q_b - s q_a), in the order of the scan.  A rotation's second column is the first column of the next one
// while a chain of close poles lasts: it stays in a register.  grid = (row chunks of 256, merges).
__global__ void __launch_bounds__(256) k_dc_rotate(Level L, double* Q, long ldq, const Rot* rots, const MergeInfo* info) {
  const int mi = blockIdx.y, b0 = L.b0(mi);
  if (b0 >= L.n) return;
  const int nrot = info[mi].nrot, m = L.b1(mi) - b0;
  const int r = blockIdx.x * 256 + threadIdx.x;
  if (nrot == 0 || r >= m) return;
  double* Qr = Q + (long)(b0 + r) + (long)b0 * ldq;
  int carry_col = -1;
  double carry = 0.0;
  for (int q = 0; q < nrot; ++q) {
    const Rot R = rots[b0 + q];
    if (R.a != carry_col) {
      if (carry_col >= 0) Qr[(long)carry_col * ldq] = carry;
      carry = Qr[(long)R.a * ldq];
    }
    const double qb = Qr[(long)R.b * ldq];
    Qr[(long)R.a * ldq] = R.c * carry + R.s * qb;
    carry = R.c * qb - R.s * carry;
    carry_col = R.b;
  }
  if (carry_col >= 0) Qr[(long)carry_col * ldq] = carry;
}

// ---- merge: the kept columns, contiguous (G: P x P slot, column-major; zero in rows m .. and in columns k .. up to the next multiple of 16) ----
// grid = (row chunks of 256, column groups of 16, merges)
__global__ void __launch_bounds__(256) k_dc_gather(Level L, const double* Q, long ldq, const int* src, const int* kidx, const int* gpos, const MergeInfo* info, double* G) {
  const int mi = blockIdx.z, b0 = L.b0(mi);
  if (b0 >= L.n) return;
  const int k = info[mi].k, m = L.b1(mi) - b0;
  const int k16 = (k + 15) & ~15, i0 = blockIdx.y * 16;
  if (i0 >= k16) return;
  const long r = (long)blockIdx.x * 256 + threadIdx.x;
  if (r >= L.P) return;
  double* Gs = G + (long)mi * L.P * L.P;
#pragma unroll 4
  for (int i = i0; i < i0 + 16; ++i) {  // kept pole i goes to column gpos[i] (columns grouped by type); the padding columns k .. k16 are zeroed
    double v = 0.0;
    if (i < k && r < m) v = Q[(long)(b0 + r) + (long)(b0 + src[b0 + kidx[b0 + i]]) * ldq];
    Gs[r + (long)(i < k ? gpos[b0 + i] : i) * L.P] = v;
  }
}

// ---- merge: secular equation ------------------------------------------------------------------------------------------------------------
// Root j in (dl_j, dl_j+1) (the last one in (dl_k-1, dl_k-1 + rho |w|^2]) as (origin pole, offset mu): lam_j = dl[org_j] + mu_j, the origin being
// the nearer end of the interval (sign of the function at the midpoint).  A wave takes RW consecutive roots; its lanes stride over the poles.
// grid = (ceil(m / (4 RW)), merges), 256 threads.
__global__ void __launch_bounds__(256) k_dc_secular(Level L, const double* dl, const double* w, const MergeInfo* info, double* dorg, double* mu, double* lamnew) {
  const int mi = blockIdx.y, b0 = L.b0(mi);
  if (b0 >= L.n) return;
  const int k = info[mi].k;
  const double rho = info[mi].rho;
  const int lane = threadIdx.x & 63, wv = threadIdx.x >> 6;
  const int j0 = (blockIdx.x * 4 + wv) * RW;
  if (j0 >= k) return;
  const double* D = dl + b0;
  const double* W = w + b0;
  const double tiny = 2.2250738585072014e-308;
  double dj[RW], hi[RW], dor[RW];
  bool right[RW], live[RW];
  // midpoint values decide the origin; for the last root the bracket is (0, rho |w|^2]
  double acc[RW];
#pragma unroll
  for (int r = 0; r < RW; ++r) {
    const int j = j0 + r;
    live[r] = j < k;
    dj[r] = live[r] ? D[j] : 0.0;
    const bool last = j >= k - 1;
    hi[r] = (live[r] && !last) ? 0.5 * (D[j + 1] - dj[r]) : 0.0;
    acc[r] = 0.0;
  }
  double w2sum = 0.0;
  for (int i = lane; i < k; i += 64) {
    const double di = D[i], w2 = W[i] * W[i];
    w2sum += w2;
#pragma unroll
    for (int r = 0; r < RW; ++r) acc[r] += w2 / ((di - dj[r]) - hi[r]);
  }
  w2sum = wave_sum_d(w2sum);
#pragma unroll
  for (int r = 0; r < RW; ++r) {
    const int j = j0 + r;
    const double fm = 1.0 + rho * wave_sum_d(acc[r]);
    const bool last = j >= k - 1;
    right[r] = last || fm >= 0.0;
    if (last) hi[r] = fmax(rho * w2sum, tiny);
    dor[r] = right[r] ? dj[r] : (live[r] ? D[min(j + 1, k - 1)] : 0.0);
    acc[r] = 0.0;
  }
  // a point next to the origin pole where the function still has the sign it has at the pole: bound the other poles' contribution at hi
  double wo2[RW];
#pragma unroll
  for (int r = 0; r < RW; ++r) wo2[r] = 0.0;
  for (int i = lane; i < k; i += 64) {
    const double di = D[i], w2 = W[i] * W[i];
#pragma unroll
    for (int r = 0; r < RW; ++r) {
      const double delta = di - dor[r];
      if (delta == 0.0) wo2[r] = w2;  // (the origin itself; poles are distinct)
      if (right[r]) {
        if (delta > 0.0) acc[r] += w2 / fmax(delta - hi[r], tiny);
      } else {
        if (delta < 0.0) acc[r] += w2 / fmax(-(delta + hi[r]), tiny);
      }
    }
  }
  // dlaed4's iteration ("the middle way", Li 1994) in the offset tau from the origin pole: with psi = the sum over the poles left of the
  // root and phi = over those right of it, the function f = 1 / rho + psi + phi is modelled by c + a / (dL - t) + b / (dR - t) (dL, dR: the
  // two poles that enclose the root) matching value and slope of psi and of phi separately; the model's root inside the interval is the
  // next iterate.  An iterate that leaves the bracket [tlo, thi] (kept by the sign of f) is replaced by a bisection step in log |tau|.  It
  // stops by dlaed4's criterion - |f| at the level of its own rounding error - or when the step or the bracket is below rounding:
  // 3 - 6 sweeps over the poles per root (a plain log-bisection took ~60).
  double tau[RW], tlo[RW], thi[RW], dL[RW], dR[RW];
  bool done[RW], lastr[RW];
#pragma unroll
  for (int r = 0; r < RW; ++r) {
    const double others = rho * wave_sum_d(acc[r]);
    const double wo = wave_sum_d(wo2[r]);
    double l = right[r] ? 0.5 * rho * wo / (1.0 + others) : 0.5 * rho * wo / fmax(others - 1.0, 1.0);
    const double h = fmax(hi[r], tiny);
    l = fmax(fmin(l, h), tiny);
    const int j = j0 + r;
    lastr[r] = j >= k - 1;
    const double gap = (live[r] && !lastr[r]) ? D[j + 1] - dj[r] : 0.0;
    dL[r] = right[r] ? 0.0 : -gap;
    dR[r] = right[r] ? gap : 0.0;
    tlo[r] = right[r] ? l : -h;
    thi[r] = right[r] ? h : -l;
    tau[r] = right[r] ? fmin(2.0 * l, h) : -fmin(2.0 * l, h);
    done[r] = !live[r];
  }
  const double eps = 2.220446049250313e-16, rhoinv = 1.0 / rho;
  for (int it = 0; it < 80; ++it) {
    bool any = false;
#pragma unroll
    for (int r = 0; r < RW; ++r) any = any || !done[r];
    if (!any) break;  // (wave-uniform: every lane holds the same state)
    double psi[RW], dpsi[RW], phi[RW], dphi[RW];
#pragma unroll
    for (int r = 0; r < RW; ++r) psi[r] = dpsi[r] = phi[r] = dphi[r] = 0.0;
    for (int i = lane; i < k; i += 64) {
      const double di = D[i], w2 = W[i] * W[i];
#pragma unroll
      for (int r = 0; r < RW; ++r) {
        const double rd = 1.0 / ((di - dor[r]) - tau[r]);
        const double t = w2 * rd, t2 = t * rd;
        const bool left = i <= j0 + r;
        psi[r] += left ? t : 0.0;
        dpsi[r] += left ? t2 : 0.0;
        phi[r] += left ? 0.0 : t;
        dphi[r] += left ? 0.0 : t2;
      }
    }
#pragma unroll
    for (int r = 0; r < RW; ++r) {
      const double ps = wave_sum_d(psi[r]), dps = wave_sum_d(dpsi[r]), ph = wave_sum_d(phi[r]), dph = wave_sum_d(dphi[r]);
      if (done[r]) continue;
      const double f = rhoinv + ps + ph, t0 = tau[r];
      if (f < 0.0) tlo[r] = t0; else thi[r] = t0;
      if (fabs(f) <= eps * (8.0 * (ph - ps) + 2.0 * rhoinv + fabs(t0) * (dps + dph))) {
        done[r] = true;
        continue;
      }
      const double DL = dL[r] - t0;
      double eta;
      if (lastr[r]) {  // one pole: c + a / (DL - eta) = 0
        const double c = f - DL * dps, a = dps * DL * DL;
        eta = DL + a / c;
      } else {
        const double DR = dR[r] - t0;
        const double c = f - DL * dps - DR * dph, a = dps * DL * DL, b = dph * DR * DR;
        const double B = c * (DL + DR) + a + b, C = DL * DR * f;
        const double disc = sqrt(fabs(B * B - 4.0 * c * C));
        if (c == 0.0) eta = C / B;
        else if (B <= 0.0) eta = (B - disc) / (2.0 * c);
        else eta = 2.0 * C / (B + disc);
      }
      double tn = t0 + eta;
      if (!(tn > tlo[r] && tn < thi[r])) tn = (t0 < 0.0 ? -1.0 : 1.0) * sqrt(fabs(tlo[r])) * sqrt(fabs(thi[r]));
      if (fabs(tn - t0) <= 8.0 * eps * fabs(tn) || fabs(thi[r] - tlo[r]) <= 8.0 * eps * fabs(tn)) done[r] = true;
      tau[r] = tn;
    }
  }
  if (lane == 0) {
#pragma unroll
    for (int r = 0; r < RW; ++r) {
      const int j = j0 + r;
      if (j < k) {
        dorg[b0 + j] = dor[r];
        mu[b0 + j] = tau[r];
        lamnew[b0 + j] = dor[r] + tau[r];
      }
    }
  }
}

// ---- merge: zhat (dlaed3), one wave per pole ----------------------------------------------------------------------------------------------
// zhat_i^2 = prod_j (lam_j - dl_i) / prod_{j != i} (dl_j - dl_i) (up to the common factor rho: the vectors are normalised afterwards), every
// lam_j - dl_i as mu_j - (dl_i - dorg_j).  The running product is kept as mantissa x 2^exponent.   grid = (ceil(m / 4), merges)
__global__ void __launch_bounds__(256) k_dc_zhat(Level L, const double* dl, const double* w, const double* dorg, const double* mu, const MergeInfo* info, double* zhat) {
  const int mi = blockIdx.y, b0 = L.b0(mi);
  if (b0 >= L.n) return;
  const int k = info[mi].k, lane = threadIdx.x & 63, i = blockIdx.x * 4 + (threadIdx.x >> 6);
  if (i >= k) return;
  const double di = dl[b0 + i];
  double mant = 1.0;
  int ex = 0, cnt = 0;
  for (int j = lane; j < k; j += 64) {
    const double num = mu[b0 + j] - (di - dorg[b0 + j]);  // lam_j - dl_i
    const double den = j == i ? 1.0 : dl[b0 + j] - di;
    mant *= num / den;
    if ((++cnt & 7) == 0) {
      int e2;
      mant = frexp(mant, &e2);
      ex += e2;
    }
  }
  int e2;
  mant = frexp(mant, &e2);
  ex += e2;
#pragma unroll
  for (int o = 32; o > 0; o >>= 1) {  // product over the lanes, renormalised at every step
    mant *= __shfl_xor(mant, o, 64);
    ex += __shfl_xor(ex, o, 64);
    mant = frexp(mant, &e2);
    ex += e2;
  }
  if (lane == 0) {
    const double z2 = ldexp(fabs(mant), ex);
    zhat[b0 + i] = w[b0 + i] < 0.0 ? -sqrt(z2) : sqrt(z2);
  }
}

// ---- merge: eigenvectors of the rank-one update, one workgroup per column (U: P x P slot, column-major) -----------------------------------
// U_ij = zhat_i / (dl_i - lam_j), normalised; rows k .. up to the next multiple of 16 zeroed (the K padding of the product).  grid = (m, merges)
__global__ void __launch_bounds__(256) k_dc_vectors(Level L, const double* dl, const double* zhat, const double* dorg, const double* mu, const int* gpos, const MergeInfo* info,
                                                    double* U) {
  __shared__ double red[4];
  const int mi = blockIdx.y, b0 = L.b0(mi);
  if (b0 >= L.n) return;
  const int k = info[mi].k, j = blockIdx.x;
  if (j >= k) return;
  const int k16 = (k + 15) & ~15;
  double* Uc = U + (long)mi * L.P * L.P + (long)j * L.P;
  const double dor = dorg[b0 + j], mj = mu[b0 + j];
  double nrm = 0.0;
  for (int i = threadIdx.x; i < k16; i += 256) {  // row = the pole's column in the gathered order
    double v = 0.0;
    if (i < k) v = zhat[b0 + i] / ((dl[b0 + i] - dor) - mj);
    Uc[i < k ? gpos[b0 + i] : i] = v;
    nrm += v * v;
  }
  nrm = wave_sum_d(nrm);
  if ((threadIdx.x & 63) == 0) red[threadIdx.x >> 6] = nrm;
  __syncthreads();
  const double inv = 1.0 / sqrt((red[0] + red[1]) + (red[2] + red[3]));
  for (int i = threadIdx.x; i < k; i += 256) Uc[i] *= inv;  // (each thread scales the entries it wrote or others wrote: after the barrier)
}

// ---- merge: R = G U on the tile engine, all merges of the level in one launch ------------------------------------------------------------
// Row-major view (nls_dual_kernels.h, k_gemm): C[j][r] = sum_i A[j][i] B[i][r] with A = U^T (U column-major: lda = P), B = G^T (G column-major),
// C = R^T (R column-major).  M = k, N = m, K = k from device memory: tiles beyond them return at once.  grid = (P / 128, P / 128, merges)
__global__ void __launch_bounds__(Cfg4::NTHREADS, 2) k_dc_gemm(Level L, const double* U, const double* G, double* R, const MergeInfo* info) {
  using C = Cfg4;
  extern __shared__ double smem[];
  const int mi = blockIdx.z, b0 = L.b0(mi);
  if (b0 >= L.n) return;
  const int k = info[mi].k, m = L.b1(mi) - b0;
  const long row0 = (long)blockIdx.y * BM, col0 = (long)blockIdx.x * BN;
  if (row0 >= k || col0 >= m) return;
  const long slot = (long)mi * L.P * L.P;
  // The kept columns are grouped (k_dc_setup): [upper block only | dense | lower block only].  Rows of the upper block see zeros in the third
  // group, rows of the lower block in the first: a tile that lies inside one block walks only its part of K (dlaed3's two products).
  const int n1 = L.mid(mi) - b0, k12 = info[mi].k1 + info[mi].k2;
  int kb = 0, ke = k;
  if (col0 + BN <= n1) ke = k12;
  else if (col0 >= n1) kb = (info[mi].k1 / BK) * BK;
  v4d acc[C::MT][C::NTL];
  zero_acc(acc);
  MMajorLoader<C::NTHREADS, BM> la{U + slot, L.P, row0};
  KMajorLoader<C::NTHREADS, BN> lb{G + slot, L.P, col0};
  mainloop_real<C, false>(acc, la, lb, (long)kb, (ke - kb + BK - 1) / BK, smem);
#pragma unroll
  for (int mt = 0; mt < C::MT; ++mt)
#pragma unroll
    for (int r = 0; r < 4; ++r) {
      const long row = row0 + C::acc_row(mt, r);
#pragma unroll
      for (int nt = 0; nt < C::NTL; ++nt) R[slot + row * L.P + col0 + C::acc_col(nt)] = acc[mt][nt][r];
    }
}

// ---- merge: places in the merged ascending order -----------------------------------------------------------------------------------------
// pos[t] for t < k: the new eigenvalue t; for k <= t < k + ndefl: the deflated pole didx[t - k].  Order: by value, new before deflated on ties,
// then by index.  Also writes the eigenvalues to their places.   grid = (ceil(m / 256), merges)
__global__ void __launch_bounds__(256) k_dc_place(Level L, const double* lamnew, const double* ds, const int* didx, const MergeInfo* info, int* pos, double* lam_out) {
  __shared__ double dvs[256];  // a tile of the deflated poles' values (round 5: every thread used to fetch every deflated pole itself, two
                               // dependent loads per pole: 0.7 ms for a merge with a few thousand of them)
  const int mi = blockIdx.y, b0 = L.b0(mi);
  if (b0 >= L.n) return;
  const int k = info[mi].k, nd = info[mi].ndefl;
  if ((int)blockIdx.x * 256 >= k + nd) return;  // uniform
  const int t = blockIdx.x * 256 + threadIdx.x;
  const bool live = t < k + nd;
  int p = 0;
  double v = 0.0;
  const int q0 = t - k;  // (deflated pole: its own number)
  if (live) {
    if (t < k) {
      v = lamnew[b0 + t];
      p = t;  // the new eigenvalues are strictly increasing
    } else {
      v = ds[b0 + didx[b0 + q0]];
      int lo = 0, hi = k;  // #(new <= v)
      while (lo < hi) {
        const int c = (lo + hi) >> 1;
        if (lamnew[b0 + c] <= v) lo = c + 1; else hi = c;
      }
      p = lo;
    }
  }
  for (int qb = 0; qb < nd; qb += 256) {
    __syncthreads();
    if (qb + (int)threadIdx.x < nd) dvs[threadIdx.x] = ds[b0 + didx[b0 + qb + threadIdx.x]];
    __syncthreads();
    const int cnt = min(256, nd - qb);
    if (live) {
      if (t < k) {
        for (int q = 0; q < cnt; ++q) p += dvs[q] < v ? 1 : 0;
      } else {
        for (int q = 0; q < cnt; ++q) {
          const double u = dvs[q];
          p += (u < v || (u == v && qb + q < q0)) ? 1 : 0;
        }
      }
    }
  }
  if (!live) return;
  pos[b0 + t] = p;
  lam_out[b0 + p] = v;
}

// ---- merge: columns to their places -------------------------------------------------------------------------------------------------------
// grid = (row chunks of 256, column groups of 8, merges)
__global__ void __launch_bounds__(256) k_dc_scatter(Level L, const double* R, const double* Qin, double* Qout, long ldq, const int* src, const int* didx, const int* pos,
                                                    const MergeInfo* info) {
  const int mi = blockIdx.z, b0 = L.b0(mi);
  if (b0 >= L.n) return;
  const int k = info[mi].k, nd = info[mi].ndefl, m = L.b1(mi) - b0;
  const int r = blockIdx.x * 256 + threadIdx.x;
  if (r >= m) return;
  const double* Rs = R + (long)mi * L.P * L.P;
  for (int t = blockIdx.y * 8; t < blockIdx.y * 8 + 8 && t < k + nd; ++t) {
    const int p = pos[b0 + t];
    double v;
    if (t < k)
      v = Rs[r + (long)t * L.P];
    else
      v = Qin[(long)(b0 + r) + (long)(b0 + src[b0 + didx[b0 + t - k]]) * ldq];
    Qout[(long)(b0 + r) + (long)(b0 + p) * ldq] = v;
  }
}

}  // namespace dc
}  // namespace nls
