// alpha = cho_solve(L, y) for the REAL lower Cholesky factor of the dual fit (_neo_ls_svm.py:313-314): L x = y forwards, L^T alpha = x backwards,
// in place in one vector.  rocblas_dtrsv takes 5.2 ms per solve at n = 10^4 (400 MB of matrix at 80 GB/s: a single stream of dependent
// blocks); here both directions go in outer blocks of 256 unknowns with two launches per block - ONE workgroup finishes the block's own triangle
// (panels of 32, the 32 x 32 triangles across the lanes of one wave), and a full-width launch handles everything outside the block:
//   forwards : k_trsv_fwd_update   y[r] -= sum_{c in block} L[r][c] x[c]  for the rows below the block (one row per thread: a column of L is
//              contiguous, consecutive threads read consecutive rows);
//   backwards: k_trsv_bwd_outer_sum s[c] = sum_{r below the block} L[r][c] alpha[r] (one workgroup per column).
// (The complex counterpart for the primal fit lives in nls_zpotrf.h: its forward half is carried through the factorisation.)
#pragma once
#include <hip/hip_runtime.h>

namespace nls {
namespace trsv {

constexpr int PB = 32;    // panel
constexpr int OB = 256;   // outer block

// One workgroup: x[K0 .. K0 + W) of L x = y within the block (rows / columns K0 .. K0 + W of L), in place in y.
__global__ void __launch_bounds__(256) k_trsv_fwd_block(const double* __restrict__ L, long ldl, int K0, int W, double* __restrict__ y) {
  __shared__ double part[8][PB];
  __shared__ double blk[PB][PB + 1];
  __shared__ double sol[PB];
  const int tid = threadIdx.x, c = tid & (PB - 1), g = tid >> 5;
  const int np = (W + PB - 1) / PB;
  for (int p = 0; p < np; ++p) {
    const int k0 = K0 + p * PB, w = min(PB, K0 + W - k0);
    // row k0 + c of the panel: sum over the block's earlier columns t in [K0, k0)
    // (four products in flight per thread, the block's entries requested together: one load per loop trip is one memory round trip per trip)
    double s = 0.0;
    if (c < w) {
      double s1 = 0.0, s2 = 0.0, s3 = 0.0;
      int t = K0 + g;
      for (; t + 24 < k0; t += 32) {
        const double l0 = L[(long)(k0 + c) + (long)t * ldl], l1 = L[(long)(k0 + c) + (long)(t + 8) * ldl];
        const double l2 = L[(long)(k0 + c) + (long)(t + 16) * ldl], l3 = L[(long)(k0 + c) + (long)(t + 24) * ldl];
        const double y0 = y[t], y1 = y[t + 8], y2 = y[t + 16], y3 = y[t + 24];
        s += l0 * y0;
        s1 += l1 * y1;
        s2 += l2 * y2;
        s3 += l3 * y3;
      }
      for (; t < k0; t += 8) s += L[(long)(k0 + c) + (long)t * ldl] * y[t];
      s = (s + s1) + (s2 + s3);
    }
    part[g][c] = s;
    {
      double bv[PB * PB / 256];
#pragma unroll
      for (int it = 0; it < PB * PB / 256; ++it) {
        const int idx = tid + 256 * it, rr = idx % PB, cc = idx / PB;
        bv[it] = (rr < w && cc < w && rr >= cc) ? L[(long)(k0 + rr) + (long)(k0 + cc) * ldl] : (rr == cc ? 1.0 : 0.0);
      }
#pragma unroll
      for (int it = 0; it < PB * PB / 256; ++it) {
        const int idx = tid + 256 * it;
        blk[idx % PB][idx / PB] = bv[it];
      }
    }
    __syncthreads();
    if (tid < 64) {
      double acc = 0.0;
      if (tid < w) {
        acc = y[k0 + c];
        for (int q = 0; q < 8; ++q) acc -= part[q][c];
      }
      const double rd = 1.0 / blk[tid & (PB - 1)][tid & (PB - 1)];  // (one division per unknown, all at once, instead of one per step of the chain)
      for (int t = 0; t < PB; ++t) {
        if (tid == t) sol[t] = acc * rd;
        __builtin_amdgcn_fence(__ATOMIC_RELEASE, "wavefront");
        __builtin_amdgcn_wave_barrier();
        if (tid > t && tid < PB) acc -= blk[tid][t] * sol[t];
      }
      if (tid < w) y[k0 + c] = sol[c];
    }
    __threadfence_block();
    __syncthreads();
  }
}

// y[r] -= sum_{c in [K0, K0 + W)} L[r][c] x[c] for r in [K0 + W, n): one row per thread.
__global__ void __launch_bounds__(256) k_trsv_fwd_update(const double* __restrict__ L, long ldl, int n, int K0, int W, double* __restrict__ y) {
  __shared__ double xs[OB];
  for (int i = threadIdx.x; i < W; i += 256) xs[i] = y[K0 + i];
  __syncthreads();
  const int r = K0 + W + blockIdx.x * 256 + threadIdx.x;
  if (r >= n) return;
  double s = 0.0;
#pragma unroll 8
  for (int c = 0; c < W; ++c) s += L[(long)r + (long)(K0 + c) * ldl] * xs[c];
  y[r] -= s;
}

// s[c - K0] = sum_{r >= K0 + W} L[r][c] alpha[r], one workgroup per column c of the block.
__global__ void __launch_bounds__(256) k_trsv_bwd_outer_sum(const double* __restrict__ L, long ldl, int n, int K0, int W, const double* __restrict__ y,
                                                            double* __restrict__ sums) {
  __shared__ double red[4];
  const int c = K0 + blockIdx.x;
  double s = 0.0;
  {
    const double* col = L + (long)c * ldl;
    double s1 = 0.0, s2 = 0.0, s3 = 0.0;
    int r = K0 + W + threadIdx.x;
    for (; r + 768 < n; r += 1024) {  // four products in flight per thread
      const double l0 = col[r], l1 = col[r + 256], l2 = col[r + 512], l3 = col[r + 768];
      const double y0 = y[r], y1 = y[r + 256], y2 = y[r + 512], y3 = y[r + 768];
      s += l0 * y0;
      s1 += l1 * y1;
      s2 += l2 * y2;
      s3 += l3 * y3;
    }
    for (; r < n; r += 256) s += col[r] * y[r];
    s = (s + s1) + (s2 + s3);
  }
#pragma unroll
  for (int o = 32; o > 0; o >>= 1) s += __shfl_xor(s, o, 64);
  if ((threadIdx.x & 63) == 0) red[threadIdx.x >> 6] = s;
  __syncthreads();
  if (threadIdx.x == 0) sums[blockIdx.x] = (red[0] + red[1]) + (red[2] + red[3]);
}

// One workgroup: alpha[K0 .. K0 + W) of L^T alpha = x within the block, given the sums over the rows below it; in place in y.
__global__ void __launch_bounds__(256) k_trsv_bwd_block(const double* __restrict__ L, long ldl, int K0, int W, double* __restrict__ y,
                                                        const double* __restrict__ sums) {
  __shared__ double part[8][PB];
  __shared__ double blk[PB][PB + 1];
  __shared__ double sol[PB];
  const int tid = threadIdx.x, c = tid & (PB - 1), g = tid >> 5;
  const int bend = K0 + W, np = (W + PB - 1) / PB;
  for (int p = np - 1; p >= 0; --p) {
    const int k0 = K0 + p * PB, w = min(PB, bend - k0), rend = k0 + w;
    double s = 0.0;
    if (c < w) {
      double s1 = 0.0, s2 = 0.0, s3 = 0.0;
      int r = rend + g;
      for (; r + 24 < bend; r += 32) {
        const double* col = L + (long)(k0 + c) * ldl;
        const double l0 = col[r], l1 = col[r + 8], l2 = col[r + 16], l3 = col[r + 24];
        const double y0 = y[r], y1 = y[r + 8], y2 = y[r + 16], y3 = y[r + 24];
        s += l0 * y0;
        s1 += l1 * y1;
        s2 += l2 * y2;
        s3 += l3 * y3;
      }
      for (; r < bend; r += 8) s += L[(long)r + (long)(k0 + c) * ldl] * y[r];
      s = (s + s1) + (s2 + s3);
    }
    part[g][c] = s;
    {
      double bv[PB * PB / 256];
#pragma unroll
      for (int it = 0; it < PB * PB / 256; ++it) {
        const int idx = tid + 256 * it, rr = idx % PB, cc = idx / PB;
        bv[it] = (rr < w && cc < w && rr >= cc) ? L[(long)(k0 + rr) + (long)(k0 + cc) * ldl] : (rr == cc ? 1.0 : 0.0);
      }
#pragma unroll
      for (int it = 0; it < PB * PB / 256; ++it) {
        const int idx = tid + 256 * it;
        blk[idx % PB][idx / PB] = bv[it];
      }
    }
    __syncthreads();
    if (tid < 64) {
      double acc = 0.0;
      if (tid < w) {
        acc = y[k0 + c];
        if (sums) acc -= sums[k0 - K0 + c];
        for (int q = 0; q < 8; ++q) acc -= part[q][c];
      }
      const double rd = 1.0 / blk[tid & (PB - 1)][tid & (PB - 1)];
      for (int t = PB - 1; t >= 0; --t) {
        if (tid == t) sol[t] = acc * rd;
        __builtin_amdgcn_fence(__ATOMIC_RELEASE, "wavefront");
        __builtin_amdgcn_wave_barrier();
        if (tid < t) acc -= blk[t][tid] * sol[t];
      }
      if (tid < w) y[k0 + c] = sol[c];
    }
    __threadfence_block();
    __syncthreads();
  }
}

}  // namespace trsv
}  // namespace nls
