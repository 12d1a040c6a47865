// Feasibility probe for a fused (persistent) zhetrd panel kernel: cost of a device-wide barrier between 256 / 512
// resident workgroups, and the rate of a Hermitian matrix-vector product that re-reads a ~134 MB lower triangle
// (fits the 256 MB Infinity Cache) every step.
#include <hip/hip_runtime.h>
#include <cstdio>
#include <cstdlib>
#define CK(x) do { hipError_t e_ = (x); if (e_ != hipSuccess) { fprintf(stderr, "HIP %s line %d\n", hipGetErrorString(e_), __LINE__); exit(1);} } while (0)
typedef double v2d __attribute__((ext_vector_type(2)));

__device__ __forceinline__ void grid_barrier(unsigned* counter, unsigned nblocks, unsigned& epoch) {
  __syncthreads();
  if (threadIdx.x == 0) {
    ++epoch;
    __threadfence();
    unsigned target = epoch * nblocks;
    atomicAdd(counter, 1u);
    while (__hip_atomic_load(counter, __ATOMIC_ACQUIRE, __HIP_MEMORY_SCOPE_AGENT) < target) __builtin_amdgcn_s_sleep(1);
    __threadfence();
  }
  __syncthreads();
}

__global__ void k_barriers(unsigned* counter, int reps, double* sink) {
  unsigned epoch = 0;
  double x = threadIdx.x;
  for (int r = 0; r < reps; ++r) {
    grid_barrier(counter, gridDim.x, epoch);
    x = x * 1.0000001 + 1.0;
  }
  if (x == -1.0) sink[0] = x;
}

// y = A x for complex Hermitian A given by its lower triangle (column-major, ld = n), rows/cols [j0, n).
// Simple full-read variant: each workgroup owns 64-row strips (round robin) and reads the row strip through the
// lower triangle for cols <= row and the transposed (conjugated) column for cols > row: here we just stream the lower
// part twice-used via two passes to measure achievable read rate of the triangle from MALL/HBM.
__global__ void k_hemv_lower(const v2d* A, const v2d* x, v2d* y, int n, int j0, int steps) {
  // each thread handles one row i (strided), loops cols j0..i  (row-wise reads are strided by ld -> use column sweep instead)
  // column sweep: block b takes columns c = j0 + b, j0 + b + grid, ...; threads stride over rows c..n-1 (coalesced)
  __shared__ double red[256];
  for (int s = 0; s < steps; ++s) {
    int jj = j0 + s;  // trailing matrix shrinks by one each step
    for (int c = jj + blockIdx.x; c < n; c += gridDim.x) {
      const v2d xc = x[c];
      double ar = 0, ai = 0;
      for (int r = c + threadIdx.x; r < n; r += blockDim.x) {
        const v2d a = A[(size_t)c * n + r];
        const v2d xr = x[r];
        // contribution to y[c] of conj(a) * x[r]  (upper part via symmetry)
        ar += a.x * xr.x + a.y * xr.y;
        ai += a.x * xr.y - a.y * xr.x;
        // contribution to y[r] of a * x[c]: accumulate with atomics
        atomicAdd(reinterpret_cast<double*>(&y[r]), a.x * xc.x - a.y * xc.y);
        atomicAdd(reinterpret_cast<double*>(&y[r]) + 1, a.x * xc.y + a.y * xc.x);
      }
      red[threadIdx.x] = ar; __syncthreads();
      for (int o = 128; o > 0; o >>= 1) { if (threadIdx.x < o) red[threadIdx.x] += red[threadIdx.x + o]; __syncthreads(); }
      if (threadIdx.x == 0) atomicAdd(reinterpret_cast<double*>(&y[c]), red[0]);
      __syncthreads();
      red[threadIdx.x] = ai; __syncthreads();
      for (int o = 128; o > 0; o >>= 1) { if (threadIdx.x < o) red[threadIdx.x] += red[threadIdx.x + o]; __syncthreads(); }
      if (threadIdx.x == 0) atomicAdd(reinterpret_cast<double*>(&y[c]) + 1, red[0]);
      __syncthreads();
    }
  }
}

// Pure streaming read of the lower triangle (upper bound for any hemv): sum of all entries, same column sweep.
__global__ void k_tri_read(const v2d* A, double* out, int n, int j0, int steps) {
  double acc = 0;
  for (int s = 0; s < steps; ++s) {
    int jj = j0 + s;
    for (int c = jj + blockIdx.x; c < n; c += gridDim.x)
      for (int r = c + threadIdx.x; r < n; r += blockDim.x) { const v2d a = A[(size_t)c * n + r]; acc += a.x + a.y; }
  }
  if (acc == -1.0) out[0] = acc;
}

int main() {
  unsigned* counter; double* sink;
  CK(hipMalloc(&counter, 4)); CK(hipMalloc(&sink, 8));
  hipEvent_t e0, e1; CK(hipEventCreate(&e0)); CK(hipEventCreate(&e1));
  for (int blocks : {256, 512, 1024}) {
    for (int threads : {64, 256}) {
      CK(hipMemset(counter, 0, 4));
      const int reps = 2000;
      k_barriers<<<blocks, threads>>>(counter, 10, sink);
      CK(hipMemset(counter, 0, 4));
      CK(hipDeviceSynchronize());
      CK(hipEventRecord(e0));
      k_barriers<<<blocks, threads>>>(counter, reps, sink);
      CK(hipEventRecord(e1)); CK(hipEventSynchronize(e1));
      float ms; CK(hipEventElapsedTime(&ms, e0, e1));
      printf("grid barrier, %4d blocks x %3d threads: %.2f us per barrier\n", blocks, threads, ms * 1e3 / reps);
    }
  }
  const int n = 4097;
  v2d *A, *x, *y;
  CK(hipMalloc(&A, (size_t)n * n * 16)); CK(hipMalloc(&x, n * 16)); CK(hipMalloc(&y, n * 16));
  CK(hipMemset(A, 0, (size_t)n * n * 16)); CK(hipMemset(x, 0, n * 16)); CK(hipMemset(y, 0, n * 16));
  for (int j0 : {0, 1024, 2048, 3072}) {
    const int steps = 32;
    double bytes = 0;
    for (int s = 0; s < steps; ++s) { double m = n - j0 - s; bytes += m * (m + 1) / 2 * 16; }
    for (int blocks : {512, 2048}) {
      k_tri_read<<<blocks, 256>>>(A, sink, n, j0, steps);
      CK(hipEventRecord(e0));
      k_tri_read<<<blocks, 256>>>(A, sink, n, j0, steps);
      CK(hipEventRecord(e1)); CK(hipEventSynchronize(e1));
      float ms; CK(hipEventElapsedTime(&ms, e0, e1));
      printf("triangle read   j0=%4d (%.0f MB per step) %4d blocks: %.1f us per step, %.2f TB/s\n", j0, bytes / steps / 1e6, blocks, ms * 1e3 / steps, bytes / ms / 1e9);
      k_hemv_lower<<<blocks, 256>>>(A, x, y, n, j0, steps);
      CK(hipEventRecord(e0));
      k_hemv_lower<<<blocks, 256>>>(A, x, y, n, j0, steps);
      CK(hipEventRecord(e1)); CK(hipEventSynchronize(e1));
      CK(hipEventElapsedTime(&ms, e0, e1));
      printf("hemv (atomics)  j0=%4d                      %4d blocks: %.1f us per step, %.2f TB/s\n", j0, blocks, ms * 1e3 / steps, bytes / ms / 1e9);
    }
  }
  return 0;
}
