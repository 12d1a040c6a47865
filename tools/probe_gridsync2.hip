// Grid barrier variants for a persistent panel kernel (one workgroup per CU): flat atomic counter, two-level tree,
// flag array gathered by workgroup 0.
#include <hip/hip_runtime.h>
#include <cstdio>
#include <cstdlib>
#define CK(x) do { hipError_t e_ = (x); if (e_ != hipSuccess) { fprintf(stderr, "HIP %s line %d\n", hipGetErrorString(e_), __LINE__); exit(1);} } while (0)

__device__ __forceinline__ unsigned ld_acq(const unsigned* p) { return __hip_atomic_load(p, __ATOMIC_ACQUIRE, __HIP_MEMORY_SCOPE_AGENT); }
__device__ __forceinline__ unsigned ld_rlx(const unsigned* p) { return __hip_atomic_load(p, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT); }
__device__ __forceinline__ void st_rel(unsigned* p, unsigned v) { __hip_atomic_store(p, v, __ATOMIC_RELEASE, __HIP_MEMORY_SCOPE_AGENT); }

template <int KIND, int SLEEP>
__device__ __forceinline__ void grid_barrier(unsigned* ws, unsigned nblocks, unsigned& epoch) {
  __syncthreads();
  ++epoch;
  if (KIND == 0) {  // flat counter
    if (threadIdx.x == 0) {
      __hip_atomic_fetch_add(ws, 1u, __ATOMIC_RELEASE, __HIP_MEMORY_SCOPE_AGENT);
      while (ld_acq(ws) < epoch * nblocks) { if (SLEEP) __builtin_amdgcn_s_sleep(SLEEP); }
    }
  } else if (KIND == 3) {  // flat counter, relaxed polling, one fence on each side
    if (threadIdx.x == 0) {
      __builtin_amdgcn_fence(__ATOMIC_RELEASE, "agent");
      __hip_atomic_fetch_add(ws, 1u, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
      while (ld_rlx(ws) < epoch * nblocks) { if (SLEEP) __builtin_amdgcn_s_sleep(SLEEP); }
      __builtin_amdgcn_fence(__ATOMIC_ACQUIRE, "agent");
    }
  } else if (KIND == 4) {  // tree with relaxed polling
    if (threadIdx.x == 0) {
      const unsigned g = blockIdx.x >> 4, ng = (nblocks + 15) >> 4;
      const unsigned gsize = (g + 1) * 16 <= nblocks ? 16 : nblocks - g * 16;
      __builtin_amdgcn_fence(__ATOMIC_RELEASE, "agent");
      unsigned old = __hip_atomic_fetch_add(ws + 16 + g * 16, 1u, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
      if (old + 1 == epoch * gsize) __hip_atomic_fetch_add(ws, 1u, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
      while (ld_rlx(ws) < epoch * ng) { if (SLEEP) __builtin_amdgcn_s_sleep(SLEEP); }
      __builtin_amdgcn_fence(__ATOMIC_ACQUIRE, "agent");
    }
  } else if (KIND == 1) {  // tree: groups of 16 -> ws[16 + g * 16], root ws[0]
    if (threadIdx.x == 0) {
      const unsigned g = blockIdx.x >> 4, ng = (nblocks + 15) >> 4;
      const unsigned gsize = (g + 1) * 16 <= nblocks ? 16 : nblocks - g * 16;
      unsigned old = __hip_atomic_fetch_add(ws + 16 + g * 16, 1u, __ATOMIC_ACQ_REL, __HIP_MEMORY_SCOPE_AGENT);
      if (old + 1 == epoch * gsize) __hip_atomic_fetch_add(ws, 1u, __ATOMIC_RELEASE, __HIP_MEMORY_SCOPE_AGENT);
      while (ld_acq(ws) < epoch * ng) { if (SLEEP) __builtin_amdgcn_s_sleep(SLEEP); }
    }
  } else {  // flags: block b writes ws[64 + b] = epoch; block 0 gathers with its threads, then publishes ws[0] = epoch
    if (threadIdx.x == 0) st_rel(ws + 64 + blockIdx.x, epoch);
    if (blockIdx.x == 0) {
      for (unsigned b = threadIdx.x; b < nblocks; b += blockDim.x)
        while (ld_acq(ws + 64 + b) < epoch) { if (SLEEP) __builtin_amdgcn_s_sleep(SLEEP); }
      __syncthreads();
      if (threadIdx.x == 0) st_rel(ws, epoch);
    } else if (threadIdx.x == 0) {
      while (ld_acq(ws) < epoch) { if (SLEEP) __builtin_amdgcn_s_sleep(SLEEP); }
    }
  }
  __syncthreads();
}

template <int KIND, int SLEEP>
__global__ void k_barriers(unsigned* ws, int reps, double* sink) {
  unsigned epoch = 0;
  double x = threadIdx.x;
  for (int r = 0; r < reps; ++r) {
    grid_barrier<KIND, SLEEP>(ws, gridDim.x, epoch);
    x = x * 1.0000001 + 1.0;
  }
  if (x == -1.0) sink[0] = x;
}

template <int KIND, int SLEEP>
void run(const char* name, unsigned* ws, double* sink, int blocks, int threads) {
  hipEvent_t e0, e1; CK(hipEventCreate(&e0)); CK(hipEventCreate(&e1));
  const int reps = 3000;
  CK(hipMemset(ws, 0, 8192));
  CK(hipEventRecord(e0));
  k_barriers<KIND, SLEEP><<<blocks, threads>>>(ws, reps, sink);
  CK(hipEventRecord(e1)); CK(hipEventSynchronize(e1));
  float ms; CK(hipEventElapsedTime(&ms, e0, e1));
  printf("%-28s sleep %d  %4d blocks x %3d threads: %6.2f us per barrier\n", name, SLEEP, blocks, threads, ms * 1e3 / reps);
}

int main() {
  unsigned* ws; double* sink;
  CK(hipMalloc(&ws, 8192)); CK(hipMalloc(&sink, 8));
  for (int blocks : {64, 128, 256}) {
    run<0, 1>("flat counter", ws, sink, blocks, 256);
    run<0, 0>("flat counter", ws, sink, blocks, 256);
    run<1, 1>("tree 16 x 16", ws, sink, blocks, 256);
    run<1, 0>("tree 16 x 16", ws, sink, blocks, 256);
    run<3, 1>("flat, relaxed poll", ws, sink, blocks, 256);
    run<3, 4>("flat, relaxed poll", ws, sink, blocks, 256);
    run<4, 1>("tree, relaxed poll", ws, sink, blocks, 256);
    run<4, 4>("tree, relaxed poll", ws, sink, blocks, 256);
  }
  return 0;
}
