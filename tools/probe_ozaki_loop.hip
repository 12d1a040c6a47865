// Gate probe for fp64 products on the int8 matrix cores (Ozaki splitting): the INNER LOOP an exact fp64-equivalent product would
// run - 8 x 8 int8 slices, the 36 slice pairs with s + t <= 7, 8 diagonal i32 accumulators per 16 x 16 tile, a 32 x 32 tile per wave (128 accumulator
// registers: what the register file allows at two waves per SIMD), operand fragments read from LDS in fragment order - WITHOUT global loads, LDS refills,
// slicing or recombination.  It is the most optimistic form of the kernel: if this loop does not reach 1.25 x the 73 TFLOP/s the fp64 3M kernels execute
// (= 3.28 POP/s of int8 work: 36 slice-pair products per fp64 product), nothing built on it can.
//   hipcc -O3 --offload-arch=gfx950 tools/probe_ozaki_loop.hip -o /tmp/probe_ozaki_loop && /tmp/probe_ozaki_loop
#include <hip/hip_runtime.h>
#include <cstdio>
#include <cstdlib>
#include <vector>
#define CK(x) do { hipError_t e_ = (x); if (e_ != hipSuccess) { fprintf(stderr, "HIP error %s at %d\n", hipGetErrorString(e_), __LINE__); exit(1);} } while (0)
typedef int v4i __attribute__((ext_vector_type(4)));

// LDS image of one K step (64 k) of a 64 x 64 workgroup tile: A[slice][m-tile 0..3][lane], B[slice][n-tile 0..3][lane], 16 bytes per lane = the fragment
// of v_mfma_i32_16x16x64_i8.  FROM_LDS = false: operands stay in registers (the pipe's own ceiling for this accumulator pattern).
template <bool FROM_LDS>
__global__ void __launch_bounds__(256, 2) ozaki_loop(int* out, const int* in, unsigned long long* clk, int iters) {
  __shared__ v4i As[8][4][64], Bs[8][4][64];
  const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6, wm = wave >> 1, wn = wave & 1;
  for (int i = threadIdx.x; i < 8 * 4 * 64; i += 256) {
    (&As[0][0][0])[i] = *(const v4i*)(in + 4 * (i % 1024));
    (&Bs[0][0][0])[i] = *(const v4i*)(in + 4 * ((i + 517) % 1024));
  }
  __syncthreads();
  v4i acc[2][2][8];
#pragma unroll
  for (int m = 0; m < 2; ++m)
#pragma unroll
    for (int n = 0; n < 2; ++n)
#pragma unroll
      for (int q = 0; q < 8; ++q) acc[m][n][q] = v4i{0, 0, 0, 0};
  v4i b[2][8], a[2];
#pragma unroll
  for (int n = 0; n < 2; ++n)
#pragma unroll
    for (int t = 0; t < 8; ++t) b[n][t] = Bs[t][2 * wn + n][lane];
  a[0] = As[0][2 * wm][lane];
  a[1] = As[0][2 * wm + 1][lane];
  const unsigned long long t0 = __builtin_amdgcn_s_memtime(), r0 = __builtin_amdgcn_s_memrealtime();
  for (int it = 0; it < iters; ++it) {
    asm volatile("" ::: "memory");  // the LDS reads below belong to this K step
    if (FROM_LDS) {
#pragma unroll
      for (int n = 0; n < 2; ++n)
#pragma unroll
        for (int t = 0; t < 8; ++t) b[n][t] = Bs[t][2 * wn + n][lane];
    }
#pragma unroll
    for (int s = 0; s < 8; ++s) {
      if (FROM_LDS) {
        a[0] = As[s][2 * wm][lane];
        a[1] = As[s][2 * wm + 1][lane];
      }
#pragma unroll
      for (int t = 0; t + s < 8; ++t)
#pragma unroll
        for (int m = 0; m < 2; ++m)
#pragma unroll
          for (int n = 0; n < 2; ++n) acc[m][n][s + t] = __builtin_amdgcn_mfma_i32_16x16x64_i8(a[m], b[n][t], acc[m][n][s + t], 0, 0, 0);
    }
  }
  const unsigned long long t1 = __builtin_amdgcn_s_memtime(), r1 = __builtin_amdgcn_s_memrealtime();
  int sum = 0;
#pragma unroll
  for (int m = 0; m < 2; ++m)
#pragma unroll
    for (int n = 0; n < 2; ++n)
#pragma unroll
      for (int q = 0; q < 8; ++q) sum += acc[m][n][q][0] + acc[m][n][q][1] + acc[m][n][q][2] + acc[m][n][q][3];
  out[blockIdx.x * blockDim.x + threadIdx.x] = sum;
  if (threadIdx.x == 0) { clk[2 * blockIdx.x] = t1 - t0; clk[2 * blockIdx.x + 1] = r1 - r0; }
}

template <bool FROM_LDS>
static void run(int cus, int* out, const int* in, unsigned long long* clk) {
  const int blocks = 2 * cus, iters = 4000;
  hipEvent_t e0, e1; CK(hipEventCreate(&e0)); CK(hipEventCreate(&e1));
  double best = 1e30;
  for (int rep = 0; rep < 4; ++rep) {
    CK(hipEventRecord(e0));
    hipLaunchKernelGGL((ozaki_loop<FROM_LDS>), dim3(blocks), dim3(256), 0, 0, out, in, clk, iters);
    CK(hipEventRecord(e1)); CK(hipEventSynchronize(e1));
    float ms; CK(hipEventElapsedTime(&ms, e0, e1));
    if (rep > 0 && ms < best) best = ms;
  }
  std::vector<unsigned long long> hc(2 * blocks);
  CK(hipMemcpy(hc.data(), clk, hc.size() * 8, hipMemcpyDeviceToHost));
  const double ghz = (double)hc[0] / (double)hc[1] * 0.1;
  const double mfma = (double)blocks * 4 * iters * 144.0, pops = mfma * 2.0 * 16 * 16 * 64 / (best * 1e-3) / 1e15;
  printf("%-34s %7.3f ms  %6.3f POP/s int8  = %6.1f TFLOP/s fp64-equivalent (36 slice pairs)  clock %.2f GHz  %5.1f cycles per MFMA and SIMD  -> %.2f x the 73 TFLOP/s executed today (gate: 1.25 x)\n",
         FROM_LDS ? "fragments from LDS (2 waves/SIMD)" : "fragments in registers (2 waves/SIMD)", best, pops, pops * 1e3 / 36.0, ghz,
         (double)hc[0] / ((double)iters * 144.0 * 2.0), pops * 1e3 / 36.0 / 73.0);
}

int main() {
  hipDeviceProp_t prop; CK(hipGetDeviceProperties(&prop, 0));
  const int cus = prop.multiProcessorCount;
  int *out, *in; unsigned long long* clk;
  CK(hipMalloc(&out, 4 * 256 * 2 * cus)); CK(hipMalloc(&in, 4096 * 4)); CK(hipMalloc(&clk, 32 * cus));
  std::vector<int> h(4096);
  for (auto& v : h) v = rand();  // random int8 operands (the clock the chip holds depends on the data)
  CK(hipMemcpy(in, h.data(), h.size() * 4, hipMemcpyHostToDevice));
  printf("%s, %d CUs\n", prop.name, cus);
  run<false>(cus, out, in, clk);
  run<true>(cus, out, in, clk);
  return 0;
}
