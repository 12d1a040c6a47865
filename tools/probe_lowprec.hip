// Issue rates of the low-precision MFMAs an Ozaki-scheme fp64 emulation would run on (next-round feasibility probe):
// v_mfma_i32_16x16x64_i8 and v_mfma_f32_16x16x32_bf16, random operands, 1/2 waves per SIMD, in-kernel clock.
#include <hip/hip_runtime.h>
#include <cstdio>
#include <cstdlib>
#include <vector>
#define CK(x) do { hipError_t e_ = (x); if (e_ != hipSuccess) { fprintf(stderr, "HIP error %s at %d\n", hipGetErrorString(e_), __LINE__); exit(1);} } while (0)
typedef int v4i __attribute__((ext_vector_type(4)));
typedef float v4f __attribute__((ext_vector_type(4)));
typedef __bf16 v8bf __attribute__((ext_vector_type(8)));

template <int KIND, int NACC>
__global__ void rate(int* out, const int* in, unsigned long long* clk, int iters) {
  v4i acci[NACC]; v4f accf[NACC];
#pragma unroll
  for (int i = 0; i < NACC; ++i) { acci[i] = v4i{0, 0, 0, 0}; accf[i] = v4f{0, 0, 0, 0}; }
  v4i a[2], b[2];
#pragma unroll
  for (int i = 0; i < 2; ++i) { a[i] = *(const v4i*)(in + threadIdx.x * 16 + 4 * i); b[i] = *(const v4i*)(in + threadIdx.x * 16 + 8 + 4 * i); }
  unsigned long long t0 = __builtin_amdgcn_s_memtime(), r0 = __builtin_amdgcn_s_memrealtime();
  for (int it = 0; it < iters; ++it) {
#pragma unroll
    for (int i = 0; i < NACC; ++i) {
      if (KIND == 0) acci[i] = __builtin_amdgcn_mfma_i32_16x16x64_i8(a[i & 1], b[(i >> 1) & 1], acci[i], 0, 0, 0);
      else accf[i] = __builtin_amdgcn_mfma_f32_16x16x32_bf16(__builtin_bit_cast(v8bf, a[i & 1]), __builtin_bit_cast(v8bf, b[(i >> 1) & 1]), accf[i], 0, 0, 0);
    }
  }
  unsigned long long t1 = __builtin_amdgcn_s_memtime(), r1 = __builtin_amdgcn_s_memrealtime();
  int s = 0;
#pragma unroll
  for (int i = 0; i < NACC; ++i) s += acci[i][0] + acci[i][1] + acci[i][2] + acci[i][3] + (int)(accf[i][0] + accf[i][1] + accf[i][2] + accf[i][3]);
  out[blockIdx.x * blockDim.x + threadIdx.x] = s;
  if (threadIdx.x == 0) { clk[2 * blockIdx.x] = t1 - t0; clk[2 * blockIdx.x + 1] = r1 - r0; }
}

template <int KIND, int NACC>
void run(int cus, int wps, int* out, int* in, unsigned long long* clk) {
  int threads = 256 * wps, blocks = cus, iters = 20000;
  hipEvent_t e0, e1; CK(hipEventCreate(&e0)); CK(hipEventCreate(&e1));
  float best = 1e30f;
  for (int rep = 0; rep < 3; ++rep) {
    CK(hipEventRecord(e0));
    rate<KIND, NACC><<<blocks, threads>>>(out, in, clk, iters);
    CK(hipEventRecord(e1)); CK(hipEventSynchronize(e1));
    float ms; CK(hipEventElapsedTime(&ms, e0, e1)); if (ms < best) best = ms;
  }
  std::vector<unsigned long long> hc(2 * blocks);
  CK(hipMemcpy(hc.data(), clk, hc.size() * 8, hipMemcpyDeviceToHost));
  double ghz = (double)hc[0] / (double)hc[1] * 0.1;
  double nm = (double)blocks * (threads / 64) * iters * NACC;
  double ops = KIND == 0 ? 2.0 * 16 * 16 * 64 : 2.0 * 16 * 16 * 32;
  printf("%-26s waves/SIMD %d acc %2d: %8.1f T%s/s  clock %.2f GHz  %.1f cyc/MFMA/SIMD\n", KIND == 0 ? "v_mfma_i32_16x16x64_i8" : "v_mfma_f32_16x16x32_bf16",
         wps, NACC, nm * ops / best / 1e9, KIND == 0 ? "OP" : "FLOP", ghz, (double)hc[0] / ((double)iters * NACC * wps));
}

int main() {
  hipDeviceProp_t prop; CK(hipGetDeviceProperties(&prop, 0));
  int cus = prop.multiProcessorCount;
  int *out, *in; unsigned long long* clk;
  CK(hipMalloc(&out, 4 * 1024 * cus)); CK(hipMalloc(&in, 1024 * 16 * 4)); CK(hipMalloc(&clk, 16 * cus));
  std::vector<int> h(1024 * 16);
  for (auto& v : h) v = rand();
  CK(hipMemcpy(in, h.data(), h.size() * 4, hipMemcpyHostToDevice));
  run<0, 4>(cus, 1, out, in, clk); run<0, 8>(cus, 1, out, in, clk); run<0, 8>(cus, 2, out, in, clk);
  run<1, 4>(cus, 1, out, in, clk); run<1, 8>(cus, 1, out, in, clk); run<1, 8>(cus, 2, out, in, clk);
  return 0;
}
