// fp64 MFMA issue-rate probe: waves/SIMD x accumulators x operand data, with the in-kernel clock
// (delta s_memtime / delta s_memrealtime x 100 MHz, MI355X_MICROARCH.md "DVFS give-back" item 6).
#include <hip/hip_runtime.h>
#include <cstdio>
#include <cstdlib>
#include <vector>
#define CK(x) do { hipError_t e_ = (x); if (e_ != hipSuccess) { fprintf(stderr, "HIP error %s at %d\n", hipGetErrorString(e_), __LINE__); exit(1);} } while (0)
typedef double v4d __attribute__((ext_vector_type(4)));

template <int NACC>
__global__ void mfma_rate(double* out, const double* in, unsigned long long* clk, int iters) {
  v4d acc[NACC];
#pragma unroll
  for (int i = 0; i < NACC; ++i) acc[i] = v4d{0, 0, 0, 0};
  double a[4], b[4];
#pragma unroll
  for (int i = 0; i < 4; ++i) { a[i] = in[threadIdx.x * 8 + i]; b[i] = in[threadIdx.x * 8 + 4 + i]; }
  unsigned long long t0 = __builtin_amdgcn_s_memtime(), r0 = __builtin_amdgcn_s_memrealtime();
  for (int it = 0; it < iters; ++it) {
#pragma unroll
    for (int i = 0; i < NACC; ++i) acc[i] = __builtin_amdgcn_mfma_f64_16x16x4f64(a[i & 3], b[(i >> 2) & 3], acc[i], 0, 0, 0);
  }
  unsigned long long t1 = __builtin_amdgcn_s_memtime(), r1 = __builtin_amdgcn_s_memrealtime();
  double s = 0;
#pragma unroll
  for (int i = 0; i < NACC; ++i) s += acc[i][0] + acc[i][1] + acc[i][2] + acc[i][3];
  out[blockIdx.x * blockDim.x + threadIdx.x] = s;
  if (threadIdx.x == 0) { clk[2 * blockIdx.x] = t1 - t0; clk[2 * blockIdx.x + 1] = r1 - r0; }
}

template <int NACC>
void run(int cus, int waves_per_simd, bool randomdata, double* out, double* in, unsigned long long* clk) {
  int threads = 256 * waves_per_simd > 1024 ? 1024 : 256 * waves_per_simd;
  int blocks_per_cu = (256 * waves_per_simd) / threads;
  int blocks = cus * blocks_per_cu;
  int iters = 40000 / NACC * 4;
  std::vector<double> h(1024 * 8);
  for (size_t i = 0; i < h.size(); ++i) h[i] = randomdata ? (rand() / (double)RAND_MAX * 2 - 1) : 1.0;
  CK(hipMemcpy(in, h.data(), h.size() * 8, hipMemcpyHostToDevice));
  hipEvent_t e0, e1; CK(hipEventCreate(&e0)); CK(hipEventCreate(&e1));
  float best = 1e30f;
  for (int rep = 0; rep < 3; ++rep) {
    CK(hipEventRecord(e0));
    mfma_rate<NACC><<<blocks, threads>>>(out, in, clk, iters);
    CK(hipEventRecord(e1)); CK(hipEventSynchronize(e1));
    float ms; CK(hipEventElapsedTime(&ms, e0, e1)); if (ms < best) best = ms;
  }
  std::vector<unsigned long long> hc(2 * blocks);
  CK(hipMemcpy(hc.data(), clk, hc.size() * 8, hipMemcpyDeviceToHost));
  double ghz = (double)hc[0] / (double)hc[1] * 0.1;
  double nm = (double)blocks * (threads / 64) * iters * NACC;
  double cyc_per_mfma_simd = (double)hc[0] / ((double)iters * NACC * waves_per_simd);
  printf("waves/SIMD %d acc %2d data %s: %.2f TFLOP/s  clock %.2f GHz  %.1f cyc/MFMA/SIMD\n", waves_per_simd, NACC,
         randomdata ? "rand" : "ones", nm * 2048 / best / 1e9, ghz, cyc_per_mfma_simd);
}

int main() {
  hipDeviceProp_t prop; CK(hipGetDeviceProperties(&prop, 0));
  int cus = prop.multiProcessorCount;
  double *out, *in; unsigned long long* clk;
  CK(hipMalloc(&out, 8 * 1024 * 8 * cus)); CK(hipMalloc(&in, 1024 * 8 * 8)); CK(hipMalloc(&clk, 16 * 8 * cus));
  for (int rd = 0; rd < 2; ++rd) {
    run<4>(cus, 1, rd, out, in, clk); run<8>(cus, 1, rd, out, in, clk); run<16>(cus, 1, rd, out, in, clk);
    run<4>(cus, 2, rd, out, in, clk); run<8>(cus, 2, rd, out, in, clk); run<16>(cus, 2, rd, out, in, clk);
    run<4>(cus, 4, rd, out, in, clk); run<8>(cus, 4, rd, out, in, clk);
  }
  return 0;
}
