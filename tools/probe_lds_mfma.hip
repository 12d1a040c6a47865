// How many matrix-pipe cycles does one extra instruction cost when it is issued between fp64 MFMAs by the same
// wave (one wave per SIMD, 4 per CU)?  24 independent MFMAs per iteration with NX extra instructions of one kind,
// one after each of the first NX MFMAs, written as inline asm so that nothing else rides along.
#include <hip/hip_runtime.h>
#include <cstdio>
#include <cstdlib>
#define CK(x) do { hipError_t e_ = (x); if (e_ != hipSuccess) { fprintf(stderr, "HIP %s line %d\n", hipGetErrorString(e_), __LINE__); exit(1);} } while (0)
typedef double v4d __attribute__((ext_vector_type(4)));
typedef double v2d __attribute__((ext_vector_type(2)));
typedef unsigned int u4 __attribute__((ext_vector_type(4)));

enum { NONE, DSR64, DSR128, DSW64, DSW128, GLD128, FMA64, ADD64, ADDU32, MOV32, FMA32, SALU };

template <int KIND, int NX>
__global__ void __launch_bounds__(256, 1) k(double* out, const double* in, int iters) {
  extern __shared__ double sm[];
  v4d acc[24];
  for (int i = 0; i < 24; ++i) acc[i] = v4d{0, 0, 0, 0};
  for (int i = threadIdx.x; i < 8192; i += 256) sm[i] = in[i];
  __syncthreads();
  double a = in[threadIdx.x], b = in[threadIdx.x + 256];
  double xd[24]; v2d yd[24]; unsigned xi[24]; float xf[24];
  for (int i = 0; i < 24; ++i) { xd[i] = a; yd[i] = v2d{a, b}; xi[i] = threadIdx.x; xf[i] = 1.f; }
  // conflict-free lane addresses: b64 -> consecutive 8 B, b128 -> consecutive 16 B
  unsigned a64 = (unsigned)(size_t)sm + threadIdx.x * 8, a128 = (unsigned)(size_t)sm + threadIdx.x * 16;
  const double* gp = in + threadIdx.x * 2;
  int sacc = 0;
  for (int it = 0; it < iters; ++it) {
#pragma unroll
    for (int i = 0; i < 24; ++i) {
      acc[i] = __builtin_amdgcn_mfma_f64_16x16x4f64(a, b, acc[i], 0, 0, 0);
      __builtin_amdgcn_sched_barrier(0);
      if (i < NX) {
        if (KIND == DSR64) asm volatile("ds_read_b64 %0, %1 offset:2048" : "=v"(xd[i]) : "v"(a64));
        if (KIND == DSR128) asm volatile("ds_read_b128 %0, %1 offset:4096" : "=v"(yd[i]) : "v"(a128));
        if (KIND == DSW64) asm volatile("ds_write_b64 %0, %1 offset:2048" : : "v"(a64), "v"(a));
        if (KIND == DSW128) asm volatile("ds_write_b128 %0, %1 offset:4096" : : "v"(a128), "v"(yd[0]));
        if (KIND == GLD128) asm volatile("global_load_dwordx4 %0, %1, off" : "=v"(yd[i]) : "v"(gp + (size_t)((it * 24 + i) & 1023) * 512));
        if (KIND == FMA64) asm volatile("v_fma_f64 %0, %1, %2, %0" : "+v"(xd[i]) : "v"(a), "v"(b));
        if (KIND == ADD64) asm volatile("v_add_f64 %0, %1, %0" : "+v"(xd[i]) : "v"(a));
        if (KIND == ADDU32) asm volatile("v_add_u32 %0, %1, %0" : "+v"(xi[i]) : "v"(xi[(i + 1) % 24]));
        if (KIND == MOV32) asm volatile("v_mov_b32 %0, %1" : "=v"(xi[i]) : "v"(xi[(i + 1) % 24]));
        if (KIND == FMA32) asm volatile("v_fma_f32 %0, %1, %1, %0" : "+v"(xf[i]) : "v"(xf[(i + 1) % 24]));
        if (KIND == SALU) asm volatile("s_add_u32 %0, %0, 1" : "+s"(sacc));
        __builtin_amdgcn_sched_barrier(0);
      }
    }
    if (KIND == DSR64 || KIND == DSR128 || KIND == DSW64 || KIND == DSW128) asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory");
    if (KIND == GLD128) asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
  }
  double s = sacc;
  for (int i = 0; i < 24; ++i) s += acc[i][0] + acc[i][1] + acc[i][2] + acc[i][3] + xd[i] + yd[i].x + yd[i].y + xi[i] + xf[i];
  out[blockIdx.x * 256 + threadIdx.x] = s + sm[threadIdx.x];
}

static double base = 1545;
template <int KIND, int NX>
void run(const char* name, double* out, double* in) {
  const int iters = 4000, blocks = 256;
  CK(hipFuncSetAttribute(reinterpret_cast<const void*>(k<KIND, NX>), hipFuncAttributeMaxDynamicSharedMemorySize, 65536));
  hipEvent_t e0, e1; CK(hipEventCreate(&e0)); CK(hipEventCreate(&e1));
  float best = 1e30f;
  for (int rep = 0; rep < 3; ++rep) {
    CK(hipEventRecord(e0));
    hipLaunchKernelGGL((k<KIND, NX>), dim3(blocks), dim3(256), 65536, 0, out, in, iters);
    CK(hipEventRecord(e1)); CK(hipEventSynchronize(e1));
    float ms; CK(hipEventElapsedTime(&ms, e0, e1)); best = ms < best ? ms : best;
  }
  double cyc = best * 1e-3 * 2.39e9 / iters;
  if (KIND == NONE) base = cyc;
  printf("%-22s x%2d per 24 MFMAs: %7.1f cycles/iter -> %+6.1f cycles per extra instruction\n", name, NX, cyc, NX ? (cyc - base) / NX : 0.0);
}

int main() {
  double *out, *in;
  CK(hipMalloc(&out, 256 * 256 * 8)); CK(hipMalloc(&in, (1 << 22) * 8)); CK(hipMemset(in, 0, (1 << 22) * 8));
  run<NONE, 0>("MFMA only", out, in);
  run<DSR64, 12>("ds_read_b64", out, in); run<DSR64, 24>("ds_read_b64", out, in);
  run<DSR128, 12>("ds_read_b128", out, in); run<DSR128, 24>("ds_read_b128", out, in);
  run<DSW64, 12>("ds_write_b64", out, in); run<DSW128, 12>("ds_write_b128", out, in);
  run<GLD128, 6>("global_load_dwordx4", out, in); run<GLD128, 12>("global_load_dwordx4", out, in);
  run<FMA64, 12>("v_fma_f64", out, in); run<FMA64, 24>("v_fma_f64", out, in);
  run<ADD64, 12>("v_add_f64", out, in);
  run<ADDU32, 24>("v_add_u32", out, in); run<MOV32, 24>("v_mov_b32", out, in); run<FMA32, 24>("v_fma_f32", out, in);
  run<SALU, 24>("s_add_u32", out, in);
  return 0;
}
