// Do fp64 MFMA and fp64 VALU work of DIFFERENT waves on one SIMD overlap on gfx950?  (K1: can the sincos epilogue of one
// workgroup hide under the MFMA main loop of another?)  One 512-thread workgroup per CU: waves 0-3 (one per SIMD) run a
// chain of v_mfma_f64_16x16x4, waves 4-7 run fp64 FMA chains / sincos.  Modes: 1 = MFMA only, 2 = VALU only, 3 = both.
//   hipcc --offload-arch=gfx950 -O3 tools/probe_f64_coexec.hip -o tools/probe_f64_coexec && tools/probe_f64_coexec
#include <hip/hip_runtime.h>

#include <cstdio>

typedef double v4d __attribute__((ext_vector_type(4)));

template <int VKIND>
__global__ void __launch_bounds__(512) k(int mode, int iters, double* out) {
  const int wave = threadIdx.x >> 6;
  if (wave < 4) {
    if (!(mode & 1)) return;
    v4d a0 = {0, 0, 0, 0}, a1 = a0, a2 = a0, a3 = a0;
    double x = 1.0 + threadIdx.x * 1e-9, y = 1.0 - threadIdx.x * 1e-9;
    for (int i = 0; i < iters; ++i) {
      a0 = __builtin_amdgcn_mfma_f64_16x16x4f64(x, y, a0, 0, 0, 0);
      a1 = __builtin_amdgcn_mfma_f64_16x16x4f64(x, y, a1, 0, 0, 0);
      a2 = __builtin_amdgcn_mfma_f64_16x16x4f64(x, y, a2, 0, 0, 0);
      a3 = __builtin_amdgcn_mfma_f64_16x16x4f64(x, y, a3, 0, 0, 0);
    }
    out[blockIdx.x * 512 + threadIdx.x] = a0[0] + a1[1] + a2[2] + a3[3];
  } else {
    if (!(mode & 2)) return;
    double acc = 0.0;
    if (VKIND == 0) {  // 8 independent fp64 FMA chains
      double c0 = threadIdx.x, c1 = 1, c2 = 2, c3 = 3, c4 = 4, c5 = 5, c6 = 6, c7 = 7;
      const double m = 1.0000001, b = 1e-9;
      for (int i = 0; i < iters * 4; ++i) {
        c0 = fma(c0, m, b); c1 = fma(c1, m, b); c2 = fma(c2, m, b); c3 = fma(c3, m, b);
        c4 = fma(c4, m, b); c5 = fma(c5, m, b); c6 = fma(c6, m, b); c7 = fma(c7, m, b);
      }
      acc = c0 + c1 + c2 + c3 + c4 + c5 + c6 + c7;
    } else {  // sincos as the feature map calls it
      double t = threadIdx.x * 0.37 + blockIdx.x;
      for (int i = 0; i < iters / 4; ++i) {
        double s, c;
        sincos(t, &s, &c);
        acc += s * 0.5 + c;
        t += 1.7;
      }
    }
    out[blockIdx.x * 512 + threadIdx.x] = acc;
  }
}

template <int VKIND>
static void run(const char* name, int iters) {
  double* out;
  hipMalloc(&out, 256 * 512 * sizeof(double));
  hipEvent_t e0, e1;
  hipEventCreate(&e0);
  hipEventCreate(&e1);
  float ms[4] = {0, 0, 0, 0};
  for (int mode = 1; mode <= 3; ++mode) {
    hipLaunchKernelGGL(k<VKIND>, dim3(256), dim3(512), 0, 0, mode, iters, out);
    hipDeviceSynchronize();
    hipEventRecord(e0);
    for (int r = 0; r < 5; ++r) hipLaunchKernelGGL(k<VKIND>, dim3(256), dim3(512), 0, 0, mode, iters, out);
    hipEventRecord(e1);
    hipEventSynchronize(e1);
    hipEventElapsedTime(&ms[mode], e0, e1);
    ms[mode] /= 5;
  }
  const double mf = 256.0 * 4 * iters * 4 * (2.0 * 16 * 16 * 4) / (ms[1] * 1e-3) / 1e12;
  printf("%-8s iters %d: mfma only %.3f ms (%.1f TFLOP/s), valu only %.3f ms, both %.3f ms  -> overlap %.2f (1 = perfect, 0 = serial)\n", name,
         iters, ms[1], mf, ms[2], ms[3], (ms[1] + ms[2] - ms[3]) / (ms[1] < ms[2] ? ms[1] : ms[2]));
  hipFree(out);
}

int main() {
  run<0>("fma", 20000);
  run<1>("sincos", 20000);
  run<0>("fma", 60000);
  return 0;
}
