// Ablation harness for the tile engine's complex main loop (diagnostic build, not part of the product).
// Variants drop one ingredient at a time to see where the non-MFMA time goes.
//   hipcc -O3 --offload-arch=gfx950 -std=c++17 -Ineo_ls_svm_amd/csrc tools/ablate_gemm.hip -o tools/ablate_gemm
#include <hip/hip_runtime.h>
#include <cstdio>
#include <cstdlib>
#include <vector>
#include "nls_gemm.h"
#include "nls_gemm3m.h"
using namespace nls;
#define CK(x) do { hipError_t e_ = (x); if (e_ != hipSuccess) { fprintf(stderr, "HIP %s line %d\n", hipGetErrorString(e_), __LINE__); exit(1);} } while (0)

enum { NO_GLOAD = ABL_NO_GLOAD, NO_LDS_STORE = ABL_NO_LDS_STORE, NO_BARRIER = ABL_NO_BARRIER, NO_FRAG = ABL_NO_FRAG };

// The REAL engine (k_sweep / dual GEMMs): m-major A, k-major B, two workgroups per CU.
template <int FLAGS>
__global__ void __launch_bounds__(Cfg4::NTHREADS, 2) k_abl(const double* Fc, const double* Fs, int Kp, int ktiles, double* out) {
  using C = Cfg4;
  extern __shared__ double smem[];
  v4d acc[C::MT][C::NTL];
  zero_acc(acc);
  const long col0 = (long)(blockIdx.x % (Kp / BN)) * BN;
  const long row0 = (long)((blockIdx.x / 7) % 500) * BM;
  MMajorLoader<C::NTHREADS, BM> la{Fc, Kp, row0};
  KMajorLoader<C::NTHREADS, BN> lb{Fs, Kp, col0};
  mainloop_real<C, false, MMajorLoader<C::NTHREADS, BM>, KMajorLoader<C::NTHREADS, BN>, FLAGS>(acc, la, lb, 0, ktiles, smem);
  double s = 0;
#pragma unroll
  for (int mt = 0; mt < C::MT; ++mt)
#pragma unroll
    for (int nt = 0; nt < C::NTL; ++nt)
#pragma unroll
      for (int r = 0; r < 4; ++r) s += acc[mt][nt][r];
  out[(long)blockIdx.x * C::NTHREADS + threadIdx.x] = s;
}

template <int FLAGS, bool A_KMAJOR>
__global__ void __launch_bounds__(m3::NT3, 1) k_abl3(const double* Fc, const double* Fs, int Kp, int ktiles, double* out) {
  using namespace m3;
  extern __shared__ double smem[];
  acc_zero();
  const long col0 = (long)(blockIdx.x % (Kp / BN3)) * BN3;
  const long colA = (long)((blockIdx.x / 7) % (Kp / BM3)) * BM3;
  using AL = typename std::conditional<A_KMAJOR, KMajorLoader3<BM3, STAGE_A>, MMajorLoader3>::type;
  AL lac{Fc, Kp, colA}, las{Fs, Kp, colA};
  KMajorLoader3<BN3, STAGE_B> lbr{Fc, Kp, col0}, lbi{Fs, Kp, col0};
  mainloop_3m<A_KMAJOR, AL, KMajorLoader3<BN3, STAGE_B>, FLAGS>(lac, las, lbr, lbi, 0, ktiles, smem);
  acc_settle();
  double s = 0;
  static_for<24>([&](auto t) {
    const v4d a = acc_get<decltype(t)::value>();
    s += a[0] + a[1] + a[2] + a[3];
  });
  out[(long)blockIdx.x * NT3 + threadIdx.x] = s;
}

template <int FLAGS, bool AK>
void run3(const char* name, const double* Fc, const double* Fs, int Kp, int ktiles, int blocks, double* out) {
  const size_t smem = m3::SMEM3;
  CK(hipFuncSetAttribute(reinterpret_cast<const void*>(k_abl3<FLAGS, AK>), hipFuncAttributeMaxDynamicSharedMemorySize, (int)smem));
  hipEvent_t e0, e1; CK(hipEventCreate(&e0)); CK(hipEventCreate(&e1));
  float best = 1e30f;
  for (int rep = 0; rep < 3; ++rep) {
    CK(hipEventRecord(e0));
    hipLaunchKernelGGL((k_abl3<FLAGS, AK>), dim3(blocks), dim3(m3::NT3), smem, 0, Fc, Fs, Kp, ktiles, out);
    CK(hipEventRecord(e1)); CK(hipEventSynchronize(e1));
    float ms; CK(hipEventElapsedTime(&ms, e0, e1)); best = ms < best ? ms : best;
  }
  double flops4m = (double)blocks * ktiles * 16.0 * 128 * 64 * 8;  // 4M-equivalent (algorithmic) flops
  double cyc = best * 1e-3 * 2.39e9 / ktiles / ((blocks + 255) / 256);
  printf("%-34s %8.2f ms  %6.2f TFLOP/s (4M-equivalent)  %7.0f cycles/slice (ideal 6144)\n", name, best, flops4m / best / 1e9, cyc);
}

template <int FLAGS>
void run(const char* name, const double* Fc, const double* Fs, int Kp, int ktiles, int blocks, double* out) {
  const size_t smem = 4 * TILE_DOUBLES * sizeof(double);
  CK(hipFuncSetAttribute(reinterpret_cast<const void*>(k_abl<FLAGS>), hipFuncAttributeMaxDynamicSharedMemorySize, (int)smem));
  hipEvent_t e0, e1; CK(hipEventCreate(&e0)); CK(hipEventCreate(&e1));
  float best = 1e30f;
  for (int rep = 0; rep < 3; ++rep) {
    CK(hipEventRecord(e0));
    hipLaunchKernelGGL((k_abl<FLAGS>), dim3(blocks), dim3(Cfg4::NTHREADS), smem, 0, Fc, Fs, Kp, ktiles, out);
    CK(hipEventRecord(e1)); CK(hipEventSynchronize(e1));
    float ms; CK(hipEventElapsedTime(&ms, e0, e1)); best = ms < best ? ms : best;
  }
  double flops = (double)blocks * ktiles * 16.0 * 128 * 128 * 2;
  double cyc = best * 1e-3 * 2.39e9 / ktiles / ((blocks + 511) / 512);  // two workgroups per CU share each SIMD
  printf("%-34s %8.2f ms  %6.2f TFLOP/s  %7.0f cycles/slice per workgroup pair (ideal 8192)\n", name, best, flops / best / 1e9, cyc);
}

int main() {
  const int Kp = 4224, rows = 65536 + 64, ktiles = 2048, blocks = 2048;
  double *Fc, *Fs, *out;
  CK(hipMalloc(&Fc, (size_t)rows * Kp * 8)); CK(hipMalloc(&Fs, (size_t)rows * Kp * 8)); CK(hipMalloc(&out, (size_t)4096 * 512 * 8));
  std::vector<double> h((size_t)1 << 20);
  for (auto& v : h) v = rand() / (double)RAND_MAX - 0.5;
  for (size_t off = 0; off < (size_t)rows * Kp; off += h.size()) {
    size_t cnt = std::min(h.size(), (size_t)rows * Kp - off);
    CK(hipMemcpy(Fc + off, h.data(), cnt * 8, hipMemcpyHostToDevice)); CK(hipMemcpy(Fs + off, h.data() + 7, (cnt - 7) * 8, hipMemcpyHostToDevice));
  }
  double* out2; CK(hipMalloc(&out2, (size_t)4096 * 256 * 8));
  printf("REAL engine (sweep-like: m-major A, k-major B), 4096 blocks x 260 slices, 2 workgroups per CU\n");
  run<0>("full", Fc, Fs, Kp, 260, 4096, out);
  run<NO_GLOAD>("no global loads", Fc, Fs, Kp, 260, 4096, out);
  run<NO_GLOAD | NO_LDS_STORE | NO_BARRIER>("no gload/LDS store/barrier", Fc, Fs, Kp, 260, 4096, out);
  run<NO_GLOAD | NO_LDS_STORE | NO_BARRIER | NO_FRAG>("MFMA only", Fc, Fs, Kp, 260, 4096, out);
  printf("3M engine, 128x64 tile, 1 wave/SIMD: gram-like\n");
  run3<0, true>("3M full", Fc, Fs, Kp, ktiles, 4096, out2);
  run3<NO_GLOAD, true>("3M no global loads", Fc, Fs, Kp, ktiles, 4096, out2);
  run3<NO_GLOAD | NO_LDS_STORE, true>("3M no gload, no LDS store", Fc, Fs, Kp, ktiles, 4096, out2);
  run3<NO_GLOAD | NO_LDS_STORE | NO_BARRIER, true>("3M ... and no barrier", Fc, Fs, Kp, ktiles, 4096, out2);
  run3<NO_GLOAD | NO_LDS_STORE | NO_BARRIER | NO_FRAG, true>("3M MFMA only", Fc, Fs, Kp, ktiles, 4096, out2);
  run3<ABL_NO_INTERLEAVE, true>("3M full, side operations not interleaved", Fc, Fs, Kp, ktiles, 4096, out2);
  printf("3M rotate-like (m-major A)\n");
  run3<0, false>("3M full", Fc, Fs, Kp, 264, 4096, out2);
  run3<NO_GLOAD, false>("3M no global loads", Fc, Fs, Kp, 264, 4096, out2);
  return 0;
}
