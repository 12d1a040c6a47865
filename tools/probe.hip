// Hardware probe for design decisions (not part of the product):
//   1. fp64 MFMA issue rate (v_mfma_f64_16x16x4_f64) -> the peak roofline.fracs are priced against
//   2. rocSOLVER Hermitian EVD / Cholesky wall time at the hot path's sizes (the serial section)
//   3. streaming HBM bandwidth with 16-B accesses
// Build: hipcc -O3 --offload-arch=gfx950 tools/probe.hip -o tools/probe -lrocsolver -lrocblas
#include <hip/hip_runtime.h>
#include <rocblas/rocblas.h>
#include <rocsolver/rocsolver.h>

#include <chrono>
#include <cstdio>
#include <cstdlib>
#include <vector>

#define CK(x)                                                                  \
  do {                                                                         \
    hipError_t e_ = (x);                                                       \
    if (e_ != hipSuccess) {                                                    \
      fprintf(stderr, "HIP error %s at %s:%d\n", hipGetErrorString(e_), __FILE__, __LINE__); \
      exit(1);                                                                 \
    }                                                                          \
  } while (0)

typedef double v4d __attribute__((ext_vector_type(4)));

template <int NACC>
__global__ void __launch_bounds__(256) mfma_rate(double* out, int iters, double a0, double b0) {
  v4d acc[NACC];
#pragma unroll
  for (int i = 0; i < NACC; ++i) acc[i] = v4d{0, 0, 0, 0};
  double a = a0 + threadIdx.x * 1e-3, b = b0 - threadIdx.x * 1e-3;
  for (int it = 0; it < iters; ++it) {
#pragma unroll
    for (int i = 0; i < NACC; ++i) acc[i] = __builtin_amdgcn_mfma_f64_16x16x4f64(a, b, acc[i], 0, 0, 0);
  }
  double s = 0;
#pragma unroll
  for (int i = 0; i < NACC; ++i) s += acc[i][0] + acc[i][1] + acc[i][2] + acc[i][3];
  out[blockIdx.x * blockDim.x + threadIdx.x] = s;
}

__global__ void stream_copy(const double2* __restrict__ in, double2* __restrict__ out, size_t n) {
  size_t i = blockIdx.x * (size_t)blockDim.x + threadIdx.x;
  size_t stride = (size_t)gridDim.x * blockDim.x;
  for (; i < n; i += stride) out[i] = in[i];
}

__global__ void fill_herm(rocblas_double_complex* A, int n, unsigned seed) {
  // A = diag-dominant-ish random Hermitian (column-major): fills both triangles consistently.
  int j = blockIdx.x * blockDim.x + threadIdx.x;
  int i = blockIdx.y;
  if (j >= n || i >= n) return;
  int lo = i < j ? i : j, hi = i < j ? j : i;
  unsigned h = (unsigned)lo * 2654435761u ^ ((unsigned)hi * 40503u + seed);
  h ^= h >> 13; h *= 0x5bd1e995u; h ^= h >> 15;
  double re = ((h & 0xffff) / 65536.0 - 0.5), im = (((h >> 16) & 0xffff) / 65536.0 - 0.5);
  if (i == j) { re = 4.0 + re; im = 0; }
  if (i > j) im = -im;  // element (i,j), i row: lower gets conj
  A[(size_t)j * n + i] = rocblas_double_complex(re, im);
}

__global__ void fill_sym(double* A, int n, unsigned seed) {
  int j = blockIdx.x * blockDim.x + threadIdx.x;
  int i = blockIdx.y;
  if (j >= n || i >= n) return;
  int lo = i < j ? i : j, hi = i < j ? j : i;
  unsigned h = (unsigned)lo * 2654435761u ^ ((unsigned)hi * 40503u + seed);
  h ^= h >> 13; h *= 0x5bd1e995u; h ^= h >> 15;
  double re = ((h & 0xffff) / 65536.0 - 0.5);
  if (i == j) re += 4.0;
  A[(size_t)j * n + i] = re;
}

static double now() {
  return std::chrono::duration<double>(std::chrono::steady_clock::now().time_since_epoch()).count();
}

int main(int argc, char** argv) {
  int big = argc > 1 ? atoi(argv[1]) : 4097;
  hipDeviceProp_t prop;
  CK(hipGetDeviceProperties(&prop, 0));
  printf("device %s CUs %d clock %d kHz mem %.1f GB\n", prop.name, prop.multiProcessorCount, prop.clockRate,
         prop.totalGlobalMem / 1e9);

  // 1. MFMA rate
  {
    int blocks = prop.multiProcessorCount * 2, iters = 20000;
    double* out;
    CK(hipMalloc(&out, sizeof(double) * blocks * 256));
    hipEvent_t e0, e1;
    CK(hipEventCreate(&e0));
    CK(hipEventCreate(&e1));
    for (int rep = 0; rep < 2; ++rep) {
      CK(hipEventRecord(e0));
      mfma_rate<8><<<blocks, 256>>>(out, iters, 1.0, 2.0);
      CK(hipEventRecord(e1));
      CK(hipEventSynchronize(e1));
      float ms;
      CK(hipEventElapsedTime(&ms, e0, e1));
      double flops = (double)blocks * 4 * iters * 8 * 2048.0;
      printf("mfma_f64_16x16x4 x8 acc, %d blocks x 4 waves: %.3f ms -> %.2f TFLOP/s (%.1f cyc/MFMA/SIMD @2.4GHz, 1 wave/SIMD eq.)\n",
             blocks, ms, flops / ms / 1e9, 2.4e9 * ms * 1e-3 / (iters * 8.0) / (blocks * 4.0 / (prop.multiProcessorCount * 4)));
    }
    blocks = prop.multiProcessorCount;
    CK(hipEventRecord(e0));
    mfma_rate<16><<<blocks, 256>>>(out, iters, 1.0, 2.0);
    CK(hipEventRecord(e1));
    CK(hipEventSynchronize(e1));
    float ms;
    CK(hipEventElapsedTime(&ms, e0, e1));
    printf("mfma 1 wave/SIMD x16 acc: %.3f ms -> %.2f TFLOP/s\n", ms, (double)blocks * 4 * iters * 16 * 2048.0 / ms / 1e9);
    CK(hipFree(out));
  }
  // 3. HBM stream
  {
    size_t n = (size_t)1 << 28;  // 4 GiB of double2
    double2 *a, *b;
    CK(hipMalloc(&a, n * sizeof(double2)));
    CK(hipMalloc(&b, n * sizeof(double2)));
    CK(hipMemset(a, 1, n * sizeof(double2)));
    hipEvent_t e0, e1;
    CK(hipEventCreate(&e0));
    CK(hipEventCreate(&e1));
    for (int rep = 0; rep < 3; ++rep) {
      CK(hipEventRecord(e0));
      stream_copy<<<2048, 256>>>(a, b, n);
      CK(hipEventRecord(e1));
      CK(hipEventSynchronize(e1));
      float ms;
      CK(hipEventElapsedTime(&ms, e0, e1));
      printf("stream copy 2x%.1f GB: %.3f ms -> %.2f TB/s\n", n * 16 / 1e9, ms, 2.0 * n * 16 / ms / 1e9);
    }
    CK(hipFree(a));
    CK(hipFree(b));
  }
  // 2. rocSOLVER
  rocblas_handle h;
  rocblas_create_handle(&h);
  int sizes[] = {1025, big};
  for (int si = 0; si < 2; ++si) {
    int n = sizes[si];
    rocblas_double_complex *A, *A0;
    double *W, *E;
    rocblas_int* info;
    CK(hipMalloc(&A, sizeof(rocblas_double_complex) * (size_t)n * n));
    CK(hipMalloc(&A0, sizeof(rocblas_double_complex) * (size_t)n * n));
    CK(hipMalloc(&W, sizeof(double) * n));
    CK(hipMalloc(&E, sizeof(double) * n));
    CK(hipMalloc(&info, sizeof(rocblas_int)));
    dim3 g((n + 255) / 256, n);
    fill_herm<<<g, 256>>>(A0, n, 7u);
    CK(hipDeviceSynchronize());
    for (int rep = 0; rep < 2; ++rep) {
      CK(hipMemcpy(A, A0, sizeof(rocblas_double_complex) * (size_t)n * n, hipMemcpyDeviceToDevice));
      CK(hipDeviceSynchronize());
      double t0 = now();
      rocblas_status st = rocsolver_zheevd(h, rocblas_evect_original, rocblas_fill_lower, n, A, n, W, E, info);
      CK(hipDeviceSynchronize());
      double t1 = now();
      int hinfo;
      CK(hipMemcpy(&hinfo, info, sizeof(int), hipMemcpyDeviceToHost));
      printf("zheevd n=%d: %.3f s (status %d info %d)\n", n, t1 - t0, (int)st, hinfo);
      fflush(stdout);
    }
    {
      CK(hipMemcpy(A, A0, sizeof(rocblas_double_complex) * (size_t)n * n, hipMemcpyDeviceToDevice));
      CK(hipDeviceSynchronize());
      double t0 = now();
      rocblas_status st = rocsolver_zheev(h, rocblas_evect_original, rocblas_fill_lower, n, A, n, W, E, info);
      CK(hipDeviceSynchronize());
      printf("zheev  n=%d: %.3f s (status %d)\n", n, now() - t0, (int)st);
      fflush(stdout);
    }
    for (int rep = 0; rep < 2; ++rep) {
      CK(hipMemcpy(A, A0, sizeof(rocblas_double_complex) * (size_t)n * n, hipMemcpyDeviceToDevice));
      CK(hipDeviceSynchronize());
      double t0 = now();
      rocblas_status st = rocsolver_zpotrf(h, rocblas_fill_lower, n, A, n, info);
      CK(hipDeviceSynchronize());
      printf("zpotrf n=%d: %.4f s (status %d)\n", n, now() - t0, (int)st);
    }
    {  // tridiagonalisation alone (the BLAS2-bound half of the EVD)
      rocblas_double_complex* tau;
      CK(hipMalloc(&tau, sizeof(rocblas_double_complex) * n));
      CK(hipMemcpy(A, A0, sizeof(rocblas_double_complex) * (size_t)n * n, hipMemcpyDeviceToDevice));
      CK(hipDeviceSynchronize());
      double t0 = now();
      rocblas_status st = rocsolver_zhetrd(h, rocblas_fill_lower, n, A, n, W, E, tau);
      CK(hipDeviceSynchronize());
      printf("zhetrd n=%d: %.3f s (status %d)\n", n, now() - t0, (int)st);
      CK(hipFree(tau));
    }
    CK(hipFree(A));
    CK(hipFree(A0));
    CK(hipFree(W));
    CK(hipFree(E));
    CK(hipFree(info));
    fflush(stdout);
  }
  // real symmetric EVD for the dual path
  int dsizes[] = {2048, argc > 2 ? atoi(argv[2]) : 10000};
  for (int si = 0; si < 2; ++si) {
    int n = dsizes[si];
    double *A, *W, *E;
    rocblas_int* info;
    CK(hipMalloc(&A, sizeof(double) * (size_t)n * n));
    CK(hipMalloc(&W, sizeof(double) * n));
    CK(hipMalloc(&E, sizeof(double) * n));
    CK(hipMalloc(&info, sizeof(rocblas_int)));
    dim3 g((n + 255) / 256, n);
    fill_sym<<<g, 256>>>(A, n, 11u);
    CK(hipDeviceSynchronize());
    double t0 = now();
    rocblas_status st = rocsolver_dsyevd(h, rocblas_evect_original, rocblas_fill_lower, n, A, n, W, E, info);
    CK(hipDeviceSynchronize());
    printf("dsyevd n=%d: %.3f s (status %d)\n", n, now() - t0, (int)st);
    fflush(stdout);
    CK(hipFree(A));
    CK(hipFree(W));
    CK(hipFree(E));
    CK(hipFree(info));
  }
  rocblas_destroy_handle(h);
  return 0;
}
