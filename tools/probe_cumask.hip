// Probe: does hipExtStreamCreateWithCUMask confine a stream's kernels to a subset of the compute units on this box, and how do mask bits map
// to (XCC, SE, CU)?  Prints, per mask, the set of (xcc, se, cu) triples that ran workgroups and the time of a bandwidth-bound kernel.
#include <hip/hip_runtime.h>
#include <cstdio>
#include <cstdint>
#include <set>
#include <vector>
#define CK(x) do { hipError_t e_ = (x); if (e_ != hipSuccess) { std::printf("%s -> %s\n", #x, hipGetErrorString(e_)); return 1; } } while (0)

__global__ void k_where(unsigned* out, int spin) {
  unsigned hw, xcc;
  asm volatile("s_getreg_b32 %0, hwreg(HW_REG_HW_ID)" : "=s"(hw));
  asm volatile("s_getreg_b32 %0, hwreg(HW_REG_XCC_ID)" : "=s"(xcc));
  long long t0 = wall_clock64();
  while (wall_clock64() - t0 < spin) {}
  if (threadIdx.x == 0) out[blockIdx.x] = (hw & 0xffff) | ((xcc & 0xf) << 16);
}
__global__ void k_stream(const double* a, double* b, long n) {
  for (long i = blockIdx.x * (long)blockDim.x + threadIdx.x; i < n; i += (long)gridDim.x * blockDim.x) b[i] = a[i] * 1.0000001;
}

int main() {
  hipDeviceProp_t p;
  CK(hipGetDeviceProperties(&p, 0));
  std::printf("CUs %d\n", p.multiProcessorCount);
  const int nblk = 4096;
  unsigned* d;
  CK(hipMalloc(&d, nblk * 4));
  double *a, *b;
  const long n = 1L << 27;
  CK(hipMalloc(&a, n * 8));
  CK(hipMalloc(&b, n * 8));
  CK(hipMemset(a, 0, n * 8));
  std::vector<std::vector<uint32_t>> masks;
  masks.push_back(std::vector<uint32_t>(8, 0xffffffffu));                       // all 256
  masks.push_back({0xffffffffu, 0, 0, 0, 0, 0, 0, 0});                          // bits 0..31
  masks.push_back({0, 0, 0, 0, 0, 0, 0, 0xffffffffu});                          // bits 224..255
  masks.push_back({0xffffffffu, 0xffffffffu, 0xffffffffu, 0xffffffffu, 0xffffffffu, 0xffffffffu, 0xffffffffu, 0});  // bits 0..223
  masks.push_back(std::vector<uint32_t>(8, 0x01010101u));                       // every 8th bit
  for (size_t mi = 0; mi < masks.size(); ++mi) {
    hipStream_t s;
    hipError_t e = hipExtStreamCreateWithCUMask(&s, (uint32_t)masks[mi].size(), masks[mi].data());
    if (e != hipSuccess) { std::printf("mask %zu: create failed: %s\n", mi, hipGetErrorString(e)); continue; }
    CK(hipMemsetAsync(d, 0xff, nblk * 4, s));
    hipLaunchKernelGGL(k_where, dim3(nblk), dim3(64), 0, s, d, 2000);
    CK(hipStreamSynchronize(s));
    std::vector<unsigned> h(nblk);
    CK(hipMemcpy(h.data(), d, nblk * 4, hipMemcpyDeviceToHost));
    std::set<unsigned> cus;
    std::set<unsigned> xccs;
    for (unsigned v : h) { cus.insert(((v >> 16) << 16) | (v & 0xff00)); xccs.insert(v >> 16); }
    hipEvent_t e0, e1;
    CK(hipEventCreate(&e0)); CK(hipEventCreate(&e1));
    hipLaunchKernelGGL(k_stream, dim3(4096), dim3(256), 0, s, a, b, n);
    CK(hipEventRecord(e0, s));
    for (int r = 0; r < 5; ++r) hipLaunchKernelGGL(k_stream, dim3(4096), dim3(256), 0, s, a, b, n);
    CK(hipEventRecord(e1, s));
    CK(hipStreamSynchronize(s));
    float ms; CK(hipEventElapsedTime(&ms, e0, e1));
    std::printf("mask %zu: distinct (xcc,se,sh,cu) = %zu over %zu xccs; stream copy %.1f GB/s\n", mi, cus.size(), xccs.size(), 5 * 2.0 * n * 8 / ms / 1e6);
    if (mi == 1 || mi == 4) { std::printf("   ids:"); int c = 0; for (unsigned v : cus) { if (c++ < 40) std::printf(" %x", v); } std::printf("\n"); }
    CK(hipStreamDestroy(s));
  }
  // two disjoint masked streams at once: a spin kernel on the small set, the copy on the large one
  return 0;
}
