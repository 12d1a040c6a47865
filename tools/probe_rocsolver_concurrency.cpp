// Round-4 root-cause probe for profiles/r03_sigma_overlap.log: are rocSOLVER / rocBLAS calls safe when TWO handles of one process run them at the
// same time on two streams of one GPU?  Two host threads, each with its own handle, stream and matrices, repeat one operation; every result is
// compared bit for bit with the same operation run alone.
//   hipcc -O2 tools/probe_rocsolver_concurrency.cpp -o tools/probe_rocsolver_concurrency -lrocsolver -lrocblas
//   ./probe_rocsolver_concurrency [n = 512] [reps = 40]
#include <hip/hip_runtime.h>
#include <rocblas/rocblas.h>
#include <rocsolver/rocsolver.h>

#include <complex>
#include <cstdio>
#include <cstdlib>
#include <cstring>
#include <random>
#include <thread>
#include <vector>

using Z = std::complex<double>;
#define CK(x)                                                            \
  do {                                                                   \
    if ((x) != 0) {                                                      \
      std::fprintf(stderr, "%s failed at line %d\n", #x, __LINE__);     \
      std::exit(2);                                                      \
    }                                                                    \
  } while (0)

struct Worker {
  rocblas_handle h;
  hipStream_t s;
  Z *dA, *dB, *dC;
  rocblas_int* dinfo;
  std::vector<Z> A, B;  // host inputs
  int n, m;
};

enum Op { POTRF, TRSM, HERK, STEDC_NONE };

static void run_op(Worker& w, Op op, std::vector<Z>& out, int* info) {
  const int n = w.n, m = w.m;
  auto* zA = reinterpret_cast<rocblas_double_complex*>(w.dA);
  auto* zB = reinterpret_cast<rocblas_double_complex*>(w.dB);
  auto* zC = reinterpret_cast<rocblas_double_complex*>(w.dC);
  CK(hipMemcpyAsync(w.dA, w.A.data(), sizeof(Z) * n * n, hipMemcpyHostToDevice, w.s));
  CK(hipMemcpyAsync(w.dB, w.B.data(), sizeof(Z) * m * n, hipMemcpyHostToDevice, w.s));
  const rocblas_double_complex one(1.0, 0.0);
  const double minus = -1.0, onef = 1.0;
  *info = 0;
  if (op == POTRF) {
    CK(rocsolver_zpotrf(w.h, rocblas_fill_lower, n, zA, n, w.dinfo));
    CK(hipMemcpyAsync(info, w.dinfo, sizeof(int), hipMemcpyDeviceToHost, w.s));
    out.resize((size_t)n * n);
    CK(hipMemcpyAsync(out.data(), w.dA, sizeof(Z) * n * n, hipMemcpyDeviceToHost, w.s));
  } else if (op == TRSM) {  // B (m x n) <- B L^-H with L = tril(A) (diagonally dominant input: no factorisation needed)
    CK(rocblas_ztrsm(w.h, rocblas_side_right, rocblas_fill_lower, rocblas_operation_conjugate_transpose, rocblas_diagonal_non_unit, m, n, &one, zA, n, zB, m));
    out.resize((size_t)m * n);
    CK(hipMemcpyAsync(out.data(), w.dB, sizeof(Z) * m * n, hipMemcpyDeviceToHost, w.s));
  } else {  // C (m x m) <- C - B B^H
    CK(hipMemsetAsync(w.dC, 0, sizeof(Z) * m * m, w.s));
    CK(rocblas_zherk(w.h, rocblas_fill_lower, rocblas_operation_none, m, n, &minus, zB, m, &onef, zC, m));
    out.resize((size_t)m * m);
    CK(hipMemcpyAsync(out.data(), w.dC, sizeof(Z) * m * m, hipMemcpyDeviceToHost, w.s));
  }
  CK(hipStreamSynchronize(w.s));
}

int main(int argc, char** argv) {
  const int n = argc > 1 ? std::atoi(argv[1]) : 512, reps = argc > 2 ? std::atoi(argv[2]) : 40;
  const int m = 4 * n;
  Worker w[2];
  for (int t = 0; t < 2; ++t) {
    w[t].n = n;
    w[t].m = m;
    CK(hipStreamCreateWithFlags(&w[t].s, hipStreamNonBlocking));
    CK(rocblas_create_handle(&w[t].h));
    CK(rocblas_set_stream(w[t].h, w[t].s));
    CK(hipMalloc(&w[t].dA, sizeof(Z) * n * n));
    CK(hipMalloc(&w[t].dB, sizeof(Z) * m * n));
    CK(hipMalloc(&w[t].dC, sizeof(Z) * m * m));
    CK(hipMalloc(&w[t].dinfo, 16));
    std::mt19937_64 g(1234 + t);
    std::normal_distribution<double> N(0.0, 1.0);
    std::vector<Z> M((size_t)n * n);
    for (auto& v : M) v = Z(N(g), N(g));
    w[t].A.assign((size_t)n * n, Z(0, 0));
    for (int i = 0; i < n; ++i)
      for (int j = 0; j <= i; ++j) {  // A = M M^H / n + I, lower triangle (Hermitian positive definite)
        Z sum(0, 0);
        for (int k = 0; k < n; ++k) sum += M[i + (size_t)k * n] * std::conj(M[j + (size_t)k * n]);
        w[t].A[i + (size_t)j * n] = sum / (double)n + (i == j ? Z(1.0, 0) : Z(0, 0));
      }
    w[t].B.resize((size_t)m * n);
    for (auto& v : w[t].B) v = Z(N(g), N(g));
  }
  const char* names[3] = {"rocsolver_zpotrf", "rocblas_ztrsm", "rocblas_zherk"};
  int total_bad = 0;
  for (int op = 0; op < 3; ++op) {
    std::vector<Z> ref[2];
    int info_ref[2];
    for (int t = 0; t < 2; ++t) run_op(w[t], (Op)op, ref[t], &info_ref[t]);  // one at a time: the reference
    int bad[2] = {0, 0}, bad_info[2] = {0, 0};
    auto body = [&](int t) {
      std::vector<Z> out;
      for (int r = 0; r < reps; ++r) {
        int info = 0;
        run_op(w[t], (Op)op, out, &info);
        if (info != info_ref[t]) ++bad_info[t];
        if (std::memcmp(out.data(), ref[t].data(), sizeof(Z) * out.size()) != 0) ++bad[t];
      }
    };
    std::thread t0(body, 0), t1(body, 1);
    t0.join();
    t1.join();
    std::printf("%-18s n = %d: two handles / two streams / two threads, %d reps each: %d + %d results differ from the solo run, %d + %d info words differ\n",
                names[op], n, reps, bad[0], bad[1], bad_info[0], bad_info[1]);
    total_bad += bad[0] + bad[1];
    // the same loop on ONE thread at a time: must be clean
    int solo_bad = 0;
    for (int t = 0; t < 2; ++t) {
      std::vector<Z> out;
      for (int r = 0; r < 5; ++r) {
        int info = 0;
        run_op(w[t], (Op)op, out, &info);
        if (std::memcmp(out.data(), ref[t].data(), sizeof(Z) * out.size()) != 0) ++solo_bad;
      }
    }
    std::printf("%-18s          one at a time again: %d results differ\n", names[op], solo_bad);
  }
  std::printf("verdict: %s\n", total_bad ? "NOT safe with two handles running concurrently in one process (this ROCm build)" : "no interference observed");
  return 0;
}
