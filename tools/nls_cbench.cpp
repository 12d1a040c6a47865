// Native driver of the C ABI (no Python): times nls_featuremap / nls_gram_only / nls_primal_fit on random data.
// Used under rocprofv3 (the program itself goes after "--") and as a minimal C integration example.
//   g++ -O2 tools/nls_cbench.cpp -Iinclude -Lneo_ls_svm_amd -lneolssvm_hip -Wl,-rpath,'$ORIGIN/../neo_ls_svm_amd' -o tools/nls_cbench
#include <chrono>
#include <cmath>
#include <cstdio>
#include <cstdlib>
#include <cstring>
#include <random>
#include <vector>

#include "neolssvm_hip.h"

static double now() { return std::chrono::duration<double>(std::chrono::steady_clock::now().time_since_epoch()).count(); }

int main(int argc, char** argv) {
  long n = argc > 1 ? atol(argv[1]) : 65536;
  int d = argc > 2 ? atoi(argv[2]) : 128, D = argc > 3 ? atoi(argv[3]) : 4096, G = argc > 4 ? atoi(argv[4]) : 1024;
  const char* what = argc > 5 ? argv[5] : "fit";
  int reps = argc > 6 ? atoi(argv[6]) : 2;
  nls_ctx* ctx = nullptr;
  if (nls_ctx_create(0, &ctx)) { fprintf(stderr, "ctx: %s\n", nls_last_error(nullptr)); return 1; }
  std::mt19937_64 rng(1);
  std::normal_distribution<double> N01(0, 1);
  std::vector<double> X((size_t)n * d), y(n), s(n, 1.0), shift(d, 0.0), scale(d, 1.0), B((size_t)d * D), gam(G);
  for (auto& v : X) v = N01(rng);
  for (auto& v : B) v = N01(rng) / std::sqrt((double)d);
  for (long i = 0; i < n; ++i) y[i] = std::sin(X[(size_t)i * d]) + 0.1 * N01(rng);
  for (int g = 0; g < G; ++g) gam[g] = std::pow(10.0, std::log10(1e-6) + (std::log10(20.0) - std::log10(1e-6)) * g / (G > 1 ? G - 1 : 1));
  void *dX, *dy, *ds;
  nls_device_malloc(ctx, X.size() * 8, &dX); nls_device_malloc(ctx, n * 8, &dy); nls_device_malloc(ctx, n * 8, &ds);
  nls_memcpy_h2d(ctx, dX, X.data(), X.size() * 8); nls_memcpy_h2d(ctx, dy, y.data(), n * 8); nls_memcpy_h2d(ctx, ds, s.data(), n * 8);
  const int D1 = D + 1;
  std::vector<double> beta(2 * D1), errs(G), tm(NLS_NUM_TIMINGS), res(n);
  double score; int opt;
  for (int rep = 0; rep < reps; ++rep) {
    double t0 = now();
    int rc = 0;
    if (!strcmp(what, "fit")) {
      nls_primal_fit_args a; memset(&a, 0, sizeof(a));
      a.X = (double*)dX; a.y = (double*)dy; a.s = (double*)ds; a.shift = shift.data(); a.scale = scale.data(); a.B = B.data();
      a.gammas = gam.data(); a.n = n; a.d = d; a.D = D; a.G = G; a.is_classifier = 0; a.gamma_index_in = -1;
      a.beta = beta.data(); a.loo_errors = errs.data(); a.loo_residuals = res.data(); a.loo_score = &score; a.gamma_index = &opt; a.timings = tm.data();
      rc = nls_primal_fit(ctx, &a);
    } else if (!strcmp(what, "rotate")) {
      static std::vector<double> Q, v;
      if (Q.empty()) { Q.resize((size_t)2 * D1 * D1); v.resize(2 * D1); for (auto& q : Q) q = N01(rng) / std::sqrt((double)D1); for (auto& q : v) q = N01(rng); }
      rc = nls_rotate_only(ctx, (double*)dX, n, d, shift.data(), scale.data(), B.data(), D, Q.data(), v.data(), nullptr, nullptr);
    } else if (!strcmp(what, "gram")) {
      rc = nls_gram_only(ctx, (double*)dX, (double*)dy, (double*)ds, n, d, shift.data(), scale.data(), B.data(), D, nullptr, nullptr);
    } else {
      void* dphi; nls_device_malloc(ctx, (size_t)n * D1 * 16, &dphi);
      rc = nls_featuremap(ctx, (double*)dX, n, d, shift.data(), scale.data(), B.data(), D, (double*)dphi);
      nls_device_free(ctx, dphi);
    }
    nls_synchronize(ctx);
    if (rc) { fprintf(stderr, "error %d: %s\n", rc, nls_last_error(ctx)); return 1; }
    printf("%s n=%ld d=%d D=%d G=%d: %.3f s", what, n, d, D, G, now() - t0);
    if (!strcmp(what, "fit"))
      printf("  [fm %.1f gram %.1f evd %.1f rot %.1f sweep %.1f loo %.1f chol %.1f ms] opt=%d score=%.4f", tm[NLS_T_FEATUREMAP] * 1e3,
             tm[NLS_T_GRAM] * 1e3, tm[NLS_T_EVD] * 1e3, tm[NLS_T_ROTATE] * 1e3, tm[NLS_T_SWEEP] * 1e3, tm[NLS_T_LOO] * 1e3, tm[NLS_T_CHOLESKY] * 1e3, opt, score);
    printf("\n");
  }
  nls_ctx_destroy(ctx);
  return 0;
}
