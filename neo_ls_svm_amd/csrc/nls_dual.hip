// Dual path (D1-D6): placeholder translation unit, replaced by the real kernels in the next milestone.
#include "../../include/neolssvm_hip.h"

extern "C" int nls_dual_fit(nls_ctx* ctx, const nls_dual_fit_args* args) {
  (void)ctx;
  (void)args;
  return NLS_ERR_ARG;
}
extern "C" int nls_dual_predict(nls_ctx* ctx, const double* Xq, int64_t m, const double* Xt, int64_t n, int r,
                                const double* alpha, const double* L, double* yhat, double* sigma) {
  (void)ctx; (void)Xq; (void)m; (void)Xt; (void)n; (void)r; (void)alpha; (void)L; (void)yhat; (void)sigma;
  return NLS_ERR_ARG;
}
