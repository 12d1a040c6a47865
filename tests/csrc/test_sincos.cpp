#include <cstdio>
#include <random>
#include "nls_sincos.h"
static const nls::SinCosCoef K = nls::sincos_coef();
static const nls::SinCosTabCoef KT = nls::sincos_tab_coef();
static double TAB[4 * nls::SINCOS_TAB_N];
int main() {
  nls::sincos_tab_fill(TAB);
  std::mt19937_64 rng(7);
  double worst_s = 0, worst_c = 0, at_s = 0, at_c = 0, worst_ts = 0, worst_tc = 0, at_ts = 0, at_tc = 0;
  auto check = [&](double t) {
    double s, c;
    nls::sincos_fast(t, s, c, K);
    long double rs = sinl((long double)t), rc = cosl((long double)t);
    double es = (double)fabsl((long double)s - rs), ec = (double)fabsl((long double)c - rc);
    if (es > worst_s) { worst_s = es; at_s = t; }
    if (ec > worst_c) { worst_c = ec; at_c = t; }
    if (fabs(t) <= 1073741824.0) {  // the table form the kernels run (same range)
      nls::sincos_table(t, s, c, KT, TAB);
      es = (double)fabsl((long double)s - rs), ec = (double)fabsl((long double)c - rc);
      if (es > worst_ts) { worst_ts = es; at_ts = t; }
      if (ec > worst_tc) { worst_tc = ec; at_tc = t; }
    }
  };
  std::uniform_real_distribution<double> U(-1, 1);
  for (int e = -30; e <= 30; ++e)
    for (int i = 0; i < 400000; ++i) check(ldexp(U(rng), e));
  // near multiples of pi/2 (worst case for the reduction)
  for (int n = -2000000; n <= 2000000; n += 7) { double t = n * 1.5707963267948966; check(t); check(nextafter(t, 1e300)); check(nextafter(t, -1e300)); }
  printf("max abs err sin %.3e at %.17g, cos %.3e at %.17g\n", worst_s, at_s, worst_c, at_c);
  printf("table form: max abs err sin %.3e at %.17g, cos %.3e at %.17g\n", worst_ts, at_ts, worst_tc, at_tc);
  { double s2, c2; nls::sincos_table(NAN, s2, c2, KT, TAB); printf("table nan: %g %g\n", s2, c2);
    nls::sincos_table(0.0, s2, c2, KT, TAB); printf("table zero: %g %g\n", s2, c2);
    nls::sincos_table(-0.0, s2, c2, KT, TAB); printf("table -zero: %g %g\n", s2, c2); }
  if (!(worst_ts < 1.2e-16 && worst_tc < 1.2e-16)) return 2;
  double s, c; nls::sincos_fast(1e300, s, c, K); printf("huge: %g %g\n", s, c);
  nls::sincos_fast(NAN, s, c, K); printf("nan: %g %g\n", s, c);
  nls::sincos_fast(0.0, s, c, K); printf("zero: %g %g\n", s, c);
  return (worst_s < 3e-16 && worst_c < 3e-16) ? 0 : 1;
}
