// sin and cos of one fp64 argument for the feature map exp(-i t) (P2: _feature_maps.py:197-200).
//
// Why not ocml's sincos: it evaluates the Payne-Hanek large-argument reduction (v_trig_preop_f64 x 3) AND the small-
// argument path for every element and selects - ~170 vector instructions per call.  On gfx950 the fp64 MFMA shares its
// datapath with the vector ALU (tools/probe_f64_coexec.hip: no overlap even between different waves of a SIMD), so in
// K1 those instructions add to the matrix time: 18 of the 31.6 ms per 10^6 x 4096 features were sincos.
//
// Here: n = rint(t 2/pi);  r = fma(-n, c2, fma(-n, c1, t)) with c1 + c2 = pi/2 to 107 bits - the fused multiply-adds
// subtract the exact products, so r is correct to half an ulp of r plus |n| 1.5e-33 for |t| <= 2^30 - then the two
// minimax polynomials on [-pi/4, pi/4] of fdlibm's __kernel_sin / __kernel_cos (public domain, Sun Microsystems 1993:
// < 1 ulp) and a quadrant select: ~35 vector instructions.  |t| > 2^30 and Inf take the library routine (the K1
// kernels test the range once per thread, over all of its accumulators).
// Compiles for host and device (tests/test_sincos_cpu.py checks it against long double on the CPU).
#pragma once
#include <cmath>
#if defined(__HIPCC__)
#include <hip/hip_runtime.h>
#define NLS_HD __host__ __device__ __forceinline__
#else
#define NLS_HD inline
#endif

namespace nls {

// The constants travel as kernel arguments: they then sit in SGPRs and enter the fp64 instructions as scalar operands.
// As literals the compiler materialises each one into a VGPR pair with two v_mov_b32 per use (fp64 instructions take no
// 64-bit literal) - 20 extra vector instructions per sincos, each of which costs matrix-pipe time in K1.
struct SinCosCoef {
  double two_over_pi, pio2_hi, pio2_lo;
  double s[6], c[6];
};
inline SinCosCoef sincos_coef() {
  SinCosCoef k;
  k.two_over_pi = 6.36619772367581382433e-01;
  k.pio2_hi = 1.5707963267948966;     // pi / 2 rounded:         0x1.921fb54442d18p+0
  k.pio2_lo = 6.123233995736766e-17;  // pi / 2 - that, rounded: 0x1.1a62633145c07p-54
  const double S[6] = {-1.66666666666666324348e-01, 8.33333333332248946124e-03, -1.98412698298579493134e-04,
                       2.75573137070700676789e-06, -2.50507602534068634195e-08, 1.58969099521155010221e-10};
  const double C[6] = {4.16666666666666019037e-02, -1.38888888888741095749e-03, 2.48015872894767294178e-05,
                       -2.75573143513906633035e-07, 2.08757232129817482790e-09, -1.13596475577881948265e-11};
  for (int i = 0; i < 6; ++i) {
    k.s[i] = S[i];
    k.c[i] = C[i];
  }
  return k;
}

// (sin t, cos t) for |t| <= 2^30 (NaN in -> NaN out); the caller has tested the range.
NLS_HD void sincos_reduced_full(double t, double& s, double& c, const SinCosCoef& k) {
  const double fn = rint(t * k.two_over_pi);
  double r = fma(-fn, k.pio2_hi, t);
  r = fma(-fn, k.pio2_lo, r);
  const double z = r * r;
  const double ps = fma(z, fma(z, fma(z, fma(z, fma(z, k.s[5], k.s[4]), k.s[3]), k.s[2]), k.s[1]), k.s[0]);
  const double sr = fma(r * z, ps, r);
  const double pc = fma(z, fma(z, fma(z, fma(z, fma(z, k.c[5], k.c[4]), k.c[3]), k.c[2]), k.c[1]), k.c[0]);
  // cos r = 1 - z/2 + z^2 pc with two fused steps (fdlibm compensates the rounding of 1 - z/2 with four more operations;
  // without them the error stays below 1.2 ulp of 1, checked by tests/test_sincos_cpu.py)
  const double cr = fma(z * z, pc, fma(-0.5, z, 1.0));
  const int q = (int)fn;  // exact: |fn| < 2^30
  const bool swap = q & 1;
  const double s0 = swap ? cr : sr, c0 = swap ? sr : cr;
  // sin flips sign in quadrants 2, 3 (bit 1 of q), cos in quadrants 1, 2 (bit 1 of q + 1): move that bit onto the sign bit
#if defined(__HIP_DEVICE_COMPILE__)
  s = __hiloint2double(__double2hiint(s0) ^ (int)(((unsigned)q << 30) & 0x80000000u), __double2loint(s0));
  c = __hiloint2double(__double2hiint(c0) ^ (int)(((unsigned)(q + 1) << 30) & 0x80000000u), __double2loint(c0));
#else
  s = (q & 2) ? -s0 : s0;
  c = ((q + 1) & 2) ? -c0 : c0;
#endif
}

// ---- table form (what the kernels run since round 3) ----------------------------------------------------------------------------
// One reduction straight onto a 256-entry table of the full circle: n = rint(t 128/pi) (magic-number rounding: the low bits of the
// biased sum ARE n mod 256), r0 = t - n pi/128 in two fused steps (pi/128 to 107 bits; |r0| <= pi/256), then
//   sin t = S + (S_lo + S (cos r0 - 1) + C sin r0),   cos t = C + (C_lo + C (cos r0 - 1) - S sin r0)
// with (S, C) = (sin, cos)(2 pi n / 256) held as hi + lo pairs and degree-5 / degree-6 Taylor polynomials in r0 (truncation < 1e-17):
// the bracket is < 0.013, so the only rounding that matters is the final add - error <= 0.5 ulp + 2e-18, no quadrant logic, 19 vector
// instructions + two 16-byte LDS reads instead of ~35.  Valid for |t| <= 2^30 like the form above.
constexpr int SINCOS_TAB_N = 256;
struct SinCosTabCoef {
  double n_over_pi, h_hi, h_lo, magic;
  double s3, s5, c2, c4, c6;
};
inline SinCosTabCoef sincos_tab_coef() {
  SinCosTabCoef k;
  k.n_over_pi = 40.74366543152520595687;    // 128 / pi
  k.h_hi = 0.0245436926061702596754;        // pi / 128 rounded:         0x1.921fb54442d18p-6
  k.h_lo = 9.56755311833869685e-19;         // pi / 128 - that, rounded: 0x1.1a62633145c07p-60
  k.magic = 6755399441055744.0;             // 1.5 * 2^52
  k.s3 = -1.0 / 6.0;
  k.s5 = 1.0 / 120.0;
  k.c2 = -0.5;
  k.c4 = 1.0 / 24.0;
  k.c6 = -1.0 / 720.0;
  return k;
}
// tab[4 n + {0,1,2,3}] = {S_hi, C_hi, S_lo, C_lo} of the angle 2 pi n / 256 (host, long double)
inline void sincos_tab_fill(double* tab) {
  const long double two_pi = 6.283185307179586476925286766559005768L;
  for (int n = 0; n < SINCOS_TAB_N; ++n) {
    long double sv = sinl(two_pi * n / SINCOS_TAB_N), cv = cosl(two_pi * n / SINCOS_TAB_N);
    if (n % 64 == 0) {  // exact values on the axes
      sv = (n == 64) ? 1.0L : (n == 192 ? -1.0L : 0.0L);
      cv = (n == 0) ? 1.0L : (n == 128 ? -1.0L : 0.0L);
    }
    const double sh = (double)sv, ch = (double)cv;
    tab[4 * n + 0] = sh;
    tab[4 * n + 1] = ch;
    tab[4 * n + 2] = (double)(sv - (long double)sh);
    tab[4 * n + 3] = (double)(cv - (long double)ch);
  }
}
NLS_HD void sincos_table(double t, double& s, double& c, const SinCosTabCoef& k, const double* tab) {
  const double fm = fma(t, k.n_over_pi, k.magic);
  const double fn = fm - k.magic;
#if defined(__HIP_DEVICE_COMPILE__)
  const int idx = __double2loint(fm) & (SINCOS_TAB_N - 1);
#else
  long long bits;
  __builtin_memcpy(&bits, &fm, 8);
  const int idx = (int)(bits & (SINCOS_TAB_N - 1));
#endif
  double r0 = fma(-fn, k.h_hi, t);
  r0 = fma(-fn, k.h_lo, r0);
  const double z0 = r0 * r0;
  const double sr0 = fma(r0, fma(z0, k.s5, k.s3) * z0, r0);              // sin r0
  const double cm = fma(z0, fma(z0, k.c6, k.c4), k.c2) * z0;             // cos r0 - 1
  const double S = tab[4 * idx], C = tab[4 * idx + 1], Sl = tab[4 * idx + 2], Cl = tab[4 * idx + 3];
  s = S + fma(C, sr0, fma(S, cm, Sl));
  c = C + fma(-S, sr0, fma(C, cm, Cl));
}

NLS_HD void sincos_fast(double t, double& s, double& c, const SinCosCoef& k) {
  if (fabs(t) <= 1073741824.0 || t != t) return sincos_reduced_full(t, s, c, k);
  // |t| > 2^30, Inf: rare, the library's full-range routine
#if defined(__HIP_DEVICE_COMPILE__)
  ::sincos(t, &s, &c);
#else
  s = std::sin(t);
  c = std::cos(t);
#endif
}

}  // namespace nls
