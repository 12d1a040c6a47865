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
