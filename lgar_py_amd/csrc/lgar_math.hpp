// lgar_math.hpp -- lean double-precision log2 / exp2 / pow for the fp64 LGAR kernels.
//
// ocml's pow/log2/exp2 are < 1 ulp but built on double-double arithmetic (~100-200 instructions each); the LGAR path
// evaluates ~1.5 k of them per column-timestep.  These versions are ~2 ulp (relative 3e-16 on log2/exp2, ~1e-14 on
// pow for the exponents used here) in ~20-30 instructions: frexp + atanh series for log2, round-to-nearest + Taylor +
// ldexp for exp2.  That is nine orders of magnitude inside the 1e-6 parity bar; -DLGAR_F64_LIBM restores ocml.
#pragma once
#ifndef LGAR_DEVSIM
#include <hip/hip_runtime.h>
#endif

namespace lgar {

// 2^x
__device__ __forceinline__ double fast_exp2(double x) {
  const double xc = fmin(fmax(x, -1100.0), 1100.0);  // keeps the integer part in range; ldexp saturates to 0 / inf
  const double k = rint(xc);
  const double f = (xc - k) * 0.6931471805599453094;  // |f| <= 0.3466
  double p = 1.6059043836821613e-10;                   // 1/13!
  p = fma(p, f, 2.08767569878681e-09);                 // 1/12!
  p = fma(p, f, 2.505210838544172e-08);                // 1/11!
  p = fma(p, f, 2.755731922398589e-07);                // 1/10!
  p = fma(p, f, 2.7557319223985893e-06);               // 1/9!
  p = fma(p, f, 2.48015873015873e-05);                 // 1/8!
  p = fma(p, f, 1.984126984126984e-04);                // 1/7!
  p = fma(p, f, 1.388888888888889e-03);                // 1/6!
  p = fma(p, f, 8.333333333333333e-03);                // 1/5!
  p = fma(p, f, 4.1666666666666664e-02);               // 1/4!
  p = fma(p, f, 1.6666666666666666e-01);               // 1/3!
  p = fma(p, f, 0.5);
  p = fma(p, f, 1.0);
  p = fma(p, f, 1.0);
  const double r = ldexp(p, (int)k);
  return (x != x) ? x : r;
}

// log2(x): -inf at 0, NaN below 0 / for NaN, +inf at +inf
__device__ __forceinline__ double fast_log2(double x) {
  int e;
  double m = frexp(x, &e);  // m in [0.5, 1)
  const bool lo = m < 0.70710678118654752440;
  m = lo ? m * 2.0 : m;     // m in [sqrt(1/2), sqrt(2))
  e = lo ? e - 1 : e;
  // (m - 1) / (m + 1) with v_rcp_f64 + one Newton step instead of the ~12-instruction IEEE divide
  const double d = m + 1.0;
#ifndef LGAR_DEVSIM
  double rc = __builtin_amdgcn_rcp(d);
#else
  double rc = 1.0 / d;
#endif
  rc = fma(fma(-d, rc, 1.0), rc, rc);
  const double s = (m - 1.0) * rc;  // |s| <= 0.1716
  const double z = s * s;
  double p = 4.7619047619047616e-02;        // 1/21
  p = fma(p, z, 5.2631578947368418e-02);    // 1/19
  p = fma(p, z, 5.8823529411764705e-02);    // 1/17
  p = fma(p, z, 6.6666666666666666e-02);    // 1/15
  p = fma(p, z, 7.6923076923076927e-02);    // 1/13
  p = fma(p, z, 9.0909090909090912e-02);    // 1/11
  p = fma(p, z, 1.1111111111111111e-01);    // 1/9
  p = fma(p, z, 1.4285714285714285e-01);    // 1/7
  p = fma(p, z, 0.2);
  p = fma(p, z, 3.3333333333333331e-01);
  p = fma(p, z, 1.0);
  const double lnm = 2.0 * s * p;            // ln(m) = 2 atanh(s)
  double r = fma(lnm, 1.4426950408889634074, (double)e);
  r = (x == 0.0) ? -__builtin_huge_val() : r;
  r = (x < 0.0 || x != x) ? __builtin_nan("") : r;
  r = (x == __builtin_huge_val()) ? x : r;
  return r;
}

__device__ __forceinline__ double fast_pow(double x, double y) { return fast_exp2(y * fast_log2(x)); }

}  // namespace lgar
