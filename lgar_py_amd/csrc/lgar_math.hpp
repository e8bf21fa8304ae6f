// lgar_math.hpp -- lean double-precision log2 / exp2 / pow for the fp64 LGAR kernels.
//
// ocml's pow/log2/exp2 are < 1 ulp but built on double-double arithmetic (~100-200 instructions each); the LGAR path
// evaluates ~1.5 k of them per column-timestep (the fp64 kernel spends ~88 % of its time in the Geff trapezoid, four of
// these per node).  The versions here take ~20 instructions each:
//   exp2: round-to-nearest split x = k + f, |f| <= 1/2, degree-11 near-minimax polynomial for 2^f (|error| <= 3.2e-18),
//         ldexp;
//   log2: x = m 2^e with m in [sqrt(1/2), sqrt(2)) (e from the exponent of x sqrt(2): no renormalisation selects),
//         s = (m - 1)/(m + 1), |s| <= 0.1716, log2 m = s Q(s^2) with Q a degree-7 near-minimax polynomial for
//         (2/ln 2) atanh(s)/s (|error| <= 3.4e-18); the quotient by v_rcp_f64 + one Newton step instead of the
//         ~12-instruction IEEE divide.
// Both are good to ~2.5e-16 (relative; log2 relative to max(1, |log2 x|), and relative to its own value near x = 1),
// i.e. pow is good to ~max(1, |y log2 x|) * 7e-16 relative, nine orders of magnitude inside the 1e-6 parity bar.
// The polynomial coefficients were fitted with mpmath.chebyfit at 60 digits (tests/test_device_math.py re-derives the
// error bounds).  search_mode 0 (verification) uses the library pow instead (pwx).
#pragma once
#ifndef LGAR_DEVSIM
#include <hip/hip_runtime.h>
#endif

namespace lgar {

// 2^x for finite x and -inf .. +inf clamped (CLAMP = false: finite |x| < 2^31 only, as inside the Geff trapezoid, where the
// exponent is a few thousand at most); NaN is NOT preserved (callers that need NaN to survive use fast_exp2)
template <bool CLAMP = true, bool ESTRIN = false> __device__ __forceinline__ double fast_exp2_core(double x) {
  const double xc = CLAMP ? fmin(fmax(x, -1100.0), 1100.0) : x;  // keeps the integer part in range; ldexp saturates to 0 / inf
  const double k = rint(xc);
  const double f = xc - k;  // |f| <= 0.5, exact
  double p;
  if constexpr (ESTRIN) {
  // degree-11 polynomial for 2^f.  The eight high coefficients (terms f^4 .. f^11, < 7e-4 of the result) are combined pairwise
  // (Estrin): three levels instead of seven; the last four steps stay a Horner chain, so the rounding of the result is that of
  // the plain Horner form.  A chain of dependent fma is what a lone wave waits on (a dependent v_fma_f64 cannot issue in the
  // slot after its producer): 7 levels instead of 11.
  const double f2 = f * f;
  const double a0 = fma(1.33335581464064708e-03, f, 9.61812910758725638e-03);  // c4 + c5 f
  const double a1 = fma(1.52527338415567733e-05, f, 1.54035304637243530e-04);  // c6 + c7 f
  const double a2 = fma(1.01780570877339407e-07, f, 1.32154325359123753e-06);  // c8 + c9 f
  const double a3 = fma(4.45581790833606449e-10, f, 7.07419429728852106e-09);  // c10 + c11 f
  const double f4 = f2 * f2;
  const double b0 = fma(a1, f2, a0);
  const double b1 = fma(a3, f2, a2);
  p = fma(b1, f4, b0);  // c4 + c5 f + ... + c11 f^7
  p = fma(p, f, 5.55041086648216248e-02);
  p = fma(p, f, 2.40226506959101582e-01);
  p = fma(p, f, 6.93147180559945286e-01);
  p = fma(p, f, 1.0);
  } else {
  p = 4.45581790833606449e-10;
  p = fma(p, f, 7.07419429728852106e-09);
  p = fma(p, f, 1.01780570877339407e-07);
  p = fma(p, f, 1.32154325359123753e-06);
  p = fma(p, f, 1.52527338415567733e-05);
  p = fma(p, f, 1.54035304637243530e-04);
  p = fma(p, f, 1.33335581464064708e-03);
  p = fma(p, f, 9.61812910758725638e-03);
  p = fma(p, f, 5.55041086648216248e-02);
  p = fma(p, f, 2.40226506959101582e-01);
  p = fma(p, f, 6.93147180559945286e-01);
  p = fma(p, f, 1.0);
  }
  return ldexp(p, (int)k);
}

// 2^x, NaN in -> NaN out
template <bool ESTRIN = false> __device__ __forceinline__ double fast_exp2(double x) {
  const double r = fast_exp2_core<true, ESTRIN>(x);
  return (x != x) ? x : r;
}

// log2(x) for positive finite x (anything else: unspecified, no trap)
template <bool ESTRIN = false> __device__ __forceinline__ double fast_log2_core(double x) {
  // x = m 2^e with m in [sqrt(1/2), sqrt(2)): e = exponent of x sqrt(2) (no compare-and-select renormalisation), so
  // x near 1 gives e = 0 and a tiny s: the result is accurate RELATIVE to itself there (1 - Se^(1/m) in K(Se) needs that)
  int e;
  (void)frexp(x * 1.41421356237309515, &e);
  e -= 1;
  const double m = ldexp(x, -e);
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
  double q;
  if constexpr (ESTRIN) {
  // degree-7 polynomial in z: the six high coefficients (z^2 .. z^7 terms, < 2e-4 of the sum) pairwise, the last two steps a
  // Horner chain (five levels instead of seven; see fast_exp2_core)
  const double z2 = z * z;
  const double a0 = fma(4.12198585840901910e-01, z, 5.77078016345520250e-01);  // q2 + q3 z
  const double a1 = fma(2.62334352512281266e-01, z, 3.20598534913810962e-01);  // q4 + q5 z
  const double a2 = fma(2.13658959211262989e-01, z, 2.20913084014299627e-01);  // q6 + q7 z
  q = fma(fma(a2, z2, a1), z2, a0);
  q = fma(q, z, 9.61796693925989765e-01);
  q = fma(q, z, 2.88539008177792677e+00);
  } else {
  q = 2.13658959211262989e-01;
  q = fma(q, z, 2.20913084014299627e-01);
  q = fma(q, z, 2.62334352512281266e-01);
  q = fma(q, z, 3.20598534913810962e-01);
  q = fma(q, z, 4.12198585840901910e-01);
  q = fma(q, z, 5.77078016345520250e-01);
  q = fma(q, z, 9.61796693925989765e-01);
  q = fma(q, z, 2.88539008177792677e+00);
  }
  return fma(s, q, (double)e);
}

// log2(x): -inf at 0, NaN below 0 / for NaN, +inf at +inf
template <bool ESTRIN = false> __device__ __forceinline__ double fast_log2(double x) {
  double r = fast_log2_core<ESTRIN>(x);
  r = (x == 0.0) ? -__builtin_huge_val() : r;
  r = (x < 0.0 || x != x) ? __builtin_nan("") : r;
  r = (x == __builtin_huge_val()) ? x : r;
  return r;
}

// 1/x for finite non-zero x to ~1 ulp: v_rcp_f64 + two Newton steps (the IEEE divide is ~2x the instructions); used where a
// quotient feeds a DERIVATIVE (dual numbers), never where the reference's own rounding matters
__device__ __forceinline__ double fast_recip(double x) {
#ifndef LGAR_DEVSIM
  double r = __builtin_amdgcn_rcp(x);
  r = fma(fma(-x, r, 1.0), r, r);
  r = fma(fma(-x, r, 1.0), r, r);
  return r;
#else
  return 1.0 / x;
#endif
}
__device__ __forceinline__ float fast_recip(float x) {
#ifndef LGAR_DEVSIM
  return __builtin_amdgcn_rcpf(x);  // v_rcp_f32, 1 ulp: feeds derivatives only
#else
  return 1.0f / x;
#endif
}

// ESTRIN: the polynomials' high-order terms combined pairwise -- seven and five levels of dependent fma instead of eleven and
// seven, for three more instructions per function.  Same coefficients, same last (Horner) steps; the result differs from the
// Horner form's in the last bit at most (tests/test_device_math.py holds both to the same bounds).  It pays where a wavefront
// waits on its own chain (the mixed-precision kernels, whose second wave spends most of its time in the fp32 trapezoid) and
// costs where the vector ALU is already full (the native fp64 kernels: their trapezoid IS these polynomials).
template <bool ESTRIN = false> __device__ __forceinline__ double fast_pow(double x, double y) { return fast_exp2<ESTRIN>(y * fast_log2<ESTRIN>(x)); }

}  // namespace lgar
