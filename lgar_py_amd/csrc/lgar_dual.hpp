// lgar_dual.hpp -- forward-mode dual numbers for the differentiable LGAR path.
//
// The reference differentiates dpLGAR.forward() with torch autograd over 0-d tensors
// (/root/reference/dpLGAR/agents/DifferentiableLGAR.py:119,163).  Here the same device physics
// (lgar_device.hpp, templated on the scalar type) is instantiated with Dual<R> = value + one
// tangent: control flow looks only at values, every arithmetic op carries d/dp along.
// Derivative conventions follow torch: min() passes the gradient of the smaller argument (half
// each on ties), abs() uses sign(x), pow() has d/dx = y x^(y-1) and d/dy = x^y ln x.
#pragma once
#include "lgar_device.hpp"

namespace lgar {

template <typename R> struct Dual {
  R v, d;
  __device__ __forceinline__ Dual() {}
  __device__ __forceinline__ explicit Dual(R a) : v(a), d(R(0)) {}
  __device__ __forceinline__ Dual(R a, R b) : v(a), d(b) {}
};
template <typename R> struct Real<Dual<R>> { using type = R; };

template <typename R> __device__ __forceinline__ R val(const Dual<R> &x) { return x.v; }
template <typename R> __device__ __forceinline__ Dual<R> choose(bool c, const Dual<R> &a, const Dual<R> &b) {
  return Dual<R>(c ? a.v : b.v, c ? a.d : b.d);
}

#define LGAR_DUAL_BIN(OP, VEXPR, DEXPR_DD, DEXPR_DR, DEXPR_RD)                                                          \
  template <typename R> __device__ __forceinline__ Dual<R> operator OP(const Dual<R> &a, const Dual<R> &b) {            \
    const R av = a.v, bv = b.v, ad = a.d, bd = b.d; (void)av; (void)bv; (void)ad; (void)bd;                             \
    return Dual<R>(VEXPR, DEXPR_DD);                                                                                    \
  }                                                                                                                     \
  template <typename R> __device__ __forceinline__ Dual<R> operator OP(const Dual<R> &a, R bv) {                        \
    const R av = a.v, ad = a.d; (void)av; (void)ad;                                                                     \
    return Dual<R>(VEXPR, DEXPR_DR);                                                                                    \
  }                                                                                                                     \
  template <typename R> __device__ __forceinline__ Dual<R> operator OP(R av, const Dual<R> &b) {                        \
    const R bv = b.v, bd = b.d; (void)bv; (void)bd;                                                                     \
    return Dual<R>(VEXPR, DEXPR_RD);                                                                                    \
  }

LGAR_DUAL_BIN(+, av + bv, ad + bd, ad, bd)
LGAR_DUAL_BIN(-, av - bv, ad - bd, ad, -bd)
LGAR_DUAL_BIN(*, av * bv, ad * bv + av * bd, ad * bv, av * bd)
LGAR_DUAL_BIN(/, av / bv, (ad - (av / bv) * bd) / bv, ad / bv, -((av / bv) * bd) / bv)
#undef LGAR_DUAL_BIN

template <typename R> __device__ __forceinline__ Dual<R> operator-(const Dual<R> &a) { return Dual<R>(-a.v, -a.d); }

// torch.pow: d/dx = y x^(y-1), d/dy = x^y ln x; at x <= 0 the tangent is dropped (the value path flags NaN there)
template <typename R> __device__ __forceinline__ Dual<R> pw(const Dual<R> &x, const Dual<R> &y) {
  // one log2 serves both the value and d/dy: x^y = 2^(y log2 x), ln x = ln 2 * log2 x
  const R l2 = lg2(x.v);
  const R v = ex2(y.v * l2);
  R d = R(0);
  if (x.v > R(0)) d = v * (y.d * (R(0.6931471805599453) * l2) + y.v * x.d / x.v);
  return Dual<R>(v, d);
}
// division policy (dv in lgar_device.hpp): dual numbers always divide exactly
template <int POL, typename R> __device__ __forceinline__ Dual<R> dv(const Dual<R> &a, const Dual<R> &b) { return a / b; }
template <int POL, typename R> __device__ __forceinline__ Dual<R> dv(const Dual<R> &a, R b) { return a / b; }
template <int POL, typename R> __device__ __forceinline__ Dual<R> dv(R a, const Dual<R> &b) { return a / b; }
// verification mode (see pwx in lgar_device.hpp): correctly rounded pow / log for the value and the derivative
template <bool EX, typename R> __device__ __forceinline__ Dual<R> pwx(const Dual<R> &x, const Dual<R> &y) {
  if constexpr (EX && sizeof(R) == 8) {
    const R v = pow(x.v, y.v);
    R d = R(0);
    if (x.v > R(0)) d = v * (y.d * log(x.v) + y.v * x.d / x.v);
    return Dual<R>(v, d);
  }
  return pw(x, y);
}
template <typename R> __device__ __forceinline__ Dual<R> sq(const Dual<R> &x) {
  const R v = sq(x.v);
  return Dual<R>(v, (v > R(0)) ? x.d / (R(2) * v) : R(0));
}
template <typename R> __device__ __forceinline__ Dual<R> ab(const Dual<R> &x) {
  return Dual<R>(ab(x.v), (x.v > R(0)) ? x.d : ((x.v < R(0)) ? -x.d : R(0)));
}
template <typename R> __device__ __forceinline__ Dual<R> mn(const Dual<R> &a, const Dual<R> &b) {
  if (a.v < b.v) return a;
  if (b.v < a.v) return b;
  return Dual<R>(a.v, R(0.5) * (a.d + b.d));
}

// log2 / exp2 (the fused Geff node, lgar_device.hpp): d log2 x = dx / (x ln 2), d 2^y = 2^y ln 2 dy
template <typename R> __device__ __forceinline__ Dual<R> lg2(const Dual<R> &x) {
  return Dual<R>(lg2(x.v), (x.v > R(0)) ? x.d / (x.v * R(0.6931471805599453)) : R(0));
}
template <typename R> __device__ __forceinline__ Dual<R> ex2(const Dual<R> &y) {
  const R v = ex2(y.v);
  return Dual<R>(v, v * R(0.6931471805599453) * y.d);
}
// inside the Geff trapezoid (x > 0, y finite): the select-free cores, and the tangent's 1/x by reciprocal + Newton
template <typename R> __device__ __forceinline__ Dual<R> lg2p(const Dual<R> &x) {
  return Dual<R>(lg2p(x.v), x.d * (fast_recip(x.v) * R(1.4426950408889634)));
}
template <typename R> __device__ __forceinline__ Dual<R> ex2p(const Dual<R> &y) {
  const R v = ex2p(y.v);
  return Dual<R>(v, v * R(0.6931471805599453) * y.d);
}
#ifndef LGAR_NO_FUSED_GEFF
template <> __device__ __forceinline__ Dual<double> geff<Dual<double>>(const LayerK<Dual<double>> &l, Dual<double> t1,
                                                                       Dual<double> t2, int nint) {
  return geff_fused<Dual<double>>(l, t1, t2, nint);
}
template <> __device__ __forceinline__ Dual<float> geff<Dual<float>>(const LayerK<Dual<float>> &l, Dual<float> t1,
                                                                     Dual<float> t2, int nint) {
  return geff_fused<Dual<float>>(l, t1, t2, nint);
}
#endif

}  // namespace lgar
