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
// The fused Geff node for dual numbers: the value is computed by the same operations in the same order as the plain node
// (lgar_device.hpp geff_node), the tangent in logarithmic form -- with a = x P, A = 1 + a, s = A^(-m/2), t = 1 - P s^2:
//   dln x = dx / x,  dln P = d(n-1) ln x + (n-1) dln x,  dln A = (a / A)(dln x + dln P),  dln s = d(-m/2) ln A - (m/2) dln A,
//   dt = -P s^2 (dln P + 2 dln s),  dK = dKsat s t^2 + K dln s + 2 Ksat s t dt
// (two reciprocals and ~25 multiply-adds per node instead of the ~60 operations of operator-by-operator propagation).
template <typename R>
__device__ __forceinline__ Dual<R> geff_node(const LayerK<Dual<R>> &l, const Dual<R> &nm1, const Dual<R> &half_m, const Dual<R> &h) {
  const R LN2 = R(0.6931471805599453);
  const R xv = l.alpha.v * h.v;
  const R xd = l.alpha.d * h.v + l.alpha.v * h.d;
  const R lg = lg2p(xv);
  const R Pv = ex2p(nm1.v * lg);
  const R av = xv * Pv;
  const R Av = R(1.0) + av;
  const R l1 = lg2p(Av);
  const R sv = ex2p(half_m.v * l1);
  const R Ps2 = Pv * (sv * sv);
  const R tv = R(1.0) - Ps2;
  const R ks = l.ksat.v * sv;
  const R tt = tv * tv;
  const R Kv = ks * tt;
  const R r = fast_recip(xv * Av);  // one reciprocal serves 1/x and 1/A
  const R dlnx = xd * (r * Av);
  const R dlnP = (nm1.d * LN2) * lg + nm1.v * dlnx;
  const R dlnA = (av * (r * xv)) * (dlnx + dlnP);
  const R dlns = (half_m.d * LN2) * l1 + half_m.v * dlnA;
  const R dt = -Ps2 * (dlnP + R(2.0) * dlns);
  const R Kd = l.ksat.d * (sv * tt) + Kv * dlns + (R(2.0) * (ks * tv)) * dt;
  return Dual<R>(Kv, Kd);
}
#ifndef LGAR_DEVSIM
// Eight nodes for eight lanes that integrate the SAME column along eight parameter directions (autograd.parameter_vjp lays
// them side by side, LgarDims.tangent_share): the values -- and with them every branch -- are identical in the eight lanes,
// only the tangents differ.  Lane r evaluates node r of an eight-node block and leaves the result in the wave's LDS buffer; every
// lane then finishes all eight nodes with its own tangent.  The VALUES and their sums are those of eight passes of the plain loop,
// bit for bit: same operations on the same operands, only computed once instead of eight times.
// The tangent of a node is LINEAR in what differs between the eight lanes -- dx = d(alpha h), d(n-1), d(-m/2), dKsat:
//   dK = A1 dx + A2 d(n-1) + A3 d(-m/2) + A4 dKsat,  with (B = 2 Ksat s t P s^2, W = K - 2 B, U = -m/2 (a/A) W)
//   A1 = ((n U - (n-1) B) / x,  A2 = ln x (U - B),  A3 = ln A W,  A4 = s t^2
// (geff_node's logarithmic-form tangent, collected by input).  So the lane that evaluates node r also evaluates K and A1..A4,
// and the other seven finish that node with four multiply-adds.
__device__ __forceinline__ void geff_block8(const LayerK<Dual<double>> &l, const Dual<double> &nm1, const Dual<double> &half_m,
                                            Dual<double> &h2, const Dual<double> &dh, const Dual<double> &hdh, Dual<double> &g,
                                            Dual<double> &k1, double *xchg) {
  const double LN2 = 0.6931471805599453;
  const int lane = (int)(threadIdx.x & 63u);
  const int r = lane & 7;
  // the eight heads, by the running sum of the plain loop
  Dual<double> h[8];
  h[0] = h2;
#pragma unroll
  for (int j = 1; j < 8; j++) h[j] = h[j - 1] + dh;
  double hv = h[0].v;
#pragma unroll
  for (int j = 1; j < 8; j++) hv = (r == j) ? h[j].v : hv;
  // my node: geff_node's value operations ...
  const double xv = l.alpha.v * hv;
  const double lg = lg2p(xv);
  const double Pv = ex2p(nm1.v * lg);
  const double av = xv * Pv;
  const double Av = 1.0 + av;
  const double l1 = lg2p(Av);
  const double sv = ex2p(half_m.v * l1);
  const double Ps2 = Pv * (sv * sv);
  const double tv = 1.0 - Ps2;
  const double ks = l.ksat.v * sv;
  const double tt = tv * tv;
  const double Kv = ks * tt;
  // ... and the coefficients of its tangent
  const double rc = fast_recip(xv * Av);  // one reciprocal serves 1/x and 1/A
  const double B = (2.0 * (ks * tv)) * Ps2;
  const double W = Kv - 2.0 * B;
  const double U = (half_m.v * (av * (rc * xv))) * W;
  const double A1 = (rc * Av) * ((1.0 + nm1.v) * U - nm1.v * B);
  const double A2 = (LN2 * lg) * (U - B);
  const double A3 = (LN2 * l1) * W;
  const double A4 = sv * tt;
  // one wave = one workgroup: LDS operations of a wave complete in order, the fences only pin the compiler
  __builtin_amdgcn_fence(__ATOMIC_RELEASE, "workgroup");
  xchg[0 * 64 + lane] = Kv;
  xchg[1 * 64 + lane] = A1;
  xchg[2 * 64 + lane] = A2;
  xchg[3 * 64 + lane] = A3;
  xchg[4 * 64 + lane] = A4;
  __builtin_amdgcn_fence(__ATOMIC_ACQ_REL, "workgroup");
  __builtin_amdgcn_wave_barrier();
  // node j of this block was evaluated by lane j of my group of eight (same address in all eight lanes: an LDS broadcast)
  const double *grp = xchg + (lane & ~7);
  double pr[8][5];
#pragma unroll
  for (int j = 0; j < 8; j++) {
#pragma unroll
    for (int v = 0; v < 5; v++) pr[j][v] = grp[v * 64 + j];
  }
  __builtin_amdgcn_fence(__ATOMIC_ACQ_REL, "workgroup");  // the next block's stores stay behind these loads
  __builtin_amdgcn_wave_barrier();
#pragma unroll
  for (int j = 0; j < 8; j++) {
    const double xd = l.alpha.d * h[j].v + l.alpha.v * h[j].d;
    const double Kd = pr[j][1] * xd + pr[j][2] * nm1.d + pr[j][3] * half_m.d + pr[j][4] * l.ksat.d;
    const Dual<double> k2(pr[j][0], Kd);
    g = g + ((k1 + k2) * hdh);
    k1 = k2;
  }
  h2 = h[7] + dh;
}
#endif
template <> __device__ __forceinline__ Dual<double> geff<Dual<double>>(const LayerK<Dual<double>> &l, Dual<double> t1,
                                                                       Dual<double> t2, int nint) {
  return geff_fused<Dual<double>>(l, t1, t2, nint);
}
template <> __device__ __forceinline__ Dual<float> geff<Dual<float>>(const LayerK<Dual<float>> &l, Dual<float> t1,
                                                                     Dual<float> t2, int nint) {
  return geff_fused<Dual<float>>(l, t1, t2, nint);
}

}  // namespace lgar
