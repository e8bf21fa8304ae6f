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
template <typename R> __device__ __forceinline__ bool same_bits(const Dual<R> &a, const Dual<R> &b) { return a.v == b.v && a.d == b.d; }
template <typename R> __device__ __forceinline__ Dual<R> choose(bool c, const Dual<R> &a, const Dual<R> &b) {
  return Dual<R>(c ? a.v : b.v, c ? a.d : b.d);
}

// a / b where the quotient feeds a DERIVATIVE only: in double precision by reciprocal (v_rcp_f64 + two Newton steps, ~1 ulp: a
// third of the instructions of the IEEE divide, in kernels whose every instruction is on a lone wave's critical path); single
// precision keeps its divide (the fp32 tangent kernels sit exactly at their register budget)
template <typename R> __device__ __forceinline__ R dquot(R a, R b) {
  if constexpr (sizeof(R) == 8) return a * fast_recip(b);
  else return a / b;
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
// (the value is the IEEE quotient the plain kernels form where they write a / b; the tangent divides by reciprocal)
LGAR_DUAL_BIN(/, av / bv, dquot(ad - (av / bv) * bd, bv), dquot(ad, bv), dquot(-((av / bv) * bd), bv))
#undef LGAR_DUAL_BIN

template <typename R> __device__ __forceinline__ Dual<R> operator-(const Dual<R> &a) { return Dual<R>(-a.v, -a.d); }

// torch.pow: d/dx = y x^(y-1), d/dy = x^y ln x; at x <= 0 the tangent is dropped (the value path flags NaN there)
template <typename R> __device__ __forceinline__ Dual<R> pw(const Dual<R> &x, const Dual<R> &y) {
  // one log2 serves both the value and d/dy: x^y = 2^(y log2 x), ln x = ln 2 * log2 x
  const R l2 = lg2(x.v);
  const R v = ex2(y.v * l2);
  R d = R(0);
  if (x.v > R(0)) d = v * (y.d * (R(0.6931471805599453) * l2) + dquot(y.v * x.d, x.v));
  return Dual<R>(v, d);
}
// division policy (dv in lgar_device.hpp).  The VALUE of a quotient goes through exactly what the plain kernels' dv does with the
// same policy (a tangent launch and the forward launch it differentiates must walk the same trajectory): in the double-precision
// fast modes that is lean_div; the tangent reuses its refined reciprocal instead of dividing a second and a third time.
template <int POL, typename R> __device__ __forceinline__ Dual<R> dv(const Dual<R> &a, const Dual<R> &b) {
  if constexpr ((POL == 0 || POL == 3) && sizeof(R) == 8) {
    const R q = lean_div(a.v, b.v);
    return Dual<R>(q, (a.d - q * b.d) * fast_recip(b.v));
  } else {
    return a / b;
  }
}
template <int POL, typename R> __device__ __forceinline__ Dual<R> dv(const Dual<R> &a, R b) {
  if constexpr ((POL == 0 || POL == 3) && sizeof(R) == 8) return Dual<R>(lean_div(a.v, b), a.d * fast_recip(b));
  else return a / b;
}
template <int POL, typename R> __device__ __forceinline__ Dual<R> dv(R a, const Dual<R> &b) {
  if constexpr ((POL == 0 || POL == 3) && sizeof(R) == 8) {
    const R q = lean_div(a, b.v);
    return Dual<R>(q, -(q * b.d) * fast_recip(b.v));
  } else {
    return a / b;
  }
}
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
  return Dual<R>(v, (v > R(0)) ? dquot(x.d, R(2) * v) : R(0));
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
  return Dual<R>(lg2(x.v), (x.v > R(0)) ? dquot(x.d, x.v * R(0.6931471805599453)) : R(0));
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
// The safe nodes of the trapezoid for W lanes that integrate the SAME column along W parameter directions
// (autograd.parameter_vjp lays them side by side, LgarDims.tangent_share = W): the values -- and with them every branch --
// are identical in the W lanes, only the tangents differ.
//
// VALUES.  Lane r evaluates node r of every W-node block and leaves K in the wave's LDS buffer; every lane then adds the W
// values up in order.  Values and their sums are those of the plain loop bit for bit: same operations on the same operands,
// computed once instead of W times.
//
// TANGENTS.  The tangent of a node is LINEAR in what differs between the lanes -- dx = d(alpha h), d(n-1), d(-m/2), dKsat
// (geff_node's logarithmic-form tangent, collected by input):
//   dK = A1 dx + A2 d(n-1) + A3 d(-m/2) + A4 dKsat,  with (B = 2 Ksat s t P s^2, W' = K - 2 B, U = -m/2 (a/A) W')
//   A1 = ((n U - (n-1) B) / x,  A2 = ln x (U - B),  A3 = ln A W',  A4 = s t^2,
// and dx_j = d alpha h_j + alpha (dh_0 + j d(dh)) for the node with head h_j = h_0 + j dh.  The trapezoid adds every block node
// twice (the last one once), so the tangent of the whole sum needs only FIVE direction-independent sums over the nodes:
//   sum A1, sum A1 j, sum A2, sum A3, sum A4        (dx_j = c0 + j c1: both constants of the lane)
// Each lane accumulates them over ITS nodes while the blocks go by -- nothing but K crosses lanes per block -- and the W
// partial sums are combined once at the end of the call.  Because nothing else is direction-dependent, ALL 3 x L directions of
// a column fit one group (W = 9 for three layers: 7 groups per wavefront), instead of 8 sharing lanes plus a ninth direction
// that ran the whole trapezoid alone.  (Round 2 exchanged K and A1..A4 of every node and had every lane redo the tangent of
// all eight nodes of a block: 45 LDS accesses and ~100 multiply-adds per block and lane; now W + 1 and ~30.)
#define LGAR_XCHG_GROUP 33  /* doubles per group in the K buffer: up to 32 lanes + 1 of padding (groups on different bank pairs) */
#define LGAR_XCHG_RED 9     /* values per lane in the end-of-call reduction area: five sums + the last node's four coefficients */
#define LGAR_XCHG_WORDS (64 + 32 + 64 * LGAR_XCHG_RED + 64) /* K buffer (<= 32 groups x padded stride < 128) + reduction area */
// WC: the group width as a compile-time constant (loops over the lanes of a group unroll), or 0: run-time width `Wrt`
template <int WC>
__device__ __forceinline__ void geff_shared_blocks_w(const LayerK<Dual<double>> &l, const Dual<double> &nm1, const Dual<double> &half_m,
                                                     Dual<double> &h2, const Dual<double> &dh, const Dual<double> &hdh, Dual<double> &g,
                                                     Dual<double> &k1, int nb, int Wrt, double *xchg, int rem) {
  const int W = WC ? WC : Wrt;
  const double LN2 = 0.6931471805599453;
  const int lane = (int)(threadIdx.x & 63u);
  const int grp_id = lane / W, r = lane - grp_id * W;
  double *grp = xchg + grp_id * (W + 1);  // padded: neighbouring groups start on different bank pairs
  double s1 = 0.0, s1j = 0.0, s2 = 0.0, s3 = 0.0, s4 = 0.0;  // my nodes' share of the five sums
  double a1 = 0.0, a2 = 0.0, a3 = 0.0, a4 = 0.0;                        // tangent coefficients of my latest node
  double gv = g.v, k1v = k1.v, hv = h2.v, pairsum = 0.0;
  // my node of a block: geff_node's value operations and the coefficients of its tangent
  auto node = [&](double hm, double &Kv, double &c1, double &c2, double &c3, double &c4) {
    const double xv = l.alpha.v * hm;
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
    Kv = ks * tt;
    const double rc = fast_recip(xv * Av);  // one reciprocal serves 1/x and 1/A
    const double B = (2.0 * (ks * tv)) * Ps2;
    const double Wk = Kv - 2.0 * B;
    const double U = (half_m.v * (av * (rc * xv))) * Wk;
    c1 = (rc * Av) * ((1.0 + nm1.v) * U - nm1.v * B);
    c2 = (LN2 * lg) * (U - B);
    c3 = (LN2 * l1) * Wk;
    c4 = sv * tt;
  };
  // every lane adds the first `cnt` values of a block up in order (same address in all lanes of the group: an LDS broadcast)
  auto add_up = [&](int cnt) {
    if constexpr (WC != 0) {
      double kq[WC ? WC : 1];
#pragma unroll
      for (int j = 0; j < WC; j++) kq[j] = grp[j];
#pragma unroll
      for (int j = 0; j < WC; j++)
        if (j < cnt) {  // (wave-uniform; always true in a full block)
          const double pr = k1v + kq[j];
          gv = gv + (pr * hdh.v);
          pairsum += pr;
          k1v = kq[j];
        }
    } else {
      for (int j0 = 0; j0 < cnt; j0 += 8) {  // eight reads in flight at a time
        double kq[8];
#pragma unroll
        for (int j = 0; j < 8; j++) kq[j] = grp[(j0 + j < W) ? j0 + j : W - 1];
#pragma unroll
        for (int j = 0; j < 8; j++)
          if (j0 + j < cnt) {
            const double pr = k1v + kq[j];
            gv = gv + (pr * hdh.v);
            pairsum += pr;
            k1v = kq[j];
          }
      }
    }
  };
  // MY head: number r of the plain loop's running sum, found once; from then on every block advances it by the W additions
  // of that running sum -- the same sequence of values (nine additions per block instead of nine compare-select-add steps)
  double hm = hv;
  if constexpr (WC != 0) {
#pragma unroll
    for (int j = 0; j < WC; j++) {
      hm = (j == r) ? hv : hm;
      hv = hv + dh.v;
    }
  } else {
    for (int j = 0; j < W; j++) {
      hm = (j == r) ? hv : hm;
      hv = hv + dh.v;
    }
  }
  // one wave = one workgroup: LDS operations of a wave complete in order, the fences only pin the compiler.
  // (A software pipeline -- block b's values read before block b+1's node is evaluated and added up after it -- was built
  // and measured: 4 ms SLOWER on the 100 000-column backward pass, 26 more AGPRs of shuffling for latency that is a small
  // part of a block by now.)
  int b = 0;
#ifndef LGAR_TAN_BLOCKS
#define LGAR_TAN_BLOCKS 2
#endif
  if constexpr (WC != 0 && LGAR_TAN_BLOCKS > 1) {
    // BL blocks at a time: my nodes of blocks b .. b + BL - 1 are INDEPENDENT chains of ~120 dependent double-precision
    // operations each, evaluated side by side -- this kernel's wave is alone on its SIMD and a dependent operation issues only
    // every ~10 cycles, so the other chains run in the gaps of the first.  Heads, node values, the order of every sum:
    // unchanged (the later blocks' values wait in buffers of their own -- the reduction area, unused until the blocks are done).
    constexpr int BL = LGAR_TAN_BLOCKS;
    double *more = xchg + 96 + grp_id * ((BL - 1) * (WC + 1));
    for (; b + BL <= nb; b += BL) {
      double hh[BL], K[BL], c1[BL], c2[BL], c3[BL], c4[BL];
#pragma unroll
      for (int u = 0; u < BL; u++) {
        hh[u] = hm;
#pragma unroll
        for (int j = 0; j < WC; j++) hm = hm + dh.v;
      }
#pragma unroll
      for (int u = 0; u < BL; u++) node(hh[u], K[u], c1[u], c2[u], c3[u], c4[u]);
#pragma unroll
      for (int u = 0; u < BL; u++) {
        const double jj = (double)(WC * (b + u) + r);
        s1 += c1[u]; s1j = fma(c1[u], jj, s1j); s2 += c2[u]; s3 += c3[u]; s4 += c4[u];
      }
      a1 = c1[BL - 1]; a2 = c2[BL - 1]; a3 = c3[BL - 1]; a4 = c4[BL - 1];
      __builtin_amdgcn_fence(__ATOMIC_RELEASE, "workgroup");
      grp[r] = K[0];
#pragma unroll
      for (int u = 1; u < BL; u++) more[(u - 1) * (WC + 1) + r] = K[u];
      __builtin_amdgcn_fence(__ATOMIC_ACQ_REL, "workgroup");
      __builtin_amdgcn_wave_barrier();
      double kq[BL * WC];
#pragma unroll
      for (int j = 0; j < WC; j++) {
        kq[j] = grp[j];
#pragma unroll
        for (int u = 1; u < BL; u++) kq[u * WC + j] = more[(u - 1) * (WC + 1) + j];
      }
#pragma unroll
      for (int j = 0; j < BL * WC; j++) {
        const double pr = k1v + kq[j];
        gv = gv + (pr * hdh.v);
        pairsum += pr;
        k1v = kq[j];
      }
      __builtin_amdgcn_fence(__ATOMIC_ACQ_REL, "workgroup");  // the next blocks' stores stay behind these loads
      __builtin_amdgcn_wave_barrier();
    }
  }
  for (; b < nb; b++) {
    double Kv;
    node(hm, Kv, a1, a2, a3, a4);
    if constexpr (WC != 0) {
#pragma unroll
      for (int j = 0; j < WC; j++) hm = hm + dh.v;
    } else {
      for (int j = 0; j < W; j++) hm = hm + dh.v;
    }
    const double jj = (double)(W * b + r);
    s1 += a1; s1j = fma(a1, jj, s1j); s2 += a2; s3 += a3; s4 += a4;
    __builtin_amdgcn_fence(__ATOMIC_RELEASE, "workgroup");
    grp[r] = Kv;
    __builtin_amdgcn_fence(__ATOMIC_ACQ_REL, "workgroup");
    __builtin_amdgcn_wave_barrier();
    add_up(W);
    __builtin_amdgcn_fence(__ATOMIC_ACQ_REL, "workgroup");  // the next block's store stays behind these loads
    __builtin_amdgcn_wave_barrier();
  }
  // the `rem` < W safe nodes left over after the full blocks: one more block in which only the first `rem` lanes of the group
  // have a node (the others evaluate one for nothing and leave it out of every sum)
  if (rem > 0) {
    double Kv, t1, t2, t3, t4;
    node(hm, Kv, t1, t2, t3, t4);
    for (int j = 0; j < rem; j++) hm = hm + dh.v;
    if (r < rem) {
      a1 = t1; a2 = t2; a3 = t3; a4 = t4;
      const double jj = (double)(W * nb + r);
      s1 += a1; s1j = fma(a1, jj, s1j); s2 += a2; s3 += a3; s4 += a4;
    }
    __builtin_amdgcn_fence(__ATOMIC_RELEASE, "workgroup");
    grp[r] = Kv;
    __builtin_amdgcn_fence(__ATOMIC_ACQ_REL, "workgroup");
    __builtin_amdgcn_wave_barrier();
    add_up(rem);
    __builtin_amdgcn_fence(__ATOMIC_ACQ_REL, "workgroup");
    __builtin_amdgcn_wave_barrier();
  }
  // the five sums over all block nodes, and the last node's coefficients (the group's last lane evaluated it)
  double *red = xchg + 96 + grp_id * (W * LGAR_XCHG_RED + 1);
  __builtin_amdgcn_fence(__ATOMIC_RELEASE, "workgroup");
  red[0 * W + r] = s1; red[1 * W + r] = s1j; red[2 * W + r] = s2; red[3 * W + r] = s3; red[4 * W + r] = s4;
  red[5 * W + r] = a1; red[6 * W + r] = a2; red[7 * W + r] = a3; red[8 * W + r] = a4;
  if (r == 0) red[9 * W] = hm;  // (the group's padding slot) the running sum after the last block node: lane 0's head by now
  __builtin_amdgcn_fence(__ATOMIC_ACQ_REL, "workgroup");
  __builtin_amdgcn_wave_barrier();
  double S[5];
#pragma unroll
  for (int v = 0; v < 5; v++) {
    double t = red[v * W];
#pragma unroll
    for (int j = 1; j < W; j++) t += red[v * W + j];
    S[v] = t;
  }
  const int last = (rem > 0) ? rem - 1 : W - 1;  // the lane that evaluated the last block node
  const double L1 = red[5 * W + last], L2 = red[6 * W + last], L3 = red[7 * W + last], L4 = red[8 * W + last];
  hv = red[9 * W];
  __builtin_amdgcn_fence(__ATOMIC_ACQ_REL, "workgroup");  // a later call's stores stay behind these loads
  __builtin_amdgcn_wave_barrier();
  // sum of dK over the block nodes and dK of the last one, for MY direction
  const double m = (double)(W * nb + rem);
  // d(alpha h_j) = c0 + j c1 with the node's head h_j = h_0 + j dh.  c0 and c1 are formed FIRST: for the alpha direction
  // alpha h does not depend on alpha at all (h = f(Se) / alpha), the two products in each cancel, and they must cancel before
  // anything is summed -- grouped by input instead (d alpha sum A1 h + alpha sum A1 dh_j) the big sums cancel to ~1e-8 of the
  // gradient (measured against the unshared launch; now ~1e-12).
  const double c0 = l.alpha.d * h2.v + l.alpha.v * h2.d;
  const double c1 = l.alpha.d * dh.v + l.alpha.v * dh.d;
  const double sumKd = c0 * S[0] + c1 * S[1] + nm1.d * S[2] + half_m.d * S[3] + l.ksat.d * S[4];
  const double Kd_last = L1 * (c0 + (m - 1.0) * c1) + L2 * nm1.d + L3 * half_m.d + L4 * l.ksat.d;
  // g += sum over the block terms of (k_prev + k) hdh:  k_in + 2 sum K - K_last
  g = Dual<double>(gv, g.d + hdh.v * (k1.d + 2.0 * sumKd - Kd_last) + hdh.d * pairsum);
  k1 = Dual<double>(k1v, Kd_last);
  h2 = Dual<double>(hv, h2.d + m * dh.d);
}
__device__ __forceinline__ void geff_shared_blocks(const LayerK<Dual<double>> &l, const Dual<double> &nm1, const Dual<double> &half_m,
                                                   Dual<double> &h2, const Dual<double> &dh, const Dual<double> &hdh, Dual<double> &g,
                                                   Dual<double> &k1, int nb, int W, double *xchg, int rem) {
  // 9 = the 3 x L parameters of a three-layer column (BASELINE configs[4]); other widths run the generic loops
  if (W == 9) geff_shared_blocks_w<9>(l, nm1, half_m, h2, dh, hdh, g, k1, nb, W, xchg, rem);
  else geff_shared_blocks_w<0>(l, nm1, half_m, h2, dh, hdh, g, k1, nb, W, xchg, rem);
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
