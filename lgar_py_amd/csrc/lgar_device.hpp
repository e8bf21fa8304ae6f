// lgar_device.hpp -- device-side LGAR column physics for gfx950 (CDNA4), lane-per-column.
//
// One wavefront = 64 independent soil columns.  Per-front state of the wave lives in LDS as
// [field][front][lane] (lane-contiguous => bank-conflict-free for any per-lane front index), the
// per-layer van Genuchten parameters and the column scalars live in registers, and the time loop
// runs inside the kernel.  The scalar type S is float, double, or a forward-mode dual number
// (lgar_dual.hpp) for the differentiable path.
//
// What each routine computes follows the reference's Python (paths relative to
// /root/reference/dpLGAR/); the data layout and control structure do not: the reference keeps a
// linked list of Layer objects each owning a Python list of WettingFront objects, here a column is
// ONE flat top->bottom front array tagged with layer numbers.
#pragma once
#ifndef LGAR_DEVSIM
#include <hip/hip_runtime.h>
#endif
#include <stdint.h>

#include "../../include/lgar.h"
#include "lgar_math.hpp"

// Kernel arguments are read where they are used, as scalar loads from the kernarg segment (constant address space),
// instead of being held in SGPRs for the whole kernel: the argument block (35 pointers + the run-time constants) is
// larger than the SGPR file, and everything the register allocator cannot keep becomes v_writelane / v_readlane traffic
// on the vector ALU -- the unit this kernel is bound by.
#ifndef LGAR_DEVSIM
#define LGAR_KARG __attribute__((address_space(4)))
#else
#define LGAR_KARG
#endif

namespace lgar {

constexpr int WAVE = 64;

// ---------------------------------------------------------------------------------------------
// scalar-type plumbing
// ---------------------------------------------------------------------------------------------
template <typename S> struct Real { using type = S; };
template <typename S> using real_t = typename Real<S>::type;

__device__ __forceinline__ double val(double x) { return x; }
__device__ __forceinline__ float val(float x) { return x; }
__device__ __forceinline__ double pw(double x, double y) { return fast_pow<false>(x, y); }  // lgar_math.hpp, ~1e-14 relative
// fp32: v_log_f32 / v_exp_f32 (quarter-rate transcendentals), ~2-3 ulp for the exponents used here
// log2 / exp2 (fused Geff node, dual-number pow)
#ifndef LGAR_DEVSIM
__device__ __forceinline__ float lg2(float x) { return __builtin_amdgcn_logf(x); }
__device__ __forceinline__ float ex2(float x) { return __builtin_amdgcn_exp2f(x); }
__device__ __forceinline__ float sq(float x) { return __builtin_amdgcn_sqrtf(x); }
__device__ __forceinline__ float clamp01(float x) { return __builtin_amdgcn_fmed3f(x, 0.0f, 1.0f); }  // v_med3_f32
__device__ __forceinline__ unsigned long long any_lane(bool p) { return __ballot(p); }
__device__ __forceinline__ bool first_active_lane() {
  const unsigned long long m = __ballot(1);
  return (int)(threadIdx.x & 63u) == __ffsll((long long)m) - 1;
}
// between the stores and the loads of an exchange through the wave's LDS (cooperating lanes): one wave = one workgroup and a
// wave's LDS operations complete in order, so this only pins the compiler
__device__ __forceinline__ void lds_exchange_point() {
  __builtin_amdgcn_fence(__ATOMIC_ACQ_REL, "workgroup");
  __builtin_amdgcn_wave_barrier();
}
// a value loaded from memory a step ahead is taken into its register HERE (the wait for the load is placed at this point):
// loads and stores share one counter, and a wait placed after later stores would wait for those as well
template <typename T> __device__ __forceinline__ void settle_load(T &x) { asm volatile("" : "+v"(x)); }
#else  // tests/devsim: the same device code compiled for the host, one lane at a time (test infrastructure only)
__device__ __forceinline__ float lg2(float x) { return log2f(x); }
__device__ __forceinline__ float ex2(float x) { return exp2f(x); }
__device__ __forceinline__ float sq(float x) { return sqrtf(x); }
__device__ __forceinline__ float clamp01(float x) { return fminf(fmaxf(x, 0.0f), 1.0f); }
__device__ __forceinline__ unsigned long long any_lane(bool p) { return p ? 1ull : 0ull; }
__device__ __forceinline__ bool first_active_lane() { return true; }
__device__ __forceinline__ void lds_exchange_point() {}
template <typename T> __device__ __forceinline__ void settle_load(T &) {}
#endif
__device__ __forceinline__ float pw(float x, float y) { return ex2(y * lg2(x)); }
// EX = true (verification mode, double precision only): the correctly rounded library pow, as the reference's torch.pow;
// otherwise the lean pow above
template <bool EX> __device__ __forceinline__ double pwx(double x, double y) {
  if constexpr (EX) return pow(x, y);
  return pw(x, y);
}
template <bool EX> __device__ __forceinline__ float pwx(float x, float y) { return pw(x, y); }
// pow by arithmetic policy POL (see dv): 1 = the library's, 3 = the lean pow with pairwise-combined polynomials (lgar_math.hpp,
// ESTRIN), anything else the lean pow
template <int POL> __device__ __forceinline__ double pwp(double x, double y) {
  if constexpr (POL == 1) return pow(x, y);
  if constexpr (POL == 3) return fast_pow<true>(x, y);
  return fast_pow<false>(x, y);
}
template <int POL> __device__ __forceinline__ float pwp(float x, float y) { return pw(x, y); }
// ... for any scalar type of the column physics: plain reals by pwp, dual numbers by their own pwx (lgar_dual.hpp)
template <typename S, int POL> __device__ __forceinline__ S pwq(const S &x, const S &y) {
  if constexpr (sizeof(S) == sizeof(real_t<S>)) return pwp<POL>(x, y);
  else return pwx<POL == 1>(x, y);
}

// Arithmetic policy of the leaf functions and the column physics (template parameter POL):
//   0  lean pow (lgar_math.hpp), IEEE division                        -- fp64 fast mode, dual numbers
//   1  library pow, IEEE division                                     -- verification mode in double precision
//   2  v_log/v_exp pow, division as a * v_rcp_f32(b) (<= 1.5 ulp)     -- fp32 fast mode: the correctly rounded fp32 divide
//      is ~10 instructions, the path takes ~20 of them per column-step.  Se = (theta - theta_r)/(theta_e - theta_r) keeps
//      the IEEE divide in every policy: Se must be exactly 1 at saturation (x/x == 1), or the pow bases go negative.
#ifndef LGAR_DEVSIM
__device__ __forceinline__ float rcp32(float b) { return __builtin_amdgcn_rcpf(b); }
#else
__device__ __forceinline__ float rcp32(float b) { return 1.0f / b; }
#endif
template <int POL> __device__ __forceinline__ float dv(float a, float b) {
  if constexpr (POL == 2) return a * rcp32(b);
  return a / b;
}
// POL 0 in double precision (fast modes): a / b as a * (1 / b) from v_rcp_f64, one Newton step on the reciprocal and one
// correction of the quotient -- six instructions, within an ulp of the IEEE quotient, against the ~14 of the correctly rounded
// divide (v_div_scale / v_div_fmas / v_div_fixup); a column step takes ~25 of them, each on its wave's critical path.  The
// divisors of the column physics are as a rule finite and non-zero (depths, theta differences that were tested > 0,
// 1 + (alpha psi)^n); where one is not -- b = 0, inf or denormal, a = inf: the Newton steps meet inf * 0 and come out NaN
// where the quotient is inf, 0 or finite (an overflowed (alpha h)^n in theta_from_h must give theta_r, not NaN) -- the NaN
// result sends the lanes concerned through the IEEE divide (a compare and a branch that is practically never taken).
// Se = (theta - theta_r) / (theta_e - theta_r) keeps the IEEE divide (se_from_theta: x / x must be exactly 1).
#ifndef LGAR_DEVSIM
__device__ __forceinline__ double lean_div(double a, double b) {
  double r = __builtin_amdgcn_rcp(b);
  r = __builtin_fma(__builtin_fma(-b, r, 1.0), r, r);
  const double q = a * r;
  double res = __builtin_fma(__builtin_fma(-b, q, a), r, q);
  if (__builtin_expect(res != res, 0)) res = a / b;
  return res;
}
#else
__device__ __forceinline__ double lean_div(double a, double b) { return a / b; }
#endif
template <int POL> __device__ __forceinline__ double dv(double a, double b) {
  if constexpr (POL == 0 || POL == 3) return lean_div(a, b);
  return a / b;
}
__device__ __forceinline__ double lg2(double x) { return fast_log2<false>(x); }
__device__ __forceinline__ double ex2(double x) { return fast_exp2<false>(x); }
// (the mixed-precision trapezoid's own double-precision logarithms and exponentials: see fast_pow on ESTRIN)
__device__ __forceinline__ double lg2e(double x) { return fast_log2<true>(x); }
__device__ __forceinline__ double ex2e(double x) { return fast_exp2<true>(x); }
// log2 / exp2 of arguments known to be positive / not NaN (the interior of the Geff trapezoid): no special-case selects
__device__ __forceinline__ float lg2p(float x) { return lg2(x); }
__device__ __forceinline__ float ex2p(float x) { return ex2(x); }
__device__ __forceinline__ double lg2p(double x) { return fast_log2_core<false>(x); }
__device__ __forceinline__ double ex2p(double x) { return fast_exp2_core<false, false>(x); }
__device__ __forceinline__ double sq(double x) { return sqrt(x); }
__device__ __forceinline__ double ab(double x) { return fabs(x); }
__device__ __forceinline__ float ab(float x) { return fabsf(x); }
__device__ __forceinline__ double mn(double a, double b) { return fmin(a, b); }
__device__ __forceinline__ float mn(float a, float b) { return fminf(a, b); }
__device__ __forceinline__ bool is_nan(double x) { return x != x; }
__device__ __forceinline__ bool is_nan(float x) { return x != x; }
__device__ __forceinline__ bool same_bits(double a, double b) { return a == b; }  // (NaN never matches: recomputed)
__device__ __forceinline__ bool same_bits(float a, float b) { return a == b; }

// Measurement points (cost attribution, tools/ablate.py): LGAR_MEASURE_POINT(NAME, args) marks a place where a measurement
// build can run a routine twice or count something, LGAR_ABLATABLE(NAME, statement) a statement such a build can leave out.
// In the product both are transparent: the point is empty, the statement is just the statement.  Only
// lgar_py_amd.build.build_variant passes -DLGAR_MEASURE, which takes the definitions from lgar_measure.hpp instead.
#ifdef LGAR_MEASURE
#include "lgar_measure.hpp"
#else
#define LGAR_MEASURE_POINT(NAME, ...)
#define LGAR_ABLATABLE(NAME, ...) __VA_ARGS__
// one lane per wave-level Geff evaluation adds 1 above the fault bits of its status word (no register, no LDS word; summed
// over the wave at the end of the block): bits 8..31 are otherwise unused while a column is integrated
#define LGAR_COUNT_GEFF_CALL(site) \
  if (count_geff && first_active_lane()) status += (1 << LGAR_ST_STEP_SHIFT);
#endif

// tolerances: the reference's absolute 1e-12 (layers/Layer.py:60) is unreachable in fp32
template <typename R> struct Tol;
template <> struct Tol<double> {
  static constexpr double mass = 1e-12;      // Layer.tolerance
  static constexpr double nochange = 1e-15;  // Layer.py:307,309
  static constexpr double tiny = 1e-50;      // Layer.py:315
};
template <> struct Tol<float> {
  static constexpr float mass = 2e-5f;
  static constexpr float nochange = 1e-9f;
  static constexpr float tiny = 1e-30f;
};

template <typename S> struct LayerK {
  S alpha, n, m, inv_m, inv_n, ksat, te, tr;
};

template <typename S, int NL> struct ColParams {
  S alpha[NL], n[NL], m[NL], inv_m[NL], inv_n[NL], ksat[NL], te[NL], tr[NL], thick[NL], cum[NL];
};

// c ? a : b on VALUES (a dual number selects component by component: a ternary on two structs can become a select of
// addresses + a load, which sends the operands through scratch memory)
__device__ __forceinline__ float choose(bool c, float a, float b) { return c ? a : b; }
__device__ __forceinline__ double choose(bool c, double a, double b) { return c ? a : b; }

template <typename S, int NL> __device__ __forceinline__ S sel(const S (&a)[NL], int k) {
  // load every element unconditionally, then select on VALUES: a lazily evaluated a[j] becomes a
  // select of addresses + one load, which pins the whole parameter block in scratch memory
  S v[NL];
#pragma unroll
  for (int j = 0; j < NL; j++) v[j] = a[j];
  S r = v[0];
#pragma unroll
  for (int j = 1; j < NL; j++) r = choose(k == j, v[j], r);
  return r;
}

template <typename S, int NL> __device__ __forceinline__ LayerK<S> pick(const ColParams<S, NL> &P, int k) {
  LayerK<S> l;
  l.alpha = sel<S, NL>(P.alpha, k);
  l.n = sel<S, NL>(P.n, k);
  l.m = sel<S, NL>(P.m, k);
  l.ksat = sel<S, NL>(P.ksat, k);
  l.inv_m = sel<S, NL>(P.inv_m, k);
  l.inv_n = sel<S, NL>(P.inv_n, k);
  l.te = sel<S, NL>(P.te, k);
  l.tr = sel<S, NL>(P.tr, k);
  return l;
}
template <typename S, int NL> __device__ __forceinline__ LayerK<S> pick_static(const ColParams<S, NL> &P, int j) {
  LayerK<S> l;
  l.alpha = P.alpha[j]; l.n = P.n[j]; l.m = P.m[j]; l.inv_m = P.inv_m[j];
  l.inv_n = P.inv_n[j]; l.ksat = P.ksat[j]; l.te = P.te[j]; l.tr = P.tr[j];
  return l;
}

// ---------------------------------------------------------------------------------------------
// van Genuchten leaf functions (models/physics/utils.py)
// ---------------------------------------------------------------------------------------------
// calc_theta_from_h, utils.py:35-51
template <typename S, int POL = 0> __device__ __forceinline__ S theta_from_h(const LayerK<S> &l, S h) {
  using R = real_t<S>;
  S ap = pwq<S, POL>(l.alpha * h, l.n);
  S op = pwq<S, POL>(R(1.0) + ap, l.m);
  return (dv<POL>(S(R(1.0)), op) * (l.te - l.tr)) + l.tr;
}
// ... also handing out (alpha h)^n, the quantity that says how close to saturation the head is
template <typename S, int POL = 0> __device__ __forceinline__ S theta_from_h_ap(const LayerK<S> &l, S h, S &ap) {
  using R = real_t<S>;
  ap = pwq<S, POL>(l.alpha * h, l.n);
  S op = pwq<S, POL>(R(1.0) + ap, l.m);
  return (dv<POL>(S(R(1.0)), op) * (l.te - l.tr)) + l.tr;
}
// calc_se_from_theta, utils.py:102-112
template <typename S> __device__ __forceinline__ S se_from_theta(const LayerK<S> &l, S theta) {
  return (theta - l.tr) / (l.te - l.tr);
}

// calc_se_from_h, utils.py:115-131 (exactly 1 for |h| < 0.1)
template <typename S, int POL = 0> __device__ __forceinline__ S se_from_h(const LayerK<S> &l, S h) {
  using R = real_t<S>;
  if (ab(val(h)) < R(1.0e-01)) return S(R(1.0));
  S is = pwq<S, POL>(l.alpha * h, l.n);
  return dv<POL>(S(R(1.0)), pwq<S, POL>(R(1.0) + is, l.m));
}
// calc_k_from_se, utils.py:134-156; torch.isclose(base, 0, rtol=1e-12) => |base| <= 1e-8 (default atol)
template <typename S, int POL = 0> __device__ __forceinline__ S k_from_se(const LayerK<S> &l, S se) {
  using R = real_t<S>;
  S sp = pwq<S, POL>(se, l.inv_m);
  S base = R(1.0) - sp;
  if (ab(val(base)) <= R(1e-8)) base = base + R(1e-12);
  S op = pwq<S, POL>(base, l.m);
  S t = R(1.0) - op;
  return l.ksat * sq(se) * (t * t);
}
// ... at Se == 1 (the trapezoid's K under the |h| < 0.1 rule): Se^(1/m) is exactly 1, the base exactly 0 and nudged to 1e-12 --
// calc_k_from_se's value and tangent, bit for bit, without the first pow
template <typename S, int POL = 0> __device__ __forceinline__ S k_from_se_one(const LayerK<S> &l) {
  using R = real_t<S>;
  S op = pwq<S, POL>(S(R(1e-12)), l.m);
  S t = R(1.0) - op;
  return l.ksat * sq(S(R(1.0))) * (t * t);
}
// calc_h_from_se, utils.py:159-174
template <typename S, int POL = 0> __device__ __forceinline__ S h_from_se(const LayerK<S> &l, S se) {
  using R = real_t<S>;
  S sp = pwq<S, POL>(se, -l.inv_m);
  S base = sp - R(1.0);
  if (ab(val(base)) <= R(1e-8)) base = base + R(1e-12);
  S op = pwq<S, POL>(base, l.inv_n);
  return dv<POL>(S(R(1.0)), l.alpha) * op;
}
// calc_geff, models/physics/lgar/green_ampt.py:45-84: nint-interval trapezoid of K(h) dh / Ksat.
// The discretisation error is part of the answer: same nodes (h accumulated by repeated += dh).
template <typename S, int POL = 0> __device__ __forceinline__ S geff_literal(const LayerK<S> &l, S theta1, S theta2, int nint) {
  using R = real_t<S>;
  S se_i = se_from_theta(l, theta1);
  S se_f = se_from_theta(l, theta2);
  S h_i = h_from_se<S, POL>(l, se_i);
  S h_f = h_from_se<S, POL>(l, se_f);
  S dh = (h_f - h_i) / R(nint);
  S g = S(R(0.0));
  S k1 = k_from_se<S, POL>(l, se_i);
  S h2 = h_i + dh;
  S hdh = dh / R(2.0);
  for (int i = 0; i < nint; i++) {
    // rounding in the repeated h2 += dh can carry the last nodes past 0 (by ~1e-10 in fp64): a negative head is
    // saturation (Se = 1, the |h| < 0.1 rule), never pow of a negative base
    S se2 = (val(h2) < R(0.0)) ? S(R(1.0)) : se_from_h<S, POL>(l, h2);
    S k2 = k_from_se<S, POL>(l, se2);
    g = g + ((k1 + k2) * hdh);
    k1 = k2;
    h2 = h2 + dh;
  }
  return ab(g / l.ksat);
}
// the trapezoid the kernels use by default: specialised below for float / double (and the dual numbers, lgar_dual.hpp)
template <typename S> __device__ __forceinline__ S geff(const LayerK<S> &l, S theta1, S theta2, int nint) {
  return geff_literal<S, 0>(l, theta1, theta2, nint);
}

// Fused Geff (fp64, and the dual numbers of the differentiable path): the same 121 nodes with Se(h) -> K(Se) fused per
// node.  With x = alpha h, a = x^n and n m = n - 1:
//   P = x^(n-1) = a^m,  a = x P,  sqrt(Se) = (1+a)^(-m/2),  (a/(1+a))^m = P Se
// so K = Ksat sqrt(Se) (1 - P Se)^2 needs 2 log2 + 2 exp2 and no division, instead of the reference's 4 pow + sqrt +
// divide.  (The 1e-12 nudge of calc_k_from_se applies only for a <= 1e-8, i.e. |h| far below the 0.1 cm cut where Se is 1
// anyway.)  fp64 keeps the reference's running sum h2 += dh; a float instantiation would place the nodes directly
// (h_i + (i+1) dh, last node = h_f): its running sum drifts by ~nint ulps of h_i, cm-scale for very dry soil, and the last
// trapezoid dominates the integral.  The plain-float kernels use the packed loop further down instead.
// K(h) of one trapezoid node, fused (see above); nm1 = n - 1, half_m = -m/2.  lgar_dual.hpp overloads it for dual numbers
// (same value operations, hand-derived tangent).
template <typename S> __device__ __forceinline__ S geff_node(const LayerK<S> &l, const S &nm1, const S &half_m, const S &h) {
  using R = real_t<S>;
  const S x = l.alpha * h;
  const S P = ex2p(nm1 * lg2p(x));
  const S l1 = lg2p(R(1.0) + x * P);
  const S sqrt_se = ex2p(half_m * l1);
  const S t = R(1.0) - P * (sqrt_se * sqrt_se);
  return l.ksat * sqrt_se * (t * t);
}
// `nb` blocks of W consecutive safe nodes of the trapezoid (h2, g, k1 advanced as W nb passes of the plain loop would).
// lgar_dual.hpp overloads it for dual numbers whose W neighbouring lanes carry the SAME column with different parameter
// directions: each lane evaluates one node of a block and the W exchange the values.
template <typename S>
__device__ __forceinline__ void geff_shared_blocks(const LayerK<S> &l, const S &nm1, const S &half_m, S &h2, const S &dh, const S &hdh, S &g,
                                                   S &k1, int nb, int W, real_t<S> *xchg, int rem) {
  (void)xchg;
  for (int j = 0; j < W * nb + rem; j++) {
    const S k2 = geff_node(l, nm1, half_m, h2);
    g = g + ((k1 + k2) * hdh);
    k1 = k2;
    h2 = h2 + dh;
  }
}
// (sizes of the cooperating lanes' exchange table: plain constants, the simulator's build sees them too)
#define LGAR_COOP_TAB 128      /* most trapezoid intervals a cooperating job may have (LgarDims.nint; 120 in every bundled config) */
#define LGAR_COOP_TAB_ROW 130  /* doubles per group's table in LDS: padded so that the groups' tables start on different banks */
#define LGAR_COOP_PAIR_LANES 12 /* groups of at least this many lanes take two moving fronts at a time (Column::calc_dzdt_pairs) */
#ifndef LGAR_DEVSIM
// The trapezoid's interior for COOPERATING lanes (forward kernels on jobs too small to fill the chip, LgarDims.forward_lanes):
// the `lanes` lanes of an aligned group all carry the SAME column -- same values, same branches.  `tab` is the group's table
// of LGAR_COOP_TAB doubles in LDS.
//   1. the group's first lane runs the plain loop's running sum h2 += dh and leaves head j in tab[j];
//   2. lane r evaluates nodes r, r + lanes, r + 2 lanes, ... -- two or four at a time -- and puts K(head) where the head was
//      (no other lane reads that head);
//   3. lane r turns its nodes into the trapezoid's terms (K_{j-1} + K_j) dh/2, in place;
//   4. every lane adds the 120 terms up in order.
// Heads, node values, terms and the sum are those of the plain loop bit for bit: each goes through the same operations on the
// same operands, only once per group instead of once per lane.
// (r: my place in the group.  The last group of a wavefront also takes the lanes left over when `lanes` does not divide 64:
// their r >= lanes; they evaluate no node -- a node's table slot must be read as a head and rewritten by ONE lane -- and take
// part in everything else.)
// W nodes per lane and round (independent chains of ~100 dependent double-precision operations each: one wave alone on its SIMD
// issues a dependent operation every ~10 cycles, so the chains of a round overlap)
template <int W>
__device__ __forceinline__ void geff_coop_node_rounds(const LayerK<double> &l, double nm1, double half_m, double k_sat1, double k_first,
                                                      double hdh, int nint, int lanes, double *tab, int r) {
  const int stride = W * lanes;
  for (int first = 0; first < nint; first += stride) {
    double h[W], k[W];
    bool ok[W];
#pragma unroll
    for (int u = 0; u < W; u++) {
      const int j = first + r + u * lanes;
      ok[u] = (r < lanes) && (j < nint);
      h[u] = tab[ok[u] ? j : 0];
    }
#pragma unroll
    for (int u = 0; u < W; u++) k[u] = geff_node(l, nm1, half_m, h[u]);
#pragma unroll
    for (int u = 0; u < W; u++) {
      k[u] = (fabs(h[u]) < 0.1 || h[u] < 0.0) ? k_sat1 : k[u];  // utils.py:124-128 (never true on the nodes the plain loop leaves unchecked)
      if (ok[u]) tab[first + r + u * lanes] = k[u];
    }
  }
  lds_exchange_point();
  LGAR_MEASURE_POINT(CLK, 14)
  // the trapezoid's terms (K_{j-1} + K_j) dh/2, in place: the rounds run from the LAST to the first, so that a round only
  // overwrites nodes no later round reads (its own, whose predecessors belong to it or to a round still to come)
  for (int first = ((nint - 1) / stride) * stride; first >= 0; first -= stride) {
    double p[W], k[W];
    bool ok[W];
#pragma unroll
    for (int u = 0; u < W; u++) {
      const int j = first + r + u * lanes;
      ok[u] = (r < lanes) && (j < nint);
      p[u] = tab[(ok[u] && j > 0) ? j - 1 : 0];
      k[u] = tab[ok[u] ? j : 0];
    }
    lds_exchange_point();  // every lane has its operands before any term replaces a node
#pragma unroll
    for (int u = 0; u < W; u++) {
      const int j = first + r + u * lanes;
      if (ok[u]) tab[j] = (((j > 0) ? p[u] : k_first) + k[u]) * hdh;
    }
  }
}
__device__ __forceinline__ void geff_nodes_cooperative(const LayerK<double> &l, double nm1, double half_m, double k_sat1, double &h2,
                                                       double dh, double hdh, double &g, double &k1, int nint, int lanes, double *tab, int r) {
  // one wave = one workgroup: LDS operations of a wave complete in order, the fences only pin the compiler
  __builtin_amdgcn_fence(__ATOMIC_ACQ_REL, "workgroup");  // (earlier loads of the table stay in front of these stores)
  // 1. the heads: a chain of nint dependent additions; ONE lane of the group runs it (64 lanes storing to one address are
  //    64 conflicting writes: 24 cycles per head, measured)
  if (r == 0) {
    double h = h2;
    int j = 0;
    for (; j + 16 <= nint; j += 16) {
      // (sixteen heads in registers of their own, then their stores: an addition that overwrites a register a store still
      // reads waits for that store)
      double hh[16];
#pragma unroll
      for (int u = 0; u < 16; u++) {
        hh[u] = h;
        h = h + dh;
      }
#pragma unroll
      for (int u = 0; u < 16; u++) tab[j + u] = hh[u];
    }
    for (; j < nint; j++) {
      tab[j] = h;
      h = h + dh;
    }
  }
  lds_exchange_point();
  LGAR_MEASURE_POINT(CLK, 13)
  // 2. nodes, 3. terms: two per lane and round, four when two would take more than one round
  if (nint > 2 * lanes) geff_coop_node_rounds<4>(l, nm1, half_m, k_sat1, k1, hdh, nint, lanes, tab, r);
  else geff_coop_node_rounds<2>(l, nm1, half_m, k_sat1, k1, hdh, nint, lanes, tab, r);
  lds_exchange_point();
  LGAR_MEASURE_POINT(CLK, 15)
  // 4. every lane adds the terms up in order (the same address in every lane of the group: a broadcast).  The additions are
  //    one dependent chain; the reads of the next sixteen terms are in flight while the chain works through the present ones.
  {
    const int nb = nint >> 4;  // batches of sixteen terms, alternately in two sets of registers
    const double *tail = tab + (nb << 4);
    double ta[16], tb[16], tc[8];
    if (nb > 0) {
#pragma unroll
      for (int j = 0; j < 16; j++) ta[j] = tab[j];
    }
    if (nint & 8) {
#pragma unroll
      for (int j = 0; j < 8; j++) tc[j] = tail[j];
    }
#pragma unroll
    for (int i = 0; i < LGAR_COOP_TAB / 16; i++) {
      if (i < nb) {
        const double *next = tab + 16 * (i + 1);
        // (the next batch is read whether it exists or not -- unconditional loads keep the waits exact; the table has
        // LGAR_COOP_TAB_ROW >= 16 (i + 2) entries for every i that gets here with another batch to come)
        if (i & 1) {
          if (i + 1 < LGAR_COOP_TAB / 16) {
#pragma unroll
            for (int j = 0; j < 16; j++) ta[j] = next[j];
          }
#pragma unroll
          for (int j = 0; j < 16; j++) g = g + tb[j];
        } else {
          if (i + 1 < LGAR_COOP_TAB / 16) {
#pragma unroll
            for (int j = 0; j < 16; j++) tb[j] = next[j];
          }
#pragma unroll
          for (int j = 0; j < 16; j++) g = g + ta[j];
        }
      }
    }
    int q0 = nb << 4;
    if (nint & 8) {
#pragma unroll
      for (int j = 0; j < 8; j++) g = g + tc[j];
      q0 += 8;
    }
    for (; q0 < nint; q0++) g = g + tab[q0];
  }
  lds_exchange_point();  // the next call's stores stay behind these loads
  LGAR_MEASURE_POINT(CLK, 16)
}
#endif
// (cooperating lanes, calc_dzdt: conductivities that ride along with the evaluations that open a trapezoid -- see
// geff_ends_cooperative)
struct CoopRiders {
  int n = 0;           // riders of this call (the group needs 4 + n lanes)
  LayerK<double> l;    // MY rider's layer and Se (lanes 4 .. 4 + n - 1; anything elsewhere)
  double se = 1.0;
  double k[LGAR_LMAX]; // out: K of rider e
};
#ifndef LGAR_DEVSIM
// The four two-pow evaluations that open a trapezoid -- h(Se_i), h(Se_f) (calc_h_from_se), K(Se_i), K(1) (calc_k_from_se) -- for
// COOPERATING lanes: lane r of a group evaluates number r mod 4 and the group exchanges the results.  Both functions are
// "pow, offset from 1, nudge, pow, finish": the lanes run ONE instruction stream with their own exponents and pick their own
// finish, every value going through exactly the operations h_from_se / k_from_se apply to it (bit-identical results; the
// serial chain of eight pows becomes one of two).
// Riders (calc_dzdt): up to LGAR_LMAX more conductivities K(Se) of OTHER evaluations -- the moving front's own K(theta) and the
// K of every layer above at the front's psi (calc_bottom_sum, Layer.py:1557-1582) -- are the same "pow, offset, nudge, pow,
// finish": lane 4 + e of the group takes rider e, with that rider's layer (xl) and Se (xse) as ITS operands, and every lane
// reads the n_riders results back into xk before the table is reused for the trapezoid's heads.
__device__ __forceinline__ void geff_ends_cooperative(const LayerK<double> &l, double se_i, double se_f, double &h_i, double &h_f,
                                                      double &k_i, double &k_sat1, double *xchg, int r, CoopRiders *rd = nullptr) {
  const bool rider = (rd != nullptr) && (r >= 4) && (r - 4 < rd->n);
  const int which = rider ? 2 : (r & 3);  // 0: h(Se_i), 1: h(Se_f), 2: K(Se_i), 3: K(1)
  const bool is_h = which < 2;
  double se = (which == 1) ? se_f : ((which == 3) ? 1.0 : se_i);
  double inv_m = l.inv_m, m = l.m, ksat = l.ksat;
  if (rd != nullptr) {
    se = rider ? rd->se : se;
    inv_m = rider ? rd->l.inv_m : inv_m; m = rider ? rd->l.m : m; ksat = rider ? rd->l.ksat : ksat;
  }
  const double sp = pw(se, is_h ? -inv_m : inv_m);
  double base = is_h ? sp - 1.0 : 1.0 - sp;
  if (fabs(base) <= 1e-8) base = base + 1e-12;
  const double op = pw(base, is_h ? l.inv_n : m);
  const double t = 1.0 - op;
  const double mine = is_h ? (1.0 / l.alpha) * op : ksat * sqrt(se) * (t * t);
  double *grp = xchg;  // the group's table
  __builtin_amdgcn_fence(__ATOMIC_ACQ_REL, "workgroup");
  if (r < 4) grp[which] = mine;  // lanes 0..3 of the group (a group has at least 4)
  if (rider) grp[r] = mine;
  __builtin_amdgcn_fence(__ATOMIC_ACQ_REL, "workgroup");
  __builtin_amdgcn_wave_barrier();
  h_i = grp[0]; h_f = grp[1]; k_i = grp[2]; k_sat1 = grp[3];
  if (rd != nullptr) {
#pragma unroll
    for (int e = 0; e < LGAR_LMAX; e++) rd->k[e] = grp[4 + e];  // (slots past the last rider: stale values nobody uses)
  }
  __builtin_amdgcn_fence(__ATOMIC_ACQ_REL, "workgroup");
  __builtin_amdgcn_wave_barrier();
}
#endif
template <typename S>
__device__ __forceinline__ S geff_fused(const LayerK<S> &l, S theta1, S theta2, int nint, real_t<S> *xchg = nullptr, int coop = 0,
                                        int rank = 0, CoopRiders *riders = nullptr) {
  using R = real_t<S>;
  const S se_i = se_from_theta(l, theta1);
  const S se_f = se_from_theta(l, theta2);
  S h_i, h_f, k_sat1, k1;  // h(Se) of both ends; K at Se == 1 (|h| < 0.1); K(Se_i)
  bool ends_done = false;
#ifndef LGAR_DEVSIM
  if constexpr (sizeof(S) == 8 && sizeof(R) == 8) {
    if (coop >= 4 && xchg != nullptr) {
      LGAR_MEASURE_POINT(CLK, 11)
      geff_ends_cooperative(l, se_i, se_f, h_i, h_f, k1, k_sat1, xchg, rank, riders);
      LGAR_MEASURE_POINT(CLK, 12)
      ends_done = true;
    }
  }
#endif
  if (!ends_done) {
    LGAR_MEASURE_POINT(CLK, 25)
    h_i = h_from_se(l, se_i);
    h_f = h_from_se(l, se_f);
    k_sat1 = k_from_se_one(l);
    k1 = k_from_se(l, se_i);
    LGAR_MEASURE_POINT(CLK, 26)
  }
  // lg2p / ex2p take finite positive arguments (the exponent of 2^x goes through a float -> int conversion, undefined for
  // NaN): an end point outside the domain (Se > 1 -> negative pow base -> NaN head; the reference raises ValueError there,
  // physics/utils.py:25-27) is replaced by a harmless one for the nodes and put back into the result below
  const S h_i_own = h_i, h_f_own = h_f;
  const bool outside = is_nan(val(h_i)) || is_nan(val(h_f));
  if (outside) { h_i = S(R(1.0)); h_f = S(R(1.0)); }
  const S dh = (h_f - h_i) / R(nint);
  const S hdh = dh / R(2.0);
  const S half_m = R(-0.5) * l.m;
  const S nm1 = l.n - R(1.0);  // n m = n - 1: a^m = (alpha h)^(n-1)
  S g = S(R(0.0));
  S h2 = h_i + dh;
  // four transcendentals per node: P = a^m = x^(n-1), a = x P, sqrt(Se) = (1+a)^(-m/2), (a/(1+a))^m = P Se
  auto node = [&](const S &h) { return geff_node(l, nm1, half_m, h); };
  int i = 0;
#ifndef LGAR_DEVSIM
  bool nodes_done = false;
  if constexpr (sizeof(S) == 8 && sizeof(R) == 8) {
    if (coop > 1 && xchg != nullptr) {  // cooperating lanes: every node is checked against the |h| < 0.1 rule by its evaluator
      geff_nodes_cooperative(l, nm1, half_m, k_sat1, h2, dh, hdh, g, k1, nint, coop, xchg, rank);
      nodes_done = true;
    }
  }
  if (nodes_done) i = nint;
#endif
  // The |h| < 0.1 -> Se = 1 rule (utils.py:124-128) can only bind on a SUFFIX of the nodes (h falls monotonically from
  // h_i to h_f): a wave-uniform count of leading nodes that no lane needs to test runs select-free.
  int n_safe = nint;
  if (i < nint) {
    // (node j sits at h_i + j dh: nodes up to floor((h_i - 0.1) / -dh) - 1 lie a whole interval above the cut, far more than the
    // running sum's rounding can move them)
    const R jf = (val(h_i) - R(0.1)) / -val(dh) - R(1.0);
    const int safe = (jf > R(0.0)) ? ((jf < R(nint)) ? int(jf) : nint) : 0;  // NaN (dh == 0) -> 0
    if (any_lane(safe < nint) != 0ull) {
      n_safe = 0;
      for (int bit = 128; bit; bit >>= 1) {
        const int cand = n_safe + bit;
        if (cand <= nint && any_lane(safe < cand) == 0ull) n_safe = cand;
      }
    }
  }
  if constexpr (sizeof(S) != sizeof(R) && sizeof(R) == 8) {
    if (xchg != nullptr && coop >= 2 && n_safe >= coop) {  // coop: the W lanes that share this column (LgarDims.tangent_share)
      LGAR_MEASURE_POINT(CLK, 27)
      geff_shared_blocks(l, nm1, half_m, h2, dh, hdh, g, k1, n_safe / coop, coop, xchg, n_safe % coop);
      i = n_safe;  // (the safe nodes left over after the full blocks are one more, partial block)
      LGAR_MEASURE_POINT(CLK, 28)
    }
  }

  // fp64 and its dual numbers: two nodes per iteration give the scheduler two independent chains (same sums in the same
  // order; measured: backward -1.5 %, fp64 forward -1 %, a job of 157 waves -3 %)
  if constexpr (sizeof(R) == 8) {
    for (; i + 1 < n_safe; i += 2) {
      const S hb = h2 + dh;
      const S ka = node(h2);
      const S kb = node(hb);
      g = g + ((k1 + ka) * hdh);
      g = g + ((ka + kb) * hdh);
      k1 = kb;
      h2 = hb + dh;
    }
  }
  for (; i < n_safe; i++) {
    if (sizeof(R) == 4) h2 = (i + 1 >= nint) ? h_f : h_i + R(i + 1) * dh;
    const S k2 = node(h2);
    g = g + ((k1 + k2) * hdh);
    k1 = k2;
    if (sizeof(R) != 4) h2 = h2 + dh;
  }
  for (; i < nint; i++) {
    if (sizeof(R) == 4) h2 = (i + 1 >= nint) ? h_f : h_i + R(i + 1) * dh;
    S k2 = node(h2);
    k2 = choose(ab(val(h2)) < R(0.1) || val(h2) < R(0.0), k_sat1, k2);
    g = g + ((k1 + k2) * hdh);
    k1 = k2;
    if (sizeof(R) != 4) h2 = h2 + dh;
  }
  // (an end point outside the domain must still surface as NaN for the status word)
  const S res = ab(g / l.ksat);
  LGAR_MEASURE_POINT(CLK, 29)
  return outside ? res + (h_i_own + h_f_own) : res;
}
// fp32 Geff, lean form (what bench.py measures).  Per node, with x = alpha h and K_r = K / Ksat (Ksat cancels in
// G = |integral of K dh| / Ksat):
//     lg = log2 x;  P = 2^((n-1) lg) = x^(n-1) = a^m   [a = x^n, n m = n - 1];   a = x P;
//     l1 = log2(1 + a);  s = 2^(-m/2 l1) = sqrt(Se);  K_r = s (1 - P s^2)^2        [(a/(1+a))^m = a^m Se]
// i.e. FOUR transcendentals per node (v_log, v_exp, v_log, v_exp) instead of five, and no division.  The vector ALU
// issues a transcendental in 8 cycles, a packed op in 4 and a v_cndmask_b32 in 16 (measured, tools/valu_probe.py), so
// the loop is laid out to be select-free: the |h| < 0.1 -> Se = 1 rule (utils.py:124-128) can only bind on a SUFFIX of
// the nodes (h falls monotonically from h_i to h_f), so a wave-uniform count of leading node pairs that no lane needs
// to test runs in a select-free loop, the rest in a checked loop.  The trapezoid is summed as
// dh/2 (K_0 + K_n + 2 sum of interior nodes): one packed add per node pair.  Interior nodes sit at h_i + j dh (no running
// sum: it drifts by cm for very dry soil); the last node is h_f itself, which dominates the integral for dry soil.
typedef float f32x2 __attribute__((ext_vector_type(2)));
// (the trapezoid given its two heads; kn_out, when asked for: K_r of the wet end node)
__device__ __forceinline__ float geff_f32_from_heads(const LayerK<float> &l, float h_i, float h_f, int nint, float *kn_out = nullptr) {
  const float dh = (h_f - h_i) * (1.0f / float(nint));
  const float nm1 = l.n - 1.0f;
  const float hm = -0.5f * l.m;
  const float x0 = l.alpha * h_i, dx = l.alpha * dh, xcut = 0.1f * l.alpha;
  // K_r at Se == 1 (calc_k_from_se's 1e-12 nudge, utils.py:147-150): (1 - (1e-12)^m)^2
  const float tsat = 1.0f - ex2(l.m * -39.863137f);
  const float ksat1 = tsat * tsat;
  auto node = [&](float x) {
    const float lg = lg2(x);
    const float P = ex2(nm1 * lg);
    const float l1 = lg2(__builtin_fmaf(x, P, 1.0f));
    const float sr = ex2(hm * l1);
    const float t = __builtin_fmaf(-P, sr * sr, 1.0f);
    const float k = sr * (t * t);
    return (x < xcut) ? ksat1 : k;
  };
  const int M = nint - 1;       // interior nodes j = 1 .. nint-1
  const int pairs = M >> 1;
  // leading interior nodes with h >= 0.1 for certain: j < (x0 - xcut) / -dx (one node of margin for rounding)
#ifndef LGAR_DEVSIM
  const float jf = (x0 - xcut) * __builtin_amdgcn_rcpf(-dx) - 1.5f;
#else
  const float jf = (x0 - xcut) / -dx - 1.5f;
#endif
  int safe = (jf > 0.0f) ? ((jf < float(M)) ? int(jf) : M) : 0;  // NaN (dx == 0) -> 0
  safe >>= 1;
  // wave-uniform minimum: usually every lane is safe for the whole interior (only the end point h_f is near zero)
  int safe_pairs = pairs;
  if (any_lane(safe < pairs) != 0ull) {
    safe_pairs = 0;  // bisection on ballots (plain compares + scalar ops)
    for (int bit = 64; bit; bit >>= 1) {
      const int cand = safe_pairs + bit;
      if (cand <= pairs && any_lane(safe < cand) == 0ull) safe_pairs = cand;
    }
  }
  const f32x2 dx2 = {dx, dx}, x02 = {x0, x0}, nm12 = {nm1, nm1}, hm2 = {hm, hm};
  const f32x2 one2 = {1.0f, 1.0f}, two2 = {2.0f, 2.0f}, four2 = {4.0f, 4.0f};
  // A column's result must not depend on which columns share its wavefront (safe_pairs is a property of the wave): node
  // pair p always sits at x0 + (2p+1, 2p+2) dx (one fma, never a running sum), always goes through the same operations,
  // and always lands in accumulator p & 1 -- whichever of the three loops below handles it.
  f32x2 ja = {1.0f, 2.0f};
  f32x2 acc = {0.0f, 0.0f}, accb = {0.0f, 0.0f};
  // one node pair: 4 transcendentals + 4 packed ops per node-pair -> sqrt(Se) and (1 - P Se)^2 of two nodes
#define LGAR_GEFF_PAIR(X, SR, TT)                                              \
  {                                                                            \
    f32x2 lg, P, l1;                                                           \
    lg.x = lg2((X).x); lg.y = lg2((X).y);                                      \
    const f32x2 e0 = nm12 * lg;                                                \
    P.x = ex2(e0.x); P.y = ex2(e0.y);                                          \
    const f32x2 opa = __builtin_elementwise_fma((X), P, one2);                 \
    l1.x = lg2(opa.x); l1.y = lg2(opa.y);                                      \
    const f32x2 e1 = hm2 * l1;                                                 \
    (SR).x = ex2(e1.x); (SR).y = ex2(e1.y);                                    \
    const f32x2 t = __builtin_elementwise_fma(-P, (SR) * (SR), one2);          \
    (TT) = t * t;                                                              \
  }
  int it = 0;
  // four nodes per iteration: two independent chains
  for (; it + 1 < safe_pairs; it += 2) {
    const f32x2 xa = __builtin_elementwise_fma(ja, dx2, x02);
    const f32x2 xb = __builtin_elementwise_fma(ja + two2, dx2, x02);
    ja = ja + four2;
    f32x2 sa, ta, sb, tb;
    LGAR_GEFF_PAIR(xa, sa, ta)
    acc = __builtin_elementwise_fma(sa, ta, acc);
    LGAR_GEFF_PAIR(xb, sb, tb)
    accb = __builtin_elementwise_fma(sb, tb, accb);
  }
  if (it < safe_pairs) {  // `it` is even here
    const f32x2 x = __builtin_elementwise_fma(ja, dx2, x02);
    f32x2 sr, tt;
    LGAR_GEFF_PAIR(x, sr, tt)
    acc = __builtin_elementwise_fma(sr, tt, acc);
    it++;
  }
  for (; it < pairs; it++) {  // nodes that may fall under the |h| < 0.1 cut: K_r = ksat1 there
    const float j0 = float(2 * it + 1);
    const f32x2 j2 = {j0, j0 + 1.0f};
    const f32x2 x = __builtin_elementwise_fma(j2, dx2, x02);
    f32x2 sr, tt;
    LGAR_GEFF_PAIR(x, sr, tt)
    sr.x = (x.x < xcut) ? ksat1 : sr.x;
    tt.x = (x.x < xcut) ? 1.0f : tt.x;
    sr.y = (x.y < xcut) ? ksat1 : sr.y;
    tt.y = (x.y < xcut) ? 1.0f : tt.y;
    if (it & 1) accb = __builtin_elementwise_fma(sr, tt, accb);
    else acc = __builtin_elementwise_fma(sr, tt, acc);
  }
#undef LGAR_GEFF_PAIR
  acc = acc + accb;
  float sum = acc.x + acc.y;
  if (M & 1) sum += node(__builtin_fmaf(float(M), dx, x0));
  const float k0 = node(x0), kn = node(l.alpha * h_f);
  if (kn_out != nullptr) *kn_out = kn;
  return fabsf((0.5f * dh) * ((k0 + kn) + 2.0f * sum));
}
template <> __device__ __forceinline__ float geff<float>(const LayerK<float> &l, float theta1, float theta2, int nint) {
  const float se_i = se_from_theta(l, theta1);
  const float se_f = se_from_theta(l, theta2);
  // h(Se) of both end points (calc_h_from_se, utils.py:159-174) with one reciprocal of alpha
  const float inv_alpha = 1.0f / l.alpha;
  auto head = [&](float se) {
    float base = pw(se, -l.inv_m) - 1.0f;
    if (fabsf(base) <= 1e-8f) base = base + 1e-12f;
    return inv_alpha * pw(base, l.inv_n);
  };
  return geff_f32_from_heads(l, head(se_i), head(se_f), nint);
}
template <> __device__ __forceinline__ double geff<double>(const LayerK<double> &l, double t1, double t2, int nint) {
  return geff_fused<double>(l, t1, t2, nint);
}

// Mixed-precision Geff (LgarDims.geff_mode = 1, fp64 runs): the column state, every branch and the mass bookkeeping stay
// in double precision; of the trapezoid, only the 119 INTERIOR nodes are evaluated with the fp32 hardware transcendentals.
//   * both heads h(Se) (calc_h_from_se, utils.py:159-174), dh, and the two END nodes K(Se_i), K(Se_f)
//     (calc_k_from_se, utils.py:134-156) are double precision.  K falls steeply with h (like h^-3.7 for the bundled soils),
//     so over a wide range the wet end node carries most of the sum: it must not carry fp32 rounding;
//   * interior node j sits at x_j = alpha h_i + j alpha dh, formed from fp32 hi + lo pairs of both terms (within 1 ulp of
//     the double-precision value); the exponents
//     n - 1 and -m/2 enter as fp32 pairs hi + lo (their rounding would otherwise be a SYSTEMATIC relative error of
//     ~1e-7 |log2 x| in every node; what remains -- the 1-ulp errors of v_log_f32 / v_exp_f32 and of x -- is random from
//     node to node and averages over the sum);
//   * the nodes' K_r are added up in double (per group of four nodes: two fp32 adds, one convert, one fp64 add).
// Per interior node: five hardware transcendentals + ~12 packed fp32 operations (the node is written without the cancelling
// difference 1 - (a/(1+a))^m of the fp32 loop: see LGAR_GEFFM_PAIR), against the ~93 fp64 instructions of the fused fp64 node.
// The |h| < 0.1 -> Se = 1 rule and the wave-uniform select-free prefix are those of the fp32 loop above; as there, a group of
// nodes always goes through the same operations in the same order, so a column's result does not depend on its wavefront.
// One end of the trapezoid in double precision: h(Se) (calc_h_from_se, utils.py:159-174) and K(Se) / Ksat (calc_k_from_se,
// utils.py:134-156) from SHARED logarithms.  With q = log2 Se^(1/m), C = 2^q and u = log2(1 - C):
//     K_r = sqrt(Se) (1 - 2^(m u))^2,     h = (1/alpha) 2^((u - q)/n)      [Se^(-1/m) - 1 = (1 - C)/C]
// i.e. two logarithms and three exponentials where the two functions on their own take four pows.  The reference's nudges
// (|base| <= 1e-8 -> base + 1e-12, both functions) are reproduced: at Se == 1 both bases are exactly 0 and share the
// logarithm of 1e-12; a base in (0, 1e-8] (Se within 1e-8 of 1 but not 1) takes its own logarithm on a wave-uniform branch.
#define LGAR_LOG2_1EM12 -39.863137138648355  // log2(1e-12): the nudged base at Se == 1
__device__ __forceinline__ void mixed_end(const LayerK<double> &l, double se, double &h, double &kr) {
  const double q = lg2e(se) * l.inv_m;
  const double C = ex2e(q);  // Se^(1/m)
  const double omc = 1.0 - C;
  const bool k_nudged = fabs(omc) <= 1e-8;
  const double bk = k_nudged ? omc + 1e-12 : omc;
  const double u = (omc == 0.0) ? LGAR_LOG2_1EM12 : lg2e(bk);  // (the constant: the saturated end below takes the same value)
  const double t = 1.0 - ex2e(l.m * u);
  kr = sqrt(se) * (t * t);
  const double bh = omc / C;  // Se^(-1/m) - 1
  const bool h_nudged = fabs(bh) <= 1e-8;
  double lbh = (omc == 0.0) ? u : u - q;  // log2 of the (nudged) base of h
  if (any_lane((k_nudged || h_nudged) && omc != 0.0) != 0ull) {
    const double own = lg2e(h_nudged ? bh + 1e-12 : bh);
    lbh = ((k_nudged || h_nudged) && omc != 0.0) ? own : lbh;
  }
  h = (1.0 / l.alpha) * ex2e(lbh * l.inv_n);
}
// The interior of the mixed-precision trapezoid and its closing formula, given both heads, K_r of both end nodes and K_r at
// Se == 1 (see geff_mixed / geff_mixed_heads for where they come from).
// COOPERATE (cooperating lanes, MODE 6): the lanes of a group carry the same column; lane r of `lanes` evaluates the
// four-node groups r, r + lanes, ... in the general, checked form -- the value every form gives a group, bit for bit -- and
// leaves each group's sum in the group's LDS table `tab`; every lane then adds the groups up in order, as the loops below do.
template <bool COOPERATE = false>
__device__ __forceinline__ double geff_mixed_core(const LayerK<double> &l, double h_i, double h_f, double k0, double kn_own, double ksat1,
                                                  int nint, double *tab = nullptr, int lanes = 0, int r = 0) {
  (void)tab; (void)lanes; (void)r;
  const float ksat1f = (float)ksat1;
  const double dh = (h_f - h_i) / double(nint);
  const double x0 = l.alpha * h_i, dx = l.alpha * dh, xcut = 0.1 * l.alpha;
  // exponents as fp32 pairs
  const double hmd = -0.5 * l.m;
  const float hm = (float)hmd, hm_lo = (float)(hmd - (double)hm);
  const int M = nint - 1;  // interior nodes j = 1 .. nint-1
  const int pairs = M >> 1;
  // (node counts from abscissae: one reciprocal of -dx serves the three of them; each count keeps a node or more of margin, and
  // which loop evaluates a node never changes its value)
  const double inv_ndx = 1.0 / -dx;
  const double jf = (x0 - xcut) * inv_ndx - 1.5;
  int safe = (jf > 0.0) ? ((jf < double(M)) ? int(jf) : M) : 0;  // NaN (dx == 0, or a head outside the domain) -> 0
  safe >>= 1;
  int safe_pairs = pairs;
  if (!COOPERATE && any_lane(safe < pairs) != 0ull) {
    safe_pairs = 0;
    for (int bit = 64; bit; bit >>= 1) {
      const int cand = safe_pairs + bit;
      if (cand <= pairs && any_lane(safe < cand) == 0ull) safe_pairs = cand;
    }
  }
  const float xcutf = (float)xcut;
  const float nf = (float)l.n, nf_lo = (float)(l.n - (double)nf), mmf = (float)(-l.m), mmf_lo = (float)(-l.m - (double)mmf);
  const f32x2 n2 = {nf, nf}, nl2 = {nf_lo, nf_lo}, hm2 = {hm, hm}, hml2 = {hm_lo, hm_lo}, one2 = {1.0f, 1.0f};
  // expm1(E ln 2) / E = ln 2 + E (ln^2 2 / 2 + E (ln^3 2 / 6 + E (ln^4 2 / 24 + E ln^5 2 / 120)))
  const f32x2 ln2_2 = {0.693147181f, 0.693147181f}, c2_2 = {0.240226507f, 0.240226507f}, c3_2 = {0.0555041087f, 0.0555041087f};
  const f32x2 c4_2 = {0.00961812911f, 0.00961812911f}, c5_2 = {0.00133335581f, 0.00133335581f};
  const f32x2 mm2 = {mmf, mmf}, mml2 = {mmf_lo, mmf_lo}, ilog2 = {1.44269504f, 1.44269504f}, two_ilog2 = {2.88539008f, 2.88539008f};
  const f32x2 half2 = {0.5f, 0.5f};
  // one node pair: sqrt(Se) and (1 - P Se)^2 of the nodes at X.x, X.y.
  // KIND says what is KNOWN about the pair (a compile-time literal; it never changes a result, only which of two values that
  // the general form computes and then discards is not computed at all):
  //   0  nothing: t = 1 - 2^E and its series are both formed and chosen between by 2^E > 7/8 (the general form);
  //   1  2^E > 7/8 for certain (dry nodes): the series only -- no 2^E, no select, and 1 + r < 2 needs no clamp;
  //   2  2^E <= 7/8 for certain (wet nodes): the difference only -- no series, no select.
#define LGAR_GEFFM_PAIR(X, SR, TT, KIND)                                                    \
  {                                                                                         \
    /* log2 a = n log2 x;  r = 1/a;  L = log2(1 + r) from c = fl(1 + r), compensated for the rounding of the sum  */ \
    f32x2 lg, r, Lc;                                                                        \
    lg.x = lg2((X).x); lg.y = lg2((X).y);                                                   \
    const f32x2 la = __builtin_elementwise_fma(n2, lg, nl2 * lg);                           \
    r.x = ex2(-la.x); r.y = ex2(-la.y);                                                     \
    const f32x2 c = one2 + r;                                                               \
    const f32x2 rho = r - (c - one2);                                                       \
    Lc.x = lg2(c.x); Lc.y = lg2(c.y);                                                       \
    /* (2 - c) / ln 2 ~ 1 / (c ln 2) where the correction matters (c near 1); nothing for c >= 2 */ \
    f32x2 ic = __builtin_elementwise_fma(-ilog2, c, two_ilog2);                             \
    if ((KIND) != 1) { ic.x = fmaxf(ic.x, 0.0f); ic.y = fmaxf(ic.y, 0.0f); }                \
    const f32x2 L = __builtin_elementwise_fma(rho, ic, Lc);                                 \
    /* E = -m L = log2 (a/(1+a))^m;  sqrt(Se) = (1 + a)^(-m/2) = 2^(-m/2 (log2 a + L)) = 2^(-m/2 log2 a + E/2) */ \
    const f32x2 E = __builtin_elementwise_fma(mm2, L, mml2 * L);                            \
    const f32x2 e1 = __builtin_elementwise_fma(hm2, la, __builtin_elementwise_fma(hml2, la, half2 * E)); \
    (SR).x = ex2(e1.x); (SR).y = ex2(e1.y);                                                 \
    /* t = 1 - 2^E.  Dry nodes have 2^E -> 1: there t = -expm1(E ln 2) by its series in E (5 terms for 2^E > 7/8), not by \
       the cancelling difference */                                                         \
    f32x2 t;                                                                                \
    if ((KIND) == 1) {                                                                      \
      f32x2 p = __builtin_elementwise_fma(c5_2, E, c4_2);                                   \
      p = __builtin_elementwise_fma(p, E, c3_2);                                            \
      p = __builtin_elementwise_fma(p, E, c2_2);                                            \
      p = __builtin_elementwise_fma(p, E, ln2_2);                                           \
      t = -E * p;                                                                           \
    } else if ((KIND) == 2) {                                                               \
      f32x2 w;                                                                              \
      w.x = ex2(E.x); w.y = ex2(E.y);                                                       \
      t = one2 - w;                                                                         \
    } else {                                                                                \
      f32x2 w;                                                                              \
      w.x = ex2(E.x); w.y = ex2(E.y);                                                       \
      f32x2 p = __builtin_elementwise_fma(c5_2, E, c4_2);                                   \
      p = __builtin_elementwise_fma(p, E, c3_2);                                            \
      p = __builtin_elementwise_fma(p, E, c2_2);                                            \
      p = __builtin_elementwise_fma(p, E, ln2_2);                                           \
      const f32x2 ts = -E * p;                                                              \
      t = one2 - w;                                                                         \
      t.x = (w.x > 0.875f) ? ts.x : t.x;                                                    \
      t.y = (w.y > 0.875f) ? ts.y : t.y;                                                    \
    }                                                                                       \
    (TT) = t * t;                                                                           \
  }
  // node abscissae x_j = x0 + j dx in fp32 from hi + lo pairs of x0 and dx: t = fma(j, dx_hi, x0_hi) is rounded once (and is
  // exact where x0 and j dx cancel), the lo parts restore what the hi parts dropped; x_j is within 1 ulp of the double-precision
  // value rounded to fp32 (the node's own v_log_f32 error is ten times that)
  const float x0h = (float)x0, dxh = (float)dx;
  const float x0l = (fabsf(x0h) < __builtin_inff()) ? (float)(x0 - (double)x0h) : 0.0f;
  const float dxl = (fabsf(dxh) < __builtin_inff()) ? (float)(dx - (double)dxh) : 0.0f;
  const f32x2 x0h2 = {x0h, x0h}, x0l2 = {x0l, x0l}, dxh2 = {dxh, dxh}, dxl2 = {dxl, dxl};
  const f32x2 four2 = {4.0f, 4.0f};
  double sum = 0.0;
  int it = 0;
  f32x2 ja = {1.0f, 2.0f}, jb = {3.0f, 4.0f};  // node indices of the current pairs: small integers, exact in fp32
  // four nodes per iteration, two independent chains; their K_r are added in fp32 ({K_j + K_j+2, K_j+1 + K_j+3}, then the two
  // halves) and the group's sum goes into the double-precision accumulator.  Groups whose nodes may fall under the
  // |h| < 0.1 cut (wave-uniform test) take K_r = ksat1 there: the same operations otherwise, so a column's result does not
  // depend on its wavefront
  // The two pairs of a group go through the node formula in LOCKSTEP, statement by statement: every packed operation of a
  // node depends on the one before it, and a dependent operation cannot issue in the slot after its producer (the compiler
  // fills those slots with s_nop when it has nothing else) -- a wave that walks one chain after the other spends half its issue
  // slots waiting.  Same operations on the same operands as LGAR_GEFFM_PAIR, pair by pair: the same values bit for bit.
#define LGAR_GEFFM_GROUP(CUT, KIND, SINK)                                                   \
  {                                                                                         \
    const f32x2 xha = __builtin_elementwise_fma(ja, dxh2, x0h2), xhb = __builtin_elementwise_fma(jb, dxh2, x0h2); \
    const f32x2 xla = __builtin_elementwise_fma(ja, dxl2, x0l2), xlb = __builtin_elementwise_fma(jb, dxl2, x0l2); \
    const f32x2 xa = xha + xla, xb = xhb + xlb;                                             \
    f32x2 lga, lgb, ra, rb, Lca, Lcb, sa, sb, ta, tb;                                       \
    lga.x = lg2(xa.x); lgb.x = lg2(xb.x); lga.y = lg2(xa.y); lgb.y = lg2(xb.y);             \
    const f32x2 ua = nl2 * lga, ub = nl2 * lgb;                                             \
    const f32x2 laa = __builtin_elementwise_fma(n2, lga, ua), lab = __builtin_elementwise_fma(n2, lgb, ub); \
    ra.x = ex2(-laa.x); rb.x = ex2(-lab.x); ra.y = ex2(-laa.y); rb.y = ex2(-lab.y);         \
    const f32x2 ca = one2 + ra, cb = one2 + rb;                                             \
    Lca.x = lg2(ca.x); Lcb.x = lg2(cb.x); Lca.y = lg2(ca.y); Lcb.y = lg2(cb.y);             \
    const f32x2 da = ca - one2, db = cb - one2;                                             \
    f32x2 ica = __builtin_elementwise_fma(-ilog2, ca, two_ilog2), icb = __builtin_elementwise_fma(-ilog2, cb, two_ilog2); \
    const f32x2 rhoa = ra - da, rhob = rb - db;                                             \
    if ((KIND) != 1) {                                                                      \
      ica.x = fmaxf(ica.x, 0.0f); icb.x = fmaxf(icb.x, 0.0f); ica.y = fmaxf(ica.y, 0.0f); icb.y = fmaxf(icb.y, 0.0f); \
    }                                                                                       \
    const f32x2 La = __builtin_elementwise_fma(rhoa, ica, Lca), Lb = __builtin_elementwise_fma(rhob, icb, Lcb); \
    const f32x2 va = mml2 * La, vb = mml2 * Lb;                                             \
    const f32x2 Ea = __builtin_elementwise_fma(mm2, La, va), Eb = __builtin_elementwise_fma(mm2, Lb, vb); \
    const f32x2 ha = half2 * Ea, hb = half2 * Eb;                                           \
    const f32x2 ga = __builtin_elementwise_fma(hml2, laa, ha), gb = __builtin_elementwise_fma(hml2, lab, hb); \
    const f32x2 e1a = __builtin_elementwise_fma(hm2, laa, ga), e1b = __builtin_elementwise_fma(hm2, lab, gb); \
    sa.x = ex2(e1a.x); sb.x = ex2(e1b.x); sa.y = ex2(e1a.y); sb.y = ex2(e1b.y);             \
    f32x2 wa, wb, pa, pb, t1a, t1b;                                                         \
    if ((KIND) != 1) {                                                                      \
      wa.x = ex2(Ea.x); wb.x = ex2(Eb.x); wa.y = ex2(Ea.y); wb.y = ex2(Eb.y);               \
      t1a = one2 - wa; t1b = one2 - wb;                                                     \
    }                                                                                       \
    if ((KIND) != 2) {                                                                      \
      pa = __builtin_elementwise_fma(c5_2, Ea, c4_2); pb = __builtin_elementwise_fma(c5_2, Eb, c4_2); \
      pa = __builtin_elementwise_fma(pa, Ea, c3_2); pb = __builtin_elementwise_fma(pb, Eb, c3_2); \
      pa = __builtin_elementwise_fma(pa, Ea, c2_2); pb = __builtin_elementwise_fma(pb, Eb, c2_2); \
      pa = __builtin_elementwise_fma(pa, Ea, ln2_2); pb = __builtin_elementwise_fma(pb, Eb, ln2_2); \
      pa = -Ea * pa; pb = -Eb * pb;                                                         \
    }                                                                                       \
    if ((KIND) == 1) { t1a = pa; t1b = pb; }                                                \
    if ((KIND) == 0) {                                                                      \
      t1a.x = (wa.x > 0.875f) ? pa.x : t1a.x; t1b.x = (wb.x > 0.875f) ? pb.x : t1b.x;       \
      t1a.y = (wa.y > 0.875f) ? pa.y : t1a.y; t1b.y = (wb.y > 0.875f) ? pb.y : t1b.y;       \
    }                                                                                       \
    ta = t1a * t1a; tb = t1b * t1b;                                                         \
    if (CUT) {                                                                              \
      sa.x = (xa.x < xcutf) ? ksat1f : sa.x; sb.x = (xb.x < xcutf) ? ksat1f : sb.x;         \
      ta.x = (xa.x < xcutf) ? 1.0f : ta.x; tb.x = (xb.x < xcutf) ? 1.0f : tb.x;             \
      sa.y = (xa.y < xcutf) ? ksat1f : sa.y; sb.y = (xb.y < xcutf) ? ksat1f : sb.y;         \
      ta.y = (xa.y < xcutf) ? 1.0f : ta.y; tb.y = (xb.y < xcutf) ? 1.0f : tb.y;             \
    }                                                                                       \
    const f32x2 ka = sa * ta;                                                               \
    const f32x2 kb = __builtin_elementwise_fma(sb, tb, ka);                                 \
    SINK((double)(kb.x + kb.y))                                                             \
  }
#define LGAR_GEFFM_TO_SUM(v) sum = sum + (v);
  // Which of the two forms of t a node takes depends on 2^E > 7/8, and E rises monotonically with x = alpha h: the series
  // region is a PREFIX of the nodes (x > x_thr, the dry end), the difference region a suffix.  x_thr -- (1 + x^-n)^-m = 7/8 --
  // is a function of the layer's n alone; a node further than 1e-4 (relative) from it is decided whatever the rounding of
  // its E (1e-4 in x moves E by >= 1e-5, the computed E and 2^E are good to ~3e-7).  Each lane counts the leading nodes that
  // are series for certain and the first node from which all are differences for certain; the wavefront runs the series-only
  // form up to the smallest of the former, the difference-only form from the largest of the latter, the general form in
  // between -- every node gets exactly the value the general form alone would give it (bit for bit: tests/devsim).
  int ser_pairs = 0, dir_pair = pairs + 1;  // (a reversed or empty range, or NaN: the general form throughout)
  if constexpr (COOPERATE) {
    const int ngroups = pairs >> 1;  // the full groups of four interior nodes
    lds_exchange_point();            // (earlier loads of the table stay in front of these stores)
    for (int q = (r < lanes) ? r : ngroups; q < ngroups; q += lanes) {
      const float j0 = (float)(4 * q);
      ja = f32x2{j0 + 1.0f, j0 + 2.0f};
      jb = f32x2{j0 + 3.0f, j0 + 4.0f};
#define LGAR_GEFFM_TO_TAB(v) tab[q] = (v);
      LGAR_GEFFM_GROUP(true, 0, LGAR_GEFFM_TO_TAB)
#undef LGAR_GEFFM_TO_TAB
    }
    lds_exchange_point();
    {
      int q0 = 0;
      for (; q0 + 8 <= ngroups; q0 += 8) {
        double tq[8];
#pragma unroll
        for (int j = 0; j < 8; j++) tq[j] = tab[q0 + j];  // same address in every lane of the group: a broadcast
#pragma unroll
        for (int j = 0; j < 8; j++) sum = sum + tq[j];
      }
      for (; q0 < ngroups; q0++) sum = sum + tab[q0];
    }
    lds_exchange_point();
    it = 2 * ngroups;
    {
      const float j0 = (float)(4 * ngroups);
      ja = f32x2{j0 + 1.0f, j0 + 2.0f};
      jb = f32x2{j0 + 3.0f, j0 + 4.0f};
    }
  }
  if (!COOPERATE && dx < 0.0) {
    const float x_thr = pw(pw(8.0f / 7.0f, (float)l.inv_m) - 1.0f, -(float)l.inv_n);
    const double js_f = (x0 - (double)(x_thr * 1.0001f)) * inv_ndx - 1.0;  // nodes 1 .. js: x_j > x_thr (1 + 1e-4)
    const int js = (js_f > 0.0) ? ((js_f < double(M)) ? int(js_f) : M) : 0;
    ser_pairs = js >> 1;                                               // pairs 0 .. ser_pairs - 1 hold only such nodes
    const double jd_f = (x0 - (double)(x_thr * 0.9999f)) * inv_ndx + 2.0;  // nodes jd ..: x_j < x_thr (1 - 1e-4)
    const int jd = !(jd_f < double(M + 2)) ? M + 2 : ((jd_f > 0.0) ? int(jd_f) : 0);
    dir_pair = jd >> 1;                                                // pairs from dir_pair on hold only such nodes
  }
  LGAR_MEASURE_POINT(GEFFM_GENERAL_ONLY, ser_pairs, dir_pair, pairs)
  int ser_all = safe_pairs;  // the wavefront's: min of ser_pairs (at most safe_pairs), max of dir_pair
  if (!COOPERATE && any_lane(ser_pairs < ser_all) != 0ull) {
    ser_all = 0;
    for (int bit = 64; bit; bit >>= 1) {
      const int cand = ser_all + bit;
      if (cand <= safe_pairs && any_lane(ser_pairs < cand) == 0ull) ser_all = cand;
    }
  }
  int dir_all = 0;
  if constexpr (!COOPERATE) {
  for (int bit = 64; bit; bit >>= 1) {
    const int cand = dir_all + bit;
    if (any_lane(dir_pair >= cand) != 0ull) dir_all = cand;
  }
  LGAR_MEASURE_POINT(CLK, 13)
  for (; it + 1 < ser_all; it += 2, ja = ja + four2, jb = jb + four2) LGAR_GEFFM_GROUP(false, 1, LGAR_GEFFM_TO_SUM)
  const int it_a = it;
  for (; it + 1 < safe_pairs && it < dir_all; it += 2, ja = ja + four2, jb = jb + four2) LGAR_GEFFM_GROUP(false, 0, LGAR_GEFFM_TO_SUM)
  const int it_b = it;
  for (; it + 1 < safe_pairs; it += 2, ja = ja + four2, jb = jb + four2) LGAR_GEFFM_GROUP(false, 2, LGAR_GEFFM_TO_SUM)
  const int it_c = it;
  for (; it + 1 < pairs; it += 2, ja = ja + four2, jb = jb + four2) LGAR_GEFFM_GROUP(true, 0, LGAR_GEFFM_TO_SUM)
  LGAR_MEASURE_POINT(GEFFM_REGIONS, it_a >> 1, (it_b - it_a) >> 1, (it_c - it_b) >> 1, (it - it_c) >> 1)
  LGAR_MEASURE_POINT(CLK, 14)
  }
#undef LGAR_GEFFM_GROUP
#undef LGAR_GEFFM_TO_SUM
  const int rem = M - 2 * it;  // interior nodes left over by the groups of four: 0..3 (3 for the reference's 120 intervals)
  if (rem > 0) {             // ... as one more group whose surplus nodes count as zero
    const f32x2 xa = __builtin_elementwise_fma(ja, dxh2, x0h2) + __builtin_elementwise_fma(ja, dxl2, x0l2);
    const f32x2 xb = __builtin_elementwise_fma(jb, dxh2, x0h2) + __builtin_elementwise_fma(jb, dxl2, x0l2);
    f32x2 sa, ta, sb, tb;
    LGAR_GEFFM_PAIR(xa, sa, ta, 0)
    LGAR_GEFFM_PAIR(xb, sb, tb, 0)
    f32x2 ka = sa * ta, kb = sb * tb;
    ka.x = (xa.x < xcutf) ? ksat1f : ka.x;
    ka.y = (xa.y < xcutf) ? ksat1f : ka.y;
    kb.x = (xb.x < xcutf) ? ksat1f : kb.x;
    ka.y = (rem >= 2) ? ka.y : 0.0f;
    kb.x = (rem >= 3) ? kb.x : 0.0f;
    sum = sum + (double)((ka.x + kb.x) + ka.y);
  }
#undef LGAR_GEFFM_PAIR
  // end nodes in double precision, with the reference's own formulas (|h| < 0.1 -> Se = 1 applies to the LAST node only:
  // the first node's K is calc_k_from_se(Se_i) as it stands, green_ampt.py:60)
  // (the last node's Se is Se(h(Se_f)) in the reference: Se_f up to the rounding of the round trip)
  const double kn = (fabs(h_f) < 0.1 || h_f < 0.0) ? ksat1 : kn_own;
  const double res = fabs((0.5 * dh) * ((k0 + kn) + 2.0 * sum));
  const bool outside = is_nan(h_i) || is_nan(h_f);
  LGAR_MEASURE_POINT(CLK, 16)
  return outside ? res + (h_i + h_f) : res;
}
// calc_geff(theta1 -> theta2) in the mixed-precision mode: heads and end nodes from the two water contents (mixed_end).
// NOT inlined: its callers are the two call sites that run rarely (insert_water on a memo miss: 16 % of the wave-level
// evaluations; the dry-depth evaluation: 1 %) -- calc_dzdt, the hot one, has its own inlined copy (geff_mixed_heads).  Two fewer
// copies of the trapezoid in the kernel: 93 -> 52 spilled registers, and the size of the code is part of its speed (build.py).
// (Round 5, same-box A/B: inlined it is 1.7 % slower -- 30.2 against 29.7 ms, 130 spilled registers against 86 -- although the
// register saves around this call are 3.6 GB of the kernel's 13.5 GB of scratch write-back per launch.)
// (The layer's parameters travel as eight scalar arguments -- in registers.  A LayerK by reference is a struct the caller must
// first build in scratch memory: 64 bytes per lane stored at every call and loaded back by the callee, and by value the
// aggregate is past the 16 argument registers the ABI gives a struct, so it would go through scratch all the same.)
__device__ __attribute__((noinline)) double geff_mixed(double alpha, double n, double m, double inv_m, double inv_n, double ksat, double te,
                                                       double tr, double theta1, double theta2, int nint) {
  const LayerK<double> l{alpha, n, m, inv_m, inv_n, ksat, te, tr};
  const double se_i = se_from_theta(l, theta1);
  const double se_f = se_from_theta(l, theta2);
  // K_r at Se == 1 (the 1e-12 nudge of calc_k_from_se): (1 - (1e-12)^m)^2
  const double tsat = 1.0 - ex2p(l.m * LGAR_LOG2_1EM12);
  const double ksat1 = tsat * tsat;
  double h_i, h_f, k0, kn_own;
  mixed_end(l, se_i, h_i, k0);
  if (any_lane(se_f != 1.0) != 0ull) {
    mixed_end(l, se_f, h_f, kn_own);
  } else {  // every lane's wet end is saturated (theta_2 == theta_e: new fronts, infiltration): what mixed_end returns for Se == 1
    h_f = (1.0 / l.alpha) * ex2e(LGAR_LOG2_1EM12 * l.inv_n);
    kn_own = ksat1;
  }
  return geff_mixed_core(l, h_i, h_f, k0, kn_own, ksat1, nint);
}
// ... for cooperating lanes (MODE 6: insert_water, the dry-depth evaluation): the same ends, the interior split over the lanes
__device__ __attribute__((noinline)) double geff_mixed_coop(double alpha, double n, double m, double inv_m, double inv_n, double ksat,
                                                            double te, double tr, double theta1, double theta2, int nint, double *tab,
                                                            int lanes, int r) {
  const LayerK<double> l{alpha, n, m, inv_m, inv_n, ksat, te, tr};
  const double se_i = se_from_theta(l, theta1);
  const double se_f = se_from_theta(l, theta2);
  const double tsat = 1.0 - ex2p(l.m * LGAR_LOG2_1EM12);
  const double ksat1 = tsat * tsat;
  double h_i, h_f, k0, kn_own;
  mixed_end(l, se_i, h_i, k0);
  mixed_end(l, se_f, h_f, kn_own);  // (geff_mixed's wave-wide shortcut for Se_f == 1 returns what mixed_end returns there)
  return geff_mixed_core<true>(l, h_i, h_f, k0, kn_own, ksat1, nint, tab, lanes, r);
}
// K(Se) / Ksat of both ends of a trapezoid (calc_k_from_se, utils.py:134-156, nudge included), the two evaluated in lockstep:
// each is a chain of two logarithms and two exponentials in which every operation waits for the one before it.
__device__ __forceinline__ void mixed_k_pair(const LayerK<double> &l, double se_a, double se_b, double &kr_a, double &kr_b) {
  const double qa = lg2e(se_a) * l.inv_m, qb = lg2e(se_b) * l.inv_m;
  const double oa = 1.0 - ex2e(qa), ob = 1.0 - ex2e(qb);  // 1 - Se^(1/m)
  const double ba = (fabs(oa) <= 1e-8) ? oa + 1e-12 : oa, bb = (fabs(ob) <= 1e-8) ? ob + 1e-12 : ob;
  const double ua = (oa == 0.0) ? LGAR_LOG2_1EM12 : lg2e(ba), ub = (ob == 0.0) ? LGAR_LOG2_1EM12 : lg2e(bb);
  const double ta = 1.0 - ex2e(l.m * ua), tb = 1.0 - ex2e(l.m * ub);
  kr_a = sqrt(se_a) * (ta * ta);
  kr_b = sqrt(se_b) * (tb * tb);
}
// calc_geff(theta1 -> theta2) for two FRONTS of the table, mixed-precision mode (calc_dzdt, calc_dry_depth): the heads are the
// fronts' own psi -- every front carries psi = h(Se(theta)) or the psi its theta was computed from (Column::move_wetting_front),
// so h(Se(theta)) need not be formed again: it would differ from psi by the rounding of the round trip, ~1e-10 relative, three
// orders below what the fp32 interior resolves -- and only K_r of the two end nodes is evaluated.  kr_f: K(theta2) / Ksat, which
// calc_dzdt needs as the front's own conductivity (Layer.py:1212-1216) and would otherwise compute a second time.
template <bool COOPERATE = false>
__device__ __forceinline__ double geff_mixed_heads(const LayerK<double> &l, double theta1, double theta2, double psi1, double psi2, int nint,
                                                   double &kr_f, double *tab = nullptr, int lanes = 0, int r = 0) {
  const double se_i = se_from_theta(l, theta1);
  const double se_f = se_from_theta(l, theta2);
  const double tsat = 1.0 - ex2p(l.m * LGAR_LOG2_1EM12);
  const double ksat1 = tsat * tsat;
  double k0, kn_own;
  LGAR_MEASURE_POINT(CLK, 11)
  mixed_k_pair(l, se_i, se_f, k0, kn_own);
  LGAR_MEASURE_POINT(CLK, 12)
  kr_f = kn_own;
  return geff_mixed_core<COOPERATE>(l, psi1, psi2, k0, kn_own, ksat1, nint, tab, lanes, r);
}
// calc_geff with use_closed_form_G (lgar/green_ampt.py:85-98): Brooks-Corey estimate from the van Genuchten parameters
// (calc_bc_lambda / calc_bc_psib, physics/utils.py:54-64, 84-99).  Operator precedence as written in the reference:
// geff = h_c * Se_i^e - Se_f^e / (1 - Se_f^e), with Se_f from theta_1 and Se_i from theta_2; inf/nan -> h_c.
template <typename S, int POL = 0> __device__ __forceinline__ S geff_closed(const LayerK<S> &l, S theta1, S theta2) {
  using R = real_t<S>;
  const S p = R(1.0) + (R(2.0) / l.m);
  const S lambda = R(2.0) / (p - R(3.0));
  const S psib = (p + R(3.0)) * (R(147.8) + R(8.1) * p + R(0.092) * p * p) /
                 (R(2.0) * l.alpha * p * (p - R(1.0)) * (R(55.6) + R(7.4) * p + p * p));
  const S se_f = se_from_theta(l, theta1);
  const S se_i = se_from_theta(l, theta2);
  const S h_c = psib * (R(2.0) + R(3.0) * lambda) / (R(1.0) + R(3.0) * lambda);
  const S e = R(3.0) + R(1.0) / lambda;
  const S pf = pwq<S, POL>(se_f, e);
  S g = h_c * pwq<S, POL>(se_i, e) - pf / (R(1.0) - pf);
  const R gv = val(g);
  if (gv != gv || gv - gv != R(0.0)) g = h_c;  // torch.isinf / torch.isnan
  return g;
}

// calc_aet, models/physics/lgar/aet.py:17-51 (0.75: GlobalParams.py:75; clamp upper bound = PET rate).  The head at which
// uptake halves (aet.py:31-40) depends on the top layer's parameters only: aet_psi_wp computes it, the column keeps it.
template <typename S, int POL = 0> __device__ __forceinline__ S aet_psi_wp(const LayerK<S> &l, real_t<S> wp_psi) {
  using R = real_t<S>;
  S theta_fc = (l.te - l.tr) * R(0.75) + l.tr;
  S wp_head_theta = theta_from_h<S, POL>(l, S(wp_psi));
  S theta_wp = (theta_fc - wp_head_theta) * R(0.5) + wp_head_theta;
  S se = se_from_theta(l, theta_wp);
  return h_from_se<S, POL>(l, se);
}
template <typename S, int POL = 0> __device__ __forceinline__ S aet_from_psi_wp(S pet, real_t<S> dt_h, S psi, S psi_wp) {
  using R = real_t<S>;
  S r = psi / psi_wp;
  S h_ratio = R(1.0) + r * r * r;
  S a = pet * (R(1.0) / h_ratio) * dt_h;
  if (val(a) < R(0.0)) a = S(R(0.0));
  if (val(a) > val(pet)) a = pet;
  return a;
}
template <typename S, int POL = 0> __device__ __forceinline__ S aet_fn(const LayerK<S> &l, S pet, real_t<S> dt_h, S psi, real_t<S> wp_psi) {
  using R = real_t<S>;
  S psi_wp = aet_psi_wp<S, POL>(l, wp_psi);
  S r = psi / psi_wp;
  S h_ratio = R(1.0) + r * r * r;
  S a = pet * (R(1.0) / h_ratio) * dt_h;
  if (val(a) < R(0.0)) a = S(R(0.0));
  if (val(a) > val(pet)) a = pet;
  return a;
}

// ---------------------------------------------------------------------------------------------
// run-time constants shared by all columns (kernel argument -> SGPRs)
// ---------------------------------------------------------------------------------------------
template <typename R> struct Glob {
  R dt_h, initial_psi, pdm, wp_psi, frozen;
  R giuh[LGAR_GMAX];
  int nint, nsub, ng, bottom_mode, closed_form;
  long long iter_cap;
};

// LDS view of one lane's fronts: element i of field f sits at base[(f * FMAX + i) * WAVE] (one base address per lane; the
// field offsets are compile-time constants folded into the ds_read / ds_write offsets)
// STRIDE: table slots per front row -- 64, one per lane; the cooperating-lanes kernels (MODE 4) keep ONE table per group of
// lanes (LGAR_COOP_GROUPS slots: the lanes of a group hold the same column, read the same address -- an LDS broadcast -- and
// write the same value to it), which is what lets a 32-front table of such a wave fit four waves per CU.
#define LGAR_COOP_GROUPS 16
constexpr bool coop_mode(int mode) { return mode == 4 || mode == 6; }
constexpr bool mixed_mode(int mode) { return mode == 3 || mode == 6; }  // (MODE 6: the mixed-precision trapezoid AND cooperating lanes)
template <typename S, int FMAX, int STRIDE = WAVE> struct FrontsView {
  S *base;             // &lds.f[0][0][slot]
  unsigned char *fl;   // &lds.fl[0][slot]
  __device__ __forceinline__ S &Z(int i) const { return base[(0 * FMAX + i) * STRIDE]; }
  __device__ __forceinline__ S &TH(int i) const { return base[(1 * FMAX + i) * STRIDE]; }
  __device__ __forceinline__ S &PS(int i) const { return base[(2 * FMAX + i) * STRIDE]; }
  __device__ __forceinline__ S &DZ(int i) const { return base[(3 * FMAX + i) * STRIDE]; }
  __device__ __forceinline__ unsigned char &flag(int i) const { return fl[i * STRIDE]; }
  __device__ __forceinline__ int layer(int i) const { return fl[i * STRIDE] & 0x7f; }
  __device__ __forceinline__ bool bottom(int i) const { return (fl[i * STRIDE] & LGAR_FLAG_BOTTOM) != 0; }
  __device__ __forceinline__ void set_flag(int i, int layer, bool bottom) const {
    fl[i * STRIDE] = (unsigned char)(layer | (bottom ? LGAR_FLAG_BOTTOM : 0));
  }
  __device__ __forceinline__ void copy(int dst, int src) const {
    Z(dst) = Z(src); TH(dst) = TH(src); PS(dst) = PS(src); DZ(dst) = DZ(src);
    fl[dst * STRIDE] = fl[src * STRIDE];
  }
};

// ---------------------------------------------------------------------------------------------
// one soil column
// ---------------------------------------------------------------------------------------------
// MODE 0: the reference's literal line searches, its update_psi pass and per-sub-step NaN scan.
// MODE 1 (default of the engine, what bench.py measures): the same roots by bracketed Newton / closed-form jumps; the
// passes that are provably no-ops between events are skipped (see forward()).
// MODE 3 (double precision only, LgarDims.geff_mode = 1): MODE 1 with the mixed-precision trapezoid (geff_mixed).
// MODE 4 (double precision only, jobs under one wave per SIMD): MODE 1 with cooperating lanes (geff_*_cooperative).
template <typename S, int NL, int FMAX, int MODE> struct Column {
  using R = real_t<S>;
  // arithmetic policy (see dv / pwx): verification mode in double precision uses the library pow (the reference's
  // torch.pow); the plain-float fast mode divides by reciprocal; everything else is lean pow + IEEE division
  // (3: MODE 3, the mixed-precision kernels -- lean pow with pairwise-combined polynomials, see fast_pow)
  static constexpr int POL = ((MODE == 0) && (sizeof(R) == 8)) ? 1 : (((MODE != 0) && (sizeof(S) == 4)) ? 2 : ((mixed_mode(MODE) && sizeof(S) == 8) ? 3 : 0));
  static constexpr int STRIDE = coop_mode(MODE) ? LGAR_COOP_GROUPS : WAVE;  // front-table slots per row (see FrontsView)
  const ColParams<S, NL> &P;
  const LGAR_KARG Glob<R> *G;  // run-time constants, in the kernarg segment (re-pointed by the kernel's time loop)
  FrontsView<S, FMAX, STRIDE> F;
  int nf;
  int status;
  S ponded_water, previous_precip, ending_volume;
  S giuh_q[LGAR_GMAX];
  // MODE 3, one lane per column: the GIUH queue is not held in registers at all.  It is touched once per sub-step and only
  // on storm steps -- which made its 16 registers the allocator's first spill victims, and a spilled read-modify-write is a
  // scratch store every time (5.5 of the kernel's 13.5 GB of write traffic, profiles/r05/ablate_giuh_queue.jsonl).  The
  // queue is updated in place in the column's own rows of the state arrays (`scalars` rows 3.., coalesced across the lanes),
  // where it is loaded from and stored to anyway; a flag remembers whether anything is queued (giuh_live, the reference's
  // `sum(queue) > 0` test evaluated when the queue was last written, lgar/giuh.py:8-20).
  static constexpr bool GIUH_MEM = (MODE == 3) && (sizeof(S) == sizeof(R));
  // MODE 3: the accumulators of the forcing alone are added at the END of the sub-step, not where the reference adds them
  // (dpLGAR.py:185-190): nothing between the two places leaves the sub-step, the sums are the same, and two fewer values wait
  // through all of the column physics.  Same-box A/B, register allocation being what it is: the mixed-precision kernel 0.9 %
  // faster, the fp32 kernel 1.7 % slower -- so the former only.
  static constexpr bool LATE_FORCING_SUMS = (MODE == 3);
  static constexpr int DEAD = LGAR_ST_BOTTOM | LGAR_ST_OVERFLOW | LGAR_ST_STRUCT;  // forward() leaves such a column alone
  R *giuh_mem = nullptr;   // &scalars[3 * N + c]
  size_t giuh_stride = 0;  // N
  bool giuh_live = false;
  // K(theta) is not kept per front: the reference refreshes it from theta for every front but the deepest at the end
  // of each move (update_psi, Layer.py:1166-1170) and reads it only in calc_dzdt, so it is computed there, for moving
  // fronts only.  k_deepest is the one K that is never refreshed (the domain's deepest front keeps its initial K);
  // new_front_frozen marks a front created in the current sub-step, whose K carries frozen_factor (Layer.py:1410-1412).
  S k_deepest;
  bool new_front_frozen;
  // insert_water's Geff of the last call and what it was computed from (see insert_water): layer -1 = nothing remembered
  S memo_theta, memo_g;
  int memo_layer = -1;
  unsigned near_sat_fronts = 0u;  // fronts whose psi the fast modes must send through the reference's theta -> psi round trip
  // fast modes, one lane per column: the column mass as of the end of move_wetting_front (mass_and_events), still the ending
  // volume of the sub-step when no pass fired and no front was created after it.  (Cooperating lanes keep the separate walks:
  // their column mass sums a register copy of the table, and the event scan added to it costs a lone wave more in selects than
  // the scan's own walk -- configs[1] 78.4 -> 81.8 ms, one column 43.3 -> 45.9 ms, measured.)
  static constexpr bool FUSED_WALK = (MODE != 0) && !coop_mode(MODE);
  S post_mass;
  bool post_mass_valid = false;
  bool post_event = false;
  bool psi_rewritten = false;  // a dry-over-wet deletion below the top layer rewrote theta and psi of the fronts above (q13)
  S aet_psi_wp_memo;          // calc_aet's half-uptake head of this column (a function of the top layer's parameters only)
  bool aet_psi_wp_known = false;
  bool count_geff = false;              // measurement: count wave-level Geff evaluations in the unused upper bits of `status`
  int share_lanes = 0;                  // tangent kernels: W = 2..32 adjacent lanes carry this same column (other directions);
                                        // forward kernels: 2..64 = that many adjacent lanes carry this very column (small jobs)
  R *xchg = nullptr;                    // ... and the wave's LDS buffer they exchange trapezoid nodes through
  int coop_rank = 0;                    // forward kernels: my place in my group of cooperating lanes
  int cap = FMAX;                       // fronts this column may hold: min(kernel capacity, rows of the state arrays)
  // accumulators drained every forcing step (physics/MassBalance.py:45-53)
  S a_precip, a_pet, a_aet, a_infil, a_runoff, a_perc, a_giuh, a_disch;

  __device__ Column(const ColParams<S, NL> &p, const LGAR_KARG Glob<R> *g, const FrontsView<S, FMAX, STRIDE> &f) : P(p), G(g), F(f) {}

  __device__ __forceinline__ S cum_at(int k) const { return sel<S, NL>(P.cum, k); }
  // calc_geff (lgar/green_ampt.py:19-99): trapezoid or closed form, per cfg.data.use_closed_form_G
  __device__ __forceinline__ S capillary_drive(const LayerK<S> &lk, S theta1, S theta2, int site = 0, CoopRiders *riders = nullptr) {
    (void)site; (void)riders;
    LGAR_MEASURE_POINT(NOGEFF, lk, theta1, theta2)
    LGAR_COUNT_GEFF_CALL(site)
    LGAR_MEASURE_POINT(DUP_GEFF, lk, theta1, theta2)
    if constexpr (MODE == 0 && sizeof(R) == 8) {
      // verification mode: the reference's trapezoid operation by operation (4 pow + sqrt per node, running h)
      return G->closed_form ? geff_closed<S, POL>(lk, theta1, theta2) : geff_literal<S, POL>(lk, theta1, theta2, G->nint);
    }
    if constexpr (sizeof(S) != sizeof(R) && sizeof(R) == 8 && MODE != 0) {
      if (share_lanes >= 2 && !G->closed_form) return geff_fused<S>(lk, theta1, theta2, G->nint, xchg, share_lanes);
    }
    if constexpr (coop_mode(MODE) && !mixed_mode(MODE) && sizeof(S) == 8 && sizeof(R) == 8) {  // plain double: cooperating lanes (small jobs)
      if (share_lanes > 1 && !G->closed_form) return geff_fused<S>(lk, theta1, theta2, G->nint, xchg, share_lanes, coop_rank, riders);
    }
    if constexpr (MODE == 6 && sizeof(S) == 8 && sizeof(R) == 8) {  // mixed-precision trapezoid, its groups of nodes split over the lanes
      if (share_lanes > 1 && !G->closed_form)
        return geff_mixed_coop(val(lk.alpha), val(lk.n), val(lk.m), val(lk.inv_m), val(lk.inv_n), val(lk.ksat), val(lk.te), val(lk.tr),
                               theta1, theta2, G->nint, xchg, share_lanes, coop_rank);
    }
    if constexpr (mixed_mode(MODE) && sizeof(S) == 8 && sizeof(R) == 8) {  // plain double, LgarDims.geff_mode = 1
      if (!G->closed_form)
        return geff_mixed(val(lk.alpha), val(lk.n), val(lk.m), val(lk.inv_m), val(lk.inv_n), val(lk.ksat), val(lk.te), val(lk.tr),
                          theta1, theta2, G->nint);
    }
    return G->closed_form ? geff_closed<S, POL>(lk, theta1, theta2) : geff(lk, theta1, theta2, G->nint);
  }
  // calc_geff between two FRONTS of the table in the mixed-precision mode (calc_dzdt): the fronts' own psi are the trapezoid's
  // heads, and K(theta2) / Ksat of its wet end node is handed back -- the front's own conductivity (geff_mixed_heads).
  __device__ __forceinline__ double capillary_drive_fronts(const LayerK<double> &lk, double theta1, double theta2, double psi1,
                                                           double psi2, double &kr_end) {
    LGAR_MEASURE_POINT(NOGEFF_FRONTS, lk, theta1, theta2, kr_end)
    LGAR_COUNT_GEFF_CALL(1)
    if constexpr (MODE == 6) {
      if (share_lanes > 1) return geff_mixed_heads<true>(lk, theta1, theta2, psi1, psi2, G->nint, kr_end, xchg, share_lanes, coop_rank);
    }
    return geff_mixed_heads(lk, theta1, theta2, psi1, psi2, G->nint, kr_end);
  }
  __device__ __forceinline__ S cum_prev(int k) const {  // cum[k-1], 0 for k == 0
    S r = S(R(0.0));
#pragma unroll
    for (int j = 0; j < NL - 1; j++) {
      const S cj = P.cum[j];
      r = choose(k == j + 1, cj, r);
    }
    return r;
  }

  // WettingFront.is_equal (by VALUE), layers/WettingFront.py:76-84
  __device__ __forceinline__ bool feq(int a, int b) const {
    return val(F.Z(a)) == val(F.Z(b)) && val(F.PS(a)) == val(F.PS(b)) && val(F.DZ(a)) == val(F.DZ(b));
  }
  __device__ __forceinline__ void fdel(int i) {
    for (int j = i; j < nf - 1; j++) F.copy(j, j + 1);
    nf--;
  }
  // first flat index of layer k and the number of fronts tagged k
  __device__ __forceinline__ void range_of(int k, int &lo, int &len) const {
    lo = -1; len = 0;
    for (int i = 0; i < nf; i++)
      if (F.layer(i) == k) { if (len == 0) lo = i; len++; }
  }

  // Cooperating lanes (one wave alone on its SIMD): every dependent LDS read is ~130 cycles nothing else covers, and the column
  // mass reads the front table row by row.  A table of at most SCAN fronts (the usual case) is fetched in ONE round trip
  // instead, and the sum runs on registers with the row index a compile-time constant: the same terms in the same order, so the
  // same result bit for bit.  (The event scan and the free-drainage search gained nothing from the same treatment: what they
  // save in waits they spend on selects.)
  static constexpr bool COOP = coop_mode(MODE) && (sizeof(S) == 8) && (sizeof(R) == 8);
  static constexpr int SCAN = 8;
  struct Rows {
    R z[SCAN], th[SCAN];
    int fl[SCAN];
  };
  __device__ __forceinline__ void fetch_rows(Rows &r) const {
#pragma unroll
    for (int q = 0; q < SCAN; q++) {
      r.z[q] = val(F.Z(q));
      r.th[q] = val(F.TH(q));
      r.fl[q] = F.flag(q);
    }
  }

  // Layer.mass_balance, layers/Layer.py:795-824 (layer sums combined s0 + (s1 + (s2 ...)))
  __device__ __forceinline__ S mass_balance() const {
    S ls[NL];
    if constexpr (COOP) {
      if (nf <= SCAN) {  // cooperating lanes: the table's first rows in one fetch (see Rows)
        Rows r;
        fetch_rows(r);
#pragma unroll
        for (int j = 0; j < NL; j++) ls[j] = S(R(0.0));
#pragma unroll
        for (int q = 0; q < SCAN; q++) {
          const bool on = q < nf;
          const int lay = r.fl[q] & 0x7f;
          const bool next_same = (q + 1 < nf) && ((r.fl[(q + 1 < SCAN) ? q + 1 : q] & 0x7f) == lay);
          const R dth = next_same ? r.th[q] - r.th[(q + 1 < SCAN) ? q + 1 : q] : r.th[q];
          R base = R(0.0);
#pragma unroll
          for (int j = 1; j < NL; j++) base = choose(lay == j, val(P.cum[j]) - val(P.thick[j]), base);
          const R term = (r.z[q] - base) * dth;
#pragma unroll
          for (int j = 0; j < NL; j++) ls[j] = choose(on && lay == j, ls[j] + term, ls[j]);  // (a layer's fronts are adjacent:
        }                                                                                  //  each sum grows in table order)
        S tot = ls[NL - 1];
#pragma unroll
        for (int j = NL - 2; j >= 0; j--) tot = ls[j] + tot;
        return tot;
      }
    }
    int i = 0;
#pragma unroll
    for (int j = 0; j < NL; j++) {
      S base = (j == 0) ? S(R(0.0)) : P.cum[j] - P.thick[j];
      S sum = S(R(0.0));
      while (i < nf && F.layer(i) == j) {
        bool next_same = (i + 1 < nf) && (F.layer(i + 1) == j);
        S dth = next_same ? (F.TH(i) - F.TH(i + 1)) : F.TH(i);
        sum = sum + (F.Z(i) - base) * dth;
        i++;
      }
      ls[j] = sum;
    }
    S tot = ls[NL - 1];
#pragma unroll
    for (int j = NL - 2; j >= 0; j--) tot = ls[j] + tot;
    return tot;
  }

  // Layer.mass_balance AND the triggers of the post-sweep passes (front_event_pending) in ONE walk over the front table: both
  // read every front's depth, theta and tag, and between the sweep and the end of a sub-step nothing else changes them unless a
  // pass fires or a surficial front is created -- so the fast modes walk the table once per sub-step instead of three times
  // (column-mass check, event scan, ending volume).  The mass is mass_balance()'s, term by term in the same order.
  __device__ __forceinline__ S mass_and_events(bool &ev) const {
    S ls[NL];
    bool e = false;
    int i = 0;
#pragma unroll
    for (int j = 0; j < NL; j++) {
      S base = (j == 0) ? S(R(0.0)) : P.cum[j] - P.thick[j];
      S sum = S(R(0.0));
      while (i < nf && F.layer(i) == j) {
        const bool has_next = i + 1 < nf;
        const int fn = has_next ? F.flag(i + 1) : 0;
        const bool next_same = has_next && ((fn & 0x7f) == j);
        const S zi = F.Z(i), ti = F.TH(i);
        if (has_next) {
          const S zn = F.Z(i + 1), tn = F.TH(i + 1);
          e = e || (next_same && !(fn & LGAR_FLAG_BOTTOM) && val(zi) > val(zn));
          e = e || (next_same && val(ti) <= val(tn));
          e = e || (val(zi) > val(P.cum[j]));
          sum = sum + (zi - base) * (next_same ? (ti - tn) : ti);
        } else {
          sum = sum + (zi - base) * ti;
        }
        i++;
      }
      ls[j] = sum;
    }
    S tot = ls[NL - 1];
#pragma unroll
    for (int j = NL - 2; j >= 0; j--) tot = ls[j] + tot;
    ev = e;
    return tot;
  }

  // calc_wetting_front_free_drainage, Layer.py:134-162: argmin psi, ties (and isclose) go deeper
  __device__ __forceinline__ int free_drainage_front() const {
    R psi = val(F.PS(0));
    int idx = 0;
    for (int i = 0; i < nf; i++) {
      R pi = val(F.PS(i));
      if (pi <= psi) { psi = pi; idx = i; }
      else if (ab(pi - psi) <= R(1e-8) + R(1e-5) * ab(psi)) { psi = pi; idx = i; }
    }
    return idx;
  }

  // theta(psi) and d theta / d psi of one layer in plain reals (no extra pow for the slope:
  // theta' = -(theta - theta_r) m n (a psi)^n / (psi (1 + (a psi)^n)))
  __device__ __forceinline__ static void theta_slope(R alpha, R n, R m, R te, R tr, R psi, R &th, R &dth) {
    R ap = pwp<POL>(alpha * psi, n);
    R one_ap = R(1.0) + ap;
    R op = pwp<POL>(one_ap, m);
    R span = dv<POL>(R(1.0), op) * (te - tr);
    th = span + tr;
    dth = (psi > R(0.0)) ? dv<POL>(-(span * m * n * ap), psi * one_ap) : R(0.0);
  }

  // ... and the second derivative (the double-precision searches fit a power law through value, slope and curvature):
  // theta'' = theta' ((n - 1) - n (a psi)^n) / (psi (1 + (a psi)^n)).  theta is theta_from_h's operation by operation.
  __device__ __forceinline__ static void theta_slope2(R alpha, R n, R m, R te, R tr, R psi, R &th, R &dth, R &d2th) {
    R ap = pwp<POL>(alpha * psi, n);
    R one_ap = R(1.0) + ap;
    R op = pwp<POL>(one_ap, m);
    R span = dv<POL>(R(1.0), op) * (te - tr);
    th = span + tr;
    const R r = (psi > R(0.0)) ? dv<POL>(R(1.0), psi * one_ap) : R(0.0);
    dth = -(span * m * n * ap) * r;
    d2th = dth * ((n - R(1.0)) - n * ap) * r;
  }

  // search_mode 1: the same root -- psi with |sum_j thick_j (theta_j(psi) - dtheta_j) - prior_mass| <= tolerance --
  // found by a bracketed iteration on the mass and its derivatives (3-5 mass evaluations) instead of the reference's
  // fixed-step decimal search (58-82 on average, Layer.py:275-317).  theta differs from the literal search by
  // <= tolerance / thickness.
  // Gradient semantics (dual numbers): psi_final = psi_init + constant, as in the reference (Layer.py:277-288).
  // Cooperating lanes (MODE 4: the lanes of a group carry the SAME column): a mass evaluation is K + 1 independent
  // theta(psi) -- two pows each -- so lane r of the group evaluates layer min(r, K) with that layer's parameters as its
  // operands (ONE instruction stream) and the group exchanges the results through its LDS table; the sums are formed from
  // them in the serial order.  Every value goes through exactly the operations of the serial evaluation: bit-identical.
  // FirstEval: what the sweep evaluated BEFORE a search (sweep_layer; coop_sweep_thetas for cooperating lanes): theta and its
  // two derivatives for the layers above and the front's own layer at the front's psi -- the search's first mass evaluation
  struct FirstEval {
    bool have = false;
    R M = R(0.0), dM = R(0.0), d2M = R(0.0);  // the mass sum of the search (layers above first, in order, then the own layer)
    R thk = R(0.0);                           // theta of the front's own layer at its psi
    __device__ __forceinline__ void add(R thick, R th, R below, R dth, R d2th) {
      M += thick * (th - below);
      dM += thick * dth;
      d2M += thick * d2th;
    }
  };

  // Single precision keeps the plain bracketed Newton iteration (its pow is three instructions; the kernel is not bound by
  // this search).
  template <int K>
  __device__ __forceinline__ S theta_mass_balance_newton_f32(const LayerK<S> &lk, S psi0, S new_mass, S prior_mass,
                                                             const S (&dth)[NL], const S (&dthick)[NL], S dth_k, S dthick_k) {
    const R prior = val(prior_mass);
    R psi = val(psi0);
    R f = val(new_mass) - prior;
    if (ab(f) <= Tol<R>::mass) return theta_from_h<S, POL>(lk, psi0);
    R M = R(0.0), dM = R(0.0);
    auto eval = [&](R x) {
      R thk, dthk;
      theta_slope(val(lk.alpha), val(lk.n), val(lk.m), val(lk.te), val(lk.tr), x, thk, dthk);
      M = val(dthick_k) * (thk - val(dth_k));
      dM = val(dthick_k) * dthk;
#pragma unroll
      for (int j = 0; j < K; j++) {
        R t1, d1;
        theta_slope(val(P.alpha[j]), val(P.n[j]), val(P.m[j]), val(P.te[j]), val(P.tr[j]), x, t1, d1);
        M += val(dthick[j]) * (t1 - val(dth[j]));
        dM += val(dthick[j]) * d1;
      }
    };
    R lo = R(0.0), hi = R(-1.0);  // bracket f(lo) > 0 > f(hi); hi < 0: not found yet
    bool lo_ok = false;
    eval(psi);
    f = M - prior;
    for (int it = 0; it < 64; it++) {
      if (ab(f) <= Tol<R>::mass) break;
      if (f > R(0.0)) {
        lo = psi;
        lo_ok = true;
      } else {
        hi = psi;
        if (!lo_ok) {
          // is the target reachable at all?  mass at psi = 0 (saturation) needs no pow
          R M0 = val(dthick_k) * (((val(lk.te) - val(lk.tr)) + val(lk.tr)) - val(dth_k));
#pragma unroll
          for (int j = 0; j < K; j++) M0 += val(dthick[j]) * (((val(P.te[j]) - val(P.tr[j])) + val(P.tr[j])) - val(dth[j]));
          if (M0 - prior <= Tol<R>::mass) { psi = R(0.0); break; }  // saturated: the reference walks psi -> 0 (Layer.py:287-316)
          lo_ok = true;
        }
      }
      R pn = (dM < R(0.0)) ? psi - dv<POL>(f, dM) : R(-1.0);
      const bool inside = (pn > lo) && (hi < R(0.0) || pn < hi);
      if (!inside) pn = (hi >= R(0.0)) ? R(0.5) * (lo + hi) : psi * R(2.0) + R(1.0);
      if (pn == psi) break;  // step below resolution
      psi = pn;
      eval(psi);
      f = M - prior;
      if (it == 63) status |= LGAR_ST_ITERCAP;
    }
    const S psi_final = psi0 + (psi - val(psi0));
    return theta_from_h<S, POL>(lk, psi_final);
  }

  // Double precision (plain and dual numbers): the column mass M(psi) is, layer by layer, close to a shifted power law of psi
  // -- theta - theta_e ~ -(alpha psi)^n towards saturation, theta - theta_r ~ (alpha psi)^(1-n) in dry soil -- so every step
  // fits M ~ A - C psi^p through the value, slope and curvature of the present iterate (p = 1 + psi M''/M') and solves that
  // model for the target: psi_next = psi (1 - p f / (psi M'))^(1/p).  Third order like Halley's method, exact for a power law:
  // 3 mass evaluations where Newton took 4-5 (f: 1e-2 -> 1e-6 -> 1e-16), and 4-5 where Newton, started from a front that had
  // just crossed a layer boundary all but saturated (psi ~ 1e-8, M' ~ 0), overshot by six decades and bisected its way back
  // in 12-16 -- a wavefront waits for its slowest lane, so that tail was most of the search's wave-level cost.  The step itself
  // costs no double-precision pow: a long step (|t| > 1/64, t = f / (psi M')) takes ratio^(1/p) from the hardware's single-
  // precision log2 / exp2 (an iterate need not be better than the model), a short one its series in t to third order
  // (psi_next good to ~t^4: the final steps, t ~ 1e-5, lose nothing).  The bracket (lo, hi) and the bisection fallback are
  // the Newton version's; so are the termination test -- the true mass, in double precision, within the reference's tolerance
  // of the target -- and the saturated exit.
  template <int K>
  __device__ __forceinline__ S theta_mass_balance_newton(const LayerK<S> &lk, S psi0, S new_mass, S prior_mass,
                                                         const S (&dth)[NL], const S (&dthick)[NL], S dth_k, S dthick_k,
                                                         const FirstEval &fe) {
    if constexpr (sizeof(R) == 4) {
      (void)fe;
      return theta_mass_balance_newton_f32<K>(lk, psi0, new_mass, prior_mass, dth, dthick, dth_k, dthick_k);
    } else {
    const R prior = val(prior_mass);
    R psi = val(psi0);
    R f = val(new_mass) - prior;
    if (ab(f) <= Tol<R>::mass) return theta_from_h<S, POL>(lk, psi0);
    R M = R(0.0), dM = R(0.0), d2M = R(0.0);
    // my share of a mass evaluation (cooperating lanes): layer c_q's parameters
    R c_al = val(lk.alpha), c_n = val(lk.n), c_m = val(lk.m), c_te = val(lk.te), c_tr = val(lk.tr);
    int c_q = K;
    bool together = false;
    R last_x = R(-1.0), last_th = R(0.0);  // the own layer's theta of the latest mass evaluation, and its psi
    if constexpr (COOP && K > 0) {
      together = share_lanes > K;
      c_q = coop_rank < K ? coop_rank : K;
#pragma unroll
      for (int j = 0; j < K; j++) {
        const bool mine = c_q == j;
        c_al = choose(mine, val(P.alpha[j]), c_al); c_n = choose(mine, val(P.n[j]), c_n); c_m = choose(mine, val(P.m[j]), c_m);
        c_te = choose(mine, val(P.te[j]), c_te); c_tr = choose(mine, val(P.tr[j]), c_tr);
      }
    }
    auto sums = [&](R thk, R dthk, R d2thk, const R (&tj)[NL], const R (&dj)[NL], const R (&ej)[NL]) {
      M = val(dthick_k) * (thk - val(dth_k));
      dM = val(dthick_k) * dthk;
      d2M = val(dthick_k) * d2thk;
#pragma unroll
      for (int j = 0; j < K; j++) {
        M += val(dthick[j]) * (tj[j] - val(dth[j]));
        dM += val(dthick[j]) * dj[j];
        d2M += val(dthick[j]) * ej[j];
      }
    };
    auto eval = [&](R x) {
      R thk, dthk, d2thk;
      if constexpr (COOP && K > 0) {
        if (together) {
          R th, dt, d2, tj[NL], dj[NL], ej[NL];
          theta_slope2(c_al, c_n, c_m, c_te, c_tr, x, th, dt, d2);
          xchg[3 * c_q] = th;  // (lanes with the same c_q store the same value to the same address)
          xchg[3 * c_q + 1] = dt;
          xchg[3 * c_q + 2] = d2;
          lds_exchange_point();
          thk = xchg[3 * K];
          dthk = xchg[3 * K + 1];
          d2thk = xchg[3 * K + 2];
#pragma unroll
          for (int j = 0; j < K; j++) { tj[j] = xchg[3 * j]; dj[j] = xchg[3 * j + 1]; ej[j] = xchg[3 * j + 2]; }
          lds_exchange_point();
          sums(thk, dthk, d2thk, tj, dj, ej);
          last_x = x;
          last_th = thk;
          return;
        }
      }
      theta_slope2(val(lk.alpha), val(lk.n), val(lk.m), val(lk.te), val(lk.tr), x, thk, dthk, d2thk);
      M = val(dthick_k) * (thk - val(dth_k));
      dM = val(dthick_k) * dthk;
      d2M = val(dthick_k) * d2thk;
#pragma unroll
      for (int j = 0; j < K; j++) {
        R t1, d1, e1;
        theta_slope2(val(P.alpha[j]), val(P.n[j]), val(P.m[j]), val(P.te[j]), val(P.tr[j]), x, t1, d1, e1);
        M += val(dthick[j]) * (t1 - val(dth[j]));
        dM += val(dthick[j]) * d1;
        d2M += val(dthick[j]) * e1;
      }
      last_x = x;
      last_th = thk;
    };
    R lo = R(0.0), hi = R(-1.0);  // bracket f(lo) > 0 > f(hi); hi < 0: not found yet
    bool lo_ok = false;
    if (fe.have) {  // evaluated by the sweep, with its own thetas
      M = fe.M; dM = fe.dM; d2M = fe.d2M;
      last_x = psi;
      last_th = fe.thk;
    } else {
      eval(psi);
    }
    f = M - prior;
    const R theta_sat = (val(lk.te) - val(lk.tr)) + val(lk.tr);  // theta_from_h at psi = 0, operation by operation
    for (int it = 0; it < 64; it++) {
      if (ab(f) <= Tol<R>::mass) break;
      if (f > R(0.0)) {
        lo = psi;
        lo_ok = true;
      } else {
        hi = psi;
        if (!lo_ok) {
          // is the target reachable at all?  mass at psi = 0 (saturation) needs no pow
          R M0 = val(dthick_k) * (theta_sat - val(dth_k));
#pragma unroll
          for (int j = 0; j < K; j++) M0 += val(dthick[j]) * (((val(P.te[j]) - val(P.tr[j])) + val(P.tr[j])) - val(dth[j]));
          if (M0 - prior <= Tol<R>::mass) {  // saturated: the reference walks psi -> 0 (Layer.py:287-316)
            psi = R(0.0);
            last_x = R(0.0);
            last_th = theta_sat;
            break;
          }
          lo_ok = true;
        }
      }
      R pn = R(-1.0);
      if (dM < R(0.0) && psi > R(0.0)) {
        const R ipd = dv<POL>(R(1.0), psi * dM);
        const R t = f * ipd;                                  // the Newton step, relative to psi (negated)
        const R p = R(1.0) + (psi * psi) * d2M * ipd;         // exponent of the power law through (M, M', M'')
        const R ratio = R(1.0) - p * t;
        if (ab(t) <= R(0.015625)) {
          // psi (ratio^(1/p) - 1) = psi (-t + (1 - p) t^2 / 2 - (2 p - 1)(p - 1) t^3 / 6 + O(t^4))
          const R c2 = R(0.5) * (R(1.0) - p), c3 = (R(2.0) * p - R(1.0)) * (p - R(1.0)) * R(1.0 / 6.0);
          pn = psi - (psi * t) * (R(1.0) - t * (c2 - c3 * t));
        } else if (ratio > R(0.0)) {
          const float pf = (float)p, tf = (float)t;
          const float e = (fabsf(pf) > 1e-4f) ? lg2((float)ratio) * rcp32(pf) : -1.44269504f * tf;  // p -> 0: M ~ A - C log psi
          pn = psi * (R)ex2(e);
        }
      }
      const bool inside = (pn > lo) && (hi < R(0.0) || pn < hi);
      if (!inside) pn = (hi >= R(0.0)) ? R(0.5) * (lo + hi) : psi * R(2.0) + R(1.0);
      if (pn == psi) break;  // step below resolution
      psi = pn;
      eval(psi);
      f = M - prior;
      if (it == 63) status |= LGAR_ST_ITERCAP;
    }
    if constexpr (sizeof(S) == sizeof(R)) {
      // plain reals: theta(psi) was part of the last mass evaluation (theta_slope2's theta is theta_from_h's)
      if (psi == last_x) return S(last_th);
      return theta_from_h<S, POL>(lk, S(psi));
    } else {
      // dual numbers: psi_final = psi_init + constant, as in the reference's autograd (Layer.py:277-288)
      const S psi_final = psi0 + (psi - val(psi0));
      return theta_from_h<S, POL>(lk, psi_final);
    }
    }
  }

  // theta_mass_balance, Layer.py:242-318 (+ recalculate_mass :211-240).  k = the front's layer;
  // dth/dthick hold the entries of the layers above (j < K), dth_k/dthick_k the front's own.
  template <int K>
  __device__ __forceinline__ S theta_mass_balance(const LayerK<S> &lk, S psi, S new_mass, S prior_mass, const S (&dth)[NL],
                                  const S (&dthick)[NL], S dth_k, S dthick_k, const FirstEval &fe = FirstEval()) {
    R delta_mass = ab(val(new_mass) - val(prior_mass));
    bool switched = false;
    R factor = R(1.0);
    S theta = S(R(0.0));
    S psi_prev = psi;  // a tensor in the reference: psi_prev * 0.1 below carries its gradient
    R delta_mass_prev = delta_mass;
    int count_no_change = 0;
    if (delta_mass <= Tol<R>::mass) {
      if constexpr (sizeof(S) == sizeof(R)) {
        if (fe.have) return S(fe.thk);  // theta_from_h(lk, psi), evaluated by the sweep
      }
      return theta_from_h<S, POL>(lk, psi);
    }
    LGAR_MEASURE_POINT(NOSEARCH, lk, psi, new_mass)
    if constexpr (MODE != 0) return theta_mass_balance_newton<K>(lk, psi, new_mass, prior_mass, dth, dthick, dth_k, dthick_k, fe);
    long long it = 0;
    while (delta_mass > Tol<R>::mass) {
      if (++it > G->iter_cap) { status |= LGAR_ST_ITERCAP; break; }
      if (val(new_mass) > val(prior_mass)) {
        psi = psi + (R(0.1) * factor);
        switched = false;
      } else {
        if (!switched) { switched = true; factor = factor * R(0.1); }
        psi_prev = psi;
        psi = psi - (R(0.1) * factor);
        if (val(psi) < R(0.0) && val(psi_prev) != R(0.0)) psi = psi_prev * R(0.1);
      }
      theta = theta_from_h<S, POL>(lk, psi);
      S mass = S(R(0.0));
      mass = mass + (dthick_k * (theta - dth_k));
#pragma unroll
      for (int j = 0; j < K; j++) mass = mass + dthick[j] * (theta_from_h<S, POL>(pick_static(P, j), psi) - dth[j]);
      new_mass = mass;
      delta_mass = ab(val(new_mass) - val(prior_mass));
      if (ab(val(psi) - val(psi_prev)) < Tol<R>::nochange && factor < R(1e-13)) break;
      if (ab(delta_mass - delta_mass_prev) < Tol<R>::nochange) count_no_change++; else count_no_change = 0;
      if (count_no_change == 5) break;
      if (val(psi) <= R(0.0) && val(psi_prev) < Tol<R>::tiny) break;
      delta_mass_prev = delta_mass;
    }
    return theta;
  }

  // check_column_mass, Layer.py:655-701: depth line search of the (saturated) free-drainage front
  __device__ __forceinline__ void check_column_mass(int fdd, S old_mass, S percolation, S aet) {
    S theta_e_k1 = sel<S, NL>(P.te, F.layer(fdd));
    S mass_timestep = (old_mass + percolation) - (aet + R(0.0));
    if (__builtin_expect(ab(val(F.TH(fdd)) - val(theta_e_k1)) < Tol<R>::mass, 0)) {
      S current_mass = mass_balance();
      R err = ab(val(current_mass) - val(mass_timestep));
      bool switched = false;
      R factor = R(1.0);
      S depth_new = F.Z(fdd);
      // search_mode 1: the column mass is LINEAR in this one depth (slope = theta_fdd - theta_next, or theta_fdd for
      // the last front of a layer).  (i) All but the last two fixed steps of each up/down run are taken in one jump;
      // (ii) inside the loop the mass is evaluated from the linear model instead of re-summing the front table; when
      // the model says "converged" the true mass is summed once and, if the reference's termination test does not hold
      // on it, the reference's loop simply continues on true sums.
      const bool nxt_same = (fdd + 1 < nf) && (F.layer(fdd + 1) == F.layer(fdd));
      const S slope_s = nxt_same ? F.TH(fdd) - F.TH(fdd + 1) : F.TH(fdd);
      const R slope = val(slope_s);
      const bool jump = (MODE != 0) && (slope > R(0.0));
      bool model = jump;
      const S m0 = current_mass, d0 = depth_new;
      long long it = 0;
      if constexpr (sizeof(R) == 8) {
        if (jump && ab(err - Tol<R>::mass) > Tol<R>::mass) {
          // the mass is linear in this depth: one step to the root (an offset that is a constant w.r.t. the parameters, like the
          // reference's fixed steps), verified on the true mass; the reference's own loop takes over if its test does not hold
          depth_new = depth_new + dv<POL>(val(mass_timestep) - val(current_mass), slope);
          F.Z(fdd) = depth_new;
          current_mass = mass_balance();
          err = ab(val(current_mass) - val(mass_timestep));
          model = false;
        }
      }
      while (true) {
        if (!(ab(err - Tol<R>::mass) > Tol<R>::mass)) {
          if (!model) break;
          F.Z(fdd) = depth_new;  // verify on the true mass; from here on the loop is the reference's own
          current_mass = mass_balance();
          err = ab(val(current_mass) - val(mass_timestep));
          model = false;
          continue;
        }
        if (++it > G->iter_cap) { status |= LGAR_ST_ITERCAP; break; }
        R before = val(depth_new);
        if (val(current_mass) < val(mass_timestep)) {
          if (jump) {
            R nsteps = (val(mass_timestep) - R(2.0) * Tol<R>::mass - val(current_mass)) / (slope * R(0.01) * factor);
            if (nsteps > R(3.0)) depth_new = depth_new + (R(floor(nsteps)) - R(2.0)) * (R(0.01) * factor);
          }
          depth_new = depth_new + R(0.01) * factor;
          switched = false;
        } else {
          if (!switched) { switched = true; factor = factor * R(0.001); }
          if (jump) {
            R nsteps = (val(current_mass) - val(mass_timestep) - R(2.0) * Tol<R>::mass) / (slope * R(0.01) * factor);
            if (nsteps > R(3.0)) depth_new = depth_new - (R(floor(nsteps)) - R(2.0)) * (R(0.01) * factor);
          }
          depth_new = depth_new - (R(0.01) * factor);
        }
        if (sizeof(R) == 4 && val(depth_new) == before && factor < R(1e-6)) break;  // fp32: step below resolution
        if (model) {
          current_mass = m0 + slope_s * (depth_new - d0);
        } else {
          F.Z(fdd) = depth_new;
          current_mass = mass_balance();
        }
        err = ab(val(current_mass) - val(mass_timestep));
      }
      if (model) F.Z(fdd) = depth_new;  // left the loop (cap / resolution) while still on the model
    }
  }

  // Layer.move_wetting_fronts (Layer.py:1254-1307) with its three cases: deepest_layer_front (:389-418),
  // wetting_front_in_layer (:420-547, compute_wetting_front_mass :561-644) and base_case (:320-387,
  // populate_delta_thickness :177-209).  The reference snapshots every front first (copy_states); the
  // sweep runs deepest -> shallowest and front i only needs the pre-sweep values of i and i+1, so the
  // snapshot is two register triples carried down the sweep instead of a second front table.
  // The sweep is LAYER-MAJOR with the layer index a compile-time constant: for K = NL-1 .. 0 the lane walks up its
  // fronts tagged K.  Per-layer parameters are then plain register operands (no value-selects per front), the
  // "layers above" loops have static bounds, and layer 0's cheap closed-form update carries no search code.
  struct SweepCarry {
    int i, nf0, fdd;
    unsigned near_sat;  // bit i: front i took its psi from the front below with (alpha psi)^n < 1e-6 (see psi_round_trip)
    S infiltration, aet;
    S on_z, on_th, on_ps;  // pre-sweep values of front i+1
  };

  // Cooperating lanes: the 6 K thetas an in-layer front of layer K needs before its search (theta of every layer above at
  // the front's and the next front's psi, before and after the move -- compute_wetting_front_mass, Layer.py:561-644) are three
  // distinct evaluations per layer above (the front's own psi has not changed yet: "old" and "new" are one value), and the
  // search's first mass evaluation needs K + 1 more at the front's psi.  Lane r of the group evaluates item r (items
  // 3 j + {0: psi, 1: psi of the next front before the sweep, 2: after}; item 3 K: the front's own layer at psi), in rounds of
  // `share_lanes` items; the serial chain of 2 (7 K + 1) pows becomes one of 2 per round.  theta_slope's theta is
  // theta_from_h's operation by operation.
  template <int K>
  __device__ __forceinline__ void coop_sweep_thetas(const LayerK<S> &lk, R psi, R psi_below_old, R psi_below, R (&th)[NL], R (&dt)[NL],
                                                    R (&d2)[NL], R &thk, R &dthk, R &d2thk, R (&th_below_old)[NL], R (&th_below)[NL]) {
    constexpr int CNT = 3 * K + 1;
    for (int first = 0; first < CNT; first += share_lanes) {
      const int q = first + coop_rank;
      const bool mine = (coop_rank < share_lanes) && (q < CNT);
      R al = val(lk.alpha), n = val(lk.n), m = val(lk.m), te = val(lk.te), tr = val(lk.tr), x = psi;
#pragma unroll
      for (int j = 0; j < K; j++) {
        const bool lj = (q >= 3 * j) && (q < 3 * j + 3);
        al = choose(lj, val(P.alpha[j]), al); n = choose(lj, val(P.n[j]), n); m = choose(lj, val(P.m[j]), m);
        te = choose(lj, val(P.te[j]), te); tr = choose(lj, val(P.tr[j]), tr);
        x = choose(q == 3 * j + 1, psi_below_old, x);
        x = choose(q == 3 * j + 2, psi_below, x);
      }
      R t0, t1, t2;
      theta_slope2(al, n, m, te, tr, x, t0, t1, t2);
      if (mine) { xchg[3 * q] = t0; xchg[3 * q + 1] = t1; xchg[3 * q + 2] = t2; }
    }
    lds_exchange_point();
#pragma unroll
    for (int j = 0; j < K; j++) {
      th[j] = xchg[3 * (3 * j)];
      dt[j] = xchg[3 * (3 * j) + 1];
      d2[j] = xchg[3 * (3 * j) + 2];
      th_below_old[j] = xchg[3 * (3 * j + 1)];
      th_below[j] = xchg[3 * (3 * j + 2)];
    }
    thk = xchg[3 * (3 * K)];
    dthk = xchg[3 * (3 * K) + 1];
    d2thk = xchg[3 * (3 * K) + 2];
    lds_exchange_point();
  }

  template <int K> __device__ __forceinline__ void sweep_layer(SweepCarry &c) {
    const LayerK<S> lk = pick_static(P, K);
    const int last = c.i;  // deepest front of this layer's list (if the lane has fronts tagged K)
    while (c.i >= 0 && F.layer(c.i) == K) {
      LGAR_MEASURE_POINT(CLK, 19)
      const int i = c.i;
      const S oc_z = F.Z(i), oc_th = F.TH(i), oc_ps = F.PS(i);  // pre-sweep values of front i
      bool need_psi = false;
      if (i < c.nf0 - 1) {
        if (i == last || feq(i, last)) {
          // deepest front of a layer: psi continuity with the layer below
          // (fast modes: a boundary front whose psi IS the psi below, bit for bit, took its theta from that very psi the last
          // time round -- nothing to do; decided per lane, the evaluation skipped when no column of the wavefront needs it.  The
          // boundary above an untouched layer stays like that for the whole run.)
          const bool fresh = (MODE != 0) && same_bits(F.PS(i), F.PS(i + 1));
          if (!fresh) {
            S ap;
            F.TH(i) = theta_from_h_ap<S, POL>(lk, F.PS(i + 1), ap);
            F.PS(i) = F.PS(i + 1);
            if constexpr (MODE != 0 && sizeof(R) == 8) {
              if (__builtin_expect(val(ap) < R(1e-6), 0)) c.near_sat |= 1u << i;
            }
          }
          LGAR_MEASURE_POINT(CLK, 20)
        } else if constexpr (K == 0) {
          S prior_mass = oc_z * (oc_th - c.on_th);
          if (i == c.fdd || feq(c.fdd, i)) prior_mass = prior_mass + (c.infiltration - (R(0.0) + c.aet));
          S z = F.Z(i) + (F.DZ(i) * G->dt_h);
          z = mn(z, P.cum[NL - 1]);
          F.Z(i) = z;
          bool zero_dzdt = ab(val(F.DZ(i))) <= R(1e-8);  // torch.isclose(dzdt, 0, rtol=1e-8): atol 1e-8
          if (!(zero_dzdt && !F.bottom(i))) {            // a just-created front keeps its theta (Layer.py:458-467)
            S potential = dv<POL>(prior_mass, z) + F.TH(i + 1);
            F.TH(i) = mn(lk.te, potential);
          }
          need_psi = true;
        } else {
          S dth[NL], dthick[NL];
          const S prev_thick = P.cum[(K > 0) ? K - 1 : 0];
          S z = F.Z(i) + (F.DZ(i) * G->dt_h);
          F.Z(i) = z;
          S psi_old = oc_ps, psi_below_old = c.on_ps;
          S psi = F.PS(i), psi_below = F.PS(i + 1);
          S prior_mass = (oc_z - prev_thick) * (oc_th - c.on_th);
          S new_mass = (z - prev_thick) * (F.TH(i) - F.TH(i + 1));
          const S t_dth_k = F.TH(i + 1);
          const S t_dthick_k = z - prev_thick;
          FirstEval fe;
          if constexpr (COOP) {
            if (share_lanes >= 4) {  // cooperating lanes: the thetas below, evaluated by the group together
              R tj[NL], dj[NL], ej[NL], tk, dk, ek, tbo[NL], tb[NL];
              coop_sweep_thetas<K>(lk, psi, psi_below_old, psi_below, tj, dj, ej, tk, dk, ek, tbo, tb);
#pragma unroll
              for (int j = 0; j < K; j++) {
                S lt = P.cum[j] - R(0.0);
                prior_mass = prior_mass + (lt * (tj[j] - tbo[j]));  // (psi_old is psi: the front's own psi has not moved yet)
                new_mass = new_mass + (lt * (tj[j] - tb[j]));
                dth[j] = tb[j];
                dthick[j] = lt;
                fe.add(val(lt), tj[j], tb[j], dj[j], ej[j]);
              }
              fe.add(val(t_dthick_k), tk, val(t_dth_k), dk, ek);
              fe.thk = tk;
              fe.have = true;
            }
          }
          if constexpr (sizeof(R) == 8) {
            if (!fe.have) {
              // The front's own psi has not moved yet (psi_old IS psi: theta of the layers above "before" and "after" are one
              // evaluation), and the psi of the front below is the same before and after the sweep unless that front moved --
              // for the active front of a storm it is the layer's untouched boundary front -- so theta_below is evaluated a
              // second time only when some column of the wavefront needs it.  Plain reals take theta at psi together with its
              // two derivatives: the search's first mass evaluation (FirstEval).
              const bool below_moved = any_lane(!same_bits(psi_below_old, psi_below)) != 0ull;
#pragma unroll
              for (int j = 0; j < K; j++) {
                const LayerK<S> lj = pick_static(P, j);
                const S theta_below_old = theta_from_h<S, POL>(lj, psi_below_old);
                const S theta_below = below_moved ? theta_from_h<S, POL>(lj, psi_below) : theta_below_old;
                S lt = P.cum[j] - R(0.0);  // quirk: cumulative thickness (Layer.py:603-604)
                S theta;
                if constexpr (sizeof(S) == sizeof(R)) {
                  R th, dt, d2;
                  theta_slope2(val(lj.alpha), val(lj.n), val(lj.m), val(lj.te), val(lj.tr), val(psi), th, dt, d2);
                  fe.add(val(lt), th, val(theta_below), dt, d2);
                  theta = S(th);
                } else {
                  theta = theta_from_h<S, POL>(lj, psi);
                }
                prior_mass = prior_mass + (lt * (theta - theta_below_old));
                new_mass = new_mass + (lt * (theta - theta_below));
                dth[j] = theta_below;
                dthick[j] = lt;
              }
              if constexpr (sizeof(S) == sizeof(R)) {
                R tk, dk, ek;
                theta_slope2(val(lk.alpha), val(lk.n), val(lk.m), val(lk.te), val(lk.tr), val(psi), tk, dk, ek);
                fe.add(val(t_dthick_k), tk, val(t_dth_k), dk, ek);
                fe.thk = tk;
                fe.have = true;
              }
            }
          } else {
#pragma unroll
            for (int j = 0; j < K; j++) {
              const LayerK<S> lj = pick_static(P, j);
              S theta_old = theta_from_h<S, POL>(lj, psi_old);
              S theta_below_old = theta_from_h<S, POL>(lj, psi_below_old);
              S lt = P.cum[j] - R(0.0);  // quirk: cumulative thickness (Layer.py:603-604)
              prior_mass = prior_mass + (lt * (theta_old - theta_below_old));
              S theta = theta_from_h<S, POL>(lj, psi);
              S theta_below = theta_from_h<S, POL>(lj, psi_below);
              new_mass = new_mass + (lt * (theta - theta_below));
              dth[j] = theta_below;
              dthick[j] = lt;
            }
          }
          if (i == c.fdd || feq(c.fdd, i)) prior_mass = prior_mass + c.infiltration - (R(0.0) + c.aet);
          LGAR_MEASURE_POINT(CLK, 22)
          LGAR_MEASURE_POINT(DUP_SEARCH, K, lk, psi, new_mass, prior_mass, dth, dthick, t_dth_k, t_dthick_k)
          S theta_new = theta_mass_balance<K>(lk, psi, new_mass, prior_mass, dth, dthick, t_dth_k, t_dthick_k, fe);
          F.TH(i) = mn(theta_new, lk.te);
          need_psi = true;
          LGAR_MEASURE_POINT(CLK, 23)
        }
      } else if constexpr (K == NL - 1) {
        if (c.nf0 == NL) {
          // base_case: one front per layer, uniform psi
          S dth[NL], dthick[NL];
#pragma unroll
          for (int j = 0; j < NL; j++) { dth[j] = S(R(0.0)); dthick[j] = S(R(0.0)); }
          S z = F.Z(i) + F.DZ(i) * G->dt_h;
          F.Z(i) = z;
          S psi_old = oc_ps;
          S psi = F.PS(i);
          S base = P.cum[NL - 2];
          S prior_mass = (oc_z - base) * (oc_th - R(0.0));
          S new_mass = (z - base) * (F.TH(i) - R(0.0));
          FirstEval fe;
#pragma unroll
          for (int j = 0; j < NL - 1; j++) {
            const LayerK<S> lj = pick_static(P, j);
            if constexpr (sizeof(R) == 8) {
              // (psi_old IS psi -- the front's psi has not moved yet: one evaluation; plain reals take the derivatives along,
              // the search's first mass evaluation)
              S theta;
              if constexpr (sizeof(S) == sizeof(R)) {
                R th, dt, d2;
                theta_slope2(val(lj.alpha), val(lj.n), val(lj.m), val(lj.te), val(lj.tr), val(psi), th, dt, d2);
                fe.add(val(P.thick[j]), th, R(0.0), dt, d2);
                theta = S(th);
              } else {
                theta = theta_from_h<S, POL>(lj, psi);
              }
              prior_mass = prior_mass + P.thick[j] * (theta - R(0.0));
              new_mass = new_mass + P.thick[j] * (theta - R(0.0));
            } else {
              S theta_old = theta_from_h<S, POL>(lj, psi_old);
              prior_mass = prior_mass + P.thick[j] * (theta_old - R(0.0));
              S theta = theta_from_h<S, POL>(lj, psi);
              new_mass = new_mass + P.thick[j] * (theta - R(0.0));
            }
            dthick[j] = P.thick[j];
          }
          if constexpr (sizeof(R) == 8 && sizeof(S) == sizeof(R)) {
            R tk, dk, ek;
            theta_slope2(val(lk.alpha), val(lk.n), val(lk.m), val(lk.te), val(lk.tr), val(psi), tk, dk, ek);
            fe.add(val(z) - val(base), tk, R(0.0), dk, ek);
            fe.thk = tk;
            fe.have = true;
          }
          if (F.layer(c.fdd) == NL - 1) prior_mass = prior_mass + c.infiltration - (R(0.0) + c.aet);
          S theta_new = theta_mass_balance<NL - 1>(lk, psi, new_mass, prior_mass, dth, dthick, S(R(0.0)), z - base, fe);
          F.TH(i) = mn(theta_new, lk.te);
          need_psi = true;
        }
      }
      if (need_psi) F.PS(i) = h_from_se<S, POL>(lk, se_from_theta(lk, F.TH(i)));
      LGAR_MEASURE_POINT(CLK, 24)
      c.on_z = oc_z; c.on_th = oc_th; c.on_ps = oc_ps;
      c.i = i - 1;
    }
  }
  template <int K> __device__ __forceinline__ void sweep_from(SweepCarry &c) {
    sweep_layer<K>(c);
    if constexpr (K > 0) sweep_from<K - 1>(c);
  }

  __device__ __forceinline__ void move_sweep(S infiltration, S aet, S old_mass, int fdd) {
    SweepCarry c;
    c.i = nf - 1; c.nf0 = nf; c.fdd = fdd;
    c.infiltration = infiltration; c.aet = aet;
    c.on_z = c.on_th = c.on_ps = S(R(0.0));
    c.near_sat = 0u;
    sweep_from<NL - 1>(c);
    near_sat_fronts = c.near_sat;
    if constexpr (!FUSED_WALK) {
      check_column_mass(fdd, old_mass, infiltration, aet);  // after front 0 (Layer.py:1296-1305)
    } else {
      // check_column_mass (Layer.py:655-701) in two halves around ONE walk over the front table.  The column mass is LINEAR in
      // the depth of the (saturated) free-drainage front -- slope = theta_fdd - theta_next, or theta_fdd for the last front of
      // a layer -- so the reference's fixed-step line search has a closed-form root: one step (an offset that is a constant
      // w.r.t. the parameters, like the reference's steps), then the walk that every column takes anyway (mass_and_events)
      // verifies it on the true mass against the reference's own termination test; where that fails -- and for a slope that
      // is not positive -- the reference's loop runs from where the step landed.
      const S theta_e_k1 = sel<S, NL>(P.te, F.layer(fdd));
      const S mass_timestep = (old_mass + infiltration) - (aet + R(0.0));
      bool stepped = false, literal = false;
      if (__builtin_expect(ab(val(F.TH(fdd)) - val(theta_e_k1)) < Tol<R>::mass, 0)) {
        const S current_mass = mass_balance();
        const R err = ab(val(current_mass) - val(mass_timestep));
        if (ab(err - Tol<R>::mass) > Tol<R>::mass) {
          const bool nxt_same = (fdd + 1 < nf) && (F.layer(fdd + 1) == F.layer(fdd));
          const R slope = nxt_same ? val(F.TH(fdd)) - val(F.TH(fdd + 1)) : val(F.TH(fdd));
          if (slope > R(0.0)) {
            F.Z(fdd) = F.Z(fdd) + dv<POL>(val(mass_timestep) - val(current_mass), slope);
            stepped = true;
          } else {
            literal = true;
          }
        }
      }
      if (__builtin_expect(literal, 0)) check_column_mass(fdd, old_mass, infiltration, aet);
      post_mass = mass_and_events(post_event);
      if (stepped) {
        const R err = ab(val(post_mass) - val(mass_timestep));
        if (__builtin_expect(ab(err - Tol<R>::mass) > Tol<R>::mass, 0)) {
          check_column_mass(fdd, old_mass, infiltration, aet);
          post_mass = mass_and_events(post_event);
        }
      }
      post_mass_valid = true;
    }
  }

  // merge_wetting_fronts / is_passing / pass_front / delete_front, Layer.py:826-892: per layer, the first
  // front that has passed its (same-layer, non-boundary) successor absorbs it; <= 1 merge per layer per call.
  __device__ __forceinline__ void merge_fronts() {
    int i = 0, lo = 0;
    while (i < nf - 1) {
      const int k = F.layer(i);
      if (i > 0 && F.layer(i - 1) != k) lo = i;
      const int nx = i + 1;
      const bool passing = (val(F.Z(i)) > val(F.Z(nx))) && (F.layer(nx) == k) && !F.bottom(nx);
      if (!passing) { i++; continue; }
      const int nn = i + 2;
      if (nn >= nf) { status |= LGAR_ST_STRUCT; return; }
      const LayerK<S> lk = pick(P, k);
      S mass = F.Z(i) * (F.TH(i) - F.TH(nx)) + F.Z(nx) * (F.TH(nx) - F.TH(nn));
      F.Z(i) = mass / (F.TH(i) - F.TH(nn));
      S se = se_from_theta(lk, F.TH(i));
      F.PS(i) = h_from_se<S, POL>(lk, se);
      // delete_front: the first front of THIS layer's list that is value-equal to `next`
      int j = lo;
      while (j < nf && F.layer(j) == k && !feq(j, nx)) j++;
      if (j < nf && F.layer(j) == k) fdel(j);
      // this layer is done: continue with the first front of the next layer
      i = lo;
      while (i < nf && F.layer(i) == k) i++;
      lo = i;
    }
  }

  // wetting_fronts_cross_layer_boundary / recalibrate, Layer.py:894-1008: a front that has advanced past its
  // layer's lower boundary swaps roles with the boundary front.  On the flat array the re-bucketing
  // (update_wetting_fronts :939-963) is just the tag change; the moved front is not revisited in this call.
  __device__ __forceinline__ void cross_layer_boundary() {
    int i = 0;
    while (i < nf - 1) {
      const int k = F.layer(i);
      const int nx = i + 1, nn = i + 2;
      const S cumk = cum_at(k);
      if (!(val(F.Z(i)) > val(cumk) && val(F.Z(nx)) == val(cumk))) { i++; continue; }
      if (k == NL - 1) {
        if (G->bottom_mode == 0) { status |= LGAR_ST_BOTTOM; return; }  // reference: AttributeError at Layer.py:980
        i++;  // LGAR-C intent: the bottom layer has no layer below; the domain-boundary step handles this front
        continue;
      }
      if (nn >= nf) { status |= LGAR_ST_STRUCT; return; }
      const LayerK<S> lk = pick(P, k);
      const LayerK<S> ln = pick(P, k + 1);
      S overshot = F.Z(i) - F.Z(nx);
      S se = se_from_theta(lk, F.TH(i));
      F.PS(i) = h_from_se<S, POL>(lk, se);
      S theta_new = theta_from_h<S, POL>(ln, F.PS(i));
      S mbal = overshot * (F.TH(i) - F.TH(nx));
      S zc = mbal / (theta_new - F.TH(nn));
      S depth_new = cumk + zc;
      F.Z(i) = cumk;
      F.TH(nx) = theta_new;
      F.PS(nx) = F.PS(i);
      // (fast modes: the reference's update_psi re-derives the new front's psi from its theta at the end of the move -- nothing
      // reads it before -- and right after a crossing that round trip is not a no-op: the front is all but saturated)
      if constexpr (MODE != 0) F.PS(nx) = h_from_se<S, POL>(ln, se_from_theta(ln, theta_new));
      F.Z(nx) = depth_new;
      F.DZ(nx) = F.DZ(i);
      F.DZ(i) = S(R(0.0));
      F.set_flag(i, k, true);
      F.set_flag(nx, k + 1, false);
      i += 2;
    }
  }

  // wetting_front_cross_domain_boundary, Layer.py:1010-1053 (bottom_mode 1 only; the reference cannot reach it without
  // crashing): the second-to-last front of the domain has passed the column bottom -> its overshoot leaves as
  // percolation, the bottom front takes its theta, and it is deleted.
  __device__ __forceinline__ S cross_domain_boundary() {
    S flux = S(R(0.0));
    if (nf < 2) return flux;
    const int i = nf - 2, nx = nf - 1;
    const int k = F.layer(i);
    if (val(F.Z(i)) > val(cum_at(k))) {
      const LayerK<S> lk = pick(P, k);
      flux = (F.TH(i) - F.TH(nx)) * (F.Z(i) - F.Z(nx));
      F.TH(nx) = F.TH(i);
      S se = se_from_theta(lk, F.TH(i));
      F.PS(nx) = h_from_se<S, POL>(lk, se);
      k_deepest = k_from_se<S, POL>(lk, se);
      fdel(i);
    }
    return flux;
  }

  // fix_dry_over_wet_fronts / cleanup_wetting_fronts / update_layer_fronts, Layer.py:1055-1143: per layer,
  // the first front that is not wetter than its same-layer successor is deleted (<= 1 per layer per call).
  __device__ __forceinline__ S fix_dry_over_wet() {
    S ls[NL];
#pragma unroll
    for (int j = 0; j < NL; j++) ls[j] = S(R(0.0));
    int i = 0;
    while (i < nf - 1) {
      const int k = F.layer(i);
      const int nx = i + 1;
      if (!(val(F.TH(i)) <= val(F.TH(nx)) && F.layer(nx) == k)) { i++; continue; }
      S before = mass_balance();
      fdel(i);  // the next front now sits at index i
      if (k > 0) {
        psi_rewritten = true;
        int found = 0;
        while (found < nf && !feq(found, i)) found++;
        if (found >= nf) status |= LGAR_ST_STRUCT;
        else {
          const int lj = F.layer(found);
          const LayerK<S> lf = pick(P, lj);
          F.PS(found) = h_from_se<S, POL>(lf, se_from_theta(lf, F.TH(found)));
          const S dry_th = F.TH(found), dry_ps = F.PS(found);
          for (int q = 0; q < nf; q++) {
            const int lq = F.layer(q);
            if (lq < lj) {  // quirk: EVERY front of all shallower layers is overwritten (Layer.py:1117-1143)
              const LayerK<S> lql = pick(P, lq);
              F.PS(q) = h_from_se<S, POL>(lql, se_from_theta(lql, dry_th));
              // another layer's theta can exceed this layer's theta_e: Se > 1, a negative pow base, and the reference
              // raises (utils.py:25-27) -- but the NaN psi is overwritten by update_psi before anything reads it, so it
              // must be caught here
              if (is_nan(val(F.PS(q)))) status |= LGAR_ST_NEGBASE;
              F.TH(q) = theta_from_h<S, POL>(lql, dry_ps);
            }
          }
        }
      }
      S after = mass_balance();
      S mc = ab(after - before);
#pragma unroll
      for (int j = 0; j < NL; j++) ls[j] = choose(k == j, ls[j] + mc, ls[j]);
      // this layer is done: continue with the first front of the next layer
      while (i < nf && F.layer(i) == k) i++;
    }
    S tot = ls[NL - 1];
#pragma unroll
    for (int j = NL - 2; j >= 0; j--) tot = ls[j] + tot;
    return tot;
  }

  // update_psi, Layer.py:1157-1174: psi from theta for every front but the deepest of the domain (K: see front_k)
  __device__ __forceinline__ void update_psi() {
    for (int i = 0; i < nf - 1; i++) {
      const LayerK<S> lk = pick(P, F.layer(i));
      F.PS(i) = h_from_se<S, POL>(lk, se_from_theta(lk, F.TH(i)));
    }
  }

  // K of front i as calc_dzdt / the state dump see it
  __device__ __forceinline__ S front_k(int i, const LayerK<S> &lk) const {
    S k = k_from_se<S, POL>(lk, se_from_theta(lk, F.TH(i)));
    if (i == 0 && new_front_frozen) k = k * G->frozen;
    return k;
  }

  // Does any of the post-sweep passes have work to do?  merge (Layer.py:826-892), layer-boundary crossing (:894-1008),
  // domain-boundary crossing (:1010-1053) and dry-over-wet (:1055-1143) each act only when their trigger holds for some
  // adjacent pair, and a pass that finds no trigger changes nothing; so a column without a trigger skips the four passes
  // as a block (one cheap scan instead of five; the wave skips the code when none of its columns has one).
  __device__ __forceinline__ bool front_event_pending() const {
    bool ev = false;
    S z1 = F.Z(0), t1 = F.TH(0);
    int f1 = F.flag(0);
    for (int i = 0; i + 1 < nf; i++) {
      const S z2 = F.Z(i + 1), t2 = F.TH(i + 1);
      const int f2 = F.flag(i + 1);
      const bool same = ((f1 ^ f2) & 0x7f) == 0;
      ev = ev || (same && !(f2 & LGAR_FLAG_BOTTOM) && val(z1) > val(z2));   // passing
      ev = ev || (same && val(t1) <= val(t2));                                // dry over wet
      ev = ev || (val(z1) > val(cum_at(f1 & 0x7f)));                         // past its layer (or the domain) bottom
      z1 = z2; t1 = t2; f1 = f2;
    }
    return ev;
  }

  // dpLGAR.move_wetting_front, models/dpLGAR.py:340-367.  Returns the bottom-boundary flux.
  // wetting_front_cross_domain_boundary (Layer.py:1010-1053) cannot be reached in the reference without
  // crashing in the layer-boundary step first; here that case sets LGAR_ST_BOTTOM and the flux is 0.
  __device__ __forceinline__ S move_wetting_front(S infiltration, S &aet, S old_mass, int fdd) {
    move_sweep(infiltration, aet, old_mass, fdd);
    LGAR_MEASURE_POINT(CLK, 3)
    S bottom_flux = S(R(0.0));
    // (a per-lane decision: a column's results must not depend on which other columns share its wave)
    LGAR_MEASURE_POINT(DUP_EVENT)
    bool pending;
    if constexpr (FUSED_WALK) pending = post_event;
    else pending = front_event_pending();
    if (__builtin_expect(pending, 0)) {
      post_mass_valid = false;
      psi_rewritten = false;
      for (int pass = 0; pass < 2; pass++) {
        merge_fronts();
        if (pass == 0) cross_layer_boundary();
      }
      if (G->bottom_mode != 0) bottom_flux = cross_domain_boundary();
      S mass_change = fix_dry_over_wet();
      if (ab(val(mass_change)) > R(1e-7)) aet = aet - mass_change;
      // the passes can leave psi inconsistent with theta (dry-over-wet in a deeper layer writes the psi of ANOTHER
      // layer's theta into the fronts above it, Layer.py:1117-1143): the reference's update_psi repairs that
      // ... every front a merge or a crossing touched carries psi = h(Se(theta)) already (each pass sets it); only that rewrite
      // needs the reference's full pass -- and a column with a near-saturated boundary front (below), whose bit may no longer
      // sit at the front's index after a deletion
      if constexpr (MODE != 0) { if (psi_rewritten || near_sat_fronts != 0u) update_psi(); }
    } else if constexpr (MODE != 0 && sizeof(R) == 8) {
      // A layer's deepest front takes the psi of the front below and theta(psi) of its own layer (Layer.py:389-418); the
      // reference's update_psi then re-derives psi from that theta.  The round trip returns psi to ~eps / (alpha psi)^n
      // relative: nothing a decision can see (the tie rule of calc_wetting_front_free_drainage has rtol 1e-5, atol 1e-8)
      // unless the front is all but saturated -- (alpha psi)^n < 1e-6, psi below ~0.03 cm -- where the reference's psi is
      // the round trip's noise and its tie rule decides on that noise (fixture crash_bottom_three_layer_synth3: 1.77e-7 cm
      // comes back as 4.64e-6).  Those fronts, and only those, go through the same round trip here.
      if (__builtin_expect(near_sat_fronts != 0u, 0)) {
        for (int i = 0; i < nf - 1; i++)
          if (near_sat_fronts & (1u << i)) {
            const LayerK<S> lk = pick(P, F.layer(i));
            F.PS(i) = h_from_se<S, POL>(lk, se_from_theta(lk, F.TH(i)));
          }
      }
    }
    // update_psi (Layer.py:1157-1174) re-derives psi from theta for every front but the deepest.  After the sweep alone
    // every front already carries psi = h(Se(theta)) (in-layer and base-case fronts) or the psi its theta was computed
    // from (a layer's deepest front), so the pass only adds a theta -> psi -> theta round trip: MODE 0 keeps it, MODE 1
    // runs it only after an event.
    if constexpr (MODE == 0) update_psi();
    LGAR_MEASURE_POINT(CLK, 4)
    LGAR_MEASURE_POINT(DUP_PSI)
    return bottom_flux;
  }

  // calc_dzdt, Layer.py:1176-1252 (calc_bottom_sum :1557-1582): one Geff per moving front
  // Every lane walks its OWN list of moving fronts: fronts that need no Geff (layer bottoms, delta_theta <= 0) are finished
  // on the way, and the k-th moving front of every column meets the others' k-th in one evaluation of the trapezoid --
  // columns whose moving fronts sit at different indices (one has crossed into the next layer, another has two fronts
  // above a boundary) would otherwise take turns.  The fronts' dz/dt do not depend on one another, so the order is free.
  // Cooperating lanes, groups of at least LGAR_COOP_PAIR_LANES: TWO moving fronts at a time.  A trapezoid's cost on a lone wave
  // is its dependent chains -- the two pows that open it, the 120 additions of its heads, the 120 of its sum -- not its nodes,
  // and most steps have two moving fronts: the lower half of the group takes one front, the upper half the other, through
  // the same instruction stream (each half with its own front's layer, thetas and psi as operands, its own row of the group's
  // LDS table, its lanes numbered from 0), and the halves then exchange Geff and the conductivities that rode along.  Every
  // value is computed exactly as the one-front path computes it.  Returns the index the one-front loop resumes after (it
  // finishes a last unpaired front, and everything when there is no pair).
  __device__ __forceinline__ int calc_dzdt_pairs(S h_p) {
    const int hl = share_lanes >> 1;
    const bool upper = coop_rank >= hl;
    const int r2 = upper ? coop_rank - hl : coop_rank;
    R *tab2 = xchg + (upper ? LGAR_COOP_TAB_ROW : 0);
    int i = -1;
    auto next_moving = [&]() {  // the one-front loop's scan: fronts that need no Geff are finished on the way
      while (++i < nf - 1) {
        if (F.bottom(i)) { F.DZ(i) = S(R(0.0)); continue; }
        const bool top = F.layer(i) == 0;
        if (top && val(F.TH(i + 1)) > val(F.TH(i))) status |= LGAR_ST_THETA_ORDER;  // Layer.py:1206-1208
        if (val(F.TH(i) - F.TH(i + 1)) > R(0.0)) return i;
        F.DZ(i) = S(R(0.0));
      }
      return -1;
    };
    for (;;) {
      const int ia = next_moving();
      if (ia < 0) return i;
      const int ib = next_moving();
      const int ka = F.layer(ia), kb = (ib >= 0) ? F.layer(ib) : 0;
      // no pair, or a half too small for a front's riders: the one-front loop takes over at front ia (the fronts behind it are
      // done; the scan's side effects beyond it are idempotent)
      if (ib < 0 || hl < 5 + ((ka > kb) ? ka : kb)) return ia - 1;
      const int im = upper ? ib : ia;
      const int k = upper ? kb : ka;
      const LayerK<S> lk = pick(P, k);
      const S theta_1 = F.TH(im + 1), theta_2 = F.TH(im);
      CoopRiders riders;
      {
        const int e = r2 - 5;  // my layer above (lanes 5 .. 5 + k - 1 of my half)
        LayerK<S> le = lk;
        S th_e = theta_2;
#pragma unroll
        for (int j = 0; j < NL - 1; j++) {
          const bool mine = (e == j) && (j < k);
          le.alpha = choose(mine, P.alpha[j], le.alpha); le.n = choose(mine, P.n[j], le.n); le.m = choose(mine, P.m[j], le.m);
          le.inv_m = choose(mine, P.inv_m[j], le.inv_m); le.ksat = choose(mine, P.ksat[j], le.ksat);
          le.te = choose(mine, P.te[j], le.te); le.tr = choose(mine, P.tr[j], le.tr);
        }
        if (ka > 0 || kb > 0) {
          const S tl = theta_from_h<S, POL>(le, F.PS(im));
          th_e = choose(e >= 0 && e < k, tl, th_e);
        }
        riders.n = 1 + k;
        riders.l = le;
        riders.se = se_from_theta(le, th_e);
#pragma unroll
        for (int q = 0; q < LGAR_LMAX; q++) riders.k[q] = R(1.0);
      }
      LGAR_COUNT_GEFF_CALL(1)
      LGAR_COUNT_GEFF_CALL(1)
      const S g_mine = geff_fused<S>(lk, theta_1, theta_2, G->nint, tab2, hl, r2, &riders);
      // the halves exchange what they found (every lane of a half stores the same values)
      tab2[0] = g_mine;
#pragma unroll
      for (int q = 0; q < NL; q++) tab2[1 + q] = riders.k[q];
      lds_exchange_point();
      S g2[2];
      R kk[2][NL];
#pragma unroll
      for (int h = 0; h < 2; h++) {
        g2[h] = xchg[h * LGAR_COOP_TAB_ROW];
#pragma unroll
        for (int q = 0; q < NL; q++) kk[h][q] = xchg[h * LGAR_COOP_TAB_ROW + 1 + q];
      }
      lds_exchange_point();
#pragma unroll
      for (int h = 0; h < 2; h++) {
        const int f = h ? ib : ia;
        const int kf = h ? kb : ka;
        const LayerK<S> lf = pick(P, kf);
        const S g = g2[h];
        S ki = kk[h][0];
        if (f == 0 && new_front_frozen) ki = ki * G->frozen;
        if (is_nan(val(g))) status |= LGAR_ST_NAN;
        const S delta_theta = F.TH(f) - F.TH(f + 1);
        S dzdt;
        if (kf == 0) {
          dzdt = dv<POL>(S(R(1.0)), delta_theta) * (dv<POL>(lf.ksat * (g + h_p), F.Z(f)) + ki);
        } else {
          S den = S(R(0.0)) + dv<POL>(F.Z(f) - cum_prev(kf), ki);
#pragma unroll
          for (int j = 0; j < NL - 1; j++)
            if (j < kf) {
              S pt = (j != 0) ? P.cum[(j > 0) ? j - 1 : 0] : S(R(0.0));
              den = den + dv<POL>(P.cum[j] - pt, S(kk[h][1 + j]));
            }
          dzdt = dv<POL>(S(R(1.0)), delta_theta) * (dv<POL>(F.Z(f), den) + dv<POL>(lf.ksat * (g + h_p), F.Z(f)));
        }
        F.DZ(f) = dzdt;
      }
    }
  }

  // ... and in the mixed-precision mode (MODE 6): the halves of ANY group (the mixed trapezoid's ends are evaluated by every
  // lane, nothing rides along) take one moving front each -- its trapezoid (geff_mixed_heads, the half's lanes splitting its
  // four-node groups through the half's own 32 slots of the group's table row) and the conductivities of the layers above at
  // its psi -- and exchange Geff, K(theta) / Ksat of the wet end and those conductivities.
  __device__ __forceinline__ int calc_dzdt_pairs_mixed(S h_p) {
    const int hl = share_lanes >> 1;
    const bool upper = coop_rank >= hl;
    const int r2 = upper ? coop_rank - hl : coop_rank;
    R *tab2 = xchg + (upper ? 64 : 0);
    int i = -1;
    auto next_moving = [&]() {  // the one-front loop's scan: fronts that need no Geff are finished on the way
      while (++i < nf - 1) {
        if (F.bottom(i)) { F.DZ(i) = S(R(0.0)); continue; }
        const bool top = F.layer(i) == 0;
        if (top && val(F.TH(i + 1)) > val(F.TH(i))) status |= LGAR_ST_THETA_ORDER;  // Layer.py:1206-1208
        if (val(F.TH(i) - F.TH(i + 1)) > R(0.0)) return i;
        F.DZ(i) = S(R(0.0));
      }
      return -1;
    };
    for (;;) {
      const int ia = next_moving();
      if (ia < 0) return i;
      const int ib = next_moving();
      if (ib < 0) return ia - 1;  // no pair: the one-front loop takes over at front ia
      const int ka = F.layer(ia), kb = F.layer(ib);
      const int kmax = (ka > kb) ? ka : kb;
      const int im = upper ? ib : ia;
      const int k = upper ? kb : ka;
      const LayerK<S> lk = pick(P, k);
      LGAR_COUNT_GEFF_CALL(1)
      LGAR_COUNT_GEFF_CALL(1)
      double kr_mine;
      const S g_mine = geff_mixed_heads<true>(lk, F.TH(im + 1), F.TH(im), F.PS(im + 1), F.PS(im), G->nint, kr_mine, tab2, hl, r2);
      tab2[0] = g_mine;
      tab2[1] = kr_mine;
#pragma unroll
      for (int j = 0; j < NL - 1; j++)
        if (j < kmax) {  // (both halves, for the deeper of the two fronts: one instruction stream)
          const LayerK<S> lj = pick_static(P, j);
          const S tl = theta_from_h<S, POL>(lj, F.PS(im));
          tab2[2 + j] = k_from_se<S, POL>(lj, se_from_theta(lj, tl));
        }
      lds_exchange_point();
      S g2[2];
      R kr2[2], kk[2][NL];
#pragma unroll
      for (int h = 0; h < 2; h++) {
        g2[h] = xchg[h * 64];
        kr2[h] = xchg[h * 64 + 1];
#pragma unroll
        for (int j = 0; j < NL - 1; j++) kk[h][j] = xchg[h * 64 + 2 + j];
      }
      lds_exchange_point();
#pragma unroll
      for (int h = 0; h < 2; h++) {
        const int f = h ? ib : ia;
        const int kf = h ? kb : ka;
        const LayerK<S> lf = pick(P, kf);
        const S g = g2[h];
        S ki = lf.ksat * kr2[h];
        if (f == 0 && new_front_frozen) ki = ki * G->frozen;
        if (is_nan(val(g))) status |= LGAR_ST_NAN;
        const S delta_theta = F.TH(f) - F.TH(f + 1);
        S dzdt;
        if (kf == 0) {
          dzdt = dv<POL>(S(R(1.0)), delta_theta) * (dv<POL>(lf.ksat * (g + h_p), F.Z(f)) + ki);
        } else {
          S den = S(R(0.0)) + dv<POL>(F.Z(f) - cum_prev(kf), ki);
#pragma unroll
          for (int j = 0; j < NL - 1; j++)
            if (j < kf) {
              S pt = (j != 0) ? P.cum[(j > 0) ? j - 1 : 0] : S(R(0.0));
              den = den + dv<POL>(P.cum[j] - pt, S(kk[h][j]));
            }
          dzdt = dv<POL>(S(R(1.0)), delta_theta) * (dv<POL>(F.Z(f), den) + dv<POL>(lf.ksat * (g + h_p), F.Z(f)));
        }
        F.DZ(f) = dzdt;
      }
    }
  }

  __device__ __forceinline__ void calc_dzdt(S h_p) {
    int i = -1;
    if constexpr (MODE == 6 && sizeof(S) == 8 && sizeof(R) == 8) {
      if (share_lanes >= 4 && !G->closed_form) i = calc_dzdt_pairs_mixed(h_p);
    }
    if constexpr (COOP && !mixed_mode(MODE)) {
      // (a group of 8 or more lanes owns TWO rows of the exchange table: lgar_kernels_nl.hip)
      if (share_lanes >= LGAR_COOP_PAIR_LANES && !G->closed_form) i = calc_dzdt_pairs(h_p);
    }
    for (;;) {
      // advance to the next front that moves
      bool found = false;
      while (!found && ++i < nf - 1) {
        if (F.bottom(i)) { F.DZ(i) = S(R(0.0)); continue; }
        const bool top = F.layer(i) == 0;
        if (top && val(F.TH(i + 1)) > val(F.TH(i))) status |= LGAR_ST_THETA_ORDER;  // Layer.py:1206-1208
        if (val(F.TH(i) - F.TH(i + 1)) > R(0.0)) found = true;
        else F.DZ(i) = S(R(0.0));
      }
      if (any_lane(found) == 0ull) break;
      LGAR_MEASURE_POINT(CLK, 10)
      if (found) {
        const int k = F.layer(i);
        const LayerK<S> lk = pick(P, k);
        S theta_1 = F.TH(i + 1), theta_2 = F.TH(i);
        S delta_theta = F.TH(i) - F.TH(i + 1);
        S g, ki;
        bool fronts_done = false;
        if constexpr (mixed_mode(MODE) && sizeof(S) == 8 && sizeof(R) == 8) {
          if (!G->closed_form) {  // mixed precision: heads from the fronts' psi, K(theta_i) from the trapezoid's wet end node
            double kr_end;
            g = capillary_drive_fronts(lk, theta_1, theta_2, F.PS(i + 1), F.PS(i), kr_end);
            ki = lk.ksat * kr_end;
            if (i == 0 && new_front_frozen) ki = ki * G->frozen;
            fronts_done = true;
          }
        }
        CoopRiders riders;  // (cooperating lanes only)
        if constexpr (COOP && !mixed_mode(MODE)) {
          // cooperating lanes: the front's own K(theta) and the K of the layers above at its psi (two pows each, after the
          // two of theta(psi)) ride along with the four evaluations that open the trapezoid -- see CoopRiders
          if (!G->closed_form && share_lanes >= 5 + k) {
            const int e = coop_rank - 5;  // my layer above (lanes 5 .. 5 + k - 1)
            LayerK<S> le = lk;
            S th_e = F.TH(i);
            if (k > 0) {
#pragma unroll
              for (int j = 0; j < NL - 1; j++) {
                const bool mine = (e == j) && (j < k);
                le.alpha = choose(mine, P.alpha[j], le.alpha); le.n = choose(mine, P.n[j], le.n); le.m = choose(mine, P.m[j], le.m);
                le.inv_m = choose(mine, P.inv_m[j], le.inv_m); le.ksat = choose(mine, P.ksat[j], le.ksat);
                le.te = choose(mine, P.te[j], le.te); le.tr = choose(mine, P.tr[j], le.tr);
              }
              const S tl = theta_from_h<S, POL>(le, F.PS(i));
              th_e = choose(e >= 0 && e < k, tl, th_e);
            }
            riders.n = 1 + k;
            riders.l = le;
            riders.se = se_from_theta(le, th_e);
#pragma unroll
            for (int q = 0; q < LGAR_LMAX; q++) riders.k[q] = R(1.0);
            g = capillary_drive(lk, theta_1, theta_2, 1, &riders);
            ki = riders.k[0];
            if (i == 0 && new_front_frozen) ki = ki * G->frozen;
            fronts_done = true;
          }
        }
        LGAR_MEASURE_POINT(F32_HEADS_FROM_PSI, lk, i, g, ki, fronts_done)
        if (!fronts_done) {
          g = capillary_drive(lk, theta_1, theta_2, 1);
          ki = front_k(i, lk);
        }
        if (is_nan(val(g))) status |= LGAR_ST_NAN;
        S dzdt;
        if (k == 0) {
          dzdt = dv<POL>(S(R(1.0)), delta_theta) * (dv<POL>(lk.ksat * (g + h_p), F.Z(i)) + ki);
        } else {
          S den = S(R(0.0)) + dv<POL>(F.Z(i) - cum_prev(k), ki);
#pragma unroll
          for (int j = 0; j < NL - 1; j++)
            if (j < k) {
              const LayerK<S> lj = pick_static(P, j);
              S kl;
              bool rode = false;
              if constexpr (COOP && !mixed_mode(MODE)) {
                if (riders.n > 0) { kl = riders.k[1 + j]; rode = true; }
              }
              if (!rode) {
                S tl = theta_from_h<S, POL>(lj, F.PS(i));
                kl = k_from_se<S, POL>(lj, se_from_theta(lj, tl));
              }
              S pt = (j != 0) ? P.cum[(j > 0) ? j - 1 : 0] : S(R(0.0));
              den = den + dv<POL>(P.cum[j] - pt, kl);
            }
          dzdt = dv<POL>(S(R(1.0)), delta_theta) * (dv<POL>(F.Z(i), den) + dv<POL>(lk.ksat * (g + h_p), F.Z(i)));
        }
        F.DZ(i) = dzdt;
        LGAR_MEASURE_POINT(CLK, 17)
      }
    }
  }

  // calc_dry_depth, Layer.py:1309-1334
  __device__ __forceinline__ S calc_dry_depth() {
    const LayerK<S> l0 = pick_static(P, 0);
    S delta_theta = l0.te - F.TH(0);
    S tau = G->dt_h * l0.ksat / delta_theta;
    S g = capillary_drive(l0, F.TH(0), l0.te, 2);
    if (is_nan(val(g))) status |= LGAR_ST_NAN;
    S dry = R(0.5) * (tau + sq(tau * tau + R(4.0) * tau * g));
    return mn(P.cum[0], dry);
  }

  // Layer.create_surficial_front, Layer.py:1336-1416
  __device__ __forceinline__ void create_surficial_front(S dry_depth, S &ponded, S &infiltration) {
    if (nf >= cap) { status |= LGAR_ST_OVERFLOW; return; }
    const LayerK<S> l0 = pick_static(P, 0);
    S cur_theta = F.TH(0);
    S delta_theta = l0.te - cur_theta;
    S theta_new;
    bool to_bottom = false;
    if (val(dry_depth * delta_theta) > val(ponded)) {
      infiltration = ponded;
      theta_new = mn((cur_theta + ponded / dry_depth), l0.te);
      ponded = S(R(0.0));
    } else {
      infiltration = dry_depth * delta_theta;
      ponded = ponded - (dry_depth * delta_theta);
      theta_new = l0.te;
      to_bottom = !(val(dry_depth) < val(P.cum[0]));
    }
    for (int j = nf; j > 0; j--) F.copy(j, j - 1);
    nf++;
    F.Z(0) = dry_depth;
    F.TH(0) = theta_new;
    F.set_flag(0, 0, to_bottom);
    F.PS(0) = h_from_se<S, POL>(l0, se_from_theta(l0, theta_new));
    new_front_frozen = true;
    F.DZ(0) = S(R(0.0));
  }

  // insert_water, Layer.py:1418-1536 (get_drainage_neighbors :1584-1607, calc_bottom_sum_f_p :1538-1555):
  // Green-Ampt infiltration capacity f_p and the infiltration / runoff / ponding split
  __device__ __forceinline__ void insert_water(int fdd, S precip, S &ponded, S &infiltration, S &runoff) {
    const R dt = G->dt_h;
    S h_p = (ponded - precip) * dt;
    if (val(h_p) < R(0.0)) h_p = S(R(0.0));
    const int kfp = F.layer(fdd);
    int lo, len;
    range_of(kfp, lo, len);
    const int nxt_i = lo + 1;  // the front after the FIRST front of the free-drainage front's layer (quirk)
    // reference: AttributeError (Layer.py:1606) when the free-drainage front is the lone front of the bottom layer.
    // With one front per layer no neighbour is needed (Geff = 0); bottom_mode 1 lets that case through.
    if (nxt_i >= nf && (nf != NL || G->bottom_mode == 0)) { status |= LGAR_ST_STRUCT; return; }
    const LayerK<S> lk = pick(P, kfp);
    S g = S(R(0.0));
    // quirk: with a fully saturated one-front top layer right after a layer crossing, nxt_i is a front of the
    // NEXT layer and Se > 1: the reference raises ValueError (negative pow base, physics/utils.py:25-27);
    // here the NaN is flagged and the IEEE min below drops it (all ponded water infiltrates).
    // Geff(theta_1 -> theta_e) is a pure function of theta_1 and the layer's parameters, and during a storm theta_1 -- the
    // front BEHIND the wetting front, usually the layer's untouched boundary front -- stays the same for many steps: the
    // value of the previous call is reused when its inputs are bit for bit the same (the whole trapezoid is skipped when
    // that holds for every column of the wavefront; same results either way).
    if (nf != NL) {
      const S theta_1 = F.TH(nxt_i < nf ? nxt_i : nf - 1);
      bool hit = memo_layer == kfp && same_bits(theta_1, memo_theta);
      if constexpr (sizeof(S) != sizeof(R)) {
        // Dual numbers: same_bits compares the tangent as well, the only place where a branch could depend on it.  The W
        // lanes of a shared group (LgarDims.tangent_share) hold the same values and different tangents and must enter the
        // trapezoid TOGETHER (geff_shared_blocks exchanges nodes between them): when any lane's tangent alone says "miss",
        // every lane whose values match recomputes -- the same result as its memo, bit for bit.
        if (share_lanes >= 2) {
          const bool value_hit = memo_layer == kfp && val(theta_1) == val(memo_theta);
          if (any_lane(value_hit && !hit) != 0ull) hit = false;
        }
      }
      if (hit) {
        g = memo_g;
      } else {
        g = capillary_drive(lk, theta_1, lk.te, 3);
        memo_theta = theta_1; memo_g = g; memo_layer = kfp;
      }
    }
    if (is_nan(val(g))) status |= LGAR_ST_NAN | LGAR_ST_NEGBASE;
    S f_p;
    if (kfp == 0) {
      f_p = P.ksat[0] * (R(1.0) + dv<POL>(g + h_p, F.Z(fdd)));
    } else {
      S fd_ksat = lk.ksat * G->frozen;
      S bottom_sum = dv<POL>(F.Z(fdd) - cum_prev(kfp), fd_ksat);
      bottom_sum = bottom_sum + dv<POL>(P.cum[0] - R(0.0), P.ksat[0] * G->frozen);
#pragma unroll
      for (int j = 1; j < NL - 1; j++)
        if (j < kfp) {
          const LayerK<S> lj = pick_static(P, j);
          S tl = theta_from_h<S, POL>(lj, F.PS(fdd));
          S kl = k_from_se<S, POL>(lj, se_from_theta(lj, tl));
          bottom_sum = bottom_sum + dv<POL>(P.cum[j] - P.cum[j - 1], kl);
        }
      f_p = dv<POL>(F.Z(fdd), bottom_sum) + dv<POL>((g + h_p) * fd_ksat, F.Z(fdd));
    }
    S pond_temp = ponded - f_p * dt;
    if (val(pond_temp) < R(0.0)) pond_temp = S(R(0.0));
    S fp_cm = f_p * dt;
    if (G->pdm > R(0.0)) {
      if (val(pond_temp) < G->pdm) {
        infiltration = mn(ponded, fp_cm);
        ponded = ponded - infiltration;
      } else if (val(pond_temp) > G->pdm) {
        ponded = S(G->pdm);
        infiltration = fp_cm;
      }
      S r = pond_temp - G->pdm;
      runoff = (val(r) > R(0.0)) ? r : S(R(0.0));
    } else {
      infiltration = mn(ponded, fp_cm);
      S r = ponded - infiltration;
      ponded = S(G->pdm);
      runoff = (val(r) > R(0.0)) ? r : S(R(0.0));
    }
  }

  // dpLGAR.set_internal_states, models/dpLGAR.py:97-147
  __device__ __forceinline__ void init_state() {
    nf = NL;
    status = 0;
    new_front_frozen = false;
#pragma unroll
    for (int k = 0; k < NL; k++) {
      const LayerK<S> lk = pick_static(P, k);
      F.Z(k) = P.cum[k];
      F.TH(k) = theta_from_h<S, POL>(lk, S(G->initial_psi));
      F.PS(k) = S(G->initial_psi);
      if (k == NL - 1) k_deepest = k_from_se<S, POL>(lk, se_from_theta(lk, F.TH(k)));
      F.DZ(k) = S(R(0.0));
      F.set_flag(k, k, true);
    }
    ponded_water = previous_precip = S(R(0.0));
#pragma unroll
    for (int i = 0; i < LGAR_GMAX; i++) giuh_q[i] = S(R(0.0));
    ending_volume = mass_balance();
    drain();
  }

  __device__ __forceinline__ void drain() {
    a_precip = a_pet = a_aet = a_infil = a_runoff = a_perc = a_giuh = a_disch = S(R(0.0));
  }

  // dpLGAR.forward, models/dpLGAR.py:154-299: one forcing step = nsub sub-steps
  __device__ __forceinline__ void forward(S precip, S pet) {
    const R dt = G->dt_h;
    // a NaN in the forcing slips through the reference unnoticed (every comparison with it is false; the step's runoff comes
    // out NaN, the front table stays finite): flagged here, a data fault the caller should hear about
    if (is_nan(val(precip)) || is_nan(val(pet))) status |= LGAR_ST_NAN;
    S ending_volume_sub = ending_volume;
    for (int sub = 0; sub < G->nsub; sub++) {
      if (status & DEAD) return;  // dead column
      new_front_frozen = false;
      post_mass_valid = false;
      S precip_sub = precip * dt;
      S pet_sub = pet * dt;
      S ponded_depth_sub = precip_sub + ponded_water;
      S ponded_water_sub = S(R(0.0)), runoff_sub = S(R(0.0)), infiltration_sub = S(R(0.0)), AET_sub = S(R(0.0));
      // create_surficial_front predicate, models/dpLGAR.py:310-323
      const bool create = (val(previous_precip) == R(0.0)) && (val(precip_sub) > R(0.0)) && (val(ponded_water) == R(0.0));
      LGAR_MEASURE_POINT(CLK, 0)
      LGAR_MEASURE_POINT(DUP_FDD)
      const int fdd = free_drainage_front();
      const bool saturated = val(F.TH(0)) >= val(P.te[0]);  // Layer.is_saturated, Layer.py:785-793
      if (val(pet) > R(0.0)) {
        if (!aet_psi_wp_known) {  // four pows, once per column and launch instead of once per sub-step
          aet_psi_wp_memo = aet_psi_wp<S, POL>(pick_static(P, 0), G->wp_psi);
          aet_psi_wp_known = true;
        }
        AET_sub = aet_from_psi_wp<S, POL>(pet, dt, F.PS(0), aet_psi_wp_memo);
      }
      if constexpr (!LATE_FORCING_SUMS) {
        a_precip = a_precip + precip_sub;
        a_pet = a_pet + ((val(pet_sub) > R(0.0)) ? pet_sub : S(R(0.0)));
      }
      // Single call site for the front move (models/dpLGAR.py:199-266 re-ordered, same data flow): columns
      // that create a surficial front move first with zero infiltration and then create it; the others
      // infiltrate (insert_water) and then move.  update_ponded_depth touches no front state, so doing it
      // after the move is equivalent.
      const bool inserting = !create && val(ponded_depth_sub) > R(0.0);
      LGAR_MEASURE_POINT(CLK, 1)
      if (inserting) {
        LGAR_ABLATABLE(INSERT, insert_water(fdd, precip_sub, ponded_depth_sub, infiltration_sub, runoff_sub);)
        a_infil = a_infil + infiltration_sub;
        a_runoff = a_runoff + runoff_sub;
        ponded_water_sub = ponded_depth_sub;
      }
      LGAR_MEASURE_POINT(CLK, 2)
      if (!create || !saturated) {
        S perc_sub = S(R(0.0));
        LGAR_ABLATABLE(MOVE, perc_sub = move_wetting_front(create ? S(R(0.0)) : infiltration_sub, AET_sub, ending_volume_sub, fdd);)
        if (!create) a_perc = a_perc + perc_sub;
      }
      if (__builtin_expect(create && !saturated, 0)) {
        S dry_depth = calc_dry_depth();
        create_surficial_front(dry_depth, ponded_depth_sub, infiltration_sub);
        a_infil = a_infil + infiltration_sub;
        post_mass_valid = false;
      }
      if (!inserting) {
        // update_ponded_depth, models/dpLGAR.py:369-382
        if (val(ponded_depth_sub) < G->pdm) {
          runoff_sub = S(R(0.0));
          ponded_water_sub = ponded_depth_sub;
          ponded_depth_sub = S(R(0.0));
        } else {
          runoff_sub = ponded_depth_sub - G->pdm;
          ponded_depth_sub = S(G->pdm);
          ponded_water_sub = ponded_depth_sub;
          a_runoff = a_runoff + runoff_sub;
        }
      }
      LGAR_MEASURE_POINT(CLK, 5)
      LGAR_ABLATABLE(DZDT, calc_dzdt(ponded_depth_sub);)
      LGAR_MEASURE_POINT(CLK, 6)
      LGAR_MEASURE_POINT(DUP_DZDT, ponded_depth_sub)
      LGAR_MEASURE_POINT(DUP_MB, ending_volume_sub)
      if constexpr (FUSED_WALK) {
        // (calc_dzdt touches neither depth nor theta: the walk after the sweep is still the column's mass)
        if (__builtin_expect(!post_mass_valid, 0)) post_mass = mass_balance();
        ending_volume_sub = post_mass;
      } else {
        ending_volume_sub = mass_balance();
      }
      LGAR_MEASURE_POINT(CLK, 7)
      previous_precip = precip_sub;
      ending_volume = ending_volume_sub;
      if constexpr (LATE_FORCING_SUMS) {
        a_precip = a_precip + precip_sub;
        a_pet = a_pet + ((val(pet_sub) > R(0.0)) ? pet_sub : S(R(0.0)));
      }
      a_aet = a_aet + AET_sub;
      ponded_water = ponded_water_sub;
      // GIUH, models/dpLGAR.py:292-298 and lgar/giuh.py:8-20
      // (queue entries from the ng-th on are zero and stay zero -- nothing is ever added to them and zeros shift in from above --
      // so the sum and the shift need no test against ng: the same values, eight adds and eight moves)
      if constexpr (GIUH_MEM) {
        if (giuh_live || val(runoff_sub) > R(0.0)) {
          const int ng = G->ng;
          S q[LGAR_GMAX];
#pragma unroll
          for (int i = 0; i < LGAR_GMAX; i++) q[i] = (i < ng) ? S(giuh_mem[(size_t)i * giuh_stride]) : S(R(0.0));
#pragma unroll
          for (int i = 0; i < LGAR_GMAX; i++) if (i < ng) q[i] = q[i] + (G->giuh[i] * runoff_sub);
          const S now = q[0];
          R qsum = R(0.0);
#pragma unroll
          for (int i = 0; i < LGAR_GMAX; i++) {
            const S shifted = (i + 1 < LGAR_GMAX) ? q[(i + 1 < LGAR_GMAX) ? i + 1 : i] : S(R(0.0));
            qsum += val(shifted);  // (the next sub-step's test, summed in its order)
            if (i < ng) giuh_mem[(size_t)i * giuh_stride] = val(shifted);
          }
          giuh_live = qsum > R(0.0);
          a_giuh = a_giuh + now;
          a_disch = a_disch + now;
        }
      } else {
      R qsum = R(0.0);
#pragma unroll
      for (int i = 0; i < LGAR_GMAX; i++) qsum += val(giuh_q[i]);
      if (qsum > R(0.0) || val(runoff_sub) > R(0.0)) {
#pragma unroll
        for (int i = 0; i < LGAR_GMAX; i++) if (i < G->ng) giuh_q[i] = giuh_q[i] + (G->giuh[i] * runoff_sub);
        S now = giuh_q[0];
#pragma unroll
        for (int i = 0; i < LGAR_GMAX - 1; i++) giuh_q[i] = giuh_q[i + 1];
        giuh_q[LGAR_GMAX - 1] = S(R(0.0));
        a_giuh = a_giuh + now;
        a_disch = a_disch + now;
      }
      }
      // NaN anywhere in the front table (the reference raises at the pow that produces it, physics/utils.py:17-27).
      // MODE 1: a NaN depth or theta reaches the column mass just computed, a NaN psi reaches a theta within a step.
      if constexpr (MODE == 0) {
        bool bad = false;
        for (int i = 0; i < nf; i++) bad = bad || is_nan(val(F.TH(i))) || is_nan(val(F.Z(i))) || is_nan(val(F.PS(i)));
        if (bad) status |= LGAR_ST_NAN;
      } else {
        if (is_nan(val(ending_volume_sub))) status |= LGAR_ST_NAN;
      }
      LGAR_MEASURE_POINT(CLK, 8)
    }
  }
};

}  // namespace lgar
