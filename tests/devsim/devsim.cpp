// devsim.cpp -- TEST INFRASTRUCTURE ONLY: the device code of lgar_py_amd/csrc compiled for the host.
//
// The column physics (lgar_device.hpp), the per-lane kernel bodies (lgar_forward_body.hpp, lgar_tangent_body.hpp) and the
// front-capacity chain are plain C++ templates; with -DLGAR_DEVSIM the few GPU intrinsics they use map to libm and
// wave-level operations degenerate to a single lane.  This lets the CPU test suite (-m "not gpu") run the SAME source the
// GPU executes against the reference's golden vectors, so logic errors surface without a GPU.  It is not a fallback: the
// product package (lgar_py_amd) never builds, loads or references this file, and nothing here is shipped or timed.
#define LGAR_DEVSIM 1
#include <cmath>
#include <cstdint>
#include <cstring>
#include <vector>

#define __device__
#define __host__
#define __global__
#define __forceinline__ inline __attribute__((always_inline))

#include "../../lgar_py_amd/csrc/lgar_forward_body.hpp"
#include "../../lgar_py_amd/csrc/lgar_tangent_body.hpp"

using namespace lgar;

namespace {

template <typename R> Glob<R> make_glob(const LgarDims *d) {
  Glob<R> G;
  G.dt_h = (R)d->dt_h; G.initial_psi = (R)d->initial_psi; G.pdm = (R)d->ponded_depth_max;
  G.wp_psi = (R)d->wilting_point_psi; G.frozen = (R)d->frozen_factor;
  for (int i = 0; i < LGAR_GMAX; i++) G.giuh[i] = (i < d->n_giuh) ? (R)d->giuh[i] : R(0);
  G.nint = d->nint; G.nsub = d->num_subcycles; G.ng = d->n_giuh;
  G.bottom_mode = d->bottom_mode; G.closed_form = d->use_closed_form_G;
  G.iter_cap = d->iter_cap > 0 ? d->iter_cap : (d->search_mode != 0 ? 5000LL : 2000000LL);
  return G;
}

template <typename R>
KArgs<R> make_args(const LgarDims *d, const LgarParams *p, LgarState *s, const LgarForcing *f, const LgarStepOut *o, int32_t *status) {
  KArgs<R> a;
  a.N = d->n_columns; a.T = d->n_steps; a.F = d->front_slots > 0 ? d->front_slots : LGAR_FMAX;
  a.Fg = d->forcing_group > 1 ? d->forcing_group : 1;
  a.Nf = d->forcing_columns > 0 ? d->forcing_columns : d->n_columns / a.Fg;
  a.chain_first = a.chain_last = 1;
  a.ticket = nullptr;
  a.pending_in = nullptr;
  a.pending_out = nullptr;
  a.alpha = (const R *)p->alpha; a.n = (const R *)p->n; a.ksat = (const R *)p->ksat;
  a.theta_e = (const R *)p->theta_e; a.theta_r = (const R *)p->theta_r; a.thick = (const R *)p->thickness;
  a.depth = (R *)s->depth; a.theta = (R *)s->theta; a.psi = (R *)s->psi; a.k = (R *)s->k; a.dzdt = (R *)s->dzdt;
  a.flags = s->flags; a.nf = s->n_fronts; a.scalars = (R *)s->scalars; a.totals = (R *)s->totals;
  a.precip = f ? (const R *)f->precip : nullptr; a.pet = f ? (const R *)f->pet : nullptr;
  for (int j = 0; j < LGAR_NACC; j++) a.series[j] = o ? (R *)o->series[j] : nullptr;
  a.basin = o ? o->basin : nullptr; a.basin_mask = o ? o->basin_mask : 0u; a.weights = o ? (const R *)o->weights : nullptr;
  a.counters = o ? (unsigned long long *)o->counters : nullptr;
  a.call_sums = o ? (R *)o->call_sums : nullptr;
  a.status = status; a.G = make_glob<R>(d);
  return a;
}

template <typename R, int NL, int CAP, int MODE> void run_forward(const KArgs<R> &a) {
  std::vector<WaveLDS<R, CAP>> lds(1);
  for (int c = 0; c < a.N; c++) forward_lane<R, NL, CAP, MODE>(&a, (size_t)c, true, 0, lds[0]);
}

template <typename R, int NL>
int forward_typed(const LgarDims *d, const LgarParams *p, LgarState *s, const LgarForcing *f, const LgarStepOut *o, int32_t *status) {
  KArgs<R> a = make_args<R>(d, p, s, f, o, status);
  if (d->search_mode == 0) { run_forward<R, NL, LGAR_FMAX, 0>(a); return 0; }
  // same chain selection as lgar_kernels_nl.hip (search_mode 2 forces it; the simulator has no notion of a tiny grid)
  const int need = NL + d->num_subcycles + 2;
  const bool chain = d->search_mode == 2;
  int caps[3], nc = 0;
  if (chain && need <= LGAR_CAP_SMALL && a.F > LGAR_CAP_SMALL) caps[nc++] = LGAR_CAP_SMALL;
  if (chain && need <= LGAR_CAP_MID && a.F > LGAR_CAP_MID) caps[nc++] = LGAR_CAP_MID;
  caps[nc++] = LGAR_FMAX;
  for (int i = 0; i < nc; i++) {
    a.chain_first = (i == 0); a.chain_last = (i == nc - 1);
    if constexpr (sizeof(R) == 8) {
      if (d->geff_mode == 1) {  // mixed-precision trapezoid (MODE 3)
        if (caps[i] == LGAR_CAP_SMALL) run_forward<R, NL, LGAR_CAP_SMALL, 3>(a);
        else if (caps[i] == LGAR_CAP_MID) run_forward<R, NL, LGAR_CAP_MID, 3>(a);
        else run_forward<R, NL, LGAR_FMAX, 3>(a);
        continue;
      }
    }
    if (caps[i] == LGAR_CAP_SMALL) run_forward<R, NL, LGAR_CAP_SMALL, 1>(a);
    else if (caps[i] == LGAR_CAP_MID) run_forward<R, NL, LGAR_CAP_MID, 1>(a);
    else run_forward<R, NL, LGAR_FMAX, 1>(a);
  }
  return 0;
}

template <typename R, int NL> int init_typed(const LgarDims *d, const LgarParams *p, LgarState *s, int32_t *status) {
  KArgs<R> a = make_args<R>(d, p, s, nullptr, nullptr, status);
  std::vector<WaveLDS<R, LGAR_CAP_SMALL>> lds(1);
  for (int c = 0; c < a.N; c++) init_lane<R, NL, LGAR_CAP_SMALL>(&a, (size_t)c, 0, lds[0]);
  return 0;
}

template <typename R, int NL, int CAP, int MODE> void run_tangent(const TArgs<R> &a) {
  std::vector<WaveLDS<Dual<R>, CAP, 1>> lds(1);
  for (int c = 0; c < a.N; c++) tangent_lane<R, NL, CAP, MODE>(&a, (size_t)c, 0, lds[0]);
}

template <typename R, int NL>
int tangent_typed(const LgarDims *d, const LgarParams *p, const LgarParams *dir, const LgarForcing *f, const void *wr, const void *wp,
                  void *grad, void *tser, int32_t *status) {
  TArgs<R> a{d->n_columns, d->n_steps, d->forcing_columns > 0 ? d->forcing_columns : d->n_columns / (d->forcing_group > 1 ? d->forcing_group : 1),
             d->forcing_group > 1 ? d->forcing_group : 1, 0 /* one lane: nothing to share */, d->front_slots > 0 ? d->front_slots : LGAR_FMAX, nullptr, nullptr, nullptr, 1, 1, (const R *)p->alpha, (const R *)p->n, (const R *)p->ksat, (const R *)p->theta_e,
             (const R *)p->theta_r, (const R *)p->thickness, (const R *)dir->alpha, (const R *)dir->n, (const R *)dir->ksat,
             (const R *)f->precip, (const R *)f->pet, (const R *)wr, (const R *)wp, (R *)grad, (R *)tser, status, make_glob<R>(d)};
  if (d->search_mode == 0) { run_tangent<R, NL, LGAR_FMAX, 0>(a); return 0; }
  if (d->search_mode == 2 && NL + d->num_subcycles + 2 <= LGAR_CAP_SMALL) {
    a.chain_first = 1; a.chain_last = 0;
    run_tangent<R, NL, LGAR_CAP_SMALL, 1>(a);
    a.chain_first = 0; a.chain_last = 1;
  }
  run_tangent<R, NL, LGAR_FMAX, 1>(a);
  return 0;
}

}  // namespace

#ifndef DEVSIM_LAYERS
#define DEVSIM_LAYERS(X) X(2) X(3) X(4) X(5) X(6)
#endif

extern "C" {

// element-wise access to the lean fp64 math (lgar_math.hpp): op 0 exp2, 1 log2, 2 pow(x, y), 3 exp2_core, 4 log2_core;
// 10..14: the same with the polynomials' high-order terms combined pairwise (ESTRIN: the mixed-precision kernels)
void devsim_math(int op, int n, const double *x, const double *y, double *out) {
  for (int i = 0; i < n; i++) {
    switch (op) {
      case 0: out[i] = fast_exp2(x[i]); break;
      case 1: out[i] = fast_log2(x[i]); break;
      case 2: out[i] = fast_pow(x[i], y[i]); break;
      case 3: out[i] = fast_exp2_core<false>(x[i]); break;
      case 4: out[i] = fast_log2_core(x[i]); break;
      case 10: out[i] = fast_exp2<true>(x[i]); break;
      case 11: out[i] = fast_log2<true>(x[i]); break;
      case 12: out[i] = fast_pow<true>(x[i], y[i]); break;
      case 13: out[i] = fast_exp2_core<false, true>(x[i]); break;
      case 14: out[i] = fast_log2_core<true>(x[i]); break;
    }
  }
}

// element-wise Geff variants on host doubles: 0 fused fp64 (fast modes), 1 mixed precision (geff_mode 1), 2 the reference's
// literal trapezoid with the library pow, 3 the packed fp32 loop (inputs rounded to float)
void devsim_geff(int variant, int n, const double *theta1, const double *theta2, const double *alpha, const double *nn,
                 const double *ksat, const double *te, const double *tr, int nint, double *out) {
  for (int i = 0; i < n; i++) {
    LayerK<double> l;
    l.alpha = alpha[i]; l.n = nn[i]; l.m = 1.0 - 1.0 / l.n; l.inv_m = 1.0 / l.m; l.inv_n = 1.0 / l.n;
    l.ksat = ksat[i]; l.te = te[i]; l.tr = tr[i];
    if (variant == 0) out[i] = geff_fused<double>(l, theta1[i], theta2[i], nint);
    else if (variant == 1) out[i] = geff_mixed(l.alpha, l.n, l.m, l.inv_m, l.inv_n, l.ksat, l.te, l.tr, theta1[i], theta2[i], nint);
    else if (variant == 2) out[i] = geff_literal<double, 1>(l, theta1[i], theta2[i], nint);
    else {
      LayerK<float> f;
      f.alpha = (float)l.alpha; f.n = (float)l.n; f.m = 1.0f - 1.0f / f.n; f.inv_m = 1.0f / f.m; f.inv_n = 1.0f / f.n;
      f.ksat = (float)l.ksat; f.te = (float)l.te; f.tr = (float)l.tr;
      out[i] = (double)geff<float>(f, (float)theta1[i], (float)theta2[i], nint);
    }
  }
}

int devsim_state_init(const LgarDims *d, const LgarParams *p, LgarState *s, int32_t *status, int dtype) {
#define X(n) if (d->n_layers == n) return dtype == LGAR_F64 ? init_typed<double, n>(d, p, s, status) : init_typed<float, n>(d, p, s, status);
  DEVSIM_LAYERS(X)
#undef X
  return LGAR_E_ARG;
}

int devsim_forward(const LgarDims *d, const LgarParams *p, LgarState *s, const LgarForcing *f, const LgarStepOut *o, int32_t *status, int dtype) {
#define X(n) if (d->n_layers == n) return dtype == LGAR_F64 ? forward_typed<double, n>(d, p, s, f, o, status) : forward_typed<float, n>(d, p, s, f, o, status);
  DEVSIM_LAYERS(X)
#undef X
  return LGAR_E_ARG;
}

int devsim_tangent(const LgarDims *d, const LgarParams *p, const LgarParams *dir, const LgarForcing *f, const void *wr, const void *wp,
                   void *grad, void *tser, int32_t *status, int dtype) {
#define X(n) if (d->n_layers == n) return dtype == LGAR_F64 ? tangent_typed<double, n>(d, p, dir, f, wr, wp, grad, tser, status) \
                                                             : tangent_typed<float, n>(d, p, dir, f, wr, wp, grad, tser, status);
  DEVSIM_LAYERS(X)
#undef X
  return LGAR_E_ARG;
}

}  // extern "C"
