// lgar_tangent_body.hpp -- what one lane of the forward-mode tangent kernels does (the differentiable path).
//
// Same device physics as the forward kernels, instantiated with Dual<R> (lgar_dual.hpp).  One launch integrates value +
// tangent for ONE parameter direction applied to every column's own parameters (columns are independent, so a one-hot
// direction over (alpha|n|Ksat, layer) yields every column's partial derivative at once) and contracts
// d runoff_t / d percolation_t with the caller's weights (the incoming gradient) on the fly: nothing is stored per step.
// A vector-Jacobian product over the 3 x L parameters is 3 x L such launches.
//
// Front capacity: values + tangents double the LDS per front, so the first kernel runs with LGAR_CAP_SMALL slots; a
// column that could outgrow them is flagged LGAR_ST_RESUME and the second kernel (LGAR_FMAX slots) integrates exactly
// those columns again from their fresh state (a tangent run always starts from set_internal_states).
#pragma once
#include "lgar_dual.hpp"
#include "lgar_forward_body.hpp"

namespace lgar {

template <typename R> struct TArgs {
  int N, T, Nf, Fg;  // Nf, Fg: columns of the forcing and weight arrays and group: column c reads column (c / Fg) % Nf
  int share;         // W = 2..32: each group of W consecutive columns is one soil column along W directions (LgarDims.tangent_share)
  int F;             // LgarDims.front_slots: the most fronts a column may hold, as in the forward kernels (same overflow flag)
  unsigned *ticket;  // null, or this launch's work counter (persistent waves)
  const unsigned *pending_in;  // null, or how many columns the first kernel of the chain handed over
  unsigned *pending_out;       // null, or where this kernel counts the columns it hands over
  int chain_first, chain_last;
  const R *alpha, *n, *ksat, *theta_e, *theta_r, *thick;  // [NL][N]
  const R *d_alpha, *d_n, *d_ksat;                        // [NL][N] or null
  const R *precip, *pet;                                  // [T][Nf]
  const R *w_runoff, *w_perc;                             // [T][Nf] or null
  R *grad_out;                                            // [N]
  R *tangent_runoff;                                      // [T][N] or null
  int32_t *status;                                        // [N]
  Glob<R> G;
};

template <typename R, int NL, int FMAX, int MODE>
__device__ __forceinline__ void tangent_lane(const LGAR_KARG TArgs<R> *ap, size_t c, int lane, WaveLDS<Dual<R>, FMAX, 1> &lds,
                                             R *xchg = nullptr) {
  using S = Dual<R>;
  const LGAR_KARG TArgs<R> &a = *ap;
  const size_t N = (size_t)a.N;
  if (!a.chain_first) {
    const bool mine = (a.status[c] & LGAR_ST_RESUME) != 0;
    if (any_lane(mine) == 0ull) return;
    if (!mine) return;  // (the eight lanes of a shared group are handed over together: same values, same front counts)
  }
  ColParams<S, NL> P;
#pragma unroll
  for (int k = 0; k < NL; k++) {
    const size_t o = k * N + c;
    P.alpha[k] = S(a.alpha[o], a.d_alpha ? a.d_alpha[o] : R(0));
    P.n[k] = S(a.n[o], a.d_n ? a.d_n[o] : R(0));
    P.ksat[k] = S(a.ksat[o], a.d_ksat ? a.d_ksat[o] : R(0)) * a.G.frozen;  // models/dpLGAR.py:57
    P.te[k] = S(a.theta_e[o]);
    P.tr[k] = S(a.theta_r[o]);
    P.thick[k] = S(a.thick[o]);
    P.m[k] = R(1.0) - (R(1.0) / P.n[k]);
    P.inv_m[k] = R(1.0) / P.m[k];
    P.inv_n[k] = R(1.0) / P.n[k];
    P.cum[k] = (k == 0) ? P.thick[0] : P.cum[(k > 0) ? k - 1 : 0] + P.thick[k];
  }
  Column<S, NL, FMAX, MODE> col(P, &ap->G, make_view<S, FMAX>(&lds.f[0][0][0], &lds.fl[0][0], lane));
  col.share_lanes = (xchg != nullptr) ? a.share : 0;
  col.xchg = xchg;
  col.cap = FMAX < a.F ? FMAX : a.F;
  col.init_state();
  R grad = R(0);
  bool handed_over = false;
  const size_t Nf = (size_t)a.Nf;
  const size_t cf = (Nf == N) ? c : (c / (size_t)a.Fg) % Nf;
  // One wave per SIMD: nothing else covers a load's latency, so a step's forcing and weights are loaded one step AHEAD (four
  // values in registers across a step; they are taken into their registers before the step's optional store is issued --
  // loads and stores share one counter).
  R precip_ahead = R(0), pet_ahead = R(0), wr_ahead = R(0), wp_ahead = R(0);
  if (a.T > 0) {
    precip_ahead = a.precip[cf];
    pet_ahead = a.pet[cf];
    if (a.w_runoff) wr_ahead = a.w_runoff[cf];
    if (a.w_perc) wp_ahead = a.w_perc[cf];
  }
  for (int t = 0; t < a.T; t++) {
    if (!a.chain_last && col.nf + a.G.nsub > FMAX) {
      handed_over = true;  // could outgrow this kernel's front capacity: the next kernel redoes this column
      break;
    }
    const R precip = precip_ahead, pet = pet_ahead, wr = wr_ahead, wp = wp_ahead;
    {
      const size_t o = (size_t)((t + 1 < a.T) ? t + 1 : t) * Nf + cf;
      precip_ahead = a.precip[o];
      pet_ahead = a.pet[o];
      if (a.w_runoff) wr_ahead = a.w_runoff[o];
      if (a.w_perc) wp_ahead = a.w_perc[o];
    }
    col.forward(S(precip), S(pet));
    settle_load(precip_ahead); settle_load(pet_ahead); settle_load(wr_ahead); settle_load(wp_ahead);
    if (a.w_runoff) grad += wr * col.a_runoff.d;
    if (a.w_perc) grad += wp * col.a_perc.d;
    if (a.tangent_runoff) a.tangent_runoff[(size_t)t * N + c] = col.a_runoff.d;
    col.drain();
  }
  a.grad_out[c] = handed_over ? R(0) : grad;
  a.status[c] = handed_over ? (col.status | LGAR_ST_RESUME) : col.status;
  if (a.pending_out != nullptr) {
    const unsigned long long m = any_lane(handed_over);
    if (m != 0ull && first_active_lane()) atomic_add(a.pending_out, (unsigned)__builtin_popcountll(m));
  }
}

}  // namespace lgar
