// lgar_host.hpp -- host-side helpers shared by the translation units of liblgar_hip.so
#pragma once
#include <hip/hip_runtime.h>

#include "lgar_device.hpp"

namespace lgar {

template <typename R> inline Glob<R> make_glob(const LgarDims *d) {
  Glob<R> G;
  G.dt_h = (R)d->dt_h;
  G.initial_psi = (R)d->initial_psi;
  G.pdm = (R)d->ponded_depth_max;
  G.wp_psi = (R)d->wilting_point_psi;
  G.frozen = (R)d->frozen_factor;
  for (int i = 0; i < LGAR_GMAX; i++) G.giuh[i] = (i < d->n_giuh) ? (R)d->giuh[i] : R(0);
  G.nint = d->nint;
  G.nsub = d->num_subcycles;
  G.ng = d->n_giuh;
  G.bottom_mode = d->bottom_mode;
  G.closed_form = d->use_closed_form_G;
  // literal searches (mode 0) are unbounded in the reference: generous cap.  In the fast modes the depth search needs a
  // few dozen iterations when it converges at all, so a diverging column (reference: endless loop) is cut off early.
  G.iter_cap = d->iter_cap > 0 ? d->iter_cap : (d->search_mode != 0 ? 5000LL : 2000000LL);
  return G;
}

inline int front_slots(const LgarDims *d) { return d->front_slots > 0 ? d->front_slots : LGAR_FMAX; }
inline int forcing_group(const LgarDims *d) { return d->forcing_group > 1 ? d->forcing_group : 1; }
inline int forcing_columns(const LgarDims *d) {
  return d->forcing_columns > 0 ? d->forcing_columns : d->n_columns / forcing_group(d);
}

inline int check_dims(const LgarDims *d) {
  if (!d) return LGAR_E_ARG;
  if (d->n_columns <= 0 || d->n_layers < LGAR_LMIN || d->n_layers > LGAR_LMAX) return LGAR_E_ARG;
  if (d->n_giuh < 0 || d->n_giuh > LGAR_GMAX) return LGAR_E_ARG;
  if (d->nint <= 0 || d->num_subcycles <= 0 || d->n_steps < 0 || d->n_steps >= (1 << 23)) return LGAR_E_ARG;
  if (d->search_mode < 0 || d->search_mode > 2) return LGAR_E_ARG;
  if (d->front_slots < 0 || d->front_slots > LGAR_FMAX || (d->front_slots > 0 && d->front_slots < d->n_layers + 1)) return LGAR_E_ARG;
  if (d->forcing_columns < 0 || d->forcing_group < 0 || d->n_columns % forcing_group(d) != 0) return LGAR_E_ARG;
  if (d->forcing_columns > 0 && (d->n_columns / forcing_group(d)) % d->forcing_columns != 0) return LGAR_E_ARG;
  if (d->tangent_share != 0 && (d->tangent_share < 2 || d->tangent_share > 32 || d->n_columns % d->tangent_share != 0)) return LGAR_E_ARG;
  if (d->geff_mode < 0 || d->geff_mode > 1) return LGAR_E_ARG;
  if (d->forward_lanes < 0 || d->forward_lanes > 64 || (d->forward_lanes & (d->forward_lanes - 1)) != 0 || d->forward_lanes == 2)
    return LGAR_E_ARG;
  if (!(d->dt_h > 0.0)) return LGAR_E_ARG;
  return 0;
}

inline int launch_status() {
  hipError_t e = hipGetLastError();
  return e == hipSuccess ? 0 : LGAR_E_LAUNCH;
}

}  // namespace lgar
