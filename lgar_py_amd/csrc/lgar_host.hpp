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
  if (d->forward_lanes < 0 || d->forward_lanes > 64 || d->forward_lanes == 2 || d->forward_lanes == 3) return LGAR_E_ARG;
  if (!(d->dt_h > 0.0)) return LGAR_E_ARG;
  return 0;
}

// wave slots of the chip for a kernel compiled for `waves` waves per SIMD
inline unsigned wave_slots(int waves) {
  static int cus = 0;
  if (cus == 0) {
    int dev = 0, n = 0;
    if (hipGetDevice(&dev) != hipSuccess || hipDeviceGetAttribute(&n, hipDeviceAttributeMultiprocessorCount, dev) != hipSuccess || n <= 0)
      n = 256;
    cus = n;
  }
  return (unsigned)cus * 4u * (unsigned)waves;
}


// lanes per column for this job: LgarDims.forward_lanes when given, else as many as keep the job within ONE wave per SIMD
// (ceil(n_columns / 1024) columns per wavefront, 64 / that lanes each) -- two such waves on a SIMD contend for its vector ALU
// in the trapezoid and the gain is gone (measured: 10 000 columns x 8 lanes = 1250 waves run slower than 157 plain ones; 6
// lanes = 1000 waves).  At most 16 columns per wavefront (that many LDS tables): 4..64 lanes; always 1 for fp32, closed-form
// G, the literal mode and more than 128 trapezoid intervals.  The rule is the same for the native double-precision trapezoid
// (MODE 4 kernels) and the mixed-precision one (LgarDims.geff_mode = 1: MODE 6 kernels).
template <typename R> inline int cooperating_lanes(const LgarDims *dims, unsigned simds) {
  if (sizeof(R) != 8 || dims->search_mode == 0 || dims->use_closed_form_G) return 1;
  if (dims->nint > LGAR_COOP_TAB) return 1;  // the groups' LDS tables hold one head / node per trapezoid interval
  if (dims->forward_lanes > 0) return dims->forward_lanes;
  if (dims->search_mode == 2) return 1;      // the capacity chain was asked for (tests): plain kernels
  const size_t groups = ((size_t)dims->n_columns + simds - 1) / simds;  // columns a wavefront has to take
  return groups <= LGAR_COOP_GROUPS ? (int)(WAVE / groups) : 1;         // 64, 32, 21, 16, 12, 10, 9, 8, 7, 6, 5, 5, 4, 4, 4, 4
}


inline int launch_status() {
  hipError_t e = hipGetLastError();
  return e == hipSuccess ? 0 : LGAR_E_LAUNCH;
}

}  // namespace lgar
