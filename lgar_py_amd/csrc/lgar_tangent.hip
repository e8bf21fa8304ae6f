// lgar_tangent.hip -- forward-mode tangent of the LGAR column integration (the differentiable path).
//
// Same device physics as lgar_kernels.hip, instantiated with Dual<R> (lgar_dual.hpp).  One launch integrates
// value + tangent for ONE parameter direction applied to every column's own parameters (columns are
// independent, so a one-hot direction over (alpha|n|Ksat, layer) yields every column's partial derivative at
// once) and contracts d runoff_t / d percolation_t with the caller's weights (the incoming gradient).
// A vector-Jacobian product over the 3 x L parameters is 3 x L such launches; nothing is stored per step.
#include <hip/hip_runtime.h>

#include "lgar_dual.hpp"
#include "lgar_host.hpp"

namespace lgar {

constexpr int TANGENT_FMAX = 8;  // front slots in the tangent kernel (values + tangents double the LDS per front)

template <typename R> struct TArgs {
  int N, T;
  const R *alpha, *n, *ksat, *theta_e, *theta_r, *thick;  // [NL][N]
  const R *d_alpha, *d_n, *d_ksat;                        // [NL][N] or null
  const R *precip, *pet;                                  // [T][N]
  const R *w_runoff, *w_perc;                             // [T][N] or null
  R *grad_out;                                            // [N]
  R *tangent_runoff;                                      // [T][N] or null
  int32_t *status;                                        // [N]
  Glob<R> G;
};

template <typename R, int NL, int FMAX> __global__ __launch_bounds__(WAVE) void lgar_tangent_kernel(TArgs<R> a) {
  using S = Dual<R>;
  __shared__ S lds_f[4 * FMAX * WAVE];
  __shared__ unsigned char lds_fl[FMAX * WAVE];
  const int lane = threadIdx.x;
  const size_t c = (size_t)blockIdx.x * WAVE + lane;
  if (c >= (size_t)a.N) return;
  const size_t N = (size_t)a.N;
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
    P.ninv_m[k] = R(-1.0) / P.m[k];
    P.inv_n[k] = R(1.0) / P.n[k];
    P.cum[k] = (k == 0) ? P.thick[0] : P.cum[(k > 0) ? k - 1 : 0] + P.thick[k];
  }
  FrontsView<S> F;
  F.z = lds_f + 0 * FMAX * WAVE + lane;
  F.th = lds_f + 1 * FMAX * WAVE + lane;
  F.ps = lds_f + 2 * FMAX * WAVE + lane;
  F.dz = lds_f + 3 * FMAX * WAVE + lane;
  F.fl = lds_fl + lane;
  Column<S, NL, FMAX> col(P, a.G, F);
  col.init_state();
  R grad = R(0);
  for (int t = 0; t < a.T; t++) {
    const size_t o = (size_t)t * N + c;
    col.forward(S(a.precip[o]), S(a.pet[o]));
    if (a.w_runoff) grad += a.w_runoff[o] * col.a_runoff.d;
    if (a.w_perc) grad += a.w_perc[o] * col.a_perc.d;
    if (a.tangent_runoff) a.tangent_runoff[o] = col.a_runoff.d;
    col.drain();
  }
  a.grad_out[c] = grad;
  a.status[c] = col.status;
}

}  // namespace lgar

using namespace lgar;

template <typename R, int NL>
static void launch_tangent(const LgarDims *dims, const LgarParams *params, const LgarParams *direction,
                           const LgarForcing *forcing, const void *w_runoff, const void *w_perc, void *grad_out,
                           void *tangent_runoff, int32_t *status, hipStream_t st) {
  const unsigned grid = (unsigned)((dims->n_columns + WAVE - 1) / WAVE);
  TArgs<R> a{dims->n_columns, dims->n_steps, (const R *)params->alpha, (const R *)params->n, (const R *)params->ksat,
             (const R *)params->theta_e, (const R *)params->theta_r, (const R *)params->thickness,
             (const R *)direction->alpha, (const R *)direction->n, (const R *)direction->ksat,
             (const R *)forcing->precip, (const R *)forcing->pet, (const R *)w_runoff, (const R *)w_perc,
             (R *)grad_out, (R *)tangent_runoff, status, make_glob<R>(dims)};
  hipLaunchKernelGGL((lgar_tangent_kernel<R, NL, TANGENT_FMAX>), dim3(grid), dim3(WAVE), 0, st, a);
}

extern "C" int32_t lgar_forward_tangent(const LgarDims *dims, const LgarParams *params, const LgarParams *direction,
                                        const LgarForcing *forcing, const void *w_runoff, const void *w_perc,
                                        void *grad_out, void *tangent_runoff, int32_t *status, int32_t dtype,
                                        void *stream) {
  int rc = check_dims(dims);
  if (rc) return rc;
  if (!params || !direction || !forcing || !grad_out || !status) return LGAR_E_ARG;
  if (!params->alpha || !params->n || !params->ksat || !params->theta_e || !params->theta_r || !params->thickness)
    return LGAR_E_ARG;
  if (dims->n_steps > 0 && (!forcing->precip || !forcing->pet)) return LGAR_E_ARG;
  hipStream_t st = (hipStream_t)stream;
  if (dtype == LGAR_F64) {
    switch (dims->n_layers) {
      case 2: launch_tangent<double, 2>(dims, params, direction, forcing, w_runoff, w_perc, grad_out, tangent_runoff, status, st); break;
      case 3: launch_tangent<double, 3>(dims, params, direction, forcing, w_runoff, w_perc, grad_out, tangent_runoff, status, st); break;
      case 4: launch_tangent<double, 4>(dims, params, direction, forcing, w_runoff, w_perc, grad_out, tangent_runoff, status, st); break;
      default: return LGAR_E_ARG;
    }
  } else if (dtype == LGAR_F32) {
    switch (dims->n_layers) {
      case 2: launch_tangent<float, 2>(dims, params, direction, forcing, w_runoff, w_perc, grad_out, tangent_runoff, status, st); break;
      case 3: launch_tangent<float, 3>(dims, params, direction, forcing, w_runoff, w_perc, grad_out, tangent_runoff, status, st); break;
      case 4: launch_tangent<float, 4>(dims, params, direction, forcing, w_runoff, w_perc, grad_out, tangent_runoff, status, st); break;
      default: return LGAR_E_ARG;
    }
  } else {
    return LGAR_E_ARG;
  }
  return launch_status();
}
