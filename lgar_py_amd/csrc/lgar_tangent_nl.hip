// lgar_tangent_nl.hip -- forward-mode tangent kernels for ONE soil-layer count (compiled with -DLGAR_NL=<n>); see
// lgar_tangent_body.hpp.
#include <hip/hip_runtime.h>

#include "lgar_host.hpp"
#include "lgar_launch.hpp"
#include "lgar_tangent_body.hpp"

#ifndef LGAR_NL
#error "compile with -DLGAR_NL=<number of soil layers>"
#endif

namespace lgar {

template <typename R, int NL, int CAP, int MODE> __global__ __launch_bounds__(WAVE) void lgar_tangent_kernel(TArgs<R> a) {
  __shared__ WaveLDS<Dual<R>, CAP, 1> lds;
  __shared__ R xchg[5][WAVE];  // tangent_share: K and the four tangent coefficients of a trapezoid node, one node per lane (lgar_dual.hpp)
  const int lane = threadIdx.x;
  const size_t c = (size_t)blockIdx.x * WAVE + lane;
  if (c >= (size_t)a.N) return;
  tangent_lane<R, NL, CAP, MODE>((const LGAR_KARG TArgs<R> *)__builtin_amdgcn_kernarg_segment_ptr(), c, lane, lds, &xchg[0][0]);
}

template <typename R, int NL, int CAP, int MODE> static void launch_one(const TArgs<R> &a, unsigned grid, hipStream_t st) {
  // the 32-slot dual-number table of a wave exceeds the 64 KiB default dynamic-LDS window only for double
  hipLaunchKernelGGL((lgar_tangent_kernel<R, NL, CAP, MODE>), dim3(grid), dim3(WAVE), 0, st, a);
}

template <typename R, int NL>
static int tangent_typed(const LgarDims *dims, const LgarParams *params, const LgarParams *direction, const LgarForcing *forcing,
                         const void *w_runoff, const void *w_perc, void *grad_out, void *tangent_runoff, int32_t *status,
                         hipStream_t st) {
  const unsigned grid = (unsigned)((dims->n_columns + WAVE - 1) / WAVE);
  TArgs<R> a{dims->n_columns, dims->n_steps, forcing_columns(dims), forcing_group(dims), dims->tangent_share, 1, 1, (const R *)params->alpha, (const R *)params->n, (const R *)params->ksat,
             (const R *)params->theta_e, (const R *)params->theta_r, (const R *)params->thickness,
             (const R *)direction->alpha, (const R *)direction->n, (const R *)direction->ksat,
             (const R *)forcing->precip, (const R *)forcing->pet, (const R *)w_runoff, (const R *)w_perc,
             (R *)grad_out, (R *)tangent_runoff, status, make_glob<R>(dims)};
  if (dims->search_mode == 0) {
    launch_one<R, NL, LGAR_FMAX, 0>(a, grid, st);
    return launch_status();
  }
  const bool chain = (NL + dims->num_subcycles + 2 <= LGAR_CAP_SMALL) && (grid > 1024u || dims->search_mode == 2);
  if (chain) {
    a.chain_first = 1; a.chain_last = 0;
    launch_one<R, NL, LGAR_CAP_SMALL, 1>(a, grid, st);
    int rc = launch_status();
    if (rc) return rc;
    a.chain_first = 0; a.chain_last = 1;
  }
  launch_one<R, NL, LGAR_FMAX, 1>(a, grid, st);
  return launch_status();
}

template <int NL>
int launch_tangent_nl(const LgarDims *dims, const LgarParams *params, const LgarParams *direction, const LgarForcing *forcing,
                      const void *w_runoff, const void *w_perc, void *grad_out, void *tangent_runoff, int32_t *status,
                      int dtype, hipStream_t st) {
  if (dtype == LGAR_F64)
    return tangent_typed<double, NL>(dims, params, direction, forcing, w_runoff, w_perc, grad_out, tangent_runoff, status, st);
  if (dtype == LGAR_F32)
    return tangent_typed<float, NL>(dims, params, direction, forcing, w_runoff, w_perc, grad_out, tangent_runoff, status, st);
  return LGAR_E_ARG;
}

template int launch_tangent_nl<LGAR_NL>(const LgarDims *, const LgarParams *, const LgarParams *, const LgarForcing *,
                                        const void *, const void *, void *, void *, int32_t *, int, hipStream_t);

}  // namespace lgar
