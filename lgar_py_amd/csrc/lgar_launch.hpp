// lgar_launch.hpp -- host-side launch entry points, one set per soil-layer count.
//
// The column physics is templated on the number of layers (per-layer parameters live in registers, loops over layers are
// unrolled), so each layer count is its own translation unit: lgar_kernels_nl.hip / lgar_tangent_nl.hip are compiled once
// per NL in LGAR_LMIN..LGAR_LMAX with -DLGAR_NL=<n> (lgar_py_amd/build.py), concurrently.  lgar_kernels.hip holds the
// C-ABI and dispatches on dims->n_layers.
#pragma once
#include <hip/hip_runtime.h>

#include "../../include/lgar.h"

namespace lgar {

// dtype: LGAR_F32 / LGAR_F64.  Returns 0 or LGAR_E_*.
template <int NL> int launch_init_nl(const LgarDims *, const LgarParams *, LgarState *, int32_t *status, int dtype, hipStream_t);
template <int NL> int launch_forward_nl(const LgarDims *, const LgarParams *, LgarState *, const LgarForcing *, const LgarStepOut *,
                                        int32_t *status, int dtype, hipStream_t);
template <int NL> int launch_tangent_nl(const LgarDims *, const LgarParams *, const LgarParams *direction, const LgarForcing *,
                                        const void *w_runoff, const void *w_perc, void *grad_out, void *tangent_runoff,
                                        int32_t *status, int dtype, hipStream_t, unsigned *tickets);

#define LGAR_DECLARE_NL(NL)                                                                                                    \
  extern template int launch_init_nl<NL>(const LgarDims *, const LgarParams *, LgarState *, int32_t *, int, hipStream_t);      \
  extern template int launch_forward_nl<NL>(const LgarDims *, const LgarParams *, LgarState *, const LgarForcing *,            \
                                            const LgarStepOut *, int32_t *, int, hipStream_t);                                 \
  extern template int launch_tangent_nl<NL>(const LgarDims *, const LgarParams *, const LgarParams *, const LgarForcing *,     \
                                            const void *, const void *, void *, void *, int32_t *, int, hipStream_t, unsigned *);
LGAR_DECLARE_NL(2)
LGAR_DECLARE_NL(3)
LGAR_DECLARE_NL(4)
LGAR_DECLARE_NL(5)
LGAR_DECLARE_NL(6)
#undef LGAR_DECLARE_NL

}  // namespace lgar
