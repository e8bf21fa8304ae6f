// lgar_kernels.hip -- the C-ABI (include/lgar.h) of the many-column LGAR engine: argument checks and dispatch on the
// soil-layer count to the per-layer-count translation units (lgar_launch.hpp), plus the element-wise leaf kernel used by
// the known-answer tests.
#include <hip/hip_runtime.h>

#include "lgar_device.hpp"
#include "lgar_host.hpp"
#include "lgar_launch.hpp"

namespace lgar {

template <typename R> struct LeafArgs {
  int op, n, nint;
  const R *x, *y, *alpha, *nn, *ksat, *te, *tr;
  R z, wp_psi;
  R *out;
};

template <typename R> __global__ void lgar_leaf_kernel(LeafArgs<R> a) {
  const int i = blockIdx.x * blockDim.x + threadIdx.x;
  if (i >= a.n) return;
  LayerK<R> l;
  l.alpha = a.alpha[i];
  l.n = a.nn[i];
  l.m = R(1.0) - (R(1.0) / l.n);
  l.inv_m = R(1.0) / l.m;
  l.inv_n = R(1.0) / l.n;
  l.ksat = a.ksat[i];
  l.te = a.te[i];
  l.tr = a.tr[i];
  const R x = a.x[i];
  const R y = a.y ? a.y[i] : R(0);
  R r = R(0);
  switch (a.op) {
    case 0: r = theta_from_h(l, x); break;
    case 1: r = se_from_h(l, x); break;
    case 2: r = k_from_se(l, x); break;
    case 3: r = h_from_se(l, x); break;
    case 4: r = geff(l, x, y, a.nint); break;
    case 5: r = aet_fn(l, y, a.z, x, a.wp_psi); break;
    case 6: r = geff_literal<R, (sizeof(R) == 8) ? 1 : 0>(l, x, y, a.nint); break;
    case 7: r = lg2p(x); break;
    case 8: r = ex2p(x); break;
    case 9: r = pw(x, y); break;
    case 10:
      if constexpr (sizeof(R) == 8) r = geff_mixed(l.alpha, l.n, l.m, l.inv_m, l.inv_n, l.ksat, l.te, l.tr, x, y, a.nint);
      else r = geff(l, x, y, a.nint);
      break;
    case 11: r = dv<0>(x, y); break;        // the fast modes' quotient (double precision: lean_div)
    case 12: r = pwp<3>(x, y); break;       // the mixed-precision kernels' pow (pairwise-combined polynomials)
    case 13:
      if constexpr (sizeof(R) == 8) r = lg2e(x);
      else r = lg2p(x);
      break;
    case 14:
      if constexpr (sizeof(R) == 8) r = ex2e(x);
      else r = ex2p(x);
      break;
  }
  a.out[i] = r;
}

static int check_state(const LgarParams *p, const LgarState *s, const int32_t *status) {
  if (!p || !s || !status) return LGAR_E_ARG;
  if (!p->alpha || !p->n || !p->ksat || !p->theta_e || !p->theta_r || !p->thickness) return LGAR_E_ARG;
  if (!s->depth || !s->theta || !s->psi || !s->k || !s->dzdt || !s->flags || !s->n_fronts || !s->scalars || !s->totals)
    return LGAR_E_ARG;
  return 0;
}

}  // namespace lgar

using namespace lgar;

#ifdef LGAR_ONLY_LAYERS  // measurement variants (build.py build_variant): only these layer counts are linked in
#define LGAR_BY_LAYERS(FN, ...)                        \
  switch (dims->n_layers) {                            \
    case 3: return FN<3>(__VA_ARGS__);                 \
    default: return LGAR_E_ARG;                        \
  }
#else
#define LGAR_BY_LAYERS(FN, ...)                        \
  switch (dims->n_layers) {                            \
    case 2: return FN<2>(__VA_ARGS__);                 \
    case 3: return FN<3>(__VA_ARGS__);                 \
    case 4: return FN<4>(__VA_ARGS__);                 \
    case 5: return FN<5>(__VA_ARGS__);                 \
    case 6: return FN<6>(__VA_ARGS__);                 \
    default: return LGAR_E_ARG;                        \
  }
#endif

// Basin aggregation of a STORED series (LgarStepOut.basin with series[j] present): basin[t] += sum_c weight[c] series[t][c]
// (physics/MassBalance.py:77-108 over many columns).  One workgroup per step; every thread sums a fixed strided subset of the
// columns in independent chains, then wave, then workgroup, each in a fixed order: the same inputs give the same bits on
// every run.  The forward kernels' own path (one fp64 atomic per wave and step, for callers that keep no series) costs 11 % of
// the fp32 launch on 1M columns -- 2.4M atomics on T addresses -- against 1 % for this pass over data the launch wrote anyway.
template <typename R> struct Vec16;  // 16-byte packets of the series
template <> struct Vec16<float> { typedef float4 type; static constexpr int n = 4; };
template <> struct Vec16<double> { typedef double2 type; static constexpr int n = 2; };
__device__ __forceinline__ void basin_add(double (&s)[4], const float4 &v) {
  s[0] += (double)v.x; s[1] += (double)v.y; s[2] += (double)v.z; s[3] += (double)v.w;
}
__device__ __forceinline__ void basin_add(double (&s)[4], const double2 &v) { s[0] += v.x; s[1] += v.y; }
__device__ __forceinline__ void basin_add(double (&s)[4], const float4 &v, const float4 &w) {
  s[0] += (double)w.x * (double)v.x; s[1] += (double)w.y * (double)v.y; s[2] += (double)w.z * (double)v.z; s[3] += (double)w.w * (double)v.w;
}
__device__ __forceinline__ void basin_add(double (&s)[4], const double2 &v, const double2 &w) { s[0] += w.x * v.x; s[1] += w.y * v.y; }

// VEC: rows are 16-byte aligned (N a multiple of the packet, base pointers aligned): every thread keeps four 16-byte loads in
// flight (a workgroup streams its 4 MB row at the rate of a CU's memory pipeline instead of one 4-byte load at a time)
template <typename R, bool VEC>
__global__ __launch_bounds__(1024) void lgar_basin_reduce_kernel(const R *series, const R *weights, double *basin, long long N) {
  __shared__ double part[16];
  const int t = blockIdx.x;
  const R *row = series + (size_t)t * (size_t)N;
  double s[4] = {0.0, 0.0, 0.0, 0.0};
  if constexpr (VEC) {
    typedef typename Vec16<R>::type V;
    const long long NV = N / Vec16<R>::n;
    const V *rv = (const V *)row;
    const V *wv = (const V *)weights;
    long long c = threadIdx.x;
    for (; c + 3072 < NV; c += 4096) {
      const V a0 = rv[c], a1 = rv[c + 1024], a2 = rv[c + 2048], a3 = rv[c + 3072];
      if (wv == nullptr) {
        basin_add(s, a0); basin_add(s, a1); basin_add(s, a2); basin_add(s, a3);
      } else {
        const V w0 = wv[c], w1 = wv[c + 1024], w2 = wv[c + 2048], w3 = wv[c + 3072];
        basin_add(s, a0, w0); basin_add(s, a1, w1); basin_add(s, a2, w2); basin_add(s, a3, w3);
      }
    }
    for (; c < NV; c += 1024) {
      if (wv == nullptr) basin_add(s, rv[c]);
      else basin_add(s, rv[c], wv[c]);
    }
  } else {
    for (long long c = threadIdx.x; c < N; c += 1024) s[0] += (weights ? (double)weights[c] : 1.0) * (double)row[c];
  }
  double v = (s[0] + s[1]) + (s[2] + s[3]);
#pragma unroll
  for (int off = 32; off > 0; off >>= 1) v += __shfl_xor(v, off);
  if ((threadIdx.x & 63u) == 0u) part[threadIdx.x >> 6] = v;
  __syncthreads();
  if (threadIdx.x == 0) {
    double tot = 0.0;
    for (int i = 0; i < 16; i++) tot += part[i];
    basin[t] += tot;
  }
}

template <typename R>
static void launch_basin_reduce(const void *series, const void *weights, double *row, long long N, int T, hipStream_t st) {
  const bool vec = (N % Vec16<R>::n == 0) && ((uintptr_t)series % 16 == 0) && ((uintptr_t)weights % 16 == 0);
  if (vec)
    hipLaunchKernelGGL((lgar_basin_reduce_kernel<R, true>), dim3((unsigned)T), dim3(1024), 0, st, (const R *)series,
                       (const R *)weights, row, N);
  else
    hipLaunchKernelGGL((lgar_basin_reduce_kernel<R, false>), dim3((unsigned)T), dim3(1024), 0, st, (const R *)series,
                       (const R *)weights, row, N);
}

static int forward_by_layers(const LgarDims *dims, const LgarParams *params, LgarState *state, const LgarForcing *forcing,
                             const LgarStepOut *out, int32_t *status, int32_t dtype, hipStream_t stream) {
  LGAR_BY_LAYERS(launch_forward_nl, dims, params, state, forcing, out, status, dtype, stream)
}

extern "C" {

const char *lgar_version(void) { return "lgar-hip 0.4 (gfx950)"; }
int32_t lgar_abi_version(void) { return LGAR_ABI_VERSION; }
int32_t lgar_sizeof_dims(void) { return (int32_t)sizeof(LgarDims); }
int32_t lgar_fmax(void) { return LGAR_FMAX; }
int32_t lgar_lmax(void) { return LGAR_LMAX; }
int32_t lgar_cooperating_lanes(const LgarDims *dims, int32_t dtype) {
  if (check_dims(dims) != 0 || (dtype != LGAR_F32 && dtype != LGAR_F64)) return LGAR_E_ARG;
  // (the function forward_typed itself calls, lgar_host.hpp)
  return dtype == LGAR_F64 ? cooperating_lanes<double>(dims, wave_slots(1)) : cooperating_lanes<float>(dims, wave_slots(1));
}

int32_t lgar_state_init(const LgarDims *dims, const LgarParams *params, LgarState *state, int32_t *status, int32_t dtype,
                        void *stream) {
  int rc = check_dims(dims);
  if (rc) return rc;
  rc = check_state(params, state, status);
  if (rc) return rc;
  LGAR_BY_LAYERS(launch_init_nl, dims, params, state, status, dtype, (hipStream_t)stream)
}

int32_t lgar_forward(const LgarDims *dims, const LgarParams *params, LgarState *state, const LgarForcing *forcing,
                     const LgarStepOut *out, int32_t *status, int32_t dtype, void *stream) {
  int rc = check_dims(dims);
  if (rc) return rc;
  rc = check_state(params, state, status);
  if (rc) return rc;
  if (dims->n_steps == 0) return 0;  // empty run: nothing to read
  if (!forcing || !forcing->precip || !forcing->pet) return LGAR_E_ARG;
  // basin sums of accumulators whose series this call stores anyway are taken from the stored series afterwards (one pass,
  // fixed summation order); the forward kernels aggregate only what has no series
  LgarStepOut kernel_out;
  uint32_t from_series = 0;
  if (out && out->basin && out->basin_mask) {
    for (int j = 0; j < LGAR_NACC; j++)
      if (((out->basin_mask >> j) & 1u) && out->series[j]) from_series |= 1u << j;
    if (from_series) {
      kernel_out = *out;
      kernel_out.basin_mask &= ~from_series;
      out = &kernel_out;
    }
  }
  rc = forward_by_layers(dims, params, state, forcing, out, status, dtype, (hipStream_t)stream);
  if (rc || !from_series) return rc;
  const long long N = dims->n_columns;
  const int T = dims->n_steps;
  for (int j = 0; j < LGAR_NACC; j++) {
    if (!((from_series >> j) & 1u)) continue;
    double *row = out->basin + (size_t)j * (size_t)T;
    if (dtype == LGAR_F64) launch_basin_reduce<double>(out->series[j], out->weights, row, N, T, (hipStream_t)stream);
    else launch_basin_reduce<float>(out->series[j], out->weights, row, N, T, (hipStream_t)stream);
  }
  return launch_status();
}

#ifndef LGAR_NO_TANGENT
int32_t lgar_forward_tangent(const LgarDims *dims, const LgarParams *params, const LgarParams *direction,
                             const LgarForcing *forcing, const void *w_runoff, const void *w_perc, void *grad_out,
                             void *tangent_runoff, int32_t *status, int32_t dtype, void *stream, uint32_t *tickets) {
  int rc = check_dims(dims);
  if (rc) return rc;
  if (!params || !direction || !forcing || !grad_out || !status) return LGAR_E_ARG;
  if (!params->alpha || !params->n || !params->ksat || !params->theta_e || !params->theta_r || !params->thickness)
    return LGAR_E_ARG;
  if (dims->n_steps > 0 && (!forcing->precip || !forcing->pet)) return LGAR_E_ARG;
  LGAR_BY_LAYERS(launch_tangent_nl, dims, params, direction, forcing, w_runoff, w_perc, grad_out, tangent_runoff, status,
                 dtype, (hipStream_t)stream, tickets)
}
#endif

int32_t lgar_leaf_batch(int32_t op, int32_t n_items, const void *x, const void *y, double z, const void *alpha,
                        const void *n, const void *ksat, const void *theta_e, const void *theta_r, int32_t nint,
                        double wilting_point_psi, void *out, int32_t dtype, void *stream) {
  if (op < 0 || op > 14 || n_items <= 0 || !x || !alpha || !n || !ksat || !theta_e || !theta_r || !out) return LGAR_E_ARG;
  if ((op == 4 || op == 5 || op == 6 || op == 10) && !y) return LGAR_E_ARG;
  const unsigned grid = (unsigned)((n_items + 255) / 256);
  hipStream_t st = (hipStream_t)stream;
  if (dtype == LGAR_F64) {
    LeafArgs<double> a{op, n_items, nint, (const double *)x, (const double *)y, (const double *)alpha, (const double *)n,
                       (const double *)ksat, (const double *)theta_e, (const double *)theta_r, z, wilting_point_psi,
                       (double *)out};
    hipLaunchKernelGGL(lgar_leaf_kernel<double>, dim3(grid), dim3(256), 0, st, a);
  } else if (dtype == LGAR_F32) {
    LeafArgs<float> a{op, n_items, nint, (const float *)x, (const float *)y, (const float *)alpha, (const float *)n,
                      (const float *)ksat, (const float *)theta_e, (const float *)theta_r, (float)z,
                      (float)wilting_point_psi, (float *)out};
    hipLaunchKernelGGL(lgar_leaf_kernel<float>, dim3(grid), dim3(256), 0, st, a);
  } else {
    return LGAR_E_ARG;
  }
  return launch_status();
}

}  // extern "C"
