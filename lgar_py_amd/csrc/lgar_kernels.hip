// lgar_kernels.hip -- gfx950 kernels and the C-ABI (include/lgar.h) of the many-column LGAR engine.
//
// Launch geometry: one 64-thread workgroup == one wavefront == 64 soil columns; a grid of
// ceil(N/64) workgroups (>> 256 CUs for the 1M-column configs).  One-wave workgroups keep the LDS
// allocation per wave, so the number of resident waves per CU is set by LDS (front tables) and VGPRs
// alone, and no barrier is ever needed: lanes never share data.
//
// HBM traffic per launch (all coalesced, column-fastest): parameters 6*L*N, front state 2*(5*F+1)*N,
// scalars/totals, and per forcing step 2 loads + (number of requested series) stores per column.
#include <hip/hip_runtime.h>

#include "lgar_device.hpp"
#include "lgar_host.hpp"

namespace lgar {

template <typename R> struct KArgs {
  int N, T;
  const R *alpha, *n, *ksat, *theta_e, *theta_r, *thick;  // [NL][N]
  R *depth, *theta, *psi, *k, *dzdt;                      // [FMAX][N]
  uint8_t *flags;                                         // [FMAX][N]
  int32_t *nf;                                            // [N]
  R *scalars;                                             // [NSCAL][N]
  R *totals;                                              // [NACC][N]
  const R *precip, *pet;                                  // [T][N]
  R *series[LGAR_NACC];                                   // [T][N] or null
  double *basin;                                          // [NACC][T] or null
  const R *weights;                                       // [N] or null
  unsigned basin_mask;
  int32_t *status;                                        // [N]
  Glob<R> G;
};

template <typename S, int FMAX> struct WaveLDS {
  S f[4][FMAX][WAVE];
  unsigned char fl[FMAX][WAVE];
};

template <typename R, int NL>
__device__ __forceinline__ void load_params(const KArgs<R> &a, size_t c, ColParams<R, NL> &P) {
  const size_t N = (size_t)a.N;
#pragma unroll
  for (int k = 0; k < NL; k++) {
    P.alpha[k] = a.alpha[k * N + c];
    P.n[k] = a.n[k * N + c];
    P.ksat[k] = a.ksat[k * N + c] * a.G.frozen;  // models/dpLGAR.py:57
    P.te[k] = a.theta_e[k * N + c];
    P.tr[k] = a.theta_r[k * N + c];
    P.thick[k] = a.thick[k * N + c];
    P.m[k] = R(1.0) - (R(1.0) / P.n[k]);  // calc_m, physics/utils.py:67-69
    P.inv_m[k] = R(1.0) / P.m[k];
    P.ninv_m[k] = R(-1.0) / P.m[k];
    P.inv_n[k] = R(1.0) / P.n[k];
    P.cum[k] = (k == 0) ? P.thick[0] : P.cum[(k > 0) ? k - 1 : 0] + P.thick[k];  // GlobalParams.py:99-109
  }
}

template <typename S> __device__ __forceinline__ FrontsView<S> make_view(S *f, unsigned char *fl, int fmax, int lane) {
  FrontsView<S> F;
  F.z = f + 0 * fmax * WAVE + lane;
  F.th = f + 1 * fmax * WAVE + lane;
  F.ps = f + 2 * fmax * WAVE + lane;
  F.dz = f + 3 * fmax * WAVE + lane;
  F.fl = fl + lane;
  return F;
}

template <typename R, int NL, int FMAX>
__device__ __forceinline__ void store_state(const KArgs<R> &a, size_t c, const Column<R, NL, FMAX> &col) {
  const size_t N = (size_t)a.N;
  for (int i = 0; i < FMAX; i++) {
    const bool live = i < col.nf;
    a.depth[i * N + c] = live ? col.F.Z(i) : R(0);
    a.theta[i * N + c] = live ? col.F.TH(i) : R(0);
    a.psi[i * N + c] = live ? col.F.PS(i) : R(0);
    a.k[i * N + c] = live ? ((i < col.nf - 1) ? col.front_k(i, pick(col.P, col.F.layer(i))) : col.k_deepest) : R(0);
    a.dzdt[i * N + c] = live ? col.F.DZ(i) : R(0);
    a.flags[i * N + c] = live ? col.F.fl[i * WAVE] : (uint8_t)0;
  }
  a.nf[c] = col.nf;
  a.scalars[0 * N + c] = col.ponded_water;
  a.scalars[1 * N + c] = col.previous_precip;
  a.scalars[2 * N + c] = col.ending_volume;
#pragma unroll
  for (int i = 0; i < LGAR_GMAX; i++) a.scalars[(3 + i) * N + c] = col.giuh_q[i];
  a.status[c] = col.status;
}

// dpLGAR.set_internal_states (models/dpLGAR.py:97-147) for every column
template <typename R, int NL, int FMAX>
__global__ __launch_bounds__(WAVE) void lgar_init_kernel(KArgs<R> a) {
  __shared__ WaveLDS<R, FMAX> lds;
  const int lane = threadIdx.x;
  const size_t c = (size_t)blockIdx.x * WAVE + lane;
  if (c >= (size_t)a.N) return;
  ColParams<R, NL> P;
  load_params<R, NL>(a, c, P);
  Column<R, NL, FMAX> col(P, a.G, make_view<R>(&lds.f[0][0][0], &lds.fl[0][0], FMAX, lane));
  col.init_state();
  store_state<R, NL, FMAX>(a, c, col);
  const size_t N = (size_t)a.N;
#pragma unroll
  for (int j = 0; j < LGAR_NACC; j++) a.totals[j * N + c] = (j == 9) ? col.ending_volume : R(0);
}

__device__ __forceinline__ double wave_sum(double v) {
#pragma unroll
  for (int off = 32; off > 0; off >>= 1) v += __shfl_xor(v, off);
  return v;
}

// T x (dpLGAR.forward + MassBalance.change_mass) for every column; time loop inside the kernel.
// Lanes past the last column of a ragged tail wave integrate a copy of the last column (all 64 lanes stay active
// for the wave reductions) and store nothing.
// BASIN = false is the lean variant (no epilogue code, tail lanes exit at once); BASIN = true adds the basin epilogue.
#ifndef LGAR_WAVES_F32
#define LGAR_WAVES_F32 3
#endif
#ifndef LGAR_WAVES_F64
#define LGAR_WAVES_F64 2
#endif
template <typename R, int NL, int FMAX, bool BASIN>
__global__ __launch_bounds__(WAVE, (sizeof(R) == 4) ? LGAR_WAVES_F32 : LGAR_WAVES_F64) void lgar_forward_kernel(KArgs<R> a) {
  __shared__ WaveLDS<R, FMAX> lds;
  const int lane = threadIdx.x;
  const size_t N = (size_t)a.N;
  const size_t c0 = (size_t)blockIdx.x * WAVE + lane;
  const bool live = c0 < N;
  if (!BASIN && !live) return;
  const size_t c = live ? c0 : N - 1;
  ColParams<R, NL> P;
  load_params<R, NL>(a, c, P);
  Column<R, NL, FMAX> col(P, a.G, make_view<R>(&lds.f[0][0][0], &lds.fl[0][0], FMAX, lane));
  // state HBM -> LDS / registers
  int nf = a.nf[c];
  nf = nf < 0 ? 0 : (nf > FMAX ? FMAX : nf);
  col.nf = nf;
  for (int i = 0; i < nf; i++) {
    col.F.Z(i) = a.depth[i * N + c];
    col.F.TH(i) = a.theta[i * N + c];
    col.F.PS(i) = a.psi[i * N + c];
    col.F.DZ(i) = a.dzdt[i * N + c];
    col.F.fl[i * WAVE] = a.flags[i * N + c];
  }
  col.ponded_water = a.scalars[0 * N + c];
  col.previous_precip = a.scalars[1 * N + c];
  col.ending_volume = a.scalars[2 * N + c];
#pragma unroll
  for (int i = 0; i < LGAR_GMAX; i++) col.giuh_q[i] = a.scalars[(3 + i) * N + c];
  col.status = a.status[c];
  if (nf < NL) col.status |= LGAR_ST_STRUCT;  // not a state lgar_state_init / lgar_forward produced: column is skipped
  col.k_deepest = (nf > 0) ? a.k[(size_t)(nf - 1) * N + c] : R(0);
  col.new_front_frozen = false;
  col.drain();
  R tot[8];
#pragma unroll
  for (int j = 0; j < 8; j++) tot[j] = a.totals[j * N + c];
  double wgt = 0.0;
  if (BASIN) wgt = live ? (a.weights ? (double)a.weights[c] : 1.0) : 0.0;

  // software prefetch: the next step's forcing is requested before this step is integrated, so its HBM latency
  // hides under ~10^4 cycles of VALU work
  R precip_nx = a.T > 0 ? a.precip[c] : R(0);
  R pet_nx = a.T > 0 ? a.pet[c] : R(0);
  for (int t = 0; t < a.T; t++) {
    const size_t o = (size_t)t * N + c;
    const R precip = precip_nx;
    const R pet = pet_nx;
    if (t + 1 < a.T) {
      precip_nx = a.precip[o + N];
      pet_nx = a.pet[o + N];
    }
    col.forward(precip, pet);
    const R acc[LGAR_NACC] = {col.a_precip, col.a_pet, col.a_aet, col.a_infil, col.a_runoff,
                              col.a_perc, col.a_giuh, col.a_disch, col.ponded_water, col.ending_volume};
    if (live) {
#pragma unroll
      for (int j = 0; j < LGAR_NACC; j++)
        if (a.series[j]) a.series[j][o] = acc[j];
    }
    if (BASIN) {
      // basin aggregation in the epilogue of the step (physics/MassBalance.py:77-108 over many columns)
#pragma unroll
      for (int j = 0; j < LGAR_NACC; j++)
        if (a.basin_mask & (1u << j)) {
          const double s = wave_sum(wgt * (double)acc[j]);
          if (lane == 0) atomicAdd(&a.basin[(size_t)j * a.T + t], s);
        }
    }
#pragma unroll
    for (int j = 0; j < 8; j++) tot[j] = tot[j] + acc[j];  // MassBalance.change_mass, MassBalance.py:31-44
    col.drain();
  }

  if (!live) return;
  store_state<R, NL, FMAX>(a, c, col);
#pragma unroll
  for (int j = 0; j < 8; j++) a.totals[j * N + c] = tot[j];
  a.totals[8 * N + c] = col.ponded_water;
  a.totals[9 * N + c] = col.ending_volume;
}

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
  l.ninv_m = R(-1.0) / l.m;
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

template <typename R>
static KArgs<R> make_args(const LgarDims *d, const LgarParams *p, LgarState *s, const LgarForcing *f, const LgarStepOut *o,
                          int32_t *status) {
  KArgs<R> a;
  a.N = d->n_columns;
  a.T = d->n_steps;
  a.alpha = (const R *)p->alpha; a.n = (const R *)p->n; a.ksat = (const R *)p->ksat;
  a.theta_e = (const R *)p->theta_e; a.theta_r = (const R *)p->theta_r; a.thick = (const R *)p->thickness;
  a.depth = (R *)s->depth; a.theta = (R *)s->theta; a.psi = (R *)s->psi; a.k = (R *)s->k; a.dzdt = (R *)s->dzdt;
  a.flags = s->flags;
  a.nf = s->n_fronts;
  a.scalars = (R *)s->scalars;
  a.totals = (R *)s->totals;
  a.precip = f ? (const R *)f->precip : nullptr;
  a.pet = f ? (const R *)f->pet : nullptr;
  for (int j = 0; j < LGAR_NACC; j++) a.series[j] = o ? (R *)o->series[j] : nullptr;
  a.basin = o ? o->basin : nullptr;
  a.basin_mask = o ? o->basin_mask : 0u;
  a.weights = o ? (const R *)o->weights : nullptr;
  a.status = status;
  a.G = make_glob<R>(d);
  return a;
}

}  // namespace lgar

using namespace lgar;

template <typename R, int NL>
static void launch_init(const LgarDims *dims, const LgarParams *params, LgarState *state, int32_t *status, hipStream_t st) {
  const unsigned grid = (unsigned)((dims->n_columns + WAVE - 1) / WAVE);
  KArgs<R> a = make_args<R>(dims, params, state, nullptr, nullptr, status);
  hipLaunchKernelGGL((lgar_init_kernel<R, NL, LGAR_FMAX>), dim3(grid), dim3(WAVE), 0, st, a);
}
template <typename R, int NL>
static void launch_forward(const LgarDims *dims, const LgarParams *params, LgarState *state, const LgarForcing *forcing,
                           const LgarStepOut *out, int32_t *status, hipStream_t st) {
  const unsigned grid = (unsigned)((dims->n_columns + WAVE - 1) / WAVE);
  KArgs<R> a = make_args<R>(dims, params, state, forcing, out, status);
  if (a.basin != nullptr && a.basin_mask != 0u)
    hipLaunchKernelGGL((lgar_forward_kernel<R, NL, LGAR_FMAX, true>), dim3(grid), dim3(WAVE), 0, st, a);
  else
    hipLaunchKernelGGL((lgar_forward_kernel<R, NL, LGAR_FMAX, false>), dim3(grid), dim3(WAVE), 0, st, a);
}

#define LGAR_BY_LAYERS(R, FN, ...)                 \
  switch (dims->n_layers) {                        \
    case 2: FN<R, 2>(__VA_ARGS__); break;          \
    case 3: FN<R, 3>(__VA_ARGS__); break;          \
    case 4: FN<R, 4>(__VA_ARGS__); break;          \
    default: return LGAR_E_ARG;                    \
  }

extern "C" {

const char *lgar_version(void) { return "lgar-hip 0.1 (gfx950)"; }
int32_t lgar_fmax(void) { return LGAR_FMAX; }
int32_t lgar_lmax(void) { return LGAR_LMAX; }

int32_t lgar_state_init(const LgarDims *dims, const LgarParams *params, LgarState *state, int32_t *status, int32_t dtype,
                        void *stream) {
  int rc = check_dims(dims);
  if (rc) return rc;
  rc = check_state(params, state, status);
  if (rc) return rc;
  hipStream_t st = (hipStream_t)stream;
  if (dtype == LGAR_F64) {
    LGAR_BY_LAYERS(double, launch_init, dims, params, state, status, st)
  } else if (dtype == LGAR_F32) {
    LGAR_BY_LAYERS(float, launch_init, dims, params, state, status, st)
  } else {
    return LGAR_E_ARG;
  }
  return launch_status();
}

int32_t lgar_forward(const LgarDims *dims, const LgarParams *params, LgarState *state, const LgarForcing *forcing,
                     const LgarStepOut *out, int32_t *status, int32_t dtype, void *stream) {
  int rc = check_dims(dims);
  if (rc) return rc;
  rc = check_state(params, state, status);
  if (rc) return rc;
  if (dims->n_steps == 0) return 0;  // empty run: nothing to read
  if (!forcing || !forcing->precip || !forcing->pet) return LGAR_E_ARG;
  hipStream_t st = (hipStream_t)stream;
  if (dtype == LGAR_F64) {
    LGAR_BY_LAYERS(double, launch_forward, dims, params, state, forcing, out, status, st)
  } else if (dtype == LGAR_F32) {
    LGAR_BY_LAYERS(float, launch_forward, dims, params, state, forcing, out, status, st)
  } else {
    return LGAR_E_ARG;
  }
  return launch_status();
}

int32_t lgar_leaf_batch(int32_t op, int32_t n_items, const void *x, const void *y, double z, const void *alpha,
                        const void *n, const void *ksat, const void *theta_e, const void *theta_r, int32_t nint,
                        double wilting_point_psi, void *out, int32_t dtype, void *stream) {
  if (op < 0 || op > 5 || n_items <= 0 || !x || !alpha || !n || !ksat || !theta_e || !theta_r || !out) return LGAR_E_ARG;
  if ((op == 4 || op == 5) && !y) return LGAR_E_ARG;
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
