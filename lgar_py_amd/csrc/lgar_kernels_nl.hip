// lgar_kernels_nl.hip -- gfx950 forward / init kernels for ONE soil-layer count (compiled with -DLGAR_NL=<n>).
//
// Launch geometry: one 64-thread workgroup == one wavefront == 64 soil columns; a grid of ceil(N/64) workgroups
// (>> 256 CUs for the 1M-column configs).  One-wave workgroups keep the LDS allocation per wave, so the number of
// resident waves per CU is set by LDS (front tables) and VGPRs alone, and no barrier is ever needed: lanes never share
// data.  Occupancy by front capacity (fp32): 8 slots = 8.7 KB LDS/wave and <= 128 VGPRs -> 4 waves/SIMD, which also
// makes the 1M-column grid (16 384 waves) exactly 4 rounds of the chip's 4 096 wave slots; 16 / 32 slots serve the
// (rare) columns handed over by the capacity chain (lgar_forward_body.hpp).
//
// HBM traffic per launch (all coalesced, column-fastest): parameters 6*L*N, front state 2*(5*n_fronts+1)*N,
// scalars/totals, and per forcing step 2 loads + (number of requested series) stores per column.
#include <hip/hip_runtime.h>

#include "lgar_forward_body.hpp"
#include "lgar_host.hpp"
#include "lgar_launch.hpp"

#ifndef LGAR_NL
#error "compile with -DLGAR_NL=<number of soil layers>"
#endif

namespace lgar {

// waves per SIMD the register allocator has to make room for
#ifndef LGAR_OCC_F32_SMALL
#define LGAR_OCC_F32_SMALL 4
#endif
#ifndef LGAR_OCC_F64_SMALL
#define LGAR_OCC_F64_SMALL 2
#endif
template <typename R, int CAP> struct Occupancy {
  static constexpr int waves = (sizeof(R) == 4) ? ((CAP <= LGAR_CAP_SMALL) ? LGAR_OCC_F32_SMALL : ((CAP <= LGAR_CAP_MID) ? 2 : 1))
                                                : ((CAP <= LGAR_CAP_SMALL) ? LGAR_OCC_F64_SMALL : 1);
};

template <typename R, int NL, int CAP>
__global__ __launch_bounds__(WAVE) void lgar_init_kernel(KArgs<R> a) {
  __shared__ WaveLDS<R, CAP> lds;
  const int lane = threadIdx.x;
  const size_t c = (size_t)blockIdx.x * WAVE + lane;
  if (c >= (size_t)a.N) return;
  init_lane<R, NL, CAP>((const LGAR_KARG KArgs<R> *)__builtin_amdgcn_kernarg_segment_ptr(), c, lane, lds);
}

// LDS tables through which cooperating lanes exchange trapezoid heads and nodes (lgar_device.hpp geff_nodes_cooperative,
// geff_mixed_core<true>), sweep thetas and mass evaluations: one per group of lanes, in the cooperating-lanes kernels only
// (MODE 4: native double-precision trapezoid, MODE 6: mixed-precision trapezoid).  With the front table kept once per group as
// well, a 32-front wave of such a kernel takes 34 KB of LDS (front table 17 KB, these tables 16.6 KB, the groups' sums): four
// waves per CU, one per SIMD -- all a cooperating job may have.
template <typename R, int MODE> struct CoopLDS {
  static constexpr bool on = (sizeof(R) == 8) && coop_mode(MODE);
  R tab[on ? LGAR_COOP_GROUPS : 1][on ? LGAR_COOP_TAB_ROW : 1];
};

template <typename R, int NL, int CAP, int MODE>
__global__ __launch_bounds__(WAVE, (Occupancy<R, CAP>::waves)) void lgar_forward_kernel(KArgs<R> a) {
  __shared__ ForwardLDS<R, CAP, MODE> lds;
  __shared__ CoopLDS<R, MODE> coop_lds;
  const int lane = threadIdx.x;
  // the argument block is read in place (kernarg segment), see LGAR_KARG in lgar_device.hpp
  const LGAR_KARG KArgs<R> *ap = (const LGAR_KARG KArgs<R> *)__builtin_amdgcn_kernarg_segment_ptr();
  const size_t N = (size_t)ap->N;
  unsigned *ticket = ap->ticket;
  if (ap->pending_in != nullptr && *ap->pending_in == 0u) return;  // no column was handed over to this kernel
  // With a ticket counter: persistent waves.  The grid is one wave per wave slot of the chip; each pulls 64-column blocks
  // from the counter until none is left, so no round of the grid is partially filled whatever the column count.
  // Without: one workgroup per block.
  // cooperating lanes: `coop` adjacent lanes per column (4..64; 1 = every lane its own column), 64 / coop columns per wave
  const int coop = CoopLDS<R, MODE>::on ? ap->coop : 1;
  const unsigned cpb = (unsigned)(WAVE / coop);
  const unsigned nblocks = (unsigned)((N + cpb - 1) / cpb);
  for (bool first = true;; first = false) {
    unsigned blk = blockIdx.x;
    if (ticket != nullptr) {
      if (lane == 0) blk = atomicAdd(ticket, 1u);
      blk = __builtin_amdgcn_readfirstlane(blk);
      if (blk >= nblocks) break;
    } else if (!first) {
      break;
    }
    if constexpr (CoopLDS<R, MODE>::on) {
      if (coop > 1) {
        // my group and my place in it; the lanes left over when coop does not divide 64 join the last group
        int group = lane / coop;
        group = group < (int)cpb ? group : (int)cpb - 1;
        const int rank = lane - group * coop;
        const size_t c0 = (size_t)blk * cpb + group;
        const bool live = c0 < N;
        // (a wave of at most 8 columns leaves every group two rows of the exchange table: Column::calc_dzdt_pairs)
        const int rows = (cpb <= LGAR_COOP_GROUPS / 2) ? 2 : 1;
        forward_lane<R, NL, CAP, MODE>(ap, live ? c0 : N - 1, live, lane, lds, rank == 0, &coop_lds.tab[group * rows][0], group, rank);
        continue;
      }
    }
    const size_t c0 = (size_t)blk * WAVE + lane;
    const bool live = c0 < N;
    forward_lane<R, NL, CAP, MODE>(ap, live ? c0 : N - 1, live, lane, lds);
  }
}

template <typename R>
static KArgs<R> make_args(const LgarDims *d, const LgarParams *p, LgarState *s, const LgarForcing *f, const LgarStepOut *o,
                          int32_t *status) {
  KArgs<R> a;
  a.N = d->n_columns;
  a.T = d->n_steps;
  a.F = front_slots(d);
  a.Nf = forcing_columns(d);
  a.Fg = forcing_group(d);
  a.coop = 1;
  a.ticket = nullptr;
  a.pending_in = nullptr;
  a.pending_out = nullptr;
  a.chain_first = a.chain_last = 1;
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
  a.counters = o ? (unsigned long long *)o->counters : nullptr;
  a.call_sums = o ? (R *)o->call_sums : nullptr;
  a.status = status;
  a.G = make_glob<R>(d);
  return a;
}

template <typename R, int NL>
static int init_typed(const LgarDims *dims, const LgarParams *params, LgarState *state, int32_t *status, hipStream_t st) {
  const unsigned grid = (unsigned)((dims->n_columns + WAVE - 1) / WAVE);
  KArgs<R> a = make_args<R>(dims, params, state, nullptr, nullptr, status);
  hipLaunchKernelGGL((lgar_init_kernel<R, NL, LGAR_CAP_SMALL>), dim3(grid), dim3(WAVE), 0, st, a);
  return launch_status();
}

template <typename R, int NL, int CAP, int MODE>
static void launch_forward_kernel(KArgs<R> &a, unsigned nblocks, unsigned *ticket, hipStream_t st) {
  a.ticket = ticket;
  if (!CoopLDS<R, MODE>::on) a.coop = 1;
  const unsigned cpb = (unsigned)(WAVE / a.coop);  // columns per wavefront
  nblocks = (unsigned)(((size_t)a.N + cpb - 1) / cpb);
  unsigned grid = nblocks;
  if (ticket != nullptr) {
    const unsigned slots = wave_slots(Occupancy<R, CAP>::waves);
    grid = nblocks < slots ? nblocks : slots;
  }
  hipLaunchKernelGGL((lgar_forward_kernel<R, NL, CAP, MODE>), dim3(grid), dim3(WAVE), 0, st, a);
}

// fast-mode kernel of capacity CAP: MODE 1, or -- double precision with LgarDims.geff_mode = 1 -- MODE 3 (mixed-precision
// trapezoid, lgar_device.hpp geff_mixed)
template <typename R, int NL, int CAP>
static void launch_fast_kernel(KArgs<R> &a, unsigned nblocks, unsigned *ticket, hipStream_t st, bool mixed) {
  if constexpr (sizeof(R) == 8) {
    if (mixed) {
      if (a.coop > 1) {  // cooperating lanes with the mixed-precision trapezoid: the 32-front kernel only, as MODE 4
        if constexpr (CAP == LGAR_FMAX) launch_forward_kernel<R, NL, CAP, 6>(a, nblocks, ticket, st);
        return;
      }
      launch_forward_kernel<R, NL, CAP, 3>(a, nblocks, ticket, st);
      return;
    }
  }
#if defined(LGAR_MEASURE) && defined(LGAR_ONLY_MIXED)  // measurement builds that compile the mixed-precision kernels only
  (void)nblocks; (void)ticket; (void)st;
#else
  if constexpr (sizeof(R) == 8) {
    if (a.coop > 1) {  // cooperating lanes: the 32-front kernel, whose LDS is per group of lanes (forward_typed)
      if constexpr (CAP == LGAR_FMAX) launch_forward_kernel<R, NL, CAP, 4>(a, nblocks, ticket, st);
      return;
    }
  }
  (void)mixed;
  launch_forward_kernel<R, NL, CAP, 1>(a, nblocks, ticket, st);
#endif
}

// The front-capacity chain of one lgar_forward call (see lgar_forward_body.hpp).
template <typename R, int NL>
static int forward_typed(const LgarDims *dims, const LgarParams *params, LgarState *state, const LgarForcing *forcing,
                         const LgarStepOut *out, int32_t *status, hipStream_t st) {
  const unsigned grid = (unsigned)((dims->n_columns + WAVE - 1) / WAVE);
  KArgs<R> a = make_args<R>(dims, params, state, forcing, out, status);
  const int slots = a.F;
  unsigned *tickets = state->tickets;
  if (tickets != nullptr && hipMemsetAsync(tickets, 0, LGAR_NTICKETS * sizeof(unsigned), st) != hipSuccess) return LGAR_E_LAUNCH;
#if !(defined(LGAR_MEASURE) && (defined(LGAR_ONLY_MIXED) || defined(LGAR_ONLY_F32)))
  if (dims->search_mode == 0) {
    // the reference's literal searches: verification mode, one kernel at the full capacity
    launch_forward_kernel<R, NL, LGAR_FMAX, 0>(a, grid, tickets, st);
    return launch_status();
  }
#endif
  // smallest capacity that leaves room for a forcing step (one front per layer + one new front per sub-step + slack);
  // small jobs (under one wave per SIMD) gain nothing from occupancy and start at the full capacity
  const int need = NL + dims->num_subcycles + 2;
  // Jobs that cannot fill the chip (the reference's own use is ONE column, agents/DifferentiableLGAR.py:117-125): in double
  // precision every column gets 4..64 cooperating lanes that split the Geff trapezoid's nodes, the pows that open it and the
  // front sweep's independent evaluations (lgar_device.hpp geff_nodes_cooperative / geff_ends_cooperative / coop_sweep_thetas /
  // calc_dzdt_pairs).  Results are bit for bit those of one lane per column.  Such a job runs the 32-front kernel directly
  // (MODE 4: front table and exchange table once per GROUP of lanes, 34 KB of LDS per wave, one wave per SIMD): no capacity
  // chain, no hand-over.
  a.coop = cooperating_lanes<R>(dims, wave_slots(1));
  const bool tiny = (grid <= 1024u && dims->search_mode != 2) || a.coop > 1;  // search_mode 2: chain forced (tests)
  int caps[3], nc = 0;
  if (!tiny && need <= LGAR_CAP_SMALL && slots > LGAR_CAP_SMALL) caps[nc++] = LGAR_CAP_SMALL;
  if (!tiny && need <= LGAR_CAP_MID && slots > LGAR_CAP_MID) caps[nc++] = LGAR_CAP_MID;
  caps[nc++] = LGAR_FMAX;
  for (int i = 0; i < nc; i++) {
    a.chain_first = (i == 0);
    a.chain_last = (i == nc - 1);
    unsigned *tk = tickets ? tickets + i : nullptr;
    a.pending_in = (tickets && i > 0) ? tickets + 4 + (i - 1) : nullptr;   // tickets[4..5]: columns handed over by kernel 0, 1
    a.pending_out = (tickets && i < nc - 1) ? tickets + 4 + i : nullptr;
    const bool mixed = dims->geff_mode == 1;
    switch (caps[i]) {
      case LGAR_CAP_SMALL: launch_fast_kernel<R, NL, LGAR_CAP_SMALL>(a, grid, tk, st, mixed); break;
      case LGAR_CAP_MID: launch_fast_kernel<R, NL, LGAR_CAP_MID>(a, grid, tk, st, mixed); break;
      default: launch_fast_kernel<R, NL, LGAR_FMAX>(a, grid, tk, st, mixed); break;
    }
    const int rc = launch_status();
    if (rc) return rc;
  }
  return 0;
}

template <int NL>
int launch_init_nl(const LgarDims *dims, const LgarParams *params, LgarState *state, int32_t *status, int dtype, hipStream_t st) {
  if (dtype == LGAR_F64) return init_typed<double, NL>(dims, params, state, status, st);
  if (dtype == LGAR_F32) return init_typed<float, NL>(dims, params, state, status, st);
  return LGAR_E_ARG;
}
template <int NL>
int launch_forward_nl(const LgarDims *dims, const LgarParams *params, LgarState *state, const LgarForcing *forcing,
                      const LgarStepOut *out, int32_t *status, int dtype, hipStream_t st) {
#if !(defined(LGAR_MEASURE) && defined(LGAR_ONLY_F32))  // (measurement builds of the fp32 kernels alone)
  if (dtype == LGAR_F64) return forward_typed<double, NL>(dims, params, state, forcing, out, status, st);
#endif
  if (dtype == LGAR_F32) return forward_typed<float, NL>(dims, params, state, forcing, out, status, st);
  return LGAR_E_ARG;
}

#ifdef LGAR_MEASURE
// measurement builds: read (and zero) the debug counters of this translation unit (lgar_measure.hpp)
extern "C" int lgar_debug_counters(unsigned long long *out, int reset) {
  unsigned long long z[8] = {0, 0, 0, 0, 0, 0, 0, 0};
  if (hipMemcpyFromSymbol(out, HIP_SYMBOL(lgar_dbg_counters), sizeof(z)) != hipSuccess) return -1;
  if (reset && hipMemcpyToSymbol(HIP_SYMBOL(lgar_dbg_counters), z, sizeof(z)) != hipSuccess) return -1;
  return 0;
}
#endif

#if defined(LGAR_MEASURE) && defined(LGAR_CLOCKS)
extern "C" int lgar_debug_clocks(unsigned long long *out, int reset) {
  unsigned long long z[64] = {0};
  if (hipMemcpyFromSymbol(out, HIP_SYMBOL(lgar_dbg_clk), sizeof(z)) != hipSuccess) return -1;
  if (reset && hipMemcpyToSymbol(HIP_SYMBOL(lgar_dbg_clk), z, sizeof(z)) != hipSuccess) return -1;
  return 0;
}
#endif

template int launch_init_nl<LGAR_NL>(const LgarDims *, const LgarParams *, LgarState *, int32_t *, int, hipStream_t);
template int launch_forward_nl<LGAR_NL>(const LgarDims *, const LgarParams *, LgarState *, const LgarForcing *,
                                        const LgarStepOut *, int32_t *, int, hipStream_t);

}  // namespace lgar
