// lgar_forward_body.hpp -- what ONE lane (= one soil column) of the forward / init kernels does.
//
// Kept apart from the __global__ wrappers (lgar_kernels.hip) so that the very same code can be compiled for the host by
// the test-only device-code simulator (tests/devsim, -DLGAR_DEVSIM: one lane at a time, wave reductions = identity).
//
// Front-capacity chain.  The front table of a wave lives in LDS, so a kernel is compiled for a front capacity CAP
// (LGAR_CAP_SMALL / _MID / _MAX slots per column; the reference's lists are unbounded, layers/Layer.py:1336-1416).
// lgar_forward launches the smallest capacity that fits n_layers + num_subcycles, and then the larger ones over the SAME
// arrays: a column that could outgrow its kernel's capacity during the coming forcing step (n_fronts + num_subcycles >
// CAP: a sub-step creates at most one front) stops BEFORE that step with its state stored, status = LGAR_ST_RESUME and
// the step index in the status word's upper bits; the next kernel of the chain picks exactly those columns up at that
// step (waves without such a column exit at once).  Only the last kernel of the chain can report LGAR_ST_OVERFLOW.
#pragma once
#include "lgar_device.hpp"

namespace lgar {

template <typename R> struct KArgs {
  int N, T, F;                                            // columns, forcing steps, rows of the per-front state arrays
  int Nf, Fg;                                             // forcing columns and group: column c reads forcing column (c / Fg) % Nf
  int coop;                                               // lanes per column (1, or 4..64: cooperating lanes, small jobs)
  unsigned *ticket;                                       // null, or the work counter of this launch (persistent waves)
  int chain_first, chain_last;                            // position in the capacity chain (see above)
  const unsigned *pending_in;                             // null, or how many columns the previous kernel of the chain handed over
  unsigned *pending_out;                                  // null, or where this kernel counts the columns it hands over
  const R *alpha, *n, *ksat, *theta_e, *theta_r, *thick;  // [NL][N]
  R *depth, *theta, *psi, *k, *dzdt;                      // [F][N]
  uint8_t *flags;                                         // [F][N]
  int32_t *nf;                                            // [N]
  R *scalars;                                             // [NSCAL][N]
  R *totals;                                              // [NACC][N]
  const R *precip, *pet;                                  // [T][Nf]
  R *series[LGAR_NACC];                                   // [T][N] or null
  double *basin;                                          // [NACC][T] or null
  const R *weights;                                       // [N] or null
  unsigned basin_mask;
  int32_t *status;                                        // [N]
  unsigned long long *counters;                           // [LGAR_NCOUNTERS] or null (measurement: Geff wave-calls ...)
  R *call_sums;                                           // [NACC][N] or null: accumulators summed over this call's steps
  Glob<R> G;
};

// Per-call sums of the accumulators that fit in the wave's LDS budget next to the front table.  They are touched once per
// forcing step, which makes them the register allocator's first spill victims -- and a spilled read-modify-write is
// scratch (HBM) write traffic on every step.  The 8-slot kernels have 10 KB (fp32, 4 waves/SIMD) / 20 KB (fp64, 2 waves/SIMD)
// of LDS per wave: fp32, 8.5 KB of fronts + 6 rows of sums = exactly 10 KB, percolation (zero in the reference's mode) stays
// in a register; fp64, 16.5 KB of fronts + 7 rows = exactly 20 KB (the register pair of the seventh was stored to scratch on
// every step: 8 of the 28 bytes per column and step the mixed-precision kernel still spilled in round 5).  Discharge is
// the same sum as giuh_runoff.
template <typename S, int FMAX> struct LdsSums {
  static constexpr int rows = (FMAX <= LGAR_CAP_SMALL) ? (sizeof(S) == 8 ? 7 : 6) : 8;
};

// LDS row of accumulator j's per-call sum, or -1 if it stays in a register: with 6 rows the one left out is percolation
// (with 7 it has the last row)
template <int SR> __device__ __forceinline__ constexpr int sum_row(int j) {
  return (SR >= 8) ? (j < 8 ? j : -1) : ((j < 5) ? (j < SR ? j : -1) : ((j == 6 && SR >= 6) ? 5 : ((j == 5 && SR >= 7) ? 6 : -1)));
}

template <typename S, int FMAX, int SUMROWS = LdsSums<S, FMAX>::rows, int STRIDE = WAVE> struct WaveLDS {
  S f[4][FMAX][STRIDE];
  unsigned char fl[FMAX][STRIDE];
  S sums[SUMROWS][STRIDE];
};
// the wave's LDS block of a forward kernel: one front table (and one row of sums) per lane, or (MODE 4) per group of
// cooperating lanes
template <typename R, int FMAX, int MODE>
using ForwardLDS = WaveLDS<R, FMAX, LdsSums<R, FMAX>::rows, coop_mode(MODE) ? LGAR_COOP_GROUPS : WAVE>;

#ifndef LGAR_DEVSIM
__device__ __forceinline__ double wave_sum(double v) {
#pragma unroll
  for (int off = 32; off > 0; off >>= 1) v += __shfl_xor(v, off);
  return v;
}
// Atomic adds to a WAVE-UNIFORM address, nothing returned.  Such an atomic is encoded as scalar base + a vector offset of zero;
// written in C++ the compiler keeps ONE zero register for all of them, alive from the kernel's first atomic to its last --
// across all of the column physics -- and the allocator of the 8-slot fp64 kernels kept it by storing it to scratch at the top
// of every step (8 bytes per column and step).  Here the zero is made where it is used.  (Those kernels only -- forward_lane's
// PARK: in the fp32 kernel, which does not spill the zero, the same change costs 1.7 % through register allocation, measured.)
__device__ __forceinline__ void atomic_add_uniform(double *p, double v) {
  unsigned z;
  asm volatile("v_mov_b32 %0, 0\n\tglobal_atomic_add_f64 %0, %1, %2" : "=&v"(z) : "v"(v), "s"(p) : "memory");
}
__device__ __forceinline__ void atomic_add_uniform(unsigned long long *p, unsigned long long v) {
  unsigned z;
  asm volatile("v_mov_b32 %0, 0\n\tglobal_atomic_add_x2 %0, %1, %2" : "=&v"(z) : "v"(v), "s"(p) : "memory");
}
__device__ __forceinline__ void atomic_add_uniform(unsigned *p, unsigned v) {
  unsigned z;
  asm volatile("v_mov_b32 %0, 0\n\tglobal_atomic_add %0, %1, %2" : "=&v"(z) : "v"(v), "s"(p) : "memory");
}
__device__ __forceinline__ void atomic_add(double *p, double v) { atomicAdd(p, v); }
__device__ __forceinline__ void atomic_add(unsigned long long *p, unsigned long long v) { atomicAdd(p, v); }
__device__ __forceinline__ void atomic_add(unsigned *p, unsigned v) { atomicAdd(p, v); }
#else
__device__ __forceinline__ double wave_sum(double v) { return v; }
__device__ __forceinline__ void atomic_add(double *p, double v) { *p += v; }
__device__ __forceinline__ void atomic_add_uniform(double *p, double v) { *p += v; }
__device__ __forceinline__ void atomic_add_uniform(unsigned long long *p, unsigned long long v) { *p += v; }
__device__ __forceinline__ void atomic_add_uniform(unsigned *p, unsigned v) { *p += v; }
__device__ __forceinline__ void atomic_add(unsigned long long *p, unsigned long long v) { *p += v; }
__device__ __forceinline__ void atomic_add(unsigned *p, unsigned v) { *p += v; }
#endif

// a lane's integer in a register, or (PARKED) in the lane's scratch memory: every get() a load, every set() a store
template <bool PARKED> struct LaneInt {
  int v;
  __device__ __forceinline__ int get() const { return v; }
  __device__ __forceinline__ void set(int x) { v = x; }
};
template <> struct LaneInt<true> {
  volatile int v;
  __device__ __forceinline__ int get() const { return v; }
  __device__ __forceinline__ void set(int x) { v = x; }
};

// the kernel's argument block where it lies (kernarg segment); `launder` makes the compiler forget what it has already
// loaded through the pointer, so values needed again later are re-read (one scalar load) rather than kept or spilled
#ifndef LGAR_DEVSIM
template <typename T> __device__ __forceinline__ const LGAR_KARG T *launder(const LGAR_KARG T *p) {
  const unsigned long long v = (unsigned long long)p;
  unsigned lo = __builtin_amdgcn_readfirstlane((unsigned)v), hi = __builtin_amdgcn_readfirstlane((unsigned)(v >> 32));
  asm volatile("" : "+s"(lo), "+s"(hi));
  return (const LGAR_KARG T *)(((unsigned long long)hi << 32) | lo);
}
#else
template <typename T> __device__ __forceinline__ const T *launder(const T *p) { return p; }
#endif

template <typename R, int NL>
__device__ __forceinline__ void load_params(const LGAR_KARG KArgs<R> &a, size_t c, ColParams<R, NL> &P) {
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
    P.inv_n[k] = R(1.0) / P.n[k];
    P.cum[k] = (k == 0) ? P.thick[0] : P.cum[(k > 0) ? k - 1 : 0] + P.thick[k];  // GlobalParams.py:99-109
  }
}

template <typename S, int FMAX, int STRIDE = WAVE>
__device__ __forceinline__ FrontsView<S, FMAX, STRIDE> make_view(S *f, unsigned char *fl, int slot) {
  FrontsView<S, FMAX, STRIDE> F;
  F.base = f + slot;
  F.fl = fl + slot;
  return F;
}

// state LDS/registers -> HBM.  Rows [0, nf) are written, rows [nf, nf_before) (fronts that disappeared) are zeroed.
template <typename R, int NL, int FMAX, int MODE>
__device__ __forceinline__ void store_state(const LGAR_KARG KArgs<R> &a, size_t c, const Column<R, NL, FMAX, MODE> &col, int nf_before,
                                            int status_word) {
  const size_t N = (size_t)a.N;
  const int rows = (col.nf > nf_before ? col.nf : nf_before) < a.F ? (col.nf > nf_before ? col.nf : nf_before) : a.F;
  for (int i = 0; i < rows; i++) {
    const bool live = i < col.nf;
    a.depth[i * N + c] = live ? col.F.Z(i) : R(0);
    a.theta[i * N + c] = live ? col.F.TH(i) : R(0);
    a.psi[i * N + c] = live ? col.F.PS(i) : R(0);
    a.k[i * N + c] = live ? ((i < col.nf - 1) ? col.front_k(i, pick(col.P, col.F.layer(i))) : col.k_deepest) : R(0);
    a.dzdt[i * N + c] = live ? col.F.DZ(i) : R(0);
    a.flags[i * N + c] = live ? col.F.flag(i) : (uint8_t)0;
  }
  a.nf[c] = col.nf;
  a.scalars[0 * N + c] = col.ponded_water;
  a.scalars[1 * N + c] = col.previous_precip;
  a.scalars[2 * N + c] = col.ending_volume;
  if constexpr (!Column<R, NL, FMAX, MODE>::GIUH_MEM) {  // (GIUH_MEM: the queue lives there already)
#pragma unroll
    for (int i = 0; i < LGAR_GMAX; i++) a.scalars[(3 + i) * N + c] = col.giuh_q[i];
  }
  a.status[c] = status_word;
}

// dpLGAR.set_internal_states (models/dpLGAR.py:97-147) for one column
template <typename R, int NL, int FMAX>
__device__ __forceinline__ void init_lane(const LGAR_KARG KArgs<R> *ap, size_t c, int lane, WaveLDS<R, FMAX> &lds) {
  const LGAR_KARG KArgs<R> &a = *ap;
  ColParams<R, NL> P;
  load_params<R, NL>(a, c, P);
  Column<R, NL, FMAX, 1> col(P, &ap->G, make_view<R, FMAX>(&lds.f[0][0][0], &lds.fl[0][0], lane));
  col.init_state();
  const int nf_before = a.nf[c];
  store_state<R, NL, FMAX, 1>(a, c, col, nf_before < 0 ? 0 : nf_before, 0);
  const size_t N = (size_t)a.N;
#pragma unroll
  for (int j = 0; j < LGAR_NACC; j++) a.totals[j * N + c] = (j == 9) ? col.ending_volume : R(0);
}

// T x (dpLGAR.forward + MassBalance.change_mass) for one column; the time loop is inside.
// `live` = false for the padding lanes of a ragged tail wave: they integrate a copy of the last column (all 64 lanes stay
// active for the wave reductions) and store nothing.
// Cooperating lanes (a.coop = 4..64, jobs too small to fill the chip): that many adjacent lanes integrate the SAME column c
// redundantly -- same loads, same values, same branches -- and split the nodes of the Geff trapezoid between them (xchg: the
// wave's exchange buffer); `leader` is true for the one lane of the group that stores the column's results.
template <typename R, int NL, int FMAX, int MODE>
__device__ __forceinline__ void forward_lane(const LGAR_KARG KArgs<R> *ap, size_t c, bool live, int lane,
                                             ForwardLDS<R, FMAX, MODE> &lds, bool leader = true, R *xchg = nullptr, int group = 0,
                                             int rank = 0) {
  const LGAR_KARG KArgs<R> &a = *ap;
  const size_t N = (size_t)a.N;
  const bool basin_on = (a.basin != nullptr) && (a.basin_mask != 0u);
  int status = a.status[c];
  // which step this lane starts at: 0 in the first kernel of the chain, the recorded step for a column handed over by
  // the previous kernel, never (T) for everybody else in a later kernel
  // (Lane-varying conditions that live across the whole time loop are kept as two integers, t_begin and t_stop, not as
  // booleans: a long-lived lane mask costs an SGPR pair each, and the SGPR file is what this kernel runs out of first.)
  int t_begin = live ? 0 : a.T;  // padding lanes of a ragged tail wave never start: they only take part in wave reductions
  if (!a.chain_first) {
    t_begin = (live && (status & LGAR_ST_RESUME)) ? (int)((unsigned)status >> LGAR_ST_STEP_SHIFT) : a.T;
    if (any_lane(t_begin < a.T) == 0ull) return;  // nothing handed over to this wave
  }
  status &= LGAR_ST_FAULT_MASK;
  LGAR_MEASURE_POINT(POISON_LDS, lds, coop_mode(MODE) ? group : lane)
  ColParams<R, NL> P;
  load_params<R, NL>(a, c, P);
  // front-table slot: my own, or (MODE 4) my group's -- the lanes of a group hold the same column
  constexpr int STRIDE = Column<R, NL, FMAX, MODE>::STRIDE;
  const int slot = coop_mode(MODE) ? group : lane;  // (also my column of the LDS sums)
  Column<R, NL, FMAX, MODE> col(P, &ap->G, make_view<R, FMAX, STRIDE>(&lds.f[0][0][0], &lds.fl[0][0], slot));
  // state HBM -> LDS / registers
  const int nf_stored = a.nf[c];
  const int cap = FMAX < a.F ? FMAX : a.F;
  int nf = nf_stored < 0 ? 0 : (nf_stored > cap ? cap : nf_stored);
  const int nf_before = nf;
  col.nf = nf;
  for (int i = 0; i < nf; i++) {
    col.F.Z(i) = a.depth[i * N + c];
    col.F.TH(i) = a.theta[i * N + c];
    col.F.PS(i) = a.psi[i * N + c];
    col.F.DZ(i) = a.dzdt[i * N + c];
    col.F.flag(i) = a.flags[i * N + c];
  }
  col.ponded_water = a.scalars[0 * N + c];
  col.previous_precip = a.scalars[1 * N + c];
  col.ending_volume = a.scalars[2 * N + c];
  if constexpr (Column<R, NL, FMAX, MODE>::GIUH_MEM) {
    col.giuh_mem = &a.scalars[3 * N + c];
    col.giuh_stride = N;
    R qsum = R(0);
#pragma unroll
    for (int i = 0; i < LGAR_GMAX; i++) qsum += a.scalars[(3 + i) * N + c];
    col.giuh_live = qsum > R(0);
  } else {
#pragma unroll
    for (int i = 0; i < LGAR_GMAX; i++) col.giuh_q[i] = a.scalars[(3 + i) * N + c];
  }
  col.status = status;
  if (nf < NL) col.status |= LGAR_ST_STRUCT;  // not a state lgar_state_init / lgar_forward produced: column is skipped
  col.k_deepest = (nf > 0) ? a.k[(size_t)(nf - 1) * N + c] : R(0);
  col.new_front_frozen = false;
  col.cap = FMAX < a.F ? FMAX : a.F;
  col.count_geff = a.counters != nullptr;
  col.share_lanes = (xchg != nullptr) ? a.coop : 0;
  col.xchg = xchg;
  col.coop_rank = rank;
  col.drain();
  // accumulators summed over the steps this kernel integrates: the first SR in LDS, the rest in registers
  constexpr int SR = LdsSums<R, FMAX>::rows;
  R tot[8];
#pragma unroll
  for (int j = 0; j < 8; j++) {
    tot[j] = R(0);
    if (j < SR) lds.sums[j][slot] = R(0);
  }
  double wgt = 0.0;
  if (basin_on && leader) wgt = a.weights ? (double)a.weights[c] : 1.0;

  // the column is integrated for steps t_begin <= t < t_stop; t_stop < T: handed to the next kernel of the chain at
  // t_stop (or, t_stop = -1, stopped for good before its first step); t_stop <= t_begin: the state in HBM is left as it is
  // (fp64 kernels of the smallest capacity: t_stop is PARKED in scratch memory -- read at the top of every step, written at a
  // hand-over only.  In a register it lives across all of the column physics, and the allocator of these kernels, 60 registers
  // short, kept it by storing it to scratch at the top of every step: 4 of the 20 bytes per column and step still spilled.)
  constexpr bool PARK = (sizeof(R) == 8) && (FMAX <= LGAR_CAP_SMALL) && !coop_mode(MODE);
  LaneInt<PARK> t_stop_at;
  bool overflow_on_load = false;
  {
    int t_stop = a.T;
    if (t_begin < a.T && nf_stored > cap) {
      // more fronts than this kernel can hold: next kernel of the chain, or (last kernel) front overflow
      if constexpr (PARK) t_stop = a.chain_last ? -1 : t_begin;
      else if (a.chain_last) { overflow_on_load = true; t_stop = -1; } else t_stop = t_begin;
    }
    t_stop_at.set(t_stop);
  }
  if constexpr (PARK) overflow_on_load = t_begin < a.T && nf_stored > cap && a.chain_last;
  if (overflow_on_load) {
    col.status |= LGAR_ST_OVERFLOW;
    // no step of this column will run: its rows of the requested series read as zero (lgar_basin_reduce_kernel sums them).
    // Only the last kernel of a chain rejects a column, and that is a kernel of the full capacity: the hot 8- and 16-front
    // kernels do not carry this loop (in the 8-front fp32 kernel it cost 1.1 ms of 13.5 through register allocation alone)
    if constexpr (FMAX == LGAR_FMAX) {
      if (leader)
        for (int t = t_begin; t < a.T; t++)
#pragma unroll
          for (int j = 0; j < LGAR_NACC; j++)
            if (a.series[j]) a.series[j][(size_t)t * N + c] = R(0);
    }
  }
  // (No software prefetch of the next step's forcing: the two values would have to live in registers across a whole step
  // -- ~10^4 cycles of VALU work -- and at 128 VGPRs they end up as scratch traffic; the load latency of a step's own
  // forcing is covered by the other three waves of the SIMD.)
  // which per-step series the caller asked for: one word instead of ten pointer tests per step (the pointers themselves are
  // re-read from the argument block where they are stored through)
  unsigned series_mask = 0u;
#pragma unroll
  for (int j = 0; j < LGAR_NACC; j++)
    if (a.series[j]) series_mask |= 1u << j;
  const int T = a.T;
  const size_t Nf = (size_t)a.Nf;
  const size_t cf = (Nf == N) ? c : (c / (size_t)a.Fg) % Nf;  // this column's forcing column
  // (MODE 4, one wave per SIMD and registers to spare: the NEXT step's forcing is loaded a step ahead -- a lone wave has nothing
  // else to cover the ~1 us of that load with)
  constexpr bool AHEAD = coop_mode(MODE);
  R precip_ahead = R(0), pet_ahead = R(0);
  if constexpr (AHEAD) {
    if (T > 0) {
      precip_ahead = a.precip[cf];
      pet_ahead = a.pet[cf];
    }
    settle_load(precip_ahead);  // (in their registers before the loop: see the loop's own note)
    settle_load(pet_ahead);
  }
  LGAR_MEASURE_POINT(BALLAST_INIT, a, c)
  for (int t = 0; t < T; t++) {
    ap = launder(ap);
    const LGAR_KARG KArgs<R> &a = *ap;  // (shadows the outer reference on purpose)
    col.G = &ap->G;
    const size_t o = (size_t)t * N + c;
    R precip, pet;
    if constexpr (AHEAD) {
      precip = precip_ahead;
      pet = pet_ahead;
      const size_t tn = (size_t)((t + 1 < T) ? t + 1 : t);
      precip_ahead = a.precip[tn * Nf + cf];
      pet_ahead = a.pet[tn * Nf + cf];
    } else {
      precip = a.precip[(size_t)t * Nf + cf];
      pet = a.pet[(size_t)t * Nf + cf];
    }
    LGAR_MEASURE_POINT(CLK, 9)
    bool active = (t >= t_begin) && (t < t_stop_at.get());
    if (active && !a.chain_last && col.nf + a.G.nsub > FMAX) {
      // this step could outgrow the kernel's front capacity: hand the column over, state as of the end of step t-1
      active = false;
      t_stop_at.set(t);
    }
    if (any_lane(active) == 0ull) continue;
    if (active) col.forward(precip, pet);
    LGAR_MEASURE_POINT(BALLAST_TOUCH)
    if constexpr (AHEAD) {
      // the forcing loaded a step ahead is taken into its registers HERE, before this step's stores are issued: loads and
      // stores share one counter, and a wait for the load placed after the stores would wait for the stores as well
      settle_load(precip_ahead);
      settle_load(pet_ahead);
    }
    const R acc[LGAR_NACC] = {col.a_precip, col.a_pet, col.a_aet, col.a_infil, col.a_runoff,
                              col.a_perc, col.a_giuh, col.a_disch, col.ponded_water, col.ending_volume};
    if (active && leader) {
#pragma unroll
      for (int j = 0; j < LGAR_NACC; j++)
        if (series_mask & (1u << j)) a.series[j][o] = acc[j];
    }
    LGAR_MEASURE_POINT(CLK, 18)
    if (basin_on) {
      // basin aggregation in the epilogue of the step (physics/MassBalance.py:77-108 over many columns)
      const double w = active ? wgt : 0.0;
#pragma unroll
      for (int j = 0; j < LGAR_NACC; j++)
        if (a.basin_mask & (1u << j)) {
          const double s = wave_sum(w * (double)acc[j]);
          if (lane == 0) {
            if constexpr (PARK) atomic_add_uniform(&a.basin[(size_t)j * T + t], s);
            else atomic_add(&a.basin[(size_t)j * T + t], s);
          }
        }
    }
    if (active) {
#pragma unroll
      for (int j = 0; j < 7; j++) {  // MassBalance.change_mass, MassBalance.py:31-44
        if (j == 5 && a.G.bottom_mode == 0) continue;  // percolation is identically zero in the reference's mode
        const int row = sum_row<SR>(j);
        if (row >= 0) lds.sums[row][slot] = lds.sums[row][slot] + acc[j];
        else tot[j] = tot[j] + acc[j];
      }  // (discharge [7] is the same sum as giuh_runoff [6]: both gain the same routed runoff, models/dpLGAR.py:293-297)
      col.drain();
    }
    LGAR_MEASURE_POINT(CLK, 21)
  }
  LGAR_MEASURE_POINT(BALLAST_FOLD, col)
  ap = launder(ap);
  const LGAR_KARG KArgs<R> &z = *ap;  // the epilogue re-reads what it needs
  if (z.counters != nullptr) {
    // measurement: wave-level Geff evaluations (the dominant instruction stream) of this block, counted above the fault
    // bits of the lanes' status words
    const double calls = wave_sum((double)((unsigned)col.status >> LGAR_ST_STEP_SHIFT));
    if (lane == 0 && calls > 0.0) {
      if constexpr (PARK) atomic_add_uniform(&z.counters[0], (unsigned long long)calls);
      else atomic_add(&z.counters[0], (unsigned long long)calls);
    }
  }
  col.status &= LGAR_ST_FAULT_MASK;
  const int t_stop = t_stop_at.get();
  if (t_begin >= z.T) return;  // not this kernel's column (or a padding lane)
  if (!leader) return;  // the column's results are stored by the leader of its group of cooperating lanes
  if (z.pending_out != nullptr) {
    // columns handed to the next kernel of the chain are counted, so that a next kernel with nothing to do (the usual
    // case) leaves after one load instead of scanning every status word
    const unsigned long long m = any_lane(t_stop >= 0 && t_stop < z.T);
    if (m != 0ull && first_active_lane()) {
      if constexpr (PARK) atomic_add_uniform(z.pending_out, (unsigned)__builtin_popcountll(m));
      else atomic_add(z.pending_out, (unsigned)__builtin_popcountll(m));
    }
  }
  int word = col.status;
  if (t_stop >= 0 && t_stop < z.T) word |= LGAR_ST_RESUME | (int)((unsigned)t_stop << LGAR_ST_STEP_SHIFT);
  if (t_stop <= t_begin) {
    z.status[c] = word;
    if (z.call_sums != nullptr && z.chain_first) {
#pragma unroll
      for (int j = 0; j < 8; j++) z.call_sums[j * N + c] = R(0);
      z.call_sums[8 * N + c] = col.ponded_water;
      z.call_sums[9 * N + c] = col.ending_volume;
    }
    return;
  }
  store_state<R, NL, FMAX, MODE>(z, c, col, nf_before, word);
#pragma unroll
  for (int j = 0; j < 7; j++)
    if (sum_row<SR>(j) >= 0) tot[j] = lds.sums[sum_row<SR>(j)][slot];
  tot[7] = tot[6];
#pragma unroll
  for (int j = 0; j < 8; j++) z.totals[j * N + c] = z.totals[j * N + c] + tot[j];  // MassBalance's run totals
  z.totals[8 * N + c] = col.ponded_water;
  z.totals[9 * N + c] = col.ending_volume;
  if (z.call_sums != nullptr) {
    // a column handed over by an earlier kernel of the chain already has its first part in place
    const bool add = !z.chain_first;
#pragma unroll
    for (int j = 0; j < 8; j++) z.call_sums[j * N + c] = add ? z.call_sums[j * N + c] + tot[j] : tot[j];
    z.call_sums[8 * N + c] = col.ponded_water;
    z.call_sums[9 * N + c] = col.ending_volume;
  }
}

}  // namespace lgar
