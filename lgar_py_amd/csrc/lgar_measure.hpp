// lgar_measure.hpp -- MEASUREMENT BUILDS ONLY (lgar_py_amd.build.build_variant passes -DLGAR_MEASURE; tools/ablate.py).
//
// Definitions of the measurement points the device code marks (lgar_device.hpp: LGAR_MEASURE_POINT, LGAR_ABLATABLE,
// LGAR_COUNT_GEFF_CALL).  The product library never includes this file: there the points are empty.
//   -DLGAR_DUP_<X>     routine X runs twice on opaque copies of its inputs, results unchanged: the time difference to the
//                      plain build is X's cost with the column dynamics (and so all other work) untouched
//   -DLGAR_ABL_NO<X>   statement X is left out (what the register allocator / the schedule does without it)
//   -DLGAR_COUNT_*     what the Geff-call counter counts (one call site, only sparse evaluations, lanes instead of waves)
// The macros expand inside member functions of lgar::Column and use its members and the locals passed to them.
#pragma once

// (included from inside namespace lgar)
__device__ __forceinline__ float opaque(float x) { asm volatile("" : "+v"(x)); return x; }
__device__ __forceinline__ double opaque(double x) { asm volatile("" : "+v"(x)); return x; }

#define LGAR_MEASURE_POINT(NAME, ...) LGAR_POINT_##NAME(__VA_ARGS__)
#define LGAR_ABLATABLE(NAME, ...) LGAR_STMT_##NAME(__VA_ARGS__)

// ---- statements a build can leave out
#ifdef LGAR_ABL_NOINSERT
#define LGAR_STMT_INSERT(...)
#else
#define LGAR_STMT_INSERT(...) __VA_ARGS__
#endif
#ifdef LGAR_ABL_NOMOVE
#define LGAR_STMT_MOVE(...)
#else
#define LGAR_STMT_MOVE(...) __VA_ARGS__
#endif
#ifdef LGAR_ABL_NODZDT
#define LGAR_STMT_DZDT(...)
#else
#define LGAR_STMT_DZDT(...) __VA_ARGS__
#endif

// ---- capillary_drive
#ifdef LGAR_ABL_NOGEFF  // the kernel without the trapezoid
#define LGAR_POINT_NOGEFF(lk, theta1, theta2) return theta1 * theta2 + lk.alpha;
#else
#define LGAR_POINT_NOGEFF(lk, theta1, theta2)
#endif
#ifdef LGAR_ABL_NOGEFF
#define LGAR_POINT_NOGEFF_FRONTS(lk, theta1, theta2, kr) { kr = theta1; return theta1 * theta2 + lk.alpha; }
#else
#define LGAR_POINT_NOGEFF_FRONTS(lk, theta1, theta2, kr)
#endif
#ifdef LGAR_COUNT_SITE  // count one call site only (1 dzdt, 2 dry depth, 3 insert)
#define LGAR_COUNT_IF_SITE(site) if (site == LGAR_COUNT_SITE)
#else
#define LGAR_COUNT_IF_SITE(site)
#endif
#ifdef LGAR_COUNT_MAXLANES  // ... and only the evaluations with at most this many lanes taking part
#define LGAR_COUNT_IF_SPARSE if (__builtin_popcountll(any_lane(true)) <= LGAR_COUNT_MAXLANES)
#else
#define LGAR_COUNT_IF_SPARSE
#endif
#ifdef LGAR_COUNT_LANES  // every evaluating lane counts
#define LGAR_COUNT_WHO count_geff
#else
#define LGAR_COUNT_WHO (count_geff && first_active_lane())
#endif
#define LGAR_COUNT_GEFF_CALL(site) \
  LGAR_COUNT_IF_SITE(site) LGAR_COUNT_IF_SPARSE { if (LGAR_COUNT_WHO) status += (1 << LGAR_ST_STEP_SHIFT); }
#ifdef LGAR_DUP_GEFF
#define LGAR_POINT_DUP_GEFF(lk, theta1, theta2)                                                    \
  if constexpr (sizeof(S) == sizeof(R)) {                                                          \
    const S extra = geff(lk, opaque(theta1), opaque(theta2), G->nint);                             \
    if (val(extra) == R(12345.678)) return extra; /* practically never true: keeps the duplicate alive */ \
  }
#else
#define LGAR_POINT_DUP_GEFF(lk, theta1, theta2)
#endif

// ---- theta_mass_balance
#ifdef LGAR_ABL_NOSEARCH
#define LGAR_POINT_NOSEARCH(lk, psi, new_mass) return theta_from_h<S, POL>(lk, psi + new_mass);
#else
#define LGAR_POINT_NOSEARCH(lk, psi, new_mass)
#endif
#ifdef LGAR_DUP_SEARCH
#define LGAR_POINT_DUP_SEARCH(K, lk, psi, new_mass, prior_mass, dth, dthick, dth_k, dthick_k)      \
  if constexpr (sizeof(S) == sizeof(R)) {                                                          \
    const S extra = theta_mass_balance<K>(lk, opaque(psi), opaque(new_mass), opaque(prior_mass), dth, dthick, dth_k, dthick_k); \
    if (val(extra) == R(-1.0)) status |= LGAR_ST_STRUCT; /* never true */                          \
  }
#else
#define LGAR_POINT_DUP_SEARCH(...)
#endif

// ---- move_wetting_front / forward
#ifdef LGAR_DUP_EVENT
#define LGAR_POINT_DUP_EVENT()                                                                     \
  if (front_event_pending() && val(F.Z(0)) == R(-12345.0)) status |= LGAR_ST_STRUCT; /* never true */ \
  asm volatile("" ::: "memory");
#else
#define LGAR_POINT_DUP_EVENT()
#endif
#ifdef LGAR_DUP_PSI
#define LGAR_POINT_DUP_PSI() asm volatile("" ::: "memory"); update_psi();
#else
#define LGAR_POINT_DUP_PSI()
#endif
#ifdef LGAR_DUP_FDD
#define LGAR_POINT_DUP_FDD()                                                                       \
  if (free_drainage_front() == 12345) status |= LGAR_ST_STRUCT; /* never true */                   \
  asm volatile("" ::: "memory");
#else
#define LGAR_POINT_DUP_FDD()
#endif
#ifdef LGAR_DUP_DZDT  // idempotent: a second pass recomputes the same dz/dt (Geff included)
#define LGAR_POINT_DUP_DZDT(h_p) asm volatile("" ::: "memory"); calc_dzdt(h_p);
#else
#define LGAR_POINT_DUP_DZDT(h_p)
#endif
#ifdef LGAR_DUP_MB
#define LGAR_POINT_DUP_MB(volume)                                                                  \
  volume = mass_balance();                                                                         \
  asm volatile("" ::: "memory");                                                                   \
  if (val(volume) == R(-1.0)) status |= LGAR_ST_STRUCT; /* never true */
#else
#define LGAR_POINT_DUP_MB(volume)
#endif

// ---- geff_mixed: the general node form throughout (what round 3 ran)
#ifdef LGAR_GEFFM_GENERAL_ONLY
#define LGAR_POINT_GEFFM_GENERAL_ONLY(ser_pairs, dir_pair, pairs) ser_pairs = 0; dir_pair = pairs + 1;
#else
#define LGAR_POINT_GEFFM_GENERAL_ONLY(ser_pairs, dir_pair, pairs)
#endif
// ---- geff_mixed: how many four-node groups each of its loops ran (series-only / general / difference-only / checked), per
// wave-level evaluation; read back with lgar_debug_counters (lgar_kernels_nl.hip, measurement builds only)
static __device__ unsigned long long lgar_dbg_counters[8];
#ifndef LGAR_COUNT_GEFFM_REGIONS
#define LGAR_POINT_GEFFM_REGIONS(a, b, c, d)
#else
#define LGAR_POINT_GEFFM_REGIONS(a, b, c, d)                                                        \
  if (first_active_lane()) {                                                                       \
    atomicAdd(&lgar_dbg_counters[0], (unsigned long long)(a));                                     \
    atomicAdd(&lgar_dbg_counters[1], (unsigned long long)(b));                                     \
    atomicAdd(&lgar_dbg_counters[2], (unsigned long long)(c));                                     \
    atomicAdd(&lgar_dbg_counters[3], (unsigned long long)(d));                                     \
    atomicAdd(&lgar_dbg_counters[4], 1ull);                                                        \
    atomicAdd(&lgar_dbg_counters[5], (unsigned long long)__builtin_popcountll(any_lane(true)));    \
  }
#endif

// ---- cycle attribution (-DLGAR_CLOCKS): LGAR_MEASURE_POINT(CLK, id) adds the shader-clock cycles since the previous point
// of the same wave to slot id -- the first wave of the grid only, so a one-wave job (a single column with cooperating lanes)
// reads as a profile of its step; read back with lgar_debug_clocks (lgar_kernels_nl.hip, measurement builds only)
#ifdef LGAR_CLOCKS
static __device__ unsigned long long lgar_dbg_clk[64];
static __shared__ unsigned long long lgar_dbg_clk_last;
#define LGAR_POINT_CLK(id)                                                                         \
  if (blockIdx.x == 0) {                                                                           \
    const unsigned long long now_ = __builtin_readcyclecounter();                                  \
    if (first_active_lane()) {                                                                     \
      if (now_ - lgar_dbg_clk_last < (1ull << 36)) { /* (not the wave's first point: no previous one) */ \
        atomicAdd(&lgar_dbg_clk[id], now_ - lgar_dbg_clk_last);                                    \
        atomicAdd(&lgar_dbg_clk[32 + id], 1ull);                                                   \
      }                                                                                            \
      lgar_dbg_clk_last = now_;                                                                    \
    }                                                                                              \
  }
#else
#define LGAR_POINT_CLK(id)
#endif

// ---- register ballast (-DLGAR_BALLAST=K): K values per lane that live through the whole time loop of a forward kernel and must
// be in registers once per step -- K more registers for the allocator to spill around the trapezoid.  How
// tools/spill_determinism.sh studies what spill code does to the results of the 128-register fp32 kernel.
#ifdef LGAR_BALLAST
#define LGAR_POINT_BALLAST_INIT(a, c)                                                               \
  R ballast_[LGAR_BALLAST];                                                                        \
  _Pragma("unroll") for (int q_ = 0; q_ < LGAR_BALLAST; q_++) ballast_[q_] = a.theta_e[c] * R(q_ + 1);
#define LGAR_POINT_BALLAST_TOUCH()                                                                  \
  _Pragma("unroll") for (int q_ = 0; q_ < LGAR_BALLAST; q_++) ballast_[q_] = opaque(ballast_[q_]);
#define LGAR_POINT_BALLAST_FOLD(col)                                                                \
  {                                                                                                \
    R bs_ = R(0);                                                                                  \
    _Pragma("unroll") for (int q_ = 0; q_ < LGAR_BALLAST; q_++) bs_ += ballast_[q_];                \
    if (bs_ == R(-12345.0)) col.status |= LGAR_ST_STRUCT; /* never true */                         \
  }
#else
#define LGAR_POINT_BALLAST_INIT(a, c)
#define LGAR_POINT_BALLAST_TOUCH()
#define LGAR_POINT_BALLAST_FOLD(col)
#endif

// ---- LDS poison (-DLGAR_POISON_LDS): before a wave loads a block's state, every row of its front table and of its sums is filled
// with NaN -- what the block before left there is gone.  A persistent wave integrates whatever blocks the ticket counter hands it,
// so a kernel that read a row it had not written (a front index at or beyond n_fronts) would give results that depend on the
// order of the blocks, i.e. differ from run to run; with the poison such a read turns the column's results into NaN.  The
// product's results must not change (tools/determinism_probe.py prints a digest to compare).
#ifdef LGAR_POISON_LDS
#define LGAR_POINT_POISON_LDS(lds, slot)                                                            \
  {                                                                                                \
    const R nan_ = R(__builtin_nan(""));                                                           \
    for (int f_ = 0; f_ < 4; f_++)                                                                 \
      for (int i_ = 0; i_ < FMAX; i_++) lds.f[f_][i_][slot] = nan_;                                \
    for (int i_ = 0; i_ < FMAX; i_++) lds.fl[i_][slot] = (unsigned char)0x7f;                      \
    for (int i_ = 0; i_ < (int)(sizeof(lds.sums) / sizeof(lds.sums[0])); i_++) lds.sums[i_][slot] = nan_; \
  }
#else
#define LGAR_POINT_POISON_LDS(lds, slot)
#endif

// ---- the fp32 kernel's calc_dzdt with the trapezoid's heads taken from the fronts' own psi and K(theta_i) from the wet end node
// (-DLGAR_F32_HEADS_FROM_PSI): what the mixed-precision kernel does, tried on the fp32 kernel in round 4, where it made the
// results differ from run to run.  Kept as a measurement variant to study that (tools/spill_determinism.sh).
#ifdef LGAR_F32_HEADS_FROM_PSI
#define LGAR_POINT_F32_HEADS_FROM_PSI(lk, i, g, ki, fronts_done)                                    \
  if constexpr (sizeof(S) == 4 && sizeof(R) == 4 && MODE != 0) {                                   \
    if (!G->closed_form) {                                                                         \
      float kn_ = 0.0f;                                                                            \
      LGAR_COUNT_GEFF_CALL(1)                                                                      \
      g = geff_f32_from_heads(lk, F.PS(i + 1), F.PS(i), G->nint, &kn_);                            \
      ki = lk.ksat * kn_;                                                                          \
      if (i == 0 && new_front_frozen) ki = ki * G->frozen;                                         \
      fronts_done = true;                                                                          \
    }                                                                                              \
  }
#else
#define LGAR_POINT_F32_HEADS_FROM_PSI(lk, i, g, ki, fronts_done)
#endif
