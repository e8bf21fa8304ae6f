/*
 * lgar.h -- C-ABI of the MI355X-native many-column LGAR engine (liblgar_hip.so).
 *
 * The reference (LGAR-py, "dpLGAR") has no FFI: its only seam is the Python object surface of
 * dpLGAR(nn.Module) (dpLGAR/models/dpLGAR.py:30-299) as driven by the agent
 * (dpLGAR/agents/DifferentiableLGAR.py:94-172) and drained by MassBalance
 * (dpLGAR/models/physics/MassBalance.py:31-53).  The entry points below are what a binding for that
 * path needs; each cites the reference interface it replaces.  INTEGRATION.md shows the ctypes stub.
 *
 * Conventions
 *  - Plain pointers + sizes; no torch types.  Every pointer is DEVICE memory owned by the caller
 *    (the library never allocates, frees or copies host<->device), valid on the given stream.
 *  - All arrays are struct-of-arrays, COLUMN-FASTEST: a per-layer quantity is [n_layers][n_columns],
 *    a per-front quantity [front_slots][n_columns], a forcing / per-step series [n_steps][n_columns].
 *  - dtype: LGAR_F32 or LGAR_F64 selects the element type of every `void*` array (float / double).
 *  - Return value: 0 ok; <0 argument / launch error (LGAR_E_*).  Physics faults never abort the
 *    launch: they set bits in status[column] (the reference raises Python exceptions instead:
 *    physics/utils.py:17-27,181-183; layers/Layer.py:1206-1208,1115,980).
 *  - Re-entrant, no globals; one in-flight call per state buffer.  `stream` is a hipStream_t (NULL =
 *    default stream).
 */
#ifndef LGAR_H
#define LGAR_H

#include <stdint.h>

#ifdef __cplusplus
extern "C" {
#endif

/* Front capacity.  The reference's per-layer front lists are unbounded (layers/Layer.py:1336-1416; observed <= 8 on the
 * bundled cases).  The kernels keep a wave's front table in LDS and are compiled for three capacities; lgar_forward
 * starts with the smallest that fits and hands the (rare) columns that outgrow it to the next one in the same call, so a
 * column can hold up to LGAR_FMAX fronts (LGAR_ST_OVERFLOW beyond) while the common case runs at the occupancy of 8. */
#define LGAR_CAP_SMALL 8
#define LGAR_CAP_MID 16
#define LGAR_FMAX 32
#define LGAR_LMIN 2   /* the reference itself needs >= 2 layers (Layer.py:204) */
#define LGAR_LMAX 6   /* soil layers: kernels are compiled for 2 .. 6 (BASELINE configs: 3; the reference builds any number, Layer.py:77-89) */
#define LGAR_GMAX 8   /* GIUH ordinates */
#define LGAR_NSCAL (3 + LGAR_GMAX) /* scalars row count: ponded_water, previous_precip, ending_volume, giuh_queue[GMAX] */
#define LGAR_NACC 10  /* precip, PET, AET, infiltration, runoff, percolation, giuh_runoff, discharge, ponded_water, ending_volume */

#define LGAR_F32 0
#define LGAR_F64 1

#define LGAR_E_ARG (-1)     /* bad argument (null pointer, size out of range, unsupported n_layers) */
#define LGAR_E_LAUNCH (-2)  /* HIP launch error */
#define LGAR_E_NODEVICE (-3)

/* status bits, one int32 per column */
#define LGAR_ST_NAN 1
#define LGAR_ST_NEGBASE 2
#define LGAR_ST_THETA_ORDER 4
#define LGAR_ST_OVERFLOW 8
#define LGAR_ST_ITERCAP 16
#define LGAR_ST_BOTTOM 32
#define LGAR_ST_STRUCT 64
#define LGAR_ST_FAULT_MASK 0x7f
/* internal to a lgar_forward call (never set when it returns): the column was handed to the next kernel of the
 * front-capacity chain at the step index held in bits 8..31 */
#define LGAR_ST_RESUME 128
#define LGAR_ST_STEP_SHIFT 8
#define LGAR_NTICKETS 8   /* LgarState.tickets: [0..2] one work counter per kernel of a call's capacity chain; [4..5] columns
                             handed over by the first / second kernel (a next kernel with nothing to do exits at once) */
#define LGAR_NCOUNTERS 4  /* LgarStepOut.counters: [0] wave-level Geff evaluations of the launch; [1..3] reserved */

/* front flag byte: low 7 bits layer number, bit 7 = to_bottom (layers/WettingFront.py:39,49) */
#define LGAR_FLAG_BOTTOM 0x80

/* Run-time constants: the cfg keys dpLGAR(cfg) reads (models/dpLGAR.py:31-95, physics/GlobalParams.py:79-138). */
typedef struct {
  int32_t n_columns;      /* N */
  int32_t n_layers;       /* len(cfg.data.layer_thickness); LGAR_LMIN..LGAR_LMAX */
  int32_t n_steps;        /* T forcing rows processed by this call */
  int32_t num_subcycles;  /* cfg.models.num_subcycles */
  int32_t nint;           /* cfg.constants.nint (120) */
  int32_t n_giuh;         /* len(cfg.data.giuh_ordinates) <= LGAR_GMAX */
  int32_t search_mode;    /* 0 = the reference's literal line searches (Layer.py:275-317, 681-701) */
  int32_t bottom_mode;    /* 0 = reference behaviour: a front reaching the domain bottom sets LGAR_ST_BOTTOM and the
                             column stops (the reference crashes, Layer.py:980); 1 = LGAR-C intent: the bottom layer is
                             exempt from the layer-boundary step and wetting_front_cross_domain_boundary
                             (Layer.py:1010-1053) turns the overshoot into percolation.  Parity of mode 1 is unpinned. */
  int32_t use_closed_form_G; /* cfg.data.use_closed_form_G: Brooks-Corey closed-form capillary drive
                                (lgar/green_ampt.py:85-98) instead of the nint-interval trapezoid (:45-84) */
  int32_t front_slots;       /* rows of the per-front state arrays, 3 .. LGAR_FMAX (0 = LGAR_FMAX): the most fronts a
                                column can hold in this state buffer */
  double dt_h;               /* cfg.models.subcycle_length_h */
  double initial_psi;        /* cfg.data.initial_psi */
  double ponded_depth_max;   /* cfg.data.ponded_depth_max */
  double wilting_point_psi;  /* cfg.data.wilting_point_psi */
  double frozen_factor;      /* cfg.constants.frozen_factor */
  double giuh[LGAR_GMAX];    /* cfg.data.giuh_ordinates */
  int64_t iter_cap;          /* cap on the reference's unbounded line searches (0 = default) */
  int32_t forcing_columns;   /* columns of the forcing (and tangent-weight) arrays: 0 or n_columns = one forcing column per
                                soil column; a divisor Nf of n_columns = broadcast, soil column c reads forcing column c % Nf
                                (Nf = 1: one basin series for every column, the reference's Data yields exactly that,
                                data/Data.py:32-37; the differentiable path lays its parameter directions side by side this
                                way without replicating the forcing) */
  int32_t forcing_group;     /* 0 or 1: as above.  G > 1: G CONSECUTIVE soil columns share a forcing column -- column c reads
                                forcing column (c / G) % forcing_columns; n_columns must be a multiple of G * forcing_columns.
                                (The differentiable path puts the G parameter directions of one column in adjacent lanes:
                                they take the same branches, so a wavefront diverges over 64 / G columns instead of 64.) */
  int32_t tangent_share;     /* lgar_forward_tangent only.  0: every column stands alone.  W = 2..32: the caller guarantees that
                                each group of W consecutive columns is ONE soil column (identical parameters and forcing) with W
                                different directions; the W lanes then share the Geff trapezoid -- each evaluates every W-th node,
                                and the tangent of the whole sum comes from five direction-independent sums (fp64 fast modes;
                                ignored elsewhere).  n_columns must be a multiple of W; a wavefront carries floor(64 / W) such
                                groups (W = 9, the 3 x L parameters of a 3-layer column: 7 groups, one idle lane). */
  int32_t geff_mode;         /* 0: the Geff trapezoid (lgar/green_ampt.py:45-84) in the precision of `dtype`.  1 (LGAR_F64 fast
                                modes only; ignored elsewhere): mixed precision -- column state, branches and mass bookkeeping
                                stay fp64, the two heads and the two end nodes of the trapezoid stay fp64, its 119 interior
                                nodes use the fp32 hardware transcendentals and are summed in fp64 */
  int32_t forward_lanes;     /* lgar_forward, LGAR_F64 fast modes with the trapezoid, native (geff_mode 0) or mixed precision
                                (geff_mode 1) alike (ignored elsewhere).  0: the library decides -- jobs under one wave per SIMD
                                get 4..64 cooperating lanes per column, which split the nodes of the Geff trapezoid and the
                                front sweep's independent evaluations between them (results bit for bit those of one lane per
                                column in the same geff_mode); 1: one lane per column whatever the job size; 4 .. 64: that
                                many (64 / lanes columns per wavefront) */
  int32_t reserved4;
} LgarDims;

/* Per-column soil parameters, each [n_layers][n_columns].  Replaces dpLGAR.alpha/.n/.ksat
 * (models/dpLGAR.py:50-57), theta_e/theta_r from the soil table (data/utils.py:66-67) and
 * cfg.data.layer_thickness.  ksat is the value BEFORE frozen_factor is applied. */
typedef struct {
  const void *alpha, *n, *ksat, *theta_e, *theta_r, *thickness;
} LgarParams;

/* Per-column model state (caller-owned, updated in place).  Replaces the Layer/WettingFront object
 * graph (layers/Layer.py:62-90, layers/WettingFront.py:38-49) and the model attributes
 * ponded_water / previous_precip / ending_volume / giuh_runoff_queue (models/dpLGAR.py:128-147). */
typedef struct {
  void *depth, *theta, *psi, *k, *dzdt; /* each [front_slots][n_columns], fronts ordered top -> bottom; rows at and
                                           beyond n_fronts[column] are not meaningful */
  uint8_t *flags;                        /* [front_slots][n_columns] */
  int32_t *n_fronts;                     /* [n_columns] */
  void *scalars;                         /* [LGAR_NSCAL][n_columns] */
  void *totals;                          /* [LGAR_NACC][n_columns]: run totals, what MassBalance accumulates
                                            (physics/MassBalance.py:31-44); rows 8,9 hold the latest value */
  uint32_t *tickets;                     /* NULL or device uint32[LGAR_NTICKETS]: work counters of the persistent-wave
                                            schedule (the library zeroes them on the stream at every call).  With tickets the
                                            forward kernels run as ONE resident wave per wave slot of the chip, each pulling
                                            64-column blocks until none is left (no partially filled last round); without,
                                            one workgroup per block. */
} LgarState;

/* Forcing, each [n_steps][forcing_columns], cm/h (data/Data.py:32-37). */
typedef struct {
  const void *precip, *pet;
} LgarForcing;

/* Optional per-step outputs.
 * series[j]: [n_steps][n_columns] or NULL: the model accumulators as they stand after forward() and before
 *   MassBalance.change_mass zeroes them (order = LGAR_NACC list).  series[4] (runoff) and series[5] (percolation)
 *   are forward()'s return pair (models/dpLGAR.py:299).
 * basin: [LGAR_NACC][n_steps] fp64 or NULL; for every bit j set in basin_mask,
 *   basin[j][t] += sum over this launch's columns of weight[c] * accumulator_j[t][c]
 *   (the caller zeroes it; what MassBalance.report_mass / the agent's y_hat aggregate over a basin).  Where series[j]
 *   is given as well, the sum is taken from the stored series by a second kernel on the same stream (one pass, fixed
 *   summation order: the same bits on every run; ~1 % of the launch).  Without series[j] it is reduced inside the forward
 *   kernels (wave reduction + one fp64 atomic per wave and step: no [n_steps][n_columns] array is needed, summation order
 *   across waves is not fixed, ~10 % of an fp32 launch on 1M columns).  Calls that add into the same basin block must
 *   be on one stream.
 * weights: [n_columns] (dtype) or NULL = 1: e.g. area fractions.
 * counters[0] += number of wave-level Geff evaluations (calc_geff, lgar/green_ampt.py:19-99) of the launch: what
 *   bench.py prices against the chip's measured transcendental issue rate. */
typedef struct {
  void *series[LGAR_NACC];
  double *basin;
  const void *weights;
  uint32_t basin_mask;
  uint32_t reserved;
  uint64_t *counters; /* NULL or device uint64[LGAR_NCOUNTERS], accumulated (caller zeroes): measurement only */
  void *call_sums;    /* NULL or [LGAR_NACC][n_columns] (dtype): rows 0..7 = the accumulators summed over THIS call's
                         steps (what the model attributes gain over a block of forward() calls, models/dpLGAR.py:271-298),
                         rows 8, 9 = latest ponded_water / ending_volume */
} LgarStepOut;

const char *lgar_version(void);
/* ABI revision of this header: bumped whenever a struct of this file changes size or layout or an entry point changes its
 * argument list (3: LgarDims.geff_mode / forward_lanes, lgar_forward_tangent's `tickets`).  A caller compares it with the
 * LGAR_ABI_VERSION it was compiled against, and lgar_sizeof_dims() with its own sizeof(LgarDims), before the first call. */
#define LGAR_ABI_VERSION 3
int32_t lgar_abi_version(void);
int32_t lgar_sizeof_dims(void);
int32_t lgar_fmax(void);
int32_t lgar_lmax(void);
/* Lanes per column lgar_forward would use for these dims and dtype (LgarDims.forward_lanes = 0: the library's choice for jobs
 * under one wave per SIMD; 1 = every column its own lane).  A pure function of its arguments and of the device's CU count
 * (256 when no device is visible). */
int32_t lgar_cooperating_lanes(const LgarDims *dims, int32_t dtype);

/* dpLGAR.set_internal_states() (models/dpLGAR.py:97-147): one to_bottom front per layer at
 * psi = initial_psi, theta = theta(initial_psi); zero scalars/totals; status = 0. */
int32_t lgar_state_init(const LgarDims *dims, const LgarParams *params, LgarState *state, int32_t *status,
                        int32_t dtype, void *stream);

/* n_steps x dpLGAR.forward(x) (models/dpLGAR.py:154-299) for every column, each followed by the
 * MassBalance.change_mass drain (physics/MassBalance.py:31-53), i.e. the agent's inner loop
 * (agents/DifferentiableLGAR.py:117-125).  `out` may be NULL. */
int32_t lgar_forward(const LgarDims *dims, const LgarParams *params, LgarState *state, const LgarForcing *forcing,
                     const LgarStepOut *out, int32_t *status, int32_t dtype, void *stream);

/* Forward-mode tangent of lgar_forward (the differentiable path; replaces torch autograd through
 * forward(), agents/DifferentiableLGAR.py:119,163).  For a parameter direction (d_alpha, d_n, d_ksat,
 * each [n_layers][n_columns], NULL = 0) it integrates value and tangent together from a FRESH state
 * (set_internal_states) over n_steps and accumulates, per column,
 *     grad_out[column] = sum_t  w_runoff[t][column] * d runoff_t + w_perc[t][column] * d percolation_t
 * (w_* may be NULL).  Line-search offsets are constants w.r.t. the parameters, as in the reference
 * (Layer.py:277-288,683-696).  tangent_runoff ([n_steps][n_columns]) may be NULL.
 * tickets: NULL or device uint32[LGAR_NTICKETS] (zeroed by the library on the stream): with it the kernels run as one
 * resident wave per wave slot of the chip, each pulling blocks of columns until none is left. */
int32_t lgar_forward_tangent(const LgarDims *dims, const LgarParams *params, const LgarParams *direction,
                             const LgarForcing *forcing, const void *w_runoff, const void *w_perc,
                             void *grad_out, void *tangent_runoff, int32_t *status, int32_t dtype, void *stream,
                             uint32_t *tickets);

/* Leaf kernels (known-answer tests on the GPU), element-wise over n items:
 * op 0 theta_from_h(x), 1 se_from_h(x), 2 k_from_se(x), 3 h_from_se(x)      (physics/utils.py:35-174)
 * op 4 geff(theta1 = x, theta2 = y)                                        (lgar/green_ampt.py:45-84)
 * op 6 the same trapezoid evaluated operation by operation like the reference (4 pow + sqrt per node, running h)
 * op 5 aet(psi = x, pet = y, dt_h = z)                                      (lgar/aet.py:17-51)
 * op 7 log2(x), 8 exp2(x) as the Geff trapezoid evaluates them, 9 pow(x, y) as the fast modes evaluate torch.pow
 *      (accuracy of the device arithmetic on the real hardware; soil parameters unused but required)
 * op 10 geff(theta1 = x, theta2 = y) by the mixed-precision trapezoid of LgarDims.geff_mode = 1 (LGAR_F64; LGAR_F32: op 4)
 * op 11 x / y as the fast modes divide (LGAR_F64: reciprocal + Newton + correction, within an ulp of the IEEE quotient)
 * op 12 pow(x, y), 13 log2(x), 14 exp2(x) as the mixed-precision kernels evaluate them (polynomial terms combined pairwise)
 * alpha, n, ksat, theta_e, theta_r: [n] per-item soil parameters. */
int32_t lgar_leaf_batch(int32_t op, int32_t n_items, const void *x, const void *y, double z, const void *alpha,
                        const void *n, const void *ksat, const void *theta_e, const void *theta_r, int32_t nint,
                        double wilting_point_psi, void *out, int32_t dtype, void *stream);

/* Measurement only (no reference counterpart): vector-ALU issue-rate probe, the compute-side roof bench.py prices the
 * path against (the path is VALU-bound: ~10^3 flop per algorithmic byte).  Launches n_workgroups one-wave workgroups,
 * each running `iters` iterations of lgar_valu_probe_insts(op) instructions of kind `op`; lds_bytes_per_workgroup sets the resident waves
 * per SIMD (160 KiB / (4 k) => k).  sink: device float[n_workgroups * 64] (never written in practice).  The caller
 * times the launch on `stream`. */
#define LGAR_PROBE_EXP 0      /* v_exp_f32, 8 independent chains */
#define LGAR_PROBE_LOG 1      /* v_log_f32 */
#define LGAR_PROBE_RCP 2      /* v_rcp_f32 */
#define LGAR_PROBE_SQRT 3     /* v_sqrt_f32 */
#define LGAR_PROBE_FMA 4      /* v_fma_f32 */
#define LGAR_PROBE_MUL 5      /* v_mul_f32 */
#define LGAR_PROBE_PK_FMA 6   /* v_pk_fma_f32 */
#define LGAR_PROBE_PK_MUL 7   /* v_pk_mul_f32 */
#define LGAR_PROBE_CNDMASK 8  /* v_cndmask_b32 */
#define LGAR_PROBE_CMP 9      /* v_cmp_lt_f32 */
#define LGAR_PROBE_EXP_DEP 10 /* v_exp_f32, ONE dependent chain (latency) */
#define LGAR_PROBE_FMA_DEP 11 /* v_fma_f32, ONE dependent chain */
#define LGAR_PROBE_FMA64 12   /* v_fma_f64 */
#define LGAR_PROBE_MUL64 13   /* v_mul_f64 */
#define LGAR_PROBE_ADD64 14   /* v_add_f64 */
#define LGAR_PROBE_RCP64 15   /* v_rcp_f64 */
#define LGAR_PROBE_GEFF_MIX 16 /* the instruction stream of the lean fp32 Geff loop: per node pair 4 v_log, 4 v_exp, 10 packed */
#define LGAR_PROBE_CNDMASK_SGPR 17 /* v_cndmask_b32 with an SGPR-pair mask */
#define LGAR_PROBE_BFI 18      /* v_bfi_b32 (bitwise select on a vector mask) */
#define LGAR_PROBE_CMP_CNDMASK 19 /* v_cmp_lt_f32 + v_cndmask_b32 pairs */
#define LGAR_PROBE_ADD 20      /* v_add_f32 */
#define LGAR_PROBE_READLANE 21 /* v_readlane_b32 (what an SGPR spill reload costs) */
#define LGAR_PROBE_DS_READ 22  /* ds_read_b32, lane-contiguous */
#define LGAR_PROBE_MIN 23      /* v_min_f32 */
#define LGAR_PROBE_LDEXP64 24       /* v_ldexp_f64 */
#define LGAR_PROBE_FREXP_EXP64 25   /* v_frexp_exp_i32_f64 */
#define LGAR_PROBE_RNDNE64 26       /* v_rndne_f64 */
#define LGAR_PROBE_CVT_I32_F64 27   /* v_cvt_i32_f64 */
#define LGAR_PROBE_CVT_F64_I32 28   /* v_cvt_f64_i32 */
#define LGAR_PROBE_ADD_U32 29       /* v_add_u32 */
int32_t lgar_valu_probe_insts(int32_t op); /* wave-instructions per probe iteration: 64 (72 for LGAR_PROBE_GEFF_MIX) */
int32_t lgar_valu_probe(int32_t op, int32_t n_workgroups, int32_t lds_bytes_per_workgroup, int32_t iters, void *sink,
                        void *stream);

#ifdef __cplusplus
}
#endif
#endif
