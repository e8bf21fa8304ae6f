/*
 * lgar_oracle.h -- CPU restatement (plain C, fp64, one soil column at a time) of the LGAR-py
 * per-timestep wetting-front update.
 *
 * THIS IS TEST INFRASTRUCTURE, NOT PRODUCT CODE.  Only tests/, __graft_entry__.smoke() and
 * bench.py's cpu_baseline leg may load it, and only as the checker / reported CPU baseline.
 * The product path (lgar-py_amd/) never links, imports or falls back to anything in oracle/.
 *
 * Parity pin: checked against golden vectors captured by importing the Python reference itself
 * (tests/golden/make_golden.py -> tests/golden/*.npz); see tests/test_oracle_golden.py.
 *
 * Every function cites the reference file:line it follows (paths relative to the reference root,
 * dpLGAR/...).  The reference keeps one Python list of WettingFront objects per Layer object in a
 * doubly linked list; here the fronts of a column are ONE flat array ordered top -> bottom, each
 * front tagged with its layer number.  "Layer k's list" == the contiguous run of fronts tagged k.
 */
#ifndef LGAR_ORACLE_H
#define LGAR_ORACLE_H

#ifdef __cplusplus
extern "C" {
#endif

#define LGO_LMAX 8   /* soil layers */
#define LGO_FMAX 32  /* wetting fronts per column (reference lists are unbounded) */
#define LGO_GMAX 16  /* GIUH ordinates */

/* per-column status bits (the reference raises Python exceptions instead; SURVEY 8b "Errors") */
#define LGO_ST_NAN 1          /* NaN produced (physics/utils.py:17-19,181-183) */
#define LGO_ST_NEGBASE 2      /* pow of a negative base (physics/utils.py:25-27) */
#define LGO_ST_THETA_ORDER 4  /* theta_1 > theta_2 in layer 0 (layers/Layer.py:1206-1208) */
#define LGO_ST_OVERFLOW 8     /* more than FMAX fronts */
#define LGO_ST_ITERCAP 16     /* a line search hit the iteration cap (the reference loops are unbounded) */
#define LGO_ST_BOTTOM 32      /* a front reached the domain bottom (reference crashes, Layer.py:980) */
#define LGO_ST_STRUCT 64      /* missing neighbour / front not found (reference AttributeError/IndexError) */

typedef struct {
  double depth, theta, psi, k, dzdt;
  int layer;
  int to_bottom;
} lgo_front;

typedef struct {
  int L;
  double alpha[LGO_LMAX], n[LGO_LMAX], m[LGO_LMAX], ksat[LGO_LMAX];
  double theta_e[LGO_LMAX], theta_r[LGO_LMAX], thick[LGO_LMAX], cum[LGO_LMAX];
  double initial_psi, pdm, wp_psi, frozen_factor, dt_h;
  int nint, num_subcycles, ngiuh;
  double giuh[LGO_GMAX];
  long iter_cap; /* cap for the two unbounded line searches */
  int closed_form; /* cfg.data.use_closed_form_G (lgar/green_ampt.py:85-98) */
  int bottom_mode; /* 0 = reference (a front at the domain bottom kills the column); 1 = LGAR-C intent: percolate */
} lgo_params;

typedef struct {
  int nf;
  lgo_front f[LGO_FMAX];
  double ponded_water, previous_precip, ending_volume;
  double giuh_queue[LGO_GMAX];
  /* accumulators drained by MassBalance.change_mass (physics/MassBalance.py:31-53) */
  double precip, PET, AET, infiltration, runoff, percolation, giuh_runoff, discharge;
  int status;
  /* instrumentation */
  long n_geff, n_tmb_calls, n_tmb_iters, n_ccm_iters;
} lgo_state;

/* leaf functions (physics/utils.py) */
double lgo_theta_from_h(double h, double alpha, double m, double n, double theta_e, double theta_r, int *st);
double lgo_se_from_theta(double theta, double theta_e, double theta_r);
double lgo_se_from_h(double h, double alpha, double m, double n, int *st);
double lgo_k_from_se(double se, double ksat, double m, int *st);
double lgo_h_from_se(double se, double alpha, double m, double n, int *st);
double lgo_geff(double theta1, double theta2, double alpha, double n, double m, double ksat, double theta_e,
                double theta_r, int nint, int *st);
double lgo_aet(double pet, double dt_h, double psi, double alpha, double n, double m, double theta_e, double theta_r,
               double wp_psi, int *st);
double lgo_giuh(double *queue, const double *ordinates, int ng, double runoff);

/* fill derived params (m, cum) and defaults */
void lgo_params_init(lgo_params *p, int L, const double *alpha, const double *n, const double *ksat,
                     const double *theta_e, const double *theta_r, const double *thick, double initial_psi,
                     double pdm, double wp_psi, double frozen_factor, double dt_h, int nint, int num_subcycles,
                     const double *giuh, int ngiuh);
/* dpLGAR.set_internal_states (models/dpLGAR.py:97-147) */
void lgo_state_init(const lgo_params *p, lgo_state *s);
/* one dpLGAR.forward(x) (models/dpLGAR.py:154-299); accumulators are NOT drained here */
void lgo_forward(const lgo_params *p, lgo_state *s, double precip, double pet);
/* MassBalance.change_mass drain (physics/MassBalance.py:45-53) */
void lgo_drain(lgo_state *s);
double lgo_mass_balance(const lgo_params *p, const lgo_state *s);

/*
 * Run T forcing steps for one column, draining after each like the agent loop
 * (agents/DifferentiableLGAR.py:117-125).  out_acc[T][10] = precip, PET, AET, infiltration, runoff,
 * percolation, giuh_runoff, discharge, ponded_water, ending_volume (values before the drain).
 * Optional per-step front tables: out_fronts[T][frec][5] (depth,theta,psi,k,dzdt), out_layer[T][frec],
 * out_bottom[T][frec], out_nf[T] (any may be NULL).
 */
void lgo_run(const lgo_params *p, lgo_state *s, int T, const double *precip, const double *pet, double *out_acc,
             int frec, double *out_fronts, signed char *out_layer, signed char *out_bottom, int *out_nf);

/*
 * Many columns (the CPU baseline timed beside the GPU): SoA inputs column-fastest,
 * params [L][N], forcing [T][N]; out_runoff/out_perc [T][N] (may be NULL), out_acc [10][N] totals
 * over the run, status [N].  Parallel over columns with OpenMP when built with -fopenmp.
 */
void lgo_run_columns(int N, int L, int T, const double *alpha, const double *n, const double *ksat,
                     const double *theta_e, const double *theta_r, const double *thick, double initial_psi,
                     double pdm, double wp_psi, double frozen_factor, double dt_h, int nint, int num_subcycles,
                     const double *giuh, int ngiuh, const double *precip, const double *pet, double *out_runoff,
                     double *out_perc, double *out_acc, int *status, int nthreads);

#ifdef __cplusplus
}
#endif
#endif
