/*
 * lgar_oracle.c -- CPU restatement of the LGAR-py hot path.  See lgar_oracle.h.
 * TEST INFRASTRUCTURE ONLY: never imported, linked or called by the product path.
 *
 * Reference paths below are relative to /root/reference/dpLGAR/.
 * Floating-point expression order follows the Python source so that results agree with the
 * reference to pow()-ulp level.
 */
#include "lgar_oracle.h"

#include <math.h>
#include <stdlib.h>
#include <string.h>
#ifdef _OPENMP
#include <omp.h>
#endif

#define TOL 1e-12 /* Layer.tolerance, models/physics/layers/Layer.py:60 */

/* ------------------------------------------------------------------------------------------ */
/* leaf functions: models/physics/utils.py                                                     */
/* ------------------------------------------------------------------------------------------ */

/* safe_pow, utils.py:12-32 */
static double spow(double b, double e, int *st) {
  if (isnan(b) || isnan(e)) *st |= LGO_ST_NAN;
  if (b < 0) *st |= LGO_ST_NEGBASE;
  return pow(b, e);
}

/* calc_theta_from_h, utils.py:35-51 */
double lgo_theta_from_h(double h, double alpha, double m, double n, double theta_e, double theta_r, int *st) {
  double ap = spow(alpha * h, n, st);
  double op = spow(1.0 + ap, m, st);
  return (1.0 / op * (theta_e - theta_r)) + theta_r;
}

/* calc_se_from_theta, utils.py:102-112 */
double lgo_se_from_theta(double theta, double theta_e, double theta_r) { return (theta - theta_r) / (theta_e - theta_r); }

/* calc_se_from_h, utils.py:115-131 */
double lgo_se_from_h(double h, double alpha, double m, double n, int *st) {
  if (fabs(h) < 1.0e-01) return 1.0;
  double is = spow(alpha * h, n, st);
  return 1.0 / spow(1.0 + is, m, st);
}

/* calc_k_from_se, utils.py:134-156.  torch.isclose(base, 0, rtol=1e-12) has atol=1e-8. */
double lgo_k_from_se(double se, double ksat, double m, int *st) {
  double sp = spow(se, 1.0 / m, st);
  double base = 1.0 - sp;
  if (fabs(base) <= 1e-8) base = base + 1e-12;
  double op = spow(base, m, st);
  return ksat * sqrt(se) * spow(1.0 - op, 2.0, st);
}

/* calc_h_from_se, utils.py:159-174 */
double lgo_h_from_se(double se, double alpha, double m, double n, int *st) {
  double sp = spow(se, (-1.0 / m), st);
  double base = sp - 1.0;
  if (fabs(base) <= 1e-8) base = base + 1e-12;
  double op = spow(base, (1.0 / n), st);
  return 1.0 / alpha * op;
}

/* calc_geff, models/physics/lgar/green_ampt.py:45-84 (use_closed_form_G False) */
double lgo_geff(double theta1, double theta2, double alpha, double n, double m, double ksat, double theta_e,
                double theta_r, int nint, int *st) {
  double se_i = lgo_se_from_theta(theta1, theta_e, theta_r);
  double se_f = lgo_se_from_theta(theta2, theta_e, theta_r);
  double h_i = lgo_h_from_se(se_i, alpha, m, n, st);
  double h_f = lgo_h_from_se(se_f, alpha, m, n, st);
  double dh = (h_f - h_i) / nint;
  double geff = 0.0;
  double k1 = lgo_k_from_se(se_i, ksat, m, st);
  double h2 = h_i + dh;
  for (int i = 0; i < nint; i++) {
    double se2 = lgo_se_from_h(h2, alpha, m, n, st);
    double k2 = lgo_k_from_se(se2, ksat, m, st);
    geff = geff + ((k1 + k2) * (dh / 2.0));
    k1 = k2;
    h2 = h2 + dh;
  }
  return fabs(geff / ksat);
}

/* calc_aet, models/physics/lgar/aet.py:17-51; 0.75 = GlobalParams.py:75 */
double lgo_aet(double pet, double dt_h, double psi, double alpha, double n, double m, double theta_e, double theta_r,
               double wp_psi, int *st) {
  double theta_fc = (theta_e - theta_r) * 0.75 + theta_r;
  double wp_head_theta = lgo_theta_from_h(wp_psi, alpha, m, n, theta_e, theta_r, st);
  double theta_wp = (theta_fc - wp_head_theta) * 0.5 + wp_head_theta;
  double se = lgo_se_from_theta(theta_wp, theta_e, theta_r);
  double psi_wp = lgo_h_from_se(se, alpha, m, n, st);
  double h_ratio = 1.0 + spow(psi / psi_wp, 3.0, st);
  double a = pet * (1 / h_ratio) * dt_h;
  /* torch.clamp(min=0.0, max=pet): the upper bound is the PET *rate* */
  if (a < 0.0) a = 0.0;
  if (a > pet) a = pet;
  return a;
}

/* calc_giuh, models/physics/lgar/giuh.py:8-20 */
double lgo_giuh(double *queue, const double *ord, int ng, double runoff) {
  for (int i = 0; i < ng; i++) queue[i] = queue[i] + (ord[i] * runoff);
  double now = queue[0];
  for (int i = 0; i < ng - 1; i++) queue[i] = queue[i + 1];
  queue[ng - 1] = 0.0;
  return now;
}

/* ------------------------------------------------------------------------------------------ */
/* set-up                                                                                      */
/* ------------------------------------------------------------------------------------------ */

void lgo_params_init(lgo_params *p, int L, const double *alpha, const double *n, const double *ksat,
                     const double *theta_e, const double *theta_r, const double *thick, double initial_psi,
                     double pdm, double wp_psi, double frozen_factor, double dt_h, int nint, int num_subcycles,
                     const double *giuh, int ngiuh) {
  memset(p, 0, sizeof(*p));
  p->L = L;
  for (int k = 0; k < L; k++) {
    p->alpha[k] = alpha[k];
    p->n[k] = n[k];
    p->m[k] = 1.0 - (1.0 / n[k]); /* calc_m, utils.py:67-69 */
    p->ksat[k] = ksat[k] * frozen_factor; /* models/dpLGAR.py:57 */
    p->theta_e[k] = theta_e[k];
    p->theta_r[k] = theta_r[k];
    p->thick[k] = thick[k];
    p->cum[k] = (k == 0) ? thick[0] : p->cum[k - 1] + thick[k]; /* GlobalParams.py:99-109 */
  }
  p->initial_psi = initial_psi;
  p->pdm = pdm;
  p->wp_psi = wp_psi;
  p->frozen_factor = frozen_factor;
  p->dt_h = dt_h;
  p->nint = nint;
  p->num_subcycles = num_subcycles;
  p->ngiuh = ngiuh;
  for (int i = 0; i < ngiuh; i++) p->giuh[i] = giuh[i];
  p->iter_cap = 2000000;
}

static double th_k(const lgo_params *p, int k, double h, int *st) {
  return lgo_theta_from_h(h, p->alpha[k], p->m[k], p->n[k], p->theta_e[k], p->theta_r[k], st);
}
static double se_k(const lgo_params *p, int k, double theta) { return lgo_se_from_theta(theta, p->theta_e[k], p->theta_r[k]); }
static double h_k(const lgo_params *p, int k, double se, int *st) { return lgo_h_from_se(se, p->alpha[k], p->m[k], p->n[k], st); }
static double K_k(const lgo_params *p, int k, double se, int *st) { return lgo_k_from_se(se, p->ksat[k], p->m[k], st); }
/* calc_geff closed form, lgar/green_ampt.py:85-98; calc_bc_lambda / calc_bc_psib, physics/utils.py:54-64,84-99 */
static double geff_closed(const lgo_params *p, int k, double theta1, double theta2, int *st) {
  double m = p->m[k], alpha = p->alpha[k];
  double pp = 1.0 + (2.0 / m);
  double bc_lambda = 2.0 / (pp - 3.0);
  double bc_psib = (pp + 3.0) * (147.8 + 8.1 * pp + 0.092 * pp * pp) / (2.0 * alpha * pp * (pp - 1.0) * (55.6 + 7.4 * pp + pp * pp));
  double se_f = lgo_se_from_theta(theta1, p->theta_e[k], p->theta_r[k]);
  double se_i = lgo_se_from_theta(theta2, p->theta_e[k], p->theta_r[k]);
  double h_c = bc_psib * (2 + 3 * bc_lambda) / (1 + 3 * bc_lambda);
  double e = (3 + 1 / bc_lambda);
  double g = h_c * (spow(se_i, e, st)) - spow(se_f, e, st) / (1 - spow(se_f, e, st));
  if (isinf(g) || isnan(g)) g = h_c;
  return g;
}

static double geff_k(const lgo_params *p, lgo_state *s, int k, double t1, double t2) {
  s->n_geff++;
  if (p->closed_form) return geff_closed(p, k, t1, t2, &s->status);
  return lgo_geff(t1, t2, p->alpha[k], p->n[k], p->m[k], p->ksat[k], p->theta_e[k], p->theta_r[k], p->nint, &s->status);
}

/* Layer.mass_balance, layers/Layer.py:795-824.  Layer sums are combined s0 + (s1 + (s2 + ...)). */
double lgo_mass_balance(const lgo_params *p, const lgo_state *s) {
  double ls[LGO_LMAX] = {0};
  for (int k = 0; k < p->L; k++) ls[k] = 0.0;
  int i = 0;
  for (int k = 0; k < p->L; k++) {
    double base = (k == 0) ? 0.0 : p->cum[k] - p->thick[k];
    double sum = 0.0;
    int lo = i;
    while (i < s->nf && s->f[i].layer == k) i++;
    int len = i - lo;
    for (int j = 0; j < len - 1; j++)
      sum = sum + (s->f[lo + j].depth - base) * (s->f[lo + j].theta - s->f[lo + j + 1].theta);
    if (len > 0) sum = sum + (s->f[lo + len - 1].depth - base) * s->f[lo + len - 1].theta;
    ls[k] = sum;
  }
  double tot = ls[p->L - 1];
  for (int k = p->L - 2; k >= 0; k--) tot = ls[k] + tot;
  return tot;
}

/* dpLGAR.set_internal_states, models/dpLGAR.py:97-147; Layer ctor Layer.py:62-72;
 * WettingFront ctor WettingFront.py:38-49; theta_init data/utils.py:80-82 */
void lgo_state_init(const lgo_params *p, lgo_state *s) {
  memset(s, 0, sizeof(*s));
  s->nf = p->L;
  for (int k = 0; k < p->L; k++) {
    lgo_front *f = &s->f[k];
    f->depth = p->cum[k];
    f->layer = k;
    f->theta = th_k(p, k, p->initial_psi, &s->status);
    f->dzdt = 0.0;
    f->psi = p->initial_psi;
    f->k = K_k(p, k, se_k(p, k, f->theta), &s->status);
    f->to_bottom = 1;
  }
  s->ending_volume = lgo_mass_balance(p, s);
}

void lgo_drain(lgo_state *s) {
  s->precip = s->PET = s->AET = s->infiltration = s->runoff = s->percolation = s->giuh_runoff = s->discharge = 0.0;
}

/* ------------------------------------------------------------------------------------------ */
/* list plumbing on the flat array                                                             */
/* ------------------------------------------------------------------------------------------ */

static void ranges(const lgo_params *p, const lgo_state *s, int *lo, int *len) {
  for (int k = 0; k < p->L; k++) { lo[k] = -1; len[k] = 0; }
  for (int i = 0; i < s->nf; i++) {
    int k = s->f[i].layer;
    if (len[k] == 0) lo[k] = i;
    len[k]++;
  }
}

/* WettingFront.is_equal (by VALUE), WettingFront.py:76-84 */
static int feq(const lgo_front *a, const lgo_front *b) { return a->depth == b->depth && a->psi == b->psi && a->dzdt == b->dzdt; }

static void fdel(lgo_state *s, int i) {
  for (int j = i; j < s->nf - 1; j++) s->f[j] = s->f[j + 1];
  s->nf--;
}

/* calc_wetting_front_free_drainage, Layer.py:134-162 (entry models/dpLGAR.py:328-338) */
static int free_drainage_front(const lgo_state *s) {
  double psi = s->f[0].psi;
  int idx = 0;
  for (int i = 0; i < s->nf; i++) {
    double pi = s->f[i].psi;
    if (pi <= psi) { psi = pi; idx = i; }
    else if (fabs(pi - psi) <= 1e-8 + 1e-5 * fabs(psi)) { psi = pi; idx = i; } /* torch.isclose(atol=1e-8), rtol 1e-5 */
  }
  return idx;
}

/* ------------------------------------------------------------------------------------------ */
/* theta_mass_balance, Layer.py:242-318; recalculate_mass Layer.py:211-240                      */
/* k = layer of the front, nlen = len(delta_thickness)                                          */
/* ------------------------------------------------------------------------------------------ */
static double theta_mass_balance(const lgo_params *p, lgo_state *s, int k, double psi, double new_mass, double prior_mass,
                                 const double *dth, const double *dthick, int nlen) {
  int *st = &s->status;
  double delta_mass = fabs(new_mass - prior_mass);
  int switched = 0;
  double factor = 1.0;
  double theta = 0.0;
  double psi_prev = psi;
  double delta_mass_prev = delta_mass;
  int count_no_change = 0;
  s->n_tmb_calls++;
  if (delta_mass <= TOL) return th_k(p, k, psi, st);
  long it = 0;
  while (delta_mass > TOL) {
    if (++it > p->iter_cap) { *st |= LGO_ST_ITERCAP; break; }
    s->n_tmb_iters++;
    if (new_mass > prior_mass) {
      psi = psi + (0.1 * factor);
      switched = 0;
    } else {
      if (!switched) { switched = 1; factor = factor * 0.1; }
      psi_prev = psi;
      psi = psi - (0.1 * factor);
      if (psi < 0 && psi_prev != 0) psi = psi_prev * 0.1;
    }
    theta = th_k(p, k, psi, st);
    double mass = 0.0;
    mass = mass + (dthick[k] * (theta - dth[k]));
    for (int j = 0; j < nlen - 1; j++) mass = mass + dthick[j] * (th_k(p, j, psi, st) - dth[j]);
    new_mass = mass;
    delta_mass = fabs(new_mass - prior_mass);
    if (fabs(psi - psi_prev) < 1e-15 && factor < 1e-13) break;
    if (fabs(delta_mass - delta_mass_prev) < 1e-15) count_no_change++; else count_no_change = 0;
    if (count_no_change == 5) break;
    if (psi <= 0 && psi_prev < 1e-50) break;
    delta_mass_prev = delta_mass;
  }
  return theta;
}

typedef struct { double depth, theta, psi; } snap_t;

/* base_case, Layer.py:320-387; populate_delta_thickness Layer.py:177-209 */
static void base_case(const lgo_params *p, lgo_state *s, const snap_t *prev, int i, int k, double percolation, double aet,
                      int fdd) {
  int *st = &s->status;
  lgo_front *cur = &s->f[i];
  double dth[LGO_LMAX], dthick[LGO_LMAX];
  for (int j = 0; j < p->L; j++) { dth[j] = 0.0; dthick[j] = 0.0; }
  cur->depth = cur->depth + cur->dzdt * p->dt_h;
  double psi_old = prev[i].psi;
  double psi = cur->psi;
  double base = (k > 0) ? p->cum[k - 1] : 0.0;
  double prior_mass = (prev[i].depth - base) * (prev[i].theta - 0.0);
  double new_mass = (cur->depth - base) * (cur->theta - 0.0);
  for (int j = 0; j < p->L - 1; j++) {
    double theta_old = th_k(p, j, psi_old, st);
    prior_mass = prior_mass + p->thick[j] * (theta_old - 0.0);
    double theta = th_k(p, j, psi, st);
    new_mass = new_mass + p->thick[j] * (theta - 0.0);
    dth[j] = 0.0;
    dthick[j] = p->thick[j];
  }
  dthick[k] = cur->depth - base;
  if (s->f[fdd].layer == k) prior_mass = prior_mass + percolation - (0 + aet);
  double theta_new = theta_mass_balance(p, s, k, psi, new_mass, prior_mass, dth, dthick, p->L);
  cur->theta = fmin(theta_new, p->theta_e[k]);
  cur->psi = h_k(p, k, se_k(p, k, cur->theta), st);
}

/* wetting_front_in_layer, Layer.py:420-547; compute_wetting_front_mass Layer.py:561-644 */
static void front_in_layer(const lgo_params *p, lgo_state *s, const snap_t *prev, int i, int k, double infiltration,
                           double aet, int fdd) {
  int *st = &s->status;
  lgo_front *cur = &s->f[i];
  lgo_front *nxt = &s->f[i + 1];
  double theta_e = p->theta_e[k];
  if (k == 0) {
    double prior_mass = prev[i].depth * (prev[i].theta - prev[i + 1].theta);
    if (i == fdd || feq(&s->f[fdd], cur)) prior_mass = prior_mass + (infiltration - (0 + aet));
    cur->depth = cur->depth + (cur->dzdt * p->dt_h);
    cur->depth = fmin(cur->depth, p->cum[p->L - 1]);
    int zero_dzdt = fabs(cur->dzdt) <= 1e-8; /* torch.isclose(dzdt, 0, rtol=1e-8): atol 1e-8 */
    if (zero_dzdt && !cur->to_bottom) {
      /* a new front was just created: theta unchanged (Layer.py:460-464) */
    } else {
      double potential = (prior_mass / cur->depth) + nxt->theta;
      cur->theta = fmin(theta_e, potential);
    }
  } else {
    double prev_thick = p->cum[k - 1];
    cur->depth = cur->depth + (cur->dzdt * p->dt_h);
    double psi_old = prev[i].psi, psi_below_old = prev[i + 1].psi;
    double psi = cur->psi, psi_below = nxt->psi;
    double prior_mass = (prev[i].depth - prev_thick) * (prev[i].theta - prev[i + 1].theta);
    double new_mass = (cur->depth - prev_thick) * (cur->theta - nxt->theta);
    double dth[LGO_LMAX], dthick[LGO_LMAX];
    for (int j = 0; j <= k; j++) { dth[j] = 0.0; dthick[j] = 0.0; }
    for (int j = 0; j < k; j++) {
      double theta_old = th_k(p, j, psi_old, st);
      double theta_below_old = th_k(p, j, psi_below_old, st);
      double local_delta_old = theta_old - theta_below_old;
      double lt = p->cum[j] - 0.0; /* quirk: cumulative, not individual, thickness (Layer.py:603-604) */
      prior_mass = prior_mass + (lt * local_delta_old);
      double theta = th_k(p, j, psi, st);
      double theta_below = th_k(p, j, psi_below, st);
      new_mass = new_mass + (lt * (theta - theta_below));
      dth[j] = theta_below;
      dthick[j] = lt;
    }
    dth[k] = nxt->theta;
    dthick[k] = cur->depth - p->cum[k - 1];
    if (i == fdd || feq(&s->f[fdd], cur)) prior_mass = prior_mass + infiltration - (0 + aet);
    double theta_new = theta_mass_balance(p, s, k, psi, new_mass, prior_mass, dth, dthick, k + 1);
    cur->theta = fmin(theta_new, theta_e);
  }
  cur->psi = h_k(p, k, se_k(p, k, cur->theta), st);
}

/* check_column_mass, Layer.py:655-701 */
static void check_column_mass(const lgo_params *p, lgo_state *s, int fdd, double old_mass, double percolation, double aet) {
  double theta_e_k1 = p->theta_e[s->f[fdd].layer];
  double mass_timestep = (old_mass + percolation) - (aet + 0.0);
  if (fabs(s->f[fdd].theta - theta_e_k1) < TOL) {
    double current_mass = lgo_mass_balance(p, s);
    double err = fabs(current_mass - mass_timestep);
    int switched = 0;
    double factor = 1.0;
    double depth_new = s->f[fdd].depth;
    long it = 0;
    while (fabs(err - TOL) > 1e-12) {
      if (++it > p->iter_cap) { s->status |= LGO_ST_ITERCAP; break; }
      s->n_ccm_iters++;
      if (current_mass < mass_timestep) {
        depth_new = depth_new + 0.01 * factor;
        switched = 0;
      } else {
        if (!switched) { switched = 1; factor = factor * 0.001; }
        depth_new = depth_new - (0.01 * factor);
      }
      s->f[fdd].depth = depth_new;
      current_mass = lgo_mass_balance(p, s);
      err = fabs(current_mass - mass_timestep);
    }
  }
}

/* Layer.move_wetting_fronts, Layer.py:1254-1307 (bottom layer first, recursing to previous layers) */
static void move_fronts_sweep(const lgo_params *p, lgo_state *s, const snap_t *prev, double infiltration, double aet,
                              double old_mass, int fdd) {
  int lo[LGO_LMAX], len[LGO_LMAX];
  ranges(p, s, lo, len);
  int nf = s->nf;
  int count = nf;
  for (int k = p->L - 1; k >= 0; k--) {
    int is_bottom = (k == p->L - 1);
    for (int ii = len[k] - 1; ii >= 0; ii--) {
      int i = lo[k] + ii;
      int last = lo[k] + len[k] - 1;
      if (count < nf) {
        if (feq(&s->f[i], &s->f[last])) {
          /* deepest_layer_front, Layer.py:389-418 */
          s->f[i].theta = th_k(p, k, s->f[i + 1].psi, &s->status);
          s->f[i].psi = s->f[i + 1].psi;
        } else {
          front_in_layer(p, s, prev, i, k, infiltration, aet, fdd);
        }
      }
      if (nf == p->L && is_bottom) base_case(p, s, prev, i, k, infiltration, aet, fdd);
      if (count == 1) check_column_mass(p, s, fdd, old_mass, infiltration, aet);
      count--;
    }
  }
}

/* merge_wetting_fronts / is_passing / pass_front / delete_front, Layer.py:826-892 */
static void merge_fronts(const lgo_params *p, lgo_state *s) {
  int *st = &s->status;
  for (int k = 0; k < p->L; k++) {
    int lo[LGO_LMAX], len[LGO_LMAX];
    ranges(p, s, lo, len);
    int nit = (k == p->L - 1) ? len[k] - 1 : len[k];
    for (int ii = 0; ii < nit; ii++) {
      int i = lo[k] + ii, nx = i + 1, nn = i + 2;
      if (nx >= s->nf) { *st |= LGO_ST_STRUCT; break; }
      lgo_front *c = &s->f[i], *n = &s->f[nx];
      int passing = (c->depth > n->depth) && (c->layer == n->layer) && !n->to_bottom;
      if (passing) {
        if (nn >= s->nf) { *st |= LGO_ST_STRUCT; break; }
        lgo_front *q = &s->f[nn];
        double mass = c->depth * (c->theta - n->theta) + n->depth * (n->theta - q->theta);
        c->depth = mass / (c->theta - q->theta);
        double se = se_k(p, k, c->theta);
        c->psi = h_k(p, k, se, st);
        c->k = K_k(p, k, se, st);
        /* delete_front: first front of THIS layer's list that is value-equal to next */
        for (int j = lo[k]; j < lo[k] + len[k]; j++)
          if (feq(&s->f[j], n)) { fdel(s, j); break; }
        break;
      }
    }
  }
}

/* wetting_fronts_cross_layer_boundary / recalibrate, Layer.py:894-1008.
 * The per-layer lists are only re-bucketed at the end (update_wetting_fronts, :939-963), which on the
 * flat array is a no-op because order is preserved; list membership inside = membership at entry. */
static void cross_layer_boundary(const lgo_params *p, lgo_state *s) {
  int *st = &s->status;
  int lo[LGO_LMAX], len[LGO_LMAX];
  ranges(p, s, lo, len);
  for (int k = 0; k < p->L; k++) {
    int nit = (k == p->L - 1) ? len[k] - 1 : len[k];
    for (int ii = 0; ii < nit; ii++) {
      int i = lo[k] + ii, nx = i + 1, nn = i + 2;
      if (nx >= s->nf) { *st |= LGO_ST_STRUCT; break; }
      lgo_front *c = &s->f[i], *n = &s->f[nx];
      if (c->depth > p->cum[k] && n->depth == p->cum[k]) {
        if (k == p->L - 1) {
          if (p->bottom_mode == 0) { *st |= LGO_ST_BOTTOM; return; } /* reference: AttributeError at Layer.py:980 */
          continue; /* LGAR-C intent: the domain-boundary step handles this front */
        }
        if (nn >= s->nf) { *st |= LGO_ST_STRUCT; return; }
        lgo_front *q = &s->f[nn];
        double overshot = c->depth - n->depth;
        double se = se_k(p, k, c->theta);
        c->psi = h_k(p, k, se, st);
        c->k = K_k(p, k, se, st);
        double theta_new = th_k(p, k + 1, c->psi, st);
        double mbal = overshot * (c->theta - n->theta);
        double zc = mbal / (theta_new - q->theta);
        double depth_new = p->cum[k] + zc;
        c->depth = p->cum[k];
        n->theta = theta_new;
        n->psi = c->psi;
        n->depth = depth_new;
        n->layer = k + 1;
        n->dzdt = c->dzdt;
        c->dzdt = 0.0;
        c->to_bottom = 1;
        n->to_bottom = 0;
      }
    }
  }
}

/* wetting_front_cross_domain_boundary, Layer.py:1010-1053 (unreachable in the reference without crashing first) */
static double cross_domain_boundary(const lgo_params *p, lgo_state *s) {
  int *st = &s->status;
  double ls[LGO_LMAX] = {0};
  for (int k = 0; k < p->L; k++) {
    int lo[LGO_LMAX], len[LGO_LMAX];
    ranges(p, s, lo, len);
    int nit = (k == p->L - 1) ? len[k] - 1 : len[k];
    double flux = 0.0;
    for (int ii = 0; ii < nit; ii++) {
      int i = lo[k] + ii, nx = i + 1, nn = i + 2;
      double tmp = 0.0;
      if (nx >= s->nf) break;
      if (nn >= s->nf) {
        lgo_front *c = &s->f[i], *n = &s->f[nx];
        if (c->depth > p->cum[k]) {
          tmp = (c->theta - n->theta) * (c->depth - n->depth);
          n->theta = c->theta;
          double se = se_k(p, k, c->theta);
          n->psi = h_k(p, k, se, st);
          n->k = K_k(p, k, se, st);
          fdel(s, i);
          if (p->bottom_mode == 0) *st |= LGO_ST_BOTTOM;
        }
      }
      flux = flux + tmp;
    }
    ls[k] = flux;
  }
  double tot = ls[p->L - 1];
  for (int k = p->L - 2; k >= 0; k--) tot = ls[k] + tot;
  return tot;
}

/* fix_dry_over_wet_fronts / cleanup_wetting_fronts / update_layer_fronts, Layer.py:1055-1143 */
static double fix_dry_over_wet(const lgo_params *p, lgo_state *s) {
  int *st = &s->status;
  double ls[LGO_LMAX] = {0};
  for (int k = 0; k < p->L; k++) {
    int lo[LGO_LMAX], len[LGO_LMAX];
    ranges(p, s, lo, len);
    double mass_change = 0.0;
    for (int ii = 0; ii < len[k]; ii++) {
      int i = lo[k] + ii, nx = i + 1;
      if (nx >= s->nf) continue;
      if (s->f[i].theta <= s->f[nx].theta && s->f[i].layer == s->f[nx].layer) {
        double before = lgo_mass_balance(p, s);
        int popped_layer = s->f[i].layer;
        fdel(s, i); /* next front is now at index i */
        if (popped_layer > 0) {
          lgo_front target = s->f[i];
          int j, found = -1;
          for (j = 0; j < s->nf; j++)
            if (feq(&s->f[j], &target)) { found = j; break; }
          if (found < 0) { *st |= LGO_ST_STRUCT; }
          else {
            int lj = s->f[found].layer;
            s->f[found].psi = h_k(p, lj, se_k(p, lj, s->f[found].theta), st);
            lgo_front dry = s->f[found];
            for (int q = 0; q < s->nf; q++) {
              int lq = s->f[q].layer;
              if (lq < dry.layer) {
                s->f[q].psi = h_k(p, lq, se_k(p, lq, dry.theta), st);
                s->f[q].theta = th_k(p, lq, dry.psi, st);
              }
            }
          }
        }
        double after = lgo_mass_balance(p, s);
        mass_change = mass_change + fabs(after - before);
        break;
      }
    }
    ls[k] = mass_change;
  }
  double tot = ls[p->L - 1];
  for (int k = p->L - 2; k >= 0; k--) tot = ls[k] + tot;
  return tot;
}

/* update_psi, Layer.py:1157-1174: every front except the deepest front of the domain */
static void update_psi(const lgo_params *p, lgo_state *s) {
  int lo[LGO_LMAX], len[LGO_LMAX];
  ranges(p, s, lo, len);
  for (int k = 0; k < p->L; k++) {
    int nit = (k == p->L - 1) ? len[k] - 1 : len[k];
    for (int ii = 0; ii < nit; ii++) {
      lgo_front *c = &s->f[lo[k] + ii];
      double se = se_k(p, k, c->theta);
      c->psi = h_k(p, k, se, &s->status);
      c->k = K_k(p, k, se, &s->status);
    }
  }
}

/* dpLGAR.move_wetting_front, models/dpLGAR.py:340-367.  Returns bottom flux; *aet updated. */
static double move_wetting_front(const lgo_params *p, lgo_state *s, const snap_t *prev, double infiltration, double *aet,
                                 double old_mass, int fdd) {
  move_fronts_sweep(p, s, prev, infiltration, *aet, old_mass, fdd);
  merge_fronts(p, s);
  cross_layer_boundary(p, s);
  merge_fronts(p, s);
  double bottom_flux = 0.0 + cross_domain_boundary(p, s);
  double mass_change = fix_dry_over_wet(p, s);
  if (fabs(mass_change) > 1e-7) *aet = *aet - mass_change;
  update_psi(p, s);
  return bottom_flux;
}

/* calc_dzdt, Layer.py:1176-1252; calc_bottom_sum Layer.py:1557-1582 */
static void calc_dzdt(const lgo_params *p, lgo_state *s, double h_p) {
  int *st = &s->status;
  int lo[LGO_LMAX], len[LGO_LMAX];
  ranges(p, s, lo, len);
  for (int k = 0; k < p->L; k++) {
    int nit = (k == p->L - 1) ? len[k] - 1 : len[k];
    for (int ii = 0; ii < nit; ii++) {
      int i = lo[k] + ii;
      lgo_front *c = &s->f[i], *n = &s->f[i + 1];
      double bottom_sum = 0.0;
      double theta_1 = n->theta, theta_2 = c->theta;
      if (c->to_bottom) { c->dzdt = 0.0; continue; }
      if (c->layer > 0) {
        bottom_sum = bottom_sum + (c->depth - p->cum[k - 1]) / c->k;
      } else if (theta_1 > theta_2) {
        *st |= LGO_ST_THETA_ORDER; /* reference raises ValueError, Layer.py:1206-1208 */
      }
      double geff = geff_k(p, s, k, theta_1, theta_2);
      double delta_theta = c->theta - n->theta;
      double dzdt;
      if (c->layer == 0) {
        if (delta_theta > 0) dzdt = 1.0 / delta_theta * (p->ksat[k] * (geff + h_p) / c->depth + c->k);
        else dzdt = 0.0;
      } else {
        double den = bottom_sum;
        for (int j = 0; j < c->layer; j++) {
          double tl = th_k(p, j, c->psi, st);
          double kl = K_k(p, j, se_k(p, j, tl), st);
          double pt = (j != 0) ? p->cum[j - 1] : 0.0;
          den = den + ((p->cum[j] - pt) / kl);
        }
        double num = c->depth;
        if (delta_theta > 0) dzdt = (1.0 / delta_theta) * ((num / den) + p->ksat[k] * (geff + h_p) / c->depth);
        else dzdt = 0.0;
      }
      c->dzdt = dzdt;
    }
  }
}

/* calc_dry_depth, Layer.py:1309-1334 */
static double calc_dry_depth(const lgo_params *p, lgo_state *s) {
  lgo_front *c = &s->f[0];
  double theta_e = p->theta_e[0];
  double delta_theta = theta_e - c->theta;
  double tau = p->dt_h * p->ksat[0] / delta_theta;
  double geff = geff_k(p, s, 0, c->theta, theta_e);
  double dry = 0.5 * (tau + sqrt(tau * tau + 4.0 * tau * geff));
  return fmin(p->cum[0], dry);
}

/* Layer.create_surficial_front, Layer.py:1336-1416 */
static void create_surficial_front(const lgo_params *p, lgo_state *s, double dry_depth, double *ponded, double *infiltration) {
  int *st = &s->status;
  if (s->nf >= LGO_FMAX) { *st |= LGO_ST_OVERFLOW; return; }
  double theta_e = p->theta_e[0];
  double cur_theta = s->f[0].theta;
  double delta_theta = theta_e - cur_theta;
  double theta_new;
  int to_bottom = 0;
  if (dry_depth * delta_theta > *ponded) {
    *infiltration = *ponded;
    theta_new = fmin((cur_theta + *ponded / dry_depth), theta_e);
    *ponded = 0.0;
  } else {
    *infiltration = dry_depth * delta_theta;
    *ponded = *ponded - (dry_depth * delta_theta);
    theta_new = theta_e;
    to_bottom = !(dry_depth < p->cum[0]);
  }
  for (int j = s->nf; j > 0; j--) s->f[j] = s->f[j - 1];
  s->nf++;
  lgo_front *nf = &s->f[0];
  nf->depth = dry_depth;
  nf->theta = theta_new;
  nf->layer = 0;
  nf->to_bottom = to_bottom;
  double se = se_k(p, 0, theta_new);
  nf->psi = h_k(p, 0, se, st);
  nf->k = K_k(p, 0, se, st) * p->frozen_factor;
  nf->dzdt = 0.0;
}

/* insert_water, Layer.py:1418-1536; get_drainage_neighbors :1584-1607; calc_bottom_sum_f_p :1538-1555 */
static void insert_water(const lgo_params *p, lgo_state *s, int fdd, double precip, double *ponded, double *infiltration,
                         double *runoff) {
  int *st = &s->status;
  double dt = p->dt_h;
  double h_p = (*ponded - precip) * dt;
  if (h_p < 0.0) h_p = 0.0;
  int lo[LGO_LMAX], len[LGO_LMAX];
  ranges(p, s, lo, len);
  lgo_front *fd = &s->f[fdd];
  int kfp = fd->layer;
  /* get_drainage_neighbors(0): (first front of fdd's layer, fdd, the front after that FIRST front) */
  int cur_i = lo[kfp];
  int nxt_i = cur_i + 1;
  /* reference: AttributeError at Layer.py:1606 when the free-drainage front is the lone front of the bottom layer.
   * With one front per layer no neighbour is needed (Geff = 0); bottom_mode 1 lets that case through. */
  if (nxt_i >= s->nf && (s->nf != p->L || p->bottom_mode == 0)) { *st |= LGO_ST_STRUCT; return; }
  double geff;
  if (s->nf == p->L) geff = 0.0;
  else geff = geff_k(p, s, kfp, s->f[nxt_i].theta, p->theta_e[kfp]);
  double f_p;
  if (kfp == 0) {
    f_p = p->ksat[0] * (1 + (geff + h_p) / fd->depth);
  } else {
    double fd_ksat = p->ksat[kfp] * p->frozen_factor;
    double bottom_sum = (fd->depth - p->cum[kfp - 1]) / fd_ksat;
    /* top layer: Ksat * frozen; deeper layers: K(psi_fdd) (calc_bottom_sum) */
    bottom_sum = bottom_sum + ((p->cum[0] - 0.0) / (p->ksat[0] * p->frozen_factor));
    for (int j = 1; j < kfp; j++) {
      double tl = th_k(p, j, fd->psi, st);
      double kl = K_k(p, j, se_k(p, j, tl), st);
      bottom_sum = bottom_sum + ((p->cum[j] - p->cum[j - 1]) / kl);
    }
    f_p = (fd->depth / bottom_sum) + ((geff + h_p) * fd_ksat / fd->depth);
  }
  /* the f_p = 0 override (Layer.py:1495-1500) needs layer_num == num_layers: never true (0-based) */
  double pond_temp = *ponded - f_p * dt - 0.0 * 0;
  if (pond_temp < 0.0) pond_temp = 0.0;
  double fp_cm = f_p * dt + 0.0 / dt;
  if (p->pdm > 0.0) {
    if (pond_temp < p->pdm) {
      *infiltration = fmin(*ponded, fp_cm);
      *ponded = *ponded - *infiltration;
    } else if (pond_temp > p->pdm) {
      *ponded = p->pdm;
      *infiltration = fp_cm;
    }
    double r = pond_temp - p->pdm;
    *runoff = r > 0 ? r : 0.0;
  } else {
    *infiltration = fmin(*ponded, fp_cm);
    double r = *ponded - *infiltration;
    *ponded = p->pdm;
    *runoff = r > 0.0 ? r : 0.0;
  }
}

/* dpLGAR.forward, models/dpLGAR.py:154-299 */
void lgo_forward(const lgo_params *p, lgo_state *s, double precip, double pet) {
  int *st = &s->status;
  double dt = p->dt_h;
  double ending_volume_sub = s->ending_volume;
  for (int sub = 0; sub < p->num_subcycles; sub++) {
    if (*st & (LGO_ST_BOTTOM | LGO_ST_OVERFLOW | LGO_ST_STRUCT)) return; /* dead column: reference would have raised */
    snap_t prev[LGO_FMAX];
    for (int i = 0; i < s->nf; i++) { prev[i].depth = s->f[i].depth; prev[i].theta = s->f[i].theta; prev[i].psi = s->f[i].psi; }
    double precip_sub = precip * dt;
    double pet_sub = pet * dt;
    double previous_precip_sub = s->previous_precip;
    double ponded_depth_sub = precip_sub + s->ponded_water;
    double ponded_water_sub = 0.0, percolation_sub = 0.0, runoff_sub = 0.0, infiltration_sub = 0.0, AET_sub = 0.0;
    /* create_surficial_front predicate, models/dpLGAR.py:310-323 */
    int create = (previous_precip_sub == 0.0) && (precip_sub > 0.0) && (s->ponded_water == 0);
    int fdd = free_drainage_front(s);
    int saturated = s->f[0].theta >= p->theta_e[0]; /* Layer.is_saturated, Layer.py:785-793 */
    if (pet > 0.0)
      AET_sub = lgo_aet(pet, dt, s->f[0].psi, p->alpha[0], p->n[0], p->m[0], p->theta_e[0], p->theta_r[0], p->wp_psi, st);
    s->precip = s->precip + precip_sub;
    s->PET = s->PET + (pet_sub > 0.0 ? pet_sub : 0.0);
    (void)lgo_mass_balance; /* starting_volume_sub only feeds the unused local_mb */
    if (create && !saturated) {
      (void)move_wetting_front(p, s, prev, 0.0, &AET_sub, ending_volume_sub, fdd);
      double dry_depth = calc_dry_depth(p, s);
      create_surficial_front(p, s, dry_depth, &ponded_depth_sub, &infiltration_sub);
      s->infiltration = s->infiltration + infiltration_sub;
    }
    if (!create && ponded_depth_sub > 0) {
      insert_water(p, s, fdd, precip_sub, &ponded_depth_sub, &infiltration_sub, &runoff_sub);
      s->infiltration = s->infiltration + infiltration_sub;
      s->runoff = s->runoff + runoff_sub;
      ponded_water_sub = ponded_depth_sub;
    } else {
      /* update_ponded_depth, models/dpLGAR.py:369-382 */
      if (ponded_depth_sub < p->pdm) {
        runoff_sub = 0.0;
        s->runoff = s->runoff + runoff_sub;
        ponded_water_sub = ponded_depth_sub;
        ponded_depth_sub = 0.0;
      } else {
        runoff_sub = ponded_depth_sub - p->pdm;
        ponded_depth_sub = p->pdm;
        ponded_water_sub = ponded_depth_sub;
        s->runoff = s->runoff + runoff_sub;
      }
    }
    if (!create) {
      double bottom_flux = move_wetting_front(p, s, prev, infiltration_sub, &AET_sub, ending_volume_sub, fdd);
      percolation_sub = bottom_flux;
      s->percolation = s->percolation + percolation_sub;
    }
    calc_dzdt(p, s, ponded_depth_sub);
    ending_volume_sub = lgo_mass_balance(p, s);
    s->previous_precip = precip_sub;
    s->ending_volume = ending_volume_sub;
    s->AET = s->AET + AET_sub;
    s->ponded_water = ponded_water_sub;
    /* GIUH, models/dpLGAR.py:292-298 */
    double qsum = 0.0;
    for (int i = 0; i < p->ngiuh; i++) qsum += s->giuh_queue[i];
    if (qsum > 0 || runoff_sub > 0) {
      double g = lgo_giuh(s->giuh_queue, p->giuh, p->ngiuh, runoff_sub);
      s->giuh_runoff = s->giuh_runoff + g;
      s->discharge = s->discharge + g;
    }
    for (int i = 0; i < s->nf; i++)
      if (isnan(s->f[i].theta) || isnan(s->f[i].depth) || isnan(s->f[i].psi)) *st |= LGO_ST_NAN;
  }
}

void lgo_run(const lgo_params *p, lgo_state *s, int T, const double *precip, const double *pet, double *out_acc,
             int frec, double *out_fronts, signed char *out_layer, signed char *out_bottom, int *out_nf) {
  for (int t = 0; t < T; t++) {
    lgo_forward(p, s, precip[t], pet[t]);
    if (out_acc) {
      double *a = out_acc + (size_t)t * 10;
      a[0] = s->precip; a[1] = s->PET; a[2] = s->AET; a[3] = s->infiltration; a[4] = s->runoff;
      a[5] = s->percolation; a[6] = s->giuh_runoff; a[7] = s->discharge; a[8] = s->ponded_water; a[9] = s->ending_volume;
    }
    if (out_nf) out_nf[t] = s->nf;
    if (out_fronts) {
      for (int i = 0; i < frec; i++) {
        double *r = out_fronts + ((size_t)t * frec + i) * 5;
        if (i < s->nf) {
          r[0] = s->f[i].depth; r[1] = s->f[i].theta; r[2] = s->f[i].psi; r[3] = s->f[i].k; r[4] = s->f[i].dzdt;
        } else r[0] = r[1] = r[2] = r[3] = r[4] = 0.0;
        if (out_layer) out_layer[(size_t)t * frec + i] = (i < s->nf) ? (signed char)s->f[i].layer : -1;
        if (out_bottom) out_bottom[(size_t)t * frec + i] = (i < s->nf) ? (signed char)s->f[i].to_bottom : 0;
      }
    }
    lgo_drain(s);
  }
}

void lgo_run_columns(int N, int L, int T, const double *alpha, const double *n, const double *ksat,
                     const double *theta_e, const double *theta_r, const double *thick, double initial_psi,
                     double pdm, double wp_psi, double frozen_factor, double dt_h, int nint, int num_subcycles,
                     const double *giuh, int ngiuh, const double *precip, const double *pet, double *out_runoff,
                     double *out_perc, double *out_acc, int *status, int nthreads) {
#ifdef _OPENMP
  if (nthreads > 0) omp_set_num_threads(nthreads);
#else
  (void)nthreads;
#endif
#pragma omp parallel for schedule(dynamic, 16)
  for (int c = 0; c < N; c++) {
    double a[LGO_LMAX], nn[LGO_LMAX], ks[LGO_LMAX], te[LGO_LMAX], tr[LGO_LMAX], th[LGO_LMAX];
    for (int k = 0; k < L; k++) {
      a[k] = alpha[(size_t)k * N + c]; nn[k] = n[(size_t)k * N + c]; ks[k] = ksat[(size_t)k * N + c];
      te[k] = theta_e[(size_t)k * N + c]; tr[k] = theta_r[(size_t)k * N + c]; th[k] = thick[(size_t)k * N + c];
    }
    lgo_params p;
    lgo_state s;
    lgo_params_init(&p, L, a, nn, ks, te, tr, th, initial_psi, pdm, wp_psi, frozen_factor, dt_h, nint, num_subcycles, giuh, ngiuh);
    lgo_state_init(&p, &s);
    double tot[8] = {0, 0, 0, 0, 0, 0, 0, 0};
    for (int t = 0; t < T; t++) {
      lgo_forward(&p, &s, precip[(size_t)t * N + c], pet[(size_t)t * N + c]);
      if (out_runoff) out_runoff[(size_t)t * N + c] = s.runoff;
      if (out_perc) out_perc[(size_t)t * N + c] = s.percolation;
      tot[0] += s.precip; tot[1] += s.PET; tot[2] += s.AET; tot[3] += s.infiltration; tot[4] += s.runoff;
      tot[5] += s.percolation; tot[6] += s.giuh_runoff; tot[7] += s.discharge;
      lgo_drain(&s);
    }
    if (out_acc) {
      for (int j = 0; j < 8; j++) out_acc[(size_t)j * N + c] = tot[j];
      out_acc[(size_t)8 * N + c] = s.ponded_water;
      out_acc[(size_t)9 * N + c] = s.ending_volume;
    }
    if (status) status[c] = s.status;
  }
}
