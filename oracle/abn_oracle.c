/*
 * abn_oracle.c — CPU ORACLE for the ABneutral hot path.  TEST INFRASTRUCTURE ONLY (see abn_oracle.h).
 *
 * Every function cites the reference lines (relative to /root/reference) it restates.  Arithmetic is
 * IEEE f64 in source order with contraction disabled; the only fused operations are the ones the
 * reference effectively executes:
 *   - 3x3 * 3x3 `ndarray::dot` -> matrixmultiply dgemm FMA micro-kernel: every output element is
 *     acc = fma(a_ik, b_kj, acc), k ascending, acc0 = 0 (pinned by src/structs.rs:233 bit-exactly);
 *   - 1x3 * 3x3 `dot` -> gemv with the same k-ascending accumulation.  Whether that one fuses is not
 *     discriminated by any reference fixture (both variants reproduce every golden value); the fused
 *     form is used here and in the HIP kernel.
 *   - ndarray `var` uses mul_add (analysis only).
 * The `lanes` argument of the cost / fit functions selects the residual summation order: 1 = the reference's serial
 * order; G (<= 64) = G accumulators (rows l, l+G, ...) and an xor-butterfly 1, 2, 4, ...; G | 3 << 8 = four-row
 * blocks (the stream kernels); 0x10040 = the canonical tree of the HIP path (64 accumulators, high lane bits first).
 * Pinning: golden vectors and known answers of the reference's own tests (tests/test_oracle_golden.py) and the
 * reference-held R outputs (tests/test_r_anchors.py); the argmin 0.8.1 trajectory itself is "parity unpinned".
 * Build: gcc -O2 -ffp-contract=off -fopenmp -fPIC -shared (see oracle/Makefile).
 */
#pragma STDC FP_CONTRACT OFF

#include "abn_oracle.h"

#include <math.h>
#include <stdlib.h>
#include <string.h>
#ifdef _OPENMP
#include <omp.h>
#endif

#define ABO_HOT

static inline double fma3(double a, double b, double c) { return __builtin_fma(a, b, c); }

/* ------------------------------------------------------------------------------------------------
 * src/divergence.rs:96-114  genmatrix.  powi(2) is x*x.
 * ---------------------------------------------------------------------------------------------- */
void abo_genmatrix(double alpha, double beta, double G[9]) {
  G[0] = (1.0 - alpha) * (1.0 - alpha);
  G[1] = 2.0 * (1.0 - alpha) * alpha;
  G[2] = alpha * alpha;
  G[3] = 0.25 * ((beta + 1.0 - alpha) * (beta + 1.0 - alpha));
  G[4] = 0.5 * (beta + 1.0 - alpha) * (alpha + 1.0 - beta);
  G[5] = 0.25 * ((alpha + 1.0 - beta) * (alpha + 1.0 - beta));
  G[6] = beta * beta;
  G[7] = 2.0 * (1.0 - beta) * beta;
  G[8] = (1.0 - beta) * (1.0 - beta);
}

/* one `result.dot(matrix)` of src/divergence.rs:28 : R <- R * M, FMA-accumulated, k ascending */
ABO_HOT static void matmul3(const double R[9], const double M[9], double out[9]) {
  for (int i = 0; i < 3; ++i)
    for (int j = 0; j < 3; ++j) {
      double acc = 0.0;
      acc = fma3(R[3 * i + 0], M[0 + j], acc);
      acc = fma3(R[3 * i + 1], M[3 + j], acc);
      acc = fma3(R[3 * i + 2], M[6 + j], acc);
      out[3 * i + j] = acc;
    }
}

/* `sv_gzero.t().dot(&M)` of src/divergence.rs:55 */
ABO_HOT static void vecmat3(const double v[3], const double M[9], double out[3]) {
  for (int j = 0; j < 3; ++j) {
    double acc = 0.0;
    acc = fma3(v[0], M[0 + j], acc);
    acc = fma3(v[1], M[3 + j], acc);
    acc = fma3(v[2], M[6 + j], acc);
    out[j] = acc;
  }
}

/* src/divergence.rs:16-31 */
int abo_matrix_power(const double M[9], int power, double out[9]) {
  if (power < 0) return ABO_ERR_BAD_PEDIGREE; /* reference inverts M (divergence.rs:17-19); out of scope */
  if (power == 0) {
    for (int i = 0; i < 9; ++i) out[i] = (i % 4 == 0) ? 1.0 : 0.0;
    return ABO_OK;
  }
  double r[9], t[9];
  memcpy(r, M, sizeof r);
  for (int k = 1; k < power; ++k) {
    matmul3(r, M, t);
    memcpy(r, t, sizeof r);
  }
  memcpy(out, r, sizeof r);
  return ABO_OK;
}

/* src/alphabeta.rs:62-65 */
double abo_p_uu_est(double alpha, double beta) {
  return (beta * ((1.0 - beta) * (1.0 - beta) - (1.0 - alpha) * (1.0 - alpha) - 1.0)) /
         ((alpha + beta) * ((alpha + beta - 1.0) * (alpha + beta - 1.0) - 2.0));
}
/* src/structs.rs:146-149 */
double abo_est_mm(double alpha, double beta) {
  return (alpha * ((1.0 - alpha) * (1.0 - alpha) - (1.0 - beta) * (1.0 - beta) - 1.0)) /
         ((alpha + beta) * ((alpha + beta - 1.0) * (alpha + beta - 1.0) - 2.0));
}
/* src/structs.rs:151-154 */
double abo_est_um(double alpha, double beta) {
  return (4.0 * alpha * beta * (alpha + beta - 2.0)) /
         ((alpha + beta) * ((alpha + beta - 1.0) * (alpha + beta - 1.0) - 2.0));
}
/* src/alphabeta.rs:73-79 */
double abo_steady_state(double alpha, double beta) {
  double pi_2 = (4.0 * alpha * beta * (alpha + beta - 2.0)) /
                ((alpha + beta) * ((alpha + beta - 1.0) * (alpha + beta - 1.0) - 2.0));
  return abo_est_mm(alpha, beta) + 0.5 * pi_2;
}

/* Rust `f64 as i8`: truncate toward zero, saturate, NaN -> 0 (src/divergence.rs:52) */
int abo_as_i8(double x) {
  if (x != x) return 0;
  if (x >= 127.0) return 127;
  if (x <= -128.0) return -128;
  return (int)x;
}

int abo_check_pedigree(const double* ped, int n) {
  if (!ped || n <= 0) return ABO_ERR_INVALID_ARG;
  for (int i = 0; i < n; ++i) {
    int t0 = abo_as_i8(ped[4 * i + 0]), t1 = abo_as_i8(ped[4 * i + 1]), t2 = abo_as_i8(ped[4 * i + 2]);
    if (t0 < 0 || t1 < t0 || t2 < t0) return ABO_ERR_BAD_PEDIGREE;
  }
  return ABO_OK;
}

/* conditional divergence of one start state: src/divergence.rs:68-87 (this exact association) */
static inline double cond_div(const double* a, const double* b) {
  return 0.5 * (a[0] * b[1] + a[1] * b[0] + a[1] * b[2] + a[2] * b[1]) + (a[0] * b[2] + a[2] * b[0]);
}

/* src/divergence.rs:33-94, reference-shaped */
ABO_HOT int abo_divergence(const double* ped, int n, double p_mm, double p_uu, double alpha,
                           double beta, double weight, double* dt1t2, double* p_uu_inf) {
  double sv0[3] = {p_uu, weight * p_mm, (1.0 - weight) * p_mm}; /* :44 */
  double G[9];
  abo_genmatrix(alpha, beta, G); /* :47 */
  for (int i = 0; i < n; ++i) {
    int t0 = abo_as_i8(ped[4 * i + 0]), t1 = abo_as_i8(ped[4 * i + 1]), t2 = abo_as_i8(ped[4 * i + 2]);
    double P0[9], A[9], B[9], s0[3];
    if (abo_matrix_power(G, t0, P0)) return ABO_ERR_BAD_PEDIGREE;
    vecmat3(sv0, P0, s0);                                                /* :55 */
    if (abo_matrix_power(G, t1 - t0, A)) return ABO_ERR_BAD_PEDIGREE;    /* :57 */
    if (abo_matrix_power(G, t2 - t0, B)) return ABO_ERR_BAD_PEDIGREE;    /* :58 */
    double d_mm = cond_div(A + 6, B + 6);                                /* :68-73 */
    double d_um = cond_div(A + 3, B + 3);                                /* :75-80 */
    double d_uu = cond_div(A + 0, B + 0);                                /* :82-87 */
    dt1t2[i] = s0[0] * d_uu + s0[1] * d_um + s0[2] * d_mm;               /* :89 */
  }
  if (p_uu_inf) *p_uu_inf = abo_p_uu_est(alpha, beta); /* :92 */
  return ABO_OK;
}

/* Same bits with the power table G^0..G^T built once.  G^k is the SAME left-accumulated chain the
 * reference computes for every k (result = result.dot(matrix), :25-30), so entries are bit-identical. */
ABO_HOT int abo_divergence_table(const double* ped, int n, double p_mm, double p_uu, double alpha,
                                 double beta, double weight, double* dt1t2, double* p_uu_inf) {
  double sv0[3] = {p_uu, weight * p_mm, (1.0 - weight) * p_mm};
  double tab[128][9];
  int tmax = 0;
  for (int i = 0; i < n; ++i) {
    int t0 = abo_as_i8(ped[4 * i + 0]), t1 = abo_as_i8(ped[4 * i + 1]), t2 = abo_as_i8(ped[4 * i + 2]);
    if (t0 < 0 || t1 < t0 || t2 < t0) return ABO_ERR_BAD_PEDIGREE;
    if (t1 > tmax) tmax = t1;
    if (t2 > tmax) tmax = t2;
  }
  for (int i = 0; i < 9; ++i) tab[0][i] = (i % 4 == 0) ? 1.0 : 0.0;
  abo_genmatrix(alpha, beta, tab[1]);
  for (int k = 2; k <= tmax; ++k) matmul3(tab[k - 1], tab[1], tab[k]);
  for (int i = 0; i < n; ++i) {
    int t0 = abo_as_i8(ped[4 * i + 0]), t1 = abo_as_i8(ped[4 * i + 1]), t2 = abo_as_i8(ped[4 * i + 2]);
    double s0[3];
    vecmat3(sv0, tab[t0], s0);
    const double* A = tab[t1 - t0];
    const double* B = tab[t2 - t0];
    double d_mm = cond_div(A + 6, B + 6);
    double d_um = cond_div(A + 3, B + 3);
    double d_uu = cond_div(A + 0, B + 0);
    dt1t2[i] = s0[0] * d_uu + s0[1] * d_um + s0[2] * d_mm;
  }
  if (p_uu_inf) *p_uu_inf = abo_p_uu_est(alpha, beta);
  return ABO_OK;
}

/* src/structs.rs:191-217 */
double abo_cost(const double* ped, int n, const double* dobs, double p_uu, double eqp,
                double eqp_weight, const double x[4], int lanes, int table) {
  double stackbuf[512];
  double* dt = (n <= 512) ? stackbuf : (double*)malloc(sizeof(double) * (size_t)n);
  double p_mm = 1.0 - p_uu; /* src/ab_neutral.rs:23, src/structs.rs:175 */
  double puu_inf;
  int rc = table ? abo_divergence_table(ped, n, p_mm, p_uu, x[0], x[1], x[2], dt, &puu_inf)
                 : abo_divergence(ped, n, p_mm, p_uu, x[0], x[1], x[2], dt, &puu_inf);
  double result;
  if (rc) {
    result = NAN;
  } else {
    /* :209-212  (ped - intercept - div)^2 + eqp_weight * nrows * (p_uu_inf - eqp)^2, per row */
    double pen = eqp_weight * (double)n * ((puu_inf - eqp) * (puu_inf - eqp));
    if (lanes <= 1) {
      double square_sum = 0.0;
      for (int i = 0; i < n; ++i) {
        double d = dobs ? dobs[i] : ped[4 * i + 3];
        double r = d - x[3] - dt[i];
        square_sum += r * r + pen;
      }
      result = square_sum;
    } else {
      /* lanes code: low byte = lanes of the butterfly; (code >> 8) + 1 = consecutive rows per lane and block
       * (1: lane l takes rows l, l+lanes, ...; 4 (the stream kernel): rows 4(l + lanes q) .. +3) */
      /* bit 16 (with 64 lanes): the canonical tree of every LDS-resident pedigree — 64 accumulators, combined from
       * the high lane bits down with the pairings the kernels' DPP / permlane steps realise:
       * l^32, l^16, l^8, then l <-> 7-l inside 8, l <-> 3-l inside 4, l^1 (abn_device.hpp: tree64_finish) */
      const int mirror = (lanes >> 16) & 1;
      const int vec = ((lanes >> 8) & 0xff) + 1;
      lanes &= 0xff;
      double part[64];
      for (int l = 0; l < lanes; ++l) {
        double acc = 0.0;
        for (long base = (long)vec * l; base < n; base += (long)vec * lanes)
          for (int e = 0; e < vec && base + e < n; ++e) {
            const long i = base + e;
            double d = dobs ? dobs[i] : ped[4 * i + 3];
            double r = d - x[3] - dt[i];
            acc += r * r + pen;
          }
        part[l] = acc;
      }
      if (mirror && lanes == 64) {
        for (int step = 0; step < 6; ++step) {
          double nxt[64];
          for (int l = 0; l < 64; ++l) {
            int p;
            switch (step) {
              case 0: p = l ^ 32; break;
              case 1: p = l ^ 16; break;
              case 2: p = l ^ 8; break;
              case 3: p = (l & ~7) | (7 - (l & 7)); break;
              case 4: p = (l & ~3) | (3 - (l & 3)); break;
              default: p = l ^ 1; break;
            }
            nxt[l] = part[l] + part[p];
          }
          memcpy(part, nxt, sizeof(double) * 64);
        }
      } else {
        for (int off = 1; off < lanes; off <<= 1) {
          double nxt[64];
          for (int l = 0; l < lanes; ++l) nxt[l] = part[l] + part[l ^ off];
          memcpy(part, nxt, sizeof(double) * (size_t)lanes);
        }
      }
      result = part[0];
    }
  }
  if (dt != stackbuf) free(dt);
  return result;
}

/* src/ab_neutral.rs:87-98 */
double abo_lse(const double* ped, int n, double p_uu, const double x[4]) {
  double stackbuf[512];
  double* dt = (n <= 512) ? stackbuf : (double*)malloc(sizeof(double) * (size_t)n);
  double s = 0.0;
  if (abo_divergence_table(ped, n, 1.0 - p_uu, p_uu, x[0], x[1], x[2], dt, NULL)) {
    s = NAN;
  } else {
    for (int i = 0; i < n; ++i) {
      double r = ped[4 * i + 3] - x[3] - dt[i];
      s += r * r;
    }
  }
  if (dt != stackbuf) free(dt);
  return s;
}

/* ------------------------------------------------------------------------------------------------
 * argmin 0.8.1 NelderMead (solver/neldermead/mod.rs) + Executor (core/executor.rs), restated from the
 * published crate; call sites src/ab_neutral.rs:49-64, src/boot_model.rs:69-84.
 *   coefficients alpha=1, gamma=2, rho=0.5, sigma=0.5, sd_tolerance=EPSILON (none overridden).
 * ---------------------------------------------------------------------------------------------- */
typedef struct {
  double x[4];
  double c;
} nm_vertex;

/* params.sort_by(|a,b| a.1.partial_cmp(&b.1).unwrap_or(Equal)) — std stable sort; for len<=20 it is
 * an insertion sort shifting left while is_less(tail, prev); NaN compares Equal (never less). */
static void nm_sort(nm_vertex v[5]) {
  for (int i = 1; i < 5; ++i) {
    if (v[i].c < v[i - 1].c) {
      nm_vertex tmp = v[i];
      int hole = i;
      do {
        v[hole] = v[hole - 1];
        --hole;
      } while (hole > 0 && tmp.c < v[hole - 1].c);
      v[hole] = tmp;
    }
  }
}

typedef struct {
  const double* ped;
  int n;
  const double* dobs;
  double p_uu, eqp, eqp_weight;
  int lanes, table;
  int evals;
} nm_problem;

static double nm_cost(nm_problem* p, const double x[4]) {
  p->evals++;
  return abo_cost(p->ped, p->n, p->dobs, p->p_uu, p->eqp, p->eqp_weight, x, p->lanes, p->table);
}

/* IterState::update(): accept when cost < best_cost, or both infinite with equal sign */
static void nm_update_best(const nm_vertex* v0, double* best_cost, double best[4], int* have_best) {
  double c = v0->c;
  if (c < *best_cost || (isinf(c) && isinf(*best_cost) && (signbit(c) == signbit(*best_cost)))) {
    memcpy(best, v0->x, sizeof(double) * 4);
    *best_cost = c;
    *have_best = 1;
  }
}

/* trace (nullable, trace_cap entries): the branch next_iter took in iteration i — 0 reflection accepted, 1 expansion
 * tried, 2 contraction accepted, 3 contraction rejected, 4 shrink.  Scheduling studies only (scripts/sched_sim.py). */
static void fit_impl(const double* ped, int n, const double* dobs, double p_uu, double eqp,
                     double eqp_weight, const double simplex0[20], int max_iters, double sd_tol,
                     int shrink_on_failed_contraction, int lanes, int table, abo_fit_result* out,
                     uint8_t* trace, int trace_cap) {
  nm_problem pb = {ped, n, dobs, p_uu, eqp, eqp_weight, lanes, table, 0};
  nm_vertex v[5];
  /* Solver::init: evaluate all vertices in the given order, sort */
  for (int k = 0; k < 5; ++k) {
    memcpy(v[k].x, simplex0 + 4 * k, sizeof(double) * 4);
    v[k].c = nm_cost(&pb, v[k].x);
  }
  nm_sort(v);
  double best[4] = {NAN, NAN, NAN, NAN};
  double best_cost = INFINITY;
  int have_best = 0;
  nm_update_best(&v[0], &best_cost, best, &have_best);

  int iter = 0;
  int status;
  for (;;) {
    /* terminate_internal: solver.terminate() -> max_iters -> target_cost (-inf) */
    double sum = 0.0;
    for (int k = 0; k < 5; ++k) sum += v[k].c;
    double c0 = sum / 5.0;
    double ss = 0.0;
    for (int k = 0; k < 5; ++k) ss += (v[k].c - c0) * (v[k].c - c0);
    double s = sqrt(1.0 / (5.0 - 1.0) * ss);
    if (s < sd_tol) {
      status = ABO_FIT_CONVERGED;
      break;
    }
    if (iter >= max_iters) {
      status = ABO_FIT_MAX_ITERS;
      break;
    }
    if (best_cost <= -INFINITY) {
      status = ABO_FIT_TARGET;
      break;
    }
    /* next_iter */
    double x0[4], xr[4];
    for (int d = 0; d < 4; ++d) {
      double a = v[0].x[d];
      a = a + v[1].x[d];
      a = a + v[2].x[d];
      a = a + v[3].x[d];
      x0[d] = a * (1.0 / 4.0);
    }
    for (int d = 0; d < 4; ++d) xr[d] = x0[d] + (x0[d] - v[4].x[d]) * 1.0;
    double fr = nm_cost(&pb, xr);
    int kind = 0;
    if (fr < v[3].c && fr >= v[0].c) {
      memcpy(v[4].x, xr, sizeof xr);
      v[4].c = fr;
    } else if (fr < v[0].c) {
      kind = 1;
      double xe[4];
      for (int d = 0; d < 4; ++d) xe[d] = x0[d] + (xr[d] - x0[d]) * 2.0;
      double fe = nm_cost(&pb, xe);
      if (fe < fr) {
        memcpy(v[4].x, xe, sizeof xe);
        v[4].c = fe;
      } else {
        memcpy(v[4].x, xr, sizeof xr);
        v[4].c = fr;
      }
    } else if (fr >= v[3].c) {
      double xc[4];
      for (int d = 0; d < 4; ++d) xc[d] = x0[d] + (v[4].x[d] - x0[d]) * 0.5;
      double fc = nm_cost(&pb, xc);
      kind = 3;
      if (fc < v[4].c) {
        kind = 2;
        memcpy(v[4].x, xc, sizeof xc);
        v[4].c = fc;
      } else if (shrink_on_failed_contraction) {
        kind = 4;
        for (int k = 1; k < 5; ++k) {
          for (int d = 0; d < 4; ++d) v[k].x[d] = v[0].x[d] + (v[k].x[d] - v[0].x[d]) * 0.5;
          v[k].c = nm_cost(&pb, v[k].x);
        }
      }
    } else { /* only reachable when fr is NaN */
      kind = 4;
      for (int k = 1; k < 5; ++k) {
        for (int d = 0; d < 4; ++d) v[k].x[d] = v[0].x[d] + (v[k].x[d] - v[0].x[d]) * 0.5;
        v[k].c = nm_cost(&pb, v[k].x);
      }
    }
    nm_sort(v);
    nm_update_best(&v[0], &best_cost, best, &have_best);
    if (trace && iter < trace_cap) trace[iter] = (uint8_t)kind;
    ++iter;
  }
  memcpy(out->best, best, sizeof best);
  out->best_cost = best_cost;
  out->iters = iter;
  out->evals = pb.evals;
  out->status = have_best ? status : ABO_FIT_NONFINITE;
  out->pad = 0;
}

void abo_fit(const double* ped, int n, const double* dobs, double p_uu, double eqp,
             double eqp_weight, const double simplex0[20], int max_iters, double sd_tol,
             int shrink_on_failed_contraction, int lanes, int table, abo_fit_result* out) {
  fit_impl(ped, n, dobs, p_uu, eqp, eqp_weight, simplex0, max_iters, sd_tol, shrink_on_failed_contraction, lanes, table,
           out, NULL, 0);
}

void abo_fit_batch(const double* ped, int n, const double* dobs_rows, int64_t f, double p_uu,
                   double eqp, double eqp_weight, const double* simplex0, int max_iters,
                   double sd_tol, int shrink_on_failed_contraction, int lanes, int table,
                   int threads, abo_fit_result* out) {
#ifdef _OPENMP
  if (threads <= 0) threads = omp_get_max_threads();
#pragma omp parallel for schedule(dynamic, 1) num_threads(threads)
#endif
  for (int64_t i = 0; i < f; ++i) {
    const double* d = dobs_rows ? dobs_rows + (size_t)i * (size_t)n : NULL;
    abo_fit(ped, n, d, p_uu, eqp, eqp_weight, simplex0 + 20 * i, max_iters, sd_tol,
            shrink_on_failed_contraction, lanes, table, out + i);
  }
}

/* src/ab_neutral.rs:83-135.  The reference stable-sorts a thread-order-dependent vector; here the
 * order is the start index, so ties resolve to the lowest index.  NaN LSE panics there; never wins here. */
int abo_select_best(const double* ped, int n, double p_uu, const double* models, int s,
                    double* lse_out, double* model_out, double* pred, double* resid) {
  int best = -1;
  double best_lse = INFINITY;
  for (int k = 0; k < s; ++k) {
    double l = abo_lse(ped, n, p_uu, models + 4 * k);
    if (lse_out) lse_out[k] = l;
    if (l == l && (best < 0 || l < best_lse)) {
      best = k;
      best_lse = l;
    }
  }
  if (best < 0) return -1;
  const double* m = models + 4 * best;
  if (model_out) memcpy(model_out, m, sizeof(double) * 4);
  double stackbuf[512];
  double* dt = (n <= 512) ? stackbuf : (double*)malloc(sizeof(double) * (size_t)n);
  abo_divergence_table(ped, n, 1.0 - p_uu, p_uu, m[0], m[1], m[2], dt, NULL);
  for (int i = 0; i < n; ++i) {
    double p = m[3] + dt[i];                    /* :123-129 */
    if (pred) pred[i] = p;
    if (resid) resid[i] = ped[4 * i + 3] - p;   /* :131-135 */
  }
  if (dt != stackbuf) free(dt);
  return best;
}

/* src/boot_model.rs:86-91 */
void abo_bootstrap_row(const double x[4], double row[7]) {
  row[0] = x[0];
  row[1] = x[1];
  row[2] = x[2];
  row[3] = x[3];
  row[4] = abo_est_mm(x[0], x[1]);
  row[5] = abo_est_um(x[0], x[1]);
  row[6] = abo_p_uu_est(x[0], x[1]);
}

/* ------------------------------------------------------------------------------------------------
 * Philox4x32-10 (Salmon, Moraes, Dror, Shaw: "Parallel random numbers: as easy as 1, 2, 3", SC'11).
 * ---------------------------------------------------------------------------------------------- */
void abo_philox4x32(uint32_t c0, uint32_t c1, uint32_t c2, uint32_t c3, uint32_t k0, uint32_t k1,
                    uint32_t out[4]) {
  for (int r = 0; r < 10; ++r) {
    uint64_t p0 = (uint64_t)0xD2511F53u * c0;
    uint64_t p1 = (uint64_t)0xCD9E8D57u * c2;
    uint32_t n0 = (uint32_t)(p1 >> 32) ^ c1 ^ k0;
    uint32_t n1 = (uint32_t)p1;
    uint32_t n2 = (uint32_t)(p0 >> 32) ^ c3 ^ k1;
    uint32_t n3 = (uint32_t)p0;
    c0 = n0; c1 = n1; c2 = n2; c3 = n3;
    k0 += 0x9E3779B9u;
    k1 += 0xBB67AE85u;
  }
  out[0] = c0; out[1] = c1; out[2] = c2; out[3] = c3;
}

/* 52 random mantissa bits -> [0,1)  (rand 0.8.5 UniformFloat: (u64 >> 12) with exponent 0, minus 1) */
static double u01_from(uint32_t lo, uint32_t hi) {
  uint64_t u = (((uint64_t)hi << 32) | lo) >> 12;
  uint64_t bits = 0x3FF0000000000000ull | u;
  double d;
  memcpy(&d, &bits, sizeof d);
  return d - 1.0;
}
static double uniform_from(uint32_t lo, uint32_t hi, double low, double high) {
  return u01_from(lo, hi) * (high - low) + low;
}

/* idx[i] = mulhi32(r, N): rows 4q..4q+3 come from Philox counter (q, boot, window, TAG_IDX) */
void abo_boot_indices(uint64_t seed, uint32_t window, uint32_t boot, int n, uint32_t* idx) {
  uint32_t k0 = (uint32_t)seed, k1 = (uint32_t)(seed >> 32);
  for (int q = 0; 4 * q < n; ++q) {
    uint32_t r[4];
    abo_philox4x32((uint32_t)q, boot, window, ABO_TAG_IDX, k0, k1, r);
    for (int j = 0; j < 4 && 4 * q + j < n; ++j)
      idx[4 * q + j] = (uint32_t)(((uint64_t)r[j] * (uint64_t)(uint32_t)n) >> 32);
  }
}

/* src/structs.rs:78-96, five times (src/ab_neutral.rs:49-55) */
void abo_start_simplex(uint64_t seed, uint32_t window, uint32_t start, double max_divergence,
                       double simplex[20]) {
  uint32_t k0 = (uint32_t)seed, k1 = (uint32_t)(seed >> 32);
  double mx = max_divergence;
  if (!(max_divergence > 0.0)) mx = 0.1; /* :80-83 */
  for (uint32_t v = 0; v < 5; ++v) {
    uint32_t r0[4], r1[4];
    abo_philox4x32(0u, start * 5u + v, window, ABO_TAG_START, k0, k1, r0);
    abo_philox4x32(1u, start * 5u + v, window, ABO_TAG_START, k0, k1, r1);
    simplex[4 * v + 0] = pow(10.0, uniform_from(r0[0], r0[1], -9.0, -2.0));
    simplex[4 * v + 1] = pow(10.0, uniform_from(r0[2], r0[3], -9.0, -2.0));
    simplex[4 * v + 2] = uniform_from(r1[0], r1[1], 0.0, 0.1);
    simplex[4 * v + 3] = uniform_from(r1[2], r1[3], 0.0, mx);
  }
}

/* src/structs.rs:100-128: U(n - 0.1|n|, n + 0.1|n|); n == 0 is treated as 0.1 */
static double vary_one(double n, uint32_t lo, uint32_t hi) {
  if (n == 0.0) n = 0.1;
  double low = n - fabs(n) * 0.1;
  double high = n + fabs(n) * 0.1;
  if (low >= high) {
    double t = low;
    low = high;
    high = t;
  }
  return uniform_from(lo, hi, low, high);
}

/* src/boot_model.rs:69-75 */
void abo_boot_simplex(uint64_t seed, uint32_t window, uint32_t boot, const double params[4],
                      double simplex[20]) {
  uint32_t k0 = (uint32_t)seed, k1 = (uint32_t)(seed >> 32);
  memcpy(simplex, params, sizeof(double) * 4);
  for (uint32_t v = 1; v < 5; ++v) {
    uint32_t r0[4], r1[4];
    abo_philox4x32((v - 1u) * 2u + 0u, boot, window, ABO_TAG_JITTER, k0, k1, r0);
    abo_philox4x32((v - 1u) * 2u + 1u, boot, window, ABO_TAG_JITTER, k0, k1, r1);
    simplex[4 * v + 0] = vary_one(params[0], r0[0], r0[1]);
    simplex[4 * v + 1] = vary_one(params[1], r0[2], r0[3]);
    simplex[4 * v + 2] = vary_one(params[2], r1[0], r1[1]);
    simplex[4 * v + 3] = vary_one(params[3], r1[2], r1[3]);
  }
}

/* src/boot_model.rs:41-100 */
static void boot_model_impl(const double* ped, int n, const double model[4], const double* pred,
                    const double* resid, double p_uu, double eqp, double eqp_weight, uint64_t seed,
                    uint32_t window, uint32_t b0, int64_t nb, int max_iters, double sd_tol,
                    int shrink_on_failed_contraction, int lanes, int table, int threads,
                    double* raw, abo_fit_result* results, uint8_t* traces, int trace_cap) {
#ifdef _OPENMP
  if (threads <= 0) threads = omp_get_max_threads();
#pragma omp parallel num_threads(threads)
#endif
  {
    uint32_t* idx = (uint32_t*)malloc(sizeof(uint32_t) * (size_t)n);
    double* dstar = (double*)malloc(sizeof(double) * (size_t)n);
#ifdef _OPENMP
#pragma omp for schedule(dynamic, 1)
#endif
    for (int64_t i = 0; i < nb; ++i) {
      uint32_t b = b0 + (uint32_t)i;
      abo_boot_indices(seed, window, b, n, idx);
      for (int j = 0; j < n; ++j) dstar[j] = pred[j] + resid[idx[j]]; /* :50-54 */
      double simplex[20];
      abo_boot_simplex(seed, window, b, model, simplex);
      abo_fit_result r;
      fit_impl(ped, n, dstar, p_uu, eqp, eqp_weight, simplex, max_iters, sd_tol,
               shrink_on_failed_contraction, lanes, table, &r, traces ? traces + (size_t)i * (size_t)trace_cap : NULL,
               trace_cap);
      abo_bootstrap_row(r.best, raw + 7 * i);
      if (results) results[i] = r;
    }
    free(idx);
    free(dstar);
  }
}

void abo_boot_model(const double* ped, int n, const double model[4], const double* pred,
                    const double* resid, double p_uu, double eqp, double eqp_weight, uint64_t seed,
                    uint32_t window, uint32_t b0, int64_t nb, int max_iters, double sd_tol,
                    int shrink_on_failed_contraction, int lanes, int table, int threads,
                    double* raw, abo_fit_result* results) {
  boot_model_impl(ped, n, model, pred, resid, p_uu, eqp, eqp_weight, seed, window, b0, nb, max_iters, sd_tol,
                  shrink_on_failed_contraction, lanes, table, threads, raw, results, NULL, 0);
}
/* the same, recording every fit's branch per iteration (traces[nb][trace_cap]; see fit_impl) */
void abo_boot_model_trace(const double* ped, int n, const double model[4], const double* pred,
                          const double* resid, double p_uu, double eqp, double eqp_weight, uint64_t seed,
                          uint32_t window, uint32_t b0, int64_t nb, int max_iters, double sd_tol,
                          int shrink_on_failed_contraction, int lanes, int table, int threads,
                          double* raw, abo_fit_result* results, uint8_t* traces, int trace_cap) {
  boot_model_impl(ped, n, model, pred, resid, p_uu, eqp, eqp_weight, seed, window, b0, nb, max_iters, sd_tol,
                  shrink_on_failed_contraction, lanes, table, threads, raw, results, traces, trace_cap);
}

/* ------------------------------------------------------------------------------------------------
 * src/analysis.rs:50-98 (ndarray 0.15.6 mean/std, ndarray-stats 0.5.1 Linear quantiles; unpinned)
 * ---------------------------------------------------------------------------------------------- */
static int cmp_double(const void* a, const void* b) {
  double x = *(const double*)a, y = *(const double*)b;
  return (x > y) - (x < y);
}
/* ndarray numeric_util::unrolled_fold over a contiguous slice */
static double unrolled_sum(const double* xs, int64_t len) {
  double acc = 0.0, p[8] = {0, 0, 0, 0, 0, 0, 0, 0};
  while (len >= 8) {
    for (int j = 0; j < 8; ++j) p[j] = p[j] + xs[j];
    xs += 8;
    len -= 8;
  }
  acc = acc + (p[0] + p[4]);
  acc = acc + (p[1] + p[5]);
  acc = acc + (p[2] + p[6]);
  acc = acc + (p[3] + p[7]);
  for (int64_t i = 0; i < len && i < 7; ++i) acc = acc + xs[i];
  return acc;
}
/* ndarray var(ddof): Welford with mul_add */
static double welford_std(const double* x, int64_t n, int64_t stride) {
  double mean = 0.0, sum_sq = 0.0;
  for (int64_t i = 0; i < n; ++i) {
    double v = x[i * stride];
    double count = (double)(i + 1);
    double delta = v - mean;
    mean = mean + delta / count;
    sum_sq = fma3(v - mean, delta, sum_sq);
  }
  return sqrt(sum_sq / ((double)n - 1.0));
}
static void linear_ci(double* sorted, int64_t n, double* lo, double* hi) {
  const double qs[2] = {0.025, 0.975};
  double res[2];
  for (int k = 0; k < 2; ++k) {
    double fi = qs[k] * (double)(n - 1);
    int64_t l = (int64_t)floor(fi), h = (int64_t)ceil(fi);
    double frac = fi - trunc(fi);
    res[k] = sorted[l] + frac * (sorted[h] - sorted[l]);
  }
  *lo = res[0];
  *hi = res[1];
}

void abo_analyze(const double* raw, int64_t b, double out[32]) {
  static const int col_of[8] = {0, 1, -1, 2, 3, 4, 5, 6}; /* -1 = beta/alpha (:54) */
  double* tmp = (double*)malloc(sizeof(double) * (size_t)b);
  double* ratio = (double*)malloc(sizeof(double) * (size_t)b);
  for (int64_t i = 0; i < b; ++i) ratio[i] = raw[7 * i + 1] / raw[7 * i + 0];
  for (int k = 0; k < 8; ++k) {
    int c = col_of[k];
    double mean, sd;
    if (c < 0) {
      mean = unrolled_sum(ratio, b) / (double)b;
      sd = welford_std(ratio, b, 1);
      memcpy(tmp, ratio, sizeof(double) * (size_t)b);
    } else {
      double s = 0.0;
      for (int64_t i = 0; i < b; ++i) s = s + raw[7 * i + c];
      mean = s / (double)b;
      sd = welford_std(raw + c, b, 7);
      for (int64_t i = 0; i < b; ++i) tmp[i] = raw[7 * i + c];
    }
    qsort(tmp, (size_t)b, sizeof(double), cmp_double);
    out[k] = mean;
    out[8 + k] = sd;
    linear_ci(tmp, b, &out[16 + k], &out[24 + k]);
  }
  free(tmp);
  free(ratio);
}

/* ------------------------------------------------------------------------------------------------
 * DMatrix::from, src/pedigree.rs:210-261: for every pair i < j (nested-loop order) the sum of
 * |status_i - status_j| over sites whose posteriormax passes the filter in BOTH samples (:249-253), the
 * number of such sites (:254) and D = divergence / (2 * compared_sites) (:257).
 * status[n*L] in {0,1,2} (status_numeric), posteriormax[n*L] doubles.
 * ---------------------------------------------------------------------------------------------- */
void abo_pairwise_divergence(const uint8_t* status, const double* posteriormax, int n, int64_t n_sites,
                             double posterior_max_filter, uint64_t* diff, uint64_t* both, double* dvalue) {
  int64_t p = 0;
  for (int i = 0; i < n; ++i)
    for (int j = i + 1; j < n; ++j, ++p) {
      uint64_t divergence = 0, compared = 0;
      for (int64_t k = 0; k < n_sites; ++k) {
        const double fp = posteriormax[(int64_t)i * n_sites + k], sp = posteriormax[(int64_t)j * n_sites + k];
        if (fp < posterior_max_filter || sp < posterior_max_filter) continue;
        const int a = status[(int64_t)i * n_sites + k], b = status[(int64_t)j * n_sites + k];
        divergence += (uint64_t)(a > b ? a - b : b - a);
        compared += 1;
      }
      if (diff) diff[p] = divergence;
      if (both) both[p] = compared;
      if (dvalue) dvalue[p] = (double)divergence / (2.0 * (double)compared);
    }
}

int abo_max_threads(void) {
#ifdef _OPENMP
  return omp_get_max_threads();
#else
  return 1;
#endif
}
