/*
 * abn_oracle.h — CPU ORACLE for the ABneutral hot path.  TEST INFRASTRUCTURE ONLY.
 *
 * This is a plain-C restatement of the reference's arithmetic (alphabeta-rs v0.2.1).  It is the
 * checker the HIP path is compared against; it is never the thing shipped or measured as the product.
 * Only tests/, __graft_entry__.smoke() and bench.py's cpu_baseline leg may load it.
 *
 * Pinning (see oracle/README.md and tests/test_oracle_golden.py):
 *   - divergence()      pinned by data/pedigree.txt + data/divergence.txt   (src/divergence.rs:138-161)
 *   - cost()            pinned bit-exactly by 0.0006700888539608879         (src/structs.rs:225-240)
 *   - matrix_power      pinned by the identity / row-extraction tests       (src/divergence.rs:129-209)
 *   - Nelder-Mead       argmin 0.8.1 (Cargo.lock:121-122) is NOT under /root/reference and the
 *                       reference holds no enabled test at that boundary: OPTIMIZER TRAJECTORY PARITY
 *                       IS UNPINNED.  The restatement follows the published argmin 0.8.1 algorithm.
 *   - RNG               rand 0.8.5 thread_rng is unseeded: parity unpinned by design (distributional).
 */
#ifndef ABN_ORACLE_H
#define ABN_ORACLE_H

#include <stdint.h>

#ifdef __cplusplus
extern "C" {
#endif

/* status codes shared with the product header (include/abneutral.h) */
#define ABO_OK 0
#define ABO_ERR_INVALID_ARG 1
#define ABO_ERR_BAD_PEDIGREE 2

/* fit status (per Nelder-Mead run) */
#define ABO_FIT_CONVERGED 0   /* sd of simplex costs < sd_tol  (argmin SolverConverged)      */
#define ABO_FIT_MAX_ITERS 1   /* iter >= max_iters            (argmin MaxItersReached)      */
#define ABO_FIT_NONFINITE 2   /* no finite best parameter vector was ever recorded           */
#define ABO_FIT_TARGET 3      /* best cost <= -inf            (argmin TargetCostReached)    */

/* Philox stream tags (must match alphabeta_rs_amd/csrc/abn_philox.h; restated, not shared) */
#define ABO_TAG_START 1u
#define ABO_TAG_JITTER 2u
#define ABO_TAG_IDX 3u

typedef struct {
  double best[4];     /* argmin state.best_param                                            */
  double best_cost;   /* argmin state.best_cost                                             */
  int32_t iters;      /* number of next_iter() calls                                        */
  int32_t evals;      /* number of cost() calls                                             */
  int32_t status;     /* ABO_FIT_*                                                          */
  int32_t pad;
} abo_fit_result;

/* src/divergence.rs:96-114 */
void abo_genmatrix(double alpha, double beta, double G[9]);
/* src/divergence.rs:16-31 (power >= 0; negative powers -> returns ABO_ERR_BAD_PEDIGREE) */
int abo_matrix_power(const double M[9], int power, double out[9]);
/* src/alphabeta.rs:62-65 */
double abo_p_uu_est(double alpha, double beta);
/* src/structs.rs:146-159 */
double abo_est_mm(double alpha, double beta);
double abo_est_um(double alpha, double beta);
/* src/alphabeta.rs:73-79 */
double abo_steady_state(double alpha, double beta);

/* f64 -> i8 `as` cast of Rust (truncate toward zero, saturate, NaN -> 0), src/divergence.rs:52 */
int abo_as_i8(double x);
/* validates that every row has integer-castable 0 <= t0 <= min(t1,t2) <= 127 */
int abo_check_pedigree(const double* ped, int n);

/* src/divergence.rs:33-94, reference-shaped: three matrix_power calls per pedigree row.
 * ped is N x 4 row-major (t0,t1,t2,D).  dt1t2[N] and *p_uu_inf are written. */
int abo_divergence(const double* ped, int n, double p_mm, double p_uu, double alpha, double beta,
                   double weight, double* dt1t2, double* p_uu_inf);
/* Same results bit for bit, but G^0..G^T computed once (the table the HIP kernel uses). */
int abo_divergence_table(const double* ped, int n, double p_mm, double p_uu, double alpha,
                         double beta, double weight, double* dt1t2, double* p_uu_inf);

/* src/structs.rs:191-217.  dobs == NULL -> column 3 of ped.  lanes == 1 reproduces the reference's
 * serial row-order accumulation; lanes in {2,4,...,64} reproduces the HIP kernel's reduction: lane l
 * accumulates rows l, l+lanes, ... in increasing order, then an xor-butterfly (offsets 1,2,4,...).
 * lanes may carry a row-block code in its upper bits: (lanes >> 8) + 1 consecutive rows per lane and block
 * (the stream-mode kernel uses 4; abn_fit_info.lanes reports the same code, so tests pass it through).
 * table != 0 uses abo_divergence_table (same bits, faster). */
double abo_cost(const double* ped, int n, const double* dobs, double p_uu, double eqp,
                double eqp_weight, const double x[4], int lanes, int table);
/* pure least-squares error of src/ab_neutral.rs:83-101 (serial row order, no penalty) */
double abo_lse(const double* ped, int n, double p_uu, const double x[4]);

/* argmin 0.8.1 NelderMead + Executor, as called at src/ab_neutral.rs:49-64 / src/boot_model.rs:69-84.
 * simplex0 = 5 vertices x 4.  shrink_on_failed_contraction = 0 restates argmin 0.8.1 (a rejected
 * contraction leaves the simplex unchanged); 1 is the textbook variant. */
void abo_fit(const double* ped, int n, const double* dobs, double p_uu, double eqp,
             double eqp_weight, const double simplex0[20], int max_iters, double sd_tol,
             int shrink_on_failed_contraction, int lanes, int table, abo_fit_result* out);

/* Batch of fits over one pedigree, OpenMP over fits (mirrors the rayon par_iter).
 * dobs_rows: NULL (all fits use ped col 3) or F x N row-major observed divergences. */
void abo_fit_batch(const double* ped, int n, const double* dobs_rows, int64_t f, double p_uu,
                   double eqp, double eqp_weight, const double* simplex0, int max_iters,
                   double sd_tol, int shrink_on_failed_contraction, int lanes, int table,
                   int threads, abo_fit_result* out);

/* src/ab_neutral.rs:83-135: stable arg-min by pure LSE over S fitted models (NaN LSE never wins),
 * then predicted divergence and residuals.  Returns the index of the winner or -1. */
int abo_select_best(const double* ped, int n, double p_uu, const double* models, int s,
                    double* lse_out, double* model_out, double* pred, double* resid);

/* src/boot_model.rs:86-91: [alpha,beta,weight,intercept,est_mm,est_um,est_uu] */
void abo_bootstrap_row(const double x[4], double row[7]);

/* ---- deterministic inputs (restatement of Philox4x32-10, Salmon et al. SC'11) ---- */
void abo_philox4x32(uint32_t c0, uint32_t c1, uint32_t c2, uint32_t c3, uint32_t k0, uint32_t k1,
                    uint32_t out[4]);
/* bootstrap indices of boot b (global index) in window w: idx[N], each in [0,N) */
void abo_boot_indices(uint64_t seed, uint32_t window, uint32_t boot, int n, uint32_t* idx);
/* start simplex of start s in window w (Model::new x5, src/structs.rs:78-96) */
void abo_start_simplex(uint64_t seed, uint32_t window, uint32_t start, double max_divergence,
                       double simplex[20]);
/* bootstrap simplex [params, vary() x4] of boot b in window w (src/structs.rs:100-128) */
void abo_boot_simplex(uint64_t seed, uint32_t window, uint32_t boot, const double params[4],
                      double simplex[20]);

/* src/boot_model.rs:41-100 for boots [b0, b0+nb) of window w: resample, refit, rows.
 * raw is nb x 7; results (optional, may be NULL) nb entries. */
void abo_boot_model(const double* ped, int n, const double model[4], const double* pred,
                    const double* resid, double p_uu, double eqp, double eqp_weight, uint64_t seed,
                    uint32_t window, uint32_t b0, int64_t nb, int max_iters, double sd_tol,
                    int shrink_on_failed_contraction, int lanes, int table, int threads,
                    double* raw, abo_fit_result* results);

/* abo_boot_model that also records the branch every Nelder-Mead iteration took (traces[nb][trace_cap] bytes: 0 reflection
 * accepted, 1 expansion tried, 2 contraction accepted, 3 contraction rejected, 4 shrink): scheduling studies only */
void abo_boot_model_trace(const double* ped, int n, const double model[4], const double* pred,
                          const double* resid, double p_uu, double eqp, double eqp_weight, uint64_t seed,
                          uint32_t window, uint32_t b0, int64_t nb, int max_iters, double sd_tol,
                          int shrink_on_failed_contraction, int lanes, int table, int threads,
                          double* raw, abo_fit_result* results, uint8_t* traces, int trace_cap);

/* src/analysis.rs:50-98: out[24] = 8 means, 8 sample SDs, 8 x (lo,hi)... laid out as
 * mean[8] (alpha,beta,beta/alpha,weight,intercept,pr_mm,pr_um,pr_uu), sd[8], ci_lo[8], ci_hi[8] -> 32 */
void abo_analyze(const double* raw, int64_t b, double out[32]);

/* DMatrix::from, src/pedigree.rs:210-261 (pairs in nested-loop order i < j) */
void abo_pairwise_divergence(const uint8_t* status, const double* posteriormax, int n, int64_t n_sites,
                             double posterior_max_filter, uint64_t* diff, uint64_t* both, double* dvalue);

int abo_max_threads(void);

#ifdef __cplusplus
}
#endif
#endif
