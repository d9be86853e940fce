/*
 * abneutral.h — C-ABI of the MI355X-native ABneutral hot path (libabneutral_hip.so).
 *
 * Drop-in boundary for alphabeta-rs v0.2.1 (citations relative to the reference tree).  The reference
 * has no FFI layer; the seams this library sits behind are Rust-level:
 *   (1) `impl CostFunction for Problem`          src/structs.rs:191-217      -> abn_cost_batch
 *   (2) `ab_neutral::run`                        src/ab_neutral.rs:13-20     -> abn_ab_neutral_run
 *   (3) `boot_model::run`                        src/boot_model.rs:17-28     -> abn_boot_model_run
 *   (4) the serial window loop of `metaprofile`  src/cli/metaprofile.rs:50-72 -> abn_plan_* (batched)
 * INTEGRATION.md shows the `extern "C"` block + safe wrappers a maintainer would add on the Rust side.
 *
 * Conventions: plain pointers and sizes, all f64 unless noted, row-major, caller owns every buffer it
 * passes; the library owns device memory behind the opaque handles.  No exceptions cross the ABI and
 * nothing aborts: every entry point returns an abn_status (0 = ok).  A handle is thread-compatible
 * (one thread at a time); there is no hidden global state.
 *
 * There is NO CPU fallback: without a HIP device every compute entry point returns ABN_ERR_NO_DEVICE.
 */
#ifndef ABNEUTRAL_H
#define ABNEUTRAL_H

#include <stdint.h>

#ifdef __cplusplus
extern "C" {
#endif

#define ABN_VERSION_MAJOR 0
#define ABN_VERSION_MINOR 1

typedef enum abn_status {
  ABN_OK = 0,
  ABN_ERR_INVALID_ARG = 1,   /* null pointer, non-positive size, unsupported option value; also a pedigree
                                whose power table and distinct-triple list (10 (T+1) + K + 4 doubles per chain) do
                                not fit the 160 KiB of LDS a workgroup can have: K up to ~17 000 distinct triples
                                at T = 255 (64 KiB per workgroup with an explicit lanes_per_chain below 64)          */
  ABN_ERR_BAD_PEDIGREE = 2,  /* a generation outside 0..127 after the `as i8` cast, or t1/t2 < t0      */
  ABN_ERR_NO_DEVICE = 3,     /* no usable HIP device (the product path never falls back to the CPU)    */
  ABN_ERR_HIP = 4,           /* a HIP runtime call failed; see abn_last_error()                        */
  ABN_ERR_NO_FINITE_FIT = 5, /* every start of a window ended non-finite (reference: panic, :28,:100)  */
  ABN_ERR_STATE = 6          /* plan used out of order (e.g. run before set_windows)                   */
} abn_status;

/* per-fit termination status, written in fit order (no push-under-lock as in src/ab_neutral.rs:74) */
#define ABN_FIT_CONVERGED 0  /* SD of the 5 simplex costs < sd_tolerance (argmin SolverConverged)     */
#define ABN_FIT_MAX_ITERS 1  /* iteration budget reached (argmin MaxItersReached)                     */
#define ABN_FIT_NONFINITE 2  /* no finite best parameter vector (reference: `.unwrap()` panic)        */
#define ABN_FIT_TARGET 3     /* best cost <= -inf (argmin TargetCostReached with default target)      */

typedef struct abn_ctx abn_ctx;
typedef struct abn_plan abn_plan;

typedef struct abn_options {
  uint64_t seed;                        /* Philox4x32-10 key (reference: unseeded thread_rng)           */
  int32_t lanes_per_chain;              /* 0 = auto; 8, 16, 32 or 64: accumulators of the residual reduction
                                           tree = lanes per chain of the packed kernels.  Auto picks it from the
                                           PEDIGREE alone (rows, generations, distinct triples) — never from the
                                           number of fits in the launch — so results are independent of batch size
                                           and of how a job is sharded over GPUs; few-chain launches use one (or
                                           four) wavefronts per chain and reproduce the same tree bit for bit     */
  int32_t strict_order;                 /* 1 = every cost sums its residuals serially in row order, the reference's
                                           `square_sum += ...` (src/structs.rs:206-213): cost AND fit entry points,
                                           bit-equal to the oracle's lanes = 1; abn_fit_info.lanes reports 1.
                                           0 = auto: serial for pedigrees of up to 16 rows (free there; the bundled
                                           data/ pedigree is then in the reference's order by default), else the
                                           pedigree's reduction tree.  -1 = the tree whatever the size.  Like the
                                           tree a function of the pedigree and the options, never of the launch    */
  int32_t shrink_on_failed_contraction; /* 0 = argmin 0.8.1 behaviour; 1 = textbook Nelder-Mead        */
  int32_t max_iters_start;              /* 10000, src/ab_neutral.rs:62                                 */
  int32_t max_iters_boot;               /* 1000,  src/boot_model.rs:81                                 */
  int32_t stream_mode;                  /* pedigrees too large for LDS: 0 = stream the bootstrap observations
                                           (gathered once per fit, 8 B/row/evaluation); 1 = re-stream the u32
                                           index row and gather residuals every evaluation (4 B/row + gather) */
  double sd_tolerance;                  /* f64::EPSILON (argmin default, never overridden)             */
  int32_t window_groups;                /* abn_plan_run: windows are cut into this many groups that run
                                           A -> select -> B concurrently on separate HIP streams
                                           (0 or 1 = one stream, the default; at most 4 overlap: HIP maps
                                           streams onto 4 in-order hardware queues.  Measured gain on a
                                           25-window shard: 2 %, the slowest start chain sets the time)                    */
  int32_t no_fixed_point_skip;          /* 0 (default): a fit whose contraction is rejected has reached a fixed
                                           point of argmin 0.8.1's iteration (the simplex is left untouched and
                                           the cost is deterministic), so it is finished at once with the counters
                                           the repetitions would have produced (iters = max_iters, evals += 2 per
                                           remaining iteration, ABN_FIT_MAX_ITERS): identical outputs.  1 = execute
                                           the repetitions like the reference does                             */
} abn_options;

typedef struct abn_fit_info {
  double best_cost; /* argmin state.best_cost (includes the equilibrium penalty)                        */
  int32_t iters;    /* next_iter() calls                                                                */
  int32_t evals;    /* cost() calls                                                                     */
  int32_t status;   /* ABN_FIT_*                                                                        */
  int32_t lanes;    /* residual reduction tree (accumulators; | rows-per-block code << 8 in stream mode) */
} abn_fit_info;

/* ------------------------------------------------------------------ context */
void abn_default_options(abn_options* opts);
int abn_device_count(int* count);
/* stream == NULL: the context creates (and owns) its own non-blocking HIP stream; ABN_STREAM_DEFAULT: the
 * device's default (null) stream — what torch.cuda.current_stream() is unless the caller switched streams;
 * otherwise the given hipStream_t is borrowed (e.g. torch.cuda.current_stream().cuda_stream != 0). */
#define ABN_STREAM_DEFAULT ((void*)(intptr_t)-1)
int abn_init(int device_ordinal, void* stream, abn_ctx** ctx);
/* Frees the context and its device-buffer pool (the per-call entry points recycle their device buffers through it
 * instead of hipMalloc / hipFree per call).  Destroy the context's plans first. */
int abn_shutdown(abn_ctx* ctx);
/* What abn_init read from hipDeviceProp and the launch geometry derived from it: out4 = {compute units, KiB of LDS per
 * CU, wavefronts of a persistent fit launch, wavefronts of a persistent launch that just about fills the GPU}.  MI355X:
 * {256, 160, 3072, 2048}.  abn_init refuses (ABN_ERR_NO_DEVICE) a device that is not gfx950 or has less LDS per CU.
 * (The reference has no counterpart: rayon sizes its pool from the host's cores, src/ab_neutral.rs:37, src/boot_model.rs:41.) */
int abn_device_info(const abn_ctx* ctx, int32_t* out4);
const char* abn_last_error(const abn_ctx* ctx);
const char* abn_status_string(int status);
int abn_version(void);

/* The residual reduction tree the fits of this pedigree use (what abn_fit_info.lanes reports): a function of the
 * pedigree's generations (N x 3) and opts->lanes_per_chain only — host arithmetic, no device needed. */
int abn_reduction_tree(const abn_options* opts, const double* generations, int32_t n_rows, int32_t* tree);

/* ------------------------------------------------------------------ (1) cost function
 * Replaces `Problem::cost` (src/structs.rs:194-216) + `divergence()` (src/divergence.rs:33-94) for M
 * candidates at once.  pedigree: N x 4 rows (t0,t1,t2,D) exactly as `Pedigree` (src/pedigree.rs:44-45).
 * Observed divergences: column 3 of the pedigree, or, when idx != NULL, the residual bootstrap
 * D*_i = pred[i] + resid[idx[cand_to_boot[m]*N + i]] (src/boot_model.rs:50-57).
 * Outputs: cost[M]; optional dt1t2[M x N] and p_uu_inf[M] (struct Divergence, src/divergence.rs:10-14). */
int abn_cost_batch(abn_ctx* ctx, const abn_options* opts, const double* pedigree, int32_t n_rows,
                   double p_uu0, double eqp, double eqp_weight, const double* candidates, int64_t m,
                   const double* pred, const double* resid, const uint32_t* idx,
                   const uint32_t* cand_to_boot, int64_t n_boot_rows, double* cost, double* dt1t2,
                   double* p_uu_inf);

/* ------------------------------------------------------------------ Nelder-Mead fits
 * Replaces `Executor::new(problem, NelderMead::new(simplex)).max_iters(k).run()` (src/ab_neutral.rs:49-64,
 * src/boot_model.rs:69-84) for F independent fits, one per simplex0[f] (5 vertices x 4).
 * dobs_rows: NULL (every fit uses pedigree column 3) or F x N observed divergences.
 * Outputs in fit order: best[F x 4] (argmin best_param), info[F]. */
int abn_fit_batch(abn_ctx* ctx, const abn_options* opts, const double* pedigree, int32_t n_rows,
                  double p_uu0, double eqp, double eqp_weight, const double* simplex0, int64_t f,
                  const double* dobs_rows, int32_t max_iters, double* best, abn_fit_info* info);

/* ------------------------------------------------------------------ deterministic inputs
 * Model::new x5 per start (src/structs.rs:78-96): simplex0[S x 5 x 4] for `window`. Host arithmetic. */
int abn_gen_start_simplices(uint64_t seed, uint32_t window, int32_t n_starts, double max_divergence,
                            double* simplex0);
/* [params, vary() x4] (src/boot_model.rs:69-75, src/structs.rs:100-128) for boots [b0, b0+nb) */
int abn_gen_boot_simplices(uint64_t seed, uint32_t window, uint32_t b0, int64_t nb,
                           const double params[4], double* simplex0);
/* residual-bootstrap indices (src/boot_model.rs:43-48) idx[nb x N], generated ON DEVICE, copied back */
int abn_gen_boot_indices(abn_ctx* ctx, uint64_t seed, uint32_t window, uint32_t b0, int64_t nb,
                         int32_t n_rows, uint32_t* idx);

/* ------------------------------------------------------------------ (2) ab_neutral::run
 * src/ab_neutral.rs:13-142: n_starts random-start fits (start simplices from opts->seed), arg-min by
 * pure LSE (:83-101), predicted divergence (:123-129) and residuals (:131-135).
 * Outputs: model[4], pred[N], resid[N]; optional all_models[S x 4], info[S], lse[S]. */
int abn_ab_neutral_run(abn_ctx* ctx, const abn_options* opts, const double* pedigree, int32_t n_rows,
                       double p0uu, double eqp, double eqp_weight, int32_t n_starts, double* model,
                       double* pred, double* resid, double* all_models, abn_fit_info* info,
                       double* lse);

/* src/ab_neutral.rs:83-135 on its own: stable arg-min by pure LSE (serial row order, no penalty term) over
 * S fitted models, then predicted divergence and residuals of the winner.  NaN never wins; *best_index = -1
 * (and ABN_ERR_NO_FINITE_FIT) if every LSE is NaN.  lse[S] optional. */
int abn_select_best(abn_ctx* ctx, const double* pedigree, int32_t n_rows, double p0uu, const double* models,
                    int32_t n_models, int32_t* best_index, double* model, double* pred, double* resid,
                    double* lse);

/* src/boot_model.rs:86-91 on its own: raw[B x 7] = [alpha, beta, weight, intercept, est_mm, est_um, est_uu]
 * (src/structs.rs:146-159) from fitted vectors best[B x 4], evaluated on the device. */
int abn_bootstrap_rows(abn_ctx* ctx, const double* best, int64_t n_boot, double* raw);

/* ------------------------------------------------------------------ (3) boot_model::run
 * src/boot_model.rs:17-115 without the PNG (:105-109): n_boot residual-bootstrap refits.
 * raw[n_boot x 7] = [alpha,beta,weight,intercept,PrMM,PrUM,PrUU] rows (RawAnalysis, src/analysis.rs:12),
 * row b = bootstrap b (the reference's row order is thread-schedule dependent).  info optional. */
int abn_boot_model_run(abn_ctx* ctx, const abn_options* opts, const double* pedigree, int32_t n_rows,
                       const double* model, const double* pred, const double* resid, double p0uu,
                       double eqp, double eqp_weight, int32_t n_boot, double* raw,
                       abn_fit_info* info);

/* src/analysis.rs:50-98 on the host: out[32] = mean[8], sd[8], ci_lo[8], ci_hi[8] in the order
 * alpha, beta, beta/alpha, weight, intercept, pr_mm, pr_um, pr_uu (struct Analysis, :15-47).
 * Precondition: no NaN in raw nor in beta/alpha (rows of fits with status ABN_FIT_NONFINITE): the quantiles sort with `<`
 * (the reference converts to n64, which rejects NaN, :57-58).  The host mirrors (RawAnalysis::analyze, the Python
 * analyze()) check it and refuse such a table with ABN_ERR_NO_FINITE_FIT. */
int abn_analyze(const double* raw, int64_t n_boot, double* out32);

/* ------------------------------------------------------------------ pedigree construction
 * DMatrix::from (src/pedigree.rs:210-261): pairwise divergence of n samples over n_sites aligned sites.
 *   codes[n x n_sites] (u8, row per sample): status_numeric 0 = U, 1 = I, 2 = M
 *   (src/methylation_site.rs:130-136), plus 0x80 when the site's posteriormax is below the filter (the
 *   site is then skipped for every pair it takes part in, src/pedigree.rs:249-251).
 * Outputs per unordered pair i < j at index p = i*n - i*(i+1)/2 + (j - i - 1) (the nested-loop order of
 * :214-215):  diff[p] = sum of |status_i - status_j| over sites valid in both (:253),  both[p] = number of
 * such sites (:254),  dvalue[p] = diff / (2 * both) in f64 (:257; NaN for both == 0 like the reference's
 * 0/0).  Integer sums are exact, so the result does not depend on the device's summation order.
 * Any number of samples up to 65535 (the reference has no limit; the sample axis is tiled in groups of 64), any
 * number of sites, codes at any byte alignment.  Bytes other than 0, 1, 2 (| 0x80) are not valid codes. */
int abn_pairwise_divergence(abn_ctx* ctx, const uint8_t* codes, int32_t n_samples, int64_t n_sites,
                            uint64_t* diff, uint64_t* both, double* dvalue);
/* The same on DEVICE-resident buffers (no PCIe in the call): dev_codes u8[n x n_sites]; dev_diff / dev_both
 * u64[pairs], dev_dvalue f64[pairs], any of the three may be NULL.  kernel_ms (nullable): HIP-event time of the
 * kernels of this call on the context's stream.  Returns after the work has completed. */
int abn_pairwise_divergence_dev(abn_ctx* ctx, const void* dev_codes, int32_t n_samples, int64_t n_sites,
                                void* dev_diff, void* dev_both, void* dev_dvalue, double* kernel_ms);

/* ------------------------------------------------------------------ (4) batched, device-resident plan
 * One pedigree topology (t0,t1,t2 of N rows), W windows that differ in D / p0uu (the metaprofile loop,
 * src/cli/metaprofile.rs:50-72, where every window shares nodelist/edgelist), S starts and B bootstraps
 * per window.  boot_offset / window_offset place this plan's shard in the global (window, bootstrap)
 * index space so that results do not depend on how the job is sharded over GPUs. */
int abn_plan_create(abn_ctx* ctx, const abn_options* opts, const double* generations /* N x 3 */,
                    int32_t n_rows, int32_t n_windows, int32_t n_starts, int32_t n_boot,
                    uint32_t window_offset, uint32_t boot_offset, abn_plan** plan);
int abn_plan_destroy(abn_plan* plan);
/* Optional, before abn_plan_set_windows: ids[W] = each window's index in the Philox counters (start simplices,
 * jitter, bootstrap indices).  Default: window_offset + w.  The metaprofile driver passes every window's own
 * position in the (region, window) enumeration (src/cli/metaprofile.rs:50-53), so that skipped windows and
 * several topology groups never make two windows share a random stream.  NULL restores the default. */
int abn_plan_set_window_ids(abn_plan* plan, const uint32_t* ids);
/* D[W x N], p0uu[W]; eqp = p0uu and eqp_weight = 1 as src/alphabeta.rs:33-54 unless eqp/eqp_weight
 * are non-NULL ([W] each).  Also draws the start simplices and generates the bootstrap index buffer
 * idx[W x B x N] (u32) in HBM on the device. */
int abn_plan_set_windows(abn_plan* plan, const double* d_obs, const double* p0uu, const double* eqp,
                         const double* eqp_weight);
/* enqueue phase A (starts) -> select -> phase B (bootstraps) on the context's stream; asynchronous */
int abn_plan_run(abn_plan* plan);
int abn_plan_run_phase(abn_plan* plan, int32_t phase /* 0 = A+select, 1 = B */);
int abn_plan_sync(abn_plan* plan);
/* Diagnostics: out2 = chains the last persistent (queue + time-sliced) launch of phase A / phase B handed to its tail — the
 * last chains of such a launch finish on abn_fit_spec_kernel (four wavefronts per chain) instead of one by one on an
 * emptying GPU; 0 when the phase did not run persistent.  Same bits either way.  Synchronises like abn_plan_sync. */
int abn_plan_tail_handed(abn_plan* plan, int64_t* out2);
/* HIP-event time of the most recent launch of each kernel, in milliseconds (fit A, select, fit B) */
int abn_plan_kernel_ms(abn_plan* plan, double* ms3);
/* device pointer of raw[W x B x 7] (for an RCCL gather by the caller) and optional rebinding to a
 * caller-owned device buffer of the same size */
int abn_plan_raw_device_ptr(abn_plan* plan, void** dev_ptr);
int abn_plan_bind_raw(abn_plan* plan, void* dev_ptr);
/* copy results to the host; any pointer may be NULL.  models[W x 4], pred[W x N], resid[W x N],
 * raw[W x B x 7], info_a[W x S], info_b[W x B], best_start[W] (int32, -1 = no start of that window ended finite).
 * Returns ABN_ERR_NO_FINITE_FIT — AFTER filling every buffer — when any window has best_start = -1 (the reference
 * panics, src/ab_neutral.rs:28,100; metaprofile prints and skips the window, src/cli/metaprofile.rs:64-65): that
 * window's model, pred, resid and bootstrap rows are NaN, the other windows are valid. */
int abn_plan_download(abn_plan* plan, double* models, double* pred, double* resid, double* raw,
                      abn_fit_info* info_a, abn_fit_info* info_b, int32_t* best_start);
/* number of windows whose selection found no finite start in the last phase-A run (cheap: W int32 come back) */
int abn_plan_failed_windows(abn_plan* plan, int32_t* n_failed);
/* sums over all fits of the last run (for evals/s): out[0] = fits, out[1] = evals (cost() calls of the reference
 * algorithm = sum of abn_fit_info.evals), out[2] = iters, out[3] / out[4] = evaluations of out[1] in the start / bootstrap
 * fits that were NOT executed because the fit had reached a fixed point (options.no_fixed_point_skip) */
int abn_plan_counters(abn_plan* plan, int64_t* out5);
/* number of bytes of device memory the plan holds (index buffer included) */
int abn_plan_device_bytes(abn_plan* plan, int64_t* bytes);
/* which fit kernel the last run of each phase used (the choice depends on the size of the launch, never the bits):
 * out[0] = kernel of phase A (ABN_KERNEL_*), out[1] = its lanes per chain, out[2] / out[3] the same for phase B */
#define ABN_KERNEL_NONE 0         /* the phase has not run                                                   */
#define ABN_KERNEL_SPECULATIVE 1  /* four wavefronts per chain (few chains: latency-bound)                   */
#define ABN_KERNEL_RESIDENT 2     /* one launch, pedigree in LDS; lanes = 64: a wavefront per chain          */
#define ABN_KERNEL_PERSISTENT 3   /* resident, persistent wavefronts with a chain queue and time slicing     */
#define ABN_KERNEL_STREAM 4       /* rows re-read from HBM every evaluation                                  */
#define ABN_KERNEL_TWO_PASS 5     /* resident, long chains parked and resumed in a second launch             */
int abn_plan_last_kernels(abn_plan* plan, int32_t* out4);

/* ------------------------------------------------------------------ (5) one process, several GPUs of a node
 * The metaprofile loop (src/cli/metaprofile.rs:50-72) over all MI355X of a node from ONE host thread: a plan per
 * device on a stream of its own (the devices run concurrently), windows dealt in contiguous blocks — or, with fewer
 * windows than devices, every device repeats the cheap phase A (same inputs, same bits) and takes a contiguous slice
 * of the bootstraps.  Every fit is independent and every random draw is a function of the GLOBAL (window,
 * bootstrap) index, and the reduction tree is the pedigree's (abn_options.lanes_per_chain): the tables are
 * byte-identical for every number of devices.  abn_multi_run ends with the one exchange step of the path: the
 * gather of the bootstrap tables (56 B per fit) into every device's copy of raw[W x B x 7] over xGMI with RCCL
 * (in-place ncclAllGather for equal window blocks, else ncclBroadcast per block), enqueued behind the kernels on
 * each device's stream.  RCCL is bound at run time (librccl.so.1) and only when n_devices > 1.
 * devices[n_devices]: distinct HIP ordinals.  On failure *out may be non-NULL (abn_multi_last_error explains it)
 * and must still be destroyed. */
typedef struct abn_multi abn_multi;
int abn_multi_create(const int32_t* devices, int32_t n_devices, const abn_options* opts,
                     const double* generations /* N x 3 */, int32_t n_rows, int32_t n_windows, int32_t n_starts,
                     int32_t n_boot, abn_multi** out);
int abn_multi_destroy(abn_multi* m);
const char* abn_multi_last_error(const abn_multi* m);
/* ids[W] for ALL windows, before abn_multi_set_windows (as abn_plan_set_window_ids); NULL restores the default */
int abn_multi_set_window_ids(abn_multi* m, const uint32_t* ids);
/* D[W x N], p0uu[W], optional eqp[W], eqp_weight[W] for ALL windows (as abn_plan_set_windows) */
int abn_multi_set_windows(abn_multi* m, const double* d_obs, const double* p0uu, const double* eqp,
                          const double* eqp_weight);
/* A -> select -> B on every device, then the gather; asynchronous */
int abn_multi_run(abn_multi* m);
int abn_multi_sync(abn_multi* m);
/* out4 = window_offset, n_windows, boot_offset, n_boot of the shard of devices[device_index] */
int abn_multi_shard(abn_multi* m, int32_t device_index, int32_t* out4);
/* the same partition as host arithmetic, no device or handle needed: which windows / bootstraps device
 * `device_index` of `n_devices` takes (contiguous balanced blocks of windows; with fewer windows than devices every
 * device takes all windows and a block of the bootstraps) — the rule of the one-process-per-GPU path too */
int abn_multi_plan_shard(int32_t n_windows, int32_t n_boot, int32_t n_devices, int32_t device_index, int32_t* out4);
/* HIP-event milliseconds of devices[device_index]'s last run: ms3 = (phase A, selection, phase B), as abn_plan_kernel_ms */
int abn_multi_kernel_ms(abn_multi* m, int32_t device_index, double* ms3);
/* the gathered table raw[W x B x 7] in the memory of devices[device_index] (valid after abn_multi_sync) */
int abn_multi_raw_device_ptr(abn_multi* m, int32_t device_index, void** dev_ptr);
/* as abn_plan_download for all W windows (raw comes from the first device's gathered table); returns
 * ABN_ERR_NO_FINITE_FIT after filling every buffer when a window has no finite start */
int abn_multi_download(abn_multi* m, double* models, double* pred, double* resid, double* raw,
                       abn_fit_info* info_a, abn_fit_info* info_b, int32_t* best_start);
/* abn_plan_counters summed over the devices (bootstrap-sharded mode: the replicated phase-A fits count per device) */
int abn_multi_counters(abn_multi* m, int64_t* out5);
/* *ok = 1 when librccl.so.1 can be loaded and exports every symbol the gather uses */
int abn_multi_rccl_available(int* ok);

#ifdef __cplusplus
}
#endif
#endif /* ABNEUTRAL_H */
