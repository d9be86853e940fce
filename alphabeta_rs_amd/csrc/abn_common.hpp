// Device code of the ABneutral hot path for gfx950 (CDNA4, wave64): what every kernel family shares — launch arguments,
// the scalar pieces of one cost evaluation (genmatrix, power table, one triple), simplex bookkeeping, reductions.
// Kernels: abn_fit_kernel.hpp (plain / stream / two-pass / strict), abn_fit_refill.hpp (persistent, time-sliced),
// abn_fit_spec.hpp (four wavefronts per chain), abn_aux_kernels.hpp (selection, cost batch, rows, observations, indices),
// abn_pairwise_mx.hpp (pedigree construction).  abn_device.hpp includes them all.  Compiled with -ffp-contract=off:
// every fused multiply-add below is explicit and corresponds to one the reference executes.
//
// Mapping: a Nelder-Mead chain (one fit) is owned by a group of G lanes of one
// wavefront (G = 64: one wavefront per chain; G < 64 packs 64/G chains into a wavefront for small
// pedigrees).  Workgroups are ONE wavefront (64 threads) so chains in different wavefronts never
// synchronise.  All groups of a wavefront advance in lock-step, one cost evaluation per step
// ("evaluation-synchronous" state machines), so the expensive part — the cost function — never diverges.
//
// One cost evaluation (Problem::cost, src/structs.rs:194-216) for candidate x = (alpha,beta,weight,c):
//   P1  genmatrix(alpha,beta)                          src/divergence.rs:96-114      (all lanes)
//   P2  power table G^0..G^T, left-accumulated         src/divergence.rs:16-31       (lane r<3 = row r)
//       -> LDS, pw[k][kPw]
//   P3  per DISTINCT (t0,t1-t0,t2-t0) triple: dt1t2    src/divergence.rs:51-90       (lane per triple)
//       -> LDS dt[K]   (rows sharing a triple share the value bit for bit)
//   P4  per row: (D_i - c - dt[tid_i])^2 + penalty     src/structs.rs:208-213        (lane per row)
//   P5  xor-butterfly over the G lanes                 (the oracle's `lanes=G` order)
#pragma once
#include <type_traits>
#include <hip/hip_runtime.h>
#include <stdint.h>

#include "abn_philox.h"

namespace abn {

constexpr int kWave = 64;
constexpr int kPw = 10;  // doubles per entry of the power table in LDS: 9 elements + 1 so that entries are 16-byte aligned
static_assert(kPw % 2 == 0 && kPw >= 9, "load_matrix reads 16-byte aligned pairs");
constexpr int kStreamVec = 4;  // consecutive rows per lane and block in stream mode
#ifndef ABN_STREAM_BLOCKS
#define ABN_STREAM_BLOCKS 6
#endif
#ifndef ABN_STREAM_WAVES
#define ABN_STREAM_WAVES 2
#endif
constexpr int kStreamWaves = ABN_STREAM_WAVES;    // wavefronts per SIMD the stream variant is compiled for
constexpr int kStreamBlocks = ABN_STREAM_BLOCKS;  // row blocks a lane keeps in flight per loop iteration

// element-aligned vector types: global loads on gfx950 need dword alignment only
typedef uint32_t u32x4 __attribute__((ext_vector_type(4), aligned(4)));
typedef uint16_t u16x4 __attribute__((ext_vector_type(4), aligned(2)));
typedef double f64x2 __attribute__((ext_vector_type(2), aligned(8)));
typedef uint32_t u32x4_lds __attribute__((ext_vector_type(4)));  // naturally aligned: ds_read_b128
typedef double f64x2_lds __attribute__((ext_vector_type(2)));    // naturally aligned: ds_read_b128

// ---- fit states of the evaluation-synchronous Nelder-Mead machine
constexpr int ST_INIT0 = 0;     // 0..4: evaluating start vertex k           (argmin Solver::init)
constexpr int ST_REFLECT = 5;   // evaluating the reflected point
constexpr int ST_EXPAND = 6;    // evaluating the expanded point
constexpr int ST_CONTRACT = 7;  // evaluating the contracted point
constexpr int ST_SHRINK1 = 8;   // 8..11: evaluating shrunk vertex k = st-7
constexpr int ST_DONE = 12;
constexpr int kParkHead = 0, kParkTail = 64, kParkAvail = 128, kParkHeaderInts = 192;  // one cache line each
constexpr int kParkShards = 64;  // independent FIFOs (workgroup b uses b mod 64): a cache line serves ~100 M atomics/s
constexpr int kFitSuspended = 4;  // internal status between the two passes of a long chain
constexpr unsigned kSliceErrLostEntry = 1u;  // FitArgs::slice_status[0]

struct FitInfoDev {  // layout of abn_fit_info (include/abneutral.h)
  double best_cost;
  int32_t iters;
  int32_t evals;
  int32_t status;
  int32_t lanes;
};

struct FitArgs {
  // pedigree topology (shared by all windows)
  const uint32_t* tri;   // [K] t0 | (t1-t0)<<8 | (t2-t0)<<16
  const uint16_t* tid;   // [N] row -> triple
  int N, K, T, TP;       // TP = table pitch (>= T+1)
  int chain_stride;      // doubles of LDS per chain (even): kPw*TP + KP + ...
  // per-window data; wstride = 0 broadcasts window 0's scalars to every chain
  const double* p_uu;    // [W] p0uu
  const double* eqp;     // [W]
  const double* eqp_w;   // [W]
  int wstride;
  // observed divergences
  int dmode;             // 0: D[w*N+i]   1: pred[w*N+i] + resid[w*N + idx[(w*C+j)*N+i]]   2: D[chain*N+i]
                         //    (2: bootstrap observations materialised once per fit by abn_make_dstar_kernel)
  const double* D;
  const double* pred;
  const double* resid;
  const uint32_t* idx;
  // start simplices
  int smode;             // 0: simplex0[chain*20]   1: [model[w], vary() x4] from Philox
  const double* simplex0;
  const double* model;   // [W*4]
  uint64_t seed;
  uint32_t window_offset, boot_offset;
  const uint32_t* wid;   // nullable [W]: the window's index in the Philox counters (default window_offset + w)
  // Residual reduction tree (the oracle's `lanes` code): a property of the PEDIGREE, not of the launch.
  //   kTreeCanon (auto options, every LDS-resident pedigree): 64 accumulators — accumulator v sums rows v, v + 64, ...
  //     in that order — combined from the high lane bits down: v^32, v^16, v^8, then v <-> 7-v inside 8, v <-> 3-v
  //     inside 4, v^1.  EVERY kernel runs it at its native cost: a wavefront per chain holds one accumulator per lane
  //     (two permlane swaps, four DPP steps); the packed kernels hold the 64/G accumulators v = gl + G j of a chain in
  //     each lane, combine them in registers (that is the v^32, v^16 (, v^8) part) and finish with the same DPP steps.
  //     So the kernel is chosen by the size of the launch and the bits are the pedigree's (tree64_finish).
  //   G (explicit lanes_per_chain, packed kernels only) or G | 3 << 8 (streamed pedigrees): G accumulators, one per
  //     lane, xor-butterfly 1, 2, 4, ... (group_sum_dpp).
  int tree;
  int strict;            // host dispatch only: the STRICT instantiation (serial row-order sum; `tree` is then 1)
  // chains: W windows x C chains
  int W, C;
  int max_iters;
  // Two-pass execution of long chains (abn_api.hip: phase A with many chains).  Pass 1: iter_cap > 0 — a chain
  // that is still running after iter_cap iterations stores its Nelder-Mead state (32 doubles) and appends its
  // index to susp_list.  Pass 2: resume != 0 — block b, group g continues chain susp_list[b*NG+g] (for
  // b*NG+g < *susp_count) from the stored state to the end.  Same arithmetic either way: results are
  // bit-identical to an uninterrupted run.
  int iter_cap;          // 0 = unlimited
  int resume;
  double* state;         // [W*C*32]
  int* susp_list;        // [W*C]
  int* susp_count;       // [1]
  int shrink_variant;
  // argmin 0.8.1 leaves the simplex untouched after a rejected contraction (shrink_variant == 0), and the cost
  // function is deterministic: from then on every iteration repeats the same two evaluations and the same
  // rejection until max_iters.  no_skip == 0: such a chain is finished on the spot with the counters it would
  // have reached (iters = max_iters, evals += 2 per remaining iteration, status MAX_ITERS) — the same outputs
  // as running the repetitions.  The evaluations not executed are summed into *skipped (nullable).
  int no_skip;
  unsigned long long* skipped;
  unsigned* queue;       // abn_fit_refill_kernel: next chain to start (zeroed by the host); nullptr = no persistent launch
  // Time slicing in the persistent kernel (quantum > 0).  A chain that has run `quantum` evaluations while others wait
  // (unstarted chains in the queue, or parked ones) stores its state (`state`, 32 doubles, as the two-pass hand-over)
  // at its next iteration boundary and appends itself to the FIFO `parked`; its group takes the next waiting chain —
  // a fresh one while there are any, else the oldest parked one.  Chains of very different length then advance
  // together and the launch no longer ends with a few long chains on an idle GPU.  Same arithmetic, same bits.
  //   park_ht[kParkHead] = next entry to take, [kParkTail] = entries reserved, [kParkAvail] = entries published and
  //   not yet claimed (a signed credit: a group claims one with an atomic subtract and gives it back if there was
  //   none — no compare-and-swap loop: thousands of groups reach the end of a quantum together); parked[] starts at
  //   -1 and an entry is published by its (agent-scope) store after the chain's state has been written through.
  int quantum;
  unsigned park_cap;     // entries of parked[] PER SHARD (kParkShards shards, each with its own three counters)
  unsigned* park_ht;
  int* parked;
  // persistent launches: slice_status[0] |= kSliceErrLostEntry when a claimed FIFO entry never appeared (the group goes
  // idle, the chain's outputs stay unwritten), slice_status[1] += chains finished (results written).  The host compares
  // the count with W x C after a time-sliced launch: a lost or never-resumed chain is an error, not stale output.
  unsigned* slice_status;
  // Tail hand-over of a time-sliced persistent launch (tail_cap > 0): once no chain waits any more (queue and this workgroup's
  // FIFO empty) and at most tail_cap chains of the launch are unfinished, the running chains park at their next iteration
  // boundary — state as for time slicing, index appended to susp_list — and the launch ends; abn_fit_spec_kernel (spec_resume
  // != 0: workgroup b takes chain susp_list[b], b < *susp_count, up from `state`) runs them to the end with four wavefronts
  // per chain: the long chains that would otherwise finish one by one on an emptying GPU at the packed kernel's step time.
  // Same arithmetic, same bits; slice_status[2] counts the chains handed over.
  int tail_cap;
  int spec_resume;
#ifdef ABN_MEASUREMENT_KNOBS
  // Wave priority by chain age in the persistent kernel (prio_mode != 0): the wavefront's s_setprio level is the number of
  // thresholds prio_t[] that the evaluations of its OLDEST running chain have passed (mode 1), or 3 minus that (mode 2).
  // Measured in round 4 (scripts/prio_sweep.sh, profiles/r04_prio_sweep.txt): no gain — kept out of the product build.
  int prio_mode;
  int prio_t[3];
  int drop_entry;        // fault injection for the tests: FIFO shard 0 never publishes its first entry
#endif
  double sd_tol;
  double gap_tol;        // 64 * sd_tol, precomputed on the host so that it stays a scalar (kernarg) operand
  // outputs (fit order)
  double* best;          // [W*C*4]
  FitInfoDev* info;      // [W*C]
  double* raw;           // nullable [W*C*7] (src/boot_model.rs:86-91)
  unsigned long long* dbg;  // diagnostic builds only (ABN_STAMPS): per-segment cycle sums of chain 0
};

// In-kernel stamps (MI355X guide §7): only in a separate diagnostic build, never in the shipped library.
#ifdef ABN_STAMPS
#define ABN_STAMP(slot)                                                           \
  do {                                                                            \
    __builtin_amdgcn_sched_barrier(0);                                            \
    unsigned long long t__;                                                       \
    asm volatile("s_memtime %0\n\ts_waitcnt lgkmcnt(0)" : "=s"(t__)::"memory"); \
    __builtin_amdgcn_sched_barrier(0);                                            \
    seg[slot] += t__ - tprev;                                                     \
    tprev = t__;                                                                  \
  } while (0)
#else
#define ABN_STAMP(slot) \
  do {                  \
  } while (0)
#endif

// ------------------------------------------------------------------------------------------------
// scalar pieces
// ------------------------------------------------------------------------------------------------
__device__ __forceinline__ double fma3(double a, double b, double c) { return __builtin_fma(a, b, c); }

// src/divergence.rs:96-114 (powi(2) = x*x).  Named scalars, not an array: a lane-dependent choice of row
// must stay a v_cndmask on registers and never become an indexed (scratch) load.
struct Gen {
  double g0, g1, g2, g3, g4, g5, g6, g7, g8;
};
__device__ __forceinline__ Gen genmatrix(double alpha, double beta) {
  Gen G;
  G.g0 = (1.0 - alpha) * (1.0 - alpha);
  G.g1 = 2.0 * (1.0 - alpha) * alpha;
  G.g2 = alpha * alpha;
  G.g3 = 0.25 * ((beta + 1.0 - alpha) * (beta + 1.0 - alpha));
  G.g4 = 0.5 * (beta + 1.0 - alpha) * (alpha + 1.0 - beta);
  G.g5 = 0.25 * ((alpha + 1.0 - beta) * (alpha + 1.0 - beta));
  G.g6 = beta * beta;
  G.g7 = 2.0 * (1.0 - beta) * beta;
  G.g8 = (1.0 - beta) * (1.0 - beta);
  return G;
}

// src/alphabeta.rs:62-65
__device__ __forceinline__ double p_uu_est(double alpha, double beta) {
  return (beta * ((1.0 - beta) * (1.0 - beta) - (1.0 - alpha) * (1.0 - alpha) - 1.0)) /
         ((alpha + beta) * ((alpha + beta - 1.0) * (alpha + beta - 1.0) - 2.0));
}
// src/structs.rs:146-149
__device__ __forceinline__ double est_mm(double alpha, double beta) {
  return (alpha * ((1.0 - alpha) * (1.0 - alpha) - (1.0 - beta) * (1.0 - beta) - 1.0)) /
         ((alpha + beta) * ((alpha + beta - 1.0) * (alpha + beta - 1.0) - 2.0));
}
// src/structs.rs:151-154
__device__ __forceinline__ double est_um(double alpha, double beta) {
  return (4.0 * alpha * beta * (alpha + beta - 2.0)) /
         ((alpha + beta) * ((alpha + beta - 1.0) * (alpha + beta - 1.0) - 2.0));
}

// xor-butterfly over the G lanes of a group; every lane ends with the same sum.
template <int G>
__device__ __forceinline__ double group_sum(double v) {
#pragma unroll
  for (int off = 1; off < G; off <<= 1) v = v + __shfl_xor(v, off, kWave);
  return v;
}

// ------------------------------------------------------------------------------------------------
// Cross-lane helpers on DPP (no LDS crossbar): data-parallel-primitive moves have VALU latency, a
// ds_bpermute round trip costs an LDS access.  All lanes of the wavefront are active at every call site.
// ------------------------------------------------------------------------------------------------
template <int CTRL>
__device__ __forceinline__ double dpp_mov(double v) {
  const int lo = __double2loint(v), hi = __double2hiint(v);
  const int nlo = __builtin_amdgcn_update_dpp(lo, lo, CTRL, 0xF, 0xF, true);
  const int nhi = __builtin_amdgcn_update_dpp(hi, hi, CTRL, 0xF, 0xF, true);
  return __hiloint2double(nhi, nlo);
}
constexpr int kDppQuadXor1 = 0xB1;       // quad_perm:[1,0,3,2]
constexpr int kDppQuadXor2 = 0x4E;       // quad_perm:[2,3,0,1]
constexpr int kDppRowHalfMirror = 0x141; // lane i <-> 7-i  inside each 8 lanes
constexpr int kDppRowMirror = 0x140;     // lane i <-> 15-i inside each 16 lanes
constexpr int kDppQuadBcast0 = 0x00, kDppQuadBcast1 = 0x55, kDppQuadBcast2 = 0xAA, kDppQuadBcast3 = 0xFF;

// P2: the power table G^0..G^T, left-accumulated exactly as the reference (result = result.dot(matrix),
// src/divergence.rs:25-30): every product element is fma(a_i2,b_2j, fma(a_i1,b_1j, fma(a_i0,b_0j, 0))) —
// matrixmultiply's k-ascending FMA accumulation.  Table layout pw[k][kPw] (entry k at k*kPw, row r at +3r; element 9
// is padding): entries are 16-byte aligned so that P3 reads a matrix with four ds_read_b128 and one ds_read_b64
// (256 B/clk) — with a pitch of 9 doubles hipcc paired the reads into ds_read2_b64, which the LDS serves at half
// that rate, and the LDS array, shared by the CU's wavefronts, was busy for most of a packed launch.
//
// G = 16, 32 (several chains per wavefront, throughput-bound): nine lanes of the group, lane 4i+j holds
// element (i,j) of the running power; the three operands R[i][0..2] are the other lanes of the same quad
// (DPP quad broadcasts), so a step is 3 FMAs + 6 DPP moves + 1 LDS store instead of 9 FMAs + 3 stores
// (-4 % time on C3's phase B).  G = 8 and G = 64: lanes 0..2 own one row each — with one chain per
// wavefront the single dependent FMA chain of the nine-lane form is slower (measured: +25 % on the
// latency-bound phase A of the 351-row pedigree), three independent row chains interleave better.
template <int G>
__device__ __forceinline__ void build_power_table(Gen Gm, int T, int TP, double* pw, int gl) {
  (void)TP;
  // Opaque register copies: without them hipcc rewrites the lane-dependent selects below into an indexed
  // load from a scratch copy of the matrix.
  asm("" : "+v"(Gm.g0), "+v"(Gm.g1), "+v"(Gm.g2));
  asm("" : "+v"(Gm.g3), "+v"(Gm.g4), "+v"(Gm.g5));
  asm("" : "+v"(Gm.g6), "+v"(Gm.g7), "+v"(Gm.g8));
  if constexpr (G == 16 || G == 32) {
    if (gl < 12) {  // quads 0..2 of the group; lane position 3 of each quad mirrors position 2
      const int i = gl >> 2, jr = gl & 3;
      const bool j1 = (jr == 1), j2 = (jr >= 2);  // position 3 computes and stores what position 2 does (same address)
      const bool i1 = (i == 1), i2 = (i == 2);
      const double gc0 = j2 ? Gm.g2 : (j1 ? Gm.g1 : Gm.g0);  // column j of G
      const double gc1 = j2 ? Gm.g5 : (j1 ? Gm.g4 : Gm.g3);
      const double gc2 = j2 ? Gm.g8 : (j1 ? Gm.g7 : Gm.g6);
      const int j = j2 ? 2 : jr;
      double* pe = pw + 3 * i + j;
      double r = (i == j) ? 1.0 : 0.0;                        // identity, :21-24
      pe[0] = r;
      if (T >= 1) {
        r = i2 ? gc2 : (i1 ? gc1 : gc0);                      // matrix.clone(), :25  (G[i][j])
        pe[kPw] = r;
        double* pk = pe + 2 * kPw;
        for (int k = 2; k <= T; ++k) {                        // :27-29 (hipcc does not unroll a loop of DPP
          const double b0 = dpp_mov<kDppQuadBcast0>(r), b1 = dpp_mov<kDppQuadBcast1>(r);  // operations with a run-time
          const double b2 = dpp_mov<kDppQuadBcast2>(r);                                     // trip count; by hand: no gain)
          r = fma3(b2, gc2, fma3(b1, gc1, fma3(b0, gc0, 0.0)));
          pk[0] = r;
          pk += kPw;
        }
      }
    }
  } else if (gl < 3) {  // one exec mask for the whole chain of products: lanes 0..2 of each group, row gl
    const bool is1 = (gl == 1), is2 = (gl == 2);
    double r0 = (is1 || is2) ? 0.0 : 1.0, r1 = is1 ? 1.0 : 0.0, r2 = is2 ? 1.0 : 0.0;  // identity, :21-24
    double* prow = pw + 3 * gl;
    prow[0] = r0;
    prow[1] = r1;
    prow[2] = r2;
    if (T >= 1) {
      r0 = is2 ? Gm.g6 : (is1 ? Gm.g3 : Gm.g0);  // matrix.clone(), :25
      r1 = is2 ? Gm.g7 : (is1 ? Gm.g4 : Gm.g1);
      r2 = is2 ? Gm.g8 : (is1 ? Gm.g5 : Gm.g2);
      prow[kPw + 0] = r0;
      prow[kPw + 1] = r1;
      prow[kPw + 2] = r2;
      double* pk = prow + 2 * kPw;
#pragma unroll 2
      for (int k = 2; k <= T; ++k) {  // :27-29
        const double n0 = fma3(r2, Gm.g6, fma3(r1, Gm.g3, fma3(r0, Gm.g0, 0.0)));
        const double n1 = fma3(r2, Gm.g7, fma3(r1, Gm.g4, fma3(r0, Gm.g1, 0.0)));
        const double n2 = fma3(r2, Gm.g8, fma3(r1, Gm.g5, fma3(r0, Gm.g2, 0.0)));
        r0 = n0;
        r1 = n1;
        r2 = n2;
        pk[0] = r0;
        pk[1] = r1;
        pk[2] = r2;
        pk += kPw;
      }
    }
  }
}

// ------------------------------------------------------------------------------------------------
// P2 on the f64 matrix instruction.  v_mfma_f64_4x4x4 multiplies four independent 4x4 blocks; its dot products
// are the k-ASCENDING chain of fused multiply-adds, rounded after every step — bit for bit the reference's
// `result.dot(matrix)` (scripts/mfma_f64_probe.hip: 1.28 M random elements, 3x3 blocks padded with zeros, NaN /
// infinities / denormals / overflow: no mismatch against fma(a2,b2, fma(a1,b1, fma(a0,b0, 0)))).  This is not a
// reshaping of the path into a GEMM: the 3x3 transition-matrix product IS the operation, the instruction is used
// as a four-chains-wide FMA chain.
//   layout (read off the probe): A[blk][i][k] at lane i + 4 blk + 16 k, B[blk][k][j] at lane j + 4 blk + 16 k,
//   D[blk][i][j] at lane j + 4 blk + 16 i.
// With A = G^T (constant) and B = (G^n)^T the result D = G^T (G^n)^T = (G^n G)^T has B's layout again, so the
// whole chain is one dependent MFMA per power (48 cycles, measured) with no data movement in between; block blk
// works for chain blk of the wavefront (64/G chains; G = 8: two instructions per power).  Lane (x = lane & 3,
// y = lane >> 4) of block blk holds element [x][y] of chain blk's running power and stores it to that chain's
// table pw[n][3 x + y].  All 64 lanes must be active (MFMA ignores EXEC).
// ------------------------------------------------------------------------------------------------
__device__ __forceinline__ double lane_fetch(double v, int src_lane) {  // ds_bpermute: v of lane src_lane
  const int lo = __builtin_amdgcn_ds_bpermute(src_lane << 2, __double2loint(v));
  const int hi = __builtin_amdgcn_ds_bpermute(src_lane << 2, __double2hiint(v));
  return __hiloint2double(hi, lo);
}

// Build switch: -DABN_NO_MATRIX_FMA compiles the VALU form (build_power_table) instead — same bits, for A/B timing.
#ifdef ABN_NO_MATRIX_FMA
constexpr bool kMatrixFma = false;
#else
constexpr bool kMatrixFma = true;
#endif

template <int G>
__device__ __forceinline__ void build_power_table_mx(double al, double be, int T, double* lds0, int chain_stride,
                                                     double* dump, int lane) {
  constexpr int NG = kWave / G;               // chains of this wavefront
  constexpr int NH = NG > 4 ? 2 : 1;          // matrix instructions per product (four chains each)
  const int x = lane & 3, blk = (lane >> 2) & 3, y = lane >> 4;
  double A[NH], B[NH];
  double* dst[NH];
  int step[NH];
  (void)dump;
#pragma unroll
  for (int h = 0; h < NH; ++h) {
    const int c = blk + 4 * h;                // the chain this lane's block works for
    double ac = al, bc = be;
    if (NG > 1) {                             // its candidate: any lane of group c holds (alpha, beta)
      const int src = (c < NG ? c : 0) * G;
      ac = lane_fetch(al, src);
      bc = lane_fetch(be, src);
    }
    Gen Gc = genmatrix(ac, bc);
    asm("" : "+v"(Gc.g0), "+v"(Gc.g1), "+v"(Gc.g2));
    asm("" : "+v"(Gc.g3), "+v"(Gc.g4), "+v"(Gc.g5));
    asm("" : "+v"(Gc.g6), "+v"(Gc.g7), "+v"(Gc.g8));
    // A = G^T: element G[y][x]; zero outside 3x3
    const double r0 = x == 0 ? Gc.g0 : (x == 1 ? Gc.g1 : Gc.g2);
    const double r1 = x == 0 ? Gc.g3 : (x == 1 ? Gc.g4 : Gc.g5);
    const double r2 = x == 0 ? Gc.g6 : (x == 1 ? Gc.g7 : Gc.g8);
    const double gyx = y == 0 ? r0 : (y == 1 ? r1 : r2);
    const bool in3 = (x < 3) && (y < 3);
    A[h] = in3 ? gyx : 0.0;
    // B = (G^1)^T: element G[x][y] = A of the lane with x and y exchanged (same block)
    B[h] = lane_fetch(A[h], y + 4 * blk + 16 * x);
    // lanes outside the 3x3 block (or of an unused block) store too — to the padding element [9] of the same entry of the
    // first chain's table, which nobody reads: no store predicate in the loop and ONE pitch for all lanes, so the
    // stores of a trip differ by an immediate offset
    const bool st = in3 && (c < NG);
    dst[h] = st ? lds0 + (size_t)c * chain_stride + 3 * x + y : lds0 + 9;
    step[h] = kPw;
    dst[h][0] = (x == y) ? 1.0 : 0.0;         // identity, :21-24
    dst[h] += step[h];
    if (T >= 1) dst[h][0] = B[h];             // matrix.clone(), :25
  }
  // :27-29.  Power n is stored while the instruction for power n+1 runs (the store needs the finished result
  // anyway; issued right behind the dependent MFMA it hides in its 48-cycle shadow).
  // (build_power_table_mx_pre below repeats this loop for one chain.)
  if (T >= 2) {   // two powers per trip, alternating registers: see build_power_table_mx_pre
    double Bn[NH], Bm[NH];
#pragma unroll
    for (int h = 0; h < NH; ++h) Bn[h] = __builtin_amdgcn_mfma_f64_4x4x4f64(A[h], B[h], 0.0, 0, 0, 0);
    int n = 3;
    for (; n + 1 <= T; n += 2) {
#pragma unroll
      for (int h = 0; h < NH; ++h) {
        Bm[h] = __builtin_amdgcn_mfma_f64_4x4x4f64(A[h], Bn[h], 0.0, 0, 0, 0);
        dst[h] += step[h];
        dst[h][0] = Bn[h];
      }
      __builtin_amdgcn_sched_barrier(0);   // (or hipcc pairs these stores with the next ones, which wait for power n)
#pragma unroll
      for (int h = 0; h < NH; ++h) {
        Bn[h] = __builtin_amdgcn_mfma_f64_4x4x4f64(A[h], Bm[h], 0.0, 0, 0, 0);
        dst[h] += step[h];
        dst[h][0] = Bm[h];
      }
    }
    if (n <= T) {
#pragma unroll
      for (int h = 0; h < NH; ++h) {
        Bm[h] = __builtin_amdgcn_mfma_f64_4x4x4f64(A[h], Bn[h], 0.0, 0, 0, 0);
        dst[h] += step[h];
        dst[h][0] = Bn[h];
        Bn[h] = Bm[h];
      }
    }
#pragma unroll
    for (int h = 0; h < NH; ++h) {
      dst[h] += step[h];
      dst[h][0] = Bn[h];
    }
  }
}

// One chain per wavefront with A = G^T and B = (G^1)^T already in the matrix-instruction layout (zeros outside the
// 3x3 block of block 0): the speculative kernel's keeper prepares them for every candidate it hands out.
__device__ __forceinline__ void build_power_table_mx_pre(double A, double B, int T, double* pw, double* dump, int lane) {
  const int x = lane & 3, blk = (lane >> 2) & 3, y = lane >> 4;
  const bool st = (x < 3) && (y < 3) && (blk == 0);
  double* dst = st ? pw + 3 * x + y : pw + 9;   // the other lanes: the entry's padding element (never read)
  constexpr int step = kPw;
  (void)dump;
  dst[0] = (x == y) ? 1.0 : 0.0;              // identity, :21-24
  dst += step;
  if (T >= 1) dst[0] = B;                     // matrix.clone(), :25
  if (T >= 2) {                               // :27-29
    // Two powers per trip with alternating registers: the instruction for power n is issued as soon as power n - 1 is there
    // and the store of power n - 1 follows in its shadow.  (One register and one power per trip made hipcc put the store
    // first — it has to read the register the instruction overwrites — and with the loop's branch a power cost 86 cycles
    // instead of the instruction's 48: in-kernel stamps, 31 powers of the 351-row pedigree 2 680 cycles.)
    double Bn = __builtin_amdgcn_mfma_f64_4x4x4f64(A, B, 0.0, 0, 0, 0);   // power 2
    int n = 3;
    for (; n + 1 <= T; n += 2) {
      const double Bm = __builtin_amdgcn_mfma_f64_4x4x4f64(A, Bn, 0.0, 0, 0, 0);  // power n
      dst += step;
      dst[0] = Bn;                                                                 // power n - 1
      __builtin_amdgcn_sched_barrier(0);   // (or hipcc pairs this store with the next one, which waits for power n)
      Bn = __builtin_amdgcn_mfma_f64_4x4x4f64(A, Bm, 0.0, 0, 0, 0);               // power n + 1
      dst += step;
      dst[0] = Bm;
    }
    if (n <= T) {
      const double Bm = __builtin_amdgcn_mfma_f64_4x4x4f64(A, Bn, 0.0, 0, 0, 0);
      dst += step;
      dst[0] = Bn;
      Bn = Bm;
    }
    dst += step;
    dst[0] = Bn;
  }
}

// conditional divergence of one start state, src/divergence.rs:68-87 (this exact association)
__device__ __forceinline__ double cond_div(double a0, double a1, double a2, double b0, double b1, double b2) {
  return 0.5 * (a0 * b1 + a1 * b0 + a1 * b2 + a2 * b1) + (a0 * b2 + a2 * b0);
}

// one table entry (16-byte aligned: kPw is even and so is every chain's LDS stride)
__device__ __forceinline__ void load_matrix(const double* m, double (&M)[9]) {
  const f64x2_lds* v = reinterpret_cast<const f64x2_lds*>(m);
#pragma unroll
  for (int e = 0; e < 4; ++e) {
    const f64x2_lds x = v[e];
    M[2 * e] = x[0];
    M[2 * e + 1] = x[1];
  }
  M[8] = m[8];
}

// P3 for one distinct triple, src/divergence.rs:52-89.  Two load batches (G^t0, then G^a and G^b) with a
// scheduling barrier in between keep the live registers of the fit kernel under 128 (4 wavefronts per SIMD).
__device__ __forceinline__ double triple_dt(uint32_t tr, const double* pw, int TP, double sv0, double sv1,
                                            double sv2) {
  (void)TP;
  const int t0 = tr & 0xff, ea = (tr >> 8) & 0xff, eb = (tr >> 16) & 0xff;
  double P[9], A[9], B[9];
  load_matrix(pw + t0 * kPw, P);
  // svt0 = sv_gzero.t().dot(G^t0), :55
  const double s0 = fma3(sv2, P[6], fma3(sv1, P[3], fma3(sv0, P[0], 0.0)));
  const double s1 = fma3(sv2, P[7], fma3(sv1, P[4], fma3(sv0, P[1], 0.0)));
  const double s2 = fma3(sv2, P[8], fma3(sv1, P[5], fma3(sv0, P[2], 0.0)));
  __builtin_amdgcn_sched_barrier(0);
  load_matrix(pw + ea * kPw, A);
  load_matrix(pw + eb * kPw, B);
  const double d_mm = cond_div(A[6], A[7], A[8], B[6], B[7], B[8]);  // :68-73
  const double d_um = cond_div(A[3], A[4], A[5], B[3], B[4], B[5]);  // :75-80
  const double d_uu = cond_div(A[0], A[1], A[2], B[0], B[1], B[2]);  // :82-87
  return s0 * d_uu + s1 * d_um + s2 * d_mm;                          // :89
}

// ------------------------------------------------------------------------------------------------
// Simplex bookkeeping, "dimension per lane".  Nelder-Mead's vector algebra is element-wise over the four
// parameters, so lane gl of a group keeps ONE parameter dimension d = gl & 3 of all five vertices, in
// rank order (vx[0] best ... vx[4] worst), next to a replicated copy of the five costs.  Everything is
// statically indexed (registers only).  The candidate is re-assembled for the cost function with four
// lane broadcasts per evaluation.
// ------------------------------------------------------------------------------------------------
// Insert element I into the sorted prefix [0, I): the inner step of std's stable insertion sort
// (len <= 20), is_less(a,b) = a.cost < b.cost; a NaN compares Equal, i.e. never moves.
template <int I>
__device__ __forceinline__ void insert_tail(double (&c)[5], double (&v)[5]) {
  const double fi = c[I], xi = v[I];
  bool b[I];
  b[I - 1] = fi < c[I - 1];
#pragma unroll
  for (int j = I - 2; j >= 0; --j) b[j] = b[j + 1] && (fi < c[j]);
  double nc[I + 1], nv[I + 1];
  nc[I] = b[I - 1] ? c[I - 1] : fi;
  nv[I] = b[I - 1] ? v[I - 1] : xi;
#pragma unroll
  for (int j = I - 1; j >= 1; --j) {
    nc[j] = b[j - 1] ? c[j - 1] : (b[j] ? fi : c[j]);
    nv[j] = b[j - 1] ? v[j - 1] : (b[j] ? xi : v[j]);
  }
  nc[0] = b[0] ? fi : c[0];
  nv[0] = b[0] ? xi : v[0];
#pragma unroll
  for (int j = 0; j <= I; ++j) {
    c[j] = nc[j];
    v[j] = nv[j];
  }
}
__device__ __forceinline__ void sort5(double (&c)[5], double (&v)[5]) {
  insert_tail<1>(c, v);
  insert_tail<2>(c, v);
  insert_tail<3>(c, v);
  insert_tail<4>(c, v);
}

// ------------------------------------------------------------------------------------------------
// Cross-lane reductions
// ------------------------------------------------------------------------------------------------
// v_permlane16_swap / v_permlane32_swap (gfx950) with both operands = v return (a, b) with
// a + b = v[l] + v[l ^ 16] (resp. ^ 32) in every lane: the xor-16 / xor-32 butterfly step without LDS.
template <int W>
__device__ __forceinline__ double swap_sum(double v) {
  const unsigned lo = (unsigned)__double2loint(v), hi = (unsigned)__double2hiint(v);
  if (W == 16) {
    const auto l = __builtin_amdgcn_permlane16_swap(lo, lo, false, false);
    const auto h = __builtin_amdgcn_permlane16_swap(hi, hi, false, false);
    return __hiloint2double((int)h[0], (int)l[0]) + __hiloint2double((int)h[1], (int)l[1]);
  } else {
    const auto l = __builtin_amdgcn_permlane32_swap(lo, lo, false, false);
    const auto h = __builtin_amdgcn_permlane32_swap(hi, hi, false, false);
    return __hiloint2double((int)h[0], (int)l[0]) + __hiloint2double((int)h[1], (int)l[1]);
  }
}

// Sum over the G lanes of a group with the value tree of an xor-butterfly (offsets 1,2,4,...): after the
// quad steps every quad is uniform, so the half-mirror / mirror partners hold exactly the values the xor-4 /
// xor-8 partners would (addition is commutative), and the result is bit-identical to group_sum<G>.
template <int G>
__device__ __forceinline__ double group_sum_dpp(double v) {
  v = v + dpp_mov<kDppQuadXor1>(v);
  v = v + dpp_mov<kDppQuadXor2>(v);
  if (G >= 8) v = v + dpp_mov<kDppRowHalfMirror>(v);
  if (G >= 16) v = v + dpp_mov<kDppRowMirror>(v);
  if (G >= 32) v = swap_sum<16>(v);
  if (G >= 64) v = swap_sum<32>(v);
  return v;
}

// ------------------------------------------------------------------------------------------------
// The canonical residual tree (FitArgs::tree == kTreeCanon): 64 accumulators, high lane bits first.
// ------------------------------------------------------------------------------------------------
constexpr int kTreeCanon = 0x10040;      // the oracle's `lanes` code: 64 accumulators | mirror-descending steps
constexpr int kDppRowRor8 = 0x128;       // lane i <- lane i ^ 8 inside each 16 lanes
constexpr int kDppQuadRev = 0x1B;        // quad_perm:[3,2,1,0]: lane i <-> 3 - i inside each quad

// steps v <-> 7-v (inside 8), v <-> 3-v (inside 4), v ^ 1
__device__ __forceinline__ double tree64_tail8(double v) {
  v = v + dpp_mov<kDppRowHalfMirror>(v);
  v = v + dpp_mov<kDppQuadRev>(v);
  v = v + dpp_mov<kDppQuadXor1>(v);
  return v;
}
// acc[j] = accumulator gl + G j of this lane's chain (G = 64: the lane's own).  Every lane of the group ends with the
// chain's sum.
template <int G>
__device__ __forceinline__ double tree64_finish(const double (&acc)[kWave / G]) {
  double v;
  if constexpr (G == 64) {
    v = swap_sum<32>(acc[0]);                                  // v ^ 32
    v = swap_sum<16>(v);                                       // v ^ 16
    v = v + dpp_mov<kDppRowRor8>(v);                           // v ^ 8
  } else if constexpr (G == 32) {
    v = acc[0] + acc[1];                                       // v ^ 32: the lane's two accumulators
    v = swap_sum<16>(v);
    v = v + dpp_mov<kDppRowRor8>(v);
  } else if constexpr (G == 16) {
    const double b0 = acc[0] + acc[2], b1 = acc[1] + acc[3];   // v ^ 32
    v = b0 + b1;                                               // v ^ 16
    v = v + dpp_mov<kDppRowRor8>(v);
  } else {
    static_assert(G == 8, "lanes per chain");
    const double b0 = acc[0] + acc[4], b1 = acc[1] + acc[5], b2 = acc[2] + acc[6], b3 = acc[3] + acc[7];
    const double c0 = b0 + b2, c1 = b1 + b3;                   // v ^ 16
    v = c0 + c1;                                               // v ^ 8
  }
  return tree64_tail8(v);
}

// acc + t[0] + t[1] + ... + t[n-1], added in that order (strict order).  t is 16-byte aligned; every lane of a group reads
// the same addresses (LDS broadcast), eight terms per batch of loads.
__device__ __forceinline__ double serial_sum_lds(const double* t, int n, double acc) {
  const f64x2_lds* v = reinterpret_cast<const f64x2_lds*>(t);
  int i = 0;
  for (; i + 8 <= n; i += 8) {
    const f64x2_lds a0 = v[i / 2], a1 = v[i / 2 + 1], a2 = v[i / 2 + 2], a3 = v[i / 2 + 3];
    acc = acc + a0[0];
    acc = acc + a0[1];
    acc = acc + a1[0];
    acc = acc + a1[1];
    acc = acc + a2[0];
    acc = acc + a2[1];
    acc = acc + a3[0];
    acc = acc + a3[1];
  }
  for (; i < n; ++i) acc = acc + t[i];
  return acc;
}

// The same sum for a pedigree of at most 16 rows held one row per lane (lane i: row i's term, +0.0 in the lanes past the
// last row) by a wavefront that serves ONE chain: 0.0 + t[0] + t[1] + ... in row order through v_readlane — no LDS round
// trip on the latency path (the bundled six-row pedigree is summed this way by default: abn_options.strict_order = 0).
// The +0.0 terms past row n - 1 change no bit (no partial sum is -0.0).
__device__ __forceinline__ double serial_sum_lanes16(double term, int n) {
  const int lo = __double2loint(term), hi = __double2hiint(term);
  double acc = 0.0;
  // Eight lane reads at constant lanes first (independent: they pipeline; read pair by pair into the same scalar registers
  // they serialise with the additions), then the dependent additions, four per wavefront-uniform exit test.
#pragma unroll
  for (int base = 0; base < 16; base += 8) {
    if (base >= n) break;
    int l[8], h[8];
#pragma unroll
    for (int j = 0; j < 8; ++j) {
      l[j] = __builtin_amdgcn_readlane(lo, base + j);
      h[j] = __builtin_amdgcn_readlane(hi, base + j);
    }
#pragma unroll
    for (int j = 0; j < 4; ++j) acc = acc + __hiloint2double(h[j], l[j]);
    if (base + 4 < n) {
#pragma unroll
      for (int j = 4; j < 8; ++j) acc = acc + __hiloint2double(h[j], l[j]);
    }
  }
  return acc;
}

}  // namespace abn
