// Device code of the ABneutral hot path for gfx950 (CDNA4, wave64).  Compiled with -ffp-contract=off:
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
  // Wave priority by chain age in the persistent kernel (prio_mode != 0): the wavefront's s_setprio level is the number of
  // thresholds prio_t[] that the evaluations of its OLDEST running chain have passed (mode 1), or 3 minus that (mode 2).
  int prio_mode;
  int prio_t[3];
#ifdef ABN_MEASUREMENT_KNOBS
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
    // lanes outside the 3x3 block (or of an unused block) store too — always to `dump`, a slot of the caller's that
    // nobody reads before it is rewritten: no store predicate in the loop
    const bool st = in3 && (c < NG);
    dst[h] = st ? lds0 + (size_t)c * chain_stride + 3 * x + y : dump;
    step[h] = st ? kPw : 0;
    dst[h][0] = (x == y) ? 1.0 : 0.0;         // identity, :21-24
    dst[h] += step[h];
    if (T >= 1) dst[h][0] = B[h];             // matrix.clone(), :25
  }
  // :27-29.  Power n is stored while the instruction for power n+1 runs (the store needs the finished result
  // anyway; issued right behind the dependent MFMA it hides in its 48-cycle shadow).
  // (build_power_table_mx_pre below repeats this loop for one chain.)
  if (T >= 2) {
#pragma unroll
    for (int h = 0; h < NH; ++h) B[h] = __builtin_amdgcn_mfma_f64_4x4x4f64(A[h], B[h], 0.0, 0, 0, 0);
    for (int n = 3; n <= T; ++n) {
#pragma unroll
      for (int h = 0; h < NH; ++h) {
        const double nxt = __builtin_amdgcn_mfma_f64_4x4x4f64(A[h], B[h], 0.0, 0, 0, 0);
        dst[h] += step[h];
        dst[h][0] = B[h];
        B[h] = nxt;
      }
    }
#pragma unroll
    for (int h = 0; h < NH; ++h) {
      dst[h] += step[h];
      dst[h][0] = B[h];
    }
  }
}

// One chain per wavefront with A = G^T and B = (G^1)^T already in the matrix-instruction layout (zeros outside the
// 3x3 block of block 0): the speculative kernel's keeper prepares them for every candidate it hands out.
__device__ __forceinline__ void build_power_table_mx_pre(double A, double B, int T, double* pw, double* dump, int lane) {
  const int x = lane & 3, blk = (lane >> 2) & 3, y = lane >> 4;
  const bool st = (x < 3) && (y < 3) && (blk == 0);
  double* dst = st ? pw + 3 * x + y : dump;
  const int step = st ? kPw : 0;
  dst[0] = (x == y) ? 1.0 : 0.0;              // identity, :21-24
  dst += step;
  if (T >= 1) dst[0] = B;                     // matrix.clone(), :25
  if (T >= 2) {                               // :27-29
    B = __builtin_amdgcn_mfma_f64_4x4x4f64(A, B, 0.0, 0, 0, 0);
    for (int n = 3; n <= T; ++n) {
      const double nxt = __builtin_amdgcn_mfma_f64_4x4x4f64(A, B, 0.0, 0, 0, 0);
      dst += step;
      dst[0] = B;
      B = nxt;
    }
    dst += step;
    dst[0] = B;
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

// ------------------------------------------------------------------------------------------------
// The fit kernel.  RMAX > 0 ("resident", needs N <= G*RMAX): the chain's observed divergences
// (bootstrap: pred_i + resid[idx_i], gathered once per fit) are staged in LDS, each lane keeps its
// triples and its rows' triple ids in registers; an evaluation touches no global memory.
// RMAX == 0 ("stream"): for larger pedigrees the rows are re-read every evaluation (bootstrap: the u32
// index row is re-streamed from HBM, coalesced).
// LDS per workgroup: 64/G chains x (kPw (T+1) + KP + 4 [+ NP]) doubles.
// ------------------------------------------------------------------------------------------------
// STRICT (abn_options.strict_order): the residuals are summed SERIALLY in row order — the reference's `square_sum += ...`
// (src/structs.rs:206-213), the oracle's lanes = 1 — instead of with the tree: the lanes write their rows' terms to LDS
// (resident: N more doubles per chain; stream: chunks of 8 G rows) and every lane of the group adds them up in order
// (same address in the whole group: an LDS broadcast).  N dependent additions per evaluation: the price of an opt-in mode.
constexpr int kStrictRowsPerLane = 8;  // rows per lane and chunk of the strict stream variant

template <int G, int RMAX, bool TWOPASS = false, bool STRICT = false>
__global__ __launch_bounds__(kWave, TWOPASS ? 2 : (RMAX == 0 ? kStreamWaves : 3)) void abn_fit_kernel(const FitArgs a) {
  // RMAX == 0: stream mode for long rows (deep loop, kStreamWaves wavefronts per SIMD); RMAX == -1: stream mode for
  // mid-size pedigrees whose rows never fill the deep loop (pairs of blocks, three wavefronts per SIMD)
  constexpr int NG = kWave / G;
  constexpr bool STREAM = (RMAX <= 0);
  constexpr int SNB = RMAX == 0 ? kStreamBlocks : 2;  // row blocks a lane keeps in flight in stream mode
  constexpr int RR = RMAX > 0 ? RMAX : 1;
  extern __shared__ __align__(16) double lds[];

  const int lane = threadIdx.x;
  const int g = lane / G;
  const int gl = lane - g * G;
  const int dim = gl & 3;
  const long long total = (long long)a.W * a.C;
  const long long slot = (long long)blockIdx.x * NG + g;
  long long chain_raw = slot;
  bool valid = slot < total;
  if constexpr (TWOPASS) {
    if (a.resume) {  // second pass: the compacted list of suspended chains
      valid = slot < (long long)*a.susp_count;
      chain_raw = valid ? (long long)a.susp_list[slot] : 0;
    }
  }
  const long long chain = valid ? chain_raw : 0;
  const int w = (int)(chain / a.C);
  const int j = (int)(chain - (long long)w * a.C);
  const int N = a.N, K = a.K, TP = a.TP;

  double* pw = lds + (size_t)g * a.chain_stride;
  double* dtab = pw + kPw * TP;
  double* wconst = dtab + ((K + 1) & ~1);                                 // p0uu, p0mm, eqp, eqp_weight*N
  double* dobs = wconst + 4;                                              // resident mode: N doubles

  const int wi = w * a.wstride;
  const size_t wN = (size_t)w * (size_t)N;
  const uint32_t* idx_row = (a.dmode == 1) ? a.idx + (size_t)chain * (size_t)N : nullptr;
  const size_t dN = (a.dmode == 2) ? (size_t)chain * (size_t)N : wN;  // base of this chain's rows in a.D

  // ---- per-chain constants live in LDS (they would otherwise pin 8 VGPRs for the whole fit)
  if (gl == 0) {
    const double p_uu0 = a.p_uu[wi];
    wconst[0] = p_uu0;
    wconst[1] = 1.0 - p_uu0;                          // p0mm, src/ab_neutral.rs:23
    wconst[2] = a.eqp[wi];
    wconst[3] = a.eqp_w[wi] * (double)N;              // eqp_weight * nrows, src/structs.rs:210-211
  }
  // ---- resident mode: observed divergences staged in LDS once per fit (bootstrap: gathered through the
  // index row); this lane's triples and row->triple ids (as LDS byte offsets into dt) in registers
  uint32_t tidp[(RR + 1) / 2];  // two 16-bit triple ids per register
  const bool canon = !STREAM && a.tree == kTreeCanon;  // the canonical 64-accumulator tree (FitArgs::tree), else G accumulators
  uint32_t* tri_s = reinterpret_cast<uint32_t*>(dobs + ((N + 1) & ~1));  // this chain's copy of the triple list
  // strict order: the rows' terms (resident: behind the triple list, N doubles; stream: behind the constants, 8 G doubles)
  double* term = STREAM ? dobs : reinterpret_cast<double*>(tri_s) + (((K + 1) / 2 + 1) & ~1);
  if (!STREAM) {
    for (int t = gl; t < K; t += G) tri_s[t] = a.tri[t];
#pragma unroll
    for (int q = 0; q < (RR + 1) / 2; ++q) tidp[q] = 0u;
#pragma unroll
    for (int q = 0; q < RR; ++q) {
      const int i = gl + G * q;
      if (i < N) {
        tidp[q / 2] |= (uint32_t)a.tid[i] << (16 * (q & 1));
        dobs[i] = (a.dmode == 1) ? a.pred[wN + i] + a.resid[wN + idx_row[i]]  // src/boot_model.rs:50-54
                                 : a.D[dN + i];
      }
    }
  }
  __syncthreads();

  // ---- start simplex: this lane's dimension of the five vertices
  double vx[5], c[5];
  if (a.smode == 0) {
    const double* s0 = a.simplex0 + (size_t)chain * 20;
#pragma unroll
    for (int k = 0; k < 5; ++k) vx[k] = s0[4 * k + dim];
  } else {  // [params, vary() x4], src/boot_model.rs:69-75
    const uint32_t k0 = (uint32_t)a.seed, k1 = (uint32_t)(a.seed >> 32);
    const uint32_t wg = a.wid ? a.wid[w] : a.window_offset + (uint32_t)w, bg = a.boot_offset + (uint32_t)j;
    vx[0] = a.model[4 * w + dim];
#pragma unroll
    for (int v = 1; v < 5; ++v) {
      uint32_t r[4];
      philox4x32_10((uint32_t)(v - 1) * 2u + (uint32_t)(dim >> 1), bg, wg, kTagJitter, k0, k1, r);
      const uint32_t r0 = r[0], r1 = r[1], r2 = r[2], r3 = r[3];
      const bool odd = (dim & 1) != 0;
      vx[v] = vary_one(vx[0], odd ? r2 : r0, odd ? r3 : r1);
    }
  }

#ifdef ABN_STAMPS
  unsigned long long seg[8] = {0, 0, 0, 0, 0, 0, 0, 0}, tprev = 0;
  asm volatile("s_memtime %0\n\ts_waitcnt lgkmcnt(0)" : "=s"(tprev)::"memory");
#endif
  // ---- one cost evaluation; xd = this lane's dimension of its group's candidate.  Lanes of a quad hold
  // dimensions 0..3 of the same chain, so the candidate is re-assembled with four quad broadcasts.
  auto eval = [&](double xd) -> double {
    ABN_STAMP(6);  // Nelder-Mead bookkeeping since the previous evaluation
    const double al = dpp_mov<kDppQuadBcast0>(xd), be = dpp_mov<kDppQuadBcast1>(xd);
    const double wt = dpp_mov<kDppQuadBcast2>(xd), ic = dpp_mov<kDppQuadBcast3>(xd);
    const double p_mm = wconst[1];
    const double sv0 = wconst[0], sv1 = wt * p_mm, sv2 = (1.0 - wt) * p_mm;  // src/divergence.rs:44
    const double puu = p_uu_est(al, be);                     // src/divergence.rs:92 (early: overlaps P2)
    const double dq = puu - wconst[2];
    const double pen = wconst[3] * (dq * dq);                // src/structs.rs:210-212
    ABN_STAMP(0);
    // this lane's first triple is fetched before the power table is built and every later one a round ahead:
    // the LDS latency of the triple list stays off the path
    uint32_t tr = STREAM ? a.tri[gl < K ? gl : 0] : tri_s[gl < K ? gl : 0];
    if constexpr (kMatrixFma && G == kWave) build_power_table_mx<G>(al, be, a.T, lds, a.chain_stride, dtab, lane);  // P1 + P2
    else build_power_table<G>(genmatrix(al, be), a.T, TP, pw, gl);
    __syncthreads();
    ABN_STAMP(1);
    if (!STREAM) {                                           // P3: ceil(K/G) rounds, one triple per lane
#pragma unroll 1
      for (int t = gl; t < K; t += G) {
        const uint32_t trn = tri_s[t + G < K ? t + G : 0];
        dtab[t] = triple_dt(tr, pw, TP, sv0, sv1, sv2);
        tr = trn;
      }
    } else {
#pragma unroll 1
      for (int t = gl; t < K; t += G) {
        const uint32_t trn = a.tri[t + G < K ? t + G : 0];
        dtab[t] = triple_dt(tr, pw, TP, sv0, sv1, sv2);
        tr = trn;
      }
    }
    __syncthreads();
    ABN_STAMP(2);
    double acc = 0.0;                                        // P4
    bool summed = false;
    if (!STREAM) {
      constexpr int RC = RR < 8 ? RR : 8;                    // eight rows per lane at a time
      constexpr int NA = kWave / G;                          // canonical tree: accumulators gl + G j held by this lane
      double av[NA];
#pragma unroll
      for (int j = 0; j < NA; ++j) av[j] = 0.0;
#pragma unroll
      for (int q0 = 0; q0 < RR; q0 += RC) {
        double dv[RC], tv[RC], x[RC];
#pragma unroll
        for (int q = 0; q < RC; ++q) {                       // all LDS reads first, then the arithmetic;
          const int i = gl + G * (q0 + q);                   // rows past the end read row N-1 and add +0.0
          dv[q] = dobs[i < N ? i : N - 1];
          tv[q] = dtab[(tidp[(q0 + q) / 2] >> (16 * ((q0 + q) & 1))) & 0xffffu];
        }
#pragma unroll
        for (int q = 0; q < RC; ++q) {
          const double r = dv[q] - ic - tv[q];
          const double term = r * r + pen;
          x[q] = ((gl + G * (q0 + q)) < N) ? term : 0.0;     // x + 0.0 == x bit for bit (no sum is -0.0)
        }
        if constexpr (STRICT) {
          if (G == kWave && RR == 1 && N <= 16) {            // one chain per wavefront, one row per lane: lane reads
            acc = serial_sum_lanes16(x[0], N);
          } else {
#pragma unroll
            for (int q = 0; q < RC; ++q)
              if ((gl + G * (q0 + q)) < N) term[gl + G * (q0 + q)] = x[q];
          }
        } else if (canon) {                                  // uniform: row gl + G q belongs to accumulator gl + G (q mod NA)
#pragma unroll
          for (int q = 0; q < RC; ++q) av[(q0 + q) % NA] = av[(q0 + q) % NA] + x[q];
        } else {
#pragma unroll
          for (int q = 0; q < RC; ++q) acc = acc + x[q];
        }
      }
      if constexpr (STRICT) {                                // `square_sum += ...` in row order, src/structs.rs:206-213
        if (!(G == kWave && RR == 1 && N <= 16)) {
          __syncthreads();
          acc = serial_sum_lds(term, N, 0.0);
        }
        summed = true;
      } else if (canon) {
        acc = tree64_finish<G>(av);                          // P5, the pedigree's tree
        summed = true;
      }
    } else if constexpr (STRICT) {
      // strict stream mode: chunks of 8 G rows — lane l computes rows base + l + G q (coalesced 8-byte loads), the terms
      // go to LDS and every lane adds them up in row order
      constexpr int CH = kStrictRowsPerLane * G;
      for (int base = 0; base < N; base += CH) {
        const int cnt = (N - base) < CH ? (N - base) : CH;
#pragma unroll
        for (int q = 0; q < kStrictRowsPerLane; ++q) {
          const int i = base + gl + G * q;
          if (i < N) {
            const double dd = (a.dmode == 1) ? a.pred[wN + i] + a.resid[wN + idx_row[i]] : a.D[dN + i];
            const double r = dd - ic - dtab[a.tid[i]];
            term[gl + G * q] = r * r + pen;
          }
        }
        __syncthreads();
        acc = serial_sum_lds(term, cnt, acc);
        __syncthreads();
      }
      summed = true;
    } else {
      // stream mode.  Lane l owns row blocks of kStreamVec = 4 consecutive rows: rows 4(l + G q) .. +3 for
      // q = 0, 1, ... — so the u32 index row is read with one 16-byte load per lane (1 KiB per wavefront
      // instruction), pred with two and the triple ids with one 8-byte load.  Two blocks (8 rows) per lane are
      // in flight per iteration, the dependent residual gathers issued together.  The per-lane accumulation
      // order (block by block, row by row) is what the oracle's lanes code `G | 3 << 8` reproduces.
      constexpr int V = kStreamVec;
      const int stride = V * G;
      int base = V * gl;
      // NBK full blocks of this lane in flight; consumed block by block, row by row
      auto blocks = [&](auto nbk) {
        constexpr int NB = decltype(nbk)::value;
        for (; base + (NB - 1) * stride + V <= N; base += NB * stride) {
          double d[NB * V], t[NB * V];
          u16x4 tq[NB];
#pragma unroll
          for (int b = 0; b < NB; ++b) tq[b] = *reinterpret_cast<const u16x4*>(a.tid + base + b * stride);
          if (a.dmode == 1) {
            u32x4 ix[NB];
            f64x2 pl[NB], ph[NB];
#pragma unroll
            for (int b = 0; b < NB; ++b) {
              ix[b] = *reinterpret_cast<const u32x4*>(idx_row + base + b * stride);
              pl[b] = *reinterpret_cast<const f64x2*>(a.pred + wN + base + b * stride);
              ph[b] = *reinterpret_cast<const f64x2*>(a.pred + wN + base + b * stride + 2);
            }
            const double* rs = a.resid + wN;
            double rg[NB * V];
#pragma unroll
            for (int b = 0; b < NB; ++b)
#pragma unroll
              for (int e = 0; e < V; ++e) rg[b * V + e] = rs[ix[b][e]];
#pragma unroll
            for (int b = 0; b < NB; ++b) {                             // src/boot_model.rs:50-54
              d[b * V + 0] = pl[b][0] + rg[b * V + 0];
              d[b * V + 1] = pl[b][1] + rg[b * V + 1];
              d[b * V + 2] = ph[b][0] + rg[b * V + 2];
              d[b * V + 3] = ph[b][1] + rg[b * V + 3];
            }
          } else {
            f64x2 ql[NB], qh[NB];
#pragma unroll
            for (int b = 0; b < NB; ++b) {
              ql[b] = *reinterpret_cast<const f64x2*>(a.D + dN + base + b * stride);
              qh[b] = *reinterpret_cast<const f64x2*>(a.D + dN + base + b * stride + 2);
            }
#pragma unroll
            for (int b = 0; b < NB; ++b) {
              d[b * V + 0] = ql[b][0];
              d[b * V + 1] = ql[b][1];
              d[b * V + 2] = qh[b][0];
              d[b * V + 3] = qh[b][1];
            }
          }
#pragma unroll
          for (int b = 0; b < NB; ++b)
#pragma unroll
            for (int e = 0; e < V; ++e) t[b * V + e] = dtab[tq[b][e]];
#pragma unroll
          for (int e = 0; e < NB * V; ++e) {
            const double r = d[e] - ic - t[e];
            acc = acc + (r * r + pen);
          }
        }
      };
      blocks(std::integral_constant<int, SNB>{});                    // deep loop for long rows (HBM latency) ...
      if (SNB > 2) blocks(std::integral_constant<int, 2>{});          // ... then pairs for what is left
      for (; base < N; base += stride) {                             // remaining (possibly partial) blocks
#pragma unroll
        for (int e = 0; e < V; ++e) {
          const int i = base + e;
          if (i < N) {
            const double dd = (a.dmode == 1) ? a.pred[wN + i] + a.resid[wN + idx_row[i]] : a.D[dN + i];
            const double r = dd - ic - dtab[a.tid[i]];
            acc = acc + (r * r + pen);
          }
        }
      }
    }
    ABN_STAMP(3);
    if (!summed) acc = group_sum_dpp<G>(acc);                // P5
    __syncthreads();
    ABN_STAMP(4);
    return acc;
  };

  // ---- evaluation-synchronous Nelder-Mead (argmin 0.8.1 NelderMead + Executor).
  // Solver::init and NelderMead::shrink evaluate "the vertex at a fixed position" and rotate the arrays,
  // so no register array is ever indexed at run time; both live outside the hot loop.
  int st = valid ? ST_REFLECT : ST_DONE;
  int iter = 0, evals = 0;
  double xc = 0.0, x0 = 0.0, xr = 0.0, bx = __builtin_nan("");
  double fr = 0.0, best_cost = __builtin_inf();
  bool have_best = false;
  int fin_status = 2;

  // IterState::update() + terminate_internal() + the head of next_iter (centroid, reflection)
  auto begin_iteration = [&](bool count_iter) {
    const double c_best = c[0];
    if (c_best < best_cost || (__builtin_isinf(c_best) && __builtin_isinf(best_cost) &&
                               (__builtin_signbit(c_best) == __builtin_signbit(best_cost)))) {
      bx = vx[0];
      best_cost = c_best;
      have_best = true;
    }
    if (count_iter) ++iter;
    // NelderMead::terminate (sample SD of the five costs < sd_tolerance) -> max_iters -> target_cost.
    // Shortcut: with sorted finite costs some |c_k - mean| >= (c4 - c0)/2, so the computed SD is at least
    // (c4 - c0)/4 (1 - 2^-50); a gap above 64*tol can never test as converged and the division and square
    // root are skipped.  Any NaN makes the gap test false and falls through to the full form.
    bool converged = false;
    if (!((c[4] - c[0]) > a.gap_tol)) {
      double sum = 0.0;
#pragma unroll
      for (int k = 0; k < 5; ++k) sum = sum + c[k];
      const double c0 = sum / 5.0;
      double ss = 0.0;
#pragma unroll
      for (int k = 0; k < 5; ++k) ss = ss + (c[k] - c0) * (c[k] - c0);
      const double sd = __builtin_sqrt(1.0 / (5.0 - 1.0) * ss);
      converged = sd < a.sd_tol;
    }
    int status = -1;
    if (converged) status = 0;
    else if (iter >= a.max_iters) status = 1;
    else if (best_cost <= -__builtin_inf()) status = 3;
    // results are written after the loop (keeps output addresses out of the loop's registers); plain
    // selects here: conditional stores to two different scalars made hipcc spill them to scratch
    const bool suspend = TWOPASS && status < 0 && a.iter_cap > 0 && iter >= a.iter_cap;  // first of two passes
    const bool done = status >= 0 || suspend;
    fin_status = (status >= 0) ? (have_best ? status : 2) : (suspend ? kFitSuspended : fin_status);
    // centroid (p0 + p1 + p2 + p3) * (1/4), reflection x0 + (x0 - worst) * alpha
    double acc = vx[0];
    acc = acc + vx[1];
    acc = acc + vx[2];
    acc = acc + vx[3];
    x0 = acc * (1.0 / 4.0);
    xr = x0 + (x0 - vx[4]) * 1.0;
    xc = xr;
    st = done ? ST_DONE : ST_REFLECT;
  };

  if (!TWOPASS || !a.resume) {
    // Solver::init: the five start costs in input order, stable sort, first termination check.  All chains
    // of a wavefront start together.
#pragma unroll 1
    for (int k = 0; k < 5; ++k) {
      const double f = eval(vx[0]);
      const double tv = vx[0];
#pragma unroll
      for (int q = 0; q < 4; ++q) {
        c[q] = c[q + 1];
        vx[q] = vx[q + 1];
      }
      c[4] = f;
      vx[4] = tv;
    }
    if (valid) {
      evals = 5;
      sort5(c, vx);
      begin_iteration(false);
    }
  } else if (TWOPASS && valid) {
    // continue a suspended chain: simplex (this lane's dimension), costs, best-so-far and counters as stored
    // at an iteration boundary; centroid and reflection are recomputed (same arithmetic, same bits)
    const double* sp = a.state + (size_t)chain * 32;
#pragma unroll
    for (int k = 0; k < 5; ++k) {
      vx[k] = sp[4 * k + dim];
      c[k] = sp[20 + k];
    }
    bx = sp[25 + dim];
    best_cost = sp[29];
    const int* ip = reinterpret_cast<const int*>(sp + 30);
    iter = ip[0];
    evals = ip[1];
    have_best = ip[2] != 0;
    double acc = vx[0];
    acc = acc + vx[1];
    acc = acc + vx[2];
    acc = acc + vx[3];
    x0 = acc * (1.0 / 4.0);
    xr = x0 + (x0 - vx[4]) * 1.0;
    xc = xr;
    st = ST_REFLECT;
  }

  while (__ballot(st != ST_DONE) != 0ull) {
    const double f = eval(xc);
    // ---- decisions of NelderMead::next_iter as predicates (no divergent control flow on the hot path)
    const bool active = st != ST_DONE;
    const bool is_ref = st == ST_REFLECT, is_exp = st == ST_EXPAND, is_con = st == ST_CONTRACT;
    const bool acc_r = is_ref && (f < c[3]) && (f >= c[0]);      // reflection accepted
    const bool go_exp = is_ref && !acc_r && (f < c[0]);           // try expansion
    const bool go_con = is_ref && !acc_r && !go_exp && (f >= c[3]);  // contraction towards the worst
    const bool nan_ref = is_ref && !acc_r && !go_exp && !go_con;  // only reachable with a NaN cost
    const bool keep_r = is_exp && !(f < fr);                      // expansion not better: keep the reflection
    const bool acc_c = is_con && (f < c[4]);
    const bool rej_c = is_con && !acc_c;
    const bool do_insert = acc_r || is_exp || acc_c;
    const bool start_shrink = nan_ref || (rej_c && a.shrink_variant != 0);
    const bool do_begin = do_insert || (rej_c && a.shrink_variant == 0);  // argmin 0.8.1: rejected contraction leaves the simplex
    evals += active ? 1 : 0;
    if (rej_c && a.shrink_variant == 0 && a.no_skip == 0) {  // fixed point: finish the chain (FitArgs::no_skip)
      const int rest = a.max_iters - iter - 1;               // iterations that would repeat this one
      evals += 2 * rest;
      iter += rest;
      if (a.skipped && gl == 0 && rest > 0) atomicAdd(a.skipped, 2ull * (unsigned long long)rest);
    }
    const double xi = keep_r ? xr : xc;
    const double fi = keep_r ? fr : f;
    fr = is_ref ? f : fr;
    const double x_e = x0 + (xr - x0) * 2.0;        // expansion  x0 + (xr - x0) * gamma
    const double x_c = x0 + (vx[4] - x0) * 0.5;     // contraction x0 + (xw - x0) * rho
    xc = go_exp ? x_e : (go_con ? x_c : xc);
    st = go_exp ? ST_EXPAND : (go_con ? ST_CONTRACT : st);
    if (do_insert) {
      c[4] = fi;
      vx[4] = xi;
      insert_tail<4>(c, vx);
    }
    if (do_begin) begin_iteration(true);
    // ---- NelderMead::shrink (NaN costs, or the textbook variant after a rejected contraction): vertices
    // 1..4 move towards the best by sigma and are re-evaluated in order.  Rare; the other chains idle.
    if (__ballot(start_shrink) != 0ull) {
#pragma unroll 1
      for (int k = 1; k < 5; ++k) {
        const double nv = vx[0] + (vx[1] - vx[0]) * 0.5;
        const double fk = eval(start_shrink ? nv : xc);
        if (start_shrink) {
          ++evals;
#pragma unroll
          for (int q = 1; q < 4; ++q) {
            c[q] = c[q + 1];
            vx[q] = vx[q + 1];
          }
          c[4] = fk;
          vx[4] = nv;
        }
      }
      if (start_shrink) {
        sort5(c, vx);
        begin_iteration(true);
      }
    }
  }

#ifdef ABN_STAMPS
  if (a.dbg && chain_raw == 0 && gl == 0) {
    for (int q = 0; q < 8; ++q) a.dbg[q] = seg[q];
    a.dbg[7] = (unsigned long long)evals;
  }
#endif
  // ---- first pass of a two-pass run: park the chains that hit the iteration cap
  if (TWOPASS && valid && fin_status == kFitSuspended) {
    double* sp = a.state + (size_t)chain * 32;
    if (gl < 4) {
#pragma unroll
      for (int k = 0; k < 5; ++k) sp[4 * k + gl] = vx[k];
      sp[25 + gl] = bx;
    }
    if (gl == 0) {
#pragma unroll
      for (int k = 0; k < 5; ++k) sp[20 + k] = c[k];
      sp[29] = best_cost;
      int* ip = reinterpret_cast<int*>(sp + 30);
      ip[0] = iter;
      ip[1] = evals;
      ip[2] = have_best ? 1 : 0;
      ip[3] = 0;
      a.susp_list[atomicAdd(a.susp_count, 1)] = (int)chain;
    }
  }
  // ---- results in fit order: best_param, (best_cost, iters, evals, status, lanes) and, for bootstraps,
  // the row [alpha, beta, weight, intercept, est_mm, est_um, est_uu] of src/boot_model.rs:86-91
  const double b0 = dpp_mov<kDppQuadBcast0>(bx), b1 = dpp_mov<kDppQuadBcast1>(bx);
  if (valid) {
    if (gl < 4) a.best[(size_t)chain * 4 + gl] = bx;
    if (gl == 0) {
      FitInfoDev fo;
      fo.best_cost = best_cost;
      fo.iters = iter;
      fo.evals = evals;
      fo.status = fin_status;
      fo.lanes = STRICT ? 1 : (STREAM ? (G | ((kStreamVec - 1) << 8)) : a.tree);  // reduction-order code (oracle: `lanes`)
      a.info[chain] = fo;
    }
    if (a.raw) {
      double* ro = a.raw + (size_t)chain * 7;
      if (gl < 4) ro[gl] = bx;
      if (gl == 4) ro[4] = est_mm(b0, b1);
      if (gl == 5) ro[5] = est_um(b0, b1);
      if (gl == 6) ro[6] = p_uu_est(b0, b1);
    }
  }
}

// ------------------------------------------------------------------------------------------------
// Persistent variant of the resident fit kernel for launches with many more chains than the GPU holds
// wavefronts (phase B of a multi-window shard, phase A of the metaprofile shape).  Chains differ in length
// (300 ... 900 evaluations on C3), so in abn_fit_kernel a wavefront lives as long as the longest of its 64/G
// chains and its other groups idle: 1.29x the wavefront-steps the chains need.  Here the grid is one resident
// set of wavefronts; a group whose fit ends writes its results and takes the next chain from an atomic queue
// (`FitArgs::queue`, zeroed by the host; initial chains are the slots themselves), so every group stays busy
// until the queue is empty.  Solver::init's five start evaluations become states of the evaluation-synchronous
// machine (ST_INIT0..4) so that a freshly started chain runs next to chains in mid-flight.
// Every chain computes exactly what it computes in abn_fit_kernel (same code for the evaluation, the same
// Nelder-Mead update), and results are written by chain index: outputs are bit-identical and independent
// of the schedule.  Resident mode only (RMAX > 0), single pass.
// Time slicing (FitArgs::quantum > 0): the queue alone leaves a long tail — the launch ends with whole long chains
// that started late, on a GPU that is emptying.  So a chain that has run a quantum of evaluations while others wait
// parks itself at its next iteration boundary (state to memory, an entry in its workgroup's FIFO shard) and its group
// takes the next waiting chain: chains of different length advance together and the groups stay busy to the end
// (C4 shard phase B 4.96 -> 4.44 ms at a quantum of 256; 128 costs more in parks than it gains, 768 gains less).
// What the protocol needs on this hardware (each learnt from a measurement): no agent-scope fence per
// park (it writes back and invalidates the XCD's L2: state through sc1 stores / loads and a wavefront-level wait); no
// compare-and-swap loop (thousands of groups end a quantum together: a credit counter instead); the counters sharded
// over 64 sets of cache lines (one line serves ~100 M device-scope atomics a second).
// ------------------------------------------------------------------------------------------------
constexpr int ST_IDLE = 13;

#ifndef ABN_REFILL_MIN_WAVES
#define ABN_REFILL_MIN_WAVES 3
#endif
template <int G, int RMAX>
__global__ __launch_bounds__(kWave, ABN_REFILL_MIN_WAVES) void abn_fit_refill_kernel(const FitArgs a) {
  static_assert(RMAX > 0, "resident mode only");
  constexpr int NG = kWave / G;
  constexpr int RR = RMAX;
  extern __shared__ __align__(16) double lds[];

  const int lane = threadIdx.x;
  const int g = lane / G;
  const int gl = lane - g * G;
  const int dim = gl & 3;
  const unsigned total = (unsigned)((long long)a.W * a.C);
  const int N = a.N, K = a.K, TP = a.TP;

  double* pw = lds + (size_t)g * a.chain_stride;
  double* dtab = pw + kPw * TP;
  double* wconst = dtab + ((K + 1) & ~1);                                 // p0uu, p0mm, eqp, eqp_weight*N
  double* dobs = wconst + 4;                                              // N doubles
  uint32_t* tri_s = reinterpret_cast<uint32_t*>(dobs + ((N + 1) & ~1));  // this group's copy of the triple list

  // ---- per-group constants of the topology: triple list in LDS, this lane's row -> triple ids in registers
  // time slicing: this workgroup's FIFO of parked chains (shards are statistically alike: no stealing)
  unsigned* const pht = a.quantum > 0 ? a.park_ht + (blockIdx.x & (kParkShards - 1)) * kParkHeaderInts : nullptr;
  int* const pk = a.quantum > 0 ? a.parked + (size_t)(blockIdx.x & (kParkShards - 1)) * a.park_cap : nullptr;
  const bool canon = a.tree == kTreeCanon;  // the canonical 64-accumulator tree (FitArgs::tree), else G accumulators
  uint32_t tidp[(RR + 1) / 2];
  for (int t = gl; t < K; t += G) tri_s[t] = a.tri[t];
#pragma unroll
  for (int q = 0; q < (RR + 1) / 2; ++q) tidp[q] = 0u;
#pragma unroll
  for (int q = 0; q < RR; ++q) {
    const int i = gl + G * q;
    if (i < N) tidp[q / 2] |= (uint32_t)a.tid[i] << (16 * (q & 1));
  }

  // ---- per-chain state
  unsigned chain = blockIdx.x * NG + g;
  double vx[5], c[5];
  int st = ST_IDLE;
  int iter = 0, evals = 0;
  double xc = 0.0, x0 = 0.0, xr = 0.0, bx = __builtin_nan("");
  double fr = 0.0, best_cost = __builtin_inf();
  bool have_best = false;
  int fin_status = 2;
  int q_start = 0;           // time slicing: evals of this chain when its current quantum began
  bool fresh_done = false;   // this group has seen the queue of unstarted chains empty
#pragma unroll
  for (int k = 0; k < 5; ++k) {
    vx[k] = 0.0;
    c[k] = 0.0;
  }

  // start chain `chain` in this group: window constants and observed divergences (bootstrap: gathered through
  // the index row, src/boot_model.rs:50-54) to LDS, start simplex, fresh optimiser state
  auto setup_chain = [&]() {
    const int w = (int)(chain / (unsigned)a.C);
    const int j = (int)(chain - (unsigned)w * (unsigned)a.C);
    const int wi = w * a.wstride;
    const size_t wN = (size_t)w * (size_t)N;
    if (gl == 0) {
      const double p_uu0 = a.p_uu[wi];
      wconst[0] = p_uu0;
      wconst[1] = 1.0 - p_uu0;                          // p0mm, src/ab_neutral.rs:23
      wconst[2] = a.eqp[wi];
      wconst[3] = a.eqp_w[wi] * (double)N;              // eqp_weight * nrows, src/structs.rs:210-211
    }
    const uint32_t* idx_row = (a.dmode == 1) ? a.idx + (size_t)chain * (size_t)N : nullptr;
    const size_t dN = (a.dmode == 2) ? (size_t)chain * (size_t)N : wN;
#pragma unroll
    for (int q = 0; q < RR; ++q) {
      const int i = gl + G * q;
      if (i < N)
        dobs[i] = (a.dmode == 1) ? a.pred[wN + i] + a.resid[wN + idx_row[i]] : a.D[dN + i];
    }
    if (a.smode == 0) {
      const double* s0 = a.simplex0 + (size_t)chain * 20;
#pragma unroll
      for (int k = 0; k < 5; ++k) vx[k] = s0[4 * k + dim];
    } else {  // [params, vary() x4], src/boot_model.rs:69-75
      const uint32_t k0 = (uint32_t)a.seed, k1 = (uint32_t)(a.seed >> 32);
      const uint32_t wg = a.wid ? a.wid[w] : a.window_offset + (uint32_t)w, bg = a.boot_offset + (uint32_t)j;
      vx[0] = a.model[4 * w + dim];
#pragma unroll
      for (int v = 1; v < 5; ++v) {
        uint32_t r[4];
        philox4x32_10((uint32_t)(v - 1) * 2u + (uint32_t)(dim >> 1), bg, wg, kTagJitter, k0, k1, r);
        const uint32_t r0 = r[0], r1 = r[1], r2 = r[2], r3 = r[3];
        const bool odd = (dim & 1) != 0;
        vx[v] = vary_one(vx[0], odd ? r2 : r0, odd ? r3 : r1);
      }
    }
    iter = 0;
    evals = 0;
    bx = __builtin_nan("");
    best_cost = __builtin_inf();
    have_best = false;
    fin_status = 2;
    fr = 0.0;
    // first quantum shortened by a per-chain amount: the chains that start together do not all park together
    q_start = a.quantum > 0 ? -(int)((chain * 2654435761u >> 16) % (unsigned)a.quantum) : 0;
    st = ST_INIT0;
  };
  // take a parked chain up again: observations to LDS as for a fresh chain, simplex / costs / best / counters as
  // stored at the iteration boundary (the caller has fenced: the state is the parking group's), centroid and
  // reflection recomputed with the same arithmetic
  auto resume_chain = [&]() {
    const int w = (int)(chain / (unsigned)a.C);
    const int wi = w * a.wstride;
    const size_t wN = (size_t)w * (size_t)N;
    if (gl == 0) {
      const double p_uu0 = a.p_uu[wi];
      wconst[0] = p_uu0;
      wconst[1] = 1.0 - p_uu0;
      wconst[2] = a.eqp[wi];
      wconst[3] = a.eqp_w[wi] * (double)N;
    }
    const uint32_t* idx_row = (a.dmode == 1) ? a.idx + (size_t)chain * (size_t)N : nullptr;
    const size_t dN = (a.dmode == 2) ? (size_t)chain * (size_t)N : wN;
#pragma unroll
    for (int q = 0; q < RR; ++q) {
      const int i = gl + G * q;
      if (i < N)
        dobs[i] = (a.dmode == 1) ? a.pred[wN + i] + a.resid[wN + idx_row[i]] : a.D[dN + i];
    }
    // agent-scope loads (past the caches, as the parking group's stores): no cache invalidation needed
    double* sp = a.state + (size_t)chain * 32;
    auto ld = [&](int i) { return __hip_atomic_load(sp + i, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT); };
#pragma unroll
    for (int k = 0; k < 5; ++k) {
      vx[k] = ld(4 * k + dim);
      c[k] = ld(20 + k);
    }
    bx = ld(25 + dim);
    best_cost = ld(29);
    const long long ie = __double_as_longlong(ld(30));
    iter = (int)(ie & 0xffffffffll);
    evals = (int)(ie >> 32);
    have_best = __double_as_longlong(ld(31)) != 0;
    fin_status = 2;
    fr = 0.0;
    double acc = vx[0];
    acc = acc + vx[1];
    acc = acc + vx[2];
    acc = acc + vx[3];
    x0 = acc * (1.0 / 4.0);
    xr = x0 + (x0 - vx[4]) * 1.0;
    xc = xr;
    q_start = evals;
    st = ST_REFLECT;
  };
  if (chain < total) setup_chain();
  __syncthreads();

  // ---- one cost evaluation: the resident branch of abn_fit_kernel's, statement for statement (keep the two in
  // step; tests/test_gpu_parity.py::test_persistent_refill_kernel_is_schedule_independent compares their outputs)
  auto eval = [&](double xd) -> double {
    const double al = dpp_mov<kDppQuadBcast0>(xd), be = dpp_mov<kDppQuadBcast1>(xd);
    const double wt = dpp_mov<kDppQuadBcast2>(xd), ic = dpp_mov<kDppQuadBcast3>(xd);
    const double p_mm = wconst[1];
    const double sv0 = wconst[0], sv1 = wt * p_mm, sv2 = (1.0 - wt) * p_mm;  // src/divergence.rs:44
    const double puu = p_uu_est(al, be);                     // src/divergence.rs:92
    const double dq = puu - wconst[2];
    const double pen = wconst[3] * (dq * dq);                // src/structs.rs:210-212
    uint32_t tr = tri_s[gl < K ? gl : 0];                    // first triple early, later ones a round ahead
    if constexpr (kMatrixFma && G == kWave) build_power_table_mx<G>(al, be, a.T, lds, a.chain_stride, dtab, lane);  // P1 + P2
    else build_power_table<G>(genmatrix(al, be), a.T, TP, pw, gl);
    __syncthreads();
#pragma unroll 1
    for (int t = gl; t < K; t += G) {                        // P3
      const uint32_t trn = tri_s[t + G < K ? t + G : 0];
      dtab[t] = triple_dt(tr, pw, TP, sv0, sv1, sv2);
      tr = trn;
    }
    __syncthreads();
    double acc = 0.0;                                        // P4
    constexpr int RC = RR < 8 ? RR : 8;
    constexpr int NA = kWave / G;                            // canonical tree: accumulators gl + G j held by this lane
    double av[NA];
#pragma unroll
    for (int j = 0; j < NA; ++j) av[j] = 0.0;
#pragma unroll
    for (int q0 = 0; q0 < RR; q0 += RC) {
      double dv[RC], tv[RC], x[RC];
#pragma unroll
      for (int q = 0; q < RC; ++q) {
        const int i = gl + G * (q0 + q);
        dv[q] = dobs[i < N ? i : N - 1];
        tv[q] = dtab[(tidp[(q0 + q) / 2] >> (16 * ((q0 + q) & 1))) & 0xffffu];
      }
#pragma unroll
      for (int q = 0; q < RC; ++q) {
        const double r = dv[q] - ic - tv[q];
        const double term = r * r + pen;
        x[q] = ((gl + G * (q0 + q)) < N) ? term : 0.0;
      }
      if (canon) {
#pragma unroll
        for (int q = 0; q < RC; ++q) av[(q0 + q) % NA] = av[(q0 + q) % NA] + x[q];
      } else {
#pragma unroll
        for (int q = 0; q < RC; ++q) acc = acc + x[q];
      }
    }
    acc = canon ? tree64_finish<G>(av) : group_sum_dpp<G>(acc);  // P5
    __syncthreads();
    return acc;
  };

  // IterState::update() + terminate_internal() + the head of next_iter (centroid, reflection); as abn_fit_kernel
  auto begin_iteration = [&](bool count_iter) {
    const double c_best = c[0];
    if (c_best < best_cost || (__builtin_isinf(c_best) && __builtin_isinf(best_cost) &&
                               (__builtin_signbit(c_best) == __builtin_signbit(best_cost)))) {
      bx = vx[0];
      best_cost = c_best;
      have_best = true;
    }
    if (count_iter) ++iter;
    bool converged = false;
    if (!((c[4] - c[0]) > a.gap_tol)) {
      double sum = 0.0;
#pragma unroll
      for (int k = 0; k < 5; ++k) sum = sum + c[k];
      const double c0 = sum / 5.0;
      double ss = 0.0;
#pragma unroll
      for (int k = 0; k < 5; ++k) ss = ss + (c[k] - c0) * (c[k] - c0);
      const double sd = __builtin_sqrt(1.0 / (5.0 - 1.0) * ss);
      converged = sd < a.sd_tol;
    }
    int status = -1;
    if (converged) status = 0;
    else if (iter >= a.max_iters) status = 1;
    else if (best_cost <= -__builtin_inf()) status = 3;
    // time slicing: the quantum is used up and somebody is waiting (an unstarted or a parked chain) -> park.
    // The counters are read once per quantum; a stale answer costs at most a park that is taken up again at once.
    // The counters change under the group's feet (other CUs): ONE lane reads them and the group takes its verdict —
    // lanes that each read for themselves could disagree at a 0/1 boundary and tear the chain apart.  (evals, q_start
    // and status are replicated in the group, so all its lanes are here together and the leader lane is active.)
    bool suspend = false;
    if (a.quantum > 0 && status < 0 && evals - q_start >= a.quantum) {
      int verdict = 0;
      if (gl == 0) {
        const unsigned fq = __hip_atomic_load(a.queue, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
        const int av = __hip_atomic_load(reinterpret_cast<int*>(pht) + kParkAvail, __ATOMIC_RELAXED,
                                         __HIP_MEMORY_SCOPE_AGENT);
        const unsigned tl = __hip_atomic_load(pht + kParkTail, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
        verdict = ((gridDim.x * NG + fq < total || av > 0) && tl + gridDim.x * NG / kParkShards + NG < a.park_cap) ? 1 : 0;
      }
      suspend = __builtin_amdgcn_ds_bpermute(4 * (g * G), verdict) != 0;  // the group leader's reading
      q_start = evals;
    }
    const bool done = status >= 0 || suspend;
    fin_status = (status >= 0) ? (have_best ? status : 2) : (suspend ? kFitSuspended : fin_status);
    double acc = vx[0];
    acc = acc + vx[1];
    acc = acc + vx[2];
    acc = acc + vx[3];
    x0 = acc * (1.0 / 4.0);
    xr = x0 + (x0 - vx[4]) * 1.0;
    xc = xr;
    st = done ? ST_DONE : ST_REFLECT;
  };

  int prio_cur = 0;
  while (__ballot(st != ST_IDLE) != 0ull) {
    if (a.prio_mode != 0) {  // wave priority by the age of the wavefront's oldest running chain (scalar code)
      const int e = st != ST_IDLE ? evals : 0;
      int m = 0;
#pragma unroll
      for (int j = 0; j < NG; ++j) {
        const int ej = __builtin_amdgcn_readlane(e, j * G);
        m = ej > m ? ej : m;
      }
      int lvl = (m >= a.prio_t[0] ? 1 : 0) + (m >= a.prio_t[1] ? 1 : 0) + (m >= a.prio_t[2] ? 1 : 0);
      if (a.prio_mode == 2) lvl = 3 - lvl;
      if (lvl != prio_cur) {
        prio_cur = lvl;
        if (lvl == 0) __builtin_amdgcn_s_setprio(0);
        else if (lvl == 1) __builtin_amdgcn_s_setprio(1);
        else if (lvl == 2) __builtin_amdgcn_s_setprio(2);
        else __builtin_amdgcn_s_setprio(3);
      }
    }
    const bool in_init = st < ST_REFLECT;                         // Solver::init: start vertex st - ST_INIT0
    const double f = eval(in_init ? vx[0] : xc);
    // ---- decisions of NelderMead::next_iter as predicates (inert for groups in init or idle)
    const bool is_ref = st == ST_REFLECT, is_exp = st == ST_EXPAND, is_con = st == ST_CONTRACT;
    const bool active = is_ref || is_exp || is_con;
    const bool acc_r = is_ref && (f < c[3]) && (f >= c[0]);
    const bool go_exp = is_ref && !acc_r && (f < c[0]);
    const bool go_con = is_ref && !acc_r && !go_exp && (f >= c[3]);
    const bool nan_ref = is_ref && !acc_r && !go_exp && !go_con;
    const bool keep_r = is_exp && !(f < fr);
    const bool acc_c = is_con && (f < c[4]);
    const bool rej_c = is_con && !acc_c;
    const bool do_insert = acc_r || is_exp || acc_c;
    const bool start_shrink = nan_ref || (rej_c && a.shrink_variant != 0);
    const bool do_begin = do_insert || (rej_c && a.shrink_variant == 0);
    evals += active ? 1 : 0;
    if (rej_c && a.shrink_variant == 0 && a.no_skip == 0) {  // fixed point: finish the chain (FitArgs::no_skip)
      const int rest = a.max_iters - iter - 1;
      evals += 2 * rest;
      iter += rest;
      if (a.skipped && gl == 0 && rest > 0) atomicAdd(a.skipped, 2ull * (unsigned long long)rest);
    }
    const double xi = keep_r ? xr : xc;
    const double fi = keep_r ? fr : f;
    fr = is_ref ? f : fr;
    const double x_e = x0 + (xr - x0) * 2.0;
    const double x_c = x0 + (vx[4] - x0) * 0.5;
    xc = go_exp ? x_e : (go_con ? x_c : xc);
    st = go_exp ? ST_EXPAND : (go_con ? ST_CONTRACT : st);
    if (do_insert) {
      c[4] = fi;
      vx[4] = xi;
      insert_tail<4>(c, vx);
    }
    if (do_begin) begin_iteration(true);
    // ---- Solver::init: costs in input order; the arrays rotate so that no register array is indexed at run time
    if (in_init) {
      const double tv = vx[0];
#pragma unroll
      for (int q = 0; q < 4; ++q) {
        c[q] = c[q + 1];
        vx[q] = vx[q + 1];
      }
      c[4] = f;
      vx[4] = tv;
      st = st + 1;
      if (st == ST_REFLECT) {  // all five: stable sort, first termination check
        evals = 5;
        sort5(c, vx);
        begin_iteration(false);
      }
    }
    // ---- NelderMead::shrink (rare; the other groups idle)
    if (__ballot(start_shrink) != 0ull) {
#pragma unroll 1
      for (int k = 1; k < 5; ++k) {
        const double nv = vx[0] + (vx[1] - vx[0]) * 0.5;
        const double fk = eval(start_shrink ? nv : xc);
        if (start_shrink) {
          ++evals;
#pragma unroll
          for (int q = 1; q < 4; ++q) {
            c[q] = c[q + 1];
            vx[q] = vx[q + 1];
          }
          c[4] = fk;
          vx[4] = nv;
        }
      }
      if (start_shrink) {
        sort5(c, vx);
        begin_iteration(true);
      }
    }
    // ---- finished fits: results in fit order, then the next chain from the queue
    if (__ballot(st == ST_DONE) != 0ull) {
      const bool fin = st == ST_DONE;
      const double b0 = dpp_mov<kDppQuadBcast0>(bx), b1 = dpp_mov<kDppQuadBcast1>(bx);
      unsigned nxt = 0xffffffffu;
      const bool parking = fin && fin_status == kFitSuspended;
      if (parking) {  // time slicing: the chain's state (32 doubles), stored past the caches (agent scope): the group
        double* sp = a.state + (size_t)chain * 32;  // that takes the chain up again may sit on another XCD
        auto sd = [&](int i, double v) { __hip_atomic_store(sp + i, v, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT); };
        if (gl < 4) {
#pragma unroll
          for (int k = 0; k < 5; ++k) sd(4 * k + gl, vx[k]);
          sd(25 + gl, bx);
        }
        if (gl == 0) {
#pragma unroll
          for (int k = 0; k < 5; ++k) sd(20 + k, c[k]);
          sd(29, best_cost);
          sd(30, __longlong_as_double((long long)(unsigned)iter | ((long long)evals << 32)));
          sd(31, __longlong_as_double(have_best ? 1ll : 0ll));
        }
      } else if (fin) {
        if (gl < 4) a.best[(size_t)chain * 4 + gl] = bx;
        if (gl == 0) {
          if (a.slice_status) atomicAdd(a.slice_status + 1, 1u);  // fits finished (no return value: nobody waits for it)
          FitInfoDev fo;
          fo.best_cost = best_cost;
          fo.iters = iter;
          fo.evals = evals;
          fo.status = fin_status;
          fo.lanes = a.tree;  // reduction-tree code (oracle: `lanes`)
          a.info[chain] = fo;
        }
        if (a.raw) {
          double* ro = a.raw + (size_t)chain * 7;
          if (gl < 4) ro[gl] = bx;
          if (gl == 4) ro[4] = est_mm(b0, b1);
          if (gl == 5) ro[5] = est_um(b0, b1);
          if (gl == 6) ro[6] = p_uu_est(b0, b1);
        }
      }
      if (a.quantum > 0) {
        // Publish the parked chains of this wavefront.  The state stores above are write-through (agent scope, sc1) and
        // the vector-memory counter of gfx9 retires in order, so once vmcnt reaches 0 every one of them has been
        // acknowledged past this XCD's L2 — only then may the FIFO entry (another sc1 store) become visible to a group on
        // another XCD.  The wait is EXPLICIT: a workgroup-scope release fence emits no instruction here (one wavefront
        // per workgroup), and an agent-scope fence would write back and invalidate the XCD's whole L2 at every park
        // (measured: a 5 ms launch took 2 s).  tests/test_isa_checks.py greps the emitted ISA for this wait between the
        // state stores and the entry store.  All of this wavefront's entries are out before any of its groups looks for
        // one (no group can wait for an entry of its own wavefront).
        if (__ballot(parking) != 0ull) asm volatile("s_waitcnt vmcnt(0) ; abn: parked state written through" ::: "memory");
        if (parking && gl == 0) {
          const unsigned pos = atomicAdd(pht + kParkTail, 1u);
#ifdef ABN_MEASUREMENT_KNOBS
          if (!(a.drop_entry != 0 && pos == 0 && (blockIdx.x & (kParkShards - 1)) == 0))
#endif
          __hip_atomic_store(pk + pos, (int)chain, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
          atomicAdd(reinterpret_cast<int*>(pht) + kParkAvail, 1);
        }
      }
      bool take_parked = false;
      if (fin && gl == 0) {
        if (!fresh_done) {
          const unsigned f = gridDim.x * NG + atomicAdd(a.queue, 1u);
          if (f < total) nxt = f;
          else fresh_done = true;
        }
        if (nxt == 0xffffffffu && a.quantum > 0) {
          // oldest parked chain, if any: claim a credit first (given back if there was none), then a ticket — a
          // ticket is only ever taken against a published entry, so none is lost and nobody loops
          int* avail = reinterpret_cast<int*>(pht) + kParkAvail;
          if (atomicSub(avail, 1) > 0) {
            const unsigned h = atomicAdd(pht + kParkHead, 1u);
            // entries are published in any order: the one of this ticket may be a few instructions away (its writer
            // is a running wavefront past its reservation).  Bounded all the same: a lost entry must neither hang the
            // launch nor abort the process (the C-ABI never crashes) — the error word is set, this group goes idle and
            // abn_plan_download reports ABN_ERR_HIP because the chain's fit was never written.
            int cpk;
            unsigned spins = 0;
            while ((cpk = __hip_atomic_load(pk + h, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT)) < 0 && spins < (1u << 24)) {
              __builtin_amdgcn_s_sleep(1);
              ++spins;
            }
            if (cpk >= 0) {
              nxt = (unsigned)cpk;
              take_parked = true;
            } else if (a.slice_status) {
              atomicOr(a.slice_status, kSliceErrLostEntry);
            }
          } else {
            atomicAdd(avail, 1);
          }
        }
      }
      nxt = (unsigned)__builtin_amdgcn_ds_bpermute(4 * (g * G), (int)nxt);  // the group leader's draw
      take_parked = __builtin_amdgcn_ds_bpermute(4 * (g * G), (int)take_parked) != 0;
      fresh_done = __builtin_amdgcn_ds_bpermute(4 * (g * G), (int)fresh_done) != 0;
      if (fin) {
        st = ST_IDLE;
        if (nxt < total) {
          chain = nxt;
          if (take_parked) resume_chain();
          else setup_chain();
        }
      }
      __syncthreads();
    }
  }
}

// ------------------------------------------------------------------------------------------------
// Speculative fit kernel for the latency-bound case (few chains: the S starts of phase A, a few hundred bootstraps).
// A Nelder-Mead iteration evaluates the reflection and then, depending on its cost, the expansion OR the
// contraction point (argmin next_iter) — two dependent evaluations on ~70 % of the iterations.  All three
// candidates are known before the first cost: x_r = x0 + (x0 - x_w), x_e = x0 + 2 (x_r - x0),
// x_c = x0 + (x_w - x0)/2.  Here a chain owns a workgroup of FOUR wavefronts, one per SIMD of a CU:
//   * wavefronts 0..2 evaluate one candidate each with the G = 64 tree and exchange the three costs through
//     LDS.  They hold no optimiser state: behind the exchange barrier they read the sorted costs of the previous
//     update and a done flag (published by the keeper), take the reference's decision — which point is accepted
//     and at which rank it sorts in: one of ten outcomes — and pick their next candidate, its generation-matrix
//     elements (matrix-instruction layout) and its penalty term out of tables.
//   * wavefront 3, the "keeper", alone holds the simplex (dimension per lane), costs, best vertex and counters.
//     While the others evaluate the candidates of iteration i it works out, for each of the ten outcomes (one per
//     quad of lanes), the three candidates of iteration i+1 and what depends on their (alpha, beta) only, and
//     publishes the costs / done flag of the update it made after iteration i-1.
// The done flag reaches the evaluation wavefronts one evaluation late (a finished fit costs one surplus evaluation,
// never counted); every wavefront derives its control flow from the same published numbers, so they reach the
// same barriers.  Results, iteration and evaluation counts (only evaluations the reference would have made are
// counted) are bit-identical to abn_fit_kernel<64,*>.
// Resident mode only (N <= 64*RMAX).  LDS: 3 x chain_stride doubles + kSpecCommDoubles.
// ------------------------------------------------------------------------------------------------
__device__ __forceinline__ void wave_lds_fence() {  // orders this wavefront's LDS writes before its reads
  __builtin_amdgcn_fence(__ATOMIC_RELEASE, "workgroup");
  __builtin_amdgcn_wave_barrier();
  __builtin_amdgcn_fence(__ATOMIC_ACQUIRE, "workgroup");
}

constexpr int kSpecOutcomes = 10;                                   // r@0..3, e@0, c@0..4
constexpr int kSpecTabDoubles = kSpecOutcomes * 12;                 // [outcome][candidate r/e/c][dimension]
constexpr int kSpecPreDoubles = kSpecOutcomes * 3 * 12;             // [outcome][candidate][G (9), penalty, 0.0, pad]
constexpr int kSpecCommDoubles = 8 + 2 * kSpecTabDoubles + 16 + 2 * kSpecPreDoubles + 16;  // cost exchange, two candidate
                                                 // tables, shrink points, two tables of prepared inputs, two control blocks

// STRICT (abn_options.strict_order): the evaluation wavefronts sum the residuals serially in row order (terms to LDS, N
// more doubles per wavefront, then serial_sum_lds) — the reference's order, the oracle's lanes = 1.
template <int RMAX, bool STRICT = false>
__global__ __launch_bounds__(4 * kWave, RMAX <= 2 ? 3 : 2) void abn_fit_spec_kernel(const FitArgs a) {
  constexpr int G = kWave;
  extern __shared__ __align__(16) double lds[];
  const int wv = threadIdx.x >> 6;      // 0: reflection, 1: expansion, 2: contraction, 3: keeper
  const int gl = threadIdx.x & 63;
  const int dim = gl & 3;
  const bool keeper = wv == 3;
  const long long chain = blockIdx.x;   // grid = W*C exactly
  const int w = (int)(chain / a.C);
  const int N = a.N, K = a.K, TP = a.TP;

  double* pw = lds + (size_t)(keeper ? 0 : wv) * a.chain_stride;  // the keeper never touches its alias
  double* dtab = pw + kPw * TP;
  double* wconst = dtab + ((K + 1) & ~1);
  double* dobs = wconst + 4;
  double* xch = lds + (size_t)3 * a.chain_stride;  // two buffers of 3 costs (+ pad)
  double* tab = xch + 8;                           // two candidate tables
  double* pts = tab + 2 * kSpecTabDoubles;         // NelderMead::shrink: the four moved vertices
  double* gtab = pts + 16;                         // per candidate: generation matrix, penalty term, a zero

  const int wi = w * a.wstride;
  const size_t wN = (size_t)w * (size_t)N;
  uint32_t triv[RMAX], tidp[(RMAX + 1) / 2];
#pragma unroll
  for (int q = 0; q < (RMAX + 1) / 2; ++q) tidp[q] = 0u;
#pragma unroll
  for (int q = 0; q < RMAX; ++q) triv[q] = 0u;
  // keeper state: the simplex, this lane's dimension of the five vertices in rank order
  double vx[5] = {0.0, 0.0, 0.0, 0.0, 0.0};
  if (!keeper) {
    if (gl == 0) {
      const double p_uu0 = a.p_uu[wi];
      wconst[0] = p_uu0;
      wconst[1] = 1.0 - p_uu0;
      wconst[2] = a.eqp[wi];
      wconst[3] = a.eqp_w[wi] * (double)N;
    }
    const uint32_t* idx_row = (a.dmode == 1) ? a.idx + (size_t)chain * (size_t)N : nullptr;
#pragma unroll
    for (int q = 0; q < RMAX; ++q) {
      const int i = gl + G * q;
      triv[q] = (i < K) ? a.tri[i] : 0u;
      if (i < N) {
        tidp[q / 2] |= (uint32_t)a.tid[i] << (16 * (q & 1));
        dobs[i] = (a.dmode == 1) ? a.pred[wN + i] + a.resid[wN + idx_row[i]]  // src/boot_model.rs:50-54
                                 : a.D[wN + i];
      }
    }
  } else {
    // start simplex (starts: given; bootstraps: [params, vary() x4], src/boot_model.rs:69-75), handed to the
    // evaluation wavefronts through the (still unused) candidate table
    if (a.smode == 0) {
      const double* s0 = a.simplex0 + (size_t)chain * 20;
#pragma unroll
      for (int k = 0; k < 5; ++k) vx[k] = s0[4 * k + dim];
    } else {
      const int j = (int)(chain - (long long)w * a.C);
      const uint32_t k0 = (uint32_t)a.seed, k1 = (uint32_t)(a.seed >> 32);
      const uint32_t wg = a.wid ? a.wid[w] : a.window_offset + (uint32_t)w, bg = a.boot_offset + (uint32_t)j;
      vx[0] = a.model[4 * w + dim];
#pragma unroll
      for (int v = 1; v < 5; ++v) {
        uint32_t r[4];
        philox4x32_10((uint32_t)(v - 1) * 2u + (uint32_t)(dim >> 1), bg, wg, kTagJitter, k0, k1, r);
        const uint32_t r0 = r[0], r1 = r[1], r2 = r[2], r3 = r[3];
        const bool odd = (dim & 1) != 0;
        vx[v] = vary_one(vx[0], odd ? r2 : r0, odd ? r3 : r1);
      }
    }
    if (gl < 4) {
#pragma unroll
      for (int k = 0; k < 5; ++k) tab[4 * k + dim] = vx[k];
    }
  }
  __syncthreads();

#ifdef ABN_STAMPS
  unsigned long long seg[8] = {0, 0, 0, 0, 0, 0, 0, 0}, tprev = 0;
  asm volatile("s_memtime %0\n\ts_waitcnt lgkmcnt(0)" : "=s"(tprev)::"memory");
#endif
  // Inputs of the next evaluation that depend on (alpha, beta) only and that the keeper prepares next to the
  // candidate (kMatrixFma builds): this lane's elements of G^T and G in the matrix-instruction layout, the
  // equilibrium penalty term.  `pre` = they are valid for the candidate being evaluated.
  double preA = 0.0, preB = 0.0, prePen = 0.0;
  const int mx_x = gl & 3, mx_y = gl >> 4;
  const bool mx_in = (mx_x < 3) && (mx_y < 3) && (((gl >> 2) & 3) == 0);
  const int preA_idx = mx_in ? 3 * mx_y + mx_x : 10, preB_idx = mx_in ? 3 * mx_x + mx_y : 10;  // [10] holds 0.0
  auto eval = [&](double xd, bool pre) -> double {
    ABN_STAMP(6);  // control flow + candidate fetch since the exchange
    const double wt = dpp_mov<kDppQuadBcast2>(xd), ic = dpp_mov<kDppQuadBcast3>(xd);
    const double p_mm = wconst[1];
    const double sv0 = wconst[0], sv1 = wt * p_mm, sv2 = (1.0 - wt) * p_mm;
    double pen;
    if (kMatrixFma && pre) {
      pen = prePen;
      ABN_STAMP(0);
      build_power_table_mx_pre(preA, preB, a.T, pw, dtab, gl);
    } else {
      const double al = dpp_mov<kDppQuadBcast0>(xd), be = dpp_mov<kDppQuadBcast1>(xd);
      const double puu = p_uu_est(al, be);
      const double dq = puu - wconst[2];
      pen = wconst[3] * (dq * dq);
      ABN_STAMP(0);
      if constexpr (kMatrixFma) build_power_table_mx<G>(al, be, a.T, pw, 0, dtab, gl);  // one chain per wavefront: block 0
      else build_power_table<G>(genmatrix(al, be), a.T, TP, pw, gl);
    }
    wave_lds_fence();
    ABN_STAMP(1);
#pragma unroll
    for (int q = 0; q < RMAX; ++q) {
      const int t = gl + G * q;
      if (t < K) dtab[t] = triple_dt(triv[q], pw, TP, sv0, sv1, sv2);
      __builtin_amdgcn_sched_barrier(0);
    }
    wave_lds_fence();
    ABN_STAMP(2);
    double acc = 0.0;
    double dv[RMAX], tv[RMAX];
#pragma unroll
    for (int q = 0; q < RMAX; ++q) {
      const bool in = (gl + G * q) < N;
      dv[q] = in ? dobs[gl + G * q] : 0.0;
      tv[q] = in ? dtab[(tidp[q / 2] >> (16 * (q & 1))) & 0xffffu] : 0.0;
    }
    if constexpr (STRICT) {   // `square_sum += ...` in row order, src/structs.rs:206-213
      if (RMAX == 1 && N <= 16) {   // one row per lane, at most 16 rows: the sum through lane reads (wavefront-uniform branch)
        const double r = dv[0] - ic - tv[0];
        acc = serial_sum_lanes16(gl < N ? r * r + pen : 0.0, N);
      } else {
        double* term = dobs + ((N + 1) & ~1);
#pragma unroll
        for (int q = 0; q < RMAX; ++q) {
          if ((gl + G * q) < N) {
            const double r = dv[q] - ic - tv[q];
            term[gl + G * q] = r * r + pen;
          }
        }
        wave_lds_fence();
        acc = serial_sum_lds(term, N, 0.0);
      }
    } else {
#pragma unroll
      for (int q = 0; q < RMAX; ++q) {
        if ((gl + G * q) < N) {
          const double r = dv[q] - ic - tv[q];
          acc = acc + (r * r + pen);
        }
      }
      // P5: the canonical tree (without strict order this kernel runs under auto options only: FitArgs::tree == kTreeCanon)
      const double one[1] = {acc};
      acc = tree64_finish<G>(one);
    }
    wave_lds_fence();
    ABN_STAMP(3);
    return acc;
  };

  // exchange: every evaluation wavefront publishes its cost, one workgroup barrier, everybody reads all three.
  // The barrier also hands the keeper's LDS writes (candidate table, shrink points) to the others.
  int phase = 0;
  auto exchange = [&](double f, double& f0, double& f1, double& f2) {
    double* buf = xch + 4 * phase;
    if (!keeper && gl == 0) buf[wv] = f;
    __syncthreads();
    f0 = buf[0];
    f1 = buf[1];
    f2 = buf[2];
    phase ^= 1;  // the other buffer next time: no second barrier needed
    ABN_STAMP(4);
  };

  // ---- control state (the keeper's; the evaluation wavefronts hold a published copy of the costs in c[])
  double c[5], best_cost = __builtin_inf();
  bool have_best = false;
  int iter = 0;
  // IterState::update() + terminate_internal(): -1 = go on, else the ABN_FIT_* status.  `improved`: the best
  // vertex is the new best_param (the keeper copies it)
  auto ctl_begin = [&](bool count_iter, bool& improved) -> int {
    const double c_best = c[0];
    improved = c_best < best_cost || (__builtin_isinf(c_best) && __builtin_isinf(best_cost) &&
                                      (__builtin_signbit(c_best) == __builtin_signbit(best_cost)));
    best_cost = improved ? c_best : best_cost;
    have_best = have_best || improved;
    if (count_iter) ++iter;
    bool converged = false;
    if (!((c[4] - c[0]) > a.gap_tol)) {
      double sum = 0.0;
#pragma unroll
      for (int k = 0; k < 5; ++k) sum = sum + c[k];
      const double c0 = sum / 5.0;
      double ss = 0.0;
#pragma unroll
      for (int k = 0; k < 5; ++k) ss = ss + (c[k] - c0) * (c[k] - c0);
      const double sd = __builtin_sqrt(1.0 / (5.0 - 1.0) * ss);
      converged = sd < a.sd_tol;
    }
    int status = -1;
    if (converged) status = 0;
    else if (iter >= a.max_iters) status = 1;
    else if (best_cost <= -__builtin_inf()) status = 3;
    return status;
  };

  // ---- keeper state: the simplex, this lane's dimension of the five vertices in rank order, the three
  // candidates of the running iteration, best_param, evaluation count
  double xr = 0.0, x_e = 0.0, x_c = 0.0, bx = __builtin_nan("");
  int evals = 0;
  // the keeper's lane group: quad q < 10 works on outcome q = (accepted point, rank): r@0..3, e@0, c@0..4
  const int oq = gl >> 2;
  const int o_kind = oq < 4 ? 0 : (oq == 4 ? 1 : 2);
  const int o_rank = oq < 4 ? oq : (oq == 4 ? 0 : oq - 5);

  // keeper: generation matrix and penalty term of the three candidates (r_, e_, c_: this lane's dimension) of
  // outcome slot `oq` — lane (quad, dimension t < 3) works for candidate t; same functions as the evaluation
  // wavefronts would call, so the same bits
  auto emit_pre = [&](double r_, double e_, double c_, int parity) {
    const double ar = dpp_mov<kDppQuadBcast0>(r_), br = dpp_mov<kDppQuadBcast1>(r_);
    const double ae = dpp_mov<kDppQuadBcast0>(e_), be = dpp_mov<kDppQuadBcast1>(e_);
    const double ac = dpp_mov<kDppQuadBcast0>(c_), bc = dpp_mov<kDppQuadBcast1>(c_);
    const double al_t = dim == 0 ? ar : (dim == 1 ? ae : ac);
    const double be_t = dim == 0 ? br : (dim == 1 ? be : bc);
    const Gen Gt = genmatrix(al_t, be_t);
    const double puu = p_uu_est(al_t, be_t);
    const double dq = puu - wconst[2];
    const double pen = wconst[3] * (dq * dq);
    if (oq < kSpecOutcomes && dim < 3) {
      double* g = gtab + parity * kSpecPreDoubles + (oq * 3 + dim) * 12;
      g[0] = Gt.g0;
      g[1] = Gt.g1;
      g[2] = Gt.g2;
      g[3] = Gt.g3;
      g[4] = Gt.g4;
      g[5] = Gt.g5;
      g[6] = Gt.g6;
      g[7] = Gt.g7;
      g[8] = Gt.g8;
      g[9] = pen;
      g[10] = 0.0;
    }
  };
  // evaluation wavefronts: the prepared inputs of candidate (outcome o, this wavefront) next to the candidate itself
  auto fetch_pre = [&](int o, int parity) {
    const double* g = gtab + parity * kSpecPreDoubles + (o * 3 + wv) * 12;
    preA = g[preA_idx];
    preB = g[preB_idx];
    prePen = g[9];
  };

  // ---- The keeper alone keeps the optimiser's state.  What the evaluation wavefronts need to follow the control
  // flow it publishes in `ctl` (two buffers): the five sorted costs BEFORE the running iteration (their decision
  // needs c0, c3, c4 and the rank of the accepted cost) and a `done` flag.  The keeper writes ctl[cb] while the
  // others evaluate; they read it behind the next exchange barrier, then everybody flips cb.  The flag therefore
  // reaches them one evaluation late: a finished fit costs one surplus evaluation (never counted, never used),
  // every iteration saves the cost insertion and the termination test on the evaluation wavefronts' path.
  double* ctl = gtab + 2 * kSpecPreDoubles;
  int cb = 0;
  auto ctl_write = [&](int status_now) {      // keeper
    if (gl < 5) ctl[8 * cb + gl] = gl == 0 ? c[0] : (gl == 1 ? c[1] : (gl == 2 ? c[2] : (gl == 3 ? c[3] : c[4])));
    if (gl == 5) ctl[8 * cb + 5] = status_now >= 0 ? 1.0 : 0.0;
  };
  bool improved;
  int status = -1;
  int par = 0;
  double f0, f1, f2;
  double cand = 0.0;

  // hand-over of a freshly sorted simplex (after Solver::init and after a shrink): the keeper writes the control
  // block and, unless the fit is finished, the three candidates with their prepared inputs; one barrier; the
  // evaluation wavefronts learn `done` and pick their candidate up.  Returns done.
  auto publish = [&]() -> bool {
    if (keeper) {
      ctl_write(status);
      if (status < 0) {  // centroid (p0 + p1 + p2 + p3) * (1/4), x0 + (x0 - xw) * alpha, x0 + (xr - x0) * gamma, x0 + (xw - x0) * rho
        double acc = vx[0];
        acc = acc + vx[1];
        acc = acc + vx[2];
        acc = acc + vx[3];
        const double x0 = acc * (1.0 / 4.0);
        xr = x0 + (x0 - vx[4]) * 1.0;
        x_e = x0 + (xr - x0) * 2.0;
        x_c = x0 + (vx[4] - x0) * 0.5;
        if (gl < 4) {
          double* t = tab + par * kSpecTabDoubles;
          t[dim] = xr;
          t[4 + dim] = x_e;
          t[8 + dim] = x_c;
        }
        if constexpr (kMatrixFma) emit_pre(xr, x_e, x_c, par);  // every quad writes its slot; slot 0 is read
      }
    }
    __syncthreads();
    bool done = status >= 0;                  // keeper
    if (!keeper) {
      done = ctl[8 * cb + 5] != 0.0;
      if (!done) {
        cand = tab[par * kSpecTabDoubles + 4 * wv + dim];
        if constexpr (kMatrixFma) fetch_pre(0, par);
      }
    }
    par ^= 1;
    return done;
  };

  // ---- Solver::init: the five start costs in input order (3 + 2), stable sort, first termination check
  cand = keeper ? 0.0 : tab[4 * wv + dim];
  exchange(keeper ? 0.0 : eval(cand, false), f0, f1, f2);
  c[0] = f0;
  c[1] = f1;
  c[2] = f2;
  cand = keeper ? 0.0 : tab[4 * (wv == 0 ? 3 : 4) + dim];
  exchange(keeper ? 0.0 : eval(cand, false), f0, f1, f2);
  c[3] = f0;
  c[4] = f1;
  evals = 5;
  if (keeper) {
    sort5(c, vx);
    status = ctl_begin(false, improved);
    if (improved) bx = vx[0];
  }
  bool done = publish();

  while (!done) {
    if (keeper) {
      // the candidates of the NEXT iteration for each way this one can end.  Outcome (A, p): the accepted
      // point A replaces the worst vertex and sorts in at rank p; the new order is v0..v3 with A at p.
      const double A = o_kind == 0 ? xr : (o_kind == 1 ? x_e : x_c);
      const double e0 = o_rank == 0 ? A : vx[0];
      const double e1 = o_rank == 1 ? A : (o_rank < 1 ? vx[0] : vx[1]);
      const double e2 = o_rank == 2 ? A : (o_rank < 2 ? vx[1] : vx[2]);
      const double e3 = o_rank == 3 ? A : (o_rank < 3 ? vx[2] : vx[3]);
      const double xw = o_rank == 4 ? A : vx[3];
      double acc = e0;
      acc = acc + e1;
      acc = acc + e2;
      acc = acc + e3;
      const double x0 = acc * (1.0 / 4.0);
      const double nr = x0 + (x0 - xw) * 1.0;
      const double ne = x0 + (nr - x0) * 2.0;
      const double nc = x0 + (xw - x0) * 0.5;
      if (oq < kSpecOutcomes) {
        double* t = tab + par * kSpecTabDoubles + oq * 12;
        t[dim] = nr;
        t[4 + dim] = ne;
        t[8 + dim] = nc;
      }
      if constexpr (kMatrixFma) emit_pre(nr, ne, nc, par);
    }
    double fr, fe, fc;
    {  // exchange + (evaluation wavefronts) the control block, read in ONE batch of LDS loads behind the barrier:
       // the sorted costs before this iteration, and whether the keeper finished the fit while this (then surplus)
       // evaluation ran
      const double f = keeper ? 0.0 : eval(cand, true);
      double* buf = xch + 4 * phase;
      if (!keeper && gl == 0) buf[wv] = f;
      __syncthreads();
      const double* b = ctl + 8 * cb;
      fr = buf[0];
      fe = buf[1];
      fc = buf[2];
      double k0 = c[0], k1 = c[1], k2 = c[2], k3 = c[3], k4 = c[4], dn = 0.0;
      if (!keeper) {
        k0 = b[0];
        k1 = b[1];
        k2 = b[2];
        k3 = b[3];
        k4 = b[4];
        dn = b[5];
      }
      c[0] = k0;
      c[1] = k1;
      c[2] = k2;
      c[3] = k3;
      c[4] = k4;
      phase ^= 1;
      ABN_STAMP(4);
#ifdef ABN_STAMPS
      ++seg[7];  // iterations seen by this wavefront
#endif
      if (dn != 0.0) break;
    }
    cb ^= 1;
    // ---- NelderMead::next_iter's decision.  Every lane holds the same costs, so the branches are uniform.
    // (Measured alternatives, all slower on a lone wavefront: the costs in scalar registers via v_readfirstlane —
    // persistent: SGPR spills; re-read every iteration: +23 % —, and the flat predicated form of abn_fit_kernel.)
    int kind, which = 0;   // kind 0: a point is accepted, 1: rejected contraction (simplex untouched), 2: shrink
    double fi = fr;
    int spent;             // cost() calls the reference makes in this branch
    if (fr < c[3] && fr >= c[0]) {          // reflection accepted
      kind = 0;
      spent = 1;
    } else if (fr < c[0]) {                 // expansion tried
      kind = 0;
      spent = 2;
      const bool take_e = fe < fr;
      which = take_e ? 1 : 0;
      fi = take_e ? fe : fr;
    } else if (fr >= c[3]) {                // contraction tried
      spent = 2;
      if (fc < c[4]) {
        kind = 0;
        which = 2;
        fi = fc;
      } else {
        kind = a.shrink_variant ? 2 : 1;
      }
    } else {                                // NaN reflection cost
      kind = 2;
      spent = 1;
    }
    evals += spent;
    if (kind == 0) {
      // rank of the accepted point: the stable insertion of insert_tail<4>
      const bool b3 = fi < c[3], b2 = b3 && (fi < c[2]), b1 = b2 && (fi < c[1]), b0 = b1 && (fi < c[0]);
      const int p = 4 - ((b3 ? 1 : 0) + (b2 ? 1 : 0) + (b1 ? 1 : 0) + (b0 ? 1 : 0));
      const int o = which == 0 ? p : (which == 1 ? 4 : 5 + p);
      const double* t = tab + par * kSpecTabDoubles + o * 12;
      if (keeper) {
        c[4] = fi;
        vx[4] = which == 0 ? xr : (which == 1 ? x_e : x_c);
        insert_tail<4>(c, vx);
        status = ctl_begin(true, improved);
        if (improved) bx = vx[0];
        ctl_write(status);
        xr = t[dim];
        x_e = t[4 + dim];
        x_c = t[8 + dim];
      } else {
        cand = t[4 * wv + dim];
        if constexpr (kMatrixFma) fetch_pre(o, par);
        ABN_STAMP(5);  // decision
      }
      par ^= 1;
    } else if (kind == 1) {
      // argmin 0.8.1: a rejected contraction leaves the simplex untouched — for good: every later iteration
      // repeats this one.  no_skip == 0: finish the chain with the counters it would reach (FitArgs::no_skip)
      if (keeper) {
        if (a.no_skip == 0) {
          const int rest = a.max_iters - iter - 1;
          evals += 2 * rest;
          iter += rest;
          if (a.skipped && gl == 0 && rest > 0) atomicAdd(a.skipped, 2ull * (unsigned long long)rest);
        }
        status = ctl_begin(true, improved);
        ctl_write(status);
      }
      par ^= 1;  // same candidates again; the keeper rebuilds the (identical) tables in the other buffers
    } else {
      // NelderMead::shrink (NaN reflection cost, or the textbook variant after a rejected contraction):
      // vertices 1..4 move towards the best by sigma and are re-evaluated in order (3 + 1)
      if (keeper) {
#pragma unroll
        for (int k = 1; k < 5; ++k) {
          vx[k] = vx[0] + (vx[k] - vx[0]) * 0.5;
          if (gl < 4) pts[4 * (k - 1) + dim] = vx[k];
        }
      }
      __syncthreads();
      if (!keeper) cand = pts[4 * wv + dim];
      exchange(keeper ? 0.0 : eval(cand, false), f0, f1, f2);
      c[1] = f0;
      c[2] = f1;
      c[3] = f2;
      if (!keeper) cand = pts[12 + dim];
      exchange(keeper ? 0.0 : eval(cand, false), f0, f1, f2);
      c[4] = f0;
      evals += 4;
      if (keeper) {
        sort5(c, vx);
        status = ctl_begin(true, improved);
        if (improved) bx = vx[0];
      }
      done = publish();
      continue;
    }
    // the keeper found the fit finished: the others are one evaluation ahead — meet them at that exchange
    if (keeper && status >= 0) {
      exchange(0.0, f0, f1, f2);
      break;
    }
  }

#ifdef ABN_STAMPS
  if (a.dbg && chain == 0 && wv == 0 && gl == 0) {
    for (int q = 0; q < 8; ++q) a.dbg[q] = seg[q];
    a.dbg[7] = seg[7];
  }
#endif
  const double b0 = dpp_mov<kDppQuadBcast0>(bx), b1 = dpp_mov<kDppQuadBcast1>(bx);
  if (keeper) {
    if (gl < 4) a.best[(size_t)chain * 4 + gl] = bx;
    if (gl == 0) {
      FitInfoDev fo;
      fo.best_cost = best_cost;
      fo.iters = iter;
      fo.evals = evals;
      fo.status = have_best ? status : 2;
      fo.lanes = a.tree;
      a.info[chain] = fo;
    }
    if (a.raw) {
      double* ro = a.raw + (size_t)chain * 7;
      if (gl < 4) ro[gl] = bx;
      if (gl == 4) ro[4] = est_mm(b0, b1);
      if (gl == 5) ro[5] = est_um(b0, b1);
      if (gl == 6) ro[6] = p_uu_est(b0, b1);
    }
  }
}

// ------------------------------------------------------------------------------------------------
// Selection kernels: src/ab_neutral.rs:83-135.  The pure LSE of each of the
// S fitted models is summed SERIALLY in row order (the reference's `.sum::<f64>()`), the stable
// arg-min taken (lowest start index on ties; NaN never wins), then predicted divergence and residuals
// of the winner written for phase B.  LDS: chain scratch (9*TP + K doubles) + kSelChunk terms.
// ------------------------------------------------------------------------------------------------
constexpr int kSelChunk = 512;

struct SelectArgs {
  const uint32_t* tri;
  const uint16_t* tid;
  int N, K, T, TP;
  const double* p_uu;   // [W]
  const double* D;      // [W*N]
  const double* models; // [W*S*4] fitted start models
  const FitInfoDev* info;  // [W*S]
  int W, S;
  double* lse;          // [W*S]
  double* model;        // [W*4]
  double* pred;         // [W*N]
  double* resid;        // [W*N]
  int32_t* best_start;  // [W]  (-1: no finite fit)
};

// P1-P3 of one model for the whole wavefront: power table and dt[K] into LDS
__device__ __forceinline__ void select_fill_dt(const SelectArgs& a, const double* x, double p_uu0, double* pw,
                                               double* dtab, int lane) {
  const double p_mm = 1.0 - p_uu0;
  const double sv0 = p_uu0, sv1 = x[2] * p_mm, sv2 = (1.0 - x[2]) * p_mm;
  if constexpr (kMatrixFma) build_power_table_mx<kWave>(x[0], x[1], a.T, pw, 0, dtab, lane);
  else build_power_table<kWave>(genmatrix(x[0], x[1]), a.T, a.TP, pw, lane);
  __syncthreads();
  for (int t = lane; t < a.K; t += kWave) dtab[t] = triple_dt(a.tri[t], pw, a.TP, sv0, sv1, sv2);
  __syncthreads();
}

// Step 1: one wavefront per (window, start) — the pure LSE of that fitted model, summed serially in row order
__global__ __launch_bounds__(kWave) void abn_select_lse_kernel(const SelectArgs a) {
  extern __shared__ __align__(16) double lds[];
  const int lane = threadIdx.x;
  const int w = blockIdx.x / a.S, sidx = blockIdx.x - w * a.S;
  const int N = a.N;
  double* pw = lds;
  double* dtab = pw + kPw * a.TP;
  double* term = dtab + ((a.K + 1) & ~1);
  const size_t wN = (size_t)w * (size_t)N;
  double x[4];
#pragma unroll
  for (int d = 0; d < 4; ++d) x[d] = a.models[((size_t)w * a.S + sidx) * 4 + d];
  select_fill_dt(a, x, a.p_uu[w], pw, dtab, lane);
  double lsum = 0.0;
  for (int base = 0; base < N; base += kSelChunk) {
    const int cnt = (N - base) < kSelChunk ? (N - base) : kSelChunk;
    for (int i = lane; i < cnt; i += kWave) {
      const double r = a.D[wN + base + i] - x[3] - dtab[a.tid[base + i]];
      term[i] = r * r;
    }
    __syncthreads();
    for (int i = 0; i < cnt; ++i) lsum = lsum + term[i];
    __syncthreads();
  }
  if (lane == 0) a.lse[(size_t)w * a.S + sidx] = lsum;
}

// Step 2: one wavefront per window — stable arg-min over the starts (lowest index on ties; NaN and non-finite
// fits never win), predicted divergence and residuals of the winner
__global__ __launch_bounds__(kWave) void abn_select_kernel(const SelectArgs a) {
  extern __shared__ __align__(16) double lds[];
  const int lane = threadIdx.x;
  const int w = blockIdx.x;
  const int N = a.N;
  double* pw = lds;
  double* dtab = pw + kPw * a.TP;
  const size_t wN = (size_t)w * (size_t)N;

  // stable arg-min = (smallest LSE, lowest start index among equals): every lane scans the starts lane, lane + 64, ...
  // in increasing order, then the 64 candidates are combined with that same rule (1000 starts: 0.2 ms -> a few us)
  int best = -1;
  double best_lse = __builtin_inf();
  for (int sidx = lane; sidx < a.S; sidx += kWave) {
    const double lsum = a.lse[(size_t)w * a.S + sidx];
    const bool ok = (lsum == lsum) && (a.info[(size_t)w * a.S + sidx].status != 2);
    if (ok && (best < 0 || lsum < best_lse)) {
      best = sidx;
      best_lse = lsum;
    }
  }
#pragma unroll
  for (int off = kWave / 2; off >= 1; off >>= 1) {
    const int ob = __shfl_xor(best, off, kWave);
    const double ol = __shfl_xor(best_lse, off, kWave);
    if (ob >= 0 && (best < 0 || ol < best_lse || (ol == best_lse && ob < best))) {
      best = ob;
      best_lse = ol;
    }
  }
  if (lane == 0) a.best_start[w] = best;
  if (best < 0) {
    for (int i = lane; i < N; i += kWave) {
      a.pred[wN + i] = __builtin_nan("");
      a.resid[wN + i] = __builtin_nan("");
    }
    if (lane < 4) a.model[4 * w + lane] = __builtin_nan("");
    return;
  }
  double x[4];
#pragma unroll
  for (int d = 0; d < 4; ++d) x[d] = a.models[((size_t)w * a.S + best) * 4 + d];
  select_fill_dt(a, x, a.p_uu[w], pw, dtab, lane);
  for (int i = lane; i < N; i += kWave) {
    const double p = x[3] + dtab[a.tid[i]];      // src/ab_neutral.rs:123-129
    a.pred[wN + i] = p;
    a.resid[wN + i] = a.D[wN + i] - p;           // src/ab_neutral.rs:131-135
  }
  if (lane < 4) a.model[4 * w + lane] = x[lane];
}

// ------------------------------------------------------------------------------------------------
// Cost kernel (abn_cost_batch): one group of G lanes per candidate, any N (rows streamed).
// strict = 1 (G must be 64): serial row-order accumulation, the reference's order exactly.
// ------------------------------------------------------------------------------------------------
struct CostArgs {
  const uint32_t* tri;
  const uint16_t* tid;
  int N, K, T, TP;
  int chain_stride;
  double p_uu0, eqp, eqp_w;
  const double* D;            // [N] (dmode 0)
  const double* pred;         // [N]
  const double* resid;        // [N]
  const uint32_t* idx;        // [n_boot_rows * N]
  const uint32_t* cand_to_boot;  // [M] or null (identity)
  int dmode;
  const double* cand;         // [M*4]
  long long M;
  int strict;
  int tree;                   // kTreeCanon or G accumulators (FitArgs::tree)
  double* cost;               // [M]
  double* dt;                 // nullable [M*N]
  double* puu;                // nullable [M]
};

template <int G>
__global__ __launch_bounds__(kWave) void abn_cost_kernel(const CostArgs a) {
  constexpr int NG = kWave / G;
  extern __shared__ __align__(16) double lds[];
  const int lane = threadIdx.x;
  const int g = lane / G;
  const int gl = lane - g * G;
  const long long m_raw = (long long)blockIdx.x * NG + g;
  const bool valid = m_raw < a.M;
  const long long m = valid ? m_raw : 0;
  const int N = a.N, K = a.K, TP = a.TP;
  double* pw = lds + (size_t)g * a.chain_stride;
  double* dtab = pw + kPw * TP;
  double* term = lds + (size_t)NG * a.chain_stride;  // strict mode only (G == 64)

  const double al = a.cand[4 * m + 0], be = a.cand[4 * m + 1], wt = a.cand[4 * m + 2], ic = a.cand[4 * m + 3];
  const double p_mm = 1.0 - a.p_uu0;
  const uint32_t* idx_row = nullptr;
  if (a.dmode) {
    const size_t b = a.cand_to_boot ? a.cand_to_boot[m] : (size_t)m;
    idx_row = a.idx + b * (size_t)N;
  }
  const double sv0 = a.p_uu0, sv1 = wt * p_mm, sv2 = (1.0 - wt) * p_mm;
  if constexpr (kMatrixFma && G == kWave) build_power_table_mx<G>(al, be, a.T, lds, a.chain_stride, dtab, lane);
  else build_power_table<G>(genmatrix(al, be), a.T, TP, pw, gl);
  __syncthreads();
  for (int t = gl; t < K; t += G) dtab[t] = triple_dt(a.tri[t], pw, TP, sv0, sv1, sv2);
  __syncthreads();
  const double puu = p_uu_est(al, be);
  const double pen = (a.eqp_w * (double)N) * ((puu - a.eqp) * (puu - a.eqp));
  double result;
  if (a.strict) {
    double ssum = 0.0;  // `square_sum += ...` in row order, src/structs.rs:206-213
    for (int base = 0; base < N; base += kSelChunk) {
      const int cnt = (N - base) < kSelChunk ? (N - base) : kSelChunk;
      for (int i = gl; i < cnt; i += G) {
        const int row = base + i;
        const double d = a.dmode ? a.pred[row] + a.resid[idx_row[row]] : a.D[row];
        const double r = d - ic - dtab[a.tid[row]];
        term[i] = r * r + pen;
      }
      __syncthreads();
      for (int i = 0; i < cnt; ++i) ssum = ssum + term[i];
      __syncthreads();
    }
    result = ssum;
  } else {
    if (a.tree == kTreeCanon) {  // the canonical 64-accumulator tree: this lane holds accumulators gl + G j
      constexpr int NA = kWave / G;
      double av[NA];
#pragma unroll
      for (int j = 0; j < NA; ++j) av[j] = 0.0;
      for (int i0 = gl; i0 < N; i0 += kWave) {  // NA rows at a time keep av[] statically indexed
#pragma unroll
        for (int j = 0; j < NA; ++j) {
          const int i = i0 + G * j;
          if (i < N) {
            const double d = a.dmode ? a.pred[i] + a.resid[idx_row[i]] : a.D[i];
            const double r = d - ic - dtab[a.tid[i]];
            av[j] = av[j] + (r * r + pen);
          }
        }
      }
      result = tree64_finish<G>(av);
    } else {
      double acc = 0.0;
      for (int i = gl; i < N; i += G) {
        const double d = a.dmode ? a.pred[i] + a.resid[idx_row[i]] : a.D[i];
        const double r = d - ic - dtab[a.tid[i]];
        acc = acc + (r * r + pen);
      }
      result = group_sum<G>(acc);
    }
  }
  if (valid) {
    if (gl == 0) {
      a.cost[m] = result;
      if (a.puu) a.puu[m] = puu;
    }
    if (a.dt)
      for (int i = gl; i < N; i += G) a.dt[(size_t)m * N + i] = dtab[a.tid[i]];
  }
}

// src/boot_model.rs:86-91 for a batch of fitted vectors (abn_bootstrap_rows)
__global__ __launch_bounds__(256) void abn_rows_kernel(const double* best, long long n, double* raw) {
  for (long long t = (long long)blockIdx.x * blockDim.x + threadIdx.x; t < n; t += (long long)gridDim.x * blockDim.x) {
    const double al = best[4 * t + 0], be = best[4 * t + 1];
    double* ro = raw + 7 * t;
    ro[0] = al;
    ro[1] = be;
    ro[2] = best[4 * t + 2];
    ro[3] = best[4 * t + 3];
    ro[4] = est_mm(al, be);
    ro[5] = est_um(al, be);
    ro[6] = p_uu_est(al, be);
  }
}

// ------------------------------------------------------------------------------------------------
// Residual bootstrap observations, materialised once per fit for the stream mode (src/boot_model.rs:50-57):
// dstar[(w*B + b)*N + i] = pred[w*N + i] + resid[w*N + idx[(w*B + b)*N + i]].  The index buffer is read
// once, coalesced; the evaluations then stream dstar (8 B per row) instead of re-gathering through the index
// row (4 B per row + an 8-byte random gather that, for tables beyond LDS, is bound by L2->L1 sector traffic).
// ------------------------------------------------------------------------------------------------
__global__ __launch_bounds__(256) void abn_make_dstar_kernel(double* dstar, const double* pred, const double* resid,
                                                             const uint32_t* idx, int N, long long rows_per_window,
                                                             long long total) {
  for (long long t = (long long)blockIdx.x * blockDim.x + threadIdx.x; t < total;
       t += (long long)gridDim.x * blockDim.x) {
    const long long w = t / rows_per_window;
    const int i = (int)(t % N);
    const size_t wN = (size_t)w * (size_t)N;
    dstar[t] = pred[wN + i] + resid[wN + idx[t]];
  }
}

// ------------------------------------------------------------------------------------------------
// Bootstrap index generation (src/boot_model.rs:43-48): idx[(w*B + b)*N + i] in [0,N).  One Philox
// call yields the indices of rows 4q..4q+3 of one bootstrap.
// ------------------------------------------------------------------------------------------------
__global__ __launch_bounds__(256) void abn_gen_idx_kernel(uint32_t* idx, int N, int B, int W, uint64_t seed,
                                                          uint32_t window_offset, uint32_t boot_offset,
                                                          const uint32_t* wid) {
  const int Q = (N + 3) / 4;
  const long long total = (long long)W * B * Q;
  const uint32_t k0 = (uint32_t)seed, k1 = (uint32_t)(seed >> 32);
  for (long long t = (long long)blockIdx.x * blockDim.x + threadIdx.x; t < total;
       t += (long long)gridDim.x * blockDim.x) {
    const int q = (int)(t % Q);
    const long long wb = t / Q;
    const int b = (int)(wb % B);
    const int w = (int)(wb / B);
    uint32_t r[4];
    philox4x32_10((uint32_t)q, boot_offset + (uint32_t)b, wid ? wid[w] : window_offset + (uint32_t)w, kTagIdx, k0, k1, r);
    uint32_t* row = idx + (size_t)wb * (size_t)N;
#pragma unroll
    for (int e = 0; e < 4; ++e) {
      const int i = 4 * q + e;
      if (i < N) row[i] = index_from(r[e], (uint32_t)N);
    }
  }
}

}  // namespace abn
