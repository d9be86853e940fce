// abn_fit_spec_kernel: four wavefronts per chain (three speculative evaluations + the keeper) for latency-bound launches.
#pragma once
#include "abn_common.hpp"

namespace abn {

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
// Registers: 88-135 per lane (the roles' loops are separate, so neither carries the other's state): four workgroups per CU for
// pedigrees of up to two rows per lane — 1024 chains resident on the MI355X —, three beyond.
// RESUME: the launch behind a time-sliced persistent launch that takes its parked tail up (FitArgs::spec_resume) — an
// instantiation of its own so that the start-up path of the ordinary launches carries none of it.
template <int RMAX, bool STRICT = false, bool RESUME = false>
__global__ __launch_bounds__(4 * kWave, RMAX <= 2 ? 4 : 3) void abn_fit_spec_kernel(const FitArgs a) {
  static_assert(!RESUME || !STRICT, "the persistent kernel (whose tail this resumes) has no strict-order form");
  constexpr int G = kWave;
  extern __shared__ __align__(16) double lds[];
  const int wv = __builtin_amdgcn_readfirstlane((int)(threadIdx.x >> 6));  // 0: reflection, 1: expansion, 2: contraction, 3: keeper (a scalar: the role branches are uniform)
  const int gl = threadIdx.x & 63;
  const int dim = gl & 3;
  const bool keeper = wv == 3;
  long long chain = blockIdx.x;         // grid = W*C exactly ...
  if constexpr (RESUME) {               // ... or the tail of a persistent launch: slot b takes a parked chain up again
    // The list, its length and the parked states were written by the wavefronts of the persistent launch on every XCD
    // (device-scope atomics on the counter, write-through stores for the states); a kernel boundary lies in between, and
    // they are read past the caches all the same, like the persistent kernel's own resume path reads them.
    if ((int)blockIdx.x >= __hip_atomic_load(a.susp_count, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT)) return;   // uniform in the workgroup, before any barrier
    chain = __hip_atomic_load(a.susp_list + blockIdx.x, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
  }
  [[maybe_unused]] auto parked = [&](int i) __attribute__((always_inline)) {   // RESUME: word i of the chain's parked state
    return __hip_atomic_load(a.state + (size_t)chain * 32 + i, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
  };
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
    if constexpr (RESUME) {   // the sorted simplex as abn_fit_refill_kernel parked it at an iteration boundary
#pragma unroll
      for (int k = 0; k < 5; ++k) vx[k] = parked(4 * k + dim);
    } else if (a.smode == 0) {
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
    if (gl == 4) xch[3] = xch[7] = 0.0;  // the exchange buffers' fourth slot: the zero the done flag is compared with
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
  // one evaluation of the candidate (al, be, wt, ic) — every lane holds all four; `pre`: preA / preB / prePen are this
  // candidate's (then al and be are not looked at)
  auto eval = [&](double al, double be, double wt, double ic, bool pre) __attribute__((always_inline)) -> double {
    ABN_STAMP(6);  // control flow + candidate fetch since the exchange
    const double p_mm = wconst[1];
    const double sv0 = wconst[0], sv1 = wt * p_mm, sv2 = (1.0 - wt) * p_mm;
    double pen;
    if (kMatrixFma && pre) {
      pen = prePen;
      ABN_STAMP(0);
      build_power_table_mx_pre(preA, preB, a.T, pw, dtab, gl);
    } else {
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

  // ---- control block (LDS, two buffers): what the evaluation wavefronts need to follow the keeper's control flow — the five
  // sorted costs BEFORE the running iteration (their decision needs c0..c4) and a `done` flag.  The keeper writes ctl[cb]
  // while the others evaluate; they read it behind the next exchange barrier, then everybody flips cb.  The flag therefore
  // reaches them one evaluation late: a finished fit costs one surplus evaluation (never counted, never used), and no
  // iteration has the cost insertion or the termination test on the evaluation wavefronts' path.
  double* ctl = gtab + 2 * kSpecPreDoubles;

  // =================================================================================================================
  // The two roles run their own loops (the role is a scalar: no lane masks, no values merged between them); what keeps
  // them in step is the barrier count, the same on both sides of every path:
  //   start: [Solver::init: 2 exchanges]  hand-over (1)
  //   per iteration: exchange (1); a shrink adds: points (1), 2 exchanges, hand-over (1)
  //   end: the keeper, finished after iteration i, meets the others at the exchange of their surplus evaluation i + 1.
  // Exchange = the evaluation wavefronts write their cost to xch[4 phase + wavefront], barrier, the keeper reads; two
  // buffers, so no second barrier.  Hand-over of a freshly sorted simplex (after Solver::init, after a shrink, after a
  // resume): the keeper writes the control block and, unless the fit is finished, the three candidates with their
  // prepared inputs; barrier; the others learn `done` and pick their candidate up.
  // =================================================================================================================
  if (!keeper) {
    int phase = 0, cb = 0, par = 0;
    double wt = 0.0, ic = 0.0, al = 0.0, be = 0.0;
    auto put = [&](double f) __attribute__((always_inline)) {
      if (gl == 0) xch[4 * phase + wv] = f;
      __syncthreads();
      phase ^= 1;
    };
    auto eval_point = [&](const double* x) __attribute__((always_inline)) { return eval(x[0], x[1], x[2], x[3], false); };
    // the candidate of outcome slot o for this wavefront: weight and intercept (every lane reads the same addresses), the
    // prepared inputs; al / be only where the matrix instruction is compiled out
    auto fetch = [&](int o) __attribute__((always_inline)) {
      const double* t = tab + par * kSpecTabDoubles + o * 12 + 4 * wv;
      const double* g = gtab + par * kSpecPreDoubles + (o * 3 + wv) * 12;
      wt = t[2];
      ic = t[3];
      if constexpr (kMatrixFma) {
        preA = g[preA_idx];
        preB = g[preB_idx];
        prePen = g[9];
      } else {
        al = t[0];
        be = t[1];
      }
    };
    auto pickup = [&]() __attribute__((always_inline)) -> bool {
      __syncthreads();
      const bool d = ctl[8 * cb + 5] != 0.0;
      if (!d) fetch(0);
      par ^= 1;
      return d;
    };
    if constexpr (!RESUME) {  // Solver::init: the five start costs in input order (3 + 2)
      put(eval_point(tab + 4 * wv));
      put(eval_point(tab + 4 * (wv == 0 ? 3 : 4)));
    }
    bool done = pickup();
    // The decision of NelderMead::next_iter on the evaluation wavefronts — the critical path: their next evaluation waits
    // for it.  The twelve comparisons it can ask for are independent of each other, so lane j < 11 loads ONE pair (X, Y) of
    // published numbers behind the barrier and two v_cmp_f64 of the whole wavefront produce all of them as bits of two
    // scalar masks (X = cost cmp_x of the exchange buffer — reflection, expansion, contraction; [3] holds 0.0 —, Y = sorted
    // cost cmp_y, the reflection cost (-1) or the done flag (5)):
    //   lane 0..3 (r,c3) (r,c2) (r,c1) (r,c0)   4..7 (c,c3) (c,c2) (c,c1) (c,c0)   8 (c,c4)   9 (e,r)   10 (0,done)
    // so that the rank of an accepted cost — insert_tail<4>'s stable insertion: b3 = f < c3, b2 = b3 && f < c2, ... — is the
    // length of the run of ones from bit 0 (reflection) or bit 4 (contraction).  ≈ 25 scalar instructions instead of a
    // ladder of vector compares and branches with every lane holding every cost (650 of an iteration's 3 700 cycles).
    const int cmp_x = gl < 11 ? (int)((0x31222220000ull >> (4 * gl)) & 0xfull) : 0;
    const int cmp_y = gl < 11 ? (int)((0x60512341234ull >> (4 * gl)) & 0xfull) - 1 : 4;
    while (!done) {
      const double f = eval(al, be, wt, ic, true);
      double* buf = xch + 4 * phase;
      if (gl == 0) buf[wv] = f;
      const double* px = buf + cmp_x;
      const double* py = cmp_y < 0 ? buf : ctl + 8 * cb + cmp_y;
      __syncthreads();
      const double X = *px, Y = *py;
      phase ^= 1;
      const uint32_t lt = (uint32_t)__builtin_amdgcn_fcmp(X, Y, 4 /* ordered < */);
      const uint32_t ge = (uint32_t)__builtin_amdgcn_fcmp(X, Y, 3 /* ordered >= */);
      ABN_STAMP(4);
#ifdef ABN_STAMPS
      ++seg[7];  // iterations seen by this wavefront
#endif
      if (lt & 0x400u) break;  // the keeper finished the fit while this (then surplus) evaluation ran
      cb ^= 1;
      const int o_r = 4 - __builtin_ctz(~lt | 0x10u), o_c = 9 - __builtin_ctz((~lt >> 4) | 0x10u);
      const uint32_t is_r = lt & (ge >> 3) & 1u;   // reflection accepted: r < c3 && r >= c0
      const uint32_t is_e = (lt >> 3) & 1u;        // else expansion tried: r < c0 — the better of the two goes in
      const uint32_t is_c = ge & 1u;               // else contraction tried: r >= c3; else: NaN reflection cost
      const int o_e = (lt & 0x200u) ? 4 : o_r;
      const int o = is_r ? o_r : (is_e ? o_e : o_c);
      if ((is_r | is_e | (is_c & (lt >> 8))) & 1u) {   // a point is accepted: outcome slot o
        fetch(o);
        par ^= 1;
        ABN_STAMP(5);  // decision
      } else if (is_c && a.shrink_variant == 0) {
        par ^= 1;  // rejected contraction, argmin 0.8.1: the simplex stays as it is — the same candidates again
      } else {     // NelderMead::shrink: vertices 1..4 re-evaluated in order (3 + 1)
        __syncthreads();
        put(eval_point(pts + 4 * wv));
        put(eval_point(pts + 12));
        done = pickup();
      }
    }
#ifdef ABN_STAMPS
    if (a.dbg && chain == 0 && wv == 0 && gl == 0) {
      for (int q = 0; q < 8; ++q) a.dbg[q] = seg[q];
      a.dbg[7] = seg[7];
    }
#endif
    return;
  }

  // =================================================================================================================
  // the keeper
  // =================================================================================================================
  int phase = 0, cb = 0, par = 0;
  double c[5], best_cost = __builtin_inf();
  bool have_best = false;
  int iter = 0;
  // IterState::update() + terminate_internal(): -1 = go on, else the ABN_FIT_* status.  `improved`: the best
  // vertex is the new best_param
  auto ctl_begin = [&](bool count_iter, bool& improved) __attribute__((always_inline)) -> int {
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
  // the three candidates of the running iteration (this lane's dimension), best_param, evaluation count
  double xr = 0.0, x_e = 0.0, x_c = 0.0, bx = __builtin_nan("");
  int evals = 0;
  // quad q < 10 of the keeper works on outcome q = (accepted point, rank): r@0..3, e@0, c@0..4
  const int oq = gl >> 2;
  const int o_kind = oq < 4 ? 0 : (oq == 4 ? 1 : 2);
  const int o_rank = oq < 4 ? oq : (oq == 4 ? 0 : oq - 5);
  // generation matrix and penalty term of the three candidates (r_, e_, c_: this lane's dimension) of outcome slot `oq` —
  // lane (quad, dimension t < 3) works for candidate t; same functions as an evaluation would call, so the same bits
  auto emit_pre = [&](double r_, double e_, double c_, int parity) __attribute__((always_inline)) {
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
  auto ctl_write = [&](int status_now) __attribute__((always_inline)) {
    if (gl < 5) ctl[8 * cb + gl] = gl == 0 ? c[0] : (gl == 1 ? c[1] : (gl == 2 ? c[2] : (gl == 3 ? c[3] : c[4])));
    if (gl == 5) ctl[8 * cb + 5] = status_now >= 0 ? 1.0 : 0.0;
  };
  auto take = [&](double& f0, double& f1, double& f2) __attribute__((always_inline)) {
    const double* buf = xch + 4 * phase;
    __syncthreads();
    f0 = buf[0];
    f1 = buf[1];
    f2 = buf[2];
    phase ^= 1;
  };
  bool improved;
  int status = -1;
  double f1, f2;
  auto publish = [&]() __attribute__((always_inline)) -> bool {
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
    __syncthreads();
    par ^= 1;
    return status >= 0;
  };

  if constexpr (RESUME) {
    // a chain parked by the persistent kernel at an iteration boundary (IterState::update and the termination test of that
    // iteration are behind it: status < 0): costs, best vertex and counters as stored, then straight to the candidates
#pragma unroll
    for (int k = 0; k < 5; ++k) c[k] = parked(20 + k);
    bx = parked(25 + dim);
    best_cost = parked(29);
    const long long ie = __double_as_longlong(parked(30));
    iter = (int)(ie & 0xffffffffll);
    evals = (int)(ie >> 32);
    have_best = __double_as_longlong(parked(31)) != 0;
  } else {
    // Solver::init: the five start costs in input order (3 + 2), stable sort, first termination check
    take(c[0], c[1], c[2]);
    take(c[3], c[4], f2);
    evals = 5;
    sort5(c, vx);
    status = ctl_begin(false, improved);
    if (improved) bx = vx[0];
  }
  bool done = publish();

  while (!done) {
    {
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
    take(fr, fe, fc);
    cb ^= 1;
    // ---- NelderMead::next_iter's decision.  Every lane holds the same costs, so the branches are uniform.
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
      c[4] = fi;
      vx[4] = which == 0 ? xr : (which == 1 ? x_e : x_c);
      insert_tail<4>(c, vx);
      status = ctl_begin(true, improved);
      if (improved) bx = vx[0];
      ctl_write(status);
      xr = t[dim];
      x_e = t[4 + dim];
      x_c = t[8 + dim];
      par ^= 1;
    } else if (kind == 1) {
      // argmin 0.8.1: a rejected contraction leaves the simplex untouched — for good: every later iteration
      // repeats this one.  no_skip == 0: finish the chain with the counters it would reach (FitArgs::no_skip)
      if (a.no_skip == 0) {
        const int rest = a.max_iters - iter - 1;
        evals += 2 * rest;
        iter += rest;
        if (a.skipped && gl == 0 && rest > 0) atomicAdd(a.skipped, 2ull * (unsigned long long)rest);
      }
      status = ctl_begin(true, improved);
      ctl_write(status);
      par ^= 1;  // same candidates again; the (identical) tables are rebuilt in the other buffers
    } else {
      // NelderMead::shrink (NaN reflection cost, or the textbook variant after a rejected contraction):
      // vertices 1..4 move towards the best by sigma and are re-evaluated in order (3 + 1)
#pragma unroll
      for (int k = 1; k < 5; ++k) {
        vx[k] = vx[0] + (vx[k] - vx[0]) * 0.5;
        if (gl < 4) pts[4 * (k - 1) + dim] = vx[k];
      }
      __syncthreads();
      take(c[1], c[2], c[3]);
      take(c[4], f1, f2);
      evals += 4;
      sort5(c, vx);
      status = ctl_begin(true, improved);
      if (improved) bx = vx[0];
      done = publish();
      continue;
    }
    // the fit is finished: the others are one evaluation ahead — meet them at that exchange
    if (status >= 0) {
      __syncthreads();
      break;
    }
  }

  const double b0 = dpp_mov<kDppQuadBcast0>(bx), b1 = dpp_mov<kDppQuadBcast1>(bx);
  if (gl < 4) a.best[(size_t)chain * 4 + gl] = bx;
  if (gl == 0) {
#ifdef ABN_DIAG_RESUME_COUNT   // diagnosis only (scripts/repro_lost_chain.py): the resume launch counts in the upper half of the word
    if constexpr (RESUME) atomicAdd(a.slice_status + 1, 0x10000u);
#else
    if constexpr (RESUME) atomicAdd(a.slice_status + 1, 1u);  // the persistent launch's count of finished fits
#endif
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

}  // namespace abn
