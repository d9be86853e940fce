// abn_fit_refill_kernel: the persistent, time-sliced form of the resident fit kernel (chain queue + FIFO of parked chains).
#pragma once
#include "abn_common.hpp"
#include "abn_fit_kernel.hpp"

namespace abn {

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
  bool wave_fd = false;      // ... some group of this wavefront has (wavefront-uniform)
  bool tail_mode = false;    // tail hand-over: every running chain of this wavefront parks at its next iteration boundary
  bool tail_park = false;    // this chain's park goes to the tail list, not to the FIFO
  unsigned wstep = 0;
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
    tail_park = false;
    if (tail_mode && status < 0 && !suspend) {   // tail hand-over: to the list of abn_fit_spec_kernel's resume launch
      suspend = true;
      tail_park = true;
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

#ifdef ABN_MEASUREMENT_KNOBS
  int prio_cur = 0;
#endif
  while (__ballot(st != ST_IDLE) != 0ull) {
#ifdef ABN_MEASUREMENT_KNOBS
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
#endif
    // tail hand-over: once a group of this wavefront has seen the queue empty, lane 0 looks every 64 steps whether nobody
    // waits any more (queue, this workgroup's FIFO) and at most tail_cap chains of the launch are unfinished
    if (a.tail_cap > 0 && wave_fd && !tail_mode && (++wstep & 63u) == 0u) {
      int t = 0;
      if (lane == 0) {
        const unsigned fq = __hip_atomic_load(a.queue, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
        const int av = __hip_atomic_load(reinterpret_cast<int*>(pht) + kParkAvail, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
        const unsigned fin = __hip_atomic_load(a.slice_status + 1, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
        t = (gridDim.x * NG + fq >= total && av <= 0 && total - fin <= (unsigned)a.tail_cap) ? 1 : 0;
      }
      tail_mode = __builtin_amdgcn_readfirstlane(t) != 0;
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
        if (parking && tail_park && gl == 0) {   // to the resume launch of abn_fit_spec_kernel (a kernel boundary away)
          a.susp_list[atomicAdd(a.susp_count, 1)] = (int)chain;
          atomicAdd(a.slice_status + 2, 1u);
        }
        if (parking && !tail_park && gl == 0) {
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
          // A failed claim leaves the counter one too low until it is given back, so a claim that runs into another
          // group's failed one can fail although a chain IS parked (the parking group's own claim, for one) — and if both
          // then went idle with no later finisher in their FIFO shard, the chain stayed parked for good: seen in round 4 as
          // "finished 13999 of 14000 chains" in 1 run of 15 of a launch whose wavefronts wind down together (tail
          // hand-over).  So whoever gives a credit back looks again: the last one to do so sees the true count.
          int* avail = reinterpret_cast<int*>(pht) + kParkAvail;
          bool claimed = false;
          for (int tries = 0; tries < 1024 && !claimed; ++tries) {
            if (atomicSub(avail, 1) > 0) {
              claimed = true;
            } else {
              atomicAdd(avail, 1);
              if (__hip_atomic_load(avail, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT) <= 0) break;
              __builtin_amdgcn_s_sleep(1);
            }
          }
          if (claimed) {
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
          }
        }
      }
      nxt = (unsigned)__builtin_amdgcn_ds_bpermute(4 * (g * G), (int)nxt);  // the group leader's draw
      take_parked = __builtin_amdgcn_ds_bpermute(4 * (g * G), (int)take_parked) != 0;
      fresh_done = __builtin_amdgcn_ds_bpermute(4 * (g * G), (int)fresh_done) != 0;
      wave_fd = wave_fd || (__ballot(fresh_done) != 0ull);
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

}  // namespace abn
