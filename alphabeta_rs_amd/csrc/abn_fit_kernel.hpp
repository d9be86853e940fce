// abn_fit_kernel: the fit kernel in its plain launch — several chains per wavefront (LDS-resident pedigrees), one wavefront
// per chain, streamed pedigrees, the two-pass form and strict (serial) order.  See abn_common.hpp for the mapping.
#pragma once
#include "abn_common.hpp"

namespace abn {

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
      // count and entry past the caches (agent scope), as abn_fit_spec_kernel's resume launch reads them: written by
      // device-scope atomics of the first pass on every XCD
      valid = slot < (long long)__hip_atomic_load(a.susp_count, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
      chain_raw = valid ? (long long)__hip_atomic_load(a.susp_list + slot, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT) : 0;
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

}  // namespace abn
