// Selection (ab_neutral::run :83-135), cost batch, bootstrap rows, materialised observations, bootstrap indices.
#pragma once
#include "abn_common.hpp"

namespace abn {

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
