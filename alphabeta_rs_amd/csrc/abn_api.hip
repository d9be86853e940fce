// C-ABI of libabneutral_hip.so (include/abneutral.h): host-side glue around the gfx950 kernels of
// abn_device.hpp.  No CPU compute path exists here by design: if HIP is unusable every compute entry
// point returns ABN_ERR_NO_DEVICE / ABN_ERR_HIP.
#include "abn_host.hpp"
#include "abn_device.hpp"

using namespace abn;

static_assert(sizeof(abn_fit_info) == sizeof(FitInfoDev), "abn_fit_info layout");


// ------------------------------------------------------------------------------------------------
// pedigree topology: distinct (t0, t1-t0, t2-t0) triples, src/divergence.rs:52,57-58
// ------------------------------------------------------------------------------------------------
static int as_i8(double x) {  // Rust `f64 as i8`: truncate, saturate, NaN -> 0
  if (x != x) return 0;
  if (x >= 127.0) return 127;
  if (x <= -128.0) return -128;
  return (int)x;
}

struct Topology {
  int N = 0, K = 0, T = 0, TP = 0, KP = 0, chain_stride = 0;
  std::vector<uint32_t> tri;
  std::vector<uint16_t> tid;
};

// rows: n rows with `stride` doubles each, generations in the first three columns
static int build_topology(const double* rows, int n, int stride, Topology& t) {
  if (!rows || n <= 0) return ABN_ERR_INVALID_ARG;
  t.N = n;
  t.tri.clear();
  t.tid.resize((size_t)n);
  std::unordered_map<uint32_t, uint32_t> seen;
  int tmax = 0;
  for (int i = 0; i < n; ++i) {
    const int t0 = as_i8(rows[(size_t)i * stride + 0]);
    const int t1 = as_i8(rows[(size_t)i * stride + 1]);
    const int t2 = as_i8(rows[(size_t)i * stride + 2]);
    // The reference inverts G for negative exponents (src/divergence.rs:17-19) and its i8 subtraction
    // can wrap; neither is meaningful for a pedigree, so such rows are rejected.
    if (t0 < 0 || t1 < t0 || t2 < t0) return ABN_ERR_BAD_PEDIGREE;
    const int ea = t1 - t0, eb = t2 - t0;
    tmax = std::max(tmax, std::max(t0, std::max(ea, eb)));
    const uint32_t key = (uint32_t)t0 | ((uint32_t)ea << 8) | ((uint32_t)eb << 16);
    auto it = seen.find(key);
    uint32_t id;
    if (it == seen.end()) {
      id = (uint32_t)t.tri.size();
      if (id >= 65535u) return ABN_ERR_INVALID_ARG;
      seen.emplace(key, id);
      t.tri.push_back(key);
    } else {
      id = it->second;
    }
    t.tid[(size_t)i] = (uint16_t)id;
  }
  t.K = (int)t.tri.size();
  t.T = tmax;
  t.TP = tmax + 1;
  t.KP = (t.K + 1) & ~1;
  t.chain_stride = kPw * t.TP + t.KP + 4;  // power table, dt1t2 per triple, 4 per-chain constants
  return ABN_OK;
}

struct DevTopology {
  DevBuf<uint32_t> tri;
  DevBuf<uint16_t> tid;
};

static int upload_topology(abn_ctx* c, const Topology& t, DevTopology& d) {
  HIPCHK(c, d.tri.alloc(t.tri.size()));
  HIPCHK(c, d.tid.alloc(t.tid.size()));
  HIPCHK(c, hipMemcpyAsync(d.tri.p, t.tri.data(), d.tri.bytes(), hipMemcpyHostToDevice, c->stream));
  HIPCHK(c, hipMemcpyAsync(d.tid.p, t.tid.data(), d.tid.bytes(), hipMemcpyHostToDevice, c->stream));
  HIPCHK(c, hipStreamSynchronize(c->stream));
  return ABN_OK;
}

// ------------------------------------------------------------------------------------------------
// launch configuration
// ------------------------------------------------------------------------------------------------
constexpr size_t kDefaultDynLds = 64 * 1024;   // what a launch may ask for without opting in
constexpr size_t kMaxDynLds = 160 * 1024;      // gfx950: the whole LDS of a CU, for the one-chain-per-workgroup kernels
// A workgroup with one chain (stream-mode fits, selection, 64-lane cost) may need more than 64 KiB for pedigrees with
// thousands of distinct triples: opt the kernel in (hipFuncAttributeMaxDynamicSharedMemorySize) before such a launch.
static hipError_t allow_lds(const void* kernel, size_t lds) {
  if (lds <= kDefaultDynLds) return hipSuccess;
  return hipFuncSetAttribute(kernel, hipFuncAttributeMaxDynamicSharedMemorySize, (int)lds);
}
constexpr size_t kLdsResidentMax = 40 * 1024;
// Launch geometry in wavefronts (or chains) PER CU; the totals in the comments are the MI355X's (256 CUs).
#ifndef ABN_PERSIST_WAVES_PER_CU
#define ABN_PERSIST_WAVES_PER_CU 12
#endif
constexpr long long kPersistWavesPerCu = ABN_PERSIST_WAVES_PER_CU;  // a persistent launch: 3 per SIMD x 4 SIMDs (3072)
// A launch that would just about fill the resident wavefronts (2048 < wavefronts <= 3072: C3's 10 000 bootstraps are 2500)
// runs persistent on 2048 of them instead: the last fifth of the chains waits in the queue, finished groups refill and
// time slicing evens out the tail (C3 phase B 2.58 -> 2.44 ms; same box: 2.61 -> 2.53 ms, 1792 / 2304 wavefronts 2.71 /
// 2.82 ms; profiles/r03_persist_waves_sweep.txt)
#ifndef ABN_PERSIST_WAVES_SMALL_PER_CU
#define ABN_PERSIST_WAVES_SMALL_PER_CU 8
#endif
constexpr long long kPersistWavesSmallPerCu = ABN_PERSIST_WAVES_SMALL_PER_CU;
static long long persist_waves(const abn_ctx* c) { return kPersistWavesPerCu * c->cus; }
static long long persist_waves_small(const abn_ctx* c) { return kPersistWavesSmallPerCu * c->cus; }
static long long persist_waves_for(const abn_ctx* c, long long blocks) {
  return blocks > persist_waves(c) ? persist_waves(c) : persist_waves_small(c);
}
#ifndef ABN_PHASE_A_SPEC_PER_CU
#define ABN_PHASE_A_SPEC_PER_CU 4
#endif
constexpr long long kPhaseASpecPerCu = ABN_PHASE_A_SPEC_PER_CU;  // chains per CU abn_fit_spec_kernel keeps resident at up to two rows per lane (spec_max_chains)
constexpr long long kPhaseAWidePerCu = 24;                      // ... and up to which it uses one wavefront per chain (6144)
// Time slicing of persistent launches (FitArgs::quantum): evaluations a chain runs before it yields to waiting chains.
#ifndef ABN_QUANTUM
#define ABN_QUANTUM 256
#endif
constexpr int kQuantum = ABN_QUANTUM;
constexpr size_t kSliceStateMax = (size_t)256 << 20;  // bytes of parked state (32 doubles per chain of the launch)
constexpr int kPhaseACap = 1000;  // first-pass iteration cap of the two-pass phase A

// Lanes of a wavefront per chain.  Auto: by pedigree rows, then widened until the workgroup's LDS
// (64/G chains x chain_stride doubles) leaves room for >= 8 workgroups per CU (160 KiB LDS).
constexpr size_t kLdsTargetPerBlock = 20 * 1024;
static int pick_lanes(int n, int requested, int chain_stride) {
  if (requested == 8 || requested == 16 || requested == 32 || requested == 64) return requested;
  int g = 64;
  if (n <= 32) g = 8;
  else if (n <= 128) g = 16;
  else if (n <= 256) g = 32;
  const size_t per_chain = ((size_t)chain_stride + (size_t)n) * sizeof(double);  // scratch + resident observations
  while (g < 64 && (size_t)(kWave / g) * per_chain > kLdsTargetPerBlock) g *= 2;
  return g;
}
static bool fit_streams(int n, int chain_stride, int lanes, int strict = 0);
static bool tree_on_wave_ok(int n_rows, int chain_stride, int tree, int strict);
static bool spec_applicable(const FitArgs& a);
static int launch_fit_spec(abn_ctx* c, FitArgs a, hipStream_t st);
// The residual reduction tree (FitArgs::tree; abn_fit_info.lanes).  Auto (lanes_per_chain == 0) and the pedigree
// LDS-resident: the canonical 64-accumulator tree, which every kernel — packed, one wavefront per chain, four
// wavefronts per chain — runs at its native cost.  Streamed pedigrees and explicit lane counts: one accumulator per
// lane of the packed kernel.
static int pick_tree(int n, int requested, int chain_stride, int lanes, int strict = 0) {
  if (strict) return 1;  // serial row order (abn_options.strict_order): no tree
  if (requested != 0 || fit_streams(n, chain_stride, lanes, strict)) return lanes;
  return kTreeCanon;
}
static int pick_rmax(int n, int lanes) {
  const int per = (n + lanes - 1) / lanes;
  if (per <= 1) return 1;
  if (per <= 2) return 2;
  if (per <= 4) return 4;
  if (per <= 8) return 8;
  // 512 < N <= 1024: still LDS-resident with one wavefront per chain (scripts/n_sweep.py, N = 820, 2000 bootstraps:
  // 5.25 -> 4.0 ms; 32 rows per lane with the triple ids in LDS gained nothing over streaming: not kept)
  if (per <= 16 && lanes == 64) return 16;
  return 0;  // stream mode
}

template <int G, bool TP, bool STRICT>
static hipError_t launch_fit_gt(const FitArgs& a, int rmax, dim3 grid, size_t lds, hipStream_t s) {
  switch (rmax) {
    case 1: hipLaunchKernelGGL((abn_fit_kernel<G, 1, TP, STRICT>), grid, dim3(kWave), lds, s, a); break;
    case 2: hipLaunchKernelGGL((abn_fit_kernel<G, 2, TP, STRICT>), grid, dim3(kWave), lds, s, a); break;
    case 4: hipLaunchKernelGGL((abn_fit_kernel<G, 4, TP, STRICT>), grid, dim3(kWave), lds, s, a); break;
    case 8: hipLaunchKernelGGL((abn_fit_kernel<G, 8, TP, STRICT>), grid, dim3(kWave), lds, s, a); break;
    case 16:  // one wavefront per chain only (pick_rmax)
      hipLaunchKernelGGL((abn_fit_kernel<64, 16, TP, STRICT>), grid, dim3(kWave), lds, s, a);
      break;
    case -1:
      if (STRICT) {  // strict order has one stream variant (chunks of 8 G rows)
        if (hipError_t e = allow_lds(reinterpret_cast<const void*>(&abn_fit_kernel<G, 0, TP, STRICT>), lds)) return e;
        hipLaunchKernelGGL((abn_fit_kernel<G, 0, TP, STRICT>), grid, dim3(kWave), lds, s, a);
        break;
      }
      if (hipError_t e = allow_lds(reinterpret_cast<const void*>(&abn_fit_kernel<G, -1, TP, false>), lds)) return e;
      hipLaunchKernelGGL((abn_fit_kernel<G, -1, TP, false>), grid, dim3(kWave), lds, s, a);
      break;
    default:
      if (hipError_t e = allow_lds(reinterpret_cast<const void*>(&abn_fit_kernel<G, 0, TP, STRICT>), lds)) return e;
      hipLaunchKernelGGL((abn_fit_kernel<G, 0, TP, STRICT>), grid, dim3(kWave), lds, s, a);
      break;
  }
  return hipGetLastError();
}
template <int G>
static hipError_t launch_fit_refill(const FitArgs& a, int rmax, dim3 grid, size_t lds, hipStream_t s) {
  switch (rmax) {
    case 1: hipLaunchKernelGGL((abn_fit_refill_kernel<G, 1>), grid, dim3(kWave), lds, s, a); break;
    case 2: hipLaunchKernelGGL((abn_fit_refill_kernel<G, 2>), grid, dim3(kWave), lds, s, a); break;
    case 4: hipLaunchKernelGGL((abn_fit_refill_kernel<G, 4>), grid, dim3(kWave), lds, s, a); break;
    default: hipLaunchKernelGGL((abn_fit_refill_kernel<G, 8>), grid, dim3(kWave), lds, s, a); break;
  }
  return hipGetLastError();
}
template <int G>
static hipError_t launch_fit_g(const FitArgs& a, int rmax, dim3 grid, size_t lds, hipStream_t s, bool refill) {
  const bool twopass = a.iter_cap > 0 || a.resume != 0;
  if (refill) return launch_fit_refill<G>(a, rmax, grid, lds, s);
  if (a.strict) return launch_fit_gt<G, false, true>(a, rmax, grid, lds, s);  // launch_fit: never two-pass, never persistent
  return twopass ? launch_fit_gt<G, true, false>(a, rmax, grid, lds, s) : launch_fit_gt<G, false, false>(a, rmax, grid, lds, s);
}

// `a.chain_stride` must be the topology's scratch stride (kPw*TP + KP + 4, even); the resident variant adds an even
// number of doubles (observations + triple list): every chain's region stays 16-byte aligned for load_matrix.
// kind (nullable): the ABN_KERNEL_* code of what was launched (PERSISTENT: a.slice_status then counts its fits)
static int launch_fit(abn_ctx* c, FitArgs a, int lanes, hipStream_t st, int* kind = nullptr) {
  if (kind) *kind = ABN_KERNEL_NONE;
  const FitArgs a0 = a;  // as the caller set it (topology's scratch stride): the tail's resume launch starts from it
  const long long chains = (long long)a.W * a.C;
  if (chains <= 0) return ABN_OK;
  const int ng = kWave / lanes;
  int rmax = pick_rmax(a.N, lanes);
  // the reduction tree: the canonical one (any resident kernel) or one accumulator per lane; strict order: none (serial)
  if (a.strict) {
    if (a.iter_cap > 0 || a.resume != 0)
      return set_err(c, ABN_ERR_INVALID_ARG, "internal: strict order has no two-pass variant");
    a.tree = 1;
    a.queue = nullptr;  // no persistent variant either
  } else if (a.tree != kTreeCanon) {
    a.tree = lanes;
  }
  // resident observations + this chain's triple list (even: 16-byte aligned chains) (+ strict order: the rows' terms)
  const int np = ((a.N + 1) & ~1) + (((a.K + 1) / 2 + 1) & ~1) + (a.strict ? ((a.N + 1) & ~1) : 0);
  if (rmax > 0) {
    if ((size_t)ng * (size_t)(a.chain_stride + np) * sizeof(double) > kLdsResidentMax) rmax = 0;
    else a.chain_stride += np;
  }
  if (rmax <= 0 && a.tree == kTreeCanon)
    return set_err(c, ABN_ERR_INVALID_ARG, "internal: the canonical tree needs an LDS-resident pedigree");
  if (rmax <= 0 && a.strict) a.chain_stride += kStrictRowsPerLane * lanes;  // one chunk of terms
  // stream mode: rows shorter than one trip of the deep loop (kStreamBlocks x 4 rows x lanes) use the pair-loop variant
  if (rmax == 0 && a.N < 2 * kStreamBlocks * kStreamVec * lanes) rmax = -1;
  const size_t lds = (size_t)ng * (size_t)a.chain_stride * sizeof(double);
  if (lds > (lanes == kWave ? kMaxDynLds : kDefaultDynLds))
    return set_err(c, ABN_ERR_INVALID_ARG, "pedigree needs more LDS per workgroup than supported (T or K too large)");
  long long blocks = (chains + ng - 1) / ng;
  if (blocks > 0x7fffffffLL || chains > 0x7fffffffLL)
    return set_err(c, ABN_ERR_INVALID_ARG, "too many chains for one launch");
  // More wavefronts than the GPU holds at once and several chains per wavefront: the persistent kernel, whose
  // groups take the next chain from a queue when their fit ends (abn_fit_refill_kernel)
  const bool refill = a.queue != nullptr && rmax > 0 && ng > 1 && a.iter_cap == 0 && a.resume == 0 &&
                      blocks > persist_waves_small(c);
  if (refill) blocks = persist_waves_for(c, blocks);
  if (!refill) a.quantum = 0;
  // The quantum grows with the queue's depth (chains per lane group of the launch): a deep queue keeps the GPU full whatever
  // the slicing, every park costs a wavefront ≈ 10 µs of dependent memory traffic, and the tail goes to the speculative
  // kernel anyway; a shallow one needs short slices to start everybody early.  scripts/quantum_sweep.sh, profiles/r04_quantum_sweep.txt:
  // C3 (1.2 chains per group) is best at 256, the C4 shard (2.0) at 256-384, the metaprofile shape (2.4) at 384-768, C4's
  // 200 000 chains (16) at >= 1024.  Results do not depend on it (the persistent kernel is schedule-independent).
  if (refill && a.quantum > 0) {
    const long long q = (5LL * kQuantum * chains / (blocks * ng) / 8 + 63) & ~63LL;
    a.quantum = (int)std::min<long long>(4LL * kQuantum, std::max<long long>(kQuantum, q));
  }
#ifdef ABN_MEASUREMENT_KNOBS  // scripts/prio_sweep.sh: wave priority by chain age, wavefronts and quantum of the persistent launch
  if (refill) {
    if (const char* e = getenv("ABN_PRIO")) sscanf(e, "%d,%d,%d,%d", &a.prio_mode, &a.prio_t[0], &a.prio_t[1], &a.prio_t[2]);
    if (const char* e = getenv("ABN_PERSIST_WAVES_SMALL_ENV")) {
      if (blocks == persist_waves_small(c)) blocks = std::max(64, atoi(e));
    }
    if (const char* e = getenv("ABN_QUANTUM_ENV")) {
      if (a.quantum > 0) a.quantum = std::max(16, atoi(e));
    }
    // tests/test_gpu_parity.py::test_lost_fifo_entry_is_an_error_at_sync: the first parked chain of FIFO shard 0 is never
    // published — the launch must end (bounded spin), and every way of taking results must report ABN_ERR_HIP
    if (const char* e = getenv("ABN_DROP_FIFO_ENTRY")) a.drop_entry = atoi(e);
  }
#endif
  // Tail hand-over (FitArgs::tail_cap): the last chains of a time-sliced launch finish on four wavefronts each instead of one
  // by one on an emptying GPU at the packed kernel's step time (metaprofile shape, phase A: 12 of 17 ms were such a tail).
  // As many as abn_fit_spec_kernel keeps resident (four per CU at up to two rows per lane), twice that behind the deep queues
  // of the 12-wavefronts-per-CU geometry, where the later workgroups start as the first end (scripts/tail_sweep.sh,
  // profiles/r04_tail_sweep.txt: C3 is best at 1024, the C4 shard and the metaprofile shape at 2048-3072: +3 % / +2 %).
  // Needs the speculative kernel to apply to the pedigree.
  a.tail_cap = 0;
  if (refill && a.quantum > 0 && a.slice_status && a.susp_list && a.tree == kTreeCanon) {
    FitArgs probe = a0;
    probe.tree = kTreeCanon;
    if (spec_applicable(probe))
      a.tail_cap = (int)((pick_rmax(a.N, kWave) <= 2 ? (blocks == persist_waves(c) ? 8LL : 4LL) : 2LL) * c->cus);
#ifdef ABN_MEASUREMENT_KNOBS
    if (const char* e = getenv("ABN_TAIL_CAP")) a.tail_cap = a.tail_cap > 0 ? (int)std::min<long long>(chains, std::max(0, atoi(e))) : 0;
#endif
    a.susp_count = reinterpret_cast<int*>(a.slice_status + 3);
  }
  if (refill && a.slice_status) HIPCHK(c, hipMemsetAsync(a.slice_status, 0, 4 * sizeof(unsigned), st));
  if (kind)
    *kind = refill ? ABN_KERNEL_PERSISTENT
            : rmax <= 0 ? ABN_KERNEL_STREAM
            : (a.iter_cap > 0 || a.resume != 0) ? ABN_KERNEL_TWO_PASS : ABN_KERNEL_RESIDENT;
  if (a.quantum > 0) {  // empty FIFO of parked chains: entries -1, head = tail = 0
    HIPCHK(c, hipMemsetAsync(a.parked, 0xff, (size_t)kParkShards * a.park_cap * sizeof(int), st));
    HIPCHK(c, hipMemsetAsync(a.park_ht, 0, (size_t)kParkShards * kParkHeaderInts * sizeof(unsigned), st));
  }
  dim3 grid((unsigned)blocks);
  hipError_t e;
  switch (lanes) {
    case 8: e = launch_fit_g<8>(a, rmax, grid, lds, st, refill); break;
    case 16: e = launch_fit_g<16>(a, rmax, grid, lds, st, refill); break;
    case 32: e = launch_fit_g<32>(a, rmax, grid, lds, st, refill); break;
    default: e = launch_fit_g<64>(a, rmax, grid, lds, st, refill); break;
  }
  HIPCHK(c, e);
  if (a.tail_cap > 0) {  // the parked tail (possibly empty: workgroups beyond *susp_count leave at once)
    FitArgs r = a0;
    r.tree = kTreeCanon;
    r.queue = nullptr;
    r.quantum = 0;
    r.spec_resume = 1;
    r.tail_cap = a.tail_cap;
    r.state = a.state;
    r.susp_list = a.susp_list;
    r.susp_count = a.susp_count;
    r.slice_status = a.slice_status;
    if (int rc = launch_fit_spec(c, r, st)) return rc;
  }
  return ABN_OK;
}

// true when launch_fit will use the stream variant for this pedigree / lane count.  Strict order keeps the rows' terms in
// LDS next to the observations (N more doubles per chain): the same footprint launch_fit computes, so that what a plan
// decides (and validates) at abn_plan_create is what runs.
static bool fit_streams(int n, int chain_stride, int lanes, int strict) {
  if (pick_rmax(n, lanes) == 0) return true;
  const int np = ((n + 1) & ~1) + n / 2 + 2 + (strict ? ((n + 1) & ~1) : 0);  // observations + triple list (K <= n) (+ terms)
  return (size_t)(kWave / lanes) * (size_t)(chain_stride + np) * sizeof(double) > kLdsResidentMax;
}

// Speculative kernel (phase A; three evaluation wavefronts + a bookkeeping wavefront per chain): resident mode
// with one wavefront per candidate only.
// Chains up to which the speculative kernel is used.  What the GPU holds at once: four workgroups per CU for pedigrees of up
// to two rows per lane (1024 chains on the MI355X), three beyond (768); workgroups beyond that start as earlier ones end.
// Phase B (bootstrap chains: similar lengths) up to 1.5 x / 1 x of that; phase A (start chains from random points: lengths
// differ several-fold, so the queue behind the resident chains drains into slots that free early) up to 4 x / 2.7 x.
// Speculative / one wavefront per chain / packed, ms (scripts/b_kernel_sweep.py, scripts/a_kernel_sweep.py,
// profiles/r04_b_kernel_sweep.txt, r04_a_kernel_sweep.txt):
//   phase B, C3 topology: 1000 chains 0.76 / 1.03 / 1.60, 1500: 1.04 / 1.15 / 1.16, 2000: 1.28 / 1.31 / 1.31, 3000: 1.70 / 1.49 / 1.48;
//            6-row pedigree 1500: 1.44 / 1.95 / 1.95, 3000: 1.95 / 2.43 / 2.26; 351-row pedigree 500: 1.21 / 1.95 / 1.95, 1000: 2.14 / 1.97 / 1.96
//   phase A, C3 topology: 1000 chains 1.89 / 2.94 / 4.80, 2000: 2.71 / 3.31 / 4.94, 3000: 3.45 / 3.67 / 5.06, 4000: 4.21 / 4.40 / 6.13,
//            5000: 5.10 / 5.16 / 6.46, 6000: 6.15 / 5.75 / 6.43; 351-row pedigree 1000: 3.25 / 3.77 / 3.78, 2000: 6.11 / 6.43 / 6.31, 3000: 6.30 / 6.55 / 6.54
static long long spec_max_chains(const abn_ctx* c, int n_rows, int phase) {
  const long long mx = kPhaseASpecPerCu * c->cus;   // 1024
  const bool small = pick_rmax(n_rows, kWave) <= 2;
  if (phase == 0) return small ? mx * 4 : mx * 2;
  return small ? mx * 3 / 2 : mx * 3 / 4;
}

// a wavefront per chain runs the canonical tree (or, strict order, the serial sum) whenever the pedigree is LDS-resident
// at 64 lanes per chain; an explicit lanes_per_chain tree only when it IS 64 lanes
static bool tree_on_wave_ok(int n_rows, int chain_stride, int tree, int strict) {
  if (!strict && tree != kTreeCanon) return tree == kWave;
  const int rmax = pick_rmax(n_rows, kWave);
  if (rmax <= 0) return false;
  const int np = ((n_rows + 1) & ~1) + n_rows / 2 + 2 + (strict ? ((n_rows + 1) & ~1) : 0);  // strict order: + the terms
  return (size_t)(chain_stride + np) * sizeof(double) <= kLdsResidentMax;
}

static bool spec_applicable(const FitArgs& a) {
  if (a.dmode == 2) return false;  // resident observations only (starts, or bootstraps gathered through the index row)
  const int rmax = pick_rmax(a.N, kWave);
  if (rmax == 0 || rmax > 8) return false;  // 16 rows per lane: the plain resident kernel
  if (!tree_on_wave_ok(a.N, a.chain_stride, a.tree, a.strict)) return false;
  const int np = ((a.N + 1) & ~1) * (a.strict ? 2 : 1);   // observations (+ strict order: the rows' terms)
  return (3 * (size_t)(a.chain_stride + np) + kSpecCommDoubles) * sizeof(double) <= kLdsResidentMax;
}

static int launch_fit_spec(abn_ctx* c, FitArgs a, hipStream_t st) {
  // spec_resume: one workgroup per slot of the tail list (at most tail_cap chains were handed over)
  const long long chains = a.spec_resume ? (long long)a.tail_cap : (long long)a.W * a.C;
  if (chains <= 0) return ABN_OK;
  const int rmax = pick_rmax(a.N, kWave);
  a.chain_stride += ((a.N + 1) & ~1) * (a.strict ? 2 : 1);
  const size_t lds = (3 * (size_t)a.chain_stride + kSpecCommDoubles) * sizeof(double);
  dim3 grid((unsigned)chains), block(4 * kWave);
  a.tree = a.strict ? 1 : kTreeCanon;  // spec_applicable admitted it
  if (a.spec_resume) {  // the tail of a persistent launch (never strict)
    switch (rmax) {
      case 1: hipLaunchKernelGGL((abn_fit_spec_kernel<1, false, true>), grid, block, lds, st, a); break;
      case 2: hipLaunchKernelGGL((abn_fit_spec_kernel<2, false, true>), grid, block, lds, st, a); break;
      case 4: hipLaunchKernelGGL((abn_fit_spec_kernel<4, false, true>), grid, block, lds, st, a); break;
      default: hipLaunchKernelGGL((abn_fit_spec_kernel<8, false, true>), grid, block, lds, st, a); break;
    }
  } else if (a.strict) {
    switch (rmax) {
      case 1: hipLaunchKernelGGL((abn_fit_spec_kernel<1, true>), grid, block, lds, st, a); break;
      case 2: hipLaunchKernelGGL((abn_fit_spec_kernel<2, true>), grid, block, lds, st, a); break;
      case 4: hipLaunchKernelGGL((abn_fit_spec_kernel<4, true>), grid, block, lds, st, a); break;
      default: hipLaunchKernelGGL((abn_fit_spec_kernel<8, true>), grid, block, lds, st, a); break;
    }
  } else {
    switch (rmax) {
      case 1: hipLaunchKernelGGL((abn_fit_spec_kernel<1, false>), grid, block, lds, st, a); break;
      case 2: hipLaunchKernelGGL((abn_fit_spec_kernel<2, false>), grid, block, lds, st, a); break;
      case 4: hipLaunchKernelGGL((abn_fit_spec_kernel<4, false>), grid, block, lds, st, a); break;
      default: hipLaunchKernelGGL((abn_fit_spec_kernel<8, false>), grid, block, lds, st, a); break;
    }
  }
  HIPCHK(c, hipGetLastError());
  return ABN_OK;
}

// ------------------------------------------------------------------------------------------------
// context
// ------------------------------------------------------------------------------------------------
extern "C" void abn_default_options(abn_options* o) {
  if (!o) return;
  o->seed = 20260101ull;
  o->lanes_per_chain = 0;
  o->strict_order = 0;
  o->shrink_on_failed_contraction = 0;
  o->max_iters_start = 10000;  // src/ab_neutral.rs:62
  o->max_iters_boot = 1000;    // src/boot_model.rs:81
  o->stream_mode = 0;
  o->sd_tolerance = 2.220446049250313e-16;  // f64::EPSILON
  o->window_groups = 0;
  o->no_fixed_point_skip = 0;
}

static abn_options resolve(const abn_options* o) {
  abn_options d;
  abn_default_options(&d);
  if (o) d = *o;
  return d;
}

// The summation order of a pedigree (abn_options.strict_order: -1 tree, 0 auto, 1 serial) resolved to 0 / 1.  Auto sums
// pedigrees of up to kSerialSumMaxRows rows SERIALLY in row order — the reference's `square_sum += ...`
// (src/structs.rs:206-213), where a serial sum costs nothing measurable and the bundled data/ pedigree (6 rows: the
// north star's parity target, on which `weight` is not identified and a last-ulp difference in a cost can move a
// bootstrap row by O(1)) is then bit-equal to the reference order by DEFAULT.  Like the tree, a function of the pedigree
// (and the options) alone: never of the launch.  An explicit lanes_per_chain keeps its per-lane tree.
constexpr int kSerialSumMaxRows = 16;
static abn_options resolve_for(const abn_options* o, int n_rows) {
  abn_options d = resolve(o);
  if (d.strict_order == 0 && d.lanes_per_chain == 0 && n_rows <= kSerialSumMaxRows) d.strict_order = 1;
  else if (d.strict_order < 0) d.strict_order = 0;
  return d;
}

// iteration budgets must leave room for the 32-bit evaluation counters (at most 2 evaluations per iteration plus the
// 4 of a shrink); lane counts are 0 (auto) or a power of two up to the wavefront; the tolerance must compare
static const char* options_error(const abn_options& o) {
  if (o.max_iters_start < 0 || o.max_iters_start > (1 << 28) || o.max_iters_boot < 0 || o.max_iters_boot > (1 << 28))
    return "max_iters_start / max_iters_boot must be in 0 .. 2^28";
  if (!(o.lanes_per_chain == 0 || o.lanes_per_chain == 8 || o.lanes_per_chain == 16 || o.lanes_per_chain == 32 ||
        o.lanes_per_chain == 64))
    return "lanes_per_chain must be 0 (auto), 8, 16, 32 or 64";
  if (o.sd_tolerance != o.sd_tolerance) return "sd_tolerance is NaN";
  if (o.stream_mode < 0 || o.stream_mode > 1) return "stream_mode must be 0 or 1";
  if (o.window_groups < 0) return "window_groups must be >= 0";
  if (o.strict_order < -1 || o.strict_order > 1) return "strict_order must be -1 (tree), 0 (auto) or 1 (serial)";
  return nullptr;
}

// The residual reduction tree of a pedigree (abn_options.lanes_per_chain; abn_fit_info.lanes): host arithmetic on
// the topology alone — no device, no launch size.
extern "C" int abn_reduction_tree(const abn_options* opts, const double* generations, int32_t n_rows, int32_t* tree) {
  if (!generations || n_rows <= 0 || !tree) return ABN_ERR_INVALID_ARG;
  if (const char* oe = options_error(resolve(opts))) {
    (void)oe;
    return ABN_ERR_INVALID_ARG;
  }
  const abn_options o = resolve_for(opts, n_rows);
  Topology t;
  const int rc = build_topology(generations, n_rows, 3, t);
  if (rc) return rc;
  const int lanes = pick_lanes(n_rows, o.lanes_per_chain, t.chain_stride);
  *tree = o.strict_order ? 1
          : fit_streams(n_rows, t.chain_stride, lanes) ? (lanes | ((kStreamVec - 1) << 8))
                                                       : pick_tree(n_rows, o.lanes_per_chain, t.chain_stride, lanes);
  return ABN_OK;
}

extern "C" int abn_version(void) { return ABN_VERSION_MAJOR * 100 + ABN_VERSION_MINOR; }

extern "C" int abn_device_count(int* count) {
  if (!count) return ABN_ERR_INVALID_ARG;
  int n = 0;
  if (hipGetDeviceCount(&n) != hipSuccess) {
    *count = 0;
    return ABN_ERR_NO_DEVICE;
  }
  *count = n;
  return n > 0 ? ABN_OK : ABN_ERR_NO_DEVICE;
}

extern "C" int abn_init(int device_ordinal, void* stream, abn_ctx** out) {
  if (!out) return ABN_ERR_INVALID_ARG;
  *out = nullptr;
  int n = 0;
  if (hipGetDeviceCount(&n) != hipSuccess || n <= 0) return ABN_ERR_NO_DEVICE;
  if (device_ordinal < 0 || device_ordinal >= n) return ABN_ERR_INVALID_ARG;
  if (hipSetDevice(device_ordinal) != hipSuccess) return ABN_ERR_HIP;
  hipDeviceProp_t prop;
  if (hipGetDeviceProperties(&prop, device_ordinal) != hipSuccess) return ABN_ERR_HIP;
  // the library holds gfx950 code only, and its one-chain-per-workgroup kernels opt in to a CU's whole 160 KiB of LDS
  if (strncmp(prop.gcnArchName, "gfx950", 6) != 0 || prop.multiProcessorCount <= 0 ||
      (size_t)prop.maxSharedMemoryPerMultiProcessor < kMaxDynLds)
    return ABN_ERR_NO_DEVICE;
  abn_ctx* c = new (std::nothrow) abn_ctx();
  if (!c) return ABN_ERR_HIP;
  c->device = device_ordinal;
  c->cus = prop.multiProcessorCount;
  c->lds_per_cu = (size_t)prop.maxSharedMemoryPerMultiProcessor;
  if (stream == ABN_STREAM_DEFAULT) {
    c->stream = nullptr;  // the null stream
    c->own_stream = false;
  } else if (stream) {
    c->stream = (hipStream_t)stream;
    c->own_stream = false;
  } else {
    if (hipStreamCreateWithFlags(&c->stream, hipStreamNonBlocking) != hipSuccess) {
      delete c;
      return ABN_ERR_HIP;
    }
    c->own_stream = true;
  }
  *out = c;
  return ABN_OK;
}

extern "C" int abn_device_info(const abn_ctx* c, int32_t* out4) {
  if (!c || !out4) return ABN_ERR_INVALID_ARG;
  out4[0] = c->cus;
  out4[1] = (int32_t)(c->lds_per_cu / 1024);
  out4[2] = (int32_t)persist_waves(c);
  out4[3] = (int32_t)persist_waves_small(c);
  return ABN_OK;
}

extern "C" int abn_shutdown(abn_ctx* c) {
  if (!c) return ABN_ERR_INVALID_ARG;
  if (c->own_stream && c->stream) (void)hipStreamDestroy(c->stream);
  for (auto st : c->side) (void)hipStreamDestroy(st);
  (void)hipSetDevice(c->device);
  c->pool->clear();
  c->pool->closed = true;
  delete c;
  return ABN_OK;
}

extern "C" const char* abn_last_error(const abn_ctx* c) { return c ? c->err.c_str() : "null context"; }

extern "C" const char* abn_status_string(int s) {
  switch (s) {
    case ABN_OK: return "ok";
    case ABN_ERR_INVALID_ARG: return "invalid argument";
    case ABN_ERR_BAD_PEDIGREE: return "bad pedigree (generation outside 0..127 or t1/t2 < t0)";
    case ABN_ERR_NO_DEVICE: return "no HIP device";
    case ABN_ERR_HIP: return "HIP runtime error";
    case ABN_ERR_NO_FINITE_FIT: return "no start produced a finite fit";
    case ABN_ERR_STATE: return "plan used out of order";
    default: return "unknown status";
  }
}

// ------------------------------------------------------------------------------------------------
// deterministic inputs (host side)
// ------------------------------------------------------------------------------------------------
// Model::new, src/structs.rs:78-96, five vertices per start (src/ab_neutral.rs:49-55)
extern "C" int abn_gen_start_simplices(uint64_t seed, uint32_t window, int32_t n_starts, double max_divergence,
                                       double* simplex0) {
  if (!simplex0 || n_starts < 0) return ABN_ERR_INVALID_ARG;
  const uint32_t k0 = (uint32_t)seed, k1 = (uint32_t)(seed >> 32);
  double mx = max_divergence;
  if (!(max_divergence > 0.0)) mx = 0.1;  // :80-83
  for (int32_t s = 0; s < n_starts; ++s)
    for (uint32_t v = 0; v < 5; ++v) {
      uint32_t r0[4], r1[4];
      philox4x32_10(0u, (uint32_t)s * 5u + v, window, kTagStart, k0, k1, r0);
      philox4x32_10(1u, (uint32_t)s * 5u + v, window, kTagStart, k0, k1, r1);
      double* o = simplex0 + ((size_t)s * 5 + v) * 4;
      o[0] = std::pow(10.0, uniform_from(r0[0], r0[1], -9.0, -2.0));
      o[1] = std::pow(10.0, uniform_from(r0[2], r0[3], -9.0, -2.0));
      o[2] = uniform_from(r1[0], r1[1], 0.0, 0.1);
      o[3] = uniform_from(r1[2], r1[3], 0.0, mx);
    }
  return ABN_OK;
}

extern "C" int abn_gen_boot_simplices(uint64_t seed, uint32_t window, uint32_t b0, int64_t nb, const double params[4],
                                      double* simplex0) {
  if (!simplex0 || !params || nb < 0) return ABN_ERR_INVALID_ARG;
  const uint32_t k0 = (uint32_t)seed, k1 = (uint32_t)(seed >> 32);
  for (int64_t i = 0; i < nb; ++i) {
    double* o = simplex0 + (size_t)i * 20;
    for (int d = 0; d < 4; ++d) o[d] = params[d];
    for (uint32_t v = 1; v < 5; ++v) {
      uint32_t r0[4], r1[4];
      philox4x32_10((v - 1u) * 2u + 0u, b0 + (uint32_t)i, window, kTagJitter, k0, k1, r0);
      philox4x32_10((v - 1u) * 2u + 1u, b0 + (uint32_t)i, window, kTagJitter, k0, k1, r1);
      o[4 * v + 0] = vary_one(params[0], r0[0], r0[1]);
      o[4 * v + 1] = vary_one(params[1], r0[2], r0[3]);
      o[4 * v + 2] = vary_one(params[2], r1[0], r1[1]);
      o[4 * v + 3] = vary_one(params[3], r1[2], r1[3]);
    }
  }
  return ABN_OK;
}

static int launch_gen_idx(abn_ctx* c, uint32_t* idx, int n, int b, int w, uint64_t seed, uint32_t woff, uint32_t boff,
                          const uint32_t* wid = nullptr) {
  const long long total = (long long)w * b * ((n + 3) / 4);
  if (total <= 0) return ABN_OK;
  const long long want = (total + 255) / 256;
  const unsigned blocks = (unsigned)std::min<long long>(want, 32LL * c->cus);
  hipLaunchKernelGGL(abn_gen_idx_kernel, dim3(blocks), dim3(256), 0, c->stream, idx, n, b, w, seed, woff, boff, wid);
  HIPCHK(c, hipGetLastError());
  return ABN_OK;
}

extern "C" int abn_gen_boot_indices(abn_ctx* c, uint64_t seed, uint32_t window, uint32_t b0, int64_t nb, int32_t n_rows,
                                    uint32_t* idx) {
  if (!c || !idx || nb < 0 || n_rows <= 0 || nb > 0x7fffffff) return ABN_ERR_INVALID_ARG;
  if (nb == 0) return ABN_OK;
  HIPCHK(c, hipSetDevice(c->device));
  PoolScope pool_scope(c);
  DevBuf<uint32_t> d;
  HIPCHK(c, d.alloc((size_t)nb * (size_t)n_rows));
  int rc = launch_gen_idx(c, d.p, n_rows, (int)nb, 1, seed, window, b0);
  if (rc) return rc;
  HIPCHK(c, hipMemcpyAsync(idx, d.p, d.bytes(), hipMemcpyDeviceToHost, c->stream));
  HIPCHK(c, hipStreamSynchronize(c->stream));
  return ABN_OK;
}

// ------------------------------------------------------------------------------------------------
// (1) cost batch
// ------------------------------------------------------------------------------------------------
extern "C" int abn_cost_batch(abn_ctx* c, const abn_options* opts, const double* pedigree, int32_t n_rows, double p_uu0,
                              double eqp, double eqp_weight, const double* candidates, int64_t m, const double* pred,
                              const double* resid, const uint32_t* idx, const uint32_t* cand_to_boot,
                              int64_t n_boot_rows, double* cost, double* dt1t2, double* p_uu_inf) {
  if (!c) return ABN_ERR_INVALID_ARG;
  if (!pedigree || n_rows <= 0 || !candidates || m < 0 || !cost) return set_err(c, ABN_ERR_INVALID_ARG, "null/size");
  if (idx && (!pred || !resid || n_boot_rows <= 0)) return set_err(c, ABN_ERR_INVALID_ARG, "bootstrap inputs");
  if (m == 0) return ABN_OK;
  if (const char* oe = options_error(resolve(opts))) return set_err(c, ABN_ERR_INVALID_ARG, oe);
  const abn_options o = resolve_for(opts, n_rows);
  HIPCHK(c, hipSetDevice(c->device));
  PoolScope pool_scope(c);
  Topology t;
  int rc = build_topology(pedigree, n_rows, 4, t);
  if (rc) return set_err(c, rc, abn_status_string(rc));
  if (idx) {
    for (int64_t i = 0; i < m; ++i) {
      const int64_t b = cand_to_boot ? (int64_t)cand_to_boot[i] : i;
      if (b >= n_boot_rows) return set_err(c, ABN_ERR_INVALID_ARG, "cand_to_boot out of range");
    }
  }
  DevTopology dt;
  rc = upload_topology(c, t, dt);
  if (rc) return rc;
  const int N = n_rows;
  std::vector<double> dcol((size_t)N);
  for (int i = 0; i < N; ++i) dcol[(size_t)i] = pedigree[(size_t)i * 4 + 3];
  DevBuf<double> dD, dpred, dresid, dcand, dcost, ddt, dpuu;
  DevBuf<uint32_t> didx, dc2b;
  HIPCHK(c, dD.alloc((size_t)N));
  HIPCHK(c, dcand.alloc((size_t)m * 4));
  HIPCHK(c, dcost.alloc((size_t)m));
  HIPCHK(c, hipMemcpyAsync(dD.p, dcol.data(), dD.bytes(), hipMemcpyHostToDevice, c->stream));
  HIPCHK(c, hipMemcpyAsync(dcand.p, candidates, dcand.bytes(), hipMemcpyHostToDevice, c->stream));
  if (idx) {
    HIPCHK(c, dpred.alloc((size_t)N));
    HIPCHK(c, dresid.alloc((size_t)N));
    HIPCHK(c, didx.alloc((size_t)n_boot_rows * (size_t)N));
    HIPCHK(c, hipMemcpyAsync(dpred.p, pred, dpred.bytes(), hipMemcpyHostToDevice, c->stream));
    HIPCHK(c, hipMemcpyAsync(dresid.p, resid, dresid.bytes(), hipMemcpyHostToDevice, c->stream));
    HIPCHK(c, hipMemcpyAsync(didx.p, idx, didx.bytes(), hipMemcpyHostToDevice, c->stream));
    for (size_t q = 0; q < (size_t)n_boot_rows * (size_t)N; ++q)
      if (idx[q] >= (uint32_t)N) return set_err(c, ABN_ERR_INVALID_ARG, "bootstrap index out of range");
    if (cand_to_boot) {
      HIPCHK(c, dc2b.alloc((size_t)m));
      HIPCHK(c, hipMemcpyAsync(dc2b.p, cand_to_boot, dc2b.bytes(), hipMemcpyHostToDevice, c->stream));
    }
  }
  if (dt1t2) HIPCHK(c, ddt.alloc((size_t)m * (size_t)N));
  if (p_uu_inf) HIPCHK(c, dpuu.alloc((size_t)m));

  const int lanes = o.strict_order ? 64 : pick_lanes(N, o.lanes_per_chain, t.chain_stride);
  const int ng = kWave / lanes;
  CostArgs a{};
  a.tri = dt.tri.p;
  a.tid = dt.tid.p;
  a.N = N;
  a.K = t.K;
  a.T = t.T;
  a.TP = t.TP;
  a.chain_stride = t.chain_stride;
  a.p_uu0 = p_uu0;
  a.eqp = eqp;
  a.eqp_w = eqp_weight;
  a.D = dD.p;
  a.pred = dpred.p;
  a.resid = dresid.p;
  a.idx = didx.p;
  a.cand_to_boot = dc2b.p;
  a.dmode = idx ? 1 : 0;
  a.cand = dcand.p;
  a.M = m;
  a.strict = o.strict_order ? 1 : 0;
  a.tree = o.strict_order ? lanes : pick_tree(N, o.lanes_per_chain, t.chain_stride, lanes);
  a.cost = dcost.p;
  a.dt = ddt.p;
  a.puu = dpuu.p;
  size_t lds = ((size_t)ng * t.chain_stride + (o.strict_order ? kSelChunk : 0)) * sizeof(double);
  if (lds > (lanes == kWave ? kMaxDynLds : kDefaultDynLds))
    return set_err(c, ABN_ERR_INVALID_ARG, "pedigree needs more LDS than supported");
  if (lanes == kWave) HIPCHK(c, allow_lds(reinterpret_cast<const void*>(&abn_cost_kernel<64>), lds));
  const long long blocks = (m + ng - 1) / ng;
  if (blocks > 0x7fffffffLL) return set_err(c, ABN_ERR_INVALID_ARG, "too many candidates");
  dim3 grid((unsigned)blocks);
  switch (lanes) {
    case 8: hipLaunchKernelGGL(abn_cost_kernel<8>, grid, dim3(kWave), lds, c->stream, a); break;
    case 16: hipLaunchKernelGGL(abn_cost_kernel<16>, grid, dim3(kWave), lds, c->stream, a); break;
    case 32: hipLaunchKernelGGL(abn_cost_kernel<32>, grid, dim3(kWave), lds, c->stream, a); break;
    default: hipLaunchKernelGGL(abn_cost_kernel<64>, grid, dim3(kWave), lds, c->stream, a); break;
  }
  HIPCHK(c, hipGetLastError());
  HIPCHK(c, hipMemcpyAsync(cost, dcost.p, dcost.bytes(), hipMemcpyDeviceToHost, c->stream));
  if (dt1t2) HIPCHK(c, hipMemcpyAsync(dt1t2, ddt.p, ddt.bytes(), hipMemcpyDeviceToHost, c->stream));
  if (p_uu_inf) HIPCHK(c, hipMemcpyAsync(p_uu_inf, dpuu.p, dpuu.bytes(), hipMemcpyDeviceToHost, c->stream));
  HIPCHK(c, hipStreamSynchronize(c->stream));
  return ABN_OK;
}

// ------------------------------------------------------------------------------------------------
// Nelder-Mead fit batch (explicit start simplices)
// ------------------------------------------------------------------------------------------------
extern "C" int abn_fit_batch(abn_ctx* c, const abn_options* opts, const double* pedigree, int32_t n_rows, double p_uu0,
                             double eqp, double eqp_weight, const double* simplex0, int64_t f, const double* dobs_rows,
                             int32_t max_iters, double* best, abn_fit_info* info) {
  if (!c) return ABN_ERR_INVALID_ARG;
  if (!pedigree || n_rows <= 0 || !simplex0 || f < 0 || !best || max_iters < 0 || f > 0x7fffffff)
    return set_err(c, ABN_ERR_INVALID_ARG, "null/size");
  if (f == 0) return ABN_OK;
  if (const char* oe = options_error(resolve(opts))) return set_err(c, ABN_ERR_INVALID_ARG, oe);
  const abn_options o = resolve_for(opts, n_rows);
  if (max_iters < 0 || max_iters > (1 << 28)) return set_err(c, ABN_ERR_INVALID_ARG, "max_iters must be in 0 .. 2^28");
  HIPCHK(c, hipSetDevice(c->device));
  PoolScope pool_scope(c);
  Topology t;
  int rc = build_topology(pedigree, n_rows, 4, t);
  if (rc) return set_err(c, rc, abn_status_string(rc));
  DevTopology dt;
  rc = upload_topology(c, t, dt);
  if (rc) return rc;
  const int N = n_rows;
  DevBuf<double> dD, ds0, dbest, dscal;
  DevBuf<FitInfoDev> dinfo;
  std::vector<double> dcol;
  const double* dsrc = dobs_rows;
  size_t dcount = (size_t)f * (size_t)N;
  if (!dobs_rows) {
    dcol.resize((size_t)N);
    for (int i = 0; i < N; ++i) dcol[(size_t)i] = pedigree[(size_t)i * 4 + 3];
    dsrc = dcol.data();
    dcount = (size_t)N;
  }
  const double scal[3] = {p_uu0, eqp, eqp_weight};
  HIPCHK(c, dD.alloc(dcount));
  HIPCHK(c, ds0.alloc((size_t)f * 20));
  HIPCHK(c, dbest.alloc((size_t)f * 4));
  HIPCHK(c, dinfo.alloc((size_t)f));
  HIPCHK(c, dscal.alloc(3));
  HIPCHK(c, hipMemcpyAsync(dD.p, dsrc, dD.bytes(), hipMemcpyHostToDevice, c->stream));
  HIPCHK(c, hipMemcpyAsync(ds0.p, simplex0, ds0.bytes(), hipMemcpyHostToDevice, c->stream));
  HIPCHK(c, hipMemcpyAsync(dscal.p, scal, sizeof scal, hipMemcpyHostToDevice, c->stream));
  HIPCHK(c, hipMemsetAsync(dinfo.p, 0, dinfo.bytes(), c->stream));

  FitArgs a{};
  a.tri = dt.tri.p;
  a.tid = dt.tid.p;
  a.N = N;
  a.K = t.K;
  a.T = t.T;
  a.TP = t.TP;
  a.chain_stride = t.chain_stride;
  a.p_uu = dscal.p;
  a.eqp = dscal.p + 1;
  a.eqp_w = dscal.p + 2;
  a.wstride = 0;
  a.dmode = 0;
  a.D = dD.p;
  a.smode = 0;
  a.simplex0 = ds0.p;
  a.seed = o.seed;
  if (dobs_rows) {  // one "window" per fit: its own observed divergences
    a.W = (int)f;
    a.C = 1;
  } else {
    a.W = 1;
    a.C = (int)f;
  }
  a.max_iters = max_iters;
  a.shrink_variant = o.shrink_on_failed_contraction ? 1 : 0;
  a.no_skip = o.no_fixed_point_skip ? 1 : 0;
  a.skipped = nullptr;
  a.sd_tol = o.sd_tolerance;
  a.gap_tol = 64.0 * o.sd_tolerance;
  a.best = dbest.p;
  a.info = dinfo.p;
  a.raw = nullptr;
  const int lanes = pick_lanes(N, o.lanes_per_chain, t.chain_stride);
  a.strict = o.strict_order ? 1 : 0;
  a.tree = pick_tree(N, o.lanes_per_chain, t.chain_stride, lanes, a.strict);
  rc = launch_fit(c, a, lanes, c->stream);
  if (rc) return rc;
  HIPCHK(c, hipMemcpyAsync(best, dbest.p, dbest.bytes(), hipMemcpyDeviceToHost, c->stream));
  if (info) HIPCHK(c, hipMemcpyAsync(info, dinfo.p, dinfo.bytes(), hipMemcpyDeviceToHost, c->stream));
  HIPCHK(c, hipStreamSynchronize(c->stream));
  return ABN_OK;
}

// ------------------------------------------------------------------------------------------------
// (4) device-resident plan
// ------------------------------------------------------------------------------------------------
struct abn_plan {
  abn_ctx* ctx = nullptr;
  abn_options opt{};
  Topology topo;
  DevTopology dtopo;
  int N = 0, W = 0, S = 0, B = 0;
  uint32_t window_offset = 0, boot_offset = 0;
  int lanes = 16;    // lanes per chain of the packed (throughput) kernels
  int tree = 32;     // the pedigree's reduction tree (accumulators): lanes or 2 x lanes (pick_tree)
  int lanes_a = 64;  // phase A (few chains: latency-bound, one wavefront per chain is fastest)
  std::vector<uint32_t> wid_host;  // Philox window ids (abn_plan_set_window_ids); empty = window_offset + w
  DevBuf<uint32_t> wid;
  bool windows_set = false, phase_a_done = false, ran_a = false, ran_b = false;
  DevBuf<double> D, pred, resid, p_uu, eqp, eqp_w, simplexA, bestA, model, lse, bestB, raw_own;
  DevBuf<FitInfoDev> infoA, infoB;
  DevBuf<int32_t> best_start;
  DevBuf<uint32_t> idx;
  DevBuf<double> dstar;  // stream mode: materialised bootstrap observations [W x B x N]
  DevBuf<double> nm_state;    // two-pass phase A: parked Nelder-Mead states [W x S x 32]
  DevBuf<int> susp_list;      // [W x S] + 1 counter at the end
  // per phase (A, B): [2*ph] evaluations not executed (fixed-point skip), [2*ph+1] chain queue of the persistent kernel
  DevBuf<unsigned long long> skipped;
  bool twopass_a = false;
  DevBuf<int> slice_buf;      // time slicing: head, tail, then the FIFO of parked chains
  unsigned slice_cap = 0;
  DevBuf<unsigned> slice_status;   // per phase four words: error word, fits finished by the persistent launch (and its tail's
                                   // resume launch), chains handed to the tail, the tail list's fill count (FitArgs::slice_status)
  long long persist_expected[2] = {0, 0};  // chains the last persistent launch of phase A / B had to finish (0: none)
  long long tail_handed[2] = {0, 0};       // ... of which its tail handed to the speculative kernel (read at the last sync)
  int32_t last_kernels[4] = {0, 0, 0, 0};  // abn_plan_last_kernels
  bool stream_b = false;
  double* raw = nullptr;  // raw_own.p or caller-bound
  hipEvent_t ev[6] = {nullptr, nullptr, nullptr, nullptr, nullptr, nullptr};
  hipEvent_t ev_fork = nullptr;
  std::vector<hipEvent_t> ev_join;
};

extern "C" int abn_plan_destroy(abn_plan* p) {
  if (!p) return ABN_ERR_INVALID_ARG;
  for (auto& e : p->ev)
    if (e) (void)hipEventDestroy(e);
  if (p->ev_fork) (void)hipEventDestroy(p->ev_fork);
  for (auto e : p->ev_join) (void)hipEventDestroy(e);
  delete p;
  return ABN_OK;
}

extern "C" int abn_plan_create(abn_ctx* c, const abn_options* opts, const double* generations, int32_t n_rows,
                               int32_t n_windows, int32_t n_starts, int32_t n_boot, uint32_t window_offset,
                               uint32_t boot_offset, abn_plan** out) {
  if (!c || !out) return ABN_ERR_INVALID_ARG;
  *out = nullptr;
  if (!generations || n_rows <= 0 || n_windows <= 0 || n_starts < 0 || n_boot < 0)
    return set_err(c, ABN_ERR_INVALID_ARG, "null/size");
  if ((long long)n_windows * std::max(n_starts, n_boot) > 0x7fffffffLL)
    return set_err(c, ABN_ERR_INVALID_ARG, "too many chains");
  {
    const abn_options o = resolve(opts);
    if (const char* oe = options_error(o)) return set_err(c, ABN_ERR_INVALID_ARG, oe);
  }
  HIPCHK(c, hipSetDevice(c->device));
  PoolScope pool_scope(c);
  abn_plan* p = new (std::nothrow) abn_plan();
  if (!p) return ABN_ERR_HIP;
  p->ctx = c;
  p->opt = resolve_for(opts, n_rows);  // strict_order resolved to 0 / 1 for this pedigree
  p->N = n_rows;
  p->W = n_windows;
  p->S = n_starts;
  p->B = n_boot;
  p->window_offset = window_offset;
  p->boot_offset = boot_offset;
  int rc = build_topology(generations, n_rows, 3, p->topo);
  if (rc) {
    delete p;
    return set_err(c, rc, abn_status_string(rc));
  }
  p->lanes = pick_lanes(n_rows, p->opt.lanes_per_chain, p->topo.chain_stride);
  p->tree = pick_tree(n_rows, p->opt.lanes_per_chain, p->topo.chain_stride, p->lanes, p->opt.strict_order);
  p->lanes_a = p->lanes;
  // Phase A is latency-bound while its chains fit the machine about twice over (3 wavefronts x 1024 SIMDs): one
  // wavefront per chain then beats packing several chains into a wavefront, and below ~1000 chains the
  // four-wavefront speculative kernel beats both (scripts/phase_a_sweep.py, C3 topology: 1000 chains 2.6 / 3.2 /
  // 4.6 ms for speculative / 64 lanes / 16 lanes, 1500 chains 4.2 / 3.4 / 4.8; 4000 chains - / 4.6 / 5.9 ms;
  // 8000 chains - / 7.7 / 7.0 ms)
  // The reduction tree stays the pedigree's (p->tree) whichever kernel runs: results do not depend on the size of
  // the launch, hence not on how a job is sharded over GPUs.
  if (p->opt.lanes_per_chain == 0 && (long long)n_windows * n_starts <= kPhaseAWidePerCu * c->cus &&
      tree_on_wave_ok(n_rows, p->topo.chain_stride, p->tree, p->opt.strict_order))
    p->lanes_a = 64;
  // the footprint launch_fit will ask for when the pedigree is streamed (resident launches stay below kLdsResidentMax by
  // construction): the scratch of the workgroup's chains, plus one chunk of terms per chain in strict order — validated
  // here, not at the first run
  const size_t stream_stride = (size_t)p->topo.chain_stride + (p->opt.strict_order ? (size_t)kStrictRowsPerLane * p->lanes : 0);
  if ((size_t)(kWave / p->lanes) * stream_stride * sizeof(double) > (p->lanes == kWave ? kMaxDynLds : kDefaultDynLds) ||
      ((size_t)p->topo.chain_stride + kSelChunk) * sizeof(double) > kMaxDynLds) {
    delete p;
    return set_err(c, ABN_ERR_INVALID_ARG, "pedigree needs more LDS per workgroup than supported (T or K too large)");
  }
  auto fail = [&](hipError_t e, const char* what) {
    abn_plan_destroy(p);
    return set_err(c, ABN_ERR_HIP, std::string(what) + ": " + hipGetErrorString(e));
  };
  rc = upload_topology(c, p->topo, p->dtopo);
  if (rc) {
    abn_plan_destroy(p);
    return rc;
  }
  const size_t W = (size_t)n_windows, N = (size_t)n_rows, S = (size_t)n_starts, B = (size_t)n_boot;
  hipError_t e;
#define PALLOC(buf, count)                         \
  if ((e = p->buf.alloc(count)) != hipSuccess) return fail(e, "hipMalloc " #buf)
  PALLOC(D, W * N);
  PALLOC(pred, W * N);
  PALLOC(resid, W * N);
  PALLOC(p_uu, W);
  PALLOC(eqp, W);
  PALLOC(eqp_w, W);
  PALLOC(model, W * 4);
  PALLOC(best_start, W);
  PALLOC(simplexA, W * S * 20);
  PALLOC(bestA, W * S * 4);
  PALLOC(infoA, W * S);
  PALLOC(lse, W * S);
  PALLOC(idx, W * B * N);
  PALLOC(bestB, W * B * 4);
  PALLOC(infoB, W * B);
  PALLOC(raw_own, W * B * 7);
  PALLOC(skipped, 4);
  PALLOC(slice_status, 8);
  // Phase A with many chains when the repetitions of stuck fits must be executed (no_fixed_point_skip): 7 % of
  // random starts run into argmin's fixed point and repeat it up to iteration 10000; dispatched late in one launch
  // such a chain runs alone for tens of milliseconds.  Two passes: every chain for at most kPhaseACap iterations,
  // then the unfinished ones, compacted, all resident at once.  With the default skip those chains end at once and
  // one pass is faster (metaprofile shape, 30000 start chains: 14.6 ms against 18.3 ms).
  p->twopass_a = (long long)n_windows * n_starts > 4096 && p->opt.max_iters_start > kPhaseACap &&
                 p->opt.no_fixed_point_skip != 0 && p->opt.shrink_on_failed_contraction == 0 && !p->opt.strict_order;
  if (p->twopass_a) {
    PALLOC(nm_state, W * S * 32);
    PALLOC(susp_list, W * S + 1);
  }
  {  // time slicing for launches that outgrow the resident set of the persistent kernel (4 x kPersistWaves chains at 16 lanes)
    const size_t chains = W * std::max(S, B);
    if (kQuantum > 0 && p->lanes < kWave && chains > (size_t)persist_waves_small(c) * (size_t)(kWave / p->lanes) &&
        chains * 32 * sizeof(double) <= kSliceStateMax && chains < (1u << 27) && p->opt.window_groups <= 1) {
      if (p->nm_state.n < chains * 32) PALLOC(nm_state, chains * 32);
      p->slice_cap = (unsigned)(chains * 16 / kParkShards + 4096);   // per shard; a full shard just stops parking
      PALLOC(slice_buf, (size_t)kParkShards * ((size_t)kParkHeaderInts + (size_t)p->slice_cap));
      if (p->susp_list.n < chains + 1) PALLOC(susp_list, chains + 1);  // the tail list of the hand-over to the speculative kernel
    }
  }
  p->stream_b = n_boot > 0 && fit_streams(n_rows, p->topo.chain_stride, p->lanes, p->opt.strict_order) && p->opt.stream_mode == 0;
  if (p->stream_b) PALLOC(dstar, W * B * N);
#undef PALLOC
  p->raw = p->raw_own.p;
  for (auto& ev : p->ev)
    if ((e = hipEventCreate(&ev)) != hipSuccess) return fail(e, "hipEventCreate");
  *out = p;
  return ABN_OK;
}

static int plan_upload_model(abn_plan* p, const double* model, const double* pred, const double* resid,
                             const double* p0uu, const double* eqp, const double* eqp_w) {
  abn_ctx* c = p->ctx;
  const size_t W = (size_t)p->W;
  HIPCHK(c, hipMemcpyAsync(p->model.p, model, p->model.bytes(), hipMemcpyHostToDevice, c->stream));
  HIPCHK(c, hipMemcpyAsync(p->pred.p, pred, p->pred.bytes(), hipMemcpyHostToDevice, c->stream));
  HIPCHK(c, hipMemcpyAsync(p->resid.p, resid, p->resid.bytes(), hipMemcpyHostToDevice, c->stream));
  HIPCHK(c, hipMemcpyAsync(p->p_uu.p, p0uu, W * sizeof(double), hipMemcpyHostToDevice, c->stream));
  HIPCHK(c, hipMemcpyAsync(p->eqp.p, eqp, W * sizeof(double), hipMemcpyHostToDevice, c->stream));
  HIPCHK(c, hipMemcpyAsync(p->eqp_w.p, eqp_w, W * sizeof(double), hipMemcpyHostToDevice, c->stream));
  int rc = launch_gen_idx(c, p->idx.p, p->N, p->B, p->W, p->opt.seed, p->window_offset, p->boot_offset, p->wid.p);
  if (rc) return rc;
  HIPCHK(c, hipStreamSynchronize(c->stream));
  p->windows_set = true;
  p->phase_a_done = true;
  return ABN_OK;
}

extern "C" int abn_plan_set_window_ids(abn_plan* p, const uint32_t* ids) {
  if (!p) return ABN_ERR_INVALID_ARG;
  abn_ctx* c = p->ctx;
  if (p->windows_set) return set_err(c, ABN_ERR_STATE, "abn_plan_set_window_ids must precede abn_plan_set_windows");
  HIPCHK(c, hipSetDevice(c->device));
  PoolScope pool_scope(c);
  if (!ids) {
    p->wid_host.clear();
    p->wid.release();
    return ABN_OK;
  }
  p->wid_host.assign(ids, ids + p->W);
  HIPCHK(c, p->wid.alloc((size_t)p->W));
  HIPCHK(c, hipMemcpyAsync(p->wid.p, p->wid_host.data(), p->wid.bytes(), hipMemcpyHostToDevice, c->stream));
  HIPCHK(c, hipStreamSynchronize(c->stream));
  return ABN_OK;
}

extern "C" int abn_plan_set_windows(abn_plan* p, const double* d_obs, const double* p0uu, const double* eqp,
                                    const double* eqp_weight) {
  if (!p) return ABN_ERR_INVALID_ARG;
  abn_ctx* c = p->ctx;
  if (!d_obs || !p0uu) return set_err(c, ABN_ERR_INVALID_ARG, "null window data");
  HIPCHK(c, hipSetDevice(c->device));
  const size_t W = (size_t)p->W, N = (size_t)p->N, S = (size_t)p->S;
  std::vector<double> ones(W, 1.0);  // eqp_weight = 1.0, src/alphabeta.rs:37,50
  const double* e1 = eqp ? eqp : p0uu;  // eqp = p0uu, src/alphabeta.rs:36,49
  const double* e2 = eqp_weight ? eqp_weight : ones.data();
  HIPCHK(c, hipMemcpyAsync(p->D.p, d_obs, p->D.bytes(), hipMemcpyHostToDevice, c->stream));
  HIPCHK(c, hipMemcpyAsync(p->p_uu.p, p0uu, W * sizeof(double), hipMemcpyHostToDevice, c->stream));
  HIPCHK(c, hipMemcpyAsync(p->eqp.p, e1, W * sizeof(double), hipMemcpyHostToDevice, c->stream));
  HIPCHK(c, hipMemcpyAsync(p->eqp_w.p, e2, W * sizeof(double), hipMemcpyHostToDevice, c->stream));
  std::vector<double> sx(W * S * 20);
  for (size_t w = 0; w < W; ++w) {
    double mx = d_obs[w * N];  // max of column 3, src/ab_neutral.rs:25-29
    for (size_t i = 1; i < N; ++i) mx = std::max(mx, d_obs[w * N + i]);
    const uint32_t wg = p->wid_host.empty() ? p->window_offset + (uint32_t)w : p->wid_host[w];
    abn_gen_start_simplices(p->opt.seed, wg, p->S, mx, sx.data() + w * S * 20);
  }
  if (!sx.empty())
    HIPCHK(c, hipMemcpyAsync(p->simplexA.p, sx.data(), p->simplexA.bytes(), hipMemcpyHostToDevice, c->stream));
  int rc = launch_gen_idx(c, p->idx.p, p->N, p->B, p->W, p->opt.seed, p->window_offset, p->boot_offset, p->wid.p);
  if (rc) return rc;
  HIPCHK(c, hipStreamSynchronize(c->stream));  // sx / ones are host temporaries
  p->windows_set = true;
  p->phase_a_done = false;
  return ABN_OK;
}

static void fill_common(const abn_plan* p, FitArgs& a) {
  a.tri = p->dtopo.tri.p;
  a.tid = p->dtopo.tid.p;
  a.N = p->N;
  a.K = p->topo.K;
  a.T = p->topo.T;
  a.TP = p->topo.TP;
  a.chain_stride = p->topo.chain_stride;
  a.p_uu = p->p_uu.p;
  a.eqp = p->eqp.p;
  a.eqp_w = p->eqp_w.p;
  a.wstride = 1;
  a.D = p->D.p;
  a.pred = p->pred.p;
  a.resid = p->resid.p;
  a.idx = p->idx.p;
  a.model = p->model.p;
  a.seed = p->opt.seed;
  a.window_offset = p->window_offset;
  a.boot_offset = p->boot_offset;
  a.wid = p->wid.p;
  a.tree = p->tree;
  a.strict = p->opt.strict_order ? 1 : 0;
  a.W = p->W;
  a.shrink_variant = p->opt.shrink_on_failed_contraction ? 1 : 0;
  a.no_skip = p->opt.no_fixed_point_skip ? 1 : 0;
  a.sd_tol = p->opt.sd_tolerance;
  a.gap_tol = 64.0 * p->opt.sd_tolerance;
}

// Phase A (starts) + selection for windows [w0, w0+wn) on stream st.  ev != nullptr: record the plan's
// timing events around the kernels.
static int enqueue_phase_a(abn_plan* p, int w0, int wn, hipStream_t st, bool timed, bool refill) {
  abn_ctx* c = p->ctx;
  const size_t N = (size_t)p->N, S = (size_t)p->S, o = (size_t)w0;
  FitArgs a{};
  fill_common(p, a);
  a.p_uu += o;
  a.eqp += o;
  a.eqp_w += o;
  a.D += o * N;
  a.window_offset += (uint32_t)w0;
  if (a.wid) a.wid += o;
  a.W = wn;
  a.dmode = 0;
  a.smode = 0;
  a.simplex0 = p->simplexA.p + o * S * 20;
  a.C = p->S;
  a.max_iters = p->opt.max_iters_start;
  a.best = p->bestA.p + o * S * 4;
  a.info = p->infoA.p + o * S;
  a.raw = nullptr;
  a.skipped = p->skipped.p;
  a.queue = refill ? reinterpret_cast<unsigned*>(p->skipped.p + 1) : nullptr;
  if (refill && p->slice_cap > 0 && w0 == 0 && wn == p->W) {
    a.quantum = kQuantum;
    a.park_cap = p->slice_cap;
    a.park_ht = reinterpret_cast<unsigned*>(p->slice_buf.p);
    a.parked = p->slice_buf.p + kParkShards * kParkHeaderInts;
    a.state = p->nm_state.p;
    a.susp_list = p->susp_list.p;   // tail hand-over (launch_fit decides whether it applies)
  }
  const bool whole = w0 == 0 && wn == p->W;   // window groups on side streams share the plan's status words: unchecked
  if (whole) a.slice_status = p->slice_status.p;
  int kind = ABN_KERNEL_NONE;
  if (timed) HIPCHK(c, hipEventRecord(p->ev[0], st));
  // few chains: latency-bound -> three wavefronts per chain evaluate reflection / expansion / contraction at once,
  // a fourth keeps the simplex and prepares the next candidates meanwhile (abn_fit_spec_kernel)
  bool spec = p->lanes_a == 64 && p->opt.lanes_per_chain == 0 &&
              (long long)p->W * p->S <= spec_max_chains(c, p->N, 0) && spec_applicable(a);
  int lanes_a = p->lanes_a;
#ifdef ABN_MEASUREMENT_KNOBS  // ABN_PHASE_A_KERNEL = spec | wide | packed  (scripts/phase_a_sweep.py)
  if (const char* e = getenv("ABN_PHASE_A_KERNEL")) {
    const bool can_wide = p->opt.lanes_per_chain == 0 && tree_on_wave_ok(p->N, p->topo.chain_stride, p->tree, p->opt.strict_order);
    if (!strcmp(e, "spec")) spec = can_wide && spec_applicable(a);
    if (!strcmp(e, "wide")) { spec = false; if (can_wide) lanes_a = kWave; }
    if (!strcmp(e, "packed")) { spec = false; lanes_a = p->lanes; }
  }
#endif
  int rc;
  if (spec) {
    rc = launch_fit_spec(c, a, st);
    kind = ABN_KERNEL_SPECULATIVE;
  } else if (p->twopass_a && w0 == 0 && wn == p->W) {
    kind = ABN_KERNEL_TWO_PASS;
    int* cnt = p->susp_list.p + (size_t)p->W * S;
    HIPCHK(c, hipMemsetAsync(cnt, 0, sizeof(int), st));
    a.state = p->nm_state.p;
    a.susp_list = p->susp_list.p;
    a.susp_count = cnt;
    a.iter_cap = kPhaseACap;
    a.resume = 0;
    rc = launch_fit(c, a, lanes_a, st);             // pass 1: everybody, capped
    if (!rc) {
      a.iter_cap = 0;
      a.resume = 1;
      rc = launch_fit(c, a, lanes_a, st);           // pass 2: the parked chains, to the end
    }
  } else {
    rc = launch_fit(c, a, lanes_a, st, &kind);
  }
  if (rc) return rc;
  if (whole) p->persist_expected[0] = kind == ABN_KERNEL_PERSISTENT ? (long long)wn * p->S : 0;
  if (timed) {
    p->last_kernels[0] = kind;
    p->last_kernels[1] = spec ? kWave : lanes_a;
  }
  if (timed) HIPCHK(c, hipEventRecord(p->ev[1], st));
  SelectArgs s{};
  s.tri = a.tri;
  s.tid = a.tid;
  s.N = a.N;
  s.K = a.K;
  s.T = a.T;
  s.TP = a.TP;
  s.p_uu = a.p_uu;
  s.D = a.D;
  s.models = a.best;
  s.info = a.info;
  s.W = wn;
  s.S = p->S;
  s.lse = p->lse.p + o * S;
  s.model = p->model.p + o * 4;
  s.pred = p->pred.p + o * N;
  s.resid = p->resid.p + o * N;
  s.best_start = p->best_start.p + o;
  const size_t lds = ((size_t)kPw * a.TP + p->topo.KP + kSelChunk) * sizeof(double);
  if (timed) HIPCHK(c, hipEventRecord(p->ev[2], st));
  HIPCHK(c, allow_lds(reinterpret_cast<const void*>(&abn_select_lse_kernel), lds));
  HIPCHK(c, allow_lds(reinterpret_cast<const void*>(&abn_select_kernel), lds));
  hipLaunchKernelGGL(abn_select_lse_kernel, dim3((unsigned)((long long)wn * p->S)), dim3(kWave), lds, st, s);
  hipLaunchKernelGGL(abn_select_kernel, dim3((unsigned)wn), dim3(kWave), lds, st, s);
  HIPCHK(c, hipGetLastError());
  if (timed) HIPCHK(c, hipEventRecord(p->ev[3], st));
  return ABN_OK;
}

// Phase B (bootstraps) for windows [w0, w0+wn) on stream st
static int enqueue_phase_b(abn_plan* p, int w0, int wn, hipStream_t st, bool timed, bool refill) {
  abn_ctx* c = p->ctx;
  const size_t N = (size_t)p->N, B = (size_t)p->B, o = (size_t)w0;
  FitArgs a{};
  fill_common(p, a);
  a.p_uu += o;
  a.eqp += o;
  a.eqp_w += o;
  a.pred += o * N;
  a.resid += o * N;
  a.idx += o * B * N;
  a.model += o * 4;
  a.window_offset += (uint32_t)w0;
  if (a.wid) a.wid += o;
  a.W = wn;
  a.dmode = 1;
  a.smode = 1;
  a.C = p->B;
  a.max_iters = p->opt.max_iters_boot;
  a.best = p->bestB.p + o * B * 4;
  a.info = p->infoB.p + o * B;
  a.raw = p->raw + o * B * 7;
  a.skipped = p->skipped.p + 2;
  a.queue = refill ? reinterpret_cast<unsigned*>(p->skipped.p + 3) : nullptr;
  if (refill && p->slice_cap > 0 && w0 == 0 && wn == p->W) {
    a.quantum = kQuantum;
    a.park_cap = p->slice_cap;
    a.park_ht = reinterpret_cast<unsigned*>(p->slice_buf.p);
    a.parked = p->slice_buf.p + kParkShards * kParkHeaderInts;
    a.state = p->nm_state.p;
    a.susp_list = p->susp_list.p;   // tail hand-over (launch_fit decides whether it applies)
  }
  const bool whole = w0 == 0 && wn == p->W;
  if (whole) a.slice_status = p->slice_status.p + 4;
  int kind = ABN_KERNEL_SPECULATIVE;
  if (timed) HIPCHK(c, hipEventRecord(p->ev[4], st));
  if (p->stream_b) {  // gather the bootstrap observations once per fit, then stream them
    double* dst = p->dstar.p + o * B * N;
    const long long total = (long long)wn * p->B * p->N;
    const unsigned blocks = (unsigned)std::min<long long>((total + 255) / 256, 64LL * c->cus);
    hipLaunchKernelGGL(abn_make_dstar_kernel, dim3(blocks), dim3(256), 0, st, dst, a.pred, a.resid, a.idx, p->N,
                       (long long)p->B * p->N, total);
    HIPCHK(c, hipGetLastError());
    a.dmode = 2;
    a.D = dst;
  }
  // few bootstraps: latency-bound like phase A -> the speculative kernel (four wavefronts per chain)
  bool spec = a.dmode == 1 && p->opt.lanes_per_chain == 0 &&
              (long long)p->W * p->B <= spec_max_chains(c, p->N, 1) && spec_applicable(a);
  int lanes_b = p->lanes;
  // ... and up to 192 chains per packed lane (3072 for the 16-lane kernels) a wavefront per chain still beats packing
  // several chains into one (scripts/b_kernel_sweep.py, C3 topology: 2000 bootstraps 1.36 ms against 1.74 ms packed and
  // 1.81 ms speculative; 4000: 1.99 against 1.75; bundled 6-row pedigree, 8 lanes: 2000 bootstraps 1.98 against 1.90)
  if (!spec && a.dmode == 1 && p->opt.lanes_per_chain == 0 && p->lanes < kWave &&
      (long long)p->W * p->B <= (3LL * c->cus / 4) * p->lanes && tree_on_wave_ok(p->N, p->topo.chain_stride, p->tree, p->opt.strict_order))
    lanes_b = kWave;
#ifdef ABN_MEASUREMENT_KNOBS  // ABN_PHASE_B_KERNEL = spec | wide | packed
  if (const char* e = getenv("ABN_PHASE_B_KERNEL")) {
    const bool can_wide = a.dmode == 1 && p->opt.lanes_per_chain == 0 && tree_on_wave_ok(p->N, p->topo.chain_stride, p->tree, p->opt.strict_order);
    if (!strcmp(e, "spec")) spec = can_wide && spec_applicable(a);
    if (!strcmp(e, "wide")) { spec = false; if (can_wide) lanes_b = kWave; }
    if (!strcmp(e, "packed")) spec = false;
  }
#endif
  int rc = spec ? launch_fit_spec(c, a, st) : launch_fit(c, a, lanes_b, st, &kind);
  if (rc) return rc;
  if (whole) p->persist_expected[1] = kind == ABN_KERNEL_PERSISTENT ? (long long)wn * p->B : 0;
  if (timed) {
    p->last_kernels[2] = kind;
    p->last_kernels[3] = spec ? kWave : lanes_b;
  }
  if (timed) HIPCHK(c, hipEventRecord(p->ev[5], st));
  return ABN_OK;
}

// zero_skipped: bit 0 / bit 1 = clear the phase-A / phase-B skip counter and chain queue before the launch
static int plan_run_phase(abn_plan* p, int32_t phase, int zero_skipped) {
  abn_ctx* c = p->ctx;
  if (!p->windows_set) return set_err(c, ABN_ERR_STATE, "abn_plan_set_windows has not been called");
  HIPCHK(c, hipSetDevice(c->device));
  if (zero_skipped == 3) {
    HIPCHK(c, hipMemsetAsync(p->skipped.p, 0, 4 * sizeof(unsigned long long), c->stream));
  } else if (zero_skipped) {
    HIPCHK(c, hipMemsetAsync(p->skipped.p + 2 * (zero_skipped >> 1), 0, 2 * sizeof(unsigned long long), c->stream));
  }
  if (phase == 0) {
    if (p->S <= 0) return set_err(c, ABN_ERR_STATE, "plan has no starts");
    int rc = enqueue_phase_a(p, 0, p->W, c->stream, true, true);
    if (rc) return rc;
    p->phase_a_done = true;
    p->ran_a = true;
    return ABN_OK;
  }
  if (phase == 1) {
    if (p->B <= 0) return set_err(c, ABN_ERR_STATE, "plan has no bootstraps");
    if (!p->phase_a_done) return set_err(c, ABN_ERR_STATE, "phase A has not run");
    int rc = enqueue_phase_b(p, 0, p->W, c->stream, true, true);
    if (rc) return rc;
    p->ran_b = true;
    return ABN_OK;
  }
  return set_err(c, ABN_ERR_INVALID_ARG, "phase must be 0 or 1");
}

extern "C" int abn_plan_run_phase(abn_plan* p, int32_t phase) {
  if (!p) return ABN_ERR_INVALID_ARG;
  return plan_run_phase(p, phase, phase == 0 ? 1 : 2);
}

// Whole pass.  opts.window_groups > 1 cuts the plan into contiguous window groups that run A -> select -> B
// on their own HIP streams (a window's bootstraps need only that window's starts), forking from and joining
// back into the context's stream with events; results do not depend on the grouping and timing events are
// recorded for group 0.  Streams share a few in-order hardware queues (4 by default), so more than 4 groups
// serialise behind each other; see scripts/groups_bench.py for the measured effect.
extern "C" int abn_plan_run(abn_plan* p) {
  if (!p) return ABN_ERR_INVALID_ARG;
  abn_ctx* c = p->ctx;
  if (!p->windows_set) return set_err(c, ABN_ERR_STATE, "abn_plan_set_windows has not been called");
  int groups = p->opt.window_groups > 0 ? p->opt.window_groups : 1;
  groups = std::max(1, std::min(groups, p->W));
  if (groups == 1 || p->S <= 0 || p->B <= 0) {
    int rc = ABN_OK;
    if (p->S > 0) rc = plan_run_phase(p, 0, 3);  // one memset clears both skip counters
    if (rc) return rc;
    if (p->B > 0) rc = plan_run_phase(p, 1, p->S > 0 ? 0 : 2);
    return rc;
  }
  HIPCHK(c, hipSetDevice(c->device));
  while ((int)c->side.size() < groups) {
    hipStream_t st = nullptr;
    HIPCHK(c, hipStreamCreateWithFlags(&st, hipStreamNonBlocking));
    c->side.push_back(st);
  }
  if (!p->ev_fork) HIPCHK(c, hipEventCreateWithFlags(&p->ev_fork, hipEventDisableTiming));
  while ((int)p->ev_join.size() < groups) {
    hipEvent_t e = nullptr;
    HIPCHK(c, hipEventCreateWithFlags(&e, hipEventDisableTiming));
    p->ev_join.push_back(e);
  }
  HIPCHK(c, hipMemsetAsync(p->skipped.p, 0, 4 * sizeof(unsigned long long), c->stream));
  HIPCHK(c, hipEventRecord(p->ev_fork, c->stream));
  for (int g = 0; g < groups; ++g) {
    const int w0 = (int)((long long)p->W * g / groups), w1 = (int)((long long)p->W * (g + 1) / groups);
    hipStream_t st = c->side[(size_t)g];
    HIPCHK(c, hipStreamWaitEvent(st, p->ev_fork, 0));
    int rc = enqueue_phase_a(p, w0, w1 - w0, st, g == 0, false);  // groups share the plan's one queue: no
    if (!rc) rc = enqueue_phase_b(p, w0, w1 - w0, st, g == 0, false);  // persistent kernel
    if (rc) return rc;
    HIPCHK(c, hipEventRecord(p->ev_join[(size_t)g], st));
  }
  // Join only after every group has been enqueued: HIP multiplexes streams onto a few in-order hardware
  // queues, and a wait packet of the main stream placed between the launches would hold back whichever side
  // stream shares its queue (scripts/stream_overlap_test.hip: 10 ms instead of 5 ms for four 5 ms kernels).
  for (int g = 0; g < groups; ++g) HIPCHK(c, hipStreamWaitEvent(c->stream, p->ev_join[(size_t)g], 0));
  p->phase_a_done = true;
  p->ran_a = true;
  p->ran_b = true;
  return ABN_OK;
}

// A persistent launch must have finished every chain it was given: a lost FIFO entry or a chain that was parked and never
// taken up again would otherwise leave stale (or, in a bound buffer, uninitialised) rows in the tables.  Synchronises the
// stream.  Every entry point a caller can take results from runs this — abn_plan_sync (bind_raw / raw_device_ptr users:
// the torch.distributed shard runner), abn_plan_failed_windows, abn_plan_download, and through them abn_multi_sync /
// abn_multi_download.
static int verify_persistent(abn_plan* p) {
  abn_ctx* c = p->ctx;
  HIPCHK(c, hipSetDevice(c->device));
  unsigned sl[8] = {0, 0, 0, 0, 0, 0, 0, 0};
  const bool check = p->persist_expected[0] > 0 || p->persist_expected[1] > 0;
  if (check) HIPCHK(c, hipMemcpyAsync(sl, p->slice_status.p, sizeof sl, hipMemcpyDeviceToHost, c->stream));
  HIPCHK(c, hipStreamSynchronize(c->stream));
  for (int ph = 0; ph < 2 && check; ++ph) {
    if (p->persist_expected[ph] <= 0) continue;
    p->tail_handed[ph] = (long long)sl[4 * ph + 2];
    if (sl[4 * ph] != 0 || (long long)sl[4 * ph + 1] != p->persist_expected[ph])
      return set_err(c, ABN_ERR_HIP, std::string("persistent fit launch of phase ") + (ph ? "B" : "A") + " finished " +
                                         std::to_string(sl[4 * ph + 1]) + " of " + std::to_string(p->persist_expected[ph]) +
                                         " chains (error word " + std::to_string(sl[4 * ph]) + ", " + std::to_string(sl[4 * ph + 2]) +
                                         " handed to the speculative kernel, list length " + std::to_string(sl[4 * ph + 3]) +
                                         "): results are incomplete");
  }
  return ABN_OK;
}

extern "C" int abn_plan_sync(abn_plan* p) {
  if (!p) return ABN_ERR_INVALID_ARG;
  return verify_persistent(p);
}

extern "C" int abn_plan_tail_handed(abn_plan* p, int64_t* out2) {
  if (!p || !out2) return ABN_ERR_INVALID_ARG;
  const int rc = verify_persistent(p);  // synchronises the stream and reads the counts of the last persistent launches
  out2[0] = p->persist_expected[0] > 0 ? p->tail_handed[0] : 0;
  out2[1] = p->persist_expected[1] > 0 ? p->tail_handed[1] : 0;
  return rc;
}

extern "C" int abn_plan_kernel_ms(abn_plan* p, double* ms3) {
  if (!p || !ms3) return ABN_ERR_INVALID_ARG;
  abn_ctx* c = p->ctx;
  HIPCHK(c, hipStreamSynchronize(c->stream));
  ms3[0] = ms3[1] = ms3[2] = 0.0;
  float ms = 0.f;
  if (p->ran_a) {
    HIPCHK(c, hipEventElapsedTime(&ms, p->ev[0], p->ev[1]));
    ms3[0] = ms;
    HIPCHK(c, hipEventElapsedTime(&ms, p->ev[2], p->ev[3]));
    ms3[1] = ms;
  }
  if (p->ran_b) {
    HIPCHK(c, hipEventElapsedTime(&ms, p->ev[4], p->ev[5]));
    ms3[2] = ms;
  }
  return ABN_OK;
}

extern "C" int abn_plan_raw_device_ptr(abn_plan* p, void** dev_ptr) {
  if (!p || !dev_ptr) return ABN_ERR_INVALID_ARG;
  *dev_ptr = p->raw;
  return ABN_OK;
}

extern "C" int abn_plan_bind_raw(abn_plan* p, void* dev_ptr) {
  if (!p) return ABN_ERR_INVALID_ARG;
  p->raw = dev_ptr ? (double*)dev_ptr : p->raw_own.p;
  return ABN_OK;
}

extern "C" int abn_plan_download(abn_plan* p, double* models, double* pred, double* resid, double* raw,
                                 abn_fit_info* info_a, abn_fit_info* info_b, int32_t* best_start) {
  if (!p) return ABN_ERR_INVALID_ARG;
  abn_ctx* c = p->ctx;
  HIPCHK(c, hipSetDevice(c->device));
  hipStream_t s = c->stream;
  const size_t W = (size_t)p->W, B = (size_t)p->B;
  if (models) HIPCHK(c, hipMemcpyAsync(models, p->model.p, p->model.bytes(), hipMemcpyDeviceToHost, s));
  if (pred) HIPCHK(c, hipMemcpyAsync(pred, p->pred.p, p->pred.bytes(), hipMemcpyDeviceToHost, s));
  if (resid) HIPCHK(c, hipMemcpyAsync(resid, p->resid.p, p->resid.bytes(), hipMemcpyDeviceToHost, s));
  if (raw && B) HIPCHK(c, hipMemcpyAsync(raw, p->raw, W * B * 7 * sizeof(double), hipMemcpyDeviceToHost, s));
  if (info_a && p->S) HIPCHK(c, hipMemcpyAsync(info_a, p->infoA.p, p->infoA.bytes(), hipMemcpyDeviceToHost, s));
  if (info_b && B) HIPCHK(c, hipMemcpyAsync(info_b, p->infoB.p, p->infoB.bytes(), hipMemcpyDeviceToHost, s));
  std::vector<int32_t> bs;
  if (p->ran_a) {  // the selection's verdict per window, whether or not the caller asked for it
    bs.resize(W);
    HIPCHK(c, hipMemcpyAsync(bs.data(), p->best_start.p, p->best_start.bytes(), hipMemcpyDeviceToHost, s));
  }
  if (int rc = verify_persistent(p)) return rc;  // synchronises the stream
  if (best_start) {
    if (p->ran_a) std::copy(bs.begin(), bs.end(), best_start);
    else std::fill(best_start, best_start + W, 0);  // model uploaded by the caller (abn_boot_model_run)
  }
  // every buffer has been filled; windows whose starts all ended non-finite carry NaN and best_start = -1
  for (int32_t b : bs)
    if (b < 0) return set_err(c, ABN_ERR_NO_FINITE_FIT, "a window has no finite start (best_start = -1): its rows are NaN");
  return ABN_OK;
}

extern "C" int abn_plan_failed_windows(abn_plan* p, int32_t* n_failed) {
  if (!p || !n_failed) return ABN_ERR_INVALID_ARG;
  abn_ctx* c = p->ctx;
  *n_failed = 0;
  if (!p->ran_a) return verify_persistent(p);
  HIPCHK(c, hipSetDevice(c->device));
  std::vector<int32_t> bs((size_t)p->W);
  HIPCHK(c, hipMemcpyAsync(bs.data(), p->best_start.p, p->best_start.bytes(), hipMemcpyDeviceToHost, c->stream));
  if (int rc = verify_persistent(p)) return rc;  // synchronises the stream
  for (int32_t b : bs) *n_failed += b < 0 ? 1 : 0;
  return ABN_OK;
}

extern "C" int abn_plan_counters(abn_plan* p, int64_t* out5) {
  int64_t* out4 = out5;
  if (!p || !out5) return ABN_ERR_INVALID_ARG;
  abn_ctx* c = p->ctx;
  out4[0] = out4[1] = out4[2] = out4[3] = out4[4] = 0;
  std::vector<FitInfoDev> h;
  auto add = [&](const DevBuf<FitInfoDev>& b, int phase) -> int {
    if (!b.n) return ABN_OK;
    h.resize(b.n);
    unsigned long long sk = 0;
    HIPCHK(c, hipMemcpyAsync(h.data(), b.p, b.bytes(), hipMemcpyDeviceToHost, c->stream));
    HIPCHK(c, hipMemcpyAsync(&sk, p->skipped.p + 2 * phase, sizeof sk, hipMemcpyDeviceToHost, c->stream));
    HIPCHK(c, hipStreamSynchronize(c->stream));
    for (const auto& f : h) {
      out4[0] += 1;
      out4[1] += f.evals;
      out4[2] += f.iters;
    }
    out4[3 + phase] = (int64_t)sk;
    return ABN_OK;
  };
  int rc = ABN_OK;
  if (p->ran_a) rc = add(p->infoA, 0);
  if (rc) return rc;
  if (p->ran_b) rc = add(p->infoB, 1);
  return rc;
}

#ifdef ABN_STAMPS
// diagnostic build only: run phase A with in-kernel stamps, return the 8 cycle sums of chain 0
static int debug_stamps(abn_plan* p, unsigned long long* out8, bool spec) {
  abn_ctx* c = p->ctx;
  DevBuf<unsigned long long> d;
  HIPCHK(c, d.alloc(8));
  HIPCHK(c, hipMemsetAsync(d.p, 0, 64, c->stream));
  FitArgs a{};
  fill_common(p, a);
  a.dmode = 0;
  a.smode = 0;
  a.simplex0 = p->simplexA.p;
  a.C = p->S;
  a.max_iters = p->opt.max_iters_start;
  a.best = p->bestA.p;
  a.info = p->infoA.p;
  a.raw = nullptr;
  a.dbg = d.p;
  int rc = spec ? launch_fit_spec(c, a, c->stream) : launch_fit(c, a, p->lanes, c->stream);
  if (rc) return rc;
  HIPCHK(c, hipMemcpyAsync(out8, d.p, 64, hipMemcpyDeviceToHost, c->stream));
  HIPCHK(c, hipStreamSynchronize(c->stream));
  return ABN_OK;
}
extern "C" int abn_plan_debug_stamps(abn_plan* p, unsigned long long* out8) { return debug_stamps(p, out8, false); }
// the same for the speculative three-wavefront kernel (wavefront 0 of chain 0; out8[7] = iterations)
extern "C" int abn_plan_debug_stamps_spec(abn_plan* p, unsigned long long* out8) { return debug_stamps(p, out8, true); }
#endif

extern "C" int abn_plan_last_kernels(abn_plan* p, int32_t* out4) {
  if (!p || !out4) return ABN_ERR_INVALID_ARG;
  for (int k = 0; k < 4; ++k) out4[k] = p->last_kernels[k];
  return ABN_OK;
}

extern "C" int abn_plan_device_bytes(abn_plan* p, int64_t* bytes) {
  if (!p || !bytes) return ABN_ERR_INVALID_ARG;
  size_t t = 0;
  t += p->D.bytes() + p->pred.bytes() + p->resid.bytes() + p->p_uu.bytes() + p->eqp.bytes() + p->eqp_w.bytes();
  t += p->simplexA.bytes() + p->bestA.bytes() + p->model.bytes() + p->lse.bytes() + p->bestB.bytes();
  t += p->raw_own.bytes() + p->infoA.bytes() + p->infoB.bytes() + p->best_start.bytes() + p->idx.bytes();
  t += p->dstar.bytes() + p->nm_state.bytes() + p->susp_list.bytes() + p->skipped.bytes() + p->slice_buf.bytes();
  t += p->slice_status.bytes() + p->wid.bytes();
  t += p->dtopo.tri.bytes() + p->dtopo.tid.bytes();
  *bytes = (int64_t)t;
  return ABN_OK;
}

// ------------------------------------------------------------------------------------------------
// (2) ab_neutral::run and (3) boot_model::run on host buffers
// ------------------------------------------------------------------------------------------------
static void split_pedigree(const double* ped, int n, std::vector<double>& gens, std::vector<double>& d) {
  gens.resize((size_t)n * 3);
  d.resize((size_t)n);
  for (int i = 0; i < n; ++i) {
    gens[(size_t)i * 3 + 0] = ped[(size_t)i * 4 + 0];
    gens[(size_t)i * 3 + 1] = ped[(size_t)i * 4 + 1];
    gens[(size_t)i * 3 + 2] = ped[(size_t)i * 4 + 2];
    d[(size_t)i] = ped[(size_t)i * 4 + 3];
  }
}

extern "C" int abn_ab_neutral_run(abn_ctx* c, const abn_options* opts, const double* pedigree, int32_t n_rows,
                                  double p0uu, double eqp, double eqp_weight, int32_t n_starts, double* model,
                                  double* pred, double* resid, double* all_models, abn_fit_info* info, double* lse) {
  if (!c) return ABN_ERR_INVALID_ARG;
  if (!pedigree || n_rows <= 0 || n_starts <= 0 || !model || !pred || !resid)
    return set_err(c, ABN_ERR_INVALID_ARG, "null/size");
  std::vector<double> gens, d;
  split_pedigree(pedigree, n_rows, gens, d);
  abn_plan* p = nullptr;
  int rc = abn_plan_create(c, opts, gens.data(), n_rows, 1, n_starts, 0, 0, 0, &p);
  if (rc) return rc;
  rc = abn_plan_set_windows(p, d.data(), &p0uu, &eqp, &eqp_weight);
  if (!rc) rc = abn_plan_run_phase(p, 0);
  int32_t best = -1;
  bool no_fit = false;
  if (!rc) {
    rc = abn_plan_download(p, model, pred, resid, nullptr, info, nullptr, &best);
    if (rc == ABN_ERR_NO_FINITE_FIT) {  // buffers are filled (NaN model); report after the optional outputs
      no_fit = true;
      rc = ABN_OK;
    }
  }
  if (!rc && all_models) {
    hipError_t e = hipMemcpy(all_models, p->bestA.p, p->bestA.bytes(), hipMemcpyDeviceToHost);
    if (e != hipSuccess) rc = set_err(c, ABN_ERR_HIP, hipGetErrorString(e));
  }
  if (!rc && lse) {
    hipError_t e = hipMemcpy(lse, p->lse.p, p->lse.bytes(), hipMemcpyDeviceToHost);
    if (e != hipSuccess) rc = set_err(c, ABN_ERR_HIP, hipGetErrorString(e));
  }
  abn_plan_destroy(p);
  if (!rc && (no_fit || best < 0)) rc = set_err(c, ABN_ERR_NO_FINITE_FIT, abn_status_string(ABN_ERR_NO_FINITE_FIT));
  return rc;
}

extern "C" int abn_boot_model_run(abn_ctx* c, const abn_options* opts, const double* pedigree, int32_t n_rows,
                                  const double* model, const double* pred, const double* resid, double p0uu,
                                  double eqp, double eqp_weight, int32_t n_boot, double* raw, abn_fit_info* info) {
  if (!c) return ABN_ERR_INVALID_ARG;
  if (!pedigree || n_rows <= 0 || n_boot <= 0 || !model || !pred || !resid || !raw)
    return set_err(c, ABN_ERR_INVALID_ARG, "null/size");
  std::vector<double> gens, d;
  split_pedigree(pedigree, n_rows, gens, d);
  abn_plan* p = nullptr;
  int rc = abn_plan_create(c, opts, gens.data(), n_rows, 1, 0, n_boot, 0, 0, &p);
  if (rc) return rc;
  rc = plan_upload_model(p, model, pred, resid, &p0uu, &eqp, &eqp_weight);
  if (!rc) rc = abn_plan_run_phase(p, 1);
  if (!rc) rc = abn_plan_download(p, nullptr, nullptr, nullptr, raw, nullptr, info, nullptr);
  abn_plan_destroy(p);
  return rc;
}

// ------------------------------------------------------------------------------------------------
// stand-alone selection and bootstrap rows
// ------------------------------------------------------------------------------------------------
extern "C" int abn_select_best(abn_ctx* c, const double* pedigree, int32_t n_rows, double p0uu, const double* models,
                               int32_t n_models, int32_t* best_index, double* model, double* pred, double* resid,
                               double* lse) {
  if (!c) return ABN_ERR_INVALID_ARG;
  if (!pedigree || n_rows <= 0 || !models || n_models <= 0 || !best_index)
    return set_err(c, ABN_ERR_INVALID_ARG, "null/size");
  HIPCHK(c, hipSetDevice(c->device));
  PoolScope pool_scope(c);
  Topology t;
  int rc = build_topology(pedigree, n_rows, 4, t);
  if (rc) return set_err(c, rc, abn_status_string(rc));
  const size_t lds = ((size_t)kPw * t.TP + t.KP + kSelChunk) * sizeof(double);
  if (lds > kMaxDynLds) return set_err(c, ABN_ERR_INVALID_ARG, "pedigree needs more LDS than supported");
  DevTopology dt;
  rc = upload_topology(c, t, dt);
  if (rc) return rc;
  const size_t N = (size_t)n_rows, S = (size_t)n_models;
  std::vector<double> dcol(N);
  for (size_t i = 0; i < N; ++i) dcol[i] = pedigree[i * 4 + 3];
  DevBuf<double> dD, dm, dlse, dmodel, dpred, dresid, dp;
  DevBuf<FitInfoDev> dinfo;
  DevBuf<int32_t> dbest;
  HIPCHK(c, dD.alloc(N));
  HIPCHK(c, dm.alloc(S * 4));
  HIPCHK(c, dlse.alloc(S));
  HIPCHK(c, dmodel.alloc(4));
  HIPCHK(c, dpred.alloc(N));
  HIPCHK(c, dresid.alloc(N));
  HIPCHK(c, dp.alloc(1));
  HIPCHK(c, dinfo.alloc(S));
  HIPCHK(c, dbest.alloc(1));
  HIPCHK(c, hipMemcpyAsync(dD.p, dcol.data(), dD.bytes(), hipMemcpyHostToDevice, c->stream));
  HIPCHK(c, hipMemcpyAsync(dm.p, models, dm.bytes(), hipMemcpyHostToDevice, c->stream));
  HIPCHK(c, hipMemcpyAsync(dp.p, &p0uu, sizeof(double), hipMemcpyHostToDevice, c->stream));
  HIPCHK(c, hipMemsetAsync(dinfo.p, 0, dinfo.bytes(), c->stream));  // status 0: every model is a candidate
  SelectArgs a{};
  a.tri = dt.tri.p;
  a.tid = dt.tid.p;
  a.N = n_rows;
  a.K = t.K;
  a.T = t.T;
  a.TP = t.TP;
  a.p_uu = dp.p;
  a.D = dD.p;
  a.models = dm.p;
  a.info = dinfo.p;
  a.W = 1;
  a.S = n_models;
  a.lse = dlse.p;
  a.model = dmodel.p;
  a.pred = dpred.p;
  a.resid = dresid.p;
  a.best_start = dbest.p;
  HIPCHK(c, allow_lds(reinterpret_cast<const void*>(&abn_select_lse_kernel), lds));
  HIPCHK(c, allow_lds(reinterpret_cast<const void*>(&abn_select_kernel), lds));
  hipLaunchKernelGGL(abn_select_lse_kernel, dim3((unsigned)a.S), dim3(kWave), lds, c->stream, a);
  hipLaunchKernelGGL(abn_select_kernel, dim3(1), dim3(kWave), lds, c->stream, a);
  HIPCHK(c, hipGetLastError());
  HIPCHK(c, hipMemcpyAsync(best_index, dbest.p, sizeof(int32_t), hipMemcpyDeviceToHost, c->stream));
  if (model) HIPCHK(c, hipMemcpyAsync(model, dmodel.p, dmodel.bytes(), hipMemcpyDeviceToHost, c->stream));
  if (pred) HIPCHK(c, hipMemcpyAsync(pred, dpred.p, dpred.bytes(), hipMemcpyDeviceToHost, c->stream));
  if (resid) HIPCHK(c, hipMemcpyAsync(resid, dresid.p, dresid.bytes(), hipMemcpyDeviceToHost, c->stream));
  if (lse) HIPCHK(c, hipMemcpyAsync(lse, dlse.p, dlse.bytes(), hipMemcpyDeviceToHost, c->stream));
  HIPCHK(c, hipStreamSynchronize(c->stream));
  if (*best_index < 0) return set_err(c, ABN_ERR_NO_FINITE_FIT, abn_status_string(ABN_ERR_NO_FINITE_FIT));
  return ABN_OK;
}

extern "C" int abn_bootstrap_rows(abn_ctx* c, const double* best, int64_t n_boot, double* raw) {
  if (!c) return ABN_ERR_INVALID_ARG;
  if (!best || !raw || n_boot < 0) return set_err(c, ABN_ERR_INVALID_ARG, "null/size");
  if (n_boot == 0) return ABN_OK;
  HIPCHK(c, hipSetDevice(c->device));
  PoolScope pool_scope(c);
  DevBuf<double> db, dr;
  HIPCHK(c, db.alloc((size_t)n_boot * 4));
  HIPCHK(c, dr.alloc((size_t)n_boot * 7));
  HIPCHK(c, hipMemcpyAsync(db.p, best, db.bytes(), hipMemcpyHostToDevice, c->stream));
  const unsigned blocks = (unsigned)std::min<long long>((n_boot + 255) / 256, 4096);
  hipLaunchKernelGGL(abn_rows_kernel, dim3(blocks), dim3(256), 0, c->stream, db.p, (long long)n_boot, dr.p);
  HIPCHK(c, hipGetLastError());
  HIPCHK(c, hipMemcpyAsync(raw, dr.p, dr.bytes(), hipMemcpyDeviceToHost, c->stream));
  HIPCHK(c, hipStreamSynchronize(c->stream));
  return ABN_OK;
}
