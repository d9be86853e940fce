// Pedigree construction and the host-side analysis: abn_pairwise_divergence*, abn_analyze.
// The scan kernels are in abn_pairwise_mx.hpp; the fit path (abn_api.hip) does not include them.
#include "abn_host.hpp"
#include "abn_pairwise_mx.hpp"

using namespace abn;

// ------------------------------------------------------------------------------------------------
// pedigree construction: pairwise divergence (src/pedigree.rs:210-261)
// ------------------------------------------------------------------------------------------------
// Exact integer Gram products on the matrix pipe (abn_pairwise_mx.hpp): any number of samples, the sample axis tiled in
// groups of 64; codes already on the device, outputs on the device (any may be null).
constexpr long long kPmxMaxJobs = 8192;  // jobs per launch: 32 KiB of packed sums each (256 MiB of `partial`)

template <bool AL4>
static hipError_t launch_pairwise_mx(int nb, bool diag, unsigned grid, hipStream_t s, const PairMxArgs& a) {
  if (!diag) {
    hipLaunchKernelGGL((abn_pairwise_mx_kernel<4, false, AL4>), dim3(grid), dim3(kPmxThreads), 0, s, a);
  } else {
    switch (nb) {
      case 1: hipLaunchKernelGGL((abn_pairwise_mx_kernel<1, true, AL4>), dim3(grid), dim3(kPmxThreads), 0, s, a); break;
      case 2: hipLaunchKernelGGL((abn_pairwise_mx_kernel<2, true, AL4>), dim3(grid), dim3(kPmxThreads), 0, s, a); break;
      case 3: hipLaunchKernelGGL((abn_pairwise_mx_kernel<3, true, AL4>), dim3(grid), dim3(kPmxThreads), 0, s, a); break;
      default: hipLaunchKernelGGL((abn_pairwise_mx_kernel<4, true, AL4>), dim3(grid), dim3(kPmxThreads), 0, s, a); break;
    }
  }
  return hipGetLastError();
}

// One family of super-pairs (the ngroups diagonal ones, or the pairs R < C), in slabs of at most kPmxMaxJobs jobs: each
// slab is a scan launch and a reduce launch that writes its pairs of the result.
static int pairwise_mx_family(abn_ctx* c, PairMxArgs a, bool diag, long long nsp, int nb, bool al4, long long cu_jobs,
                              DevBuf<unsigned long long>& partial, unsigned long long* ddiff, unsigned long long* dboth,
                              double* ddval) {
  if (nsp <= 0) return ABN_OK;
  (void)nb;
  // chunks per super-pair: enough jobs to fill the GPU (cu_jobs workgroups per CU), every wavefront at least a few K
  // steps of 64 sites, and no chunk beyond 2^30 sites (the packed 32-bit halves of a job's sums)
  const long long nk = (a.L + 63) / 64;
  long long nchunks = std::max<long long>(1, ((long long)c->cus * cu_jobs + nsp - 1) / nsp);
  nchunks = std::min<long long>(nchunks, std::max<long long>(1, nk / (4 * kPmxWaves)));
  nchunks = std::max<long long>(nchunks, (a.L >> 30) + 1);
  if (a.L == 0) nchunks = 1;
  a.nchunks = (int)nchunks;
  const long long slab = std::max<long long>(1, kPmxMaxJobs / nchunks);
  HIPCHK(c, partial.alloc((size_t)std::min(slab, nsp) * (size_t)nchunks * kPmxJobElems));
  a.partial = partial.p;
  for (long long s0 = 0; s0 < nsp; s0 += slab) {
    const long long ns = std::min(slab, nsp - s0);
    a.first = s0;
    if (a.L > 0)
      HIPCHK(c, al4 ? launch_pairwise_mx<true>(nb, diag, (unsigned)(ns * nchunks), c->stream, a)
                    : launch_pairwise_mx<false>(nb, diag, (unsigned)(ns * nchunks), c->stream, a));
    // (no sites: zero rows are summed and every pair is 0 / 0)
    hipLaunchKernelGGL(abn_pairwise_reduce_tiles_kernel, dim3((unsigned)(ns * 256)), dim3(16 * kPmxReduceGroups), 0,
                       c->stream, partial.p, a.L > 0 ? (int)nchunks : 0, a.n, a.ngroups, diag ? 1 : 0, s0, ddiff, dboth,
                       ddval);
    HIPCHK(c, hipGetLastError());
  }
  return ABN_OK;
}

static int pairwise_mx_on_device(abn_ctx* c, const uint8_t* dcodes, int n, long long L, unsigned long long* ddiff,
                                 unsigned long long* dboth, double* ddval, double* kernel_ms) {
  if (n > 65535) return set_err(c, ABN_ERR_INVALID_ARG, "too many samples");
  PairMxArgs a{};
  a.codes = dcodes;
  a.n = n;
  a.L = L;
  a.ngroups = (n + 63) / 64;
  const bool al4 = (L % 4 == 0) && ((uintptr_t)dcodes % 4 == 0);
  const int nb = a.ngroups == 1 ? (n + 15) / 16 : 4;
  DevBuf<unsigned long long> pdiag, poff;
  hipEvent_t e0 = nullptr, e1 = nullptr;
  if (kernel_ms) {
    HIPCHK(c, hipEventCreate(&e0));
    HIPCHK(c, hipEventCreate(&e1));
    HIPCHK(c, hipEventRecord(e0, c->stream));
  }
  // workgroups per CU: two (eight wavefronts streaming per CU) once the scan is long enough to pay for twice the partial
  // rows; one below (50 x 2 M sites: 28.6 against 30.5 us; 50 x 32 M: 304 against 282 us)
  long long cu_diag = (long long)n * L >= (256ll << 20) ? 2 : 1, cu_off = 1;
#ifdef ABN_MEASUREMENT_KNOBS
  if (const char* e = getenv("ABN_PMX_CU_JOBS")) cu_diag = cu_off = std::max(1, atoi(e));
#endif
  const long long g = a.ngroups;
  int rc = pairwise_mx_family(c, a, true, g, nb, al4, cu_diag, pdiag, ddiff, dboth, ddval);
  if (!rc) rc = pairwise_mx_family(c, a, false, g * (g - 1) / 2, 4, al4, cu_off, poff, ddiff, dboth, ddval);
  if (rc) return rc;
  if (kernel_ms) {
    HIPCHK(c, hipEventRecord(e1, c->stream));
    HIPCHK(c, hipEventSynchronize(e1));
    float ms = 0.f;
    HIPCHK(c, hipEventElapsedTime(&ms, e0, e1));
    *kernel_ms = ms;
    (void)hipEventDestroy(e0);
    (void)hipEventDestroy(e1);
  }
  HIPCHK(c, hipStreamSynchronize(c->stream));  // the partial rows are freed on return
  return ABN_OK;
}

extern "C" int abn_pairwise_divergence_dev(abn_ctx* c, const void* dev_codes, int32_t n_samples, int64_t n_sites,
                                           void* dev_diff, void* dev_both, void* dev_dvalue, double* kernel_ms) {
  if (!c) return ABN_ERR_INVALID_ARG;
  if (!dev_codes || n_samples <= 0 || n_sites < 0) return set_err(c, ABN_ERR_INVALID_ARG, "null/size");
  if (n_samples < 2) return ABN_OK;
  HIPCHK(c, hipSetDevice(c->device));
  PoolScope pool_scope(c);
  return pairwise_mx_on_device(c, (const uint8_t*)dev_codes, n_samples, n_sites, (unsigned long long*)dev_diff,
                            (unsigned long long*)dev_both, (double*)dev_dvalue, kernel_ms);
}

extern "C" int abn_pairwise_divergence(abn_ctx* c, const uint8_t* codes, int32_t n_samples, int64_t n_sites,
                                       uint64_t* diff, uint64_t* both, double* dvalue) {
  if (!c) return ABN_ERR_INVALID_ARG;
  if (!codes || n_samples <= 0 || n_sites < 0) return set_err(c, ABN_ERR_INVALID_ARG, "null/size");
  const size_t n = (size_t)n_samples, npairs = n * (n - 1) / 2;
  if (npairs == 0) return ABN_OK;
  HIPCHK(c, hipSetDevice(c->device));
  PoolScope pool_scope(c);
  DevBuf<uint8_t> dcodes;
  DevBuf<unsigned long long> ddiff, dboth;
  DevBuf<double> ddv;
  HIPCHK(c, dcodes.alloc(std::max<size_t>(n * (size_t)n_sites, 4)));
  HIPCHK(c, ddiff.alloc(npairs));
  HIPCHK(c, dboth.alloc(npairs));
  HIPCHK(c, ddv.alloc(npairs));
  if (n_sites > 0)
    HIPCHK(c, hipMemcpyAsync(dcodes.p, codes, n * (size_t)n_sites, hipMemcpyHostToDevice, c->stream));
  int rc = pairwise_mx_on_device(c, dcodes.p, n_samples, n_sites, ddiff.p, dboth.p, ddv.p, nullptr);
  if (rc) return rc;
  if (diff) HIPCHK(c, hipMemcpyAsync(diff, ddiff.p, ddiff.bytes(), hipMemcpyDeviceToHost, c->stream));
  if (both) HIPCHK(c, hipMemcpyAsync(both, dboth.p, dboth.bytes(), hipMemcpyDeviceToHost, c->stream));
  if (dvalue) HIPCHK(c, hipMemcpyAsync(dvalue, ddv.p, ddv.bytes(), hipMemcpyDeviceToHost, c->stream));
  HIPCHK(c, hipStreamSynchronize(c->stream));
  return ABN_OK;
}

// ------------------------------------------------------------------------------------------------
// src/analysis.rs:50-98 on the host (ndarray mean / Welford std with mul_add, ndarray-stats Linear CI)
// ------------------------------------------------------------------------------------------------
extern "C" int abn_analyze(const double* raw, int64_t n_boot, double* out32) {
  if (!raw || !out32 || n_boot <= 0) return ABN_ERR_INVALID_ARG;
  const size_t B = (size_t)n_boot;
  std::vector<double> col(B), sorted(B);
  static const int src_col[8] = {0, 1, -1, 2, 3, 4, 5, 6};
  for (int k = 0; k < 8; ++k) {
    const int cidx = src_col[k];
    for (size_t i = 0; i < B; ++i)
      col[i] = cidx < 0 ? raw[7 * i + 1] / raw[7 * i + 0] : raw[7 * i + (size_t)cidx];  // beta / alpha, :54
    double mean;
    if (cidx < 0) {  // contiguous Array1 -> ndarray's eight-lane unrolled fold
      double part[8] = {0, 0, 0, 0, 0, 0, 0, 0};
      size_t i = 0;
      for (; i + 8 <= B; i += 8)
        for (int q = 0; q < 8; ++q) part[q] = part[q] + col[i + (size_t)q];
      double acc = 0.0;
      acc = acc + (part[0] + part[4]);
      acc = acc + (part[1] + part[5]);
      acc = acc + (part[2] + part[6]);
      acc = acc + (part[3] + part[7]);
      for (; i < B; ++i) acc = acc + col[i];
      mean = acc / (double)B;
    } else {  // strided column view -> plain fold
      double acc = 0.0;
      for (size_t i = 0; i < B; ++i) acc = acc + col[i];
      mean = acc / (double)B;
    }
    double wmean = 0.0, sum_sq = 0.0;
    for (size_t i = 0; i < B; ++i) {
      const double delta = col[i] - wmean;
      wmean = wmean + delta / (double)(i + 1);
      sum_sq = std::fma(col[i] - wmean, delta, sum_sq);
    }
    const double sd = std::sqrt(sum_sq / ((double)B - 1.0));
    sorted = col;
    std::sort(sorted.begin(), sorted.end());
    const double qs[2] = {0.025, 0.975};
    double ci[2];
    for (int q = 0; q < 2; ++q) {
      const double fi = qs[q] * (double)(B - 1);
      const size_t lo = (size_t)std::floor(fi), hi = (size_t)std::ceil(fi);
      const double frac = fi - std::trunc(fi);
      ci[q] = sorted[lo] + frac * (sorted[hi] - sorted[lo]);
    }
    out32[k] = mean;
    out32[8 + k] = sd;
    out32[16 + k] = ci[0];
    out32[24 + k] = ci[1];
  }
  return ABN_OK;
}
