// Development probe (GPU box): layout and rounding of v_mfma_f64_4x4x4_4b_f64 — could the 3x3 power-table
// products (k-ascending FMA chains, bit-exact with the reference) run on it?
//   hipcc --offload-arch=gfx950 -O2 -o /tmp/mfma_probe scripts/mfma_f64_probe.hip && /tmp/mfma_probe
#include <hip/hip_runtime.h>
#include <cmath>
#include <cstdio>
#include <cstdlib>
#include <cstring>
#include <vector>

__global__ void k_mfma(const double* a, const double* b, const double* c, double* d, unsigned long long* cyc, int reps) {
  const int l = threadIdx.x;
  double av = a[l], bv = b[l], cv = c[l];
  unsigned long long t0, t1;
  double r = cv;
  asm volatile("s_waitcnt vmcnt(0) lgkmcnt(0)\n\ts_memtime %0\n\ts_waitcnt lgkmcnt(0)" : "=s"(t0)::"memory");
  for (int i = 0; i < reps; ++i) r = __builtin_amdgcn_mfma_f64_4x4x4f64(av, bv, r, 0, 0, 0);
  asm volatile("s_nop 15\n\ts_memtime %0\n\ts_waitcnt lgkmcnt(0)" : "=s"(t1) : "v"(r) : "memory");
  d[l] = r;
  if (l == 0) cyc[0] = t1 - t0;
}

// dependent through SrcB (the power table's chain: R_{n+1}^T = G^T R_n^T)
__global__ void k_chain_b(const double* a, const double* b, double* d, unsigned long long* cyc, int reps) {
  const int l = threadIdx.x;
  double av = a[l], r = b[l];
  unsigned long long t0, t1;
  asm volatile("s_waitcnt vmcnt(0) lgkmcnt(0)\n\ts_memtime %0\n\ts_waitcnt lgkmcnt(0)" : "=s"(t0)::"memory");
  for (int i = 0; i < reps; ++i) r = __builtin_amdgcn_mfma_f64_4x4x4f64(av, r, 0.0, 0, 0, 0);
  asm volatile("s_nop 15\n\ts_memtime %0\n\ts_waitcnt lgkmcnt(0)" : "=s"(t1) : "v"(r) : "memory");
  d[l] = r;
  if (l == 0) cyc[0] = t1 - t0;
}

int main() {
  double *a, *b, *c, *d;
  unsigned long long* cyc;
  hipMalloc(&a, 512); hipMalloc(&b, 512); hipMalloc(&c, 512); hipMalloc(&d, 512); hipMalloc(&cyc, 8);
  std::vector<double> ha(64), hb(64), hc(64, 0.0), hd(64);
  // ---- layout: A = e(i0,k0) one-hot per block, B all distinct -> D row i0 = B row k0
  srand(1);
  auto run = [&](int reps) {
    hipMemcpy(a, ha.data(), 512, hipMemcpyHostToDevice);
    hipMemcpy(b, hb.data(), 512, hipMemcpyHostToDevice);
    hipMemcpy(c, hc.data(), 512, hipMemcpyHostToDevice);
    hipLaunchKernelGGL(k_mfma, 1, 64, 0, 0, a, b, c, d, cyc, reps);
    hipMemcpy(hd.data(), d, 512, hipMemcpyDeviceToHost);
  };
  // hypothesis (AMD matrix instruction calculator): block = lane / 16 for D; A: i = lane % 4, k = (lane / 4) % 4 ... test
  for (int l = 0; l < 64; ++l) { ha[l] = 0; hb[l] = 1000 + l; }
  printf("one-hot A probes (lane of A set to 1): nonzero D lanes -> value\n");
  for (int la : {0, 1, 4, 5, 16, 21}) {
    std::fill(ha.begin(), ha.end(), 0.0);
    ha[la] = 1.0;
    run(1);
    printf("A lane %2d:", la);
    for (int l = 0; l < 64; ++l) if (hd[l] != 0) printf(" D[%d]=%.0f", l, hd[l]);
    printf("\n");
  }
  // ---- rounding: random A, B (block 0 only meaningful once the layout is known); print D and let the host compare
  // against fma chains in both k orders for every (i,j) under the layout found above.  Layout assumption used below:
  // A[blk][i][k] at lane i + 4 blk + 16 k, B[blk][k][j] at lane j + 4 blk + 16 k, D[blk][i][j] at lane j + 4 blk + 16 i
  // (read off the one-hot probes above).
  int bad_asc = 0, bad_desc = 0, trials = 20000;
  for (int t = 0; t < trials; ++t) {
    for (int l = 0; l < 64; ++l) {
      ha[l] = (rand() / (double)RAND_MAX - 0.5) * ldexp(1.0, rand() % 40 - 20);
      hb[l] = (rand() / (double)RAND_MAX - 0.5) * ldexp(1.0, rand() % 40 - 20);
      hc[l] = 0.0;
    }
    run(1);
    for (int blk = 0; blk < 4; ++blk)
      for (int i = 0; i < 4; ++i)
        for (int j = 0; j < 4; ++j) {
          double asc = 0.0, desc = 0.0;
          for (int k = 0; k < 4; ++k) asc = fma(ha[i + 4 * blk + 16 * k], hb[j + 4 * blk + 16 * k], asc);
          for (int k = 3; k >= 0; --k) desc = fma(ha[i + 4 * blk + 16 * k], hb[j + 4 * blk + 16 * k], desc);
          const double got = hd[j + 4 * blk + 16 * i];
          if (memcmp(&got, &asc, 8)) ++bad_asc;
          if (memcmp(&got, &desc, 8)) ++bad_desc;
        }
  }
  // the case that matters: k = 3 terms are exact zeros (3x3 products padded to 4x4)
  int bad3 = 0;
  for (int t = 0; t < trials; ++t) {
    for (int l = 0; l < 64; ++l) {
      const int k = l >> 4;
      ha[l] = k == 3 ? 0.0 : (rand() / (double)RAND_MAX) * ldexp(1.0, rand() % 6 - 3);
      hb[l] = k == 3 ? 0.0 : (rand() / (double)RAND_MAX) * ldexp(1.0, rand() % 6 - 3);
      hc[l] = 0.0;
    }
    run(1);
    for (int blk = 0; blk < 4; ++blk)
      for (int i = 0; i < 4; ++i)
        for (int j = 0; j < 4; ++j) {
          double asc = 0.0;
          for (int k = 0; k < 3; ++k) asc = fma(ha[i + 4 * blk + 16 * k], hb[j + 4 * blk + 16 * k], asc);
          const double got = hd[j + 4 * blk + 16 * i];
          if (memcmp(&got, &asc, 8)) ++bad3;
        }
  }
  printf("3x3 padded: mismatches vs fma(a2,b2,fma(a1,b1,fma(a0,b0,0))): %d\n", bad3);
  printf("rounding: %d trials x 64 elements: mismatches vs k-ascending fma chain %d, vs k-descending %d\n", trials, bad_asc, bad_desc);
  // ---- special values: NaN, +-inf, +-0, denormals, huge/tiny magnitudes at random positions
  {
    const double specials[] = {0.0, -0.0, INFINITY, -INFINITY, NAN, 4.9e-324, -2.2e-308, 1e-310, 1.7e308, -1.7e308, 1e-200, 1e200};
    int bad = 0;
    for (int t = 0; t < trials; ++t) {
      for (int l = 0; l < 64; ++l) {
        ha[l] = (rand() % 4 == 0) ? specials[rand() % 12] : (rand() / (double)RAND_MAX - 0.5) * ldexp(1.0, rand() % 600 - 300);
        hb[l] = (rand() % 4 == 0) ? specials[rand() % 12] : (rand() / (double)RAND_MAX - 0.5) * ldexp(1.0, rand() % 600 - 300);
        hc[l] = 0.0;
      }
      run(1);
      for (int blk = 0; blk < 4; ++blk)
        for (int i = 0; i < 4; ++i)
          for (int j = 0; j < 4; ++j) {
            double asc = 0.0;
            for (int k = 0; k < 4; ++k) asc = fma(ha[i + 4 * blk + 16 * k], hb[j + 4 * blk + 16 * k], asc);
            const double got = hd[j + 4 * blk + 16 * i];
            const bool same = (std::isnan(got) && std::isnan(asc)) || !memcmp(&got, &asc, 8);
            if (!same) {
              if (bad < 5) printf("  special mismatch: got %a want %a\n", got, asc);
              ++bad;
            }
          }
    }
    printf("special values (NaN compared as NaN): mismatches %d\n", bad);
  }
  // ---- latency of a dependent chain of MFMAs
  for (int l = 0; l < 64; ++l) { ha[l] = 0.5; hb[l] = 0.25; hc[l] = 1.0; }
  run(1000);
  unsigned long long h;
  hipMemcpy(&h, cyc, 8, hipMemcpyDeviceToHost);
  printf("dependent v_mfma_f64_4x4x4 chain (through SrcC): %.1f cycles per instruction\n", (double)h / 1000.0);
  for (int l = 0; l < 64; ++l) { ha[l] = ((l & 3) == (l >> 4)) ? 1.0 : 0.0; hb[l] = 0.25; }
  hipMemcpy(a, ha.data(), 512, hipMemcpyHostToDevice);
  hipMemcpy(b, hb.data(), 512, hipMemcpyHostToDevice);
  hipLaunchKernelGGL(k_chain_b, 1, 64, 0, 0, a, b, d, cyc, 1000);
  hipMemcpy(&h, cyc, 8, hipMemcpyDeviceToHost);
  printf("dependent chain through SrcB: %.1f cycles per instruction\n", (double)h / 1000.0);
  return 0;
}
