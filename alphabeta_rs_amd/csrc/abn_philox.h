// Philox4x32-10 counter-based RNG (Salmon, Moraes, Dror, Shaw, SC'11) and the uniform transforms the
// ABneutral inputs use.  Host and device share this header so that the start simplices (host), the
// bootstrap jitter (device, inside the fit kernel) and the bootstrap index buffer (device kernel) are
// pure functions of (seed, window, index).  The reference draws from an unseeded `thread_rng`
// (src/structs.rs:85, src/boot_model.rs:43), so stream parity with it is impossible by construction.
#pragma once
#include <stdint.h>
#include <string.h>

#if defined(__HIPCC__)
#define ABN_HD __host__ __device__ inline
#else
#define ABN_HD inline
#endif

namespace abn {

// stream tags (counter word 3)
constexpr uint32_t kTagStart = 1u;   // Model::new draws of the start simplices
constexpr uint32_t kTagJitter = 2u;  // Model::vary draws of the bootstrap simplices
constexpr uint32_t kTagIdx = 3u;     // residual-bootstrap row indices

ABN_HD void philox4x32_10(uint32_t c0, uint32_t c1, uint32_t c2, uint32_t c3, uint32_t k0, uint32_t k1,
                          uint32_t out[4]) {
#pragma unroll
  for (int r = 0; r < 10; ++r) {
    const uint64_t p0 = (uint64_t)0xD2511F53u * c0;
    const uint64_t p1 = (uint64_t)0xCD9E8D57u * c2;
    const uint32_t n0 = (uint32_t)(p1 >> 32) ^ c1 ^ k0;
    const uint32_t n1 = (uint32_t)p1;
    const uint32_t n2 = (uint32_t)(p0 >> 32) ^ c3 ^ k1;
    const uint32_t n3 = (uint32_t)p0;
    c0 = n0; c1 = n1; c2 = n2; c3 = n3;
    k0 += 0x9E3779B9u;
    k1 += 0xBB67AE85u;
  }
  out[0] = c0; out[1] = c1; out[2] = c2; out[3] = c3;
}

// 52 mantissa bits -> [0,1), the construction of rand 0.8.5's UniformFloat<f64>::sample
ABN_HD double u01_from(uint32_t lo, uint32_t hi) {
  const uint64_t u = (((uint64_t)hi << 32) | lo) >> 12;
  const uint64_t bits = 0x3FF0000000000000ull | u;
  double d;
#if defined(__HIP_DEVICE_COMPILE__)
  d = __longlong_as_double((long long)bits);
#else
  memcpy(&d, &bits, sizeof d);
#endif
  return d - 1.0;
}

ABN_HD double uniform_from(uint32_t lo, uint32_t hi, double low, double high) {
  return u01_from(lo, hi) * (high - low) + low;
}

// Model::vary's per-parameter range (src/structs.rs:104-119): U(n - 0.1|n|, n + 0.1|n|), 0 -> 0.1
ABN_HD double vary_one(double n, uint32_t lo, uint32_t hi) {
  if (n == 0.0) n = 0.1;
  const double mag = (n < 0.0 ? -n : n) * 0.1;
  double low = n - mag;
  double high = n + mag;
  if (low >= high) {
    const double t = low;
    low = high;
    high = t;
  }
  return uniform_from(lo, hi, low, high);
}

// row index in [0,n): high word of r * n
ABN_HD uint32_t index_from(uint32_t r, uint32_t n) { return (uint32_t)(((uint64_t)r * (uint64_t)n) >> 32); }

}  // namespace abn
