// Development probe (GPU box): does a v_mfma_f64_4x4x4 issued by ONE wavefront hold back the vector-ALU instructions of
// ANOTHER wavefront on the same SIMD?  One workgroup of 512 threads = 8 wavefronts on one CU, two per SIMD.  Wavefronts
// 0..3 run a timed chain of dependent / independent f64 FMAs; wavefronts 4..7 run (mode 0) nothing, (1) the same FMA
// loop, (2) a dependent chain of matrix instructions, (3) v_permlane32_swap pairs, (4) DPP moves.
//   hipcc --offload-arch=gfx950 -O2 -o /tmp/ovl scripts/mfma_valu_overlap.hip && /tmp/ovl
#include <hip/hip_runtime.h>
#include <cstdio>

__global__ __launch_bounds__(512) void k(int mode, int reps, unsigned long long* out, double* sink) {
  const int wv = threadIdx.x >> 6;
  double a = 1.0 + threadIdx.x * 1e-9, b = 0.999999, c = 1e-9, d = a + 1, e = a + 2, f = a + 3;
  unsigned long long t0 = 0, t1 = 0;
  __syncthreads();
  if (wv < 4) {
    asm volatile("s_memtime %0\n\ts_waitcnt lgkmcnt(0)" : "=s"(t0)::"memory");
    for (int i = 0; i < reps; ++i) {  // four independent chains of dependent FMAs
      a = __builtin_fma(a, b, c);
      d = __builtin_fma(d, b, c);
      e = __builtin_fma(e, b, c);
      f = __builtin_fma(f, b, c);
    }
    asm volatile("s_nop 15\n\ts_memtime %0\n\ts_waitcnt lgkmcnt(0)" : "=s"(t1) : "v"(a), "v"(d), "v"(e), "v"(f) : "memory");
    if ((threadIdx.x & 63) == 0) out[wv] = t1 - t0;
  } else if (mode == 1) {
    for (int i = 0; i < reps; ++i) {
      a = __builtin_fma(a, b, c);
      d = __builtin_fma(d, b, c);
      e = __builtin_fma(e, b, c);
      f = __builtin_fma(f, b, c);
    }
  } else if (mode == 2) {
    for (int i = 0; i < reps; ++i) a = __builtin_amdgcn_mfma_f64_4x4x4f64(b, a, 0.0, 0, 0, 0);
  } else if (mode == 3) {
    unsigned lo = threadIdx.x, hi = threadIdx.x * 3;
    for (int i = 0; i < reps; ++i) {
      auto r = __builtin_amdgcn_permlane32_swap(lo, hi, false, false);
      lo = r[0] + 1;
      hi = r[1];
    }
    a = lo + hi;
  } else if (mode == 4) {
    int v = threadIdx.x;
    for (int i = 0; i < 4 * reps; ++i) v = __builtin_amdgcn_update_dpp(v, v, 0xB1, 0xF, 0xF, true) + 1;
    a = v;
  }
  sink[threadIdx.x] = a + d + e + f;
}

int main() {
  unsigned long long* out;
  double* sink;
  hipMalloc(&out, 64);
  hipMalloc(&sink, 512 * 8);
  const int reps = 20000;
  const char* names[] = {"idle", "same FMA loop", "dependent v_mfma_f64_4x4x4 chain", "v_permlane32_swap chain", "DPP mov chain"};
  for (int mode = 0; mode < 5; ++mode) {
    unsigned long long h[4];
    hipLaunchKernelGGL(k, 1, 512, 0, 0, mode, reps, out, sink);
    hipMemcpy(h, out, 32, hipMemcpyDeviceToHost);
    printf("partner wavefront: %-34s -> timed FMA wavefront: %.2f cycles per FMA instruction (4 chains interleaved)\n",
           names[mode], (double)h[0] / (4.0 * reps));
  }
  return 0;
}
