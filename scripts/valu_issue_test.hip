// Development probe (GPU box): issue rate of f64 FMAs from ONE wavefront — a dependent chain against two,
// three and four interleaved independent chains (inline asm keeps the order).  Prints cycles per instruction.
//   hipcc --offload-arch=gfx950 -O3 -o /tmp/valu_issue_test scripts/valu_issue_test.hip && /tmp/valu_issue_test
#include <hip/hip_runtime.h>
#include <cstdio>

template <int CH>
__global__ void probe(double* out, unsigned long long* cyc, double m, int reps) {
  double a = out[0], b = out[1], c = out[2], d = out[3];
  unsigned long long t0, t1;
  asm volatile("s_waitcnt vmcnt(0) lgkmcnt(0)\n\ts_memtime %0\n\ts_waitcnt lgkmcnt(0)" : "=s"(t0)::"memory");
  for (int r = 0; r < reps; ++r) {
#pragma unroll
    for (int u = 0; u < 16; ++u) {
      if (CH == 1) {
        asm volatile("v_fma_f64 %0, %0, %1, %1" : "+v"(a) : "v"(m));
      } else if (CH == 2) {
        asm volatile("v_fma_f64 %0, %0, %2, %2\n\tv_fma_f64 %1, %1, %2, %2" : "+v"(a), "+v"(b) : "v"(m));
      } else if (CH == 3) {
        asm volatile("v_fma_f64 %0, %0, %3, %3\n\tv_fma_f64 %1, %1, %3, %3\n\tv_fma_f64 %2, %2, %3, %3"
                     : "+v"(a), "+v"(b), "+v"(c) : "v"(m));
      } else {
        asm volatile("v_fma_f64 %0, %0, %4, %4\n\tv_fma_f64 %1, %1, %4, %4\n\tv_fma_f64 %2, %2, %4, %4\n\t"
                     "v_fma_f64 %3, %3, %4, %4" : "+v"(a), "+v"(b), "+v"(c), "+v"(d) : "v"(m));
      }
    }
  }
  asm volatile("s_memtime %0\n\ts_waitcnt lgkmcnt(0)" : "=s"(t1)::"memory");
  out[threadIdx.x + 4] = a + b + c + d;
  if (threadIdx.x == 0) cyc[0] = t1 - t0;
}

// the same with a 32-bit move + DPP and a v_cndmask between the FMAs (what the fit kernels mix in)
__global__ void probe_mix(double* out, unsigned long long* cyc, double m, int reps) {
  double a = out[0], b = out[1];
  int x = threadIdx.x;
  unsigned long long t0, t1;
  asm volatile("s_waitcnt vmcnt(0) lgkmcnt(0)\n\ts_memtime %0\n\ts_waitcnt lgkmcnt(0)" : "=s"(t0)::"memory");
  for (int r = 0; r < reps; ++r) {
#pragma unroll
    for (int u = 0; u < 16; ++u) {
      asm volatile("v_fma_f64 %0, %0, %3, %3\n\tv_mov_b32 %2, %2\n\tv_fma_f64 %1, %1, %3, %3\n\tv_mov_b32 %2, %2"
                   : "+v"(a), "+v"(b), "+v"(x) : "v"(m));
    }
  }
  asm volatile("s_memtime %0\n\ts_waitcnt lgkmcnt(0)" : "=s"(t1)::"memory");
  out[threadIdx.x + 4] = a + b + x;
  if (threadIdx.x == 0) cyc[0] = t1 - t0;
}

int main() {
  double* out;
  unsigned long long* cyc;
  hipMalloc(&out, 1024);
  hipMalloc(&cyc, 8);
  hipMemset(out, 0, 1024);
  const int reps = 2000;
  unsigned long long h;
  for (int pass = 0; pass < 2; ++pass) {
    hipLaunchKernelGGL(probe<1>, 1, 64, 0, 0, out, cyc, 0.5, reps);
    hipMemcpy(&h, cyc, 8, hipMemcpyDeviceToHost);
    if (pass) printf("1 chain : %.2f cycles/instr\n", (double)h / (reps * 16.0));
    hipLaunchKernelGGL(probe<2>, 1, 64, 0, 0, out, cyc, 0.5, reps);
    hipMemcpy(&h, cyc, 8, hipMemcpyDeviceToHost);
    if (pass) printf("2 chains: %.2f cycles/instr\n", (double)h / (reps * 32.0));
    hipLaunchKernelGGL(probe<3>, 1, 64, 0, 0, out, cyc, 0.5, reps);
    hipMemcpy(&h, cyc, 8, hipMemcpyDeviceToHost);
    if (pass) printf("3 chains: %.2f cycles/instr\n", (double)h / (reps * 48.0));
    hipLaunchKernelGGL(probe<4>, 1, 64, 0, 0, out, cyc, 0.5, reps);
    hipMemcpy(&h, cyc, 8, hipMemcpyDeviceToHost);
    if (pass) printf("4 chains: %.2f cycles/instr\n", (double)h / (reps * 64.0));
    hipLaunchKernelGGL(probe_mix, 1, 64, 0, 0, out, cyc, 0.5, reps);
    hipMemcpy(&h, cyc, 8, hipMemcpyDeviceToHost);
    if (pass) printf("2 chains + 2 v_mov_b32: %.2f cycles/instr\n", (double)h / (reps * 64.0));
  }
  return 0;
}
