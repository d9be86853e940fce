// Development aid: do kernels on two HIP streams overlap on this box?  Two single-wavefront spin kernels of
// ~5 ms each: wall time ~5 ms (overlap) or ~10 ms (serialised).
#include <hip/hip_runtime.h>
#include <chrono>
#include <cstdio>
__global__ void spin(long long cycles, int* out) {
  long long t0 = clock64();
  while (clock64() - t0 < cycles) {}
  if (threadIdx.x == 0) out[blockIdx.x] = 1;
}
int main() {
  int* d;
  hipMalloc(&d, 4096);
  hipStream_t s[4];
  for (auto& x : s) hipStreamCreateWithFlags(&x, hipStreamNonBlocking);
  const long long cyc = 12000000;  // ~5 ms at 2.4 GHz
  for (int n : {1, 2, 4}) {
    hipDeviceSynchronize();
    auto t0 = std::chrono::steady_clock::now();
    for (int i = 0; i < n; ++i) hipLaunchKernelGGL(spin, dim3(1), dim3(64), 0, s[i], cyc, d + i);
    hipDeviceSynchronize();
    double ms = std::chrono::duration<double, std::milli>(std::chrono::steady_clock::now() - t0).count();
    std::printf("%d streams: %.2f ms\n", n, ms);
  }
  // with events between streams (fork/join like abn_plan_run)
  hipEvent_t fork, join[4];
  hipEventCreateWithFlags(&fork, hipEventDisableTiming);
  for (auto& e : join) hipEventCreateWithFlags(&e, hipEventDisableTiming);
  hipStream_t mainS;
  hipStreamCreateWithFlags(&mainS, hipStreamNonBlocking);
  hipDeviceSynchronize();
  auto t0 = std::chrono::steady_clock::now();
  hipEventRecord(fork, mainS);
  for (int i = 0; i < 4; ++i) {
    hipStreamWaitEvent(s[i], fork, 0);
    hipLaunchKernelGGL(spin, dim3(1), dim3(64), 0, s[i], cyc, d + i);
    hipEventRecord(join[i], s[i]);
    hipStreamWaitEvent(mainS, join[i], 0);
  }
  hipStreamSynchronize(mainS);
  double ms = std::chrono::duration<double, std::milli>(std::chrono::steady_clock::now() - t0).count();
  std::printf("fork/join 4 streams, joins interleaved: %.2f ms\n", ms);
  hipDeviceSynchronize();
  t0 = std::chrono::steady_clock::now();
  hipEventRecord(fork, mainS);
  for (int i = 0; i < 4; ++i) {
    hipStreamWaitEvent(s[i], fork, 0);
    hipLaunchKernelGGL(spin, dim3(1), dim3(64), 0, s[i], cyc, d + i);
    hipEventRecord(join[i], s[i]);
  }
  for (int i = 0; i < 4; ++i) hipStreamWaitEvent(mainS, join[i], 0);  // joins only after every launch
  hipStreamSynchronize(mainS);
  ms = std::chrono::duration<double, std::milli>(std::chrono::steady_clock::now() - t0).count();
  std::printf("fork/join 4 streams, joins after the launches: %.2f ms\n", ms);
  return 0;
}
