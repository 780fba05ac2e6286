// Micro-benchmark: latency of getting a small host buffer in front of a kernel on MI355X.
//   (a) hipMemcpyAsync H2D (SDMA) + dependent kernel, (b) kernel reading mapped pinned memory directly,
//   (c) empty kernel launch + sync, (d) GPU->host flag write observed by a spinning CPU thread.
#include <hip/hip_runtime.h>
#include <chrono>
#include <cstdio>
#include <cstdlib>
#include <vector>
#include <immintrin.h>
#define CK(x) do { hipError_t e = (x); if (e != hipSuccess) { printf("%s: %s\n", #x, hipGetErrorString(e)); exit(1);} } while (0)
using Clock = std::chrono::steady_clock;
static double us(Clock::time_point a) { return std::chrono::duration<double, std::micro>(Clock::now() - a).count(); }

__global__ void consume(const uint4* src, uint4* dst, size_t n16, volatile unsigned* flag, unsigned val) {
  for (size_t i = blockIdx.x * blockDim.x + threadIdx.x; i < n16; i += (size_t)gridDim.x * blockDim.x) dst[i] = src[i];
  if (flag && blockIdx.x == 0 && threadIdx.x == 0) { __threadfence_system(); *flag = val; }
}
__global__ void empty(volatile unsigned* flag, unsigned val) { if (flag && threadIdx.x == 0 && blockIdx.x == 0) { __threadfence_system(); *flag = val; } }

int main() {
  hipStream_t st; CK(hipStreamCreateWithFlags(&st, hipStreamNonBlocking));
  unsigned* hflag; CK(hipHostMalloc((void**)&hflag, 64, hipHostMallocMapped)); unsigned* dflag; CK(hipHostGetDevicePointer((void**)&dflag, hflag, 0));
  const size_t maxb = 8 << 20;
  uint8_t *hbuf, *dbuf, *dbuf2; CK(hipHostMalloc((void**)&hbuf, maxb, hipHostMallocMapped)); CK(hipMalloc(&dbuf, maxb)); CK(hipMalloc(&dbuf2, maxb));
  uint8_t* hbuf_dev; CK(hipHostGetDevicePointer((void**)&hbuf_dev, hbuf, 0));
  const int reps = 200;
  unsigned seq = 0;
  // (c) empty kernel: launch -> hipStreamSynchronize ; launch -> flag spin
  for (int mode = 0; mode < 2; ++mode) {
    double tot = 0;
    for (int r = 0; r < reps; ++r) {
      auto t0 = Clock::now();
      ++seq; *hflag = 0;
      hipLaunchKernelGGL(empty, dim3(64), dim3(256), 0, st, mode ? dflag : nullptr, seq);
      if (mode) { while (*(volatile unsigned*)hflag != seq) _mm_pause(); } else CK(hipStreamSynchronize(st));
      tot += us(t0);
    }
    printf("empty kernel, %s: %.1f us\n", mode ? "flag spin" : "hipStreamSynchronize", tot / reps);
  }
  for (size_t bytes : {65536ul, 262144ul, 655360ul, 1310720ul, 2621440ul, 4194304ul}) {
    for (int mode = 0; mode < 3; ++mode) {
      double tot = 0;
      for (int r = 0; r < reps; ++r) {
        auto t0 = Clock::now();
        ++seq;
        if (mode == 0) {        // SDMA copy then kernel consuming the device copy
          CK(hipMemcpyAsync(dbuf, hbuf, bytes, hipMemcpyHostToDevice, st));
          hipLaunchKernelGGL(consume, dim3(256), dim3(256), 0, st, (const uint4*)dbuf, (uint4*)dbuf2, bytes / 16, dflag, seq);
        } else if (mode == 1) { // kernel reads the mapped host buffer directly (16 B per lane)
          hipLaunchKernelGGL(consume, dim3(256), dim3(256), 0, st, (const uint4*)hbuf_dev, (uint4*)dbuf2, bytes / 16, dflag, seq);
        } else {                // same with more threads in flight
          hipLaunchKernelGGL(consume, dim3(1024), dim3(1024), 0, st, (const uint4*)hbuf_dev, (uint4*)dbuf2, bytes / 16, dflag, seq);
        }
        while (*(volatile unsigned*)hflag != seq) _mm_pause();
        tot += us(t0);
      }
      const char* nm[] = {"memcpyAsync+kernel", "zero-copy kernel 64K thr", "zero-copy kernel 1M thr"};
      printf("%8zu B  %-26s %.1f us  (%.1f GB/s)\n", bytes, nm[mode], tot / reps, bytes / (tot / reps) / 1e3);
    }
  }
  return 0;
}
