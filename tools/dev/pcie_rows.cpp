// How fast can waves read 260-byte rows from mapped pinned host memory?  (a) 4 B per lane (65 dwords), (b) 16 B per lane (17 lanes)
#include <hip/hip_runtime.h>
#include <chrono>
#include <cstdio>
#include <cstdlib>
#include <immintrin.h>
#define CK(x) do { hipError_t e = (x); if (e != hipSuccess) { printf("%s: %s\n", #x, hipGetErrorString(e)); exit(1);} } while (0)
using Clock = std::chrono::steady_clock;
static double us(Clock::time_point a) { return std::chrono::duration<double, std::micro>(Clock::now() - a).count(); }
typedef uint32_t u32x4a __attribute__((ext_vector_type(4), aligned(4)));

template <int MODE, int NQW>
__global__ void rows(const uint32_t* src, uint32_t* out, int nrows, volatile unsigned* flag, unsigned val) {
  const int lane = threadIdx.x & 63;
  const int wave = (blockIdx.x * blockDim.x + threadIdx.x) >> 6;
  const int nw = (gridDim.x * blockDim.x) >> 6;
  uint32_t acc = 0;
  for (int r0 = wave; r0 < nrows; r0 += nw * NQW) {
#pragma unroll
    for (int u = 0; u < NQW; ++u) {
      int r = r0 + u * nw; if (r >= nrows) r = r0;
      const uint32_t* row = src + (size_t)r * 65;
      if (MODE == 0) { acc += row[0] + row[1 + lane]; }
      else { if (lane < 17) { u32x4a v = *(const u32x4a*)(row + 4 * lane - (lane == 16 ? 3 : 0)); acc += v.x + v.y + v.z + v.w; } }
    }
  }
  out[blockIdx.x * blockDim.x + threadIdx.x] = acc;
  __syncthreads();
  if (flag && blockIdx.x == 0 && threadIdx.x == 0) { __threadfence_system(); *flag = val; }
}

int main() {
  hipStream_t st; CK(hipStreamCreateWithFlags(&st, hipStreamNonBlocking));
  unsigned* hflag; CK(hipHostMalloc((void**)&hflag, 64, hipHostMallocMapped)); unsigned* dflag; CK(hipHostGetDevicePointer((void**)&dflag, hflag, 0));
  const int maxrows = 10000;
  uint32_t *hbuf, *hdev, *dbuf, *dout; CK(hipHostMalloc((void**)&hbuf, maxrows * 260 + 64, hipHostMallocMapped)); CK(hipHostGetDevicePointer((void**)&hdev, hbuf, 0));
  CK(hipMalloc(&dbuf, maxrows * 260 + 64)); CK(hipMalloc(&dout, 1 << 22));
  unsigned seq = 0; const int reps = 200;
  for (int nrows : {1250, 2500, 10000}) {
    for (int where = 0; where < 2; ++where) {
      const uint32_t* src = where ? hdev : dbuf;
      for (int cfg = 0; cfg < 4; ++cfg) {
        double tot = 0;
        for (int r = 0; r < reps; ++r) {
          auto t0 = Clock::now(); ++seq;
          const int wgs = 64, thr = 1024;
          if (cfg == 0) hipLaunchKernelGGL((rows<0, 1>), dim3(wgs), dim3(thr), 0, st, src, dout, nrows, dflag, seq);
          if (cfg == 1) hipLaunchKernelGGL((rows<0, 4>), dim3(wgs), dim3(thr), 0, st, src, dout, nrows, dflag, seq);
          if (cfg == 2) hipLaunchKernelGGL((rows<1, 1>), dim3(wgs), dim3(thr), 0, st, src, dout, nrows, dflag, seq);
          if (cfg == 3) hipLaunchKernelGGL((rows<1, 4>), dim3(wgs), dim3(thr), 0, st, src, dout, nrows, dflag, seq);
          while (*(volatile unsigned*)hflag != seq) _mm_pause();
          tot += us(t0);
        }
        const char* nm[] = {"4B/lane NQW1", "4B/lane NQW4", "16B/lane NQW1", "16B/lane NQW4"};
        printf("%5d rows from %-6s %-14s %.1f us\n", nrows, where ? "HOST" : "device", nm[cfg], tot / reps);
      }
    }
  }
  return 0;
}
