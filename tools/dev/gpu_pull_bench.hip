// gpu_pull_bench.hip -- can the GPU fetch adjacency rows from HOST memory by itself (zero-copy reads over PCIe), and how fast?
//   rows of ROW bytes at a stride of STRIDE bytes in a pinned host table (hipHostMalloc, or mmap + hipHostRegister), one row per
//   wave-load (64 lanes x 4 B = 256 B), DEPTH independent rows in flight per wave, W waves per CU.
//   Prints rows/s, useful GB/s and (DEPTH = 1, dependent chain) the round-trip latency.
// Build: hipcc --offload-arch=gfx950 -O3 -o gpu_pull_bench gpu_pull_bench.hip
#include <hip/hip_runtime.h>
#include <sys/mman.h>
#include <cstdint>
#include <cstdio>
#include <cstdlib>
#include <cstring>

#define CHECK(x) do { hipError_t e_ = (x); if (e_ != hipSuccess) { fprintf(stderr, "%s: %s\n", #x, hipGetErrorString(e_)); exit(1); } } while (0)

__device__ __forceinline__ uint64_t mix(uint64_t z) {
  z = (z ^ (z >> 30)) * 0xBF58476D1CE4E5B9ull;
  z = (z ^ (z >> 27)) * 0x94D049BB133111EBull;
  return z ^ (z >> 31);
}

template <int DEPTH, bool CHAIN>
__global__ __launch_bounds__(1024) void pull(const uint32_t* __restrict__ table, uint64_t rows, uint64_t stride_words, uint32_t off_words,
                                             uint32_t iters, uint32_t* out) {
  const uint32_t lane = threadIdx.x & 63;
  const uint64_t wave = ((uint64_t)blockIdx.x * blockDim.x + threadIdx.x) >> 6;
  uint32_t acc = 0;
  uint64_t r = mix(wave + 1) % rows;
  for (uint32_t it = 0; it < iters; ++it) {
    uint32_t v[DEPTH];
#pragma unroll
    for (int j = 0; j < DEPTH; ++j) {
      const uint64_t rr = CHAIN ? r : mix(wave * 0x9E3779B97F4A7C15ull + (uint64_t)it * DEPTH + j) % rows;
      v[j] = __builtin_nontemporal_load(&table[rr * stride_words + off_words + lane]);
    }
#pragma unroll
    for (int j = 0; j < DEPTH; ++j) acc ^= v[j];
    if (CHAIN) r = mix(r + __shfl(v[0], 0) + it) % rows;     // the next row depends on the data of this one
  }
  if (acc == 0x12345678u) out[0] = acc;
}

template <int DEPTH, bool CHAIN>
static void run(const char* what, const uint32_t* d_table, uint64_t rows, uint64_t stride, uint32_t off, int waves, int cus_used, uint32_t iters, uint32_t* d_out) {
  dim3 grid(cus_used), block(waves * 64);
  hipEvent_t e0, e1;
  CHECK(hipEventCreate(&e0)); CHECK(hipEventCreate(&e1));
  hipLaunchKernelGGL((pull<DEPTH, CHAIN>), grid, block, 0, 0, d_table, rows, stride / 4, off / 4, 2u, d_out);
  CHECK(hipDeviceSynchronize());
  CHECK(hipEventRecord(e0));
  hipLaunchKernelGGL((pull<DEPTH, CHAIN>), grid, block, 0, 0, d_table, rows, stride / 4, off / 4, iters, d_out);
  CHECK(hipEventRecord(e1));
  CHECK(hipEventSynchronize(e1));
  float ms = 0;
  CHECK(hipEventElapsedTime(&ms, e0, e1));
  const double n = (double)cus_used * waves * iters * DEPTH;
  printf("{\"mem\": \"%s\", \"stride\": %llu, \"row_offset\": %u, \"depth\": %d, \"chain\": %d, \"waves_per_cu\": %d, \"cus\": %d, \"ms\": %.3f, "
         "\"M_rows_per_s\": %.1f, \"useful_GBps\": %.1f, \"us_per_dependent_row\": %.2f}\n", what, (unsigned long long)stride, off, DEPTH, (int)CHAIN,
         waves, cus_used, ms, n / ms / 1e3, n * 256 / ms / 1e6, CHAIN ? ms * 1e3 / iters : 0.0);
  fflush(stdout);
}

int main(int argc, char** argv) {
  const uint64_t bytes = (argc > 1 ? strtoull(argv[1], nullptr, 10) : 8ull) << 30;
  hipDeviceProp_t prop;
  CHECK(hipGetDeviceProperties(&prop, 0));
  const int cus = prop.multiProcessorCount;
  uint32_t* d_out;
  CHECK(hipMalloc(&d_out, 64));
  for (int mode = 0; mode < 2; ++mode) {
    uint32_t* h = nullptr;
    const uint32_t* d = nullptr;
    const char* what;
    if (mode == 0) {
      what = "hipHostMalloc";
      CHECK(hipHostMalloc((void**)&h, bytes, hipHostMallocMapped));
    } else {
      what = "mmap + hipHostRegister";
      h = (uint32_t*)mmap(nullptr, bytes, PROT_READ | PROT_WRITE, MAP_PRIVATE | MAP_ANONYMOUS | MAP_POPULATE, -1, 0);
      if (h == MAP_FAILED) { perror("mmap"); return 1; }
      timespec t0, t1;
      clock_gettime(CLOCK_MONOTONIC, &t0);
      hipError_t e = hipHostRegister(h, bytes, hipHostRegisterMapped);
      clock_gettime(CLOCK_MONOTONIC, &t1);
      if (e != hipSuccess) { printf("{\"mem\": \"%s\", \"error\": \"%s\"}\n", what, hipGetErrorString(e)); continue; }
      printf("{\"mem\": \"%s\", \"register_s_per_GB\": %.3f}\n", what, ((t1.tv_sec - t0.tv_sec) + 1e-9 * (t1.tv_nsec - t0.tv_nsec)) / (bytes / 1073741824.0));
    }
    memset(h, 1, bytes);
    void* dp = nullptr;
    CHECK(hipHostGetDevicePointer(&dp, h, 0));
    d = (const uint32_t*)dp;
    // (a) 256-byte aligned adjacency rows (stride 256); (b) rows inside 388-byte graph entries (offset 128: straddle lines)
    for (int layout = 0; layout < 2; ++layout) {
      const uint64_t stride = layout == 0 ? 256 : 388;
      const uint32_t off = layout == 0 ? 0 : 128;
      const uint64_t rows = (bytes - 1024) / stride;
      run<1, true>(what, d, rows, stride, off, 1, 1, 2000, d_out);          // latency: one wave, dependent chain
      run<1, true>(what, d, rows, stride, off, 16, cus, 200, d_out);        // the search kernel's shape: 4096 waves, one row each
      run<1, true>(what, d, rows, stride, off, 8, cus, 200, d_out);
      run<1, true>(what, d, rows, stride, off, 4, cus, 200, d_out);
      run<4, false>(what, d, rows, stride, off, 16, cus, 100, d_out);       // throughput ceiling
      run<8, false>(what, d, rows, stride, off, 16, cus, 100, d_out);
    }
    if (mode == 0) CHECK(hipHostFree(h)); else { CHECK(hipHostUnregister(h)); munmap(h, bytes); }
  }
  return 0;
}
