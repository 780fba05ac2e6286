// rand_sector_bench.hip -- what the MI355X memory system sustains for the access shapes of the search path:
//   uniformly random rows of ROW bytes (4-byte filter probes, 32/70/74-byte PQ code rows) from a table far larger than the
//   256 MB Infinity Cache (or small enough to live in it), read-only or read-modify-write (plain store of a dword per row),
//   with INFLIGHT independent rows in flight per lane and WAVES waves per CU.
// Prints requests/s (one row = one request stream) and the useful GB/s.  Build: hipcc --offload-arch=gfx950 -O3 -o rand_sector_bench rand_sector_bench.hip
#include <hip/hip_runtime.h>
#include <cstdint>
#include <cstdio>
#include <cstdlib>
#include <vector>

#define CHECK(x) do { hipError_t e_ = (x); if (e_ != hipSuccess) { fprintf(stderr, "%s: %s\n", #x, hipGetErrorString(e_)); exit(1); } } while (0)

__device__ __forceinline__ uint64_t mix(uint64_t z) {
  z = (z ^ (z >> 30)) * 0xBF58476D1CE4E5B9ull;
  z = (z ^ (z >> 27)) * 0x94D049BB133111EBull;
  return z ^ (z >> 31);
}

template <int DW, int INFLIGHT, bool WRITE, bool BYPASS>
__global__ __launch_bounds__(1024) void k(uint32_t* __restrict__ table, uint64_t rows, uint32_t row_bytes, uint32_t iters, uint32_t* out) {
  const uint64_t tid = (uint64_t)blockIdx.x * blockDim.x + threadIdx.x;
  uint32_t acc = 0;
  for (uint32_t it = 0; it < iters; ++it) {
    uint32_t v[INFLIGHT][DW];
    uint64_t off[INFLIGHT];
#pragma unroll
    for (int j = 0; j < INFLIGHT; ++j) {
      const uint64_t r = mix(tid * 0x9E3779B97F4A7C15ull + (uint64_t)it * INFLIGHT + j) % rows;
      off[j] = (r * row_bytes) >> 2;                       // dword index of the row start (rows may be unaligned: rounded down)
#pragma unroll
      for (int d = 0; d < DW; ++d)
        v[j][d] = BYPASS ? __hip_atomic_load(&table[off[j] + d], __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT) : table[off[j] + d];
    }
#pragma unroll
    for (int j = 0; j < INFLIGHT; ++j) {
#pragma unroll
      for (int d = 0; d < DW; ++d) acc ^= v[j][d];
      if (WRITE) table[off[j]] = v[j][0] | (1u << (it & 31));
    }
  }
  if (acc == 0x12345678u) out[0] = acc;
}

// fetch-OR with return: ONE request per probe that both reads the old word and sets the bit (vs read + plain store above)
template <int INFLIGHT, bool RETURN>
__global__ __launch_bounds__(1024) void ka(uint32_t* __restrict__ table, uint64_t rows, uint32_t iters, uint32_t* out) {
  const uint64_t tid = (uint64_t)blockIdx.x * blockDim.x + threadIdx.x;
  uint32_t acc = 0;
  for (uint32_t it = 0; it < iters; ++it) {
    uint32_t v[INFLIGHT];
#pragma unroll
    for (int j = 0; j < INFLIGHT; ++j) {
      const uint64_t r = mix(tid * 0x9E3779B97F4A7C15ull + (uint64_t)it * INFLIGHT + j) % rows;
      if (RETURN) v[j] = __hip_atomic_fetch_or(&table[r], 1u << (it & 31), __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
      else { __hip_atomic_fetch_or(&table[r], 1u << (it & 31), __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT); v[j] = 0; }
    }
#pragma unroll
    for (int j = 0; j < INFLIGHT; ++j) acc ^= v[j];
  }
  if (acc == 0x12345678u) out[0] = acc;
}

template <int INFLIGHT, bool RETURN>
static void run_atomic(const char* name, uint32_t* d_table, uint64_t table_bytes, int waves, uint32_t* d_out) {
  const uint64_t rows = (table_bytes - 512) / 4;
  const uint32_t iters = 64;
  hipDeviceProp_t prop;
  CHECK(hipGetDeviceProperties(&prop, 0));
  const int cus = prop.multiProcessorCount;
  dim3 grid(cus), block(waves * 64);
  hipEvent_t e0, e1;
  CHECK(hipEventCreate(&e0)); CHECK(hipEventCreate(&e1));
  hipLaunchKernelGGL((ka<INFLIGHT, RETURN>), grid, block, 0, 0, d_table, rows, 4u, d_out);
  CHECK(hipDeviceSynchronize());
  CHECK(hipEventRecord(e0));
  hipLaunchKernelGGL((ka<INFLIGHT, RETURN>), grid, block, 0, 0, d_table, rows, iters, d_out);
  CHECK(hipEventRecord(e1));
  CHECK(hipEventSynchronize(e1));
  float ms = 0;
  CHECK(hipEventElapsedTime(&ms, e0, e1));
  const double n = (double)cus * waves * 64 * iters * INFLIGHT;
  printf("{\"case\": \"%s\", \"table_MB\": %.0f, \"row_bytes\": 4, \"inflight\": %d, \"waves_per_cu\": %d, \"atomic_return\": %d, "
         "\"ms\": %.3f, \"G_rows_per_s\": %.2f}\n", name, table_bytes / 1e6, INFLIGHT, waves, (int)RETURN, ms, n / ms / 1e6);
}

template <int DW, int INFLIGHT, bool WRITE, bool BYPASS>
static void run(const char* name, uint32_t* d_table, uint64_t table_bytes, uint32_t row_bytes, int waves, uint32_t* d_out) {
  const uint64_t rows = (table_bytes - 512) / row_bytes;
  const uint32_t iters = 64;
  int cus = 256;
  hipDeviceProp_t prop;
  CHECK(hipGetDeviceProperties(&prop, 0));
  cus = prop.multiProcessorCount;
  dim3 grid(cus), block(waves * 64);
  hipEvent_t e0, e1;
  CHECK(hipEventCreate(&e0)); CHECK(hipEventCreate(&e1));
  hipLaunchKernelGGL((k<DW, INFLIGHT, WRITE, BYPASS>), grid, block, 0, 0, d_table, rows, row_bytes, 4u, d_out);
  CHECK(hipDeviceSynchronize());
  CHECK(hipEventRecord(e0));
  hipLaunchKernelGGL((k<DW, INFLIGHT, WRITE, BYPASS>), grid, block, 0, 0, d_table, rows, row_bytes, iters, d_out);
  CHECK(hipEventRecord(e1));
  CHECK(hipEventSynchronize(e1));
  float ms = 0;
  CHECK(hipEventElapsedTime(&ms, e0, e1));
  const double n = (double)cus * waves * 64 * iters * INFLIGHT;
  printf("{\"case\": \"%s\", \"table_MB\": %.0f, \"row_bytes\": %u, \"dwords_read\": %d, \"inflight\": %d, \"waves_per_cu\": %d, \"write\": %d, "
         "\"l1_bypass\": %d, \"ms\": %.3f, \"G_rows_per_s\": %.2f, \"useful_GBps\": %.0f}\n",
         name, table_bytes / 1e6, row_bytes, DW, INFLIGHT, waves, (int)WRITE, (int)BYPASS, ms, n / ms / 1e6, n * row_bytes / ms / 1e6);
}

int main(int argc, char** argv) {
  const uint64_t big = (uint64_t)4 << 30, mid = (uint64_t)200 << 20, small = (uint64_t)24 << 20;
  uint32_t *d_table, *d_out;
  CHECK(hipMalloc(&d_table, big));
  CHECK(hipMemset(d_table, 1, big));
  CHECK(hipMalloc(&d_out, 64));
  for (uint64_t tb : {big, mid, small}) {
    for (int waves : {16, 8}) {
      run<1, 4, false, true>("probe 4 B, read only", d_table, tb, 4, waves, d_out);
      run<1, 4, true, true>("probe 4 B + plain store", d_table, tb, 4, waves, d_out);
      run<1, 8, false, true>("probe 4 B, read only", d_table, tb, 4, waves, d_out);
      run_atomic<4, true>("probe 4 B, fetch-OR with return", d_table, tb, waves, d_out);
      run_atomic<8, true>("probe 4 B, fetch-OR with return", d_table, tb, waves, d_out);
      run_atomic<4, false>("probe 4 B, OR without return", d_table, tb, waves, d_out);
    }
    run<8, 2, false, false>("code row 32 B", d_table, tb, 32, 16, d_out);
    run<18, 1, false, false>("code row 70 B", d_table, tb, 70, 16, d_out);
    run<19, 1, false, false>("code row 74 B", d_table, tb, 74, 16, d_out);
    run<16, 2, false, false>("64-B sector", d_table, tb, 64, 16, d_out);
  }
  return 0;
}
