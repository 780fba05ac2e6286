// traffic_calib.hip -- kernels that move a KNOWN number of bytes in the access shapes of the search path, to calibrate what the
// gfx950 memory-side counters (FETCH_SIZE, WRITE_SIZE, TCC_EA0_RDREQ_32B/64B/128B, ..._DRAM_32B) report for them
// (MI355X_MICROARCH.md, HBM: "other access widths are uncalibrated: calibrate on a known byte count in your own access pattern").
//   k_rows128  : random 128-byte code rows, five adjacent lanes per row asking for 16 bytes each (CoopFetch): rows x 128 B read
//   k_rows70   : the same rows at a stride of 70 bytes (packed m = 70): rows x 70 B useful, 2.1 sixty-four-byte lines touched per row
//   k_probe4   : random 4-byte loads (filter probes): one line per probe
//   k_store4   : random 4-byte stores (filter updates): one line per store
//   k_stream   : wide coalesced read, 16 B per lane (the guide's reference case: FETCH_SIZE reads half of it)
// Build: hipcc --offload-arch=gfx950 -O3 -o tools/_build/traffic_calib tools/traffic_calib.hip ; run under rocprofv3 --pmc (tools/calibrate_traffic.sh)
#include <hip/hip_runtime.h>
#include <cstdint>
#include <cstdio>
#include <cstdlib>

#define CHECK(x) do { hipError_t e_ = (x); if (e_ != hipSuccess) { fprintf(stderr, "%s: %s\n", #x, hipGetErrorString(e_)); exit(1); } } while (0)
typedef uint32_t u32x4a __attribute__((ext_vector_type(4), aligned(4)));

__device__ __forceinline__ uint64_t mix(uint64_t z) {
  z = (z ^ (z >> 30)) * 0xBF58476D1CE4E5B9ull;
  z = (z ^ (z >> 27)) * 0x94D049BB133111EBull;
  return z ^ (z >> 31);
}

// five adjacent lanes fetch one row (16 bytes each); a wave instruction covers 12 rows; `iters` instructions per wave
__global__ __launch_bounds__(256) void k_rows(const uint8_t* __restrict__ t, uint64_t rows, uint32_t stride, uint32_t iters, uint32_t* out) {
  const uint64_t wave = ((uint64_t)blockIdx.x * blockDim.x + threadIdx.x) >> 6;
  const uint32_t lane = threadIdx.x & 63u, grp = lane / 5u, sub = lane % 5u;
  uint32_t acc = 0;
  for (uint32_t it = 0; it < iters; ++it) {
    const uint64_t r = mix((wave * iters + it) * 12ull + grp + 0x9E3779B97F4A7C15ull) % rows;
    if (grp < 12u) {
      const u32x4a v = *((const u32x4a*)(t + ((r * stride) & ~3ull)) + sub);
      acc ^= v.x ^ v.y ^ v.z ^ v.w;
    }
  }
  if (acc == 0x12345678u) out[0] = acc;
}
__global__ __launch_bounds__(256) void k_probe4(const uint32_t* __restrict__ t, uint64_t words, uint32_t iters, uint32_t* out) {
  const uint64_t tid = (uint64_t)blockIdx.x * blockDim.x + threadIdx.x;
  uint32_t acc = 0;
  for (uint32_t it = 0; it < iters; ++it) acc ^= __hip_atomic_load(t + mix(tid * iters + it + 77ull) % words, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);   // sc1, as the filter probes
  if (acc == 0x12345678u) out[0] = acc;
}
__global__ __launch_bounds__(256) void k_store4(uint32_t* __restrict__ t, uint64_t words, uint32_t iters) {
  const uint64_t tid = (uint64_t)blockIdx.x * blockDim.x + threadIdx.x;
  for (uint32_t it = 0; it < iters; ++it) t[mix(tid * iters + it + 991ull) % words] = (uint32_t)tid;
}
__global__ __launch_bounds__(256) void k_stream(const uint4* __restrict__ t, uint64_t n16, uint32_t* out) {
  uint32_t acc = 0;
  for (uint64_t i = (uint64_t)blockIdx.x * blockDim.x + threadIdx.x; i < n16; i += (uint64_t)gridDim.x * blockDim.x) {
    const uint4 v = t[i];
    acc ^= v.x ^ v.y ^ v.z ^ v.w;
  }
  if (acc == 0x12345678u) out[0] = acc;
}

int main(int argc, char** argv) {
  const uint64_t table = (argc > 1 ? strtoull(argv[1], nullptr, 10) : 16ull) << 30;      // GB: far beyond the 256 MB Infinity Cache
  uint8_t* t; uint32_t* out;
  CHECK(hipMalloc(&t, table + 4096));
  CHECK(hipMalloc(&out, 64));
  CHECK(hipMemset(t, 1, table + 4096));
  const int blocks = 256 * 8, threads = 256;
  const uint64_t waves = (uint64_t)blocks * threads / 64;
  const uint32_t it_rows = 2000, it_p = 500;
  const uint64_t rows128 = table / 128, rows70 = table / 70;
  // two launches of each (the first warms the TLBs); the per-launch counters of the SECOND are the ones to read
  for (int rep = 0; rep < 2; ++rep) {
    hipLaunchKernelGGL(k_rows, dim3(blocks), dim3(threads), 0, 0, t, rows128, 128u, it_rows, out);
    hipLaunchKernelGGL(k_rows, dim3(blocks), dim3(threads), 0, 0, t, rows70, 70u, it_rows, out);
    hipLaunchKernelGGL(k_probe4, dim3(blocks), dim3(threads), 0, 0, (const uint32_t*)t, table / 4, it_p, out);
    hipLaunchKernelGGL(k_store4, dim3(blocks), dim3(threads), 0, 0, (uint32_t*)t, table / 4, it_p);
    hipLaunchKernelGGL(k_stream, dim3(blocks), dim3(threads), 0, 0, (const uint4*)t, table / 16, out);
    CHECK(hipDeviceSynchronize());
  }
  // the same probes / stores on UNCACHED device memory (is a 4-byte access still a 128-byte line when L2 does not cache the table?)
  {
    uint8_t* u = nullptr;
    const uint64_t ub = 4ull << 30;
    if (hipExtMallocWithFlags((void**)&u, ub, hipDeviceMallocUncached) == hipSuccess) {
      CHECK(hipMemset(u, 1, ub));
      for (int rep = 0; rep < 2; ++rep) {
        hipLaunchKernelGGL(k_probe4, dim3(blocks), dim3(threads), 0, 0, (const uint32_t*)u, ub / 4, it_p, out);
        hipLaunchKernelGGL(k_store4, dim3(blocks), dim3(threads), 0, 0, (uint32_t*)u, ub / 4, it_p);
        CHECK(hipDeviceSynchronize());
      }
    } else fprintf(stderr, "no uncached device memory\n");
  }
  // known counts per launch, in launch order (JSON on stdout: the summary script joins them with the counter CSV by dispatch order)
  printf("{\"table_bytes\": %llu, \"launch_order\": [\"rows128\", \"rows70\", \"probe4\", \"store4\", \"stream\"], "
         "\"rows128\": {\"rows\": %llu, \"bytes\": %llu}, \"rows70\": {\"rows\": %llu, \"useful_bytes\": %llu}, "
         "\"probe4\": {\"probes\": %llu}, \"store4\": {\"stores\": %llu}, \"stream\": {\"bytes\": %llu}}\n",
         (unsigned long long)table, (unsigned long long)(waves * it_rows * 12), (unsigned long long)(waves * it_rows * 12 * 128),
         (unsigned long long)(waves * it_rows * 12), (unsigned long long)(waves * it_rows * 12 * 70), (unsigned long long)(waves * 64 * it_p),
         (unsigned long long)(waves * 64 * it_p), (unsigned long long)table);
  return 0;
}
