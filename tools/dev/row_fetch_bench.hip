// row_fetch_bench.hip -- how fast can MI355X fetch uniformly random PQ-code rows, by HOW the lanes ask for them?
//   per-lane : lane l fetches row r_l with NL 16-byte loads (NL wave instructions, 64 different rows each)        [what K2 does today]
//   coop     : G = NL adjacent lanes fetch ONE row with one 16-byte load each (one wave instruction covers 64 / G rows; the TA can
//              merge the lanes of a row into one request per line)
// for row strides of m bytes (packed, rows straddle lines) and padded strides (64 / 128 bytes: a row never leaves its line).
// Build: hipcc --offload-arch=gfx950 -O3 -o row_fetch_bench row_fetch_bench.hip
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

// per-lane: NL loads per lane and row, INF rows in flight per lane
template <int NL, int INF>
__global__ __launch_bounds__(1024) void k_lane(const uint8_t* __restrict__ t, uint64_t rows, uint32_t stride, uint32_t iters, uint32_t* out) {
  const uint64_t tid = (uint64_t)blockIdx.x * blockDim.x + threadIdx.x;
  uint32_t acc = 0;
  for (uint32_t it = 0; it < iters; ++it) {
    u32x4a v[INF][NL];
#pragma unroll
    for (int j = 0; j < INF; ++j) {
      const uint64_t r = mix(tid * 0x9E3779B97F4A7C15ull + (uint64_t)it * INF + j) % rows;
      const u32x4a* p = (const u32x4a*)(t + ((r * stride) & ~3ull));
#pragma unroll
      for (int d = 0; d < NL; ++d) v[j][d] = p[d];
    }
#pragma unroll
    for (int j = 0; j < INF; ++j)
#pragma unroll
      for (int d = 0; d < NL; ++d) acc ^= v[j][d].x ^ v[j][d].y ^ v[j][d].z ^ v[j][d].w;
  }
  if (acc == 0x12345678u) out[0] = acc;
}

// cooperative: G lanes per row (one 16-byte load each), INF wave instructions in flight; a wave instruction covers 64 / G rows
template <int G, int INF>
__global__ __launch_bounds__(1024) void k_coop(const uint8_t* __restrict__ t, uint64_t rows, uint32_t stride, uint32_t iters, uint32_t* out) {
  const uint64_t tid = (uint64_t)blockIdx.x * blockDim.x + threadIdx.x;
  const uint32_t lane = threadIdx.x & 63u;
  const uint32_t grp = lane / G, sub = lane % G;
  const bool on = grp < 64u / G;
  const uint64_t wave = tid >> 6;
  uint32_t acc = 0;
  for (uint32_t it = 0; it < iters; ++it) {
    u32x4a v[INF];
#pragma unroll
    for (int j = 0; j < INF; ++j) {
      const uint64_t r = mix((wave * 64 + grp) * 0x9E3779B97F4A7C15ull + (uint64_t)it * INF + j) % rows;
      const u32x4a* p = (const u32x4a*)(t + ((r * stride) & ~3ull)) + sub;
      v[j] = on ? *p : u32x4a{0, 0, 0, 0};
    }
#pragma unroll
    for (int j = 0; j < INF; ++j) acc ^= v[j].x ^ v[j].y ^ v[j].z ^ v[j].w;
  }
  if (acc == 0x12345678u) out[0] = acc;
}

template <class F>
static void timeit(const char* name, F launch, double rows_per_launch, uint32_t row_bytes, uint32_t stride, int waves) {
  hipEvent_t e0, e1;
  CHECK(hipEventCreate(&e0)); CHECK(hipEventCreate(&e1));
  launch(4u);
  CHECK(hipDeviceSynchronize());
  CHECK(hipEventRecord(e0));
  launch(64u);
  CHECK(hipEventRecord(e1));
  CHECK(hipEventSynchronize(e1));
  float ms = 0;
  CHECK(hipEventElapsedTime(&ms, e0, e1));
  printf("{\"case\": \"%s\", \"row_bytes\": %u, \"stride\": %u, \"waves_per_cu\": %d, \"ms\": %.3f, \"G_rows_per_s\": %.2f, \"useful_GBps\": %.0f}\n",
         name, row_bytes, stride, waves, ms, rows_per_launch / ms / 1e6, rows_per_launch * row_bytes / ms / 1e6);
  fflush(stdout);
}

int main() {
  // ROW_FETCH_GB=<n>: table of n GB and only the cooperative fetch of rows in their own 128-byte line -- does the table's size (the
  // reach of the address translation) bound the random row rate?
  const char* gb = getenv("ROW_FETCH_GB");
  const uint64_t big = (uint64_t)(gb ? atoi(gb) : 8) << 30;
  uint8_t* d_t; uint32_t* d_out;
  CHECK(hipMalloc(&d_t, big + 4096));
  CHECK(hipMemset(d_t, 1, big + 4096));
  CHECK(hipMalloc(&d_out, 64));
  hipDeviceProp_t prop;
  CHECK(hipGetDeviceProperties(&prop, 0));
  const int cus = prop.multiProcessorCount;
  for (int waves : {16, 12, 8}) {
    if (!gb && waves == 12) continue;
    dim3 grid(cus), block(waves * 64);
    const double lanes = (double)cus * waves * 64;
#define LANE(NL, INF, RB, ST) { const uint64_t rows = big / ST; \
      timeit("per-lane x" #NL " inflight " #INF, [&](uint32_t it) { hipLaunchKernelGGL((k_lane<NL, INF>), grid, block, 0, 0, d_t, rows, (uint32_t)ST, it, d_out); }, lanes * 64 * INF, RB, ST, waves); }
#define COOP(G, INF, RB, ST) { const uint64_t rows = big / ST; \
      timeit("coop " #G " lanes/row inflight " #INF, [&](uint32_t it) { hipLaunchKernelGGL((k_coop<G, INF>), grid, block, 0, 0, d_t, rows, (uint32_t)ST, it, d_out); }, lanes / 64 * (64 / G) * 64 * INF, RB, ST, waves); }
    if (gb) { COOP(5, 4, 70, 128) COOP(5, 8, 70, 128) COOP(8, 8, 128, 128) continue; }
    LANE(2, 2, 32, 32)  LANE(2, 4, 32, 32)  COOP(2, 4, 32, 32)  COOP(2, 8, 32, 32)
    LANE(5, 1, 70, 70)  LANE(5, 2, 70, 70)  COOP(5, 4, 70, 70)  COOP(5, 8, 70, 70)
    LANE(5, 1, 70, 128) LANE(5, 2, 70, 128) COOP(5, 4, 70, 128) COOP(5, 8, 70, 128) COOP(8, 4, 128, 128) COOP(8, 8, 128, 128)
    LANE(5, 1, 74, 74)  COOP(5, 8, 74, 74)  LANE(5, 1, 74, 128)
    LANE(4, 2, 64, 64)  COOP(4, 8, 64, 64)
  }
  return 0;
}
