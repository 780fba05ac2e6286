// lds_gather_bench.hip -- what does the LDS side of the PQ-distance stage cost, and can a layout change it?
// K2 alone (pqdist_stream_kernel) is bound by its LDS gathers: 72 `ds_read_b64` per code row, every one 64 lanes reading 64 RANDOM
// entries {p.x, p.y} of a 256-entry chunk table (profiles/r03_k2_alone.md: LDS arrays 83 % busy, 55 % of it bank conflicts).
// This bench runs the bare gather + the arithmetic of K2's inner loop on an LDS-resident [72][256] table with random codes and compares
//   v0  AoS, one ds_read_b64 per (row, chunk)                                  -- what K2 does
//   v1  SoA planes x[256] | y[256] per chunk, y displaced by 32 banks (128 B), two ds_read_b32 per (row, chunk)   -- VERDICT r3 #7 (a)
//   v2  AoS, but lane l starts at chunk (l * 9) mod 72 (a rotated schedule: the 64 lanes of an instruction read 64 different chunk tables)
//   v3  the same random codes for ALL lanes of a wave (every read a broadcast: the conflict-free floor of the same instruction stream)
// Output: ns per (row, chunk) gather per CU and G gathers/s for the chip; run under rocprofv3 --pmc SQ_LDS_BANK_CONFLICT SQ_LDS_IDX_ACTIVE
// SQ_INSTS_LDS for the conflict cycles behind them (tools/dev/run_lds_gather.sh).
// Build: hipcc --offload-arch=gfx950 -O3 -o tools/dev/lds_gather_bench tools/dev/lds_gather_bench.hip
#include <hip/hip_runtime.h>
#include <cstdint>
#include <cstdio>
#include <cstdlib>

#define CHECK(x) do { hipError_t e_ = (x); if (e_ != hipSuccess) { fprintf(stderr, "%s: %s\n", #x, hipGetErrorString(e_)); exit(1); } } while (0)
#define NC 72

__device__ __forceinline__ uint32_t mix32(uint32_t x) {
  x ^= x >> 16; x *= 0x7feb352du; x ^= x >> 15; x *= 0x846ca68bu; x ^= x >> 16;
  return x;
}

template <int V>
__global__ __launch_bounds__(768) void k_gather(const float* __restrict__ table, uint32_t rows_per_lane, float* out) {
  extern __shared__ __attribute__((aligned(16))) float lds[];
  // v0 / v2 / v3: [NC][256][2];  v1: per chunk x[256] then y[256] displaced by 32 words: [NC][512 + 32]
  const uint32_t words = (V == 1) ? NC * 544u : NC * 512u;
  for (uint32_t i = threadIdx.x; i < words; i += blockDim.x) lds[i] = table[i % (NC * 512u)];
  __syncthreads();
  const uint32_t lane = threadIdx.x & 63u;
  const uint32_t gid = blockIdx.x * blockDim.x + threadIdx.x;
  float s[8] = {0, 0, 0, 0, 0, 0, 0, 0};
  for (uint32_t r = 0; r < rows_per_lane; ++r) {
    uint32_t seed = mix32((V == 3 ? (gid >> 6) : gid) * 0x9E3779B9u + r);
    uint32_t w = 0;
#pragma unroll
    for (int c = 0; c < NC; ++c) {
      if ((c & 3) == 0) { seed = mix32(seed + 0x85EBCA6Bu); w = seed; }           // four code bytes per dword, as a code row has them
      const uint32_t code = (w >> (8 * (c & 3))) & 0xffu;
      float px, py;
      if (V == 1) {
        px = lds[(uint32_t)c * 544u + code];
        py = lds[(uint32_t)c * 544u + 256u + 32u + code];
      } else {
        uint32_t cc = (uint32_t)c;
        if (V == 2) { cc = (uint32_t)c + lane * 9u; cc -= (cc / NC) * NC; }
        const float2 p = *(const float2*)(lds + cc * 512u + code * 2u);
        px = p.x; py = p.y;
      }
      const float d0 = px - 1.5f, d1 = py + 0.25f;                                // (K2: pivot - query, then the fmaf chain)
      float t = __builtin_fmaf(d0, d0, 0.0f);
      t = __builtin_fmaf(d1, d1, t);
      s[c & 7] += t;
    }
  }
  float x = ((s[0] + s[1]) + (s[2] + s[3])) + ((s[4] + s[5]) + (s[6] + s[7]));
  if (x == 12345.678f) out[0] = x;
}

template <int V>
static void run(const char* name, const float* d_table, float* d_out, uint32_t rows_per_lane) {
  const size_t lds = (V == 1 ? NC * 544 : NC * 512) * 4;
  CHECK(hipFuncSetAttribute((const void*)k_gather<V>, hipFuncAttributeMaxDynamicSharedMemorySize, 160 * 1024));
  hipEvent_t e0, e1;
  CHECK(hipEventCreate(&e0)); CHECK(hipEventCreate(&e1));
  hipLaunchKernelGGL(k_gather<V>, dim3(256), dim3(768), lds, 0, d_table, rows_per_lane, d_out);      // warm-up
  CHECK(hipEventRecord(e0));
  hipLaunchKernelGGL(k_gather<V>, dim3(256), dim3(768), lds, 0, d_table, rows_per_lane, d_out);
  CHECK(hipEventRecord(e1));
  CHECK(hipEventSynchronize(e1));
  float ms = 0;
  CHECK(hipEventElapsedTime(&ms, e0, e1));
  const double rows = 256.0 * 768.0 * rows_per_lane;
  printf("{\"variant\": \"%s\", \"ms\": %.3f, \"rows\": %.0f, \"G_rows_per_s\": %.2f, \"ns_per_wave_gather_instruction_per_CU\": %.2f}\n", name, ms, rows,
         rows / (ms * 1e-3) / 1e9, ms * 1e6 / (12.0 * rows_per_lane * NC));
}

int main(int argc, char** argv) {
  const uint32_t rpl = argc > 1 ? (uint32_t)atoi(argv[1]) : 200u;
  float *d_table, *d_out;
  CHECK(hipMalloc(&d_table, NC * 512 * 4));
  CHECK(hipMalloc(&d_out, 64));
  float* h = (float*)malloc(NC * 512 * 4);
  for (int i = 0; i < NC * 512; ++i) h[i] = (float)((i * 2654435761u) >> 20) * 1e-3f;
  CHECK(hipMemcpy(d_table, h, NC * 512 * 4, hipMemcpyHostToDevice));
  run<0>("v0 AoS ds_read_b64 (K2 today)", d_table, d_out, rpl);
  run<1>("v1 SoA planes, y displaced by 32 banks, 2 x ds_read_b32", d_table, d_out, rpl);
  run<2>("v2 AoS, rotated chunk schedule per lane", d_table, d_out, rpl);
  run<3>("v3 AoS, one code per wave (broadcast reads: no conflicts)", d_table, d_out, rpl);
  return 0;
}
