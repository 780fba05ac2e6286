// request_ceiling.hip -- the rate at which an MI355X turns RANDOM requests that miss L2 around, in the mix the search path issues them.
//
// Measurement infrastructure of bench.py (tools/bench_legs/ceiling.py), not part of libbang.  Why it exists: per distance evaluation the
// search kernel (csrc/bang_search.hip; reference stages neighbor_filtering_new bang_search.cu:1140-1165 + compute_neighborDist_par
// :1201-1241) sends ~3.8 requests past L2 -- one 128-byte line for the PQ code row, ~1 line for a visited-filter word it has to read,
// ~1.8 scattered 4-byte stores of filter words -- and NONE of them is ever re-used from L2.  The byte roofline (m + 8 bytes per
// evaluation against 8 TB/s) says nothing about such a stream; what bounds it is how many such requests per second the L2 -> fabric ->
// Infinity Cache / HBM path completes.  This kernel issues exactly that mix with nothing else to do:
//   per lane and trip:  CODE  x one 16-byte load from a random 128-byte line of a table far larger than the Infinity Cache
//                       PROBE x one 4-byte load (past L1: sc1, as the filter probes) from a random word of a FILTER-sized table
//                       STORE x one plain 4-byte store to a random word of the same table
// with 4 trips' loads in flight per lane.  Output: requests per second (every load and every store is one request past L2: the tables are
// random-access and far beyond the 4 MB of L2 per XCD).
//
// Build: hipcc --offload-arch=gfx950 -O3 -shared -fPIC -o tools/_build/librequest_ceiling.so tools/request_ceiling.hip
#include <hip/hip_runtime.h>
#include <stdint.h>

typedef uint32_t u32x4 __attribute__((ext_vector_type(4)));

__device__ __forceinline__ uint64_t mix64(uint64_t z) {
  z = (z ^ (z >> 30)) * 0xBF58476D1CE4E5B9ull;
  z = (z ^ (z >> 27)) * 0x94D049BB133111EBull;
  return z ^ (z >> 31);
}

template <int CODE, int PROBE, int STORE>
__global__ __launch_bounds__(1024) void mix_kernel(const uint8_t* __restrict__ codes, uint64_t code_lines, uint32_t* __restrict__ filt,
                                                   uint64_t filt_words, uint32_t trips, uint32_t* out) {
  const uint64_t tid = (uint64_t)blockIdx.x * blockDim.x + threadIdx.x;
  uint32_t acc = 0;
  constexpr int FL = 4;                                    // trips in flight
  for (uint32_t t0 = 0; t0 < trips; t0 += FL) {
    u32x4 c[FL][CODE > 0 ? CODE : 1];
    uint32_t p[FL][PROBE > 0 ? PROBE : 1];
#pragma unroll
    for (int f = 0; f < FL; ++f) {
      uint64_t r = mix64(tid * 0x9E3779B97F4A7C15ull + (uint64_t)(t0 + f));
#pragma unroll
      for (int i = 0; i < CODE; ++i) { c[f][i] = *(const u32x4*)(codes + (r % code_lines) * 128u + 16u * ((uint32_t)(r >> 40) & 7u)); r = mix64(r); }
#pragma unroll
      for (int i = 0; i < PROBE; ++i) { p[f][i] = __hip_atomic_load(filt + r % filt_words, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT); r = mix64(r); }
    }
#pragma unroll
    for (int f = 0; f < FL; ++f) {
      uint64_t r = mix64(tid * 0xD1B54A32D192ED03ull + (uint64_t)(t0 + f));
#pragma unroll
      for (int i = 0; i < CODE; ++i) acc ^= c[f][i].x ^ c[f][i].w;
#pragma unroll
      for (int i = 0; i < PROBE; ++i) acc ^= p[f][i];
#pragma unroll
      for (int i = 0; i < STORE; ++i) { filt[r % filt_words] = (uint32_t)r | 1u; r = mix64(r); }
    }
  }
  if (acc == 0x12345678u) out[0] = acc;                    // (keeps the loads alive)
}

template <int CODE, int PROBE, int STORE>
static int run(const uint8_t* codes, uint64_t code_lines, uint32_t* filt, uint64_t filt_words, uint32_t trips, int waves_per_cu, int cus,
               uint32_t* d_out, double* g_per_s, double* ms_out) {
  hipEvent_t e0, e1;
  if (hipEventCreate(&e0) != hipSuccess || hipEventCreate(&e1) != hipSuccess) return 1;
  const dim3 grid(cus), block(64 * waves_per_cu);
  float best = 1e30f;
  for (int rep = 0; rep < 4; ++rep) {                      // (the first launch is the warm-up)
    (void)hipEventRecord(e0, 0);
    hipLaunchKernelGGL((mix_kernel<CODE, PROBE, STORE>), grid, block, 0, 0, codes, code_lines, filt, filt_words, trips, d_out);
    (void)hipEventRecord(e1, 0);
    if (hipEventSynchronize(e1) != hipSuccess) return 2;
    float ms = 0;
    (void)hipEventElapsedTime(&ms, e0, e1);
    if (rep > 0 && ms < best) best = ms;
  }
  (void)hipEventDestroy(e0); (void)hipEventDestroy(e1);
  const double reqs = (double)cus * 64.0 * waves_per_cu * trips * (CODE + PROBE + STORE);
  *g_per_s = reqs / (best * 1e-3) / 1e9;
  *ms_out = best;
  return 0;
}

// mix: 0 = probes only (1 load), 1 = the search kernel's mix (1 code line + 1 probe + 2 stores per trip), 2 = code lines only, 3 = probe + store
extern "C" int request_ceiling(int mix, uint64_t code_bytes, uint64_t filt_bytes, uint32_t trips, int waves_per_cu, double* g_requests_per_s, double* ms) {
  int dev = 0, cus = 256;
  hipDeviceProp_t prop;
  if (hipGetDevice(&dev) != hipSuccess) return 10;
  if (hipGetDeviceProperties(&prop, dev) == hipSuccess && prop.multiProcessorCount > 0) cus = prop.multiProcessorCount;
  uint8_t* codes = nullptr;
  uint32_t *filt = nullptr, *d_out = nullptr;
  if (hipMalloc((void**)&codes, code_bytes + 256) != hipSuccess) return 11;
  if (hipMalloc((void**)&filt, filt_bytes) != hipSuccess) { (void)hipFree(codes); return 12; }
  if (hipMalloc((void**)&d_out, 64) != hipSuccess) { (void)hipFree(codes); (void)hipFree(filt); return 13; }
  (void)hipMemset(filt, 0, filt_bytes);
  int rc = 20;
  const uint64_t cl = code_bytes / 128u, fw = filt_bytes / 4u;
  if (mix == 0) rc = run<0, 1, 0>(codes, cl, filt, fw, trips, waves_per_cu, cus, d_out, g_requests_per_s, ms);
  else if (mix == 1) rc = run<1, 1, 2>(codes, cl, filt, fw, trips, waves_per_cu, cus, d_out, g_requests_per_s, ms);
  else if (mix == 2) rc = run<1, 0, 0>(codes, cl, filt, fw, trips, waves_per_cu, cus, d_out, g_requests_per_s, ms);
  else if (mix == 3) rc = run<0, 1, 1>(codes, cl, filt, fw, trips, waves_per_cu, cus, d_out, g_requests_per_s, ms);
  (void)hipFree(codes); (void)hipFree(filt); (void)hipFree(d_out);
  return rc;
}
