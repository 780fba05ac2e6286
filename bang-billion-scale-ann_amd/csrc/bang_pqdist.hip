// bang_pqdist.hip -- K2 ALONE in its streaming form (compute_neighborDist_par, bang_search.cu:1201-1241): the stage the BASELINE
// metric quotes an HBM figure for.  Same row fetch (CoopFetch) and the same canonical reduce (pq_row_reduce) as the search kernel
// (bang_search.hip), without the dependent chain of a search iteration around them.
//
// Reference line numbers: /root/reference/BANG_Base/bang_search.cu.

#include <hip/hip_runtime.h>
#include <type_traits>
#include <utility>
#include <stdint.h>

#include "bang_c.h"
#include "bang_internal.h"
#include "bang_device.h"

// ---------------------------------------------------------------------------------------------------------------------
// K2 alone, streaming form (compute_neighborDist_par :1201-1241): dist[q][j] for the cnt[q] <= 64 neighbours of every query
// ---------------------------------------------------------------------------------------------------------------------
// The stage the BASELINE metric quotes an HBM figure for.  One wave per query row at a time, the NEXT row's ids and code rows in
// flight while the current one is reduced (two register sets, ping-pong), pivot table in LDS, centred query through scalar loads:
// the launch is bound by how fast the memory system returns random 32-74-byte rows, not by dependent round trips.
// runs f(integral_constant<0>), f(integral_constant<1>), ... until one returns false
template <class F, int... I>
__device__ __forceinline__ bool pipe_trip(F& f, std::integer_sequence<int, I...>) {
  return (f(std::integral_constant<int, I>{}) && ...);
}

// rows of at least three 16-byte pieces are fetched cooperatively (CoopFetch, bang_device.h); two-piece rows (m = 32) measured slower that way
__host__ __device__ constexpr bool k2_coop(int ndw, bool aligned) { return (ndw + (aligned ? 0 : 1) + 3) / 4 >= 3; }
template <int PSZ, int NDW, bool ALIGNED, int NHI, int MAXT>
__global__ __launch_bounds__(MAXT) void pqdist_stream_kernel(const bang_iter_params p, uint32_t lds_piv_floats) {
  extern __shared__ __attribute__((aligned(16))) float lds[];
  float* piv_lds = lds;
  constexpr bool COOP = k2_coop(NDW, ALIGNED);
  const uint32_t stride = p.code_stride ? p.code_stride : p.m;
  // staging area of this wave's cooperative row fetch, behind the pivot table
  uint32_t* coop_buf = (uint32_t*)(lds + lds_piv_floats) + (size_t)(threadIdx.x >> 6) * CoopFetch<NDW, ALIGNED>::LDS_WORDS;
  {
    const float4* src = (const float4*)p.d_pivots_packed;
    float4* dst = (float4*)piv_lds;
    const uint32_t n4 = lds_piv_floats >> 2;
    for (uint32_t i = threadIdx.x; i < n4; i += blockDim.x) dst[i] = src[i];
    __syncthreads();
  }
  // segments of the row reduce (dependency fence in pq_row_reduce): at most ~32-48 pivot floats in flight per lane
  constexpr int SB = (NDW >= 18) ? 6 : 0;
  const int lane = lane_id();
  const uint32_t nwaves = blockDim.x >> 6;
  const uint32_t step = gridDim.x * nwaves;
  const uint32_t q = blockIdx.x * nwaves + uni(threadIdx.x >> 6);
  if (q >= p.Q) return;
  constexpr int QW = NDW * 4 * PSZ;                 // floats of a centred query (padded layout)
  constexpr int NV = (QW + 63) / 64;
  typedef QcRegs<NV> Qc;                            // the centred query in registers (v_readlane): no scalar-load waits inside the reduce
  // n_all != 0: the Q neighbour rows belong to n_all distinct queries (row q -> query q mod n_all), as the rows of successive
  // iterations of a search do; 0: one query per row
  const uint32_t qc_rows = p.n_all ? p.n_all : p.Q;
  auto load_qc = [&](Qc& dst, uint32_t qq) {
    const float* src = p.d_qc + (size_t)(qq % qc_rows) * QW;
#pragma unroll
    for (int r = 0; r < NV; ++r) {
      const uint32_t i = (uint32_t)r * 64u + (uint32_t)lane;
      dst.v[r] = src[i < (uint32_t)QW ? i : 0u];
    }
  };
  // Software pipeline per wave: while row t is reduced, the code rows of rows t+1 .. t+RD-1 are in flight (requested when their ids
  // had arrived) and so are the ids, count and centred query of row t+RD -- no step waits for a dependent round trip.  Code-row
  // buffers rotate by RD, the {ids, count, query} slots by RD + 1; the steps are generated with compile-time slot numbers
  // (RD (RD + 1) of them per trip of the loop) so that everything stays in registers.
  // (RD = 3 / 4 with exact waits: m = 32 35.7 -> 34.6 / 28.6 G rows/s, the long-row instances spill and halve -- tools/dev/run_k2rd.sh)
  constexpr int RD = 2, SD = RD + 1;
  PqRow<NDW, ALIGNED> row[COOP ? 1 : RD];
  CoopFetch<NDW, ALIGNED> raw[COOP ? RD : 1];
  Qc qc[SD];
  uint32_t ids[SD], cnt[SD], qq[SD];
  bool has[SD];
  uint32_t qnext = q;
  // Every load of the pipeline is issued unconditionally -- behind the last row of this wave the slots re-read its FIRST row, a lane
  // without a neighbour reads code row 0 -- so that the compiler's count of outstanding loads is exact and a step waits for the rows
  // requested one step earlier, not for the ones it has just requested (one load under a branch and every wait becomes vmcnt(0)).
  auto load_ids = [&](int s) {
    has[s] = qnext < p.Q;
    const uint32_t qv = has[s] ? qnext : q;
    qq[s] = qv;
    cnt[s] = p.d_cnt[qv];                             // (same address in every lane: no readfirstlane, which would wait for the load here)
    ids[s] = p.d_nbrs[(size_t)qv * BANG_NBR_STRIDE + lane];
    load_qc(qc[s], qv);
    qnext += step;
  };
  auto load_rows = [&](int s, int r) {
    const uint32_t nn = cnt[s] < 64u ? cnt[s] : 64u;
    if (COOP) raw[r].issue(p.d_codes, stride, ids[s], nn, lane);
    else pq_row_load(row[r], p.d_codes, stride, (uint32_t)lane < nn ? ids[s] : 0u);
  };
#pragma unroll
  for (int i = 0; i < RD; ++i) load_ids(i);
#pragma unroll
  for (int i = 0; i < RD - 1; ++i) load_rows(i, i);
  auto pipe_step = [&](auto U) -> bool {              // one pipeline step with compile-time slot numbers
    constexpr int u = decltype(U)::value, s = u % SD, r = u % RD;
    load_ids((s + RD) % SD);
    load_rows((s + RD - 1) % SD, (r + RD - 1) % RD);
    if (!has[s]) return false;                        // (queries are handed out in increasing order: nothing behind this one)
    // all lanes, full EXEC (v_readlane / DPP read other lanes); the s_nop covers the EXEC -> DPP hazard of a branch just taken
    asm volatile("s_nop 4");
    if (COOP) raw[r].collect(row[0], coop_buf, stride, ids[s], lane);
    const float d = pq_row_reduce<PSZ, NDW, ALIGNED, NHI, SB>(row[COOP ? 0 : r], piv_lds, qc[s]);     // (segment form: 3 % faster here than the pipelined reduce)
    if ((uint32_t)lane < (cnt[s] < 64u ? cnt[s] : 64u)) p.d_dist[(size_t)qq[s] * BANG_NBR_STRIDE + lane] = d;
    return true;
  };
  for (;;)
    if (!pipe_trip(pipe_step, std::make_integer_sequence<int, RD * SD>{})) return;
}

template <int PSZ, int NDW, bool ALIGNED, int NHI>
static int launch_pqdist_inst(const bang_iter_params& p, uint32_t piv_floats, hipStream_t st) {
  // long rows: two cooperative fetches in flight + the row being reduced need more than the 128 VGPRs of a 16-wave workgroup
  // (12 waves x 168 VGPRs: rows 128 B apart 30.5 -> 32.3 (m = 70), 30.9 -> 32.9 G rows/s (m = 74) against 8 waves x 256)
  constexpr int MAXT = (NDW >= 18 || (NDW >= 16 && PSZ == 2)) ? 768 : 1024;
  static bool attr_done[BANG_MAX_DEVICES] = {false};
  const int dev = current_device();
  if (!attr_done[dev]) {
    HIP_TRY(hipFuncSetAttribute((const void*)pqdist_stream_kernel<PSZ, NDW, ALIGNED, NHI, MAXT>, hipFuncAttributeMaxDynamicSharedMemorySize,
                                160 * 1024));
    attr_done[dev] = true;
  }
  const uint32_t cus = (uint32_t)num_cus();
  const uint32_t waves = MAXT / WAVE;
  uint32_t grid = (p.Q + waves - 1) / waves;
  if (grid > cus) grid = cus;
  const size_t lds_bytes = (size_t)piv_floats * 4 + (k2_coop(NDW, ALIGNED) ? (size_t)waves * CoopFetch<NDW, ALIGNED>::LDS_WORDS * 4 : 0);
  if (lds_bytes > 160 * 1024) { bang_set_error("K2 streaming form: pivot table + staging exceed LDS"); return BANG_ERR_UNSUPPORTED; }
  hipLaunchKernelGGL((pqdist_stream_kernel<PSZ, NDW, ALIGNED, NHI, MAXT>), dim3(grid), dim3(MAXT), lds_bytes, st, p, piv_floats);
  HIP_TRY(hipGetLastError());
  return BANG_OK;
}

template <int PSZ, int NDW>
static int launch_pqdist_al(const bang_iter_params& p, uint32_t piv_floats, hipStream_t st) {
  const bool al = ((p.code_stride ? p.code_stride : p.m) % 4u) == 0;            // rows start dword-aligned
  if (p.pq_nhi) {
    constexpr int NHI = (PSZ == 2 && NDW == 18) ? 58 : (PSZ == 2 && NDW == 19) ? 22 : 0;
    if constexpr (NHI != 0) {
      if ((int)p.pq_nhi == NHI) return al ? launch_pqdist_inst<PSZ, NDW, true, NHI>(p, piv_floats, st) : launch_pqdist_inst<PSZ, NDW, false, NHI>(p, piv_floats, st);
    }
    bang_set_error("no K2 instance for the exact-size pivot table psz=%u mp=%u nhi=%u", p.psz, p.mp, p.pq_nhi);
    return BANG_ERR_UNSUPPORTED;
  }
  return al ? launch_pqdist_inst<PSZ, NDW, true, 0>(p, piv_floats, st) : launch_pqdist_inst<PSZ, NDW, false, 0>(p, piv_floats, st);
}

extern "C" int bang_k_pqdist_stream(const bang_iter_params* p, void* stream) {
  if (!p) return BANG_ERR_ARG;
  if (p->Q == 0) return BANG_OK;
  if (p->psz == 0 || p->mp < p->m || (p->mp & 3u) || p->m == 0) { bang_set_error("K2 streaming form needs the LDS-resident pivot layout"); return BANG_ERR_UNSUPPORTED; }
  if (!p->d_nbrs || !p->d_dist || !p->d_cnt || !p->d_codes || !p->d_pivots_packed || !p->d_qc) { bang_set_error("null buffer"); return BANG_ERR_ARG; }
  if (p->pq_nhi && (p->psz != 2 || p->pq_nhi > p->mp)) { bang_set_error("bad pq_nhi"); return BANG_ERR_ARG; }
  const uint32_t pf = pivot_table_floats(p->psz, p->mp, p->pq_nhi);
  hipStream_t st = (hipStream_t)stream;
  switch (p->psz * 100u + p->mp / 4u) {
#ifdef BANG_DEV_ONLY_218
    case 218: return launch_pqdist_al<2, 18>(*p, pf, st);
    default: bang_set_error("development build: psz=2 mp=72 only"); return BANG_ERR_UNSUPPORTED;
  }
  switch (0u) {
#endif
    case 108: return launch_pqdist_al<1, 8>(*p, pf, st);
    case 116: return launch_pqdist_al<1, 16>(*p, pf, st);
    case 124: return launch_pqdist_al<1, 24>(*p, pf, st);
    case 132: return launch_pqdist_al<1, 32>(*p, pf, st);
    case 208: return launch_pqdist_al<2, 8>(*p, pf, st);
    case 216: return launch_pqdist_al<2, 16>(*p, pf, st);
    case 218: return launch_pqdist_al<2, 18>(*p, pf, st);
    case 219: return launch_pqdist_al<2, 19>(*p, pf, st);
    case 404: return launch_pqdist_al<4, 4>(*p, pf, st);
    case 408: return launch_pqdist_al<4, 8>(*p, pf, st);
    case 802: return launch_pqdist_al<8, 2>(*p, pf, st);
    case 804: return launch_pqdist_al<8, 4>(*p, pf, st);
    default: bang_set_error("no K2 instance for psz=%u mp=%u", p->psz, p->mp); return BANG_ERR_UNSUPPORTED;
  }
}

