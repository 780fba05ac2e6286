// bang_kernels.hip -- hand-written gfx950 (CDNA4, wave64) kernels of the BANG_Base search path
// and their C-ABI launchers (include/bang_c.h, section 2).
//
// Design (DESIGN.md has the long version):
//  * one WAVEFRONT per query.  R = 64 neighbours == 64 lanes, so the visited filter is a
//    __ballot + popcount compaction, the parent pick is a shuffle reduce, the 65-element sort is
//    an in-LDS rank sort and the worklist merge is a merge-path done by one wave -- no
//    __syncthreads anywhere on the per-iteration path.
//  * PQ distances are "pivot-stationary": the whole pivot table (chunk-packed, <= 152 KB) lives
//    in the CU's 160 KB LDS and is shared by the 4..16 waves (= queries) of the workgroup; an
//    LUT entry LUT[c][code] = sum_j (P[j][code] - qc[j])^2 is recomputed from it with exactly the
//    float operations populate_pqDist_par uses, so the 0.3-0.7 GB per-batch LUT of the reference
//    is never materialised or re-read.  HBM traffic of the distance stage = the code rows.
//  * compiled with -ffp-contract=off; every fused multiply-add is an explicit fmaf so the float
//    results equal the CPU oracle's bit for bit.
//
// Reference line numbers: /root/reference/BANG_Base/bang_search.cu.

#include <hip/hip_runtime.h>
#include <stdint.h>
#include <stdlib.h>
#include <algorithm>

#include "bang_c.h"
#include "bang_internal.h"
#include "bang_device.h"

// LUT path (PSZ == 0): LUT[m][256] gathered from global memory as the reference does (:1236); any m.
__device__ __forceinline__ float pq_distance_lut(const uint8_t* __restrict__ codes, uint32_t m, uint32_t stride, uint32_t id,
                                                 const float* __restrict__ lut) {
  const uint8_t* row = codes + (uint64_t)id * stride;
  float s[8];
#pragma unroll
  for (int l = 0; l < 8; ++l) s[l] = 0.0f;
  uint32_t c = 0;
  for (; c + 8 <= m; c += 8) {
#pragma unroll
    for (int l = 0; l < 8; ++l) s[l] = s[l] + lut[(size_t)(c + l) * 256 + row[c + l]];
  }
#pragma unroll
  for (int l = 0; l < 8; ++l)
    if (c + l < m) s[l] = s[l] + lut[(size_t)(c + l) * 256 + row[c + l]];
  const float x = (s[0] + s[1]) + (s[2] + s[3]);
  const float y = (s[4] + s[5]) + (s[6] + s[7]);
  return x + y;
}

// ------------------------------------------------------------------------------------------
// back: K3a stable sort of the survivors + K3b merge into the worklist, one wave per query
// ------------------------------------------------------------------------------------------
#define BACK_WAVES 4
#define BACK_WL_REGS (BANG_MAX_L / WAVE)   // worklist entries a lane owns: k = lane + 64 j
#define BACK_WI_REGS 4                     // ... whose ids are kept in registers across the sort (the rest is parked in LDS)
struct __attribute__((aligned(16))) BackLds {
  float sd[BANG_NBR_STRIDE];      // unsorted distances (read 16 bytes at a time by the rank sort)
  uint32_t si[BANG_NBR_STRIDE];
  float td[BANG_NBR_STRIDE];      // sorted
  uint32_t ti[BANG_NBR_STRIDE];
  float wd[BANG_MAX_L];           // worklist distances (binary-searched); its flags and the first 256 ids stay in registers
  uint32_t wi_hi[BANG_MAX_L - BACK_WI_REGS * WAVE];   // ids of entries 256.. (L > 256 only)
};
// per-wave LDS view used by back_one_query (the static BackLds of back_kernel)
struct BackView {
  float* sd; uint32_t* si; float* td; uint32_t* ti; float* wd; uint32_t* wi_hi;
};

// Everything back_one_query reads from global memory: depends on the query only, so a wave can have the NEXT query's loads
// in flight while it sorts and merges the current one.
template <int WLR>                   // WLR = worklist entries a lane may own = ceil(L / 64), rounded up to a compiled size
struct BackIn {
  uint32_t n_raw, w_raw, mark, i_lo, i_hi;
  float d_lo, d_hi;
  float wd_r[WLR];                   // short lived: parked in LDS before the sort
  uint32_t wi_r[WLR];                // ids of the worklist entries this lane owns (k = lane + 64 j); j >= BACK_WI_REGS parked in LDS
  uint32_t wv_bits;                  // their visited flags, bit j
};
#define BACK_HI_AT(lane) (WAVE + ((uint32_t)(lane) & (BANG_NBR_STRIDE - WAVE - 1)))   // entries 64..71 (a row holds at most R + 1 = 65)

// ONE memory round trip: every load depends on q only and is issued before the first result is used (the survivor row always
// holds BANG_NBR_STRIDE words and the worklist arrays L words, whatever the counters say)
template <int WLR>
__device__ __forceinline__ void back_load(const bang_iter_params& p, uint32_t q, int lane, BackIn<WLR>& in) {
  const uint32_t L = p.L;
  const uint32_t* nbrs = p.d_nbrs + (size_t)q * BANG_NBR_STRIDE;
  const float* dist = p.d_dist + (size_t)q * BANG_NBR_STRIDE;
  const uint32_t* wl_ids = p.d_wl_ids + (size_t)q * L;
  const float* wl_dist = p.d_wl_dist + (size_t)q * L;
  const uint8_t* wl_vis = p.d_wl_vis + (size_t)q * L;
  in.n_raw = p.d_cnt[q];
  in.w_raw = p.d_wl_cnt[q];
  in.mark = p.d_mark[q];
  in.d_lo = dist[lane];
  in.i_lo = nbrs[lane];
  in.d_hi = dist[BACK_HI_AT(lane)];
  in.i_hi = nbrs[BACK_HI_AT(lane)];
  uint32_t vis[WLR];
#pragma unroll
  for (int j = 0; j < WLR; ++j) {
    in.wd_r[j] = 0.0f; vis[j] = 0u;
    in.wi_r[j] = 0u;
    if ((uint32_t)j * WAVE < L) {                       // uniform: only the ceil(L / 64) entries a lane can own are loaded
      const uint32_t i = (uint32_t)lane + (uint32_t)j * WAVE;
      const uint32_t ic = i < L ? i : 0u;               // lanes past the end re-read entry 0 (never used)
      in.wd_r[j] = wl_dist[ic]; vis[j] = wl_vis[ic];
      in.wi_r[j] = wl_ids[ic];
    }
  }
  uint32_t bits = 0;
#pragma unroll
  for (int j = 0; j < WLR; ++j) bits |= (vis[j] ? 1u : 0u) << j;
  in.wv_bits = bits;
}

// sort + merge of one query by one wave (compute_BestLSets_par_sort_msort :1533-1585, compute_BestLSets_par_merge :1605-1715)
template <int WLR>
__device__ __forceinline__ void back_one_query(const bang_iter_params& p, uint32_t q, uint32_t iter, const BackView& s, int lane,
                                               const BackIn<WLR>& in) {
  const uint32_t L = p.L;
  uint32_t* wl_ids = p.d_wl_ids + (size_t)q * L;
  float* wl_dist = p.d_wl_dist + (size_t)q * L;
  uint8_t* wl_vis = p.d_wl_vis + (size_t)q * L;
  const uint32_t n_raw = in.n_raw, w_raw = in.w_raw, mark = in.mark;
  const float d_lo = in.d_lo, d_hi = in.d_hi;
  const uint32_t i_lo = in.i_lo, i_hi = in.i_hi;
  const uint32_t hi_at = BACK_HI_AT(lane);
  const float* wd_r = in.wd_r;
  const uint32_t n = uni(n_raw);
  if (n == 0) { wave_sync(); return; }     // :1547 / :1636 -- nothing to sort or merge (mark step is a no-op then)
  s.sd[lane] = d_lo; s.si[lane] = i_lo;
  if ((uint32_t)lane < BANG_NBR_STRIDE - WAVE) { s.sd[hi_at] = d_hi; s.si[hi_at] = i_hi; }
#pragma unroll
  for (int j = 0; j < WLR; ++j) {
    const uint32_t i = (uint32_t)lane + (uint32_t)j * WAVE;
    if (i < L) {
      s.wd[i] = wd_r[j];
      if (WLR > BACK_WI_REGS && j >= BACK_WI_REGS) s.wi_hi[i - BACK_WI_REGS * WAVE] = in.wi_r[j];
    }
  }
  wave_sync();
  // K3a: stable rank sort == the reference's stable merge sort (:1553-1584).  The distances are read eight at a time (two 16-byte
  // LDS reads in flight instead of a chain of n dependent 4-byte ones: that chain was about half of a query's merge time); the
  // tail of the last group is padded with +inf, which ranks behind every real distance and ties with none.
  const uint32_t n8 = (n + 7u) & ~7u;                    // <= 72 = BANG_NBR_STRIDE
  if ((uint32_t)lane < n8 - n) s.sd[n + lane] = __builtin_inff();
  wave_sync();
  for (uint32_t i = lane; i < n; i += WAVE) {
    const float d = s.sd[i];
    uint32_t r = 0;
    for (uint32_t j = 0; j < n8; j += 8) {
      const float4 o0 = *(const float4*)(s.sd + j);
      const float4 o1 = *(const float4*)(s.sd + j + 4);
      r += (o0.x < d || (o0.x == d && j + 0 < i)) ? 1u : 0u;
      r += (o0.y < d || (o0.y == d && j + 1 < i)) ? 1u : 0u;
      r += (o0.z < d || (o0.z == d && j + 2 < i)) ? 1u : 0u;
      r += (o0.w < d || (o0.w == d && j + 3 < i)) ? 1u : 0u;
      r += (o1.x < d || (o1.x == d && j + 4 < i)) ? 1u : 0u;
      r += (o1.y < d || (o1.y == d && j + 5 < i)) ? 1u : 0u;
      r += (o1.z < d || (o1.z == d && j + 6 < i)) ? 1u : 0u;
      r += (o1.w < d || (o1.w == d && j + 7 < i)) ? 1u : 0u;
    }
    s.td[r] = d;
    s.ti[r] = s.si[i];
  }
  wave_sync();

  uint32_t new_n;
  if (iter == 1) {                                       // :1638-1649
    new_n = n < L ? n : L;
    for (uint32_t i = lane; i < new_n; i += WAVE) {
      const uint32_t id = s.ti[i];
      wl_ids[i] = id;
      wl_dist[i] = s.td[i];
      wl_vis[i] = (id == p.medoid || id == mark) ? 1 : 0;  // + mark step :1711-1714
    }
  } else {                                               // :1650-1708
    const uint32_t w_n = uni(w_raw);
    const float worst = s.wd[w_n - 1];
    const uint32_t lim = L < n ? L : n;
    // nb = number of leading new entries with dist < worst (stop at the first >=) :1653-1657
    uint32_t nb = lim;
    for (uint32_t base = 0; base < lim; base += WAVE) {
      const uint32_t i = base + lane;
      const bool ge = (i < lim) && (s.td[i] >= worst);
      const uint64_t mk = __ballot(ge);
      if (mk) { nb = base + (uint32_t)__builtin_ctzll(mk); break; }
    }
    const uint32_t room = L - w_n;
    const uint32_t fill = room < n ? room : n;
    if (fill > nb) nb = fill;                            // :1660
    new_n = (w_n + nb) < L ? (w_n + nb) : L;             // :1662
    for (uint32_t i = lane; i < nb; i += WAVE) {         // new entries: lower_bound + i :1675-1677
      const float d = s.td[i];
      const uint32_t pos = lower_bound_lds(s.wd, w_n, d) + i;
      if (pos < new_n) {
        const uint32_t id = s.ti[i];
        wl_ids[pos] = id; wl_dist[pos] = d; wl_vis[pos] = (id == mark) ? 1 : 0;
      }
    }
#pragma unroll
    for (int j = 0; j < WLR; ++j) {             // old entries: upper_bound + k :1678-1680
      const uint32_t k = (uint32_t)lane + (uint32_t)j * WAVE;
      if (k < w_n) {
        const float d = s.wd[k];
        const uint32_t pos = upper_bound_lds(s.td, nb, d) + k;
        if (pos < new_n) {
          const uint32_t id = (WLR <= BACK_WI_REGS || j < BACK_WI_REGS) ? in.wi_r[j] : s.wi_hi[k - BACK_WI_REGS * WAVE];
          wl_ids[pos] = id; wl_dist[pos] = d; wl_vis[pos] = (((in.wv_bits >> j) & 1u) || id == mark) ? 1 : 0;
        }
      }
    }
  }
  if (lane == 0) p.d_wl_cnt[q] = new_n;
  wave_sync();
}

__global__ __launch_bounds__(BACK_WAVES* WAVE) void back_kernel(const bang_iter_params p) {
  __shared__ BackLds lds_all[BACK_WAVES];
  const int lane = lane_id();
  const uint32_t wave = uni(threadIdx.x >> 6);
  BackLds& b = lds_all[wave];
  BackView s;
  s.sd = b.sd; s.si = b.si; s.td = b.td; s.ti = b.ti; s.wd = b.wd; s.wi_hi = b.wi_hi;
  for (uint32_t slot = blockIdx.x * BACK_WAVES + wave; slot < p.Q; slot += gridDim.x * BACK_WAVES) {
    const uint32_t q = p.d_qmap ? uni(p.d_qmap[slot]) : slot;
    BackIn<BACK_WL_REGS> in;
    back_load<BACK_WL_REGS>(p, q, lane, in);
    back_one_query<BACK_WL_REGS>(p, q, p.iter, s, lane, in);
  }
}

// ------------------------------------------------------------------------------------------
// front kernel: K5 filter -> K2 distances -> K4 parent, one wave per query
// ------------------------------------------------------------------------------------------
struct FrontArgs {
  bang_iter_params p;
  uint32_t stages;      // bit0 filter, bit1 distance, bit2 parent
  uint32_t debug;       // timing-only ablations (BANG_FRONT_DEBUG): 1 no filter updates, 2 no distance math, 4 no pivot staging
  uint32_t lds_piv_floats;
};

// per-wave LDS scratch (uint32 words): compacted ids [0..71]
#define FRONT_SCRATCH_WORDS 72
// ALL = true: the production instantiation (filter + distance + parent, no stage branches).  ALL = false:
// stage mask taken from a.stages (kernel-level parity tests).
// NQW = queries a wave works on AT THE SAME TIME.  The kernel is bound by dependent memory latency (row ->
// filter words -> code rows, ~2 us each) with at most 16 waves per CU (the pivot table owns the LDS), so every
// phase is executed for NQW independent queries back to back: their loads are in flight together and the wave
// pays each round trip once per NQW queries.
template <int PSZ, int NDW, bool ALIGNED, bool ALL, int NQW, int MAXT, int NHI>
__global__ __launch_bounds__(MAXT) void front_kernel(const FrontArgs a) {
  extern __shared__ __attribute__((aligned(16))) float lds[];
  const bang_iter_params& p = a.p;
  float* piv_lds = lds;
  uint32_t* scratch_all = (uint32_t*)(lds + a.lds_piv_floats);
  if (p.d_ktime && threadIdx.x == 0) p.d_ktime[2 * blockIdx.x] = __builtin_amdgcn_s_memrealtime();
  const bool do_filter = ALL || (a.stages & 1u);
  const bool do_dist = ALL || (a.stages & 2u);
  const bool do_parent = ALL || (a.stages & 4u);

  if (PSZ > 0 && do_dist && !(a.debug & 4u)) {
    // stage the chunk-packed pivot table (query independent) once per workgroup: 16 B per lane,
    // 8 unconditional loads in flight per lane before the first LDS write
    const float4* src = (const float4*)p.d_pivots_packed;
    float4* dst = (float4*)piv_lds;
    const uint32_t n4 = a.lds_piv_floats >> 2;
    for (uint32_t i0 = threadIdx.x; i0 < n4; i0 += blockDim.x * 8) {
      float4 v[8];
#pragma unroll
      for (int j = 0; j < 8; ++j) {
        const uint32_t i = i0 + (uint32_t)j * blockDim.x;
        v[j] = src[i < n4 ? i : n4 - 1];
      }
#pragma unroll
      for (int j = 0; j < 8; ++j) {
        const uint32_t i = i0 + (uint32_t)j * blockDim.x;
        if (i < n4) dst[i] = v[j];
      }
    }
    __syncthreads();
  }

  const int lane = lane_id();
  const uint32_t wave = uni(threadIdx.x >> 6);
  const uint32_t nwaves = blockDim.x >> 6;
  uint32_t* scratch = scratch_all + wave * (FRONT_SCRATCH_WORDS * NQW);
  const uint32_t medoid = p.medoid;
  const uint32_t L = p.L;
  const uint32_t lane_l = (uint32_t)lane < L ? (uint32_t)lane : L - 1;
  const uint32_t total_waves = gridDim.x * nwaves;
  const uint32_t gw = blockIdx.x * nwaves + wave;
  // slots of this wave: gw, gw + total_waves, ... < Q
  const uint32_t q_end = p.Q, q_begin = gw, q_step = total_waves;
  const uint32_t cand_stride = L + BANG_EXTRA_ITERS;
  const uint32_t cur_iter = p.iter;
  const uint32_t first = p.first ? 1u : 0u;
  uint32_t n_active = 0;

  for (uint32_t g0 = q_begin; g0 < q_end; g0 += q_step * NQW) {
    uint32_t q[NQW];
    bool valid[NQW];
#pragma unroll
    for (int u = 0; u < NQW; ++u) {
      const uint32_t slot = g0 + (uint32_t)u * q_step;
      valid[u] = slot < q_end;
      const uint32_t s_ok = valid[u] ? slot : g0;   // invalid slots shadow slot 0: loads stay legal, every store is guarded
      q[u] = p.d_qmap ? uni(p.d_qmap[s_ok]) : s_ok;  // straggler compaction: slot -> query
    }

    // ---- round trip A: every load that does not depend on another load of the query, issued
    // unconditionally (the launcher guarantees all pointers are valid).  The worklist head needed by K4 is
    // prefetched speculatively: the arrays always hold L valid words.
    uint32_t cnt_in[NQW], x0[NQW], x1[NQW], cc[NQW], w_n[NQW], pw_vis[NQW], pw_id[NQW];
    float pw_dist[NQW];
    uint2 qs[NQW];
    bool have_row[NQW];
    const uint32_t* row[NQW];     // {count, id x 64 (, 65th id for the seed list)}
    if (p.d_graph != nullptr && !first) {  // graph resident in HBM: the parent's adjacency is read in place
      uint32_t par[NQW];
#pragma unroll
      for (int u = 0; u < NQW; ++u) par[u] = p.d_parents[q[u]];
#pragma unroll
      for (int u = 0; u < NQW; ++u) {
        const uint32_t pr = uni(par[u]);
        have_row[u] = pr < BANG_IDLE_PARENT;
        row[u] = (const uint32_t*)(p.d_graph + (uint64_t)(have_row[u] ? pr : 0u) * p.entry_len + p.vec_bytes);
      }
    } else {
      // seed list [MEDOID, adj(MEDOID)...] (bang_init :467-489) or the row staged by the host walker (:827-833)
#pragma unroll
      for (int u = 0; u < NQW; ++u) {
        have_row[u] = true;
        row[u] = first ? p.d_seed : p.d_stage + (size_t)q[u] * BANG_STAGE_STRIDE;
      }
    }
#pragma unroll
    for (int u = 0; u < NQW; ++u) {
      cnt_in[u] = row[u][0];
      x0[u] = row[u][1 + lane];                          // in bounds for every row kind (R <= 64)
      x1[u] = row[u][65u * first];                       // 65th id exists only in the seed list
      cc[u] = p.d_cand_cnt[q[u]];
      w_n[u] = p.d_wl_cnt[q[u]];
      pw_vis[u] = p.d_wl_vis[(size_t)q[u] * L + lane_l];
      pw_dist[u] = p.d_wl_dist[(size_t)q[u] * L + lane_l];
      pw_id[u] = p.d_wl_ids[(size_t)q[u] * L + lane_l];
      qs[u] = make_uint2(0u, 0u);
      if (p.d_qstats && lane == 0) qs[u] = *(const uint2*)(p.d_qstats + (size_t)q[u] * 2);
    }

    uint32_t n[NQW], sid0[NQW], sid1[NQW];   // survivors; lane's survivor id; survivor 64 (lane 0 only)
    uint32_t h0a[NQW], h0b[NQW], h1a[NQW], h1b[NQW];   // filter slots of the lane's id / of element 64
    bool set0[NQW], set1[NQW];                        // slots to set once the query's loads have been issued
#pragma unroll
    for (int u = 0; u < NQW; ++u) { n[u] = 0; sid0[u] = 0; sid1[u] = 0; set0[u] = false; set1[u] = false; h0a[u] = h0b[u] = h1a[u] = h1b[u] = 0; }

    // ---------------- K5: filter (neighbor_filtering_new :1140-1165) ----------------
    if (do_filter) {
      uint32_t w0a[NQW], w0b[NQW], w1a[NQW], w1b[NQW];
      // ---- round trips B, B': visited-filter words (a hash is always a valid index).
      // CANON: every id is tested against the filter state at entry (all loads before any set)
#pragma unroll
      for (int u = 0; u < NQW; ++u) {
        // (the count arrives with the row: lanes past the list's end -- short adjacency lists, idle queries -- probe nothing)
        uint32_t ci = uni(cnt_in[u]);
        if (!have_row[u] || !valid[u]) ci = 0;
        const uint32_t cap = p.R + first;
        if (ci > cap) ci = cap;
        cnt_in[u] = ci;
      }
#pragma unroll
      for (int u = 0; u < NQW; ++u) {
        const uint32_t* bloom = p.d_bloom + (size_t)q[u] * BANG_BF_WORDS;
        const bool v0 = (uint32_t)lane < cnt_in[u];
        const bool v1 = cnt_in[u] > 64;
        h0a[u] = hash1(x0[u]); h0b[u] = hash2(x0[u]);
        if (v1) { h1a[u] = hash1(x1[u]); h1b[u] = hash2(x1[u]); }   // the 65th id exists in the seed list only (uniform branch):
                                                                 // two 64-bit multiply-and-modulo hashes less per lane otherwise
        if (a.debug & 8u) { w0a[u] = w0b[u] = w1a[u] = w1b[u] = 0; }
        else if (a.debug & 16u) {                     // both slots probed at once (one round trip, more sectors)
          w0a[u] = bloom[h0a[u] >> 5]; w0b[u] = bloom[h0b[u] >> 5];
          w1a[u] = bloom[h1a[u] >> 5]; w1b[u] = bloom[h1b[u] >> 5];
        } else {
          // the second slot is only probed where the first one is set: a never-seen id usually stops after one
          // probe (-25..-45 % filter sectors for one more dependent round trip; measured -4..-7 % kernel time)
          w0a[u] = 0; w1a[u] = 0; w0b[u] = 0; w1b[u] = 0;
          if (v0) w0a[u] = bloom[h0a[u] >> 5];
          if (v1) w1a[u] = bloom[h1a[u] >> 5];
          if ((w0a[u] >> (h0a[u] & 31)) & 1u) w0b[u] = bloom[h0b[u] >> 5];
          if ((w1a[u] >> (h1a[u] & 31)) & 1u) w1b[u] = bloom[h1b[u] >> 5];
        }
      }
#pragma unroll
      for (int u = 0; u < NQW; ++u) {
        uint32_t* sc = scratch + u * FRONT_SCRATCH_WORDS;
        const uint32_t ci = cnt_in[u];
        // round 0: lanes 0..63 ; round 1: element 64 (only the seed list has 65 entries)
        const bool v0 = (uint32_t)lane < ci;
        const bool v1 = (lane == 0) && (ci > 64);
        const bool pass0 = v0 && !(((w0a[u] >> (h0a[u] & 31)) & 1u) && ((w0b[u] >> (h0b[u] & 31)) & 1u));
        const bool pass1 = v1 && !(((w1a[u] >> (h1a[u] & 31)) & 1u) && ((w1b[u] >> (h1b[u] & 31)) & 1u));
        const uint64_t m0 = __ballot(pass0);
        const uint64_t m1 = __ballot(pass1);
        const uint32_t n0 = (uint32_t)__popcll(m0);
        n[u] = n0 + (uint32_t)__popcll(m1);
        // ordered compaction through LDS: survivors keep input order (CANON; reference uses atomicAdd
        // order :1161).  One wave executes in order, so no fence is needed between the write and the read.
        if (pass0) sc[lanes_below(m0)] = x0[u];
        if (pass1) sc[n0] = x1[u];
        wave_sync();
        if ((uint32_t)lane < n[u]) sid0[u] = sc[lane];
        if (lane == 0 && n[u] > 64) sid1[u] = sc[64];
        set0[u] = pass0 && !(a.debug & 1u);
        set1[u] = pass1 && !(a.debug & 1u);
      }
#pragma unroll
      for (int u = 0; u < NQW; ++u) {
        if (valid[u]) {
          uint32_t* nbrs = p.d_nbrs + (size_t)q[u] * BANG_NBR_STRIDE;
          if ((uint32_t)lane < n[u]) nbrs[lane] = sid0[u];
          if (lane == 0) {
            if (n[u] > 64) nbrs[64] = sid1[u];
            p.d_cnt[q[u]] = n[u];
            if (p.d_qstats) *(uint2*)(p.d_qstats + (size_t)q[u] * 2) = make_uint2(qs[u].x + n[u], qs[u].y + cnt_in[u]);
          }
        }
      }
    } else {
#pragma unroll
      for (int u = 0; u < NQW; ++u) {
        const uint32_t* nbrs = p.d_nbrs + (size_t)q[u] * BANG_NBR_STRIDE;
        n[u] = valid[u] ? uni(p.d_cnt[q[u]]) : 0u;
        if ((uint32_t)lane < n[u]) sid0[u] = nbrs[lane];
        if (lane == 0 && n[u] > 64) sid1[u] = nbrs[64];
      }
    }

    // ---------------- K2: PQ distances (compute_neighborDist_par :1201-1241) ----------------
    float d0[NQW], d1[NQW];
#pragma unroll
    for (int u = 0; u < NQW; ++u) { d0[u] = BIG_DIST; d1[u] = BIG_DIST; }
    if (do_dist) {
      const uint32_t cstride = p.code_stride ? p.code_stride : p.m;      // bytes between code rows
      if constexpr (PSZ > 0) {
        // ---- round trip C: the code rows, software pipelined over the NQW queries
        PqRow<NDW, ALIGNED> rows[2];
        if ((uint32_t)lane < n[0]) pq_row_load(rows[0], p.d_codes, cstride, sid0[0]);
#pragma unroll
        for (int u = 0; u < NQW; ++u) {
          if (u + 1 < NQW) {
            if ((uint32_t)lane < n[u + 1]) pq_row_load(rows[(u + 1) & 1], p.d_codes, cstride, sid0[u + 1]);
          }
          cfloat_p qc = (cfloat_p)(uintptr_t)(p.d_qc + (size_t)q[u] * (NDW * 4 * PSZ));
          if ((uint32_t)lane < n[u]) {
            float d = pq_row_reduce<PSZ, NDW, ALIGNED, NHI>(rows[u & 1], piv_lds, qc);
            if (a.debug & 2u) d = (float)sid0[u];
            d0[u] = d;
          }
          if (n[u] > 64) {                                  // survivor 64 (seed list only), lane 0
            if (lane == 0) {
              PqRow<NDW, ALIGNED> r1;
              pq_row_load(r1, p.d_codes, cstride, sid1[u]);
              d1[u] = pq_row_reduce<PSZ, NDW, ALIGNED, NHI>(r1, piv_lds, qc);
            }
          }
        }
      } else {
#pragma unroll
        for (int u = 0; u < NQW; ++u) {
          const float* lut = p.d_lut + (size_t)q[u] * p.m * 256;
          if ((uint32_t)lane < n[u]) d0[u] = pq_distance_lut(p.d_codes, p.m, cstride, sid0[u], lut);
          if (n[u] > 64 && lane == 0) d1[u] = pq_distance_lut(p.d_codes, p.m, cstride, sid1[u], lut);
        }
      }
#pragma unroll
      for (int u = 0; u < NQW; ++u) {
        if (valid[u]) {
          float* dist = p.d_dist + (size_t)q[u] * BANG_NBR_STRIDE;
          if ((uint32_t)lane < n[u]) dist[lane] = d0[u];
          if (lane == 0 && n[u] > 64) dist[64] = d1[u];
        }
      }
    } else if (do_parent) {
#pragma unroll
      for (int u = 0; u < NQW; ++u) {
        const float* dist = p.d_dist + (size_t)q[u] * BANG_NBR_STRIDE;
        if ((uint32_t)lane < n[u]) d0[u] = dist[lane];
        if (lane == 0 && n[u] > 64) d1[u] = dist[64];
      }
    }

    // ---------------- K4: parent (compute_parent1 :1464-1521 / compute_parent2 :1384-1459) ------
    if (do_parent) {
#pragma unroll
      for (int u = 0; u < NQW; ++u) {
        if (!valid[u]) continue;
        const uint32_t nn = n[u];
        // closest new neighbour: strict '<', first minimum wins, MEDOID skipped (:1413-1418)
        const bool elig = (uint32_t)lane < nn && sid0[u] != medoid && d0[u] < BIG_DIST;
        float bd = elig ? d0[u] : BIG_DIST;
        uint32_t bi = elig ? (uint32_t)lane : 0xFFFFu;
        uint32_t bid = sid0[u];
#pragma unroll
        for (int off = 1; off < WAVE; off <<= 1) {
          const float od = __shfl_xor(bd, off);
          const uint32_t oi = (uint32_t)__shfl_xor((int)bi, off);
          const uint32_t oid = (uint32_t)__shfl_xor((int)bid, off);
          const bool take = (oi != 0xFFFFu) && (bi == 0xFFFFu || od < bd || (od == bd && oi < bi));
          if (take) { bd = od; bi = oi; bid = oid; }
        }
        // element 64 (lane 0 of the second round) can only win with a strictly smaller distance
        if (nn > 64) {
          const float e_d = __shfl(d1[u], 0);
          const uint32_t e_id = (uint32_t)__shfl((int)sid1[u], 0);
          if (e_id != medoid && e_d < BIG_DIST && (bi == 0xFFFFu || e_d < bd)) { bd = e_d; bi = 64; bid = e_id; }
        }
        const bool have_best = (bi != 0xFFFFu);
        if (!have_best) bd = BIG_DIST;

        bool found = false, from_best = false;
        uint32_t parent = 0, w_hit = 0;
        if (first) {
          if (have_best) { found = true; parent = bid; from_best = true; }
        } else {
          const uint32_t wn = uni(w_n[u]);
          const uint8_t* wl_vis = p.d_wl_vis + (size_t)q[u] * L;
          const float* wl_dist = p.d_wl_dist + (size_t)q[u] * L;
          const uint32_t* wl_ids = p.d_wl_ids + (size_t)q[u] * L;
          float wdist = 0.0f;
          uint32_t wid = 0;
          {                                                 // first unvisited entry :1425-1439, head from registers
            const uint64_t mk = __ballot((uint32_t)lane < wn && (uint32_t)lane < L && pw_vis[u] == 0);
            if (mk) {
              w_hit = (uint32_t)__builtin_ctzll(mk);
              found = true;
              wdist = __shfl(pw_dist[u], (int)w_hit);
              wid = (uint32_t)__shfl((int)pw_id[u], (int)w_hit);
            }
          }
          for (uint32_t base = WAVE; !found && base < wn; base += WAVE) {   // rare: L > 64 and the head is all visited
            const uint32_t i = base + lane;
            const uint64_t mk = __ballot((i < wn) && (wl_vis[i < wn ? i : 0] == 0));
            if (mk) {
              w_hit = base + (uint32_t)__builtin_ctzll(mk);
              found = true;
              wdist = wl_dist[w_hit];
              wid = wl_ids[w_hit];
            }
          }
          if (found) {
            if (bd < wdist) { parent = bid; from_best = true; }
            else parent = wid;
          } else if (wn > 0) {                              // corner case :1442-1446
            const float worst = (wn <= WAVE) ? __shfl(pw_dist[u], (int)(wn - 1)) : wl_dist[wn - 1];
            if (bd < worst) { found = true; parent = bid; from_best = true; }
          }
        }
        if (lane == 0) {
          const uint32_t qq = q[u];
          if (found) {
            if (from_best) p.d_mark[qq] = parent;
            else p.d_wl_vis[(size_t)qq * L + w_hit] = 1;
            p.d_cand_ids[(size_t)qq * cand_stride + cc[u]] = parent;
            if (p.d_cand_row) p.d_cand_row[(size_t)qq * cand_stride + cc[u]] = cur_iter;
            p.d_cand_cnt[qq] = cc[u] + 1;
            __hip_atomic_store(&p.d_parents[qq], parent, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);   // write-through
          } else {
            __hip_atomic_store(&p.d_parents[qq], (nn > 0) ? BANG_IDLE_PARENT : BANG_NO_PARENT, __ATOMIC_RELAXED,
                               __HIP_MEMORY_SCOPE_AGENT);
          }
        }
        if (found || nn > 0) ++n_active;
      }
    }

    // ---------------- K5, second half: set the slots of the survivors (:1159-1160) ----------------
    // Issued LAST: vmcnt retires in order, so anything issued after an atomic waits for it (2-3 us under load);
    // here the ORs are fire-and-forget and overlap the next queries' first round trip.  A query's filter is touched by
    // exactly one wave per launch and kernel boundaries publish it, so workgroup scope is enough.
    if (do_filter) {
#pragma unroll
      for (int u = 0; u < NQW; ++u) {
        uint32_t* bloom = p.d_bloom + (size_t)q[u] * BANG_BF_WORDS;
        if (set0[u]) { bloom_set(&bloom[h0a[u] >> 5], 1u << (h0a[u] & 31)); bloom_set(&bloom[h0b[u] >> 5], 1u << (h0b[u] & 31)); }
        if (set1[u]) { bloom_set(&bloom[h1a[u] >> 5], 1u << (h1a[u] & 31)); bloom_set(&bloom[h1b[u] >> 5], 1u << (h1b[u] & 31)); }
      }
    }
  }

  // same-address atomics from thousands of waves serialise (~90 per microsecond): a plain flag store instead
  if (lane == 0 && p.d_active && n_active) *p.d_active = 1u;

  if (p.d_ktime) {
    __syncthreads();
    if (threadIdx.x == 0) p.d_ktime[2 * blockIdx.x + 1] = __builtin_amdgcn_s_memrealtime();
  }

  // Completion signal for the host walker, without any L2-wide fence (a release fence writes back every dirty
  // line of the XCD's L2: measured +85 us per launch when every wave issued one).  Parents are stored
  // write-through (agent-scope atomic stores = `sc1`), every wave drains its stores, the workgroup arrives on a
  // device counter; the LAST workgroup re-reads the parents with `sc1` loads, copies them to mapped pinned host
  // memory with system-scope stores (coalesced: one 4-byte PCIe write per query from every wave was measured at
  // +75 us per launch), drains, and publishes the iteration number.  The walker thread spins on that word.
  if (p.h_done_flag) {
    volatile uint32_t* s_last = scratch_all;        // dynamic LDS: no static allocation next to the 160 KB request
    uint32_t* counter = p.d_done_count;
    asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
    __syncthreads();
    if (threadIdx.x == 0) {
      const uint32_t old = atomicAdd(counter, 1u);
      const uint32_t last = (old + 1 == gridDim.x) ? 1u : 0u;
      *s_last = last;
      if (last) atomicExch(counter, 0u);   // every other workgroup has already arrived
    }
    __syncthreads();
    if (*s_last) {
      const uint32_t n_par = p.n_all ? p.n_all : p.Q;
      for (uint32_t i = threadIdx.x; i < n_par; i += blockDim.x) {
        const uint32_t v = __hip_atomic_load(&p.d_parents[i], __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
        __hip_atomic_store(&p.h_parents[i], v, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_SYSTEM);
      }
      asm volatile("s_waitcnt vmcnt(0)" ::: "memory");   // the flag must not overtake the stores (MI355X guide)
      __syncthreads();
      if (threadIdx.x == 0)
        __hip_atomic_store(p.h_done_flag, p.done_value, __ATOMIC_RELEASE, __HIP_MEMORY_SCOPE_SYSTEM);
    }
    __syncthreads();                                 // s_last lives in wave 0's scratch: nobody may reuse it before all have read it
  }

}

// ------------------------------------------------------------------------------------------
// K1: populate_pqDist_par (:1083-1130) -- LUT path only
// ------------------------------------------------------------------------------------------
template <typename T>
__global__ __launch_bounds__(256) void lut_build_kernel(const float* __restrict__ pivots_T, const T* __restrict__ queries,
                                                        const float* __restrict__ centroid,
                                                        const uint32_t* __restrict__ chunk_off, float* __restrict__ lut,
                                                        uint32_t D, uint32_t m, uint32_t dim_adjust) {
  extern __shared__ __attribute__((aligned(16))) float qc[];   // float(q[j]) - centroid[j]
  const uint32_t q = blockIdx.x;
  const uint32_t qdim = D - dim_adjust;
  for (uint32_t j = threadIdx.x; j < D; j += blockDim.x) {
    const float qv = (j < qdim) ? (float)queries[(size_t)q * qdim + j] : 0.0f;   // :1102-1113
    qc[j] = qv - centroid[j];
  }
  __syncthreads();
  const uint32_t k = threadIdx.x;   // one pivot column per thread: coalesced reads of pivots_T[j][0..255]
  for (uint32_t c = 0; c < m; ++c) {
    float acc = 0.0f;               // registers instead of the reference's global RMW :1126
    for (uint32_t j = chunk_off[c]; j < chunk_off[c + 1]; ++j) {
      const float diff = pivots_T[(size_t)j * 256 + k] - qc[j];
      acc = __builtin_fmaf(diff, diff, acc);
    }
    lut[((size_t)q * m + c) * 256 + k] = acc;
  }
}

// centred queries in chunk-padded layout (first half of K1, :1099-1113,1124)
template <typename T>
__global__ void center_queries_kernel(const T* __restrict__ queries, const float* __restrict__ centroid,
                                      const uint32_t* __restrict__ chunk_off, float* __restrict__ qc, uint32_t Q,
                                      uint32_t D, uint32_t m, uint32_t mp, uint32_t psz, uint32_t dim_adjust) {
  const uint32_t per_q = mp * psz;
  const uint32_t qdim = D - dim_adjust;
  for (size_t idx = (size_t)blockIdx.x * blockDim.x + threadIdx.x; idx < (size_t)Q * per_q;
       idx += (size_t)gridDim.x * blockDim.x) {
    const uint32_t q = (uint32_t)(idx / per_q);
    const uint32_t r = (uint32_t)(idx % per_q);
    const uint32_t c = r / psz, i = r % psz;
    float v = 0.0f;
    const uint32_t j = (c < m) ? chunk_off[c] + i : 0xFFFFFFFFu;
    if (c < m && j < chunk_off[c + 1]) {
      const float qv = (j < qdim) ? (float)queries[(size_t)q * qdim + j] : 0.0f;
      v = qv - centroid[j];
    }
    qc[idx] = v;
  }
}

// ------------------------------------------------------------------------------------------
// K6 + K7: exact re-rank (compute_L2Dist :1254-1299, compute_NearestNeighbours :1312-1368)
// ------------------------------------------------------------------------------------------
#define RERANK_MAX_CAND (BANG_MAX_L + BANG_EXTRA_ITERS)

// One thread per candidate, the canonical chain (ascending dimension, fmaf) -- but the vector arrives 128 bytes at a time: eight
// 16-byte loads in flight before the first use, so a 128-byte u8 vector costs ONE memory round trip, not 32 dependent ones.
template <typename T>
__device__ __forceinline__ float elem_diff(uint32_t w, int b, const T* __restrict__ qv, uint32_t j);
template <>
__device__ __forceinline__ float elem_diff<float>(uint32_t w, int, const float* __restrict__ qv, uint32_t j) {
  return __uint_as_float(w) - qv[j];
}
template <>
__device__ __forceinline__ float elem_diff<uint8_t>(uint32_t w, int b, const uint8_t* __restrict__ qv, uint32_t j) {
  return (float)((int)((w >> (8 * b)) & 0xffu) - (int)qv[j]);                       // int subtract :1294
}
template <>
__device__ __forceinline__ float elem_diff<int8_t>(uint32_t w, int b, const int8_t* __restrict__ qv, uint32_t j) {
  return (float)((int)(int8_t)((w >> (8 * b)) & 0xffu) - (int)qv[j]);
}

template <typename T>
__device__ __forceinline__ float exact_dist(const uint8_t* __restrict__ vec, const T* __restrict__ qv, uint32_t D) {
  constexpr uint32_t EPW = 4 / sizeof(T);                 // elements per dword
  float acc = 0.0f;
  uint32_t j = 0;
  if ((((uintptr_t)vec) & 3u) == 0) {
    const uint32_t ndw = (D * (uint32_t)sizeof(T)) >> 2;
    for (uint32_t w0 = 0; w0 + 4 <= ndw; w0 += 32) {
      u32x4a w[8];
#pragma unroll
      for (int i = 0; i < 8; ++i)
        if (w0 + 4 * i + 4 <= ndw) w[i] = *(const u32x4a*)(vec + (size_t)(w0 + 4 * i) * 4);
#pragma unroll
      for (int i = 0; i < 8; ++i) {
        if (w0 + 4 * i + 4 <= ndw) {
          const uint32_t ww[4] = {w[i].x, w[i].y, w[i].z, w[i].w};
#pragma unroll
          for (int d = 0; d < 4; ++d) {
#pragma unroll
            for (int b = 0; b < (int)EPW; ++b) {
              const float diff = elem_diff<T>(ww[d], b, qv, j);
              acc = __builtin_fmaf(diff, diff, acc);
              ++j;
            }
          }
        }
      }
    }
  }
  const T* v = (const T*)vec;
  for (; j < D; ++j) {                                    // unaligned vectors, and the last < 16 bytes
    float diff;
    if constexpr (sizeof(T) == 4) diff = (float)v[j] - (float)qv[j];
    else diff = (float)((int)v[j] - (int)qv[j]);
    acc = __builtin_fmaf(diff, diff, acc);
  }
  return acc;
}

struct RerankArgs {
  const uint8_t* vec_base;
  uint64_t vec_stride;
  const uint8_t* medoid_vec;
  const void* queries;
  const uint32_t* cand_ids;
  const uint32_t* cand_row;
  const uint32_t* cand_cnt;
  uint32_t cand_stride, q0, nq, Q_total, D, k, dim_adjust;
  uint32_t by_query;      // 1: vec_base is a log [query][candidate index][vec_stride] (entry 0 = the medoid, taken from medoid_vec)
  uint32_t stage_off;     // float vectors, cooperative fetch: byte offset of the staging area in dynamic LDS (0: not provided)
  uint64_t* ids_out;
  float* dists_out;
};

template <typename T>
__global__ __launch_bounds__(256) void rerank_kernel(const RerankArgs a) {
  __shared__ float e[RERANK_MAX_CAND + 2];
  __shared__ uint32_t ids[RERANK_MAX_CAND + 2];
  __shared__ __attribute__((aligned(16))) unsigned long long keys[RERANK_MAX_CAND + 2];      // {distance bits, candidate index}
  extern __shared__ __attribute__((aligned(16))) uint8_t qraw[];
  T* qv = (T*)qraw;
  const uint32_t q = a.q0 + blockIdx.x;
  const uint32_t qdim = a.D - a.dim_adjust;
  for (uint32_t j = threadIdx.x; j < a.D; j += blockDim.x)          // MIPS zero pad :1276-1286
    qv[j] = (j < qdim) ? ((const T*)a.queries)[(size_t)q * qdim + j] : (T)0;
  uint32_t n = a.cand_cnt[q];
  if (n > RERANK_MAX_CAND) n = RERANK_MAX_CAND;
  __syncthreads();
  // 8-bit vectors of up to 256 dimensions at id * stride (dword aligned): G = D / 16 ADJACENT lanes fetch one candidate's
  // vector, 16 bytes each, in one instruction -- one request per line of the vector instead of D / 16 look-ups of it by one lane (the
  // same effect as in the PQ code rows: DESIGN 4.2).  Every partial sum of squared integer differences is an integer below 2^24
  // (256 x 255^2), exact in float in ANY order, so the lanes' partial sums add up to the bits the ascending fmaf chain produces.
  if constexpr (sizeof(T) == 1) {
    const uint32_t G = a.D >> 4;
    if (!a.by_query && !a.cand_row && (a.D & 15u) == 0 && a.D <= 256 && (G & (G - 1)) == 0 && G <= 16 && (a.vec_stride & 3u) == 0 &&
        (((uintptr_t)a.vec_base) & 3u) == 0) {
      const uint32_t per_pass = blockDim.x / G;
      const uint32_t sub = threadIdx.x % G, slot = threadIdx.x / G;
      for (uint32_t i0 = 0; i0 < n; i0 += per_pass) {
        const uint32_t i = i0 + slot;
        float acc = 0.0f;
        uint32_t id = 0;
        if (i < n) {
          id = a.cand_ids[(size_t)q * a.cand_stride + i];
          const u32x4a w = *(const u32x4a*)(a.vec_base + (uint64_t)id * a.vec_stride + 16u * sub);
          const uint32_t ww[4] = {w.x, w.y, w.z, w.w};
#pragma unroll
          for (int d = 0; d < 4; ++d)
#pragma unroll
            for (int b = 0; b < 4; ++b) {
              const float diff = elem_diff<T>(ww[d], b, qv, 16u * sub + 4u * d + b);
              acc = __builtin_fmaf(diff, diff, acc);
            }
        }
        for (uint32_t off = 1; off < G; off <<= 1) acc += __shfl_xor(acc, (int)off);     // (G lanes of one candidate are adjacent)
        if (i < n && sub == 0) { e[i] = acc; ids[i] = id; }
      }
      goto ranked;
    }
  }
  // float vectors at id * stride: a lane that walks its own candidate's 384-byte vector with 16-byte loads looks every line of it up
  // eight times, 64 different lines per instruction (DEEP100M-shape: 600 us per 10 K batch).  Here a WAVE fetches a candidate's
  // vector with one dword per lane -- contiguous, one request per line -- for B candidates at a time, hands the elements over through
  // LDS, and B lanes run the fmaf chains, each over its candidate's D elements in ascending order (the order IS the result: :1295).
  if constexpr (sizeof(T) == 4) {
    if (!a.by_query && !a.cand_row && a.stage_off && a.D <= 256 && (a.vec_stride & 3u) == 0 && (((uintptr_t)a.vec_base) & 3u) == 0) {
      const uint32_t lane = threadIdx.x & 63u, wave = threadIdx.x >> 6;
      const uint32_t B = a.D <= 128 ? 16u : 8u, NT = (a.D + 63u) >> 6, Dpad = ((a.D + 3u) & ~3u) + 4u;
      float* stage = (float*)(qraw + a.stage_off) + (size_t)wave * (B * Dpad);
      for (uint32_t i = threadIdx.x; i < n; i += blockDim.x) ids[i] = a.cand_ids[(size_t)q * a.cand_stride + i];
      __syncthreads();
      float val[16][4];
      auto fetch = [&](uint32_t b0) {                                     // one dword per lane and 64 elements: contiguous, one request per line
#pragma unroll
        for (int c = 0; c < 16; ++c) {
          if ((uint32_t)c < B) {
            const uint32_t i = b0 + (uint32_t)c;
            const float* v = (const float*)(a.vec_base + (uint64_t)ids[i < n ? i : 0u] * a.vec_stride);
#pragma unroll
            for (int t = 0; t < 4; ++t) {
              const uint32_t el = (uint32_t)t * 64u + lane;
              if ((uint32_t)t < NT) val[c][t] = v[el < a.D ? el : 0u];
            }
          }
        }
      };
      if (wave * B < n) fetch(wave * B);
      for (uint32_t b0 = wave * B; b0 < n; b0 += 4u * B) {               // (uniform per wave)
#pragma unroll
        for (int c = 0; c < 16; ++c)
          if ((uint32_t)c < B) {
#pragma unroll
            for (int t = 0; t < 4; ++t) {
              const uint32_t el = (uint32_t)t * 64u + lane;
              if ((uint32_t)t < NT && el < a.D) stage[(uint32_t)c * Dpad + el] = val[c][t];
            }
          }
        __builtin_amdgcn_wave_barrier();                                 // (one wave: its LDS operations execute in order)
        if (b0 + 4u * B < n) fetch(b0 + 4u * B);                          // the next batch travels while this one is evaluated
        const uint32_t i = b0 + lane;
        if (lane < B && i < n) {
          const float* sv = stage + lane * Dpad;
          const float* qf = (const float*)qv;
          float acc = 0.0f;
          uint32_t j = 0;
          for (; j + 4 <= a.D; j += 4) {
            const float4 x = *(const float4*)(sv + j);
            const float4 y = *(const float4*)(qf + j);
            float diff = x.x - y.x; acc = __builtin_fmaf(diff, diff, acc);
            diff = x.y - y.y; acc = __builtin_fmaf(diff, diff, acc);
            diff = x.z - y.z; acc = __builtin_fmaf(diff, diff, acc);
            diff = x.w - y.w; acc = __builtin_fmaf(diff, diff, acc);
          }
          for (; j < a.D; ++j) { const float diff = sv[j] - qf[j]; acc = __builtin_fmaf(diff, diff, acc); }
          e[i] = acc;
        }
        __builtin_amdgcn_wave_barrier();
      }
      goto ranked;
    }
  }
  for (uint32_t i = threadIdx.x; i < n; i += blockDim.x) {
    const uint32_t id = a.cand_ids[(size_t)q * a.cand_stride + i];
    const uint8_t* vec;
    if (a.by_query) {
      vec = (i == 0) ? a.medoid_vec : a.vec_base + ((uint64_t)q * a.cand_stride + i) * a.vec_stride;
    } else if (a.cand_row) {
      const uint32_t row = a.cand_row[(size_t)q * a.cand_stride + i];
      vec = (row == 0) ? a.medoid_vec : a.vec_base + ((uint64_t)row * a.Q_total + q) * a.vec_stride;
    } else {
      vec = a.vec_base + (uint64_t)id * a.vec_stride;
    }
    e[i] = exact_dist<T>(vec, qv, a.D);
    ids[i] = id;
  }
ranked:
  __syncthreads();
  // stable rank by exact distance; ties keep expansion order (:1330-1363): rank = #{(distance, index) pairs below mine}.  Exact
  // distances are sums of squares -- non-negative floats, which order like their bit patterns -- so the pair is ONE unsigned 64-bit key
  // {distance bits, index} and the rank one compare per pair (the two-level float rule costs three).
  for (uint32_t i = threadIdx.x; i < n; i += blockDim.x) keys[i] = ((unsigned long long)__float_as_uint(e[i]) << 32) | i;
  if (threadIdx.x == 0) keys[n] = ~0ull;                                   // (padding of the last pair: below nobody)
  __syncthreads();
  for (uint32_t i = threadIdx.x; i < n; i += blockDim.x) {
    const unsigned long long mine = keys[i];
    uint32_t r = 0;
    for (uint32_t j = 0; j < n; j += 2) {
      const ulonglong2 o = *(const ulonglong2*)&keys[j];
      r += (o.x < mine ? 1u : 0u) + (o.y < mine ? 1u : 0u);
    }
    if (r < a.k) {
      a.ids_out[(size_t)q * a.k + r] = (uint64_t)ids[i];                 // [Q][k] u64 :1366
      a.dists_out[(size_t)r * a.Q_total + q] = e[i];                      // [rank][Q] :999,1297
    }
  }
  for (uint32_t r = n + threadIdx.x; r < a.k; r += blockDim.x) {          // CANON tail
    a.ids_out[(size_t)q * a.k + r] = ~0ull;
    a.dists_out[(size_t)r * a.Q_total + q] = BIG_DIST;
  }
}

// ------------------------------------------------------------------------------------------
// launchers (C-ABI)
// ------------------------------------------------------------------------------------------
// Supported (PSZ, NDW) instances of the LDS-resident distance kernel.  LDS need = NDW*4*256*PSZ*4 B.
static const int kNdwList[4][4] = {{8, 16, 24, 32}, {8, 16, 18, 19}, {4, 8, 0, 0}, {2, 4, 0, 0}};
static int psz_slot(int psz) { return psz == 1 ? 0 : psz == 2 ? 1 : psz == 4 ? 2 : psz == 8 ? 3 : -1; }

extern "C" int bang_pq_layout(const uint32_t* chunk_off, uint32_t D, uint32_t m, uint32_t* psz_out, uint32_t* mp_out) {
  if (!chunk_off || m == 0 || !psz_out || !mp_out) return BANG_ERR_ARG;
  *psz_out = 0;
  *mp_out = m;
  uint32_t mx = 0;
  for (uint32_t c = 0; c < m; ++c) {
    if (chunk_off[c + 1] < chunk_off[c] || chunk_off[c + 1] > D) { bang_set_error("bad chunk offsets"); return BANG_ERR_ARG; }
    const uint32_t sz = chunk_off[c + 1] - chunk_off[c];
    if (sz > mx) mx = sz;
  }
  const int psz = mx <= 1 ? 1 : mx <= 2 ? 2 : mx <= 4 ? 4 : mx <= 8 ? 8 : 0;
  if (psz == 0) return BANG_OK;                       // LUT path
  const int need = (int)((m + 3) / 4);
  const int slot = psz_slot(psz);
  for (int i = 0; i < 4; ++i) {
    const int ndw = kNdwList[slot][i];
    if (ndw >= need && ndw > 0) {
      *psz_out = (uint32_t)psz;
      *mp_out = (uint32_t)ndw * 4;
      return BANG_OK;
    }
  }
  return BANG_OK;                                     // too many chunks for LDS: LUT path
}

extern "C" int bang_pack_pivots_ragged(const float* pivots, const uint32_t* chunk_off, uint32_t D, uint32_t m, uint32_t mp,
                                       uint32_t* nhi_out, float* out, uint64_t* floats_out) {
  if (!chunk_off || !nhi_out || m == 0 || mp < m) return BANG_ERR_ARG;
  *nhi_out = 0;
  if (floats_out) *floats_out = 0;
  uint32_t nhi = 0;
  while (nhi < m && chunk_off[nhi + 1] - chunk_off[nhi] == 2) ++nhi;
  if (nhi == 0 || nhi == m) return BANG_OK;                          // no 2-dim prefix, or nothing to save
  for (uint32_t c = nhi; c < m; ++c)
    if (chunk_off[c + 1] - chunk_off[c] > 1) return BANG_OK;       // not of the form 2,..,2,1,..,1
  const uint32_t total = pivot_table_floats(2, mp, nhi);
  *nhi_out = nhi;
  if (floats_out) *floats_out = total;
  if (!out) return BANG_OK;
  if (!pivots) return BANG_ERR_ARG;
  for (uint32_t i = 0; i < total; ++i) out[i] = 0.0f;
  for (uint32_t c = 0; c < m; ++c) {
    const uint32_t sz = chunk_off[c + 1] - chunk_off[c];
    const size_t base = c < nhi ? (size_t)c * 512 : (size_t)nhi * 256 + (size_t)c * 256;
    for (uint32_t code = 0; code < 256; ++code)
      for (uint32_t i = 0; i < sz; ++i) {
        const uint32_t j = chunk_off[c] + i;
        if (j < D) out[base + (size_t)code * (c < nhi ? 2 : 1) + i] = pivots[(size_t)code * D + j];
      }
  }
  return BANG_OK;
}

extern "C" int bang_pack_pivots(const float* pivots, const uint32_t* chunk_off, uint32_t D, uint32_t m, uint32_t psz,
                                uint32_t mp, float* out) {
  if (!pivots || !chunk_off || !out || psz == 0 || mp < m) return BANG_ERR_ARG;
  for (uint32_t c = 0; c < mp; ++c)
    for (uint32_t code = 0; code < 256; ++code)
      for (uint32_t i = 0; i < psz; ++i) {
        float v = 0.0f;
        if (c < m) {
          const uint32_t j = chunk_off[c] + i;
          if (j < chunk_off[c + 1]) v = pivots[(size_t)code * D + j];
        }
        out[((size_t)c * 256 + code) * psz + i] = v;
      }
  return BANG_OK;
}

template <typename F>
static int dispatch_dtype(int dtype, F&& f) {
  switch (dtype) {
    case BANG_U8: return f((uint8_t)0);
    case BANG_I8: return f((int8_t)0);
    case BANG_F32: return f((float)0);
    default: bang_set_error("bad dtype %d", dtype); return BANG_ERR_ARG;
  }
}

extern "C" int bang_k_center_queries(const void* d_queries, int dtype, const float* d_centroid,
                                     const uint32_t* d_chunk_off, float* d_qc, uint32_t Q, uint32_t D, uint32_t m,
                                     uint32_t mp, uint32_t psz, uint32_t dim_adjust, void* stream) {
  if (Q == 0) return BANG_OK;
  if (psz == 0 || mp < m) { bang_set_error("bad pq layout"); return BANG_ERR_ARG; }
  const size_t total = (size_t)Q * mp * psz;
  const int blocks = (int)((total + 255) / 256 < 4096 ? (total + 255) / 256 : 4096);
  return dispatch_dtype(dtype, [&](auto tag) {
    using T = decltype(tag);
    hipLaunchKernelGGL(center_queries_kernel<T>, dim3(blocks), dim3(256), 0, (hipStream_t)stream, (const T*)d_queries,
                       d_centroid, d_chunk_off, d_qc, Q, D, m, mp, psz, dim_adjust);
    HIP_TRY(hipGetLastError());
    return BANG_OK;
  });
}

extern "C" int bang_k_lut_build(const float* d_pivots_T, const void* d_queries, int dtype, const float* d_centroid,
                                const uint32_t* d_chunk_off, float* d_lut, uint32_t Q, uint32_t D, uint32_t m,
                                uint32_t dim_adjust, void* stream) {
  if (Q == 0) return BANG_OK;
  return dispatch_dtype(dtype, [&](auto tag) {
    using T = decltype(tag);
    hipLaunchKernelGGL(lut_build_kernel<T>, dim3(Q), dim3(256), D * sizeof(float), (hipStream_t)stream, d_pivots_T,
                       (const T*)d_queries, d_centroid, d_chunk_off, d_lut, D, m, dim_adjust);
    HIP_TRY(hipGetLastError());
    return BANG_OK;
  });
}

template <int PSZ, int NDW, bool ALIGNED, bool ALL, int NQW, int MAXT, int NHI = 0>
static int launch_front_inst(const FrontArgs& a, dim3 grid, dim3 block, size_t lds, hipStream_t st) {
  static bool attr_done[BANG_MAX_DEVICES] = {false};      // per kernel instance AND device
  const int dev = current_device();
  if (!attr_done[dev]) {
    HIP_TRY(hipFuncSetAttribute((const void*)front_kernel<PSZ, NDW, ALIGNED, ALL, NQW, MAXT, NHI>,
                                hipFuncAttributeMaxDynamicSharedMemorySize, 160 * 1024));
    attr_done[dev] = true;
  }
  hipLaunchKernelGGL((front_kernel<PSZ, NDW, ALIGNED, ALL, NQW, MAXT, NHI>), grid, block, lds, st, a);
  HIP_TRY(hipGetLastError());
  return BANG_OK;
}

// nqw: queries in flight per wave (1, 2 or 4 compiled); block.x <= 512 selects the 256-VGPR build
template <int PSZ, int NDW>
static int launch_front_al(const FrontArgs& a, bool aligned, int nqw, dim3 grid, dim3 block, size_t lds, hipStream_t st) {
  if (a.p.pq_nhi) {
    // exact-size pivot table: compiled for the two layouts of the BASELINE configs (128 dims in 70 chunks: 58 x 2 + 12 x 1;
    // 96 dims in 74 chunks: 22 x 2 + 52 x 1), production (ALL) form only, rows of 70 / 74 bytes are never dword aligned
    constexpr int NHI = (PSZ == 2 && NDW == 18) ? 58 : (PSZ == 2 && NDW == 19) ? 22 : 0;
    if constexpr (NHI != 0) {
      if ((int)a.p.pq_nhi == NHI && !aligned && a.stages == 7u)
        return launch_front_inst<PSZ, NDW, false, true, 4, 512, NHI>(a, grid, block, lds, st);
    }
    bang_set_error("no kernel instance for the exact-size pivot table psz=%u mp=%u nhi=%u", a.p.psz, a.p.mp, a.p.pq_nhi);
    return BANG_ERR_UNSUPPORTED;
  }
  const bool all = (a.stages == 7u);
  if (!all) {
    return aligned ? launch_front_inst<PSZ, NDW, true, false, 1, 1024>(a, grid, block, lds, st)
                   : launch_front_inst<PSZ, NDW, false, false, 1, 1024>(a, grid, block, lds, st);
  }
  const bool small = block.x <= 512;
#define BANG_FRONT_PICK(AL)                                                                                    \
  switch (nqw) {                                                                                               \
    case 1: return launch_front_inst<PSZ, NDW, AL, true, 1, 1024>(a, grid, block, lds, st);                    \
    default: return small ? launch_front_inst<PSZ, NDW, AL, true, 4, 512>(a, grid, block, lds, st)             \
                          : launch_front_inst<PSZ, NDW, AL, true, 4, 1024>(a, grid, block, lds, st);           \
  }
  if (aligned) { BANG_FRONT_PICK(true) }
  BANG_FRONT_PICK(false)
#undef BANG_FRONT_PICK
}

static int launch_front(const bang_iter_params* p, uint32_t stages, void* stream) {
  if (!p) return BANG_ERR_ARG;
  if (p->Q == 0) return BANG_OK;
  if (p->R > BANG_MAX_R || p->L > BANG_MAX_L || p->m == 0) { bang_set_error("bad R/L/m"); return BANG_ERR_ARG; }
  if (p->psz != 0 && (p->mp < p->m || (p->mp & 3u))) { bang_set_error("bad padded chunk count"); return BANG_ERR_ARG; }
  if (!p->d_nbrs || !p->d_dist || !p->d_cnt || !p->d_codes || !p->d_seed) { bang_set_error("null buffer"); return BANG_ERR_ARG; }
  if ((stages & 4u) && (!p->d_wl_ids || !p->d_wl_dist || !p->d_wl_vis || !p->d_wl_cnt || !p->d_mark || !p->d_parents ||
                        !p->d_cand_ids || !p->d_cand_cnt)) { bang_set_error("null worklist/candidate buffer"); return BANG_ERR_ARG; }
  if (p->h_done_flag && (!p->d_done_count || !p->h_parents)) { bang_set_error("completion flag needs d_done_count and h_parents"); return BANG_ERR_ARG; }
  if ((stages & 1u) && (!p->d_bloom || (!p->first && !p->d_stage && !p->d_graph))) { bang_set_error("null filter buffer"); return BANG_ERR_ARG; }
  if ((stages & 2u) && (p->psz ? (!p->d_pivots_packed || !p->d_qc) : !p->d_lut)) { bang_set_error("null PQ buffer"); return BANG_ERR_ARG; }
  FrontArgs a;
  a.p = *p;
  a.stages = stages;
  {
    static int dbg = -1;
    if (dbg < 0) { const char* v = getenv("BANG_FRONT_DEBUG"); dbg = v ? atoi(v) : 0; }
    a.debug = (uint32_t)dbg;
  }
  // the kernel issues its independent loads unconditionally: give unused optional inputs a valid address
  {
    bang_iter_params& x = a.p;
    const uint32_t* dummy = p->d_seed;   // >= 67 words
    if (!x.d_stage) x.d_stage = dummy;
    if (!x.d_wl_cnt) x.d_wl_cnt = x.d_cnt;
    if (!x.d_cand_cnt) x.d_cand_cnt = x.d_cnt;
    if (!x.d_wl_vis) x.d_wl_vis = (uint8_t*)x.d_nbrs;
    if (!x.d_wl_dist) x.d_wl_dist = x.d_dist;
    if (!x.d_wl_ids) x.d_wl_ids = x.d_nbrs;
    if (!x.d_bloom) x.d_bloom = x.d_nbrs;
  }
  const bool need_piv = (p->psz != 0) && (stages & 2u);
  if (p->pq_nhi && (p->psz != 2 || p->pq_nhi > p->mp)) { bang_set_error("bad pq_nhi"); return BANG_ERR_ARG; }
  a.lds_piv_floats = need_piv ? pivot_table_floats(p->psz, p->mp, p->pq_nhi) : 0u;
  const size_t piv_bytes = (size_t)a.lds_piv_floats * 4;
  const size_t lds_cap = 160 * 1024;
  // One workgroup per CU at most (the pivot table takes most of the LDS); a lane that shares the GPU
  // with other lanes gets max_wgs of them.  Each wave keeps nqw queries in flight; waves per workgroup:
  // enough to cover Q in one sweep if possible, bounded by LDS.
  static int env_nqw = -1, env_waves = -1;
  if (env_nqw < 0) { const char* v = getenv("BANG_FRONT_NQW"); env_nqw = v ? atoi(v) : 0; }
  if (env_waves < 0) { const char* v = getenv("BANG_FRONT_WAVES"); env_waves = v ? atoi(v) : 0; }
  int wgs = need_piv ? num_cus() : num_cus() * 8;
  if (p->max_wgs && (int)p->max_wgs < wgs) wgs = (int)p->max_wgs;
  const int per_wg = (int)((p->Q + (uint32_t)wgs - 1) / (uint32_t)wgs);   // queries a workgroup must cover
  // interleaving 4 queries per wave did not pay on SIFT1M-like data (the kernel is bound by random-access
  // throughput of the visited filter, not by dependent latency); kept selectable for other shapes
  // Layouts with > 32 chunk-dwords x floats per entry (m = 68..76 at 2 floats per entry: SIFT1B, DEEP100M) need more
  // than 128 VGPRs for the straight-line distance code: run them as <= 8 waves (256-VGPR budget, no spills) with 4
  // queries in flight per wave instead of 16 waves x 1 query.
  const bool heavy = p->psz != 0 && p->psz * (p->mp / 4u) > 32u;
  int nqw = (stages != 7u) ? 1 : (env_nqw > 0 ? env_nqw : (heavy ? 4 : 1));
  nqw = (nqw >= 2) ? 4 : 1;
  int max_waves = env_waves > 0 ? env_waves : ((heavy && stages == 7u) ? 8 : 16);
  if (p->pq_nhi) { nqw = 4; if (max_waves > 8) max_waves = 8; }      // the exact-size instances are 8 waves x 4 queries in flight
  if (max_waves > 16) max_waves = 16;
  int waves = (per_wg + nqw - 1) / nqw;
  if (waves < 1) waves = 1;
  if (waves > max_waves) waves = max_waves;
  const size_t scratch_per_wave = (size_t)FRONT_SCRATCH_WORDS * 4 * (size_t)nqw;
  while (waves > 1 && piv_bytes + (size_t)waves * scratch_per_wave > lds_cap) --waves;
  const size_t lds = piv_bytes + (size_t)waves * scratch_per_wave;
  if (lds > lds_cap) { bang_set_error("pivot table does not fit LDS (%zu B)", lds); return BANG_ERR_UNSUPPORTED; }
  int grid_n = (int)((p->Q + (uint32_t)(waves * nqw) - 1) / (uint32_t)(waves * nqw));
  if (grid_n > wgs) grid_n = wgs;
  const dim3 grid(grid_n), block(waves * WAVE);
  const bool al = ((p->code_stride ? p->code_stride : p->m) % 4u) == 0;
  hipStream_t st = (hipStream_t)stream;
  const uint32_t key = p->psz * 100u + (p->psz ? p->mp / 4u : 0u);
  switch (key) {
    case 0: return launch_front_al<0, 1>(a, true, nqw, grid, block, lds, st);
    case 108: return launch_front_al<1, 8>(a, al, nqw, grid, block, lds, st);
    case 116: return launch_front_al<1, 16>(a, al, nqw, grid, block, lds, st);
    case 124: return launch_front_al<1, 24>(a, al, nqw, grid, block, lds, st);
    case 132: return launch_front_al<1, 32>(a, al, nqw, grid, block, lds, st);
    case 208: return launch_front_al<2, 8>(a, al, nqw, grid, block, lds, st);
    case 216: return launch_front_al<2, 16>(a, al, nqw, grid, block, lds, st);
    case 218: return launch_front_al<2, 18>(a, al, nqw, grid, block, lds, st);
    case 219: return launch_front_al<2, 19>(a, al, nqw, grid, block, lds, st);
    case 404: return launch_front_al<4, 4>(a, al, nqw, grid, block, lds, st);
    case 408: return launch_front_al<4, 8>(a, al, nqw, grid, block, lds, st);
    case 802: return launch_front_al<8, 2>(a, al, nqw, grid, block, lds, st);
    case 804: return launch_front_al<8, 4>(a, al, nqw, grid, block, lds, st);
    default: bang_set_error("no kernel instance for psz=%u mp=%u", p->psz, p->mp); return BANG_ERR_UNSUPPORTED;
  }
}

extern "C" int bang_k_front(const bang_iter_params* p, void* stream) { return launch_front(p, 7u, stream); }
extern "C" int bang_num_cus(void) { return num_cus(); }

// Is there a kernel instance for the exact-size pivot table of this layout (see launch_front_al)?
extern "C" int bang_ragged_supported(uint32_t psz, uint32_t mp, uint32_t nhi, uint32_t m) {
  return (psz == 2 && (m & 3u) != 0 && ((mp == 72 && nhi == 58) || (mp == 76 && nhi == 22))) ? 1 : 0;
}

extern "C" int bang_k_filter(const bang_iter_params* p, void* stream) { return launch_front(p, 1u, stream); }
extern "C" int bang_k_pqdist(const bang_iter_params* p, void* stream) { return launch_front(p, 2u, stream); }
extern "C" int bang_k_parent(const bang_iter_params* p, void* stream) { return launch_front(p, 4u, stream); }

extern "C" int bang_k_back(const bang_iter_params* p, void* stream) {
  if (!p) return BANG_ERR_ARG;
  if (p->Q == 0) return BANG_OK;
  if (p->L > BANG_MAX_L || p->L == 0) { bang_set_error("bad L"); return BANG_ERR_ARG; }
  int grid = (int)((p->Q + BACK_WAVES - 1) / BACK_WAVES);
  const int max_grid = num_cus() * 8;
  if (grid > max_grid) grid = max_grid;
  hipLaunchKernelGGL(back_kernel, dim3(grid), dim3(BACK_WAVES * WAVE), 0, (hipStream_t)stream, *p);
  HIP_TRY(hipGetLastError());
  return BANG_OK;
}

// per-batch state reset (device side of bang_init, bang_search.cu:440-464): candidate log row 0 = MEDOID
__global__ void init_state_kernel(uint32_t Q, uint32_t medoid, uint32_t cand_stride, uint32_t* cand_ids, uint32_t* cand_row,
                                  uint32_t* cand_cnt, uint32_t* wl_cnt, uint32_t* mark, uint32_t* parents, uint32_t* cnt) {
  for (uint32_t q = blockIdx.x * blockDim.x + threadIdx.x; q < Q; q += gridDim.x * blockDim.x) {
    cand_ids[(size_t)q * cand_stride] = medoid;
    if (cand_row) cand_row[(size_t)q * cand_stride] = 0;
    cand_cnt[q] = 1;
    wl_cnt[q] = 0;
    mark[q] = 0x01010101u;   // cudaMemset(d_mark, 1, ...) :446
    if (parents) parents[q] = BANG_NO_PARENT;
    cnt[q] = 0;
  }
}

// bang_init in ONE launch (bang_search.cu:427-507): the visited filters (Q x 50 KB, 16-byte stores), the per-query state above and the
// diagnostic counters -- instead of four memsets, a kernel and a device-wide synchronisation (1.15 ms per 10 K batch: VERDICT r3)
__global__ __launch_bounds__(256) void init_all_kernel(bang_init_params a) {
  const size_t tid = (size_t)blockIdx.x * blockDim.x + threadIdx.x, nt = (size_t)gridDim.x * blockDim.x;
  uint4* b = (uint4*)a.d_bloom;
  const size_t n4 = (size_t)a.Q * (BANG_BF_WORDS / 4);
  const uint4 z = make_uint4(0u, 0u, 0u, 0u);
  for (size_t i = tid; i < n4; i += nt) b[i] = z;
  for (size_t q = tid; q < a.Q; q += nt) {
    a.d_cand_ids[q * a.cand_stride] = a.medoid;
    if (a.d_cand_row) a.d_cand_row[q * a.cand_stride] = 0;
    a.d_cand_cnt[q] = 1;
    a.d_wl_cnt[q] = 0;
    a.d_mark[q] = 0x01010101u;   // cudaMemset(d_mark, 1, ...) :446
    if (a.d_parents) a.d_parents[q] = BANG_NO_PARENT;
    a.d_cnt[q] = 0;
    if (a.d_qstats) { a.d_qstats[2 * q] = 0; a.d_qstats[2 * q + 1] = 0; }
    if (a.d_qskip) a.d_qskip[q] = 0;
  }
  if (a.d_active) for (size_t i = tid; i < a.n_active; i += nt) a.d_active[i] = 0;
}
static_assert(BANG_BF_WORDS % 4 == 0, "filters are cleared with 16-byte stores");

extern "C" int bang_k_init_all(const bang_init_params* a, void* stream) {
  if (!a || !a->d_bloom || !a->d_cand_ids || !a->d_cand_cnt || !a->d_wl_cnt || !a->d_mark || !a->d_cnt) return BANG_ERR_ARG;
  if (a->Q == 0) return BANG_OK;
  const size_t n4 = (size_t)a->Q * (BANG_BF_WORDS / 4);
  size_t blocks = (n4 + 256 * 8 - 1) / (256 * 8);                   // >= 8 stores per thread
  const size_t most = (size_t)num_cus() * 8;
  if (blocks > most) blocks = most;
  if (blocks == 0) blocks = 1;
  hipLaunchKernelGGL(init_all_kernel, dim3((unsigned)blocks), dim3(256), 0, (hipStream_t)stream, *a);
  HIP_TRY(hipGetLastError());
  return BANG_OK;
}

extern "C" int bang_k_init_state(uint32_t Q, uint32_t medoid, uint32_t cand_stride, uint32_t* d_cand_ids,
                                 uint32_t* d_cand_row, uint32_t* d_cand_cnt, uint32_t* d_wl_cnt, uint32_t* d_mark,
                                 uint32_t* d_parents, uint32_t* d_cnt, void* stream) {
  if (Q == 0) return BANG_OK;
  const int blocks = (int)((Q + 255) / 256);
  hipLaunchKernelGGL(init_state_kernel, dim3(blocks), dim3(256), 0, (hipStream_t)stream, Q, medoid, cand_stride,
                     d_cand_ids, d_cand_row, d_cand_cnt, d_wl_cnt, d_mark, d_parents, d_cnt);
  HIP_TRY(hipGetLastError());
  return BANG_OK;
}

extern "C" int bang_k_rerank_range(const void* d_vec_base, uint64_t vec_stride, const void* d_medoid_vec,
                                   const void* d_queries, int dtype, const uint32_t* d_cand_ids,
                                   const uint32_t* d_cand_row, const uint32_t* d_cand_cnt, uint32_t cand_stride,
                                   uint32_t q0, uint32_t nq, uint32_t Q_total, uint32_t D, uint32_t k,
                                   uint32_t dim_adjust, uint64_t* d_ids_out, float* d_dists_out, void* stream) {
  if (nq == 0) return BANG_OK;
  if (k == 0 || k > BANG_MAX_L) { bang_set_error("bad k"); return BANG_ERR_ARG; }
  RerankArgs a;
  a.vec_base = (const uint8_t*)d_vec_base; a.vec_stride = vec_stride; a.medoid_vec = (const uint8_t*)d_medoid_vec;
  a.queries = d_queries; a.cand_ids = d_cand_ids; a.cand_row = d_cand_row; a.cand_cnt = d_cand_cnt;
  a.cand_stride = cand_stride; a.q0 = q0; a.nq = nq; a.Q_total = Q_total; a.D = D; a.k = k; a.dim_adjust = dim_adjust;
  a.by_query = (d_cand_row == (const uint32_t*)(uintptr_t)1) ? 1u : 0u;
  if (a.by_query) a.cand_row = nullptr;
  a.ids_out = d_ids_out; a.dists_out = d_dists_out;
  return dispatch_dtype(dtype, [&](auto tag) {
    using T = decltype(tag);
    size_t lds = ((size_t)D * sizeof(T) + 15) & ~(size_t)15;
    a.stage_off = 0;
    if (sizeof(T) == 4 && D <= 256 && !a.by_query && !a.cand_row) {      // staging area of the cooperative float fetch: 4 waves x B candidates x Dpad floats
      const size_t B = D <= 128 ? 16 : 8, Dpad = ((D + 3) & ~(size_t)3) + 4;
      a.stage_off = (uint32_t)lds;
      lds += (size_t)4 * B * Dpad * sizeof(float);
    }
    hipLaunchKernelGGL(rerank_kernel<T>, dim3(nq), dim3(256), lds,
                       (hipStream_t)stream, a);
    HIP_TRY(hipGetLastError());
    return BANG_OK;
  });
}

extern "C" int bang_k_rerank_byquery(const void* d_fp, uint64_t vec_stride, const void* d_medoid_vec, const void* d_queries, int dtype,
                                     const uint32_t* d_cand_ids, const uint32_t* d_cand_cnt, uint32_t cand_stride, uint32_t q0,
                                     uint32_t nq, uint32_t Q_total, uint32_t D, uint32_t k, uint32_t dim_adjust, uint64_t* d_ids_out,
                                     float* d_dists_out, void* stream) {
  return bang_k_rerank_range(d_fp, vec_stride, d_medoid_vec, d_queries, dtype, d_cand_ids, (const uint32_t*)(uintptr_t)1, d_cand_cnt,
                             cand_stride, q0, nq, Q_total, D, k, dim_adjust, d_ids_out, d_dists_out, stream);
}

extern "C" int bang_k_rerank(const void* d_vec_base, uint64_t vec_stride, const void* d_medoid_vec, const void* d_queries,
                             int dtype, const uint32_t* d_cand_ids, const uint32_t* d_cand_row, const uint32_t* d_cand_cnt,
                             uint32_t cand_stride, uint32_t Q, uint32_t D, uint32_t k, uint32_t dim_adjust,
                             uint64_t* d_ids_out, float* d_dists_out, void* stream) {
  return bang_k_rerank_range(d_vec_base, vec_stride, d_medoid_vec, d_queries, dtype, d_cand_ids, d_cand_row, d_cand_cnt,
                             cand_stride, 0, Q, Q, D, k, dim_adjust, d_ids_out, d_dists_out, stream);
}
