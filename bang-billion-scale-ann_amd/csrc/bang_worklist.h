// K3a + K3b of the search kernel on LDS-resident state: stable rank sort of the <= 65 survivors of an iteration and their merge into the
// sorted worklist (compute_BestLSets_par_sort_msort bang_search.cu:1533-1585, compute_BestLSets_par_merge :1605-1715), and the scan for the
// first unvisited entry that compute_parent2 (:1425-1439) starts from.  Wave-level device code; included by bang_search.hip only.
#pragma once
#include "bang_device.h"

// ---------------------------------------------------------------------------------------------------------------------
// K3a + K3b on LDS-resident state (compute_BestLSets_par_sort_msort :1533-1585, compute_BestLSets_par_merge :1605-1715)
// ---------------------------------------------------------------------------------------------------------------------
struct WaveLds {
  float* wd; uint32_t* wi; uint8_t* wv;     // worklist: distances, ids, visited flags (sorted ascending by distance)
  float* sd; uint32_t* ti;                  // unsorted survivor distances; after ranking: sorted ids (same words)
  float* td;                                // sorted survivor distances; before the distance stage: compaction scratch
};

// stable rank of element (d, i) among sd[0, n): #smaller + #equal with a lower index (== the reference's stable merge sort)
__device__ __forceinline__ uint32_t rank_in(const float* sd, uint32_t n8, float d, uint32_t i) {
  uint32_t r = 0;
  for (uint32_t j = 0; j < n8; j += 8) {
    const float4 o0 = *(const float4*)(sd + j);
    const float4 o1 = *(const float4*)(sd + j + 4);
    r += (o0.x < d || (o0.x == d && j + 0 < i)) ? 1u : 0u;
    r += (o0.y < d || (o0.y == d && j + 1 < i)) ? 1u : 0u;
    r += (o0.z < d || (o0.z == d && j + 2 < i)) ? 1u : 0u;
    r += (o0.w < d || (o0.w == d && j + 3 < i)) ? 1u : 0u;
    r += (o1.x < d || (o1.x == d && j + 4 < i)) ? 1u : 0u;
    r += (o1.y < d || (o1.y == d && j + 5 < i)) ? 1u : 0u;
    r += (o1.z < d || (o1.z == d && j + 6 < i)) ? 1u : 0u;
    r += (o1.w < d || (o1.w == d && j + 7 < i)) ? 1u : 0u;
  }
  return r;
}

// merge of the nb leading sorted survivors into the worklist, in place (every read precedes every write: one wave, LDS in order)
template <int WLR>
__device__ __forceinline__ uint32_t merge_in_lds(const WaveLds& s, uint32_t n, uint32_t w_n, uint32_t L, uint32_t mark, float worst,
                                                 int lane) {
  const uint32_t lim = L < n ? L : n;
  uint32_t nb = lim;                                            // leading survivors with dist < worst (stop at the first >=) :1653-1657
  {
    const bool ge = ((uint32_t)lane < lim) && (s.td[lane] >= worst);
    const uint64_t mk = __ballot(ge);
    if (mk) nb = (uint32_t)__builtin_ctzll(mk);
    else if (lim > 64 && s.td[64] >= worst) nb = 64;
  }
  const uint32_t room = L - w_n;
  const uint32_t fill = room < n ? room : n;
  if (fill > nb) nb = fill;                                     // :1660
  const uint32_t new_n = (w_n + nb) < L ? (w_n + nb) : L;       // :1662
  float od[WLR];
  uint32_t oi[WLR], po[WLR], ov = 0;
#pragma unroll
  for (int j = 0; j < WLR; ++j) {
    const uint32_t k = (uint32_t)lane + (uint32_t)j * WAVE;
    od[j] = 0.0f; oi[j] = 0; po[j] = 0xFFFFFFFFu;
    if (k < w_n) { od[j] = s.wd[k]; oi[j] = s.wi[k]; ov |= (uint32_t)s.wv[k] << j; }
  }
  uint32_t pn = 0xFFFFFFFFu, pn64 = 0xFFFFFFFFu, idn = 0, idn64 = 0;
  float dn = 0.0f, dn64 = 0.0f;
  if ((uint32_t)lane < nb) {                                    // new entries: lower_bound + i :1675-1677
    dn = s.td[lane]; idn = s.ti[lane];
    pn = lower_bound_lds(s.wd, w_n, dn) + (uint32_t)lane;
  }
  if (nb > 64 && lane == 0) {
    dn64 = s.td[64]; idn64 = s.ti[64];
    pn64 = lower_bound_lds(s.wd, w_n, dn64) + 64u;
  }
#pragma unroll
  for (int j = 0; j < WLR; ++j) {                               // old entries: upper_bound + k :1678-1680
    const uint32_t k = (uint32_t)lane + (uint32_t)j * WAVE;
    if (k < w_n) po[j] = upper_bound_lds(s.td, nb, od[j]) + k;
  }
  wave_sync();
  if (pn < new_n) { s.wi[pn] = idn; s.wd[pn] = dn; s.wv[pn] = (idn == mark) ? 1 : 0; }
  if (pn64 < new_n) { s.wi[pn64] = idn64; s.wd[pn64] = dn64; s.wv[pn64] = (idn64 == mark) ? 1 : 0; }
#pragma unroll
  for (int j = 0; j < WLR; ++j)
    if (po[j] < new_n) { s.wi[po[j]] = oi[j]; s.wd[po[j]] = od[j]; s.wv[po[j]] = (((ov >> j) & 1u) || oi[j] == mark) ? 1 : 0; }   // + mark step :1711-1714
  wave_sync();
  return new_n;
}

// The common case of K3a + K3b -- worklist full (w_n == L), survivors in lanes (n <= 64) -- without sorting and without binary
// searches.  Only survivors closer than the worklist's last entry can enter (:1653-1657; with no room left that IS the reference's
// nb), and the merged position of every element is a count:
//   new e : #{old : d_old < d_e}  (lower_bound :1675)  +  #{new f : d_f < d_e, or d_f == d_e and f before e}  (its stable rank :1559-1567)
//   old k : k  +  #{new e : d_e <= d_k}               (upper_bound :1678)
// One pass over the (few) entering survivors, broadcast from their lanes with v_readlane, updates all three counts in registers.
template <int WLR>
__device__ __forceinline__ void merge_few(const WaveLds& s, uint64_t m_in, float d0, uint32_t id0, uint32_t L, uint32_t mark, int lane) {
  float od[WLR];
  uint32_t oi[WLR], cnt[WLR], ov = 0;
#pragma unroll
  for (int j = 0; j < WLR; ++j) {
    const uint32_t k = (uint32_t)lane + (uint32_t)j * WAVE;
    od[j] = __builtin_inff(); oi[j] = 0; cnt[j] = 0;              // +inf: counted by no survivor's lower_bound, written nowhere
    if (k < L) { od[j] = s.wd[k]; oi[j] = s.wi[k]; ov |= (uint32_t)s.wv[k] << j; }
  }
  uint32_t rank = 0, below = 0;
  for (uint64_t mm = m_in; mm; mm &= mm - 1) {                     // uniform loop, input order
    const int e = __builtin_ctzll(mm);
    const float de = __uint_as_float((uint32_t)__builtin_amdgcn_readlane((int)__float_as_uint(d0), e));
    rank += (de < d0 || (de == d0 && e < lane)) ? 1u : 0u;
    uint32_t pp = 0;
#pragma unroll
    for (int j = 0; j < WLR; ++j) {
      cnt[j] += (de <= od[j]) ? 1u : 0u;
      pp += (uint32_t)__popcll(__ballot(od[j] < de));
    }
    if (lane == e) below = pp;
  }
  wave_sync();                                                     // every read of the old worklist precedes every write
  if ((m_in >> lane) & 1ull) {
    const uint32_t pn = below + rank;
    if (pn < L) { s.wi[pn] = id0; s.wd[pn] = d0; s.wv[pn] = (id0 == mark) ? 1 : 0; }
  }
#pragma unroll
  for (int j = 0; j < WLR; ++j) {
    const uint32_t k = (uint32_t)lane + (uint32_t)j * WAVE;
    const uint32_t po = k + cnt[j];
    if (k < L && po < L) { s.wi[po] = oi[j]; s.wd[po] = od[j]; s.wv[po] = (((ov >> j) & 1u) || oi[j] == mark) ? 1 : 0; }   // + mark step :1711-1714
  }
  wave_sync();
}

// sort the n survivors (lane i < 64 holds survivor i, lane 0 also survivor 64) and merge them into the worklist; returns the new length.
// worst = distance of the worklist's last entry (iter > 1)
__device__ __forceinline__ uint32_t sort_and_merge(const WaveLds& s, uint32_t n, float d0, uint32_t id0, float d1, uint32_t id1,
                                                   uint32_t iter, uint32_t w_n, uint32_t L, uint32_t medoid, uint32_t mark, float worst,
                                                   int lane) {
  if (iter > 1 && w_n == L && n <= 64) {
    const uint64_t m_in = __ballot((uint32_t)lane < n && d0 < worst);
    if (m_in == 0) return L;                                       // nobody enters: the merge is the identity (a parent taken from the survivors enters)
    if ((uint32_t)__popcll(m_in) <= L) {                           // (more than L entering survivors: only with L < 64; general path)
      const uint32_t wlr = (L + WAVE - 1) / WAVE;
      if (wlr <= 1) merge_few<1>(s, m_in, d0, id0, L, mark, lane);
      else if (wlr <= 2) merge_few<2>(s, m_in, d0, id0, L, mark, lane);
      else if (wlr <= 4) merge_few<4>(s, m_in, d0, id0, L, mark, lane);
      else merge_few<8>(s, m_in, d0, id0, L, mark, lane);
      return L;
    }
  }
  const float inf = __builtin_inff();
  s.sd[lane] = ((uint32_t)lane < n) ? d0 : inf;
  if (lane < 8) s.sd[64 + lane] = (lane == 0 && n > 64) ? d1 : inf;
  wave_sync();
  // K3a: stable rank sort.  The padding (+inf) ranks behind every real distance and ties with none.
  const uint32_t n8 = (n + 7u) & ~7u;
  uint32_t r0 = 0, r64 = 0;
  if ((uint32_t)lane < n) r0 = rank_in(s.sd, n8, d0, (uint32_t)lane);
  if (n > 64) { if (lane == 0) r64 = rank_in(s.sd, n8, d1, 64u); }
  wave_sync();
  if ((uint32_t)lane < n) { s.td[r0] = d0; s.ti[r0] = id0; }
  if (n > 64 && lane == 0) { s.td[r64] = d1; s.ti[r64] = id1; }
  wave_sync();
  if (iter == 1) {                                              // :1638-1649
    const uint32_t new_n = n < L ? n : L;
    for (uint32_t i = lane; i < new_n; i += WAVE) {
      const uint32_t id = s.ti[i];
      s.wi[i] = id; s.wd[i] = s.td[i];
      s.wv[i] = (id == medoid || id == mark) ? 1 : 0;           // + mark step :1711-1714
    }
    wave_sync();
    return new_n;
  }
  const uint32_t wlr = (w_n + WAVE - 1) / WAVE;                 // uniform: registers for the old entries a lane owns
  if (wlr <= 1) return merge_in_lds<1>(s, n, w_n, L, mark, worst, lane);
  if (wlr <= 2) return merge_in_lds<2>(s, n, w_n, L, mark, worst, lane);
  if (wlr <= 4) return merge_in_lds<4>(s, n, w_n, L, mark, worst, lane);
  return merge_in_lds<8>(s, n, w_n, L, mark, worst, lane);
}

// What K4 of the NEXT iteration needs from the worklist (compute_parent2 :1425-1446) -- its first unvisited entry and its last
// distance -- is known as soon as this iteration's merge is done: fetched here, held in scalar registers, so that the parent
// decision behind the distance stage is one compare instead of three dependent LDS round trips.
struct WlHead { bool found; uint32_t idx, id; float d, tail; };
__device__ __forceinline__ WlHead worklist_head(const WaveLds& s, uint32_t w_n, int lane) {
  WlHead h;
  h.found = false; h.idx = 0; h.id = 0; h.d = 0.0f; h.tail = 0.0f;
  if (w_n == 0) return h;
  for (uint32_t base = 0; base < w_n && !h.found; base += WAVE) {          // first unvisited entry :1425-1439
    const uint32_t i = base + (uint32_t)lane;
    const uint64_t mk = __ballot(i < w_n && s.wv[i < w_n ? i : 0] == 0);
    if (mk) { h.idx = base + (uint32_t)__builtin_ctzll(mk); h.found = true; }
  }
  const float t = s.wd[w_n - 1];
  const float d = s.wd[h.idx];
  const uint32_t id = s.wi[h.idx];
  h.tail = __uint_as_float(uni(__float_as_uint(t)));
  h.d = __uint_as_float(uni(__float_as_uint(d)));
  h.id = uni(id);
  return h;
}
