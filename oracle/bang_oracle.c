/*
 * bang_oracle.c -- CPU restatement of the BANG_Base search hot path (see bang_oracle.h).
 *
 * TEST INFRASTRUCTURE ONLY -- never linked into the product.  PARITY UNPINNED BY THE
 * REFERENCE (it cannot be built here and ships no golden vectors); pinned by hand-derived
 * known answers in tests/test_oracle_kat.py.
 *
 * Build: gcc -O2 -std=c11 -ffp-contract=off -fopenmp -fPIC -shared (oracle/Makefile).
 * -ffp-contract=off matters: every fused multiply-add below is an explicit fmaf() so
 * that the float results are the same on every compiler and on the GPU.
 *
 * All line numbers refer to /root/reference/BANG_Base/bang_search.cu unless stated.
 */
#include "bang_oracle.h"

#include <math.h>
#include <stdlib.h>
#include <string.h>
#ifdef _OPENMP
#include <omp.h>
#endif

/* ------------------------------------------------------------------ hashes */

/* hashFn1_d :1168-1178.  FNV-1a style, 4 rounds over the id's bytes LSB->MSB, in
 * uint64 wrap-around arithmetic, then mod BF_ENTRIES. */
uint32_t orc_hash1(uint32_t x) {
  uint64_t h = 0xcbf29ce4ull;
  h = (h ^ (uint64_t)(x & 0xff)) * 0x01000193ull;
  h = (h ^ (uint64_t)((x >> 8) & 0xff)) * 0x01000193ull;
  h = (h ^ (uint64_t)((x >> 16) & 0xff)) * 0x01000193ull;
  h = (h ^ (uint64_t)((x >> 24) & 0xff)) * 0x01000193ull;
  return (uint32_t)(h % ORC_BF_ENTRIES);
}

/* hashFn2_d :1180-1189 */
uint32_t orc_hash2(uint32_t x) {
  uint64_t h = 0x84222325ull;
  h = (h ^ (uint64_t)(x & 0xff)) * 0x1B3ull;
  h = (h ^ (uint64_t)((x >> 8) & 0xff)) * 0x1B3ull;
  h = (h ^ (uint64_t)((x >> 16) & 0xff)) * 0x1B3ull;
  h = (h ^ (uint64_t)((x >> 24) & 0xff)) * 0x1B3ull;
  return (uint32_t)(h % ORC_BF_ENTRIES);
}

/* ------------------------------------------------------------------ K1 */

static inline float load_as_float(const void *v, uint32_t j, int dtype) {
  switch (dtype) {
    case ORC_U8: return (float)((const uint8_t *)v)[j];
    case ORC_I8: return (float)((const int8_t *)v)[j];
    default: return ((const float *)v)[j];
  }
}

/* populate_pqDist_par :1083-1130.
 * LUT[c][k] = sum_{j=off[c]}^{off[c+1]-1} diff^2, diff = P_T[j][k] - (float(q[j]) - cen[j]),
 * accumulated in ascending j starting from +0.0f (table is memset to 0 in bang_init :442
 * and updated with `+=` :1126).  CANON: the `+= diff*diff` is an explicit fmaf (what
 * nvcc's default -fmad=true emits for that statement).
 * MIPS (:1099-1113): the query carries D-dim_adjust values, the tail is 0. */
void orc_lut_build(const orc_index *ix, const void *query, int dim_adjust, float *lut_out) {
  const uint32_t D = ix->D, m = ix->m;
  const uint32_t qdim = D - (uint32_t)dim_adjust;
  for (uint32_t c = 0; c < m; ++c) {
    float *row = lut_out + (size_t)c * 256;
    for (uint32_t k = 0; k < 256; ++k) row[k] = 0.0f;
    for (uint32_t j = ix->chunk_off[c]; j < ix->chunk_off[c + 1]; ++j) {
      const float qv = (j < qdim) ? load_as_float(query, j, ix->dtype) : 0.0f;
      const float qc = qv - ix->centroid[j]; /* (T)query_vec[j] - shm_centroid[j] :1124 */
      const float *piv = ix->pivots_T + (size_t)j * 256;
      for (uint32_t k = 0; k < 256; ++k) {
        const float diff = piv[k] - qc;
        row[k] = fmaf(diff, diff, row[k]);
      }
    }
  }
}

/* ------------------------------------------------------------------ K5 */

/* neighbor_filtering_new :1140-1165.  Keep x iff NOT(bit[h1] AND bit[h2]), then set both.
 * CANON (the reference is racy across threads :1157-1161 and emits in atomicAdd order):
 * every id of the batch is tested against the filter state at entry, then all survivors
 * set their bits; survivors keep input order. */
uint32_t orc_filter(uint8_t *bloom, const uint32_t *in, uint32_t n_in, uint32_t *out) {
  uint32_t n_out = 0;
  for (uint32_t i = 0; i < n_in; ++i) {
    const uint32_t x = in[i];
    if (!(bloom[orc_hash1(x)] && bloom[orc_hash2(x)])) out[n_out++] = x;
  }
  for (uint32_t i = 0; i < n_out; ++i) {
    bloom[orc_hash1(out[i])] = 1;
    bloom[orc_hash2(out[i])] = 1;
  }
  return n_out;
}

/* ------------------------------------------------------------------ K2 */

/* compute_neighborDist_par :1201-1241.  8 threads per neighbour (:1225), thread l sums
 * chunks c = l, l+8, ... in ascending order starting from 0.0f (:1233-1237), then
 * cub::WarpReduce<float,8>::Sum (:1239) = shfl-down tree with offsets 1,2,4 whose lane-0
 * value is ((s0+s1)+(s2+s3))+((s4+s5)+(s6+s7)).  CANON: that lane-0 value is the result
 * (the reference stores from all 8 lanes to one address). */
void orc_pqdist(const float *lut, const uint8_t *codes, uint32_t m, const uint32_t *ids,
                uint32_t n, float *dist_out) {
  for (uint32_t j = 0; j < n; ++j) {
    const uint8_t *row = codes + (uint64_t)ids[j] * m; /* 64-bit offset :1232 */
    float s[8];
    for (int l = 0; l < 8; ++l) {
      float acc = 0.0f;
      for (uint32_t c = (uint32_t)l; c < m; c += 8) acc = acc + lut[(size_t)c * 256 + row[c]];
      s[l] = acc;
    }
    const float a = (s[0] + s[1]) + (s[2] + s[3]);
    const float b = (s[4] + s[5]) + (s[6] + s[7]);
    dist_out[j] = a + b;
  }
}

/* ------------------------------------------------------------------ K3a */

/* compute_BestLSets_par_sort_msort :1533-1585 is a stable merge sort (left run placed by
 * lower_bound, right run by upper_bound :1559-1567): equal distances keep input order.
 * Insertion sort with strict '>' has the same result. */
void orc_sort_pairs(uint32_t *ids, float *dist, uint32_t n) {
  for (uint32_t i = 1; i < n; ++i) {
    const float d = dist[i];
    const uint32_t x = ids[i];
    uint32_t j = i;
    while (j > 0 && dist[j - 1] > d) {
      dist[j] = dist[j - 1];
      ids[j] = ids[j - 1];
      --j;
    }
    dist[j] = d;
    ids[j] = x;
  }
}

/* lower_bound_d :1718-1732: first index with arr[idx] >= target */
static uint32_t lower_bound_f(const float *arr, uint32_t lo, uint32_t hi, float target) {
  while (lo < hi) {
    const uint32_t mid = (lo + hi) / 2;
    if (target <= arr[mid]) hi = mid; else lo = mid + 1;
  }
  return lo;
}
/* upper_bound_d :1735-1749: first index with arr[idx] > target */
static uint32_t upper_bound_f(const float *arr, uint32_t lo, uint32_t hi, float target) {
  while (lo < hi) {
    const uint32_t mid = (lo + hi) / 2;
    if (target >= arr[mid]) lo = mid + 1; else hi = mid;
  }
  return lo;
}

/* ------------------------------------------------------------------ K3b */

/* compute_BestLSets_par_merge :1605-1715.  s_* = sorted new neighbours, w_* = worklist. */
uint32_t orc_merge(const uint32_t *s_ids, const float *s_dist, uint32_t s_n, uint32_t iter,
                   uint32_t *w_ids, float *w_dist, uint8_t *w_vis, uint32_t w_n, uint32_t L,
                   uint32_t medoid, uint32_t mark) {
  uint32_t new_n = w_n;
  if (s_n > 0) {                       /* :1636 */
    if (iter == 1) {                   /* :1638-1649 */
      const uint32_t nb = s_n < L ? s_n : L;
      for (uint32_t i = 0; i < nb; ++i) {
        w_ids[i] = s_ids[i];
        w_dist[i] = s_dist[i];
        w_vis[i] = (uint8_t)(s_ids[i] == medoid);
      }
      new_n = nb;
    } else {                           /* :1650-1708 */
      const float worst = w_dist[w_n - 1];
      const uint32_t lim = L < s_n ? L : s_n;
      uint32_t nb = 0;
      while (nb < lim && !(s_dist[nb] >= worst)) ++nb;          /* :1653-1657 */
      const uint32_t fill = (L - w_n) < s_n ? (L - w_n) : s_n;
      if (fill > nb) nb = fill;                                   /* :1660 */
      new_n = (w_n + nb) < L ? (w_n + nb) : L;                    /* :1662 */
      uint32_t t_ids[ORC_MAX_L];
      float t_dist[ORC_MAX_L];
      uint8_t t_vis[ORC_MAX_L];
      for (uint32_t i = 0; i < nb; ++i) {                         /* :1675-1677,1686-1690 */
        const uint32_t pos = lower_bound_f(w_dist, 0, w_n, s_dist[i]) + i;
        if (pos < new_n) { t_ids[pos] = s_ids[i]; t_dist[pos] = s_dist[i]; t_vis[pos] = 0; }
      }
      for (uint32_t k = 0; k < w_n; ++k) {                        /* :1678-1680,1691-1695 */
        const uint32_t pos = upper_bound_f(s_dist, 0, nb, w_dist[k]) + k;
        if (pos < new_n) { t_ids[pos] = w_ids[k]; t_dist[pos] = w_dist[k]; t_vis[pos] = w_vis[k]; }
      }
      memcpy(w_ids, t_ids, new_n * sizeof(uint32_t));
      memcpy(w_dist, t_dist, new_n * sizeof(float));
      memcpy(w_vis, t_vis, new_n * sizeof(uint8_t));
    }
  }
  for (uint32_t i = 0; i < new_n; ++i)                            /* :1711-1714 */
    if (w_ids[i] == mark) w_vis[i] = 1;
  return new_n;
}

/* ------------------------------------------------------------------ K4 */

/* closest new neighbour, strict '<', first minimum wins, MEDOID skipped :1413-1418,1491-1496 */
static int best_new(const uint32_t *s_ids, const float *s_dist, uint32_t s_n, uint32_t medoid,
                    float *dist_out) {
  float dist = ORC_BIG_DIST;
  int idx = -1;
  for (uint32_t i = 0; i < s_n; ++i) {
    if (s_dist[i] < dist && s_ids[i] != medoid) { idx = (int)i; dist = s_dist[i]; }
  }
  *dist_out = dist;
  return idx;
}

/* compute_parent1 :1464-1521: unconditional best new neighbour.
 * CANON: if no eligible neighbour exists the reference reads d_neighbors[0] of query 0
 * (:1485,1502); here the query simply has no parent. */
int orc_parent1(const uint32_t *s_ids, const float *s_dist, uint32_t s_n, uint32_t medoid,
                uint32_t *parent, uint32_t *mark) {
  float dist;
  const int idx = best_new(s_ids, s_dist, s_n, medoid, &dist);
  if (idx < 0) return 0;
  *parent = s_ids[idx];
  *mark = s_ids[idx];
  return 1;
}

/* compute_parent2 :1384-1459 */
int orc_parent2(const uint32_t *s_ids, const float *s_dist, uint32_t s_n, const uint32_t *w_ids,
                const float *w_dist, uint8_t *w_vis, uint32_t w_n, uint32_t medoid,
                uint32_t *parent, uint32_t *mark) {
  float dist;
  const int idx = best_new(s_ids, s_dist, s_n, medoid, &dist);
  int found = 0;
  for (uint32_t i = 0; i < w_n; ++i) {        /* :1425-1439 */
    if (!w_vis[i]) {
      found = 1;
      if (dist < w_dist[i]) { *parent = s_ids[idx]; *mark = s_ids[idx]; }
      else { *parent = w_ids[i]; w_vis[i] = 1; }
      break;
    }
  }
  if (!found && w_n > 0 && dist < w_dist[w_n - 1]) { /* corner case :1442-1446 */
    found = 1;
    *parent = s_ids[idx];
    *mark = s_ids[idx];
  }
  return found;
}

/* ------------------------------------------------------------------ K6 / K7 */

/* compute_L2Dist :1254-1299.  For u8/i8 the subtraction happens in int (:1294, C integer
 * promotion) then converts to float; sum in ascending j from 0.0f.  CANON: fmaf. */
float orc_exact_dist(const void *vec, const void *query, uint32_t D, int dtype, int dim_adjust) {
  const uint32_t qdim = D - (uint32_t)dim_adjust;
  float acc = 0.0f;
  for (uint32_t j = 0; j < D; ++j) {
    float diff;
    if (dtype == ORC_U8) {
      const int q = (j < qdim) ? (int)((const uint8_t *)query)[j] : 0;
      diff = (float)((int)((const uint8_t *)vec)[j] - q);
    } else if (dtype == ORC_I8) {
      const int q = (j < qdim) ? (int)((const int8_t *)query)[j] : 0;
      diff = (float)((int)((const int8_t *)vec)[j] - q);
    } else {
      const float q = (j < qdim) ? ((const float *)query)[j] : 0.0f;
      diff = ((const float *)vec)[j] - q;
    }
    acc = fmaf(diff, diff, acc);
  }
  return acc;
}

/* compute_NearestNeighbours :1312-1368: stable merge sort by exact distance (ties keep
 * expansion order), first k ids as u64 (:1364-1367).  CANON: when fewer than k candidates
 * exist the tail is UINT64_MAX / ORC_BIG_DIST (uninitialised memory in the reference). */
void orc_topk(const uint32_t *cand_ids, const float *cand_dist, uint32_t n, uint32_t k,
              uint64_t *ids_out, float *dist_out) {
  uint32_t *ids = (uint32_t *)malloc(sizeof(uint32_t) * (n ? n : 1));
  float *d = (float *)malloc(sizeof(float) * (n ? n : 1));
  memcpy(ids, cand_ids, sizeof(uint32_t) * n);
  memcpy(d, cand_dist, sizeof(float) * n);
  orc_sort_pairs(ids, d, n);
  for (uint32_t r = 0; r < k; ++r) {
    if (r < n) { ids_out[r] = (uint64_t)ids[r]; dist_out[r] = d[r]; }
    else { ids_out[r] = UINT64_MAX; dist_out[r] = ORC_BIG_DIST; }
  }
  free(ids);
  free(d);
}

/* ------------------------------------------------------------------ whole search */

static inline const uint8_t *entry_ptr(const orc_index *ix, uint64_t id) {
  return ix->graph + id * ix->entry_len;
}
static inline size_t elem_size(int dtype) { return dtype == ORC_F32 ? 4 : 1; }

/* adjacency of node id: [u32 deg][u32 nbr...] at +D*sizeof(T)  (:801-810, :467-477) */
static uint32_t fetch_adj(const orc_index *ix, uint64_t id, uint32_t *out) {
  const uint8_t *e = entry_ptr(ix, id) + (size_t)ix->D * elem_size(ix->dtype);
  uint32_t deg;
  memcpy(&deg, e, 4);
  if (deg > ix->R) deg = ix->R;
  memcpy(out, e + 4, (size_t)deg * 4);
  return deg;
}

typedef struct {
  uint8_t *bloom;
  float *lut;
  uint32_t *cand;
  float *cand_dist;
} orc_scratch;

/* One query, start to finish.  Follows bang_init :427-507 (seeding), bang_query :569-1068
 * (loop order) with the per-query view of the lock-step batch loop.
 *
 * CANON differences from the reference, all documented in DESIGN.md:
 *  - a query stays active while it has a parent OR still holds unmerged survivors (the
 *    reference's single batch-wide nextIter flag :958,1453 makes that case depend on what
 *    the other queries of the batch do);
 *  - the candidate log is compact (append), the reference indexes it by iteration (:1457)
 *    but reads rows [0,count) (:1291), which only agrees when there are no idle iterations;
 *  - the vector of the parent chosen at the iteration cap (:950-956) is part of the re-rank
 *    (the reference never fetches that row). */
static void search_one(const orc_index *ix, const void *query, uint32_t k, uint32_t L, int dim_adjust,
                       orc_scratch *sc, uint64_t *ids_out, float *dists_out, uint32_t Q, uint32_t q,
                       orc_qstats *st) {
  const uint32_t R = ix->R;
  const uint32_t medoid = (uint32_t)ix->medoid;
  const uint32_t max_cand = L + ORC_EXTRA_ITERS;          /* uMAX_PARENTS_PERQUERY :603 */
  uint32_t T[65 + 8], S[65 + 8];
  float d[65 + 8];
  uint32_t w_ids[ORC_MAX_L];
  float w_dist[ORC_MAX_L];
  uint8_t w_vis[ORC_MAX_L];
  uint32_t w_n = 0;
  uint32_t n_cand = 0;
  uint64_t evals = 0, fetched = 0;
  (void)R;

  memset(sc->bloom, 0, ORC_BF_MEMORY);                     /* :443 */
  orc_lut_build(ix, query, dim_adjust, sc->lut);           /* K1 :623 */

  sc->cand[n_cand++] = medoid;                             /* :455-462 */
  uint32_t t_n = 0;
  T[t_n++] = medoid;                                       /* :482 */
  t_n += fetch_adj(ix, medoid, T + 1);                     /* :484-487 */

  uint32_t iter = 1;                                       /* :596 */
  fetched += t_n;
  uint32_t s_n = orc_filter(sc->bloom, T, t_n, S);         /* K5 :650 */
  orc_pqdist(sc->lut, ix->codes, ix->m, S, s_n, d);        /* K2 :663 */
  evals += s_n;
  uint32_t parent = 0, mark = 0x01010101u;                 /* memset(d_mark, 1) :446 */
  int has_parent = orc_parent1(S, d, s_n, medoid, &parent, &mark); /* K4a :678 */
  if (has_parent) sc->cand[n_cand++] = parent;

  while (has_parent || s_n > 0) {
    orc_sort_pairs(S, d, s_n);                                               /* K3a :726 */
    w_n = orc_merge(S, d, s_n, iter, w_ids, w_dist, w_vis, w_n, L, medoid, mark); /* K3b :738 */
    t_n = has_parent ? fetch_adj(ix, parent, T) : 0;                         /* walker :771-813 */
    fetched += t_n;
    s_n = orc_filter(sc->bloom, T, t_n, S);                                  /* K5 :855 */
    orc_pqdist(sc->lut, ix->codes, ix->m, S, s_n, d);                        /* K2 :871 */
    evals += s_n;
    ++iter;                                                                  /* :879 */
    has_parent = orc_parent2(S, d, s_n, w_ids, w_dist, w_vis, w_n, medoid, &parent, &mark); /* :917 */
    if (has_parent) sc->cand[n_cand++] = parent;
    if (iter == max_cand - 1) break;                                         /* :950-956 */
  }

  /* re-rank: K6 :969 + K7 :980 */
  for (uint32_t i = 0; i < n_cand; ++i)
    sc->cand_dist[i] = orc_exact_dist(entry_ptr(ix, sc->cand[i]), query, ix->D, ix->dtype, dim_adjust);
  uint64_t ids_k[ORC_MAX_L];
  float dist_k[ORC_MAX_L];
  orc_topk(sc->cand, sc->cand_dist, n_cand, k, ids_k, dist_k);
  for (uint32_t r = 0; r < k; ++r) {
    ids_out[(size_t)q * k + r] = ids_k[r];                 /* [Q][k] :1366 */
    dists_out[(size_t)r * Q + q] = dist_k[r];              /* [rank][Q] :999,1297 */
  }
  if (st) {
    st->iterations = iter;
    st->candidates = n_cand;
    st->dist_evals = evals;
    st->fetched = fetched;
  }
}

int orc_num_threads(void) {
#ifdef _OPENMP
  return omp_get_max_threads();
#else
  return 1;
#endif
}

int orc_search_batch(const orc_index *ix, const void *queries, uint32_t Q, uint32_t k, uint32_t L,
                     int distfn, uint64_t *ids_out, float *dists_out, orc_qstats *stats,
                     int nthreads) {
  if (!ix || !queries || !ids_out || !dists_out) return -1;
  if (L > ORC_MAX_L || k > L || k == 0) return -2;
  if (ix->R > 64) return -3;                               /* assert(R == MAX_R) :190 */
  const int dim_adjust = (distfn == ORC_DIST_MIPS) ? 1 : 0; /* :631 */
  const size_t qstride = (size_t)(ix->D - (uint32_t)dim_adjust) * elem_size(ix->dtype);
  const uint32_t max_cand = L + ORC_EXTRA_ITERS;
  int err = 0;
#ifdef _OPENMP
  if (nthreads > 0) omp_set_num_threads(nthreads);
#else
  (void)nthreads;
#endif
#pragma omp parallel
  {
    orc_scratch sc;
    sc.bloom = (uint8_t *)malloc(ORC_BF_MEMORY);
    sc.lut = (float *)malloc(sizeof(float) * 256 * (size_t)ix->m);
    sc.cand = (uint32_t *)malloc(sizeof(uint32_t) * (max_cand + 2));
    sc.cand_dist = (float *)malloc(sizeof(float) * (max_cand + 2));
    if (!sc.bloom || !sc.lut || !sc.cand || !sc.cand_dist) {
#pragma omp atomic write
      err = -4;
    } else {
#pragma omp for schedule(dynamic, 4)
      for (int64_t q = 0; q < (int64_t)Q; ++q) {
        search_one(ix, (const uint8_t *)queries + (size_t)q * qstride, k, L, dim_adjust, &sc,
                   ids_out, dists_out, Q, (uint32_t)q, stats ? &stats[q] : NULL);
      }
    }
    free(sc.bloom); free(sc.lut); free(sc.cand); free(sc.cand_dist);
  }
  return err;
}

/* ------------------------------------------------------------------ recall */

/* calculate_recall, test_driver.cpp:43-93 (sets replaced by linear scans). */
double orc_recall(uint32_t num_queries, const uint32_t *gold_std, const float *gs_dist,
                  uint32_t dim_gs, const uint64_t *our_results, uint32_t dim_or,
                  uint32_t recall_at) {
  double total = 0;
  for (uint32_t i = 0; i < num_queries; ++i) {
    const uint32_t *gt = gold_std + (size_t)dim_gs * i;
    const uint64_t *res = our_results + (size_t)dim_or * i;
    uint32_t tie = recall_at;
    if (gs_dist) {
      const float *gd = gs_dist + (size_t)dim_gs * i;
      tie = recall_at - 1;
      while (tie < dim_gs && gd[tie] == gd[recall_at - 1]) ++tie;
    }
    uint32_t cur = 0;
    for (uint32_t a = 0; a < tie; ++a) {
      int dup = 0;                       /* std::set semantics: count distinct gt ids */
      for (uint32_t b = 0; b < a; ++b) if (gt[b] == gt[a]) { dup = 1; break; }
      if (dup) continue;
      for (uint32_t r = 0; r < recall_at; ++r)
        if (res[r] == (uint64_t)gt[a]) { ++cur; break; }
    }
    total += cur;
  }
  return total / (double)num_queries * (100.0 / (double)recall_at);
}
