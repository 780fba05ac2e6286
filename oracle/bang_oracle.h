/*
 * bang_oracle.h -- CPU restatement (plain C11) of the BANG_Base search hot path.
 *
 * TEST INFRASTRUCTURE ONLY.  Nothing under oracle/ is linked into, loaded by or
 * called from the product (libbang.so / bang_search).  Only tests/,
 * __graft_entry__.smoke() and bench.py's cpu_baseline leg may use it, and only
 * as the checker / the reported CPU baseline.
 *
 * PARITY UNPINNED BY THE REFERENCE: the reference (karthik86248/BANG-Billion-Scale-ANN)
 * is CUDA-only, cannot be compiled or run in this image (no nvcc / CUDA headers,
 * SURVEY.md 8(c)) and ships no tests, golden vectors or datasets.  This oracle is
 * therefore pinned only by (1) known answers derived by hand from the reference
 * source text (hash values, bounds, toy LUTs -- tests/test_oracle_kat.py) and
 * (2) the line-by-line citations below.  Every function names the reference
 * file:line it restates (paths relative to /root/reference/BANG_Base/).
 *
 * Canonical (deterministic) semantics where the reference is racy / undefined are
 * listed in DESIGN.md "Canonical semantics"; each is flagged CANON below.
 */
#ifndef BANG_ORACLE_H_
#define BANG_ORACLE_H_

#include <stdint.h>
#include <stddef.h>

#ifdef __cplusplus
extern "C" {
#endif

/* bang_search.cu:48-50 */
#define ORC_BF_ENTRIES 399887u
#define ORC_BF_MEMORY ((ORC_BF_ENTRIES & 0xFFFFFFFCu) + 4u)
/* bang_search.cu:53 ; bang.h:20 */
#define ORC_EXTRA_ITERS 50
#define ORC_MAX_L 512
/* compute_parent1/2 initial "infinite" distance literal, bang_search.cu:1406,1484 */
#define ORC_BIG_DIST ((float)3.402823E+38)

enum { ORC_U8 = 0, ORC_I8 = 1, ORC_F32 = 2 };
enum { ORC_DIST_L2 = 0, ORC_DIST_MIPS = 1 }; /* bang.h:26-30 */

/* In-memory view of a loaded index (bang_search.cu:138-362 loads exactly these). */
typedef struct {
  uint64_t medoid;      /* GraphMedataData.ullMedoid, bang_search.cuh:42-50 */
  uint64_t entry_len;   /* bytes per graph entry = D*sizeof(T) + 4 + 4*R   */
  uint32_t D, R, N, m;  /* dims (index side), degree bound, points, PQ chunks */
  int32_t dtype;        /* ORC_U8 / ORC_I8 / ORC_F32 */
  const uint8_t *graph; /* N entries: [T vec[D]][u32 deg][u32 nbr[R]], bang_search.cu:335-340 */
  const uint8_t *codes; /* [N][m] u8, bang_search.cu:218-243 */
  const float *pivots_T;/* [D][256] f32 (transposed at load, bang_search.cu:281-285) */
  const float *centroid;/* [D] */
  const uint32_t *chunk_off; /* [m+1] */
} orc_index;

/* per-query statistics produced by the whole-search oracle */
typedef struct {
  uint32_t iterations;   /* value of `iter` when the query stopped */
  uint32_t candidates;   /* expanded nodes logged for the re-rank (incl. medoid) */
  uint64_t dist_evals;   /* surviving neighbours whose PQ distance was computed */
  uint64_t fetched;      /* neighbour ids offered to the filter */
} orc_qstats;

/* ---- stage functions (one per reference kernel) ---- */
uint32_t orc_hash1(uint32_t x);  /* bang_search.cu:1168-1178 */
uint32_t orc_hash2(uint32_t x);  /* bang_search.cu:1180-1189 */

/* K1 populate_pqDist_par, bang_search.cu:1083-1130.  lut_out[m*256]. */
void orc_lut_build(const orc_index *ix, const void *query, int dim_adjust, float *lut_out);

/* K5 neighbor_filtering_new, bang_search.cu:1140-1165.  bloom = ORC_BF_MEMORY bytes. */
uint32_t orc_filter(uint8_t *bloom, const uint32_t *in, uint32_t n_in, uint32_t *out);

/* K2 compute_neighborDist_par, bang_search.cu:1201-1241 (canonical float order). */
void orc_pqdist(const float *lut, const uint8_t *codes, uint32_t m, const uint32_t *ids,
                uint32_t n, float *dist_out);

/* K3a compute_BestLSets_par_sort_msort, bang_search.cu:1533-1585 (stable sort asc). */
void orc_sort_pairs(uint32_t *ids, float *dist, uint32_t n);

/* K3b compute_BestLSets_par_merge, bang_search.cu:1605-1715. Returns new worklist size. */
uint32_t orc_merge(const uint32_t *s_ids, const float *s_dist, uint32_t s_n, uint32_t iter,
                   uint32_t *w_ids, float *w_dist, uint8_t *w_vis, uint32_t w_n, uint32_t L,
                   uint32_t medoid, uint32_t mark);

/* K4a/K4b compute_parent1 / compute_parent2, bang_search.cu:1464-1521 / 1384-1459.
 * Return 1 and *parent if a parent was found; update *mark / visited as the reference. */
int orc_parent1(const uint32_t *s_ids, const float *s_dist, uint32_t s_n, uint32_t medoid,
                uint32_t *parent, uint32_t *mark);
int orc_parent2(const uint32_t *s_ids, const float *s_dist, uint32_t s_n, const uint32_t *w_ids,
                const float *w_dist, uint8_t *w_vis, uint32_t w_n, uint32_t medoid,
                uint32_t *parent, uint32_t *mark);

/* K6 compute_L2Dist, bang_search.cu:1254-1299: exact squared L2 of one vector. */
float orc_exact_dist(const void *vec, const void *query, uint32_t D, int dtype, int dim_adjust);

/* K7 compute_NearestNeighbours, bang_search.cu:1312-1368: stable sort by exact dist,
 * first k.  ids_out[k] u64, dist_out[k] (sorted exact distances). */
void orc_topk(const uint32_t *cand_ids, const float *cand_dist, uint32_t n, uint32_t k,
              uint64_t *ids_out, float *dist_out);

/* ---- whole search (bang_init + bang_query, bang_search.cu:427-507, 569-1068) ----
 * queries: [Q][D - dim_adjust] of the index dtype.  ids_out [Q][k] u64;
 * dists_out [k][Q] f32 (rank-major, the layout bang_search.cu:999 returns).
 * stats may be NULL.  nthreads <= 0 -> OpenMP default.  Returns 0 or a negative error. */
int orc_search_batch(const orc_index *ix, const void *queries, uint32_t Q, uint32_t k, uint32_t L,
                     int distfn, uint64_t *ids_out, float *dists_out, orc_qstats *stats,
                     int nthreads);

/* test_driver.cpp:43-93 calculate_recall (tie-aware). gs_dist may be NULL. */
double orc_recall(uint32_t num_queries, const uint32_t *gold_std, const float *gs_dist,
                  uint32_t dim_gs, const uint64_t *our_results, uint32_t dim_or,
                  uint32_t recall_at);

int orc_num_threads(void);

#ifdef __cplusplus
}
#endif
#endif
