/*
 * bang_c.h -- C-ABI of the MI355X-native BANG_Base search engine (libbang.so).
 *
 * Plain pointers and sizes only; every function returns an int status (0 = BANG_OK).
 * Two layers:
 *
 *  (1) ENGINE level -- the live version of the C mirror the reference sketches but compiles out
 *      (BANG_Base/bang.h:89-101, bang_search.cu:1787-1807; uint8-only there).  Same call order
 *      as BANGSearch<T> (bang.h:36-84).  This is what a ctypes / cgo / JNI binding uses.
 *
 *  (2) KERNEL level -- the seam between the C++ host graph walker and the HIP kernels: one
 *      entry per reference kernel (or fused group of kernels), device pointers + an explicit
 *      hipStream_t passed as void*, no hidden globals, no allocation inside.  Each entry cites
 *      the reference kernel it replaces.  tests/ drive these one by one against the oracle.
 *
 * All "d_" pointers are device pointers; "h_" host pointers.
 */
#ifndef BANG_C_H_
#define BANG_C_H_

#include <stddef.h>
#include <stdint.h>

#ifdef __cplusplus
extern "C" {
#endif

/* ------------------------------------------------------------------ constants */
#define BANG_OK 0
#define BANG_ERR_ARG (-1)       /* bad argument / call order */
#define BANG_ERR_IO (-2)        /* missing / malformed index file (bang_load -> false) */
#define BANG_ERR_NOMEM (-3)
#define BANG_ERR_HIP (-4)       /* a HIP runtime call failed; see bang_last_error() */
#define BANG_ERR_UNSUPPORTED (-5)
#define BANG_ERR_NOGPU (-6)     /* no HIP device: the product has NO CPU fallback */

#define BANG_MAX_L 512          /* bang.h:20 */
#define BANG_MAX_R 64           /* bang_search.cu:35 */
#define BANG_EXTRA_ITERS 50     /* NAX_EXTRA_ITERATION, bang_search.cu:53 */
#define BANG_BF_ENTRIES 399887u /* bang_search.cu:48 */
#define BANG_BF_WORDS 12512u    /* bit-packed filter, u32 words per query (>= ceil(399887/32), 64-B multiple) */
#define BANG_NO_PARENT 0xFFFFFFFFu   /* parents[q]: query is finished */
#define BANG_IDLE_PARENT 0xFFFFFFFEu /* parents[q]: no parent this step, unmerged survivors pending */
#define BANG_NBR_STRIDE 72u     /* u32 stride of the per-query neighbour / distance rows (>= R+1) */
#define BANG_STAGE_STRIDE 65u   /* u32 stride of a staged adjacency row: [count][id x R] */

enum { BANG_U8 = 0, BANG_I8 = 1, BANG_F32 = 2 };            /* element type of vectors/queries */
enum { BANG_DIST_L2 = 0, BANG_DIST_MIPS = 1 };              /* DistFunc, bang.h:26-30 */
enum { BANG_GRAPH_HOST = 0,      /* graph in host RAM, C++ walker + staged H2D (BANG_Base) */
       BANG_GRAPH_DEVICE = 1,    /* graph + vectors resident in HBM (BANG_Inmemory placement) */
       BANG_GRAPH_AUTO = 2 };    /* resolved by bang_load: DEVICE when graph + vectors fit the free HBM next to the PQ codes
                                    (with 16 GB to spare for the per-batch state), else HOST.  288 GB hold SIFT1M .. DEEP100M
                                    (configs[1], [2]) whole; SIFT1B's 388 GB graph stays in host RAM (configs[3]) */

/* ------------------------------------------------------------------ (1) engine level */
typedef struct bang_engine bang_engine_t;

const char* bang_last_error(void);          /* thread-local message of the last failure */
int bang_device_count(void);                /* 0 if no HIP device */

int bang_create(int dtype, bang_engine_t** out);             /* BANGSearch<T>() bang.h:42 */
int bang_destroy(bang_engine_t* e);                          /* ~BANGSearch()   bang.h:43 */

/* options must be set before bang_load_e / bang_alloc_e:
 *   "graph"   : BANG_GRAPH_HOST | BANG_GRAPH_DEVICE | BANG_GRAPH_AUTO (default: auto; environment BANG_GRAPH=host|device|auto)
 *   "lanes"   : number of independent query groups pipelined against each other (>=1)
 *   "threads" : host walker threads per lane (>=1)
 *   "device"  : HIP device ordinal
 *   "pq"      : 0 = pivot-stationary fused distance (default when it fits LDS), 1 = LUT path (K1+K2)
 *   "timing"  : 1 = stamp every front-kernel launch in-kernel (s_memrealtime) for bang_get_stats
 *   "numa"    : graph in host RAM: 1 = pin the walker threads (and the caller during bang_query) to the CPUs of the GPU's NUMA node,
 *               one physical core each; 0 = leave them where the OS puts them; -1 = auto = 0 (pinning measured slower on the
 *               2-socket measurement box)
 *   "search"  : graph in HBM: 1 = the query-resident search kernel (bang_k_search), 0 = the round-1 loops, -1 = auto */
int bang_set_option(bang_engine_t* e, const char* key, long value);
/* The whole table of options and environment switches as text (csrc/bang_options.cpp is the one place they are defined): writes at
 * most cap bytes (NUL-terminated) to buf, returns the size needed.  buf may be NULL. */
int bang_describe_options(char* buf, size_t cap);

/* bang_load, bang.h:51 / bang_search.cu:138-362 */
int bang_load_e(bang_engine_t* e, const char* indexfile_path_prefix);

/* Same as bang_load_e but from memory (synthetic / already-mapped indices).  The graph pointer
 * must stay valid until bang_unload_e (it is NOT copied in BANG_GRAPH_HOST mode); everything
 * else is copied.  pivots is the file-order [256][D] table. */
typedef struct {
  uint64_t medoid, entry_len;
  uint32_t D, R, N, m;
  const uint8_t* graph;      /* N * entry_len bytes */
  const uint8_t* codes;      /* N * m bytes (host) -- or NULL when d_codes is given */
  const void* d_codes;       /* optional: codes already on the device (N*m + 256 bytes; N*code_stride + 256 with code_stride != 0) */
  const float* pivots;       /* [256][D] */
  const float* centroid;     /* [D] */
  const uint32_t* chunk_off; /* [m+1] */
  uint32_t code_stride;      /* d_codes only: bytes between the rows of consecutive nodes, 0 = m (packed).  Host codes are always packed;
                                the engine lays them out itself (option "code_stride") */
  uint32_t vectors_ready;    /* bang_load_shared_e: 1 = d_vectors already holds the vectors */
  void* d_vectors;           /* optional, streamed / shared loads: a DEVICE buffer of the caller, N * D * sizeof(T) + 256 bytes, that holds the
                                full-precision vectors instead of an allocation of the engine (it must outlive the index).
                                bang_load_stream_e fills it; the caller may then hand it on to the other GPUs of the node */
  uint64_t rows_hash;        /* bang_load_shared_e: what bang_get_rows_hash reported on the rank that loaded the index */
} bang_index_desc;
int bang_load_mem_e(bang_engine_t* e, const bang_index_desc* desc);

/* STREAMED load (host placement, pull mode only): the graph entries are handed over chunk by chunk and are never held in host
 * memory as a whole.  The engine calls src(ctx, first, count, dst) for consecutive node ranges; the source writes `count`
 * entries in the reference layout ([T vec[D]][u32 degree][u32 id x R], entry_len bytes each) to dst and returns 0.  Of each entry
 * the vector goes to HBM and the adjacency list into the 256-byte pull rows in pinned host memory: a 10^9-node SIFT index then
 * needs 256 GB of host RAM instead of 388 GB + 256 GB.  desc->graph must be NULL.  Fails (BANG_ERR_UNSUPPORTED / BANG_ERR_NOMEM)
 * where the pull mode is not possible -- vectors that do not fit HBM, rows that do not fit the host, R > 64, option pull = 0 --
 * because nothing else can run without the graph.  bang_load_e streams `<p>_disk.bin` through the same path when the pull mode
 * applies (BANG_STREAM_LOAD=0: map the file as before); should a walker form be asked for later, the file is mapped then. */
/* The reference's preprocessing step as a library call (host only, no device needed): DiskANN's sector-padded `_disk.index` ->
 * `<out_prefix>_disk.bin` + `<out_prefix>_disk_metadata.bin`, byte for byte what BANG_Base/bang_preprocess.py writes (header parse
 * :28-64, sector walk :75-80, ascending adjacency lists :81-109, metadata :42-51,116).  dtype: BANG_U8 / BANG_I8 / BANG_F32.
 * bang_load_e does the same conversion on the fly when only the `_disk.index` exists. */
int bang_convert_diskann_index(const char* index_path, const char* out_prefix, int dtype);

typedef int (*bang_entry_source)(void* ctx, uint64_t first, uint64_t count, uint8_t* dst);
int bang_load_stream_e(bang_engine_t* e, const bang_index_desc* desc, bang_entry_source src, void* ctx);
/* SHARED load, for the ranks of a multi-GPU node that did NOT read the index: the rank that did (bang_load_e / bang_load_stream_e with
 * desc->d_vectors) has built the node's ONE pull-rows file (BANG_PULL_ROWS_DIR) and holds the full-precision vectors in its HBM; the
 * others receive the vectors from there (RCCL broadcast over xGMI into their own desc->d_vectors, vectors_ready = 1) and the hash of
 * the adjacency lists (bang_get_rows_hash), map the rows file and are done -- no index entry is read a second time.  PQ codes as in
 * every load (desc->codes / d_codes). */
int bang_load_shared_e(bang_engine_t* e, const bang_index_desc* desc);
int bang_get_rows_hash(bang_engine_t* e, uint64_t* out);

/* PEER ROWS: the adjacency rows of a multi-GPU node in the NODE's spare HBM, read over xGMI, host DRAM only for the remainder.
 * After bang_load* in pull mode every GPU has HBM left (SIFT1B: ~46 GB = 178 M rows of 256 B): rank r of W keeps the rows
 * [r * n, (r + 1) * n) there (bang_rows_slice_e), exports that allocation (bang_rows_export_e: a 64-byte hipIpcMemHandle), and every rank
 * imports the W - 1 others (bang_rows_import_e).  The search kernel then reads the row of parent p from slice p / n -- its own HBM, a
 * peer's HBM through the IPC mapping, or, for p >= W * n, pinned host memory over PCIe as before.  W x 18 % of SIFT1B's rows: 72 % at
 * W = 4, all of them at W = 8.  Call order: load -> slice -> export / (exchange handles) / import -> bang_alloc.  Results never change.
 * bang_rows_slice_e replaces the default copy (rows [0, auto)); rows = 0 drops it.  A slice that has been exported is never freed before
 * bang_unload (bang_alloc does not take it back when HBM is short: BANG_ERR_NOMEM instead). */
#define BANG_MAX_ROW_SLICES 16
int bang_get_num_nodes(bang_engine_t* e, uint64_t* nodes_out);    /* N of the loaded index (the reference reads it from <p>_disk_metadata.bin, bang_search.cu:176-189) */
int bang_rows_capacity_e(bang_engine_t* e, uint64_t* rows_out);   /* rows of 256 B this engine's free HBM holds now (6 GB kept back for the batch state) + what its row copy holds */
int bang_rows_slice_e(bang_engine_t* e, uint64_t first_row, uint64_t rows);
int bang_rows_export_e(bang_engine_t* e, void* handle64, uint64_t* first_row, uint64_t* rows);
int bang_rows_import_e(bang_engine_t* e, uint32_t slot, uint32_t n_slots, uint64_t slice_rows, uint64_t rows /* what the exporter's bang_rows_export_e
                       * reported: a peer allocation shorter than the slot's rows min(slice_rows, N - slot * slice_rows) is refused -- the kernel reads
                       * base + p * 256 for every p of the slot; ignored for the own slice */,
                       const void* handle64 /* NULL: this engine's own slice */);
/* Tear-down in two phases: every rank closes its mappings of the OTHER ranks' slices (this call; after bang_free), the ranks meet (a barrier of
 * the caller's process group), and only then does anyone bang_unload -- which frees the slice the others had mapped. */
int bang_rows_close_peers_e(bang_engine_t* e);

int bang_set_searchparams_e(bang_engine_t* e, int recall, int worklist_length, int distfn); /* bang.h:60 */
int bang_alloc_e(bang_engine_t* e, int num_queries);                                        /* bang.h:53 */
int bang_init_e(bang_engine_t* e, int num_queries);                                         /* bang.h:56 */
/* bang_query, bang.h:75: ids [Q][k] u64, dists [k][Q] f32 (rank-major, bang_search.cu:999) */
int bang_query_e(bang_engine_t* e, const void* h_queries, int num_queries, uint64_t* h_ids, float* h_dists);
/* The same search with the results LEFT ON THE DEVICE: d_ids [Q][k] u64 and (optional, may be NULL) d_dists [k][Q] f32 are device
 * buffers of the caller on the engine's device; nothing but the per-query iteration counts returns to the host.  For callers that
 * hand the ids on from device memory -- the multi-GPU job all-gathers the shards' id blocks over xGMI straight from here (SURVEY
 * 8(e)) instead of bouncing them through the host.  Complete (stream synchronised) on return. */
int bang_query_dev_e(bang_engine_t* e, const void* h_queries, int num_queries, uint64_t* d_ids, float* d_dists);
int bang_free_e(bang_engine_t* e);                                                          /* bang.h:80 */
int bang_unload_e(bang_engine_t* e);                                                        /* bang.h:82 */

/* statistics of the last bang_query_e */
typedef struct {
  double wall_ms;             /* bang_query_e wall time */
  uint64_t iterations;        /* max `iter` reached over the lanes */
  uint64_t dist_evals;        /* total surviving neighbours whose PQ distance was computed */
  uint64_t fetched;           /* total adjacency ids offered to the filter */
  uint64_t candidates;        /* total expanded nodes re-ranked */
  uint64_t front_launches;    /* launches of the filter+distance+parent kernel */
  double front_ms;            /* sum of their durations from in-kernel s_memrealtime stamps ("timing"=1, else 0):
                                 per launch max(end) - min(start) over its workgroups */
  double back_ms;             /* unused (kept for layout; rocprofv3 reports the sort+merge kernel) */
  double rerank_ms;
  double walker_ms;           /* host time in the adjacency gather, summed over lanes */
  double sync_ms;             /* host time blocked waiting for the parents of an iteration, summed over lanes */
  double enqueue_ms;          /* host time spent inside launch / memcpy-enqueue calls, summed over lanes */
  double front_busy_ms;       /* length of the UNION of the front-kernel launch intervals of all lanes ("timing"=1): the time
                                 during which at least one front kernel was running (lanes overlap) */
  uint64_t persistent;        /* 1: the batch ran as ONE persistent search kernel (host-graph mode, "persistent" option).  Then
                                 front_launches = 1, front_busy_ms = duration of that launch (first to last in-kernel stamp) and
                                 front_ms = time a workgroup spent in its front phases, mean over the workgroups */
  uint64_t h2d_bytes;         /* host-graph mode: bytes the walker handed to the device (adjacency rows + full-precision vectors) */
  uint64_t vectors_on_device; /* host-graph mode: 1 = the re-rank read a packed copy of the vectors in HBM ("vectors" option),
                                 0 = the walker shipped every expanded node's vector (the reference's data flow) */
  /* effective configuration of the allocation the query ran on (what "auto" resolved to) */
  uint64_t graph_mode;        /* BANG_GRAPH_HOST | BANG_GRAPH_DEVICE */
  uint64_t lanes;             /* independent query groups (1 with the persistent search kernel) */
  uint64_t walker_threads;    /* host walker threads per lane (0 in device-graph mode) */
  uint64_t wg_queries;        /* persistent search kernel: queries per workgroup block, 0 otherwise */
  uint64_t workgroups;        /* persistent search kernel: workgroups of the launch, 0 otherwise */
  /* expansions (= graph hops = iterations in which the query had a parent) per query: median, 99th percentile, maximum */
  uint64_t hops_p50, hops_p99, hops_max;
  uint64_t search_kernel;     /* 1: the batch ran on the query-resident search kernel (bang_k_search) */
  uint64_t pacing_groups;     /* host-paced search kernel: groups of waves the walker threads serve (0 otherwise) */
  uint64_t graph_pull;        /* host-graph placement: 1 = PULL mode -- the adjacency lists live as 256-byte rows in pinned host memory
                                 and the self-paced search kernel fetches them over PCIe by itself (no walker thread in the loop) */
  uint64_t pulled_bytes;      /* pull mode: bytes of adjacency rows the kernel fetched over PCIe (256 per expansion) */
  uint64_t rows_in_hbm;       /* pull mode: nodes whose adjacency row ALSO sits in HBM (no PCIe read for them), option "rows_hbm" */
  uint64_t code_stride;       /* bytes between PQ code rows in HBM (m = packed; 128 = rows padded to their own 128-byte line) */
  uint64_t filter_loads_skipped; /* search kernel, self-paced form: visited-filter word loads NOT issued because the wave's on-chip
                                 summary knew the word was still zero (of 2 x `fetched` probes) */
  uint64_t rows_from_peer;    /* pull mode with peer rows: expansions of the last batch whose adjacency row came from ANOTHER GPU's HBM (over xGMI) */
  uint64_t rows_from_own_hbm; /* ... from this GPU's own HBM copy / slice */
  uint64_t walker_rows;       /* host-paced search kernel: 1 = the walker threads read the 256-byte pull rows (option "walker"), not graph entries */
  uint64_t rerank_fused;      /* 1: K6 + K7 ran inside the search launch (the wave that finished a query re-ranked it), no re-rank launch followed */
} bang_stats;
int bang_get_stats(bang_engine_t* e, bang_stats* out);
/* Per-query counters of the last bang_query_e (arrays of num_queries words; any pointer may be NULL): PQ distance evaluations,
 * adjacency ids offered to the filter, expanded nodes (candidate-log length) and -- search kernel only, else zeros -- the number
 * of iterations the query ran.  Test hook: the oracle reports the same four numbers per query. */
int bang_get_query_counters(bang_engine_t* e, uint32_t* dist_evals, uint32_t* fetched, uint32_t* candidates, uint32_t* iterations);

/* The candidate log of the last bang_query_e (bang_search.cu:1451-1458: the nodes a query expanded, in expansion order, [0] = MEDOID): ids
 * [num_queries][stride] with stride >= L + 50, counts [num_queries]; num_queries = rows the caller's buffers hold, refused when smaller than
 * the last batch (the rows of that batch are what is copied).  Analysis hook (which adjacency rows a batch really reads). */
int bang_get_candidate_log(bang_engine_t* e, uint32_t* ids, uint32_t stride, uint32_t* counts, uint32_t num_queries);

/* The reference's own C mirror (bang.h:91-100), uint8 only, one process-global engine. */
int bang_load_c(char* indexfile_path_prefix);
int bang_set_searchparams_c(int recall, int worklist_length, int nDistFunc);
int bang_alloc_c(int num_queries);
int bang_init_c(int num_queries);
int bang_query_c(uint8_t* query_array, int num_queries, unsigned long* nearestNeighbours,
                 float* nearestNeighbours_dist);
int bang_free_c(void);
int bang_unload_c(void);

/* ------------------------------------------------------------------ device helpers */
/* Raw device memory for callers without a tensor library (tests, bindings). */
int bang_dev_malloc(void** d_ptr, size_t bytes);
int bang_dev_free(void* d_ptr);
int bang_dev_memset(void* d_ptr, int value, size_t bytes);
int bang_dev_h2d(void* d_dst, const void* h_src, size_t bytes);
int bang_dev_d2h(void* h_dst, const void* d_src, size_t bytes);
int bang_dev_sync(void);

/* ------------------------------------------------------------------ (2) kernel level */

/* Layout of the LDS-resident ("pivot-stationary") distance kernel for an index: every chunk is padded
 * to psz floats (1,2,4,8 >= the largest chunk) and the chunk count to mp (a multiple of 4 with a
 * compiled kernel instance).  psz == 0 on return means "use the LUT path" (chunks wider than 8 dims or a
 * table that cannot fit the 160 KB LDS). */
int bang_pq_layout(const uint32_t* chunk_off, uint32_t D, uint32_t m, uint32_t* psz_out, uint32_t* mp_out);

/* Host-side re-layout of the file-order pivot table [256][D]:
 * out[(c*256 + code)*psz + i] = pivots[code][chunk_off[c] + i], zero beyond the chunk and for c >= m.
 * out holds mp*256*psz floats. */
int bang_pack_pivots(const float* pivots, const uint32_t* chunk_off, uint32_t D, uint32_t m, uint32_t psz,
                     uint32_t mp, float* out);
/* Exact-size table for layouts whose chunks have 2 dims first and then 1 (DiskANN's split when D/m is between 1 and 2, e.g.
 * 128 dims in 70 chunks, 96 in 74): [nhi][256][2] f32 followed by [mp - nhi][256][1], zero entries for the padding chunks, rounded up
 * to a multiple of 4 floats plus 4 (the kernel reads one float past a 1-dim entry).  *nhi_out = 0 and nothing is written if the
 * layout is not of that form.  out may be NULL to query the size.  256*D floats instead of 512*mp: 96 KB instead of 152 KB of LDS for
 * DEEP100M's layout, 128 KB instead of 144 KB for SIFT1B's. */
int bang_pack_pivots_ragged(const float* pivots, const uint32_t* chunk_off, uint32_t D, uint32_t m, uint32_t mp, uint32_t* nhi_out,
                            float* out, uint64_t* floats_out);

/* Centred queries in the same padded layout: d_qc[q][c*psz + i] = float(query[j]) - centroid[j],
 * j = chunk_off[c] + i (0 in the padding and beyond D - dim_adjust).  First half of
 * populate_pqDist_par (bang_search.cu:1099-1113,1124).  d_qc holds Q*mp*psz floats. */
int bang_k_center_queries(const void* d_queries, int dtype, const float* d_centroid, const uint32_t* d_chunk_off,
                          float* d_qc, uint32_t Q, uint32_t D, uint32_t m, uint32_t mp, uint32_t psz,
                          uint32_t dim_adjust, void* stream);

/* K1 populate_pqDist_par (bang_search.cu:1083-1130): d_lut [Q][m][256] f32. d_pivots_T is [D][256]. */
int bang_k_lut_build(const float* d_pivots_T, const void* d_queries, int dtype, const float* d_centroid,
                     const uint32_t* d_chunk_off, float* d_lut, uint32_t Q, uint32_t D, uint32_t m,
                     uint32_t dim_adjust, void* stream);

/* Parameters of the per-iteration kernels.  One row per query in every [Q][...] array. */
typedef struct {
  uint32_t Q, R, m, L, medoid, iter;   /* iter = reference's 1-based iteration number */
  uint32_t psz, mp;                    /* pivot layout (bang_pq_layout); psz == 0 => LUT path */
  uint32_t first;                      /* 1: use the seed list + compute_parent1 semantics */
  uint32_t max_wgs;                    /* cap on workgroups of the front kernel (0 = one per CU); lanes share the GPU */
  /* straggler compaction: when d_qmap != NULL the kernels iterate over Q SLOTS and slot s works on query d_qmap[s]
   * (all per-query arrays stay indexed by the query); n_all = number of queries behind the arrays (parents copy). */
  const uint32_t* d_qmap;
  uint32_t n_all;                      /* (bang_k_pqdist_stream: != 0 = the Q neighbour rows belong to n_all distinct queries, row q -> query q mod n_all) */
  /* inputs */
  const uint32_t* d_stage;             /* [Q][BANG_STAGE_STRIDE] staged adjacency {count, ids} (first==0) */
  const uint32_t* d_seed;              /* [1 + R+1] {count, MEDOID, adj(MEDOID)...}  (first==1) */
  const uint8_t* d_codes;              /* [N][m] + 256 B slack */
  const float* d_pivots_packed;        /* [mp][256][psz] (psz != 0); with pq_nhi != 0 (psz == 2): [pq_nhi][256][2] then [mp - pq_nhi][256][1] */
  const float* d_qc;                   /* [Q][mp*psz]                           (psz != 0) */
  const float* d_lut;                  /* [Q][m][256]                           (psz == 0) */
  /* graph-on-device mode: adjacency read by the kernel itself from d_graph via d_parents */
  const uint8_t* d_graph;              /* NULL in host-graph mode */
  uint64_t entry_len;
  uint32_t vec_bytes;                  /* D*sizeof(T): offset of the degree word inside an entry */
  /* state */
  uint32_t* d_bloom;                   /* [Q][BANG_BF_WORDS] bit-packed visited filter */
  uint32_t* d_nbrs;                    /* [Q][BANG_NBR_STRIDE] survivors of this iteration */
  float* d_dist;                       /* [Q][BANG_NBR_STRIDE] their PQ distances */
  uint32_t* d_cnt;                     /* [Q] survivor count */
  uint32_t* d_wl_ids;                  /* [Q][L] worklist */
  float* d_wl_dist;                    /* [Q][L] */
  uint8_t* d_wl_vis;                   /* [Q][L] */
  uint32_t* d_wl_cnt;                  /* [Q] */
  uint32_t* d_mark;                    /* [Q] */
  uint32_t* d_parents;                 /* [Q] parent id | BANG_NO_PARENT | BANG_IDLE_PARENT (device memory) */
  uint32_t* d_cand_ids;                /* [Q][L+50] expanded nodes (compact) */
  uint32_t* d_cand_row;                /* [Q][L+50] iteration row holding the node's vector */
  uint32_t* d_cand_cnt;                /* [Q] */
  uint32_t* d_active;                  /* [1] set to 1 if any query is still active (plain store; may be NULL) */
  uint32_t* d_qstats;                  /* [Q][2] per-query running totals {survivors, ids fetched} (may be NULL) */
  /* completion signal of the front kernel (host-graph mode): the last workgroup to finish stores done_value to
   * h_done_flag (mapped pinned host memory) after copying the parents to h_parents and a system-scope release, so the
   * walker thread can spin on it instead of calling into the HIP runtime.  d_done_count is a zero-initialised device word. NULL = off. */
  uint32_t* d_done_count;
  uint32_t* h_done_flag;
  unsigned long long* d_ktime;         /* [gridDim.x][2] per-workgroup {start,end} s_memrealtime stamps (100 MHz) of this launch, or NULL */
  uint32_t* h_parents;                 /* mapped pinned [Q]: the last workgroup copies d_parents there (coalesced) before the flag */
  uint32_t done_value;
  uint32_t pq_nhi;                     /* psz == 2 only: exact-size ("ragged") pivot table -- the first pq_nhi chunks have 2 dims, the rest 1
                                          (bang_pack_pivots_ragged); 0 = every chunk padded to psz floats (bang_pack_pivots) */
  uint32_t code_stride;                /* bytes between the code rows of consecutive nodes in d_codes; 0 = m (the packed layout of
                                          <p>_pq_compressed.bin).  128 for 64 < m <= 128: a row never leaves its 128-byte line */
} bang_iter_params;

/* Fused K5 + K2 + K4: neighbor_filtering_new (bang_search.cu:1140-1165) -> compute_neighborDist_par
 * (:1201-1241) -> compute_parent1 / compute_parent2 (:1464-1521 / :1384-1459), one wavefront per query. */
int bang_k_front(const bang_iter_params* p, void* stream);

/* ---- the query-resident search kernel (csrc/bang_search.hip) ----
 * ONE launch runs the whole search loop of a batch (bang_search.cu:650-958): K5 neighbor_filtering_new (:1140-1165) ->
 * K2 compute_neighborDist_par (:1201-1241) -> K4 compute_parent1/2 (:1464-1521 / :1384-1459) -> K3a/K3b
 * compute_BestLSets_par_sort_msort / _merge (:1533-1585 / :1605-1715), iteration after iteration.  A wavefront owns ONE query at a
 * time from its first iteration to its last (worklist, survivors and counters stay in LDS / registers) and then pulls the next
 * unstarted query from *d_next_query.  Graph resident in HBM (d_graph); the candidate log feeds bang_k_rerank.
 * Same per-query results as bang_k_front / bang_k_back iterated by a host loop. */
typedef struct {
  uint32_t Q, R, m, L, medoid;
  uint32_t cap_iter;                   /* last iteration a query may run: L + BANG_EXTRA_ITERS - 1 (bang_search.cu:950) */
  uint32_t psz, mp, pq_nhi;            /* pivot layout (bang_pq_layout / bang_pack_pivots[_ragged]); psz != 0 */
  uint32_t max_wgs, max_waves;         /* 0 = one workgroup per CU / as many waves per workgroup as LDS holds (<= 16) */
  const uint32_t* d_seed;              /* [2 + R + 1] {count, MEDOID, adj(MEDOID)...} */
  const uint8_t* d_codes;              /* [N][m] + 256 B slack */
  const float* d_pivots_packed;
  const float* d_qc;                   /* [Q][mp*psz] centred queries (bang_k_center_queries) */
  const uint8_t* d_graph;              /* [N][entry_len] */
  uint64_t entry_len;
  uint32_t vec_bytes;
  uint32_t row_layout;                 /* 0: d_graph = graph entries [vec][u32 degree][u32 id x R] at entry_len.  1: d_graph = ADJACENCY ROWS only,
                                          [N][64] u32 at a stride of 256 B, unused slots = 0xFFFFFFFF (BANG_ADJ_PAD) -- the layout the engine
                                          keeps in pinned HOST memory for the pull mode: the kernel fetches rows over PCIe by itself */
  uint32_t* d_bloom;                   /* [Q][BANG_BF_WORDS] visited filters, zeroed (bang_init) */
  uint32_t* d_cand_ids;                /* [Q][L + 50] out: expanded nodes in expansion order, [0] = MEDOID */
  uint32_t* d_cand_cnt;                /* [Q] out */
  uint32_t* d_qstats;                  /* [Q][2] out {distance evaluations, adjacency ids offered to the filter} or NULL */
  uint32_t* d_qiters;                  /* [Q] out: iterations the query ran, or NULL */
  uint32_t* d_next_query;              /* [1] hand-out counter, zeroed before the launch */
  unsigned long long* d_ktime;         /* [workgroups][2] {start, end} s_memrealtime stamps (100 MHz), or NULL */
  /* host-paced form (d_graph == NULL: graph in host RAM, bang_search.cu:771-813 stays on the CPU).  A wave holds nctx (1 or 2)
   * query contexts and the W waves of a workgroup (bang_search_geometry) form groups of group_waves waves; pacing GROUP grp =
   * (g * groups_per_workgroup + group) * nctx + c, whose v-th wave owns slot 16*grp + v.  Every round of a group the workgroup stores its parents to h_parents[slot] (BANG_NO_PARENT: nothing to fetch) and
   * then the round number to h_done[16 grp] (0xFFFFFFFF: the group has finished).  The walker writes the parents' adjacency ids to
   * d_rows[slot][64] and then the group's control line d_ctl[grp][16] = {round number (0xFFFFFFFF = stop), 16 count bytes, 0...} --
   * both in LOCAL fine-grained device memory, through the PCIe BAR, rows before the control line.  ship_vectors: the walker also
   * needs to know where an expanded node's full-precision vector goes in the vector log ([Q][L + 50][vec_bytes], by query and
   * candidate index): h_pub_q[slot] = query | (row wanted) << 31, h_pub_c[slot] = candidate index. */
  const uint32_t* d_rows;              /* [G*nctx*16][64] */
  const uint32_t* d_ctl;               /* [G*nctx][16] */
  uint32_t* h_done;                    /* mapped pinned [G*nctx][16] */
  uint32_t* h_parents;                 /* mapped pinned [G*nctx][16] */
  uint32_t* h_pub_q;                   /* mapped pinned [G*nctx][16] (ship_vectors) */
  uint32_t* h_pub_c;                   /* mapped pinned [G*nctx][16] (ship_vectors) */
  uint32_t* d_abort;                   /* [1] set when a workgroup gave up waiting for the host, or NULL */
  uint32_t ship_vectors;
  uint32_t nctx;                       /* 0 = auto (bang_search_geometry) */
  uint32_t group_waves;                /* waves per pacing group, 4..16 (0 = 8): a workgroup's W waves form ceil(W / group_waves) groups that
                                          advance independently; pacing group index = (g * groups_per_workgroup + group) * nctx + c */
  unsigned long long* d_prof;          /* diagnostic, host-paced form: [G][8] 100 MHz ticks thread 0 of each workgroup spent {waiting for
                                          rows, in the front half up to the publish barrier, publishing, in sort/merge}, [4] = half-rounds; or NULL */
  uint32_t code_stride;                /* bytes between code rows in d_codes; 0 = m (see bang_iter_params) */
  uint32_t n_rows_hbm;                 /* row_layout = 1: the adjacency rows of the nodes [0, n_rows_hbm) are ALSO in device memory at d_rows_hbm */
  const uint32_t* d_rows_hbm;          /* [n_rows_hbm][64] u32, or NULL */
  /* row_layout = 1, n_slices > 1 (peer rows): node p's row is read from d_row_slices[p / slice_rows] + p * 256 when p / slice_rows <
   * n_slices and that entry is not 0 (the entries are BIASED device addresses: slice s's first row is node s * slice_rows), else from
   * d_graph.  n_rows_hbm / d_rows_hbm are ignored then. */
  const uint64_t* d_row_slices;        /* [n_slices] device array of biased base addresses (this GPU's HBM or a peer's, hipIpcOpenMemHandle), or NULL */
  uint32_t n_slices, slice_rows;
  unsigned long long go_timeout_ticks; /* host-paced form: a pacing group that has waited this many 100 MHz ticks for its rows sets *d_abort and
                                          leaves (the host is gone); 0 = 30 s */
  uint32_t* d_qskip;                   /* [Q] out, or NULL: filter-word loads the query did NOT issue because its on-chip summary knew the
                                          word was still zero (self-paced form; 0 in the host-paced form) */
  uint32_t n_nodes;                    /* nodes of the index, or 0: an adjacency id >= n_nodes (and not the pad value) is never expanded nor evaluated -- the row
                                          counts as empty and *d_abort is set to 2 (a corrupt row must not become a wild read of the code table) */
  uint32_t summ_iters;                 /* self-paced form: the on-chip filter summary is consulted and maintained for a query's first summ_iters iterations only
                                          (0xFFFFFFFF = always; 0 = auto: always, except 1 -- i.e. off -- for launches of at most 5 queries per CU, where its
                                          LDS-crossbar work on the chain of every iteration costs more than the requests it saves) */
  uint32_t spec_rows;                  /* self-paced form, 70- and 74-chunk layouts (the other instances ignore it): the PQ code rows of ALL ids of an adjacency row are
                                          requested together with their filter probes (1) -- one memory latency less per iteration, the rows of the ids the
                                          filter drops fetched in vain -- or behind the filter, survivors only (2); 0 = auto: 1 where the rows are pulled (row_layout), and
                                          for launches of <= 8 queries per CU where the graph is in HBM. */
  /* K6 + K7 FUSED into the launch (self-paced form, 8-bit vectors; compute_L2Dist :1254-1299, compute_NearestNeighbours :1312-1368): the wave
   * that finishes a query re-ranks its candidate log on the spot -- exact distances to the full-precision vectors at rr_vec_base + id *
   * rr_vec_stride, stable rank by (distance, expansion order) -- and writes the query's k results; no second launch behind the search, and
   * the re-rank of all but the last queries runs under the search of the others.  rr_queries == NULL: not fused (bang_k_rerank* follows).
   * 8-bit vectors: D % 16 == 0, D <= 256, D / 16 a power of two (D / 16 lanes per candidate, exact integers by v_dot4); float vectors: D % 4 == 0, D <= 256
   * (one lane per candidate runs the ascending fmaf chain); rr_vec_stride % 4 == 0, no MIPS padding (bang_search_can_rerank).  Same bits as bang_k_rerank. */
  const void* rr_queries;              /* [rr_Q_total][D] raw queries (u8 / i8 / f32), row rr_q0 + q belongs to this launch's query q */
  const uint8_t* rr_vec_base;
  uint64_t rr_vec_stride;
  uint64_t* rr_ids_out;                /* [rr_Q_total][k] */
  float* rr_dists_out;                 /* [k][rr_Q_total] (rank-major, :999) */
  uint32_t rr_dtype, rr_D, rr_k, rr_q0, rr_Q_total;
} bang_search_params;
/* 1 if a launch with these vectors can carry the fused re-rank (bang_search_params.rr_*) */
int bang_search_can_rerank(int dtype, uint32_t D, uint64_t vec_stride, uint32_t dim_adjust);
int bang_k_search(const bang_search_params* p, void* stream);
/* waves per workgroup that fit the 160 KB of LDS beside the pivot table at worklist length L (0: the kernel cannot run) */
int bang_search_supported(uint32_t psz, uint32_t mp, uint32_t nhi, uint32_t L);
/* grid of a bang_k_search launch over Q queries: workgroups (<= CUs, <= max_wgs if nonzero), waves per workgroup (every wave that fits,
 * <= max_waves if nonzero; no more than a workgroup's share of Q needs; a self-paced batch of one to two wave-fulls per CU: half its share,
 * two equal rounds) and query contexts per wave (*nctx in: 0 = auto, out: 1 or 2) and, host-paced form, waves per pacing group (*group_waves in: 0 = auto = 8) */
int bang_search_geometry(uint32_t psz, uint32_t mp, uint32_t nhi, uint32_t L, uint32_t Q, uint32_t max_wgs, uint32_t max_waves,
                         int host_paced, uint32_t* workgroups, uint32_t* waves, uint32_t* nctx, uint32_t* group_waves);

/* Fused K3a + K3b: compute_BestLSets_par_sort_msort (bang_search.cu:1533-1585) ->
 * compute_BestLSets_par_merge (:1605-1715), one wavefront per query. */
int bang_k_back(const bang_iter_params* p, void* stream);

/* Unfused stage entries (parity tests; same device code as the fused kernels). */
int bang_k_filter(const bang_iter_params* p, void* stream);   /* K5 only: d_stage/d_seed -> d_nbrs, d_cnt */
int bang_k_pqdist(const bang_iter_params* p, void* stream);   /* K2 only: d_nbrs,d_cnt -> d_dist */
int bang_k_parent(const bang_iter_params* p, void* stream);   /* K4 only */
/* K2 only, streaming form (csrc/bang_search.hip): the same distances as bang_k_pqdist for the first min(d_cnt[q], 64) neighbours of
 * every query (R <= 64: every row the search path ever produces but the 65-entry seed list), with the next row's ids and code rows in
 * flight while the current one is reduced.  This is the launch the K2-alone HBM roofline figure is measured on. */
int bang_k_pqdist_stream(const bang_iter_params* p, void* stream);

/* Fused K6 + K7: compute_L2Dist (bang_search.cu:1254-1299) -> compute_NearestNeighbours (:1312-1368).
 * Candidate i of query q has its vector at d_vec_base + (row*Q + q)*vec_stride (row = d_cand_row, host-graph
 * mode: the per-iteration vector log) or at d_vec_base + id*vec_stride (d_cand_row == NULL: graph on device).
 * d_ids_out [Q][k] u64; d_dists_out [k][Q] f32. */
int bang_k_rerank(const void* d_vec_base, uint64_t vec_stride, const void* d_medoid_vec, const void* d_queries,
                  int dtype, const uint32_t* d_cand_ids, const uint32_t* d_cand_row, const uint32_t* d_cand_cnt,
                  uint32_t cand_stride, uint32_t Q, uint32_t D, uint32_t k, uint32_t dim_adjust,
                  uint64_t* d_ids_out, float* d_dists_out, void* stream);

#ifdef __cplusplus
}
#endif
#endif /* BANG_C_H_ */
