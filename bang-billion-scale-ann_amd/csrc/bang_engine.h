// bang_engine.h -- internal state of the host engine, shared by its translation units (not public):
//   bang_options.cpp  every option / environment switch in ONE table
//   bang_load.cpp     bang_load: files, entry sources, placement, pull rows, streamed load (bang_search.cu:138-362)
//   bang_alloc.cpp    bang_alloc / bang_free: per-batch buffers, lanes, loop form (bang_search.cu:370-425)
//   bang_walker.cpp   the C++ graph walker (bang_search.cu:771-813) in its three forms + the thread teams
//   bang_lane.cpp     bang_query of one lane: H2D -> K1 -> search -> re-rank -> D2H (bang_search.cu:569-1068)
//   bang_cabi.cpp     the engine-level C-ABI of include/bang_c.h
#ifndef BANG_ENGINE_H_
#define BANG_ENGINE_H_

#include <hip/hip_runtime_api.h>

#include <algorithm>
#include <atomic>
#include <cctype>
#include <chrono>
#include <condition_variable>
#include <mutex>
#include <cstdarg>
#include <cstdio>
#include <cstdlib>
#include <cstring>
#include <memory>
#include <string>
#include <immintrin.h>
#include <sched.h>
#include <fcntl.h>
#include <sys/mman.h>
#include <sys/stat.h>
#include <sys/statvfs.h>
#include <sys/syscall.h>
#include <unistd.h>
#include <thread>
#include <vector>

#include "bang_c.h"
#include "bang_internal.h"

#define HIP_TRY(x)                                                                                \
  do {                                                                                            \
    hipError_t _e = (x);                                                                          \
    if (_e != hipSuccess) {                                                                       \
      bang_set_error("%s failed: %s (%s:%d)", #x, hipGetErrorString(_e), __FILE__, __LINE__);     \
      return BANG_ERR_HIP;                                                                        \
    }                                                                                             \
  } while (0)
#define LANE_HIP(x) HIP_TRY(x)
#define BANG_TRY(x)            \
  do {                         \
    int _r = (x);              \
    if (_r != BANG_OK) return _r; \
  } while (0)

namespace bang {

using Clock = std::chrono::steady_clock;
inline double ms_since(Clock::time_point t0) {
  return std::chrono::duration<double, std::milli>(Clock::now() - t0).count();
}

constexpr size_t KT_WGS = 256;   // workgroups per front launch never exceed the CU count

struct Lane {
  uint32_t q0 = 0, nq = 0;
  int index = 0;
  hipStream_t s_main = nullptr, s_fp = nullptr;
  hipEvent_t ev_front = nullptr, ev_fp = nullptr;
  unsigned long long* d_ktime = nullptr;   // [max launches][KT_WGS][2] in-kernel stamps, "timing"=1
  size_t kt_launches = 0, kt_used = 0;
  // walker team of this lane: the lane thread + (threads-1) helpers, spin-synchronised while a query runs
  std::vector<std::thread> helpers;
  std::atomic<uint32_t> epoch{0};
  std::atomic<uint32_t> pending{0};
  std::atomic<bool> team_active{false};
  uint32_t job_row = 0;
  bool job_adj = false;
  // straggler compaction: slot -> query maps (double buffered), CPU-writable (BAR) or mapped pinned
  uint32_t* qmap_host[2] = {nullptr, nullptr};   // where the CPU writes
  uint32_t* qmap_dev[2] = {nullptr, nullptr};    // what the kernels read
  bool qmap_is_device = false;
  std::vector<uint32_t> parents_tmp;             // device-graph mode: parents fetched at a poll
  uint32_t* d_pcnt = nullptr;
  int job_kind = 0;                              // what the walker team does on the next epoch: 0 = slice walk, 2 = search-kernel walk (swalk)
  uint32_t pw_groups = 0;                        // pacing groups of the running search kernel
  std::unique_ptr<std::atomic<uint32_t>[]> pw_expect;   // [pacing groups] shared by the walker team (see swalk)
  std::atomic<uint32_t> pw_remaining{0};
  mutable std::atomic<uint64_t> h2d_bytes{0};   // bumped by the walker through a const Lane&
  std::atomic<int> pw_error{0};
  std::atomic<uint32_t> job_active{0}, job_parents{0};
  std::atomic<int> phase{0};          // debugging aid: what the lane thread is doing (see watchdog)
  std::atomic<uint32_t> phase_iter{0};
  // results of the last run
  int rc = BANG_OK;
  std::string err;
  uint32_t iterations = 0;
  uint64_t front_launches = 0;
  double walker_ms = 0, front_ms = 0, back_ms = 0, rerank_ms = 0, sync_ms = 0, enqueue_ms = 0;
};

struct Pool {                       // persistent lane threads, woken once per bang_query
  std::mutex m;
  std::condition_variable cv_start, cv_done, cv_team;
  uint64_t query_seq = 0;
  int lanes_done = 0;
  bool shutdown = false;
  std::atomic<bool> shutdown_flag{false};   // same, readable without the mutex (spinning helpers)
  const void* h_queries = nullptr;
  uint64_t* h_ids = nullptr;
  float* h_dists = nullptr;
  uint64_t* d_ids_user = nullptr;    // bang_query_dev_e: the results stay on the device, in the caller's buffers
  float* d_dists_user = nullptr;
  int Q = 0;
  std::vector<std::thread> lane_threads;
};

}  // namespace bang

struct bang_engine {
  int dtype = BANG_U8;
  size_t tsize = 1;
  // options
  int graph_mode = BANG_GRAPH_AUTO;   // resolved to HOST / DEVICE by bang_load
  int graph_opt = BANG_GRAPH_AUTO;    // what the caller asked for (restored by bang_unload)
  int lanes_opt = 0;      // 0 = auto
  int threads_opt = 0;    // walker threads per lane (lane thread + helpers); 0 = auto from the CPU quota
  int device = 0;
  int pq_mode = 0;        // 0 auto (pivot-stationary if possible), 1 force LUT path
  int timing = 0;
  int front_wgs_opt = -1; // -1 auto
  int front_wgs = 0;      // workgroups per front-kernel launch (0 = all CUs); set from the lane count
  int check_every = 16;   // device-graph mode: poll the active counter every N iterations
  // index
  bool loaded = false;
  uint64_t medoid = 0, entry_len = 0;
  uint32_t D = 0, R = 0, N = 0, m = 0;
  const uint8_t* graph = nullptr;   // host
  uint8_t* graph_owned = nullptr;   // private copy (fread), or
  void* graph_map = nullptr;        // the graph file mapped MAP_SHARED: N processes of one node share ONE copy in the page cache
  size_t graph_map_len = 0;
  uint8_t* d_graph = nullptr;       // BANG_GRAPH_DEVICE
  uint8_t* d_codes = nullptr;
  bool codes_owned = false;
  float* d_pivots_T = nullptr;      // [D][256]
  float* d_pivots_packed = nullptr; // [mp][256][psz]
  float* d_centroid = nullptr;
  uint32_t* d_chunk_off = nullptr;
  uint32_t* d_seed = nullptr;       // {count, MEDOID, adj(MEDOID)...}
  uint8_t* d_medoid_vec = nullptr;
  uint32_t psz = 0, mp = 0;
  int code_stride_opt = -1;            // option "code_stride": -1 auto (pad rows to a power of two when HBM allows), 0 packed (m), > 0 bytes
  uint32_t code_stride = 0;            // resolved at load: bytes between code rows in d_codes
  // search params
  int k = 0, L = 0, distfn = BANG_DIST_L2;
  bool params_set = false;
  // per-alloc state
  bool allocated = false;
  bool inited = false;
  int Qcap = 0;
  int Qcur = 0;          // batch size of the running / last query (row stride of the vector log)
  uint32_t cand_stride = 0;
  void* d_queries = nullptr;
  float* d_qc = nullptr;
  float* d_lut = nullptr;
  uint32_t* d_bloom = nullptr;
  uint32_t* d_nbrs = nullptr;
  float* d_dist = nullptr;
  uint32_t* d_cnt = nullptr;
  uint32_t* d_wl_ids = nullptr;
  float* d_wl_dist = nullptr;
  uint8_t* d_wl_vis = nullptr;
  uint32_t* d_wl_cnt = nullptr;
  uint32_t* d_mark = nullptr;
  uint32_t* d_parents_dev = nullptr;   // device-graph mode
  uint32_t* h_parents = nullptr;       // mapped pinned (host-graph mode)
  uint32_t* d_parents_map = nullptr;   // device alias of h_parents
  uint32_t* d_cand_ids = nullptr;
  uint32_t* d_cand_row = nullptr;
  uint32_t* d_cand_cnt = nullptr;
  uint32_t* d_active = nullptr;        // [L+50 + 2] per-iteration active counters (device-graph mode)
  uint32_t* d_qstats = nullptr;        // [Q][2] per-query {survivors, fetched}
  uint8_t* d_fp = nullptr;             // [(L+50)][Q][vec_bytes] vector log (host-graph mode)
  uint8_t* h_fp = nullptr;             // pinned mirror
  std::vector<uint8_t> h_fin;          // [Q] walker-side: query seen finished (its staged row count is already 0)
  uint32_t* h_stage = nullptr;         // pinned [Q][65]
  // results of a query: ids [Q][k] u64 | dists [k][Q] f32 | iterations [Q] u32 in ONE device allocation (d_results), mirrored by a
  // pinned host buffer (h_results): small batches come back in one asynchronous copy instead of three staged ones
  uint8_t* d_results = nullptr;
  uint8_t* h_results = nullptr;
  uint8_t* h_results_dev = nullptr;    // the device's address of h_results: the fused re-rank writes a small batch's results there itself (no copy behind the launch)
  size_t res_off_dists = 0, res_off_iters = 0, res_bytes = 0;
  uint64_t* d_ids_out = nullptr;
  float* d_dists_out = nullptr;
  std::vector<std::unique_ptr<bang::Lane>> lanes;
  bang::Pool pool;
  uint32_t* d_stage = nullptr;         // [Q][65] device copy of the staged adjacency rows (one H2D per lane and iteration)
  uint32_t* h_done = nullptr;          // mapped pinned [lanes*16]: completion flags written by the front kernel
  uint32_t* h_done_dev = nullptr;
  uint32_t* d_done_count = nullptr;    // [lanes*16] device arrival counters
  int stage_zero_copy = -1;            // -1: auto (2 on large-BAR devices, else 1)
                                       // 0: H2D copy of the staged rows per lane and iteration (SDMA)
                                       // 1: the front kernel reads the staged rows in place from mapped pinned memory
                                       // 2: the walker writes the rows straight into device memory through the PCIe BAR
                                       //    (large-BAR systems: hipMalloc'ed memory is CPU-writable; write-combined stores)
  uint32_t* h_stage_dev = nullptr;     // device alias of h_stage
  int threads_eff = 1, stage_mode_eff = 1;   // resolved at bang_alloc
  int use_flag = 1;                    // 0: wait for the front kernel with hipStreamSynchronize + D2H copy of the parents (debug/ablation)
  int compact = 1;                     // straggler compaction on/off
  int persistent = -1;                 // host-graph mode: 1 = ONE persistent search kernel per batch, its workgroups paced by the walker threads;
                                       // 0 = a front + back launch per iteration and lane; -1 = auto (1 where the walker can write device memory: BAR)
  int numa_opt = -1;                   // host-graph mode: 1 = pin the walker threads (and the caller for the duration of a query) to the CPUs
                                       // of the GPU's NUMA node, one physical core each; 0 / -1 (auto) = leave them to the scheduler
  cpu_set_t numa_cpus;                 // resolved at bang_alloc
  std::vector<int> numa_cores;         // one CPU per distinct physical core of that node (walker thread i is pinned to numa_cores[i % n])
  bool numa_on = false;
  int numa_node = -1;
  int search_opt = -1;                 // 1 = the query-resident search kernel (bang_search.hip), 0 = a launch per iteration,
                                       // -1 = auto (1 where the pivot table leaves LDS for at least 4 waves' worklists)
  bool search_v2 = false;              // resolved at bang_alloc: graph in HBM, self-paced form
  bool search_host = false;            // resolved at bang_alloc: graph in host RAM, the host-paced form of the same kernel (BAR mode)
  uint32_t sv_G = 0, sv_W = 0, sv_C = 1;   // its grid for the running query: workgroups, waves per workgroup, query contexts per wave
  uint32_t sv_GS = 8, sv_NG = 0;           // waves per pacing group; pacing groups = workgroups x groups per workgroup x contexts
  uint32_t* d_srows = nullptr;         // fine-grained device memory [groups*16][64]: adjacency ids per slot, written through the BAR
  uint32_t* d_sctl = nullptr;          // fine-grained device memory [groups][16]: control line per pacing group {round, 16 count bytes}
  uint32_t* h_pub_q = nullptr;         // mapped pinned [groups][16]: query | row wanted << 31 per slot (vectors shipped by the walker)
  uint32_t* h_pub_c = nullptr;         //                             candidate index per slot
  uint32_t* d_pub_q = nullptr;         // device aliases
  uint32_t* d_pub_c = nullptr;
  uint32_t* d_qiters = nullptr;        // [Q] iterations per query (search kernel)
  uint32_t* d_qskip = nullptr;         // [Q] filter-word loads saved by the on-chip summary (search kernel, self-paced)
  bool rerank_fused = false;           // the last bang_query re-ranked inside the search launch
  int fuse_rerank = -1;                // option "fuse_rerank": K6 + K7 inside the search launch (self-paced form; 8-bit and float vectors); -1 = auto = on
  std::vector<uint32_t> h_qiters;
  bool stage_local = false;            // rows are staged in local device memory (BAR mode)
  int pq_ragged = 1;                   // 2-float PQ layouts: exact-size pivot table where possible (0 = always the padded table)
  uint32_t pq_nhi_avail = 0;           // resolved at load: leading 2-dim chunks of the exact-size table in d_pivots_ragged, 0 = none
  uint32_t pq_nhi = 0;                 // resolved at bang_alloc: the table the kernels of this allocation use (0 = padded)
  float* d_pivots_ragged = nullptr;
  int vectors_opt = -1;                // host-graph mode, where the full-precision vectors for the re-rank live: 0 = host (the walker ships
                                       // every expanded node's vector, as the reference does), 1 = a packed copy [N][vec_bytes] in HBM (the
                                       // walker ships adjacency rows only), -1 = auto (1 if the copy takes at most 40 % of the free HBM)
  bool vec_on_device = false;          // resolved at load
  // PULL mode of the host-graph placement: the adjacency lists alone, as [N][64] u32 rows of 256 B (unused slots 0xFFFFFFFF), in
  // pinned host memory mapped into the GPU's address space.  The self-paced search kernel fetches a parent's row over PCIe by
  // itself (one 256-B read, ~2 us; 57 GB/s of such rows measured) -- no walker thread, no publish / poll round trip.
  int pull_opt = -1;                   // -1 auto, 0 = walker (host-paced kernel), 1 = pull
  int walker_opt = 0;                  // option "walker": 1 = the C++ walker threads serve the adjacency rows although the kernel could pull them itself
  bool walker_rows = false;            // the walker team reads the 256-byte pull rows (h_adj) instead of graph entries (resolved at bang_alloc)
  bool walker_self = false;            // last launch of the walker-from-rows form: the kernel served the rows of this GPU's HBM copy itself
  bool pull = false;                   // resolved at load
  uint32_t* h_adj = nullptr;           // [N][64]
  size_t adj_bytes = 0;
  std::string rows_path;               // the rows file the mapping h_adj belongs to ("" = anonymous memory): re-checked at bang_alloc
  std::string rows_key;                // names the shared rows file (BANG_PULL_ROWS_DIR): basename of the index prefix
  // STREAMED load: the graph entries pass through in chunks (vectors -> HBM, adjacency -> pull rows) and are not kept
  bang_entry_source entry_fn = nullptr;   // set for the duration of a streamed load
  void* entry_ctx = nullptr;
  std::string graph_path;              // file loads: `<p>_disk.bin`, mapped only if a walker form ever needs the entries
  bool graph_streamed = false;         // loaded without a resident graph (graph == nullptr): only the pull mode can run as is
  bool entry_src_rereadable = false;   // the entry source is ours (a file): placements that need the whole graph may read it all
  const uint32_t* d_adj = nullptr;     // device address of h_adj
  int rows_hbm_opt = -1;               // option "rows_hbm": MB of HBM for a copy of the first adjacency rows (-1 = whatever the index leaves beyond
                                       // 12 GB -- where the rows do not all fit --, 0 = none): read from HBM instead of pulled over PCIe
  uint32_t* d_rows_hbm = nullptr;      // [n_rows_hbm][64]: the rows [rows_first, rows_first + n_rows_hbm)
  uint32_t n_rows_hbm = 0;
  uint64_t rows_first = 0;             // first node of the HBM copy (0 unless bang_rows_slice_e moved it: peer rows)
  bool rows_exported = false;          // an IPC handle of d_rows_hbm is out: never freed before bang_unload
  // peer rows (bang_rows_import_e): slice s of the node's HBM-resident rows starts at node s * slice_rows
  uint32_t n_slices = 0, slice_rows = 0, own_slot = 0xFFFFFFFFu;
  void* peer_ptr[BANG_MAX_ROW_SLICES] = {nullptr};       // hipIpcOpenMemHandle mappings (closed at bang_unload); the own slot stays NULL
  uint64_t slice_base[BANG_MAX_ROW_SLICES] = {0};        // biased addresses handed to the kernel (0: not available -> host rows)
  uint64_t* d_slice_tab = nullptr;     // device copy of slice_base
  uint8_t* d_vecs = nullptr;           // [N][vec_bytes]
  bool vecs_owned = true;              // false: d_vecs is the caller's buffer (bang_index_desc.d_vectors), never freed here
  uint8_t* ext_vecs = nullptr;         // set for the duration of a load: the caller's vector buffer, and whether it is filled already
  bool ext_vecs_ready = false;
  uint64_t rows_hash = 0;              // hash over every adjacency list of the loaded index (the pull rows' signature)
  bool fp_direct = false;              // the walker writes the full-precision vectors straight into d_fp (BAR), no staging copy
  int stagger_us = 0;                  // lane i starts i*stagger_us later (de-synchronises the lanes' PCIe phases)
  int fp_batch = 16;                   // vector-log rows are copied to the device every fp_batch iterations
  // hand-shake timeouts of the host-paced search kernel (options; the HOST gives up first and stops the kernel through its control
  // lines, a pacing group only gives up on its own -- the host process is gone -- well after that)
  int host_walk_timeout_ms = 20000;
  int kernel_go_timeout_ms = 30000;
  int walker_stall_ms = 0;             // test hook: the walker team sleeps this long at the start of the next host-paced query (one shot)
  bang_stats stats{};
};

namespace bang {

inline size_t vec_bytes(const bang_engine* e) { return (size_t)e->D * e->tsize; }

inline int ensure_device(bang_engine* e) {
  if (bang_device_count() == 0) {
    bang_set_error("no HIP device visible: libbang has no CPU fallback");
    return BANG_ERR_NOGPU;
  }
  HIP_TRY(hipSetDevice(e->device));
  return BANG_OK;
}

template <typename T>
int dmalloc(T** p, size_t count) {
  const hipError_t err = hipMalloc((void**)p, std::max<size_t>(count * sizeof(T), 16));
  if (err == hipErrorOutOfMemory) {                 // told apart from other failures: bang_alloc retries without the HBM row cache
    (void)hipGetLastError();
    bang_set_error("out of device memory (%zu bytes wanted)", count * sizeof(T));
    return BANG_ERR_NOMEM;
  }
  HIP_TRY(err);
  return BANG_OK;
}
template <typename T>
void dfree(T*& p) {
  if (p) (void)hipFree((void*)p);
  p = nullptr;
}

// CPUs this process may really use: affinity mask capped by the cgroup CPU quota (the MI355X boxes expose 256 hardware

// ---- bang_options.cpp: every option / environment switch in one table
int set_option(bang_engine* e, const char* key, long value);
void apply_env_defaults(bang_engine* e);         // bang_create: presets from the environment
long env_long(const char* name, long dflt);      // environment switches are read when they are used
bool env_flag(const char* name);
const char* env_str(const char* name);
// ---- bang_load.cpp
int usable_cpus();                       // CPUs this process may really use (affinity mask capped by the cgroup quota)
size_t host_bytes_available();           // host memory this process may still take (MemAvailable capped by the cgroup)
int upload_index(bang_engine* e, const uint8_t* h_codes, const void* d_codes_ext, const float* pivots, const float* centroid,
                 const uint32_t* chunk_off, uint32_t desc_code_stride);
void unload_index(bang_engine* e);
int load_files(bang_engine* e, const char* prefix);
int load_shared(bang_engine* e, uint64_t expect_rows_hash);    // vectors already in the caller's device buffer, rows in the node's rows file
int map_graph_file(bang_engine* e);
int validate_pull_rows(bang_engine* e);   // pull mode, at bang_alloc: the rows mapping is still what bang_load registered
// ---- bang_alloc.cpp
int alloc_buffers(bang_engine* e, int Q);
void free_batch(bang_engine* e);
// ---- bang_walker.cpp
bool gpu_numa_cpus(int device, cpu_set_t* out, int* node_out);
std::vector<int> distinct_cores(const cpu_set_t& cpus);
void pin_walker_thread(const bang_engine* e, int index);
void swalk(bang_engine* e, Lane& ln, int t, int T);
uint32_t walk(bang_engine* e, Lane& ln, uint32_t row, bool adjacency, uint32_t* n_parents);
int wait_flag(bang_engine* e, Lane& ln, uint32_t value);
void lane_job(bang_engine* e, Lane& ln);
void start_threads(bang_engine* e);
void stop_threads(bang_engine* e);
// ---- bang_lane.cpp
int lane_run(bang_engine* e, Lane& ln, const void* h_queries, uint64_t* h_ids, float* h_dists, int Q);

}  // namespace bang
#endif
