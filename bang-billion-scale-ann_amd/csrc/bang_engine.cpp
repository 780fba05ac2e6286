// bang_engine.cpp -- host side of the MI355X-native BANG_Base search engine.
//
// Mirrors the reference's BANGSearchInner<T> (BANG_Base/bang_search.cuh:341-370; implementation
// bang_search.cu:138-1068): bang_load / bang_set_searchparams / bang_alloc / bang_init /
// bang_query / bang_free / bang_unload, behind the C-ABI of include/bang_c.h.
//
// What is different from the reference, on purpose:
//  * the batch is split into LANES (contiguous query ranges).  Each lane owns a HIP stream, pinned
//    staging buffers and a host walker thread, and runs its own iteration loop, so one lane's
//    PCIe / CPU latency hides behind another lane's kernels (the reference advances all 10K
//    queries in lock-step and pays every round trip serially, bang_search.cu:701-958).
//  * per iteration a lane issues 2 kernels (front = filter+distance+parent, back = sort+merge)
//    instead of 6 kernels + 2 memsets.  Host and device talk through mapped pinned memory, not
//    copies: the kernel writes the parents and a completion flag straight into host memory (the
//    walker spins on the flag, no HIP call).  Adjacency rows travel in one H2D copy per lane and
//    iteration (reading them in place over PCIe from the kernel was measured 5x slower: narrow
//    reads, few in flight); full-precision vectors travel by batched async copies on a second stream.
//  * lane and walker threads are persistent (created in bang_alloc), woken per query.
//  * graph placement is a run-time option: host RAM + C++ walker (BANG_Base) or HBM-resident
//    (BANG_Inmemory placement) with the identical search semantics, hence identical results.
//  * there is NO CPU fallback: without a HIP device every call fails with BANG_ERR_NOGPU.

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
#include <unistd.h>
#include <thread>
#include <vector>

#include "bang_c.h"
#include "bang_internal.h"

// ------------------------------------------------------------------ errors
static thread_local char g_err[512] = "";
extern "C" void bang_set_error(const char* fmt, ...) {
  va_list ap;
  va_start(ap, fmt);
  vsnprintf(g_err, sizeof(g_err), fmt, ap);
  va_end(ap);
}
extern "C" const char* bang_last_error(void) { return g_err; }

#define HIP_TRY(x)                                                                                \
  do {                                                                                            \
    hipError_t _e = (x);                                                                          \
    if (_e != hipSuccess) {                                                                       \
      bang_set_error("%s failed: %s (%s:%d)", #x, hipGetErrorString(_e), __FILE__, __LINE__);     \
      return BANG_ERR_HIP;                                                                        \
    }                                                                                             \
  } while (0)
#define BANG_TRY(x)            \
  do {                         \
    int _r = (x);              \
    if (_r != BANG_OK) return _r; \
  } while (0)

extern "C" int bang_device_count(void) {
  int n = 0;
  if (hipGetDeviceCount(&n) != hipSuccess) return 0;
  return n;
}

// ------------------------------------------------------------------ device helpers
extern "C" int bang_dev_malloc(void** d_ptr, size_t bytes) {
  if (!d_ptr) return BANG_ERR_ARG;
  if (bang_device_count() == 0) { bang_set_error("no HIP device"); return BANG_ERR_NOGPU; }
  HIP_TRY(hipMalloc(d_ptr, bytes ? bytes : 4));
  return BANG_OK;
}
extern "C" int bang_dev_free(void* d_ptr) { if (d_ptr) HIP_TRY(hipFree(d_ptr)); return BANG_OK; }
extern "C" int bang_dev_memset(void* d_ptr, int value, size_t bytes) { HIP_TRY(hipMemset(d_ptr, value, bytes)); return BANG_OK; }
extern "C" int bang_dev_h2d(void* d_dst, const void* h_src, size_t bytes) {
  HIP_TRY(hipMemcpy(d_dst, h_src, bytes, hipMemcpyHostToDevice));
  return BANG_OK;
}
extern "C" int bang_dev_d2h(void* h_dst, const void* d_src, size_t bytes) {
  HIP_TRY(hipMemcpy(h_dst, d_src, bytes, hipMemcpyDeviceToHost));
  return BANG_OK;
}
extern "C" int bang_dev_sync(void) { HIP_TRY(hipDeviceSynchronize()); return BANG_OK; }

// ------------------------------------------------------------------ engine state
namespace {

using Clock = std::chrono::steady_clock;
static double ms_since(Clock::time_point t0) {
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

}  // namespace

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
  size_t res_off_dists = 0, res_off_iters = 0, res_bytes = 0;
  uint64_t* d_ids_out = nullptr;
  float* d_dists_out = nullptr;
  std::vector<std::unique_ptr<Lane>> lanes;
  Pool pool;
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
  bool pull = false;                   // resolved at load
  uint32_t* h_adj = nullptr;           // [N][64]
  size_t adj_bytes = 0;
  std::string rows_key;                // names the shared rows file (BANG_PULL_ROWS_DIR): basename of the index prefix
  // STREAMED load: the graph entries pass through in chunks (vectors -> HBM, adjacency -> pull rows) and are not kept
  bang_entry_source entry_fn = nullptr;   // set for the duration of a streamed load
  void* entry_ctx = nullptr;
  std::string graph_path;              // file loads: `<p>_disk.bin`, mapped only if a walker form ever needs the entries
  bool graph_streamed = false;         // loaded without a resident graph (graph == nullptr): only the pull mode can run as is
  bool entry_src_rereadable = false;   // the entry source is ours (a file): placements that need the whole graph may read it all
  const uint32_t* d_adj = nullptr;     // device address of h_adj
  uint8_t* d_vecs = nullptr;           // [N][vec_bytes]
  bool fp_direct = false;              // the walker writes the full-precision vectors straight into d_fp (BAR), no staging copy
  int stagger_us = 0;                  // lane i starts i*stagger_us later (de-synchronises the lanes' PCIe phases)
  int fp_batch = 16;                   // vector-log rows are copied to the device every fp_batch iterations
  bang_stats stats{};
};

namespace {

size_t vec_bytes(const bang_engine* e) { return (size_t)e->D * e->tsize; }

int ensure_device(bang_engine* e) {
  if (bang_device_count() == 0) {
    bang_set_error("no HIP device visible: libbang has no CPU fallback");
    return BANG_ERR_NOGPU;
  }
  HIP_TRY(hipSetDevice(e->device));
  return BANG_OK;
}

template <typename T>
int dmalloc(T** p, size_t count) {
  HIP_TRY(hipMalloc((void**)p, std::max<size_t>(count * sizeof(T), 16)));
  return BANG_OK;
}
template <typename T>
void dfree(T*& p) {
  if (p) (void)hipFree((void*)p);
  p = nullptr;
}

// CPUs this process may really use: affinity mask capped by the cgroup CPU quota (the MI355X boxes expose 256 hardware
// threads but grant 16 CPUs; spinning walker threads beyond the quota only starve each other)
static int usable_cpus() {
  int n = (int)std::thread::hardware_concurrency();
  cpu_set_t set;
  if (sched_getaffinity(0, sizeof(set), &set) == 0) n = CPU_COUNT(&set);
  if (FILE* f = fopen("/sys/fs/cgroup/cpu.max", "r")) {
    char q[64];
    long period = 0;
    if (fscanf(f, "%63s %ld", q, &period) == 2 && strcmp(q, "max") != 0 && period > 0) {
      const long quota = atol(q);
      if (quota > 0) n = std::min<int>(n, (int)std::max<long>(1, quota / period));
    }
    fclose(f);
  }
  return std::max(1, n);
}

// host memory this process may still take: MemAvailable capped by what the cgroup has left
static size_t host_bytes_available() {
  size_t avail = ~(size_t)0;
  if (FILE* f = fopen("/proc/meminfo", "r")) {
    char line[128];
    while (fgets(line, sizeof(line), f)) {
      unsigned long long kb = 0;
      if (sscanf(line, "MemAvailable: %llu kB", &kb) == 1) { avail = (size_t)kb * 1024; break; }
    }
    fclose(f);
  }
  unsigned long long mx = 0, cur = 0;
  bool have_mx = false, have_cur = false;
  if (FILE* f = fopen("/sys/fs/cgroup/memory.max", "r")) { have_mx = fscanf(f, "%llu", &mx) == 1; fclose(f); }   // "max": no limit
  if (FILE* f = fopen("/sys/fs/cgroup/memory.current", "r")) { have_cur = fscanf(f, "%llu", &cur) == 1; fclose(f); }
  if (have_mx && have_cur) avail = std::min<size_t>(avail, mx > cur ? (size_t)(mx - cur) : 0);
  return avail;
}

// Pull mode: the adjacency lists of the host graph, re-laid as 256-byte rows the GPU can fetch with one PCIe read each.
// One copy per NODE when BANG_PULL_ROWS_DIR names a directory every rank can see (tmpfs): the rows live in the file
// <dir>/<index name>_pull_rows.bin, built by whichever rank loads first (write to a temporary name, rename) and mapped shared by
// the others; every rank registers the mapping with its own device.  Without the variable: private anonymous memory.
struct PullRowsSig { char magic[8]; uint64_t N, medoid, R, adj_hash; };
// What a rows file must match: sizes, medoid and EVERY adjacency list.  A node's list is hashed together with its number
// (word-wise FNV-style over {node, degree, ids}, then a finaliser) and the node hashes are ADDED: order independent, so the threads
// that split a chunk (or the whole graph) hash their slices on their own and the sums combine.  An index rebuilt or edited in place
// with the same N / R / medoid therefore never inherits another graph's rows.
static inline uint64_t sig_node(const bang_engine* e, uint64_t node, const uint8_t* adj /* [u32 degree][u32 id x R] */) {
  uint32_t deg;
  memcpy(&deg, adj, 4);
  if (deg > e->R) deg = e->R;
  uint64_t h = (0xcbf29ce484222325ull ^ node) * 0x100000001b3ull;
  h = (h ^ deg) * 0x100000001b3ull;
  for (uint32_t k = 0; k < deg; ++k) {
    uint32_t id;
    memcpy(&id, adj + 4 + 4 * (size_t)k, 4);
    h = (h ^ id) * 0x100000001b3ull;
  }
  h ^= h >> 29; h *= 0xBF58476D1CE4E5B9ull; h ^= h >> 32;
  return h;
}
static PullRowsSig sig_make(const bang_engine* e, uint64_t h) {
  PullRowsSig g;
  memcpy(g.magic, "BANGROW2", 8);
  g.N = e->N; g.medoid = e->medoid; g.R = e->R; g.adj_hash = h;
  return g;
}

// one node's adjacency list as a pull row: 64 slots, ids first (ascending), the rest padded
static inline void pull_row_from_entry(const bang_engine* e, uint32_t* row, const uint8_t* adj) {
  uint32_t deg;
  memcpy(&deg, adj, 4);
  if (deg > e->R) deg = e->R;
  memcpy(row, adj + 4, (size_t)deg * 4);
  for (uint32_t k = deg; k < 64; ++k) row[k] = 0xFFFFFFFFu;
}

// Pull mode: the adjacency lists of the host graph, re-laid as 256-byte rows the GPU can fetch with one PCIe read each.
// One copy per NODE when BANG_PULL_ROWS_DIR names a directory every rank can see (tmpfs): the rows live in the file
// <dir>/<index name>_pull_rows.bin, built by whichever rank loads first (write to a temporary name, rename) and mapped shared by
// the others; every rank registers the mapping with its own device.  Without the variable: private anonymous memory.
struct PullRows {
  void* m = MAP_FAILED;
  size_t bytes = 0, sig_off = 0;
  bool fill = true;                    // false: an existing rows file was mapped (its signature is checked in pull_rows_finish)
  std::string path, tmp;
};
static void pull_rows_abandon(PullRows& pr) {
  if (pr.m != MAP_FAILED) (void)munmap(pr.m, pr.bytes);
  pr.m = MAP_FAILED;
  if (!pr.tmp.empty()) (void)unlink(pr.tmp.c_str());
  pr.tmp.clear();
}
// expect: the signature the file must carry if it exists already (NULL: unknown yet -- a streamed load checks at the end)
static int pull_rows_open(bang_engine* e, PullRows& pr, const PullRowsSig* expect) {
  pr.bytes = (size_t)e->N * 256 + 4096;
  pr.sig_off = (size_t)e->N * 256 + 2048;
  if (const char* dir = getenv("BANG_PULL_ROWS_DIR"))
    if (*dir) pr.path = std::string(dir) + "/" + (e->rows_key.empty() ? std::string("index") : e->rows_key) + "_pull_rows.bin";
  if (!pr.path.empty()) {
    const int fd = open(pr.path.c_str(), O_RDWR);
    if (fd >= 0) {                                 // built by another rank of this node (or an earlier run on the same index)
      struct stat st;
      if (fstat(fd, &st) == 0 && (size_t)st.st_size == pr.bytes) {
        pr.m = mmap(nullptr, pr.bytes, PROT_READ | PROT_WRITE, MAP_SHARED, fd, 0);
        if (pr.m != MAP_FAILED && expect && memcmp((const uint8_t*)pr.m + pr.sig_off, expect, sizeof(*expect)) != 0) {
          (void)munmap(pr.m, pr.bytes);           // another index: rebuild
          pr.m = MAP_FAILED;
        }
      }
      close(fd);
    }
  }
  pr.fill = (pr.m == MAP_FAILED);
  if (!pr.fill) return BANG_OK;
  const size_t avail = host_bytes_available();
  if (pr.bytes + ((size_t)8 << 30) > avail) {      // never push the host into the OOM killer for an optimisation
    bang_set_error("pull rows: %.1f GB do not fit the %.1f GB of host memory left", pr.bytes / 1e9, avail / 1e9);
    return BANG_ERR_NOMEM;
  }
  if (!pr.path.empty()) {
    pr.tmp = pr.path + ".tmp." + std::to_string((long)getpid());
    const int fd = open(pr.tmp.c_str(), O_RDWR | O_CREAT | O_EXCL, 0600);
    if (fd >= 0) {
      if (ftruncate(fd, (off_t)pr.bytes) == 0) pr.m = mmap(nullptr, pr.bytes, PROT_READ | PROT_WRITE, MAP_SHARED, fd, 0);
      close(fd);
      if (pr.m == MAP_FAILED) { (void)unlink(pr.tmp.c_str()); pr.tmp.clear(); }
    } else pr.tmp.clear();
  }
  if (pr.m == MAP_FAILED) {
    pr.m = mmap(nullptr, pr.bytes, PROT_READ | PROT_WRITE, MAP_PRIVATE | MAP_ANONYMOUS, -1, 0);
    if (pr.m == MAP_FAILED) { bang_set_error("pull rows: cannot map %.1f GB of host memory", pr.bytes / 1e9); return BANG_ERR_NOMEM; }
    (void)madvise(pr.m, pr.bytes, MADV_HUGEPAGE);
    pr.tmp.clear();
  }
  return BANG_OK;
}
// the rows are complete: signature written (built here) or checked (mapped), file published, mapping registered with the device
static int pull_rows_finish(bang_engine* e, PullRows& pr, const PullRowsSig& sig) {
  if (pr.fill) {
    memset((uint8_t*)pr.m + (size_t)e->N * 256, 0xFF, 4096);
    memcpy((uint8_t*)pr.m + pr.sig_off, &sig, sizeof(sig));
    if (!pr.tmp.empty() && rename(pr.tmp.c_str(), pr.path.c_str()) != 0) (void)unlink(pr.tmp.c_str());   // (the mapping stays valid either way)
    pr.tmp.clear();
  } else if (memcmp((const uint8_t*)pr.m + pr.sig_off, &sig, sizeof(sig)) != 0) {
    pull_rows_abandon(pr);
    (void)unlink(pr.path.c_str());                        // stale: rows of another graph (the caller rebuilds them)
    bang_set_error("pull rows: %s belonged to another index", pr.path.c_str());
    return BANG_ERR_STALE_ROWS;
  }
  if (hipHostRegister(pr.m, pr.bytes, hipHostRegisterMapped | hipHostRegisterPortable) != hipSuccess) {
    (void)hipGetLastError();
    pull_rows_abandon(pr);
    bang_set_error("pull rows: hipHostRegister of %.1f GB failed", pr.bytes / 1e9);
    return BANG_ERR_HIP;
  }
  void* dp = nullptr;
  if (hipHostGetDevicePointer(&dp, pr.m, 0) != hipSuccess) {
    (void)hipGetLastError();
    (void)hipHostUnregister(pr.m);
    pull_rows_abandon(pr);
    bang_set_error("pull rows: no device address for the registered rows");
    return BANG_ERR_HIP;
  }
  e->h_adj = (uint32_t*)pr.m; e->adj_bytes = pr.bytes; e->d_adj = (const uint32_t*)dp; e->pull = true;
  pr.m = MAP_FAILED;
  return BANG_OK;
}

// rows from a graph that is resident in host memory
static int build_pull_rows(bang_engine* e) {
  const size_t vb = vec_bytes(e);
  const int T = std::max(1, std::min(16, usable_cpus()));
  uint64_t h = 0;
  {
    std::vector<uint64_t> part((size_t)T, 0);
    std::vector<std::thread> th;
    for (int t = 0; t < T; ++t)
      th.emplace_back([&, t]() {
        const size_t a = (size_t)e->N * t / T, b = (size_t)e->N * (t + 1) / T;
        uint64_t acc = 0;
        for (size_t i = a; i < b; ++i) acc += sig_node(e, i, e->graph + i * e->entry_len + vb);
        part[(size_t)t] = acc;
      });
    for (auto& x : th) x.join();
    for (uint64_t v : part) h += v;
  }
  const PullRowsSig sig = sig_make(e, h);
  PullRows pr;
  BANG_TRY(pull_rows_open(e, pr, &sig));                 // (a rows file of another graph is not mapped: rebuilt below)
  if (pr.fill) {
    uint32_t* rows = (uint32_t*)pr.m;
    std::vector<std::thread> th;
    for (int t = 0; t < T; ++t)
      th.emplace_back([=]() {
        const size_t a = (size_t)e->N * t / T, b = (size_t)e->N * (t + 1) / T;
        for (size_t i = a; i < b; ++i) pull_row_from_entry(e, rows + i * 64, e->graph + i * e->entry_len + vb);
      });
    for (auto& x : th) x.join();
  }
  return pull_rows_finish(e, pr, sig);
}

// seed list [MEDOID, adj(MEDOID)...] (bang_init :467-489) and the medoid's vector (:492-501), from the medoid's graph entry
static int stage_medoid(bang_engine* e, const uint8_t* me) {
  uint32_t deg;
  memcpy(&deg, me + vec_bytes(e), 4);
  if (deg > e->R) deg = e->R;
  for (uint32_t i = 0; i < deg; ++i) {                 // cheap spot check of the adjacency layout: the medoid's neighbours
    uint32_t nb;
    memcpy(&nb, me + vec_bytes(e) + 4 + 4 * (size_t)i, 4);
    if (nb >= e->N) { bang_set_error("medoid neighbour %u = %u is out of range (N = %u): wrong data type or corrupt index", i, nb, e->N); return BANG_ERR_IO; }
  }
  std::vector<uint32_t> seed(2 + BANG_MAX_R + 1, 0);
  seed[0] = deg + 1;
  seed[1] = (uint32_t)e->medoid;
  memcpy(&seed[2], me + vec_bytes(e) + 4, (size_t)deg * 4);
  BANG_TRY(dmalloc(&e->d_seed, seed.size()));
  HIP_TRY(hipMemcpy(e->d_seed, seed.data(), seed.size() * 4, hipMemcpyHostToDevice));
  HIP_TRY(hipMalloc((void**)&e->d_medoid_vec, (vec_bytes(e) + 15) & ~(size_t)15));
  HIP_TRY(hipMemcpy(e->d_medoid_vec, me, vec_bytes(e), hipMemcpyHostToDevice));
  return BANG_OK;
}

// Can this index run in pull mode without a resident graph?  (vectors in HBM, rows in host memory, no walker option forced)
static bool stream_feasible(bang_engine* e, size_t hbm_reserve, std::string* why) {
  const size_t need = (size_t)e->N * vec_bytes(e);
  size_t free_b = 0, total_b = 0;
  (void)hipMemGetInfo(&free_b, &total_b);
  if (e->pull_opt == 0) { if (why) *why = "option pull = 0"; return false; }
  if (e->vectors_opt == 0) { if (why) *why = "option vectors = 0"; return false; }
  if (e->R > 64) { if (why) *why = "R > 64"; return false; }
  if (e->persistent == 0 || e->search_opt == 0) { if (why) *why = "the search kernel is switched off"; return false; }
  if (e->vectors_opt != 1 && need + hbm_reserve > free_b) { if (why) *why = "the full-precision vectors do not fit HBM"; return false; }
  return true;
}

// STREAMED load: every chunk of graph entries the source hands over is split on the spot -- vectors into HBM (through a pinned
// staging buffer), adjacency lists into the pull rows -- and dropped.  Host memory: the rows (N x 256 B) and one chunk.
static int stage_entries_streamed(bang_engine* e, bool retried = false) {
  const size_t vb = vec_bytes(e), N = e->N, el = e->entry_len;
  HIP_TRY(hipMalloc((void**)&e->d_vecs, N * vb + 256));
  PullRows pr;
  BANG_TRY(pull_rows_open(e, pr, nullptr));
  const size_t chunk = std::max<size_t>(1024, std::min<size_t>((size_t)1 << 20, ((size_t)512 << 20) / el));
  void* bufp = nullptr;
  if (posix_memalign(&bufp, 4096, chunk * el) != 0) { pull_rows_abandon(pr); bang_set_error("streamed load: no memory for a chunk"); return BANG_ERR_NOMEM; }
  uint8_t* buf = (uint8_t*)bufp;
  uint8_t* stage[2] = {nullptr, nullptr};
  hipEvent_t ev[2] = {nullptr, nullptr};
  std::vector<uint8_t> medoid_entry(el);
  int rc = BANG_OK;
  auto cleanup = [&]() {
    free(buf);
    for (int i = 0; i < 2; ++i) { if (stage[i]) (void)hipHostFree(stage[i]); if (ev[i]) (void)hipEventDestroy(ev[i]); }
  };
  for (int b = 0; b < 2 && rc == BANG_OK; ++b)
    if (hipHostMalloc((void**)&stage[b], chunk * vb, hipHostMallocDefault) != hipSuccess || hipEventCreate(&ev[b]) != hipSuccess) {
      (void)hipGetLastError(); bang_set_error("streamed load: no pinned staging buffer"); rc = BANG_ERR_HIP;
    }
  uint64_t h = 0;
  uint32_t* rows = (uint32_t*)pr.m;
  const int T = std::max(1, std::min(16, usable_cpus()));
  std::vector<uint64_t> part((size_t)T, 0);
  int b = 0;
  for (size_t first = 0; first < N && rc == BANG_OK; first += chunk, b ^= 1) {
    const size_t n = std::min(chunk, N - first);
    if (e->entry_fn(e->entry_ctx, first, n, buf) != 0) { bang_set_error("streamed load: the entry source failed at node %zu", first); rc = BANG_ERR_IO; break; }
    if (e->medoid >= first && e->medoid < first + n) memcpy(medoid_entry.data(), buf + (e->medoid - first) * el, el);
    if (hipEventSynchronize(ev[b]) != hipSuccess) { bang_set_error("streamed load: event"); rc = BANG_ERR_HIP; break; }   // the previous copy out of this buffer is done
    {
      uint8_t* st = stage[b];
      const bool fill = pr.fill;
      uint64_t* part_p = part.data();
      std::vector<std::thread> th;
      for (int t = 0; t < T; ++t)
        th.emplace_back([=]() {
          const size_t a = n * t / T, z = n * (t + 1) / T;
          uint64_t acc = 0;
          for (size_t i = a; i < z; ++i) {
            const uint8_t* ent = buf + i * el;
            memcpy(st + i * vb, ent, vb);
            acc += sig_node(e, first + i, ent + vb);
            if (fill) pull_row_from_entry(e, rows + (first + i) * 64, ent + vb);
          }
          part_p[t] += acc;
        });
      for (auto& x : th) x.join();
    }
    if (hipMemcpyAsync(e->d_vecs + first * vb, stage[b], n * vb, hipMemcpyHostToDevice, nullptr) != hipSuccess ||
        hipEventRecord(ev[b], nullptr) != hipSuccess) { bang_set_error("streamed load: vector upload failed"); rc = BANG_ERR_HIP; break; }
  }
  if (rc == BANG_OK && hipDeviceSynchronize() != hipSuccess) { bang_set_error("streamed load: sync"); rc = BANG_ERR_HIP; }
  cleanup();
  if (rc != BANG_OK) { (void)hipGetLastError(); pull_rows_abandon(pr); return rc; }
  for (uint64_t v : part) h += v;
  {
    const int frc = pull_rows_finish(e, pr, sig_make(e, h));
    if (frc == BANG_ERR_STALE_ROWS && !retried) {
      // a rows file of ANOTHER graph sat under this name (it has been removed): the entries pass through once more to build ours
      dfree(e->d_vecs);
      return stage_entries_streamed(e, true);
    }
    if (frc != BANG_OK) return frc == BANG_ERR_STALE_ROWS ? BANG_ERR_IO : frc;
  }
  e->vec_on_device = true;
  BANG_TRY(stage_medoid(e, medoid_entry.data()));
  e->graph_streamed = true;
  return BANG_OK;
}

struct FileEntrySource { int fd; uint64_t entry_len; };
static int file_entry_source(void* ctx, uint64_t first, uint64_t count, uint8_t* dst) {
  const FileEntrySource* f = (const FileEntrySource*)ctx;
  size_t left = (size_t)(count * f->entry_len);
  off_t off = (off_t)(first * f->entry_len);
  while (left) {
    const ssize_t r = pread(f->fd, dst, left, off);
    if (r <= 0) return -1;
    dst += r; off += r; left -= (size_t)r;
  }
  return 0;
}

// DiskANN's own `_disk.index` as an entry source: what the reference's bang_preprocess.py does up front (:28-116) happens while the
// entries stream through -- sector 0 is the header, every following 4096-byte sector holds nnodes_per_sector records of
// max_node_len bytes [T vec[D]][u32 degree][u32 id x R]; a record's ids are copied in ascending order (:102-104).
struct DiskAnnSource {
  int fd = -1;
  uint64_t npts = 0, ndims = 0, medoid = 0, max_node_len = 0, per_sector = 0;
  uint64_t vec_bytes = 0, R = 0;
  std::vector<uint8_t> buf;
};
static int diskann_entry_source(void* ctx, uint64_t first, uint64_t count, uint8_t* dst) {
  DiskAnnSource* f = (DiskAnnSource*)ctx;
  const uint64_t SECTOR = 4096, el = f->max_node_len;
  uint64_t done = 0;
  while (done < count) {
    const uint64_t node = first + done, sec = node / f->per_sector, in_sec = node % f->per_sector;
    const uint64_t secs = std::min<uint64_t>(2048, (count - done + in_sec + f->per_sector - 1) / f->per_sector);     // up to 8 MB per read
    f->buf.resize((size_t)(secs * SECTOR));
    size_t left = f->buf.size();
    off_t off = (off_t)((1 + sec) * SECTOR);
    uint8_t* b = f->buf.data();
    while (left) {
      const ssize_t r = pread(f->fd, b, left, off);
      if (r < 0) return -1;
      if (r == 0) { memset(b, 0, left); break; }               // (a short last sector)
      b += r; off += r; left -= (size_t)r;
    }
    for (uint64_t s = 0; s < secs && done < count; ++s)
      for (uint64_t k = (s == 0 ? in_sec : 0); k < f->per_sector && done < count; ++k, ++done) {
        const uint8_t* rec = f->buf.data() + s * SECTOR + k * el;
        uint8_t* out = dst + done * el;
        memcpy(out, rec, (size_t)el);
        uint32_t deg;
        memcpy(&deg, rec + f->vec_bytes, 4);
        if (deg == 0 || deg > f->R) return -2;                  // bang_preprocess.py:91-94
        uint32_t ids[BANG_MAX_R];
        memcpy(ids, rec + f->vec_bytes + 4, (size_t)deg * 4);
        std::sort(ids, ids + deg);
        memcpy(out + f->vec_bytes + 4, ids, (size_t)deg * 4);
      }
  }
  return 0;
}

// `_disk.index` header (bang_preprocess.py:28-64): skip 8 B; u64 npts, ndims, medoid, max_node_len, nnodes_per_sector
static int diskann_open(DiskAnnSource& d, const char* path, size_t tsize) {
  d.fd = open(path, O_RDONLY);
  if (d.fd < 0) { bang_set_error("cannot open %s", path); return BANG_ERR_IO; }
  uint64_t h[5];
  if (pread(d.fd, h, 40, 8) != 40 || h[0] == 0 || h[0] > 0xFFFFFFFFull || h[4] == 0) { bang_set_error("bad _disk.index header"); return BANG_ERR_IO; }
  d.npts = h[0]; d.ndims = h[1]; d.medoid = h[2]; d.max_node_len = h[3]; d.per_sector = h[4];
  d.vec_bytes = d.ndims * tsize;
  if (d.max_node_len < d.vec_bytes + 8 || (d.max_node_len - d.vec_bytes - 4) % 4 != 0 || d.per_sector * d.max_node_len > 4096) {
    bang_set_error("_disk.index: record length %llu does not fit D=%llu elements of %zu B (wrong data type?)", (unsigned long long)d.max_node_len, (unsigned long long)d.ndims, tsize);
    return BANG_ERR_IO;
  }
  d.R = (d.max_node_len - d.vec_bytes - 4) / 4;
  if (d.R > BANG_MAX_R) { bang_set_error("_disk.index: degree bound R=%llu unsupported (max %d)", (unsigned long long)d.R, BANG_MAX_R); return BANG_ERR_UNSUPPORTED; }
  return BANG_OK;
}

extern "C" int bang_convert_diskann_index(const char* index_path, const char* out_prefix, int dtype) {
  if (!index_path || !out_prefix || dtype < 0 || dtype > 2) return BANG_ERR_ARG;
  DiskAnnSource d;
  const size_t tsize = (dtype == BANG_F32) ? 4 : 1;
  int rc = diskann_open(d, index_path, tsize);
  FILE* fb = nullptr;
  FILE* fm = nullptr;
  if (rc == BANG_OK) {
    const std::string pfx(out_prefix);
    fb = fopen((pfx + "_disk.bin").c_str(), "wb");
    fm = fopen((pfx + "_disk_metadata.bin").c_str(), "wb");
    if (!fb || !fm) { bang_set_error("cannot create %s_disk.bin / _disk_metadata.bin", out_prefix); rc = BANG_ERR_IO; }
  }
  if (rc == BANG_OK) {
    const size_t chunk = std::max<size_t>(1024, ((size_t)64 << 20) / d.max_node_len);
    std::vector<uint8_t> buf(chunk * d.max_node_len);
    for (uint64_t first = 0; first < d.npts && rc == BANG_OK; first += chunk) {
      const uint64_t n = std::min<uint64_t>(chunk, d.npts - first);
      const int sr = diskann_entry_source(&d, first, n, buf.data());
      if (sr != 0) { bang_set_error(sr == -2 ? "bad degree in index (bang_preprocess.py:91-94)" : "read error in %s", index_path); rc = BANG_ERR_IO; break; }
      if (fwrite(buf.data(), d.max_node_len, n, fb) != n) { bang_set_error("short write"); rc = BANG_ERR_IO; }
    }
  }
  if (rc == BANG_OK) {
    static const int32_t code_of[3] = {1 /*BANG_U8*/, 0 /*BANG_I8*/, 2 /*BANG_F32*/};      // bang_preprocess.py:12-13
    uint8_t md[32];
    const uint32_t D = (uint32_t)d.ndims, R = (uint32_t)d.R, N = (uint32_t)d.npts;
    memcpy(md, &d.medoid, 8); memcpy(md + 8, &d.max_node_len, 8); memcpy(md + 16, &code_of[dtype], 4);
    memcpy(md + 20, &D, 4); memcpy(md + 24, &R, 4); memcpy(md + 28, &N, 4);
    if (fwrite(md, 32, 1, fm) != 1) { bang_set_error("short write"); rc = BANG_ERR_IO; }
  }
  if (fb) fclose(fb);
  if (fm) fclose(fm);
  if (d.fd >= 0) close(d.fd);
  return rc;
}

// a resident private copy of the graph, filled from the entry source (placements that walk or upload the whole graph)
static int materialize_graph(bang_engine* e) {
  const size_t gsize = (size_t)e->N * e->entry_len;
  void* gp = nullptr;
  if (posix_memalign(&gp, (size_t)2 << 20, gsize) != 0 || !gp) { bang_set_error("malloc(%zu) failed", gsize); return BANG_ERR_NOMEM; }
  (void)madvise(gp, gsize, MADV_HUGEPAGE);
  e->graph_owned = (uint8_t*)gp;
  const size_t chunk = std::max<size_t>(1024, ((size_t)256 << 20) / e->entry_len);
  for (size_t first = 0; first < e->N; first += chunk) {
    const size_t n = std::min(chunk, (size_t)e->N - first);
    if (e->entry_fn(e->entry_ctx, first, n, e->graph_owned + first * e->entry_len) != 0) {
      bang_set_error("the entry source failed at node %zu", first);
      return BANG_ERR_IO;
    }
  }
  e->graph = e->graph_owned;
  return BANG_OK;
}

// The graph file in host memory, MAPPED shared and read-only, not copied: the ranks of a multi-GPU job (one process per GPU) walk
// ONE copy in the page cache instead of one 388 GB copy each (:312-328).  MAP_POPULATE reads the file in now, as the reference's
// fread does.  BANG_GRAPH_MMAP=0 (or a failing mmap) falls back to a private copy, which can ask for transparent huge pages.
static int map_graph_file(bang_engine* e) {
  if (e->graph) return BANG_OK;
  if (e->graph_path.empty()) { bang_set_error("the graph entries were streamed at load time and are not resident: only the pull mode can run"); return BANG_ERR_UNSUPPORTED; }
  const int fd = open(e->graph_path.c_str(), O_RDONLY);
  if (fd < 0) { printf("Error.. Could not open the Graph Index File: %s\n", e->graph_path.c_str()); bang_set_error("cannot open %s", e->graph_path.c_str()); return BANG_ERR_IO; }
  struct stat st;
  if (fstat(fd, &st) != 0 || (size_t)st.st_size < (size_t)e->N * e->entry_len) { close(fd); bang_set_error("graph file too small"); return BANG_ERR_IO; }
  const size_t gsize = (size_t)st.st_size;
  static const bool want_map = !(getenv("BANG_GRAPH_MMAP") && atoi(getenv("BANG_GRAPH_MMAP")) == 0);
  void* mp = want_map ? mmap(nullptr, gsize, PROT_READ, MAP_SHARED | MAP_POPULATE, fd, 0) : MAP_FAILED;
  int rc = BANG_OK;
  if (mp != MAP_FAILED) {
    e->graph_map = mp; e->graph_map_len = gsize;
    (void)madvise(mp, gsize, MADV_RANDOM);
    e->graph = (const uint8_t*)mp;
  } else {
    void* gp = nullptr;
    if (posix_memalign(&gp, (size_t)2 << 20, gsize) != 0) gp = nullptr;
    e->graph_owned = (uint8_t*)gp;
    if (gp) (void)madvise(gp, gsize, MADV_HUGEPAGE);
    if (!e->graph_owned) { printf("Error.. Malloc failed for Graph Index.\n"); bang_set_error("malloc(%zu) failed", gsize); rc = BANG_ERR_NOMEM; }
    else {
      FileEntrySource src{fd, e->entry_len};
      if (file_entry_source(&src, 0, e->N, e->graph_owned) != 0) { bang_set_error("short graph file"); rc = BANG_ERR_IO; free(e->graph_owned); e->graph_owned = nullptr; }
      else e->graph = e->graph_owned;
    }
  }
  close(fd);
  return rc;
}

int upload_index(bang_engine* e, const uint8_t* h_codes, const void* d_codes_ext, const float* pivots,
                 const float* centroid, const uint32_t* chunk_off) {
  const uint32_t D = e->D, m = e->m;
  if (e->R == 0 || e->R > BANG_MAX_R) {             // assert(R == MAX_R), bang_search.cu:190 (relaxed to R <= 64)
    bang_set_error("graph degree bound R=%u unsupported (max %d)", e->R, BANG_MAX_R);
    return BANG_ERR_UNSUPPORTED;
  }
  if (e->entry_len < (uint64_t)D * e->tsize + 4 + 4ull * e->R) {
    bang_set_error("index entry length %llu too small for D=%u R=%u", (unsigned long long)e->entry_len, D, e->R);
    return BANG_ERR_IO;
  }
  if (e->medoid >= e->N) { bang_set_error("medoid out of range"); return BANG_ERR_IO; }
  // PQ codes (+256 B slack: the distance kernel over-reads up to 19 B past a row)
  const size_t code_bytes = (size_t)e->N * m;
  if (d_codes_ext) {
    e->d_codes = (uint8_t*)d_codes_ext;
    e->codes_owned = false;
  } else {
    HIP_TRY(hipMalloc((void**)&e->d_codes, code_bytes + 256));
    e->codes_owned = true;
    HIP_TRY(hipMemset(e->d_codes + code_bytes, 0, 256));
    const size_t step = (size_t)1 << 30;
    for (size_t off = 0; off < code_bytes; off += step)
      HIP_TRY(hipMemcpy(e->d_codes + off, h_codes + off, std::min(step, code_bytes - off), hipMemcpyHostToDevice));
  }
  // Placement.  HBM left after the PQ codes decides: the whole graph (adjacency + vectors) if it fits with 16 GB to spare for
  // the per-batch state -> no host in the loop at all; else the graph stays in host RAM (C++ walker) and, if THEY fit, a packed
  // copy of the full-precision vectors goes to HBM for the re-rank (128 GB for 1e9 x 128 uint8 next to 70 GB of codes).
  const size_t hbm_reserve = (size_t)16 << 30;
  if (e->graph_mode == BANG_GRAPH_AUTO) {
    size_t free_b = 0, total_b = 0;
    (void)hipMemGetInfo(&free_b, &total_b);
    const size_t gbytes = (size_t)e->N * e->entry_len + 256;
    e->graph_mode = (gbytes + hbm_reserve <= free_b) ? BANG_GRAPH_DEVICE : BANG_GRAPH_HOST;
    if (getenv("BANG_DEBUG")) fprintf(stderr, "[bang] graph=auto -> %s (graph %.1f GB, free HBM %.1f GB)\n",
                                      e->graph_mode == BANG_GRAPH_DEVICE ? "device" : "host", gbytes / 1e9, free_b / 1e9);
  }
  // pivots: transposed [D][256] for K1 (bang_search.cu:281-285) and chunk-packed for the LDS kernel
  std::vector<float> pt((size_t)D * 256);
  for (uint32_t row = 0; row < 256; ++row)
    for (uint32_t col = 0; col < D; ++col) pt[(size_t)col * 256 + row] = pivots[(size_t)row * D + col];
  BANG_TRY(dmalloc(&e->d_pivots_T, pt.size()));
  HIP_TRY(hipMemcpy(e->d_pivots_T, pt.data(), pt.size() * 4, hipMemcpyHostToDevice));
  BANG_TRY(dmalloc(&e->d_centroid, D));
  HIP_TRY(hipMemcpy(e->d_centroid, centroid, (size_t)D * 4, hipMemcpyHostToDevice));
  BANG_TRY(dmalloc(&e->d_chunk_off, m + 1));
  HIP_TRY(hipMemcpy(e->d_chunk_off, chunk_off, (size_t)(m + 1) * 4, hipMemcpyHostToDevice));
  uint32_t psz = 0, mp = m;
  BANG_TRY(bang_pq_layout(chunk_off, D, m, &psz, &mp));
  if (e->pq_mode == 1) { psz = 0; mp = m; }
  e->psz = psz;
  e->mp = mp;
  // 2-float layouts whose chunks are 2,..,2,1,..,1 dims wide also get the exact-size table (if a kernel instance exists for it);
  // bang_alloc picks it when only it leaves room for the persistent kernel's merge scratch at the requested L
  e->pq_nhi = 0; e->pq_nhi_avail = 0;
  if (psz == 2 && e->pq_ragged) {
    uint32_t nhi = 0;
    uint64_t nfl = 0;
    BANG_TRY(bang_pack_pivots_ragged(nullptr, chunk_off, D, m, mp, &nhi, nullptr, &nfl));
    if (nhi && bang_ragged_supported(psz, mp, nhi, m)) {
      std::vector<float> packed((size_t)nfl);
      BANG_TRY(bang_pack_pivots_ragged(pivots, chunk_off, D, m, mp, &nhi, packed.data(), &nfl));
      BANG_TRY(dmalloc(&e->d_pivots_ragged, packed.size()));
      HIP_TRY(hipMemcpy(e->d_pivots_ragged, packed.data(), packed.size() * 4, hipMemcpyHostToDevice));
      e->pq_nhi_avail = nhi;
    }
  }
  if (psz) {
    std::vector<float> packed((size_t)mp * 256 * psz);
    BANG_TRY(bang_pack_pivots(pivots, chunk_off, D, m, psz, mp, packed.data()));
    BANG_TRY(dmalloc(&e->d_pivots_packed, packed.size()));
    HIP_TRY(hipMemcpy(e->d_pivots_packed, packed.data(), packed.size() * 4, hipMemcpyHostToDevice));
  }
  // ---- everything that depends on the graph entries
  e->vec_on_device = false;
  e->pull = false;
  e->graph_streamed = false;
  if (!e->graph) {
    // no resident graph: a streamed load (the caller's entry source), or a graph FILE that has not been touched yet.  If the pull
    // mode applies the entries only pass through (vectors -> HBM, adjacency -> pull rows); else a file is mapped as before.
    std::string why;
    const bool feasible = e->graph_mode != BANG_GRAPH_DEVICE && stream_feasible(e, hbm_reserve, &why);
    const bool from_file = (e->entry_fn == nullptr);
    static const bool want_stream = !(getenv("BANG_STREAM_LOAD") && atoi(getenv("BANG_STREAM_LOAD")) == 0);
    if (!from_file && !feasible && !e->entry_src_rereadable) {
      bang_set_error("a streamed load runs in pull mode on the host placement only: %s", why.empty() ? "option graph = device" : why.c_str());
      return BANG_ERR_UNSUPPORTED;
    }
    if (feasible && (!from_file || want_stream)) {
      FileEntrySource fsrc{-1, e->entry_len};
      if (from_file) {
        fsrc.fd = open(e->graph_path.c_str(), O_RDONLY);
        if (fsrc.fd < 0) { printf("Error.. Could not open the Graph Index File: %s\n", e->graph_path.c_str()); bang_set_error("cannot open %s", e->graph_path.c_str()); return BANG_ERR_IO; }
        (void)posix_fadvise(fsrc.fd, 0, 0, POSIX_FADV_SEQUENTIAL);
        e->entry_fn = file_entry_source; e->entry_ctx = &fsrc;
      }
      const int rc = stage_entries_streamed(e);
      if (from_file) { close(fsrc.fd); e->entry_fn = nullptr; e->entry_ctx = nullptr; }
      if (rc == BANG_OK) { e->loaded = true; return BANG_OK; }
      if (!((from_file || e->entry_src_rereadable) && rc == BANG_ERR_NOMEM && e->pull_opt != 1)) return rc;
      dfree(e->d_vecs);                               // the rows do not fit this host: keep the graph resident, the walker serves it
      e->vec_on_device = false;
    }
    if (e->entry_fn) BANG_TRY(materialize_graph(e));  // (a DiskANN `_disk.index`: converted into a private resident copy)
    else BANG_TRY(map_graph_file(e));
  }
  BANG_TRY(stage_medoid(e, e->graph + e->medoid * e->entry_len));
  e->vec_on_device = false;
  if (e->graph_mode != BANG_GRAPH_DEVICE && e->vectors_opt != 0) {
    // 288 GB of HBM hold the full-precision vectors of a billion uint8 points (128 GB) next to their PQ codes (70 GB): keep a
    // packed copy on the device for the re-rank, so that the walker ships adjacency rows only (a third less PCIe traffic per
    // expanded node).  "auto" = whenever the copy fits the free HBM with 16 GB to spare.
    const size_t vb = vec_bytes(e), need = (size_t)e->N * vb;
    size_t free_b = 0, total_b = 0;
    (void)hipMemGetInfo(&free_b, &total_b);
    if (e->vectors_opt == 1 || need + hbm_reserve <= free_b) {
      HIP_TRY(hipMalloc((void**)&e->d_vecs, need + 256));
      const size_t rows_per = std::max<size_t>(1, ((size_t)32 << 20) / vb);
      uint8_t* stage[2] = {nullptr, nullptr};
      hipEvent_t ev[2];
      for (int b = 0; b < 2; ++b) { HIP_TRY(hipHostMalloc((void**)&stage[b], rows_per * vb, hipHostMallocDefault)); HIP_TRY(hipEventCreate(&ev[b])); }
      int b = 0;
      for (size_t r0 = 0; r0 < e->N; r0 += rows_per, b ^= 1) {
        const size_t nr = std::min(rows_per, (size_t)e->N - r0);
        HIP_TRY(hipEventSynchronize(ev[b]));               // the previous copy out of this buffer has finished
        for (size_t r = 0; r < nr; ++r) memcpy(stage[b] + r * vb, e->graph + (r0 + r) * e->entry_len, vb);
        HIP_TRY(hipMemcpyAsync(e->d_vecs + r0 * vb, stage[b], nr * vb, hipMemcpyHostToDevice, nullptr));
        HIP_TRY(hipEventRecord(ev[b], nullptr));
      }
      HIP_TRY(hipDeviceSynchronize());
      for (int i = 0; i < 2; ++i) { (void)hipHostFree(stage[i]); (void)hipEventDestroy(ev[i]); }
      e->vec_on_device = true;
    }
  }
  if (e->graph_mode != BANG_GRAPH_DEVICE && e->pull_opt != 0) {
    // pull needs the re-rank's vectors in HBM (nothing walks the graph entries any more) and rows of <= 64 ids
    if (e->vec_on_device && e->R <= 64) {
      int rc = build_pull_rows(e);
      if (rc == BANG_ERR_STALE_ROWS) rc = build_pull_rows(e);     // (another rank swapped the file between open and check)
      if (rc == BANG_ERR_STALE_ROWS) rc = BANG_ERR_IO;
      if (rc != BANG_OK && e->pull_opt == 1) return rc;            // asked for explicitly: report; auto: the walker serves the graph
    } else if (e->pull_opt == 1) {
      bang_set_error("option pull = 1 needs the full-precision vectors resident in HBM (option vectors) and R <= 64");
      return BANG_ERR_ARG;
    }
  }
  if (e->pull && e->graph_map)
    // nothing reads the mapped graph file while the kernel pulls its rows: let the page cache have the pages back (a later
    // change of the loop form -- "persistent" = 0 -- simply faults them in again)
    (void)madvise(e->graph_map, e->graph_map_len, MADV_DONTNEED);
  if (e->graph_mode == BANG_GRAPH_DEVICE) {
    const size_t gbytes = (size_t)e->N * e->entry_len;
    HIP_TRY(hipMalloc((void**)&e->d_graph, gbytes + 256));
    const size_t step = (size_t)1 << 30;
    for (size_t off = 0; off < gbytes; off += step)
      HIP_TRY(hipMemcpy(e->d_graph + off, e->graph + off, std::min(step, gbytes - off), hipMemcpyHostToDevice));
  }
  e->loaded = true;
  return BANG_OK;
}

void unload_index(bang_engine* e) {
  if (e->codes_owned) dfree(e->d_codes);
  e->d_codes = nullptr;
  dfree(e->d_pivots_T);
  dfree(e->d_pivots_packed);
  dfree(e->d_pivots_ragged);
  e->pq_nhi = e->pq_nhi_avail = 0;
  dfree(e->d_centroid);
  dfree(e->d_chunk_off);
  dfree(e->d_seed);
  dfree(e->d_medoid_vec);
  dfree(e->d_graph);
  dfree(e->d_vecs);
  e->vec_on_device = false;
  if (e->h_adj) { (void)hipHostUnregister(e->h_adj); (void)munmap(e->h_adj, e->adj_bytes); }
  e->h_adj = nullptr; e->d_adj = nullptr; e->adj_bytes = 0; e->pull = false;
  e->graph_path.clear(); e->graph_streamed = false; e->entry_fn = nullptr; e->entry_ctx = nullptr;
  e->rows_key.clear();
  free(e->graph_owned);
  e->graph_owned = nullptr;
  if (e->graph_map) (void)munmap(e->graph_map, e->graph_map_len);
  e->graph_map = nullptr; e->graph_map_len = 0;
  e->graph = nullptr;
  e->loaded = false;
  e->graph_mode = e->graph_opt;
}

void stop_threads(bang_engine* e);

// CPUs of the NUMA node the GPU hangs off, intersected with what this process may use.  The walker threads read the host graph
// and store into the GPU's BAR: on the far socket both cross the inter-socket fabric.
bool gpu_numa_cpus(int device, cpu_set_t* out, int* node_out) {
  char bdf[64] = "";
  if (hipDeviceGetPCIBusId(bdf, (int)sizeof(bdf), device) != hipSuccess) return false;
  for (char* c = bdf; *c; ++c) *c = (char)tolower(*c);
  char path[256];
  snprintf(path, sizeof(path), "/sys/bus/pci/devices/%s/numa_node", bdf);
  int node = -1;
  if (FILE* f = fopen(path, "r")) { if (fscanf(f, "%d", &node) != 1) node = -1; fclose(f); }
  if (node < 0) return false;
  snprintf(path, sizeof(path), "/sys/devices/system/node/node%d/cpulist", node);
  FILE* f = fopen(path, "r");
  if (!f) return false;
  char list[4096] = "";
  if (!fgets(list, sizeof(list), f)) { fclose(f); return false; }
  fclose(f);
  cpu_set_t allowed, want;
  CPU_ZERO(&want);
  if (sched_getaffinity(0, sizeof(allowed), &allowed) != 0) return false;
  for (char* tok = strtok(list, ",\n"); tok; tok = strtok(nullptr, ",\n")) {
    int a = 0, b = 0;
    const int n = sscanf(tok, "%d-%d", &a, &b);
    if (n == 1) b = a;
    if (n >= 1) for (int c = a; c <= b && c < CPU_SETSIZE; ++c) if (CPU_ISSET(c, &allowed)) CPU_SET(c, &want);
  }
  if (CPU_COUNT(&want) == 0) return false;
  *out = want;
  *node_out = node;
  return true;
}

// one CPU per physical core among `cpus` (SMT siblings share a core's pipelines: two spinning walker threads on one core halve each other)
std::vector<int> distinct_cores(const cpu_set_t& cpus) {
  std::vector<int> out;
  std::vector<long long> seen;
  for (int c = 0; c < CPU_SETSIZE; ++c) {
    if (!CPU_ISSET(c, &cpus)) continue;
    char path[128];
    int core = -1, pkg = -1;
    snprintf(path, sizeof(path), "/sys/devices/system/cpu/cpu%d/topology/core_id", c);
    if (FILE* f = fopen(path, "r")) { if (fscanf(f, "%d", &core) != 1) core = -1; fclose(f); }
    snprintf(path, sizeof(path), "/sys/devices/system/cpu/cpu%d/topology/physical_package_id", c);
    if (FILE* f = fopen(path, "r")) { if (fscanf(f, "%d", &pkg) != 1) pkg = -1; fclose(f); }
    const long long key = core < 0 ? -(long long)c - 1 : ((long long)pkg << 32) | (unsigned)core;
    if (std::find(seen.begin(), seen.end(), key) != seen.end()) continue;
    seen.push_back(key);
    out.push_back(c);
  }
  return out;
}

void pin_walker_thread(const bang_engine* e, int index) {
  if (!e->numa_on) return;
  if (!e->numa_cores.empty()) {
    cpu_set_t one;
    CPU_ZERO(&one);
    CPU_SET(e->numa_cores[(size_t)index % e->numa_cores.size()], &one);
    if (sched_setaffinity(0, sizeof(one), &one) == 0) return;
  }
  (void)sched_setaffinity(0, sizeof(e->numa_cpus), &e->numa_cpus);
}

void free_batch(bang_engine* e) {
  stop_threads(e);
  for (auto& lp : e->lanes) {
    Lane& ln = *lp;
    if (ln.s_main) (void)hipStreamDestroy(ln.s_main);
    if (ln.s_fp) (void)hipStreamDestroy(ln.s_fp);
    if (ln.ev_front) (void)hipEventDestroy(ln.ev_front);
    if (ln.ev_fp) (void)hipEventDestroy(ln.ev_fp);
    if (ln.d_ktime) (void)hipFree(ln.d_ktime);
    ln.d_ktime = nullptr;
    ln.pw_groups = 0;
    if (ln.d_pcnt) (void)hipFree(ln.d_pcnt);
    ln.d_pcnt = nullptr;
    for (int b = 0; b < 2; ++b) {
      if (ln.qmap_host[b]) { if (ln.qmap_is_device) (void)hipFree(ln.qmap_host[b]); else (void)hipHostFree(ln.qmap_host[b]); }
      ln.qmap_host[b] = ln.qmap_dev[b] = nullptr;
    }
  }
  e->lanes.clear();
  dfree(e->d_queries); dfree(e->d_qc); dfree(e->d_lut); dfree(e->d_bloom); dfree(e->d_nbrs);
  dfree(e->d_dist); dfree(e->d_cnt); dfree(e->d_wl_ids); dfree(e->d_wl_dist); dfree(e->d_wl_vis); dfree(e->d_wl_cnt);
  dfree(e->d_mark); dfree(e->d_parents_dev); dfree(e->d_cand_ids); dfree(e->d_cand_row); dfree(e->d_cand_cnt);
  dfree(e->d_active); dfree(e->d_qstats); dfree(e->d_qskip); dfree(e->d_fp); dfree(e->d_results);
  e->d_ids_out = nullptr; e->d_dists_out = nullptr; e->d_qiters = nullptr;             // (inside d_results)
  if (e->h_results) { (void)hipHostFree(e->h_results); e->h_results = nullptr; }
  dfree(e->d_done_count); dfree(e->d_stage); dfree(e->d_srows); dfree(e->d_sctl);
  if (e->h_parents) (void)hipHostFree(e->h_parents);
  if (e->h_pub_q) (void)hipHostFree(e->h_pub_q);
  if (e->h_pub_c) (void)hipHostFree(e->h_pub_c);
  e->h_pub_q = e->h_pub_c = e->d_pub_q = e->d_pub_c = nullptr;
  if (e->h_fp) (void)hipHostFree(e->h_fp);
  if (e->h_stage) (void)hipHostFree(e->h_stage);
  if (e->h_done) (void)hipHostFree(e->h_done);
  e->h_parents = nullptr; e->d_parents_map = nullptr; e->h_fp = nullptr; e->h_stage = nullptr; e->h_stage_dev = nullptr;
  e->h_done = nullptr; e->h_done_dev = nullptr;
  e->allocated = false;
  e->inited = false;
}

// ---- file loading (bang_search.cu:138-362) ----
bool read_exact(FILE* f, void* dst, size_t n) { return fread(dst, 1, n, f) == n; }

int load_files(bang_engine* e, const char* prefix) {
  const std::string p(prefix);
  {
    const size_t sl = p.find_last_of('/');
    e->rows_key = (sl == std::string::npos) ? p : p.substr(sl + 1);
  }
  const std::string f_piv = p + "_pq_pivots.bin", f_cmp = p + "_pq_compressed.bin", f_graph = p + "_disk.bin",
                    f_meta = p + "_disk_metadata.bin";                       // suffixes :39-45
  FILE* fp = fopen(f_piv.c_str(), "rb");
  if (!fp) { printf("Error.. Could not open the PQ Pivots File: %s\n", f_piv.c_str()); bang_set_error("cannot open %s", f_piv.c_str()); return BANG_ERR_IO; }
  FILE* fc = fopen(f_cmp.c_str(), "rb");
  if (!fc) { fclose(fp); printf("Error.. Could not open the PQ Compressed Vectors File: %s\n", f_cmp.c_str()); bang_set_error("cannot open %s", f_cmp.c_str()); return BANG_ERR_IO; }
  FILE* fg = fopen(f_graph.c_str(), "rb");
  FILE* fm = fg ? fopen(f_meta.c_str(), "rb") : nullptr;
  DiskAnnSource dsrc;
  if (!fg) {
    // no converted graph: DiskANN's own `<p>_disk.index` is read directly (what bang_preprocess.py would have written is produced
    // while the entries stream through)
    const std::string f_index = p + "_disk.index";
    const int orc = diskann_open(dsrc, f_index.c_str(), e->tsize);
    if (orc != BANG_OK) {
      if (dsrc.fd < 0) { printf("Error.. Could not open the Graph Index File: %s\n", f_graph.c_str()); bang_set_error("cannot open %s (nor %s)", f_graph.c_str(), f_index.c_str()); }
      else close(dsrc.fd);
      fclose(fp); fclose(fc);
      return orc;
    }
  } else if (!fm) { fclose(fp); fclose(fc); fclose(fg); printf("Error.. Could not open the Metadata File: %s\n", f_meta.c_str()); bang_set_error("cannot open %s", f_meta.c_str()); return BANG_ERR_IO; }
  int rc = BANG_OK;
  std::vector<uint8_t> codes;
  std::vector<float> pivots, centroid;
  std::vector<uint32_t> chunk_off;
  do {
    // 32-byte packed metadata {u64 medoid, u64 entryLen, i32 dtype, u32 D, u32 R, u32 N} (bang_search.cuh:42-50)
    uint8_t md[32];
    int32_t md_dtype = -1;
    if (!fg) {
      e->medoid = dsrc.medoid; e->entry_len = dsrc.max_node_len; e->D = (uint32_t)dsrc.ndims; e->R = (uint32_t)dsrc.R; e->N = (uint32_t)dsrc.npts;
    } else {
    if (!read_exact(fm, md, 32)) { bang_set_error("short metadata file"); rc = BANG_ERR_IO; break; }
    memcpy(&e->medoid, md, 8);
    memcpy(&e->entry_len, md + 8, 8);
    memcpy(&md_dtype, md + 16, 4);
    memcpy(&e->D, md + 20, 4);
    memcpy(&e->R, md + 24, 4);
    memcpy(&e->N, md + 28, 4);
    }
    // The reference never looks at uDatatype (bang_search.cu:180-188) and a wrong <data type> argument makes it read vectors and
    // adjacency lists at the wrong offsets.  bang_preprocess.py:12-13 writes 0 = int8, 1 = uint8, 2 = float: refuse an index whose
    // code or entry length contradicts the element type of this engine.
    {
      static const int code_of[3] = {1 /*BANG_U8*/, 0 /*BANG_I8*/, 2 /*BANG_F32*/};
      if (md_dtype >= 0 && md_dtype <= 2 && md_dtype != code_of[e->dtype]) {
        printf("Error.. Index data type (%d) does not match the requested data type\n", md_dtype);
        bang_set_error("index metadata says dtype code %d (0 int8, 1 uint8, 2 float) but the engine was created for code %d", md_dtype, code_of[e->dtype]);
        rc = BANG_ERR_ARG; break;
      }
      if (e->entry_len != (uint64_t)e->D * e->tsize + 4 + 4ull * e->R) {
        bang_set_error("index entry length %llu does not match D=%u x %zu B + 4 + 4 x R=%u", (unsigned long long)e->entry_len, e->D, e->tsize, e->R);
        rc = BANG_ERR_IO; break;
      }
    }
    // compressed vectors {i32 N, i32 m, u8[N][m]} (:218-234)
    int32_t n_pts = 0, n_chunks = 0;
    if (!read_exact(fc, &n_pts, 4) || !read_exact(fc, &n_chunks, 4) || n_pts <= 0 || n_chunks <= 0) {
      bang_set_error("bad compressed-vector header"); rc = BANG_ERR_IO; break;
    }
    if ((uint32_t)n_pts != e->N) { bang_set_error("N mismatch: metadata %u vs compressed %d", e->N, n_pts); rc = BANG_ERR_IO; break; }
    e->m = (uint32_t)n_chunks;
    codes.resize((size_t)n_pts * n_chunks);
    if (!read_exact(fc, codes.data(), codes.size())) { bang_set_error("short compressed-vector file"); rc = BANG_ERR_IO; break; }
    // pivots file: section table then {rows, cols} + data at every offset (:246-296)
    uint32_t nsec = 0;
    if (!read_exact(fp, &nsec, 4) || nsec != 4) {
      printf("Error.. PQ Pivots File does not contain the required # of sub-sections:\n");
      bang_set_error("pivots file: bad section count"); rc = BANG_ERR_IO; break;
    }
    uint64_t offs[4];
    fseek(fp, 8, SEEK_SET);
    if (!read_exact(fp, offs, 32)) { bang_set_error("pivots file: short header"); rc = BANG_ERR_IO; break; }
    pivots.resize((size_t)256 * e->D);
    centroid.resize(e->D);
    chunk_off.resize(e->m + 1);
    fseek(fp, (long)offs[0] + 8, SEEK_SET);
    bool ok = read_exact(fp, pivots.data(), pivots.size() * 4);
    fseek(fp, (long)offs[1] + 8, SEEK_SET);
    ok = ok && read_exact(fp, centroid.data(), centroid.size() * 4);
    fseek(fp, (long)offs[2] + 8, SEEK_SET);
    ok = ok && read_exact(fp, chunk_off.data(), chunk_off.size() * 4);
    if (!ok) { bang_set_error("pivots file: short section"); rc = BANG_ERR_IO; break; }
    // graph + full-precision vectors (:312-328): only checked here; upload_index streams the file (pull mode) or maps it
    e->graph = nullptr;
    if (fg) {
      fseek(fg, 0, SEEK_END);
      const size_t gsize = (size_t)ftell(fg);
      fseek(fg, 0, SEEK_SET);
      if (gsize < (size_t)e->N * e->entry_len) { bang_set_error("graph file too small"); rc = BANG_ERR_IO; break; }
      e->graph_path = f_graph;
    } else {
      e->graph_path.clear();
      e->entry_fn = diskann_entry_source; e->entry_ctx = &dsrc; e->entry_src_rereadable = true;
    }
  } while (0);
  fclose(fp); fclose(fc);
  if (fg) fclose(fg);
  if (fm) fclose(fm);
  if (rc != BANG_OK) {
    if (dsrc.fd >= 0) close(dsrc.fd);
    e->entry_fn = nullptr; e->entry_ctx = nullptr; e->entry_src_rereadable = false;
    free(e->graph_owned); e->graph_owned = nullptr;
    if (e->graph_map) (void)munmap(e->graph_map, e->graph_map_len);
    e->graph_map = nullptr; e->graph_map_len = 0;
    e->graph = nullptr;
    return rc;
  }
  rc = upload_index(e, codes.data(), nullptr, pivots.data(), centroid.data(), chunk_off.data());
  if (dsrc.fd >= 0) close(dsrc.fd);
  e->entry_fn = nullptr; e->entry_ctx = nullptr; e->entry_src_rereadable = false;
  if (rc != BANG_OK) unload_index(e);
  return rc;
}

// ------------------------------------------------------------------ the search loop of one lane
void fill_params(bang_engine* e, const Lane& ln, bang_iter_params& p) {
  const size_t q0 = ln.q0;
  memset(&p, 0, sizeof(p));
  p.Q = ln.nq; p.R = e->R; p.m = e->m; p.L = (uint32_t)e->L; p.medoid = (uint32_t)e->medoid;
  p.psz = e->psz; p.mp = e->mp; p.pq_nhi = e->pq_nhi;
  p.max_wgs = e->front_wgs;
  p.d_stage = (e->stage_mode_eff == 1) ? (e->h_stage_dev ? e->h_stage_dev + q0 * BANG_STAGE_STRIDE : nullptr)
                                        : (e->d_stage ? e->d_stage + q0 * BANG_STAGE_STRIDE : nullptr);
  p.d_seed = e->d_seed;
  p.d_codes = e->d_codes;
  p.d_pivots_packed = e->pq_nhi ? e->d_pivots_ragged : e->d_pivots_packed;
  p.d_qc = e->d_qc ? e->d_qc + q0 * e->mp * e->psz : nullptr;
  p.d_lut = e->d_lut ? e->d_lut + q0 * e->m * 256 : nullptr;
  p.d_graph = (e->graph_mode == BANG_GRAPH_DEVICE) ? e->d_graph : nullptr;
  p.entry_len = e->entry_len;
  p.vec_bytes = (uint32_t)vec_bytes(e);
  p.d_bloom = e->d_bloom + q0 * BANG_BF_WORDS;
  p.d_nbrs = e->d_nbrs + q0 * BANG_NBR_STRIDE;
  p.d_dist = e->d_dist + q0 * BANG_NBR_STRIDE;
  p.d_cnt = e->d_cnt + q0;
  p.d_wl_ids = e->d_wl_ids + q0 * e->L;
  p.d_wl_dist = e->d_wl_dist + q0 * e->L;
  p.d_wl_vis = e->d_wl_vis + q0 * e->L;
  p.d_wl_cnt = e->d_wl_cnt + q0;
  p.d_mark = e->d_mark + q0;
  p.d_parents = e->d_parents_dev + q0;
  p.d_cand_ids = e->d_cand_ids + q0 * e->cand_stride;
  p.d_cand_row = e->d_cand_row ? e->d_cand_row + q0 * e->cand_stride : nullptr;
  p.d_cand_cnt = e->d_cand_cnt + q0;
  p.d_active = nullptr;
  p.d_qstats = e->d_qstats + q0 * 2;
  if (e->graph_mode != BANG_GRAPH_DEVICE && e->use_flag) {
    p.d_done_count = e->d_done_count + (size_t)ln.index * 16;
    p.h_done_flag = e->h_done_dev + (size_t)ln.index * 16;
    p.h_parents = e->d_parents_map + q0;
  }
}

// slot for the in-kernel {start,end} stamps of the next front launch ("timing"=1), or NULL
unsigned long long* ktime_slot(bang_engine* e, Lane& ln) {
  if (!e->timing || !ln.d_ktime || ln.kt_used >= ln.kt_launches) return nullptr;
  return ln.d_ktime + (ln.kt_used++) * KT_WGS * 2;
}

// Host graph walker (bang_search.cu:771-813) for queries [i0, i1) of a lane: for every query with a parent copy
// the parent's full-precision vector into row `row` of the vector log and its adjacency list into the staging
// row (both mapped pinned memory).  Counts queries that are still active / have a parent.
void walk_slice(bang_engine* e, const Lane& ln, uint32_t i0, uint32_t i1, uint32_t row, bool adjacency,
                uint32_t* n_active, uint32_t* n_parents) {
  const size_t vb = vec_bytes(e);
  const uint64_t elen = e->entry_len;
  const uint8_t* graph = e->graph;
  const uint32_t* parents = e->h_parents + ln.q0;
  // mode 2: the rows go straight to device memory (CPU stores through the BAR; never read back from there)
  uint32_t* stage = (e->stage_mode_eff == 2 ? e->d_stage : e->h_stage) + (size_t)ln.q0 * BANG_STAGE_STRIDE;
  const bool ship_vec = !e->vec_on_device;
  uint8_t* fp_row = ship_vec ? (e->fp_direct ? e->d_fp : e->h_fp) + ((size_t)row * e->Qcur + ln.q0) * vb : nullptr;
  const uint32_t R = e->R;
  uint8_t* fin = e->h_fin.data() + ln.q0;
  uint32_t active = 0, np = 0;
  uint64_t bytes = 0;
  const uint32_t PF = 8;                                   // software prefetch distance (entries are 388-644 B)
  const uint64_t pf0 = ship_vec ? 0 : (vb & ~(uint64_t)63);  // resident vectors: only the adjacency part of an entry is touched
  for (uint32_t i = i0; i < i1 && i < i0 + PF; ++i) {
    const uint32_t par = parents[i];
    if (par < BANG_IDLE_PARENT) {
      const uint8_t* ent = graph + (uint64_t)par * elen;
      for (uint64_t o = pf0; o < elen; o += 64) __builtin_prefetch(ent + o, 0, 0);
    }
  }
  for (uint32_t i = i0; i < i1; ++i) {
    if (i + PF < i1) {
      const uint32_t par = parents[i + PF];
      if (par < BANG_IDLE_PARENT) {
        const uint8_t* ent = graph + (uint64_t)par * elen;
        for (uint64_t o = pf0; o < elen; o += 64) __builtin_prefetch(ent + o, 0, 0);
      }
    }
    const uint32_t par = parents[i];
    uint32_t* srow = stage + (size_t)i * BANG_STAGE_STRIDE;
    if (par < BANG_IDLE_PARENT) {
      const uint8_t* ent = graph + (uint64_t)par * elen;
      if (ship_vec) memcpy(fp_row + (size_t)i * vb, ent, vb);         // :796-798
      if (adjacency) {
        uint32_t deg;
        memcpy(&deg, ent + vb, 4);                                    // :801
        if (deg > R) deg = R;
        srow[0] = deg;
        memcpy(srow + 1, ent + vb + 4, (size_t)deg * 4);              // :809-810
        bytes += 4 + (uint64_t)deg * 4;
      }
      if (ship_vec) bytes += vb;
      ++active;
      ++np;
    } else {
      // memset(numNeighbors_query) :761.  A finished query's count is zeroed ONCE: its staged row is never written again, and
      // a 4-byte store per finished query, WG service and iteration is a PCIe transaction each in the tail of a search.
      if (adjacency && !(par == BANG_NO_PARENT && fin[i])) srow[0] = 0;
      if (par == BANG_NO_PARENT) fin[i] = 1;
      if (par == BANG_IDLE_PARENT) ++active;
    }
  }
  if (e->stage_mode_eff == 2) _mm_sfence();             // drain the write-combining buffers before the launch
  *n_active = active;
  *n_parents = np;
  if (bytes) ln.h2d_bytes.fetch_add(bytes, std::memory_order_relaxed);
}

inline void slice_of(const Lane& ln, int t, int T, uint32_t* i0, uint32_t* i1) {
  *i0 = (uint32_t)((uint64_t)ln.nq * (uint32_t)t / (uint32_t)T);
  *i1 = (uint32_t)((uint64_t)ln.nq * (uint32_t)(t + 1) / (uint32_t)T);
}

// Host-paced search kernel (bang_search.hip, HOST form): walker thread t of T serves the workgroups [G*t/T, G*(t+1)/T) first and any
// other workgroup when none of those is waiting.  A workgroup publishes the parents of its <= 16 waves (one 64-byte line) and then
// its round number; the thread fetches those parents' graph entries -- adjacency rows into the waves' slots of d_stage through the
// BAR, full-precision vectors into the vector log if they are not resident in HBM -- and releases the workgroup into its next
// round by storing the round number into its pacing word.
void swalk(bang_engine* e, Lane& ln, int t, int T) {
  const uint32_t G = e->sv_NG, W = 16;                     // pacing groups (workgroups x wave groups x contexts), up to 16 slots each
  const uint32_t w0 = (uint32_t)((uint64_t)G * (uint32_t)t / (uint32_t)T), w1 = (uint32_t)((uint64_t)G * (uint32_t)(t + 1) / (uint32_t)T);
  constexpr uint32_t CLAIM = 0x80000000u, FIN = 0xFFFFFFFFu;
  std::atomic<uint32_t>* expect = ln.pw_expect.get();      // per group: the round whose parents are awaited (< CLAIM); 0 = finished
  volatile uint32_t* done = e->h_done;
  uint32_t* ctl = e->d_sctl;                               // device memory, written through the BAR (write-combining)
  const uint32_t* parents = e->h_parents;
  const size_t vb = vec_bytes(e);
  const uint64_t elen = e->entry_len;
  const uint8_t* graph = e->graph;
  const bool ship_vec = !e->vec_on_device;
  uint8_t* fp = ship_vec ? (e->fp_direct ? e->d_fp : e->h_fp) : nullptr;
  const uint32_t R = e->R, cstride = e->cand_stride;
  const uint64_t pf0 = ship_vec ? 0 : (vb & ~(uint64_t)63);  // resident vectors: only the adjacency part of an entry is touched
  uint64_t bytes = 0;
  auto t_last = Clock::now();
  uint32_t idle = 0;
  bool served_unfenced = false;
  static const bool prof = getenv("BANG_WALK_PROF") != nullptr;      // diagnostic: time spent serving vs polling, per thread
  uint64_t prof_serve = 0, prof_n = 0, prof_rows = 0, prof_t0 = prof ? __rdtsc() : 0, prof_ts = 0;
  auto try_serve = [&](uint32_t w) -> bool {
    uint32_t it = expect[w].load(std::memory_order_relaxed);
    if (it == 0 || (it & CLAIM)) return false;
    const uint32_t d = done[(size_t)w * 16];
    if (d != it && d != FIN) return false;
    if (!expect[w].compare_exchange_strong(it, it | CLAIM, std::memory_order_acquire)) return false;
    std::atomic_thread_fence(std::memory_order_acquire);
    if (prof) prof_ts = __rdtsc();
    if (d == FIN) {
      expect[w].store(0, std::memory_order_release);
      ln.pw_remaining.fetch_sub(1, std::memory_order_acq_rel);
      return true;
    }
    const uint32_t* par = parents + (size_t)w * 16;
    for (uint32_t i = 0; i < W; ++i) {
      const uint32_t p_ = par[i];
      if (p_ < BANG_IDLE_PARENT) {
        const uint8_t* ent = graph + (uint64_t)p_ * elen;
        for (uint64_t o = pf0; o < elen; o += 64) __builtin_prefetch(ent + o, 0, 0);
      }
    }
    uint32_t counts[4] = {0, 0, 0, 0};
    for (uint32_t i = 0; i < W; ++i) {
      const uint32_t p_ = par[i];
      if (p_ >= BANG_IDLE_PARENT) continue;
      const uint8_t* ent = graph + (uint64_t)p_ * elen;
      bool want_row = true;
      if (ship_vec) {
        const uint32_t qw = e->h_pub_q[(size_t)w * 16 + i], c = e->h_pub_c[(size_t)w * 16 + i];
        want_row = (qw >> 31) != 0;
        memcpy(fp + ((size_t)(ln.q0 + (qw & 0x7FFFFFFFu)) * cstride + c) * vb, ent, vb);      // :796-798
        bytes += vb;
      }
      if (want_row) {
        uint32_t* srow = e->d_srows + ((size_t)w * 16 + i) * 64;       // 256-byte aligned: whole 64-byte lines, one PCIe write each
        uint32_t deg;
        memcpy(&deg, ent + vb, 4);                                      // :801
        if (deg > R) deg = R;
        const size_t nbytes = std::min<size_t>(((size_t)deg * 4 + 63) & ~(size_t)63, (size_t)R * 4);   // ids beyond `deg` are never read
        memcpy(srow, ent + vb + 4, nbytes);                             // :809-810
        counts[i >> 2] |= deg << (8 * (i & 3u));
        bytes += nbytes;
      }
    }
    _mm_sfence();                                                       // rows before the control line
    {
      // one full 64-byte line {round number, 16 count bytes, 0...}: written whole, so the write-combining buffer goes out at once
      uint32_t* cl = ctl + (size_t)w * 16;
      cl[1] = counts[0]; cl[2] = counts[1]; cl[3] = counts[2]; cl[4] = counts[3];
      for (int z = 5; z < 16; ++z) cl[z] = 0;
      cl[0] = it;
      if (ln.pw_error.load(std::memory_order_relaxed)) cl[0] = 0xFFFFFFFFu;   // another thread has stopped the kernel meanwhile: STOP stays
      bytes += 64;
    }
    served_unfenced = true;
    expect[w].store(it + 1, std::memory_order_release);
    if (prof) { prof_serve += __rdtsc() - prof_ts; ++prof_n; for (uint32_t i = 0; i < W; ++i) prof_rows += par[i] < BANG_IDLE_PARENT; }
    return true;
  };
  while (ln.pw_remaining.load(std::memory_order_acquire) != 0) {
    bool progress = false;
    for (uint32_t w = w0; w < w1; ++w) progress |= try_serve(w);
    if (!progress) {
      for (uint32_t k = 0; k + (w1 - w0) < G; ++k) {
        const uint32_t w = (w1 + k) % G;
        if (try_serve(w)) { progress = true; break; }
      }
    }
    if (served_unfenced) { _mm_sfence(); served_unfenced = false; }   // nothing lingers in a write-combining buffer while we poll
    if (progress) { idle = 0; continue; }
    _mm_pause();
    if ((++idle & 0xFFFF) == 0) {
      if (idle == 0x10000) t_last = Clock::now();
      else if (ms_since(t_last) > BANG_HOST_WALK_TIMEOUT_MS || ln.pw_error.load(std::memory_order_relaxed)) {
        ln.pw_error.store(1);
        for (uint32_t w = 0; w < G; ++w) ctl[(size_t)w * 16] = 0xFFFFFFFFu;
        _mm_sfence();
        break;
      }
      std::this_thread::yield();
    }
  }
  if (bytes) ln.h2d_bytes.fetch_add(bytes, std::memory_order_relaxed);
  if (prof) {
    const uint64_t tot = __rdtsc() - prof_t0;
    fprintf(stderr, "[walk] thread %d/%d: %llu services, %llu rows, serving %.1f%% of %.2f Mcycles, %.0f cycles/service, %.0f cycles/row\n", t, T,
            (unsigned long long)prof_n, (unsigned long long)prof_rows, 100.0 * (double)prof_serve / (double)tot, (double)tot * 1e-6,
            prof_n ? (double)prof_serve / (double)prof_n : 0.0, prof_rows ? (double)prof_serve / (double)prof_rows : 0.0);
  }
}

// helper thread t (1..T-1) of a lane's walker team
// `seen` = the lane's job epoch at the time the thread was CREATED (captured by the creator: reading it here
// would race with a first job posted before this thread gets to run, and that job would never be done)
void helper_main(bang_engine* e, Lane* ln, int t, int T, uint32_t seen) {
  Pool& pool = e->pool;
  pin_walker_thread(e, ln->index * T + t);
  for (;;) {
    // Batches usually follow each other within a millisecond or two (bang_init in between): keep spinning for a grace period
    // before parking on the condition variable -- waking eleven parked threads at the start of every batch costs 50-100 us at
    // best and whole scheduler time slices at worst, during which the lane thread serves all workgroups alone.
    {
      static const double grace_ms = getenv("BANG_HELPER_GRACE_US") ? atoi(getenv("BANG_HELPER_GRACE_US")) * 1e-3 : 4.0;
      const auto t_idle = Clock::now();
      uint32_t spins = 0;
      while (!ln->team_active.load(std::memory_order_acquire) && !pool.shutdown_flag.load(std::memory_order_relaxed)) {
        _mm_pause();
        if ((++spins & 0xFF) == 0 && ms_since(t_idle) > grace_ms) break;
      }
    }
    if (!ln->team_active.load(std::memory_order_acquire)) {
      std::unique_lock<std::mutex> lk(pool.m);
      pool.cv_team.wait(lk, [&] { return pool.shutdown || ln->team_active.load(std::memory_order_acquire); });
      if (pool.shutdown) return;
    }
    while (ln->team_active.load(std::memory_order_acquire)) {
      const uint32_t ep = ln->epoch.load(std::memory_order_acquire);
      if (ep != seen) {
        seen = ep;
        if (ln->job_kind == 2) {
          swalk(e, *ln, t, T);
        } else {
          uint32_t i0, i1, a = 0, np = 0;
          slice_of(*ln, t, T, &i0, &i1);
          walk_slice(e, *ln, i0, i1, ln->job_row, ln->job_adj, &a, &np);
          ln->job_active.fetch_add(a, std::memory_order_relaxed);
          ln->job_parents.fetch_add(np, std::memory_order_relaxed);
        }
        ln->pending.fetch_sub(1, std::memory_order_release);
      } else {
        _mm_pause();
      }
    }
  }
}

// walk the whole lane with its team; returns the number of active queries
uint32_t walk(bang_engine* e, Lane& ln, uint32_t row, bool adjacency, uint32_t* n_parents) {
  ln.phase.store(2);
  const int T = 1 + (int)ln.helpers.size();
  ln.job_kind = 0;
  ln.job_row = row;
  ln.job_adj = adjacency;
  ln.job_active.store(0, std::memory_order_relaxed);
  ln.job_parents.store(0, std::memory_order_relaxed);
  if (T > 1) {
    ln.pending.store((uint32_t)(T - 1), std::memory_order_relaxed);
    ln.epoch.fetch_add(1, std::memory_order_release);
  }
  uint32_t i0, i1, a = 0, np = 0;
  slice_of(ln, 0, T, &i0, &i1);
  walk_slice(e, ln, i0, i1, row, adjacency, &a, &np);
  if (T > 1) {
    ln.phase.store(3);
    while (ln.pending.load(std::memory_order_acquire) != 0) _mm_pause();
  }
  ln.phase.store(4);
  *n_parents = np + ln.job_parents.load(std::memory_order_relaxed);
  return a + ln.job_active.load(std::memory_order_relaxed);
}

#define LANE_HIP(x)                                                                               \
  do {                                                                                            \
    hipError_t _e = (x);                                                                          \
    if (_e != hipSuccess) {                                                                       \
      bang_set_error("%s failed: %s (%s:%d)", #x, hipGetErrorString(_e), __FILE__, __LINE__);     \
      return BANG_ERR_HIP;                                                                        \
    }                                                                                             \
  } while (0)

// spin until the front kernel of iteration `value` has published its completion flag (no HIP call)
int wait_flag(bang_engine* e, Lane& ln, uint32_t value) {
  ln.phase.store(1); ln.phase_iter.store(value);
  if (!e->use_flag) {                                  // ablation path: runtime calls instead of the in-kernel signal
    const auto t0 = Clock::now();
    if (hipMemcpyAsync(e->h_parents + ln.q0, e->d_parents_dev + ln.q0, (size_t)ln.nq * 4, hipMemcpyDeviceToHost, ln.s_main) != hipSuccess ||
        hipStreamSynchronize(ln.s_main) != hipSuccess) { bang_set_error("parent copy failed"); return BANG_ERR_HIP; }
    ln.sync_ms += ms_since(t0);
    return BANG_OK;
  }
  volatile uint32_t* flag = e->h_done + (size_t)ln.index * 16;
  const auto t0 = Clock::now();
  uint32_t spins = 0;
  while (*flag != value) {
    _mm_pause();
    if ((++spins & 0x3FF) == 0) std::this_thread::yield();   // stay polite under a CPU quota (8 ranks x lanes may exceed it)
    if ((spins & 0xFFFFF) == 0) {
      if (ms_since(t0) > 20000.0) {
        bang_set_error("timeout waiting for the front kernel of iteration %u (lane %d)", value, ln.index);
        return BANG_ERR_HIP;
      }
      const hipError_t st = hipStreamQuery(ln.s_main);
      if (st != hipSuccess && st != hipErrorNotReady) {
        bang_set_error("stream error while waiting for iteration %u: %s", value, hipGetErrorString(st));
        return BANG_ERR_HIP;
      }
    }
  }
  std::atomic_thread_fence(std::memory_order_acquire);
  ln.sync_ms += ms_since(t0);
  return BANG_OK;
}

// diagnostic build (-DBANG_SEARCH_PHASE_PROF): per-iteration phase times of wave 0 of every workgroup, slots [8..15] of its record
static void print_phase_prof(const std::vector<unsigned long long>& pr, uint32_t G) {
  double a[8] = {0, 0, 0, 0, 0, 0, 0, 0};
  for (uint32_t w = 0; w < G; ++w) for (int k = 0; k < 8; ++k) a[k] += (double)pr[(size_t)w * 16 + 8 + k];
  if (a[7] <= 0) return;
  const double n = a[7];
  fprintf(stderr, "[search] phases of an iteration (wave 0 of %u workgroups, %.0f iterations each): row arrival + loop %.2f us, hashes + probes %.2f, "
                  "compaction %.2f, filter update %.2f, code rows + distances %.2f, parent %.2f, publish + sort/merge %.2f\n", G, n / G,
          a[0] * 0.01 / n, a[1] * 0.01 / n, a[2] * 0.01 / n, a[3] * 0.01 / n, a[4] * 0.01 / n, a[5] * 0.01 / n, a[6] * 0.01 / n);
}

static int g_dbg = -1;
#define DBG(...) do { if (g_dbg < 0) g_dbg = getenv("BANG_DEBUG") ? 1 : 0; if (g_dbg) { fprintf(stderr, __VA_ARGS__); fflush(stderr); } } while (0)

int lane_run(bang_engine* e, Lane& ln, const void* h_queries, uint64_t* h_ids, float* h_dists, int Q) {
  DBG("[lane %d] start q0=%u nq=%u\n", ln.index, ln.q0, ln.nq);
  LANE_HIP(hipSetDevice(e->device));
  const bool dev_graph = (e->graph_mode == BANG_GRAPH_DEVICE);
  const uint32_t dim_adjust = (e->distfn == BANG_DIST_MIPS) ? 1u : 0u;      // :631
  const size_t qdim = e->D - dim_adjust;
  const size_t qbytes = qdim * e->tsize;
  const size_t vb = vec_bytes(e);
  const uint32_t cap_iter = (uint32_t)e->L + BANG_EXTRA_ITERS - 1;          // :950
  if (ln.kt_used) { (void)hipMemset(ln.d_ktime, 0, ln.kt_used * KT_WGS * 16); ln.kt_used = 0; }   // stats not collected
  if (e->h_fin.size() >= (size_t)ln.q0 + ln.nq) memset(e->h_fin.data() + ln.q0, 0, ln.nq);
  ln.h2d_bytes.store(0); ln.iterations = 0; ln.front_launches = 0; ln.walker_ms = 0; ln.sync_ms = 0; ln.enqueue_ms = 0;
  auto t_enq = Clock::now();
#define ENQ_BEGIN() (t_enq = Clock::now())
#define ENQ_END() (ln.enqueue_ms += ms_since(t_enq))
  bang_iter_params p;
  fill_params(e, ln, p);
  if (!dev_graph) e->h_done[(size_t)ln.index * 16] = 0;
  p.n_all = ln.nq;
  // Straggler compaction: most queries finish after ~L+5 iterations but the batch runs until its last query does (up to
  // L+49).  Once at most half of a lane's queries are active the kernels iterate over a slot -> query map of the active
  // ones only; finished queries never change state again, so skipping them cannot change any result.
  auto set_qmap = [&](const uint32_t* parents, uint32_t active, uint32_t buf) {
    if (!e->compact || active == 0 || active * 2 > ln.nq) { p.d_qmap = nullptr; p.Q = ln.nq; return; }
    uint32_t* dst = ln.qmap_host[buf];
    uint32_t n = 0;
    for (uint32_t i = 0; i < ln.nq; ++i)
      if (parents[i] != BANG_NO_PARENT) dst[n++] = i;
    if (ln.qmap_is_device) _mm_sfence();
    p.d_qmap = ln.qmap_dev[buf];
    p.Q = n;
  };
  if (!dev_graph && e->stagger_us > 0 && ln.index > 0) {
    const auto ts = Clock::now();
    while (ms_since(ts) * 1000.0 < (double)(e->stagger_us * ln.index)) _mm_pause();
  }

  // diagnostic (BANG_TIMELINE=1): host time of every stage of one bang_query, with a stream sync behind each (stderr)
  static const bool tl_on = getenv("BANG_TIMELINE") != nullptr;
  auto tl_t = Clock::now();
  auto tl = [&](const char* what) {
    if (!tl_on) return;
    (void)hipStreamSynchronize(ln.s_main);
    fprintf(stderr, "[timeline lane %d] %-28s %8.1f us\n", ln.index, what, ms_since(tl_t) * 1000.0);
    tl_t = Clock::now();
  };
  // queries H2D (:612) + K1 (:623)
  uint8_t* dq = (uint8_t*)e->d_queries + (size_t)ln.q0 * qbytes;
  LANE_HIP(hipMemcpyAsync(dq, (const uint8_t*)h_queries + (size_t)ln.q0 * qbytes, (size_t)ln.nq * qbytes,
                          hipMemcpyHostToDevice, ln.s_main));
  if (e->psz)
    BANG_TRY(bang_k_center_queries(dq, e->dtype, e->d_centroid, e->d_chunk_off, (float*)p.d_qc, ln.nq, e->D, e->m,
                                   e->mp, e->psz, dim_adjust, ln.s_main));
  else
    BANG_TRY(bang_k_lut_build(e->d_pivots_T, dq, e->dtype, e->d_centroid, e->d_chunk_off, (float*)p.d_lut, ln.nq,
                              e->D, e->m, dim_adjust, ln.s_main));

  tl("queries H2D + K1");
  uint32_t iter = 1;                                                         // :596
  // vector-log rows [fp_lo, fp_hi] are staged in pinned memory but not yet copied to the device
  uint32_t fp_lo = 0, fp_hi = 0;
  bool fp_pending = false, fp_any = false;
  auto flush_fp = [&]() -> int {                                             // :836-838, batched
    if (!fp_pending) return BANG_OK;
    const size_t off = ((size_t)fp_lo * e->Qcur + ln.q0) * vb;
    LANE_HIP(hipMemcpy2DAsync(e->d_fp + off, (size_t)e->Qcur * vb, e->h_fp + off, (size_t)e->Qcur * vb,
                              (size_t)ln.nq * vb, (size_t)(fp_hi - fp_lo + 1), hipMemcpyHostToDevice, ln.s_fp));
    fp_pending = false;
    fp_any = true;
    return BANG_OK;
  };

  uint32_t pw_stats[2] = {0, 0};
  if (e->search_v2) {
    // graph resident in HBM: ONE launch of the query-resident search kernel; no host involvement until the re-rank
    LANE_HIP(hipMemsetAsync(ln.d_pcnt, 0, 64, ln.s_main));
    bang_search_params sp;
    memset(&sp, 0, sizeof(sp));
    sp.Q = ln.nq; sp.R = e->R; sp.m = e->m; sp.L = (uint32_t)e->L; sp.medoid = (uint32_t)e->medoid; sp.cap_iter = cap_iter;
    sp.psz = e->psz; sp.mp = e->mp; sp.pq_nhi = e->pq_nhi;
    sp.d_seed = e->d_seed; sp.d_codes = e->d_codes; sp.d_pivots_packed = p.d_pivots_packed; sp.d_qc = p.d_qc;
    sp.d_graph = e->d_graph; sp.entry_len = e->entry_len; sp.vec_bytes = (uint32_t)vb;
    if (!dev_graph) { sp.d_graph = (const uint8_t*)e->d_adj; sp.entry_len = 256; sp.vec_bytes = 0; sp.row_layout = 1; }   // pull mode
    sp.d_bloom = p.d_bloom; sp.d_cand_ids = p.d_cand_ids; sp.d_cand_cnt = p.d_cand_cnt; sp.d_qstats = p.d_qstats;
    sp.d_qiters = e->d_qiters + ln.q0; sp.d_next_query = ln.d_pcnt;
    sp.d_qskip = e->d_qskip + ln.q0;
    sp.d_ktime = ktime_slot(e, ln);
    {
      static const int env_wgs = getenv("BANG_SEARCH_MAX_WGS") ? atoi(getenv("BANG_SEARCH_MAX_WGS")) : 0;   // experiment / test knobs
      static const int env_waves = getenv("BANG_SEARCH_MAX_WAVES") ? atoi(getenv("BANG_SEARCH_MAX_WAVES")) : 0;
      const char* v1 = getenv("BANG_SEARCH_MAX_WGS");
      sp.max_wgs = v1 ? (uint32_t)std::max(0, atoi(v1)) : (uint32_t)std::max(0, env_wgs);
      sp.max_waves = (uint32_t)std::max(0, env_waves);
    }
    static const bool kprof_d = getenv("BANG_SEARCH_PROF") != nullptr;   // diagnostic build only (-DBANG_SEARCH_PHASE_PROF)
    unsigned long long* d_prof = nullptr;
    const uint32_t Gd = (uint32_t)std::min<int>((int)ln.nq, bang_num_cus());
    if (kprof_d) { LANE_HIP(hipMalloc((void**)&d_prof, (size_t)Gd * 128)); LANE_HIP(hipMemsetAsync(d_prof, 0, (size_t)Gd * 128, ln.s_main)); sp.d_prof = d_prof; }
    ENQ_BEGIN();
    BANG_TRY(bang_k_search(&sp, ln.s_main));
    ENQ_END();
    ++ln.front_launches;
    if (d_prof) {
      std::vector<unsigned long long> pr((size_t)Gd * 16);
      (void)hipStreamSynchronize(ln.s_main);
      (void)hipMemcpy(pr.data(), d_prof, pr.size() * 8, hipMemcpyDeviceToHost);
      (void)hipFree(d_prof);
      print_phase_prof(pr, Gd);
    }
    iter = cap_iter;                                                         // refined from the per-query counts below
  } else if (e->search_host) {
    // graph in host RAM: ONE launch of the host-paced form; its workgroups are served round by round by the walker team
    uint32_t G = 0, W = 0;
    {
      const char* v1 = getenv("BANG_SEARCH_MAX_WGS");
      const char* v2 = getenv("BANG_SEARCH_MAX_WAVES");
      const char* v3 = getenv("BANG_SEARCH_CTX");
      const char* v4 = getenv("BANG_SEARCH_GS");
      uint32_t C = v3 ? (uint32_t)std::max(0, atoi(v3)) : 0u, GS = v4 ? (uint32_t)std::max(0, atoi(v4)) : 0u;
      BANG_TRY(bang_search_geometry(e->psz, e->mp, e->pq_nhi, (uint32_t)e->L, ln.nq, v1 ? (uint32_t)std::max(0, atoi(v1)) : 0u,
                                    v2 ? (uint32_t)std::max(0, atoi(v2)) : 0u, 1, &G, &W, &C, &GS));
      e->sv_C = C; e->sv_GS = GS;
      e->sv_NG = G * ((W + GS - 1) / GS) * C;
      if (e->sv_NG > 8 * KT_WGS) { bang_set_error("search kernel: %u pacing groups exceed the pacing buffers", e->sv_NG); return BANG_ERR_ARG; }
    }
    e->sv_G = G; e->sv_W = W;
    const uint32_t NG = e->sv_NG;                                              // pacing groups
    for (size_t i = 0; i < (size_t)NG * 16; ++i) e->h_parents[i] = BANG_NO_PARENT;
    ln.pw_groups = NG;
    ln.pw_error.store(0);
    for (uint32_t w = 0; w < NG; ++w) ln.pw_expect[w].store(1u, std::memory_order_relaxed);
    ln.pw_remaining.store(NG, std::memory_order_release);
    for (uint32_t w = 0; w < NG; ++w) { e->h_done[(size_t)w * 16] = 0; e->d_sctl[(size_t)w * 16] = 0; }
    _mm_sfence();
    LANE_HIP(hipMemsetAsync(ln.d_pcnt, 0, 64, ln.s_main));
    bang_search_params sp;
    memset(&sp, 0, sizeof(sp));
    sp.Q = ln.nq; sp.R = e->R; sp.m = e->m; sp.L = (uint32_t)e->L; sp.medoid = (uint32_t)e->medoid; sp.cap_iter = cap_iter;
    sp.psz = e->psz; sp.mp = e->mp; sp.pq_nhi = e->pq_nhi;
    sp.max_wgs = G; sp.max_waves = W; sp.nctx = e->sv_C; sp.group_waves = e->sv_GS;
    sp.d_seed = e->d_seed; sp.d_codes = e->d_codes; sp.d_pivots_packed = p.d_pivots_packed; sp.d_qc = p.d_qc;
    sp.d_graph = nullptr; sp.entry_len = e->entry_len; sp.vec_bytes = (uint32_t)vb;
    sp.d_bloom = p.d_bloom; sp.d_cand_ids = p.d_cand_ids; sp.d_cand_cnt = p.d_cand_cnt; sp.d_qstats = p.d_qstats;
    sp.d_qiters = e->d_qiters + ln.q0; sp.d_next_query = ln.d_pcnt; sp.d_abort = ln.d_pcnt + 1;
    sp.d_ktime = ktime_slot(e, ln);
    sp.d_rows = e->d_srows; sp.d_ctl = e->d_sctl; sp.h_done = e->h_done_dev; sp.h_parents = e->d_parents_map;
    sp.h_pub_q = e->d_pub_q; sp.h_pub_c = e->d_pub_c; sp.ship_vectors = e->vec_on_device ? 0u : 1u;
    static const bool kprof = getenv("BANG_SEARCH_PROF") != nullptr;     // diagnostic: phase times of the half-rounds (stderr)
    unsigned long long* d_prof = nullptr;
    if (kprof) { LANE_HIP(hipMalloc((void**)&d_prof, (size_t)G * 128)); LANE_HIP(hipMemsetAsync(d_prof, 0, (size_t)G * 128, ln.s_main)); sp.d_prof = d_prof; }
    BANG_TRY(bang_k_search(&sp, ln.s_main));
    ++ln.front_launches;
    const auto t0 = Clock::now();
    const int T = 1 + (int)ln.helpers.size();
    ln.job_kind = 2;
    if (T > 1) {
      ln.pending.store((uint32_t)(T - 1), std::memory_order_relaxed);
      ln.epoch.fetch_add(1, std::memory_order_release);
    }
    swalk(e, ln, 0, T);
    if (T > 1) while (ln.pending.load(std::memory_order_acquire) != 0) _mm_pause();
    ln.walker_ms += ms_since(t0);
    if (ln.pw_error.load()) { bang_set_error("timeout waiting for the search kernel"); (void)hipStreamSynchronize(ln.s_main); return BANG_ERR_HIP; }
    if (d_prof) {
      std::vector<unsigned long long> pr((size_t)G * 16);
      (void)hipStreamSynchronize(ln.s_main);
      (void)hipMemcpy(pr.data(), d_prof, pr.size() * 8, hipMemcpyDeviceToHost);
      (void)hipFree(d_prof);
      print_phase_prof(pr, G);
      double a[5] = {0, 0, 0, 0, 0};
      for (uint32_t w = 0; w < G; ++w) for (int k = 0; k < 5; ++k) a[k] += (double)pr[(size_t)w * 16 + k];
      const double n = a[4] > 0 ? a[4] : 1;
      fprintf(stderr, "[search] %u workgroups x %u waves x %u contexts, %u waves per pacing group: %.0f half-rounds per group; per half-round: wait for rows %.2f us, "
                      "front (to the publish barrier) %.2f us, publish %.2f us, sort/merge %.2f us\n", G, W, e->sv_C, e->sv_GS, a[4] / G,
              a[0] * 0.01 / n, a[1] * 0.01 / n, a[2] * 0.01 / n, a[3] * 0.01 / n);
    }
    if (!e->fp_direct && !e->vec_on_device)     // vectors staged in pinned memory: one copy of this lane's part of the log
      LANE_HIP(hipMemcpyAsync(e->d_fp + (size_t)ln.q0 * e->cand_stride * vb, e->h_fp + (size_t)ln.q0 * e->cand_stride * vb,
                              (size_t)ln.nq * e->cand_stride * vb, hipMemcpyHostToDevice, ln.s_main));
    iter = cap_iter;                                                         // refined from the per-query counts below
  } else {
  p.first = 1; p.iter = iter; p.done_value = iter;
  if (dev_graph) p.d_active = e->d_active + iter;
  p.d_ktime = ktime_slot(e, ln);
  BANG_TRY(bang_k_front(&p, ln.s_main));                                     // K5+K2+K4a :650-678
  ++ln.front_launches;

  for (;;) {
    p.first = 0; p.iter = iter;
    ENQ_BEGIN();
    BANG_TRY(bang_k_back(&p, ln.s_main));                                    // K3a+K3b :726-738 (overlaps the walker)
    ENQ_END();
    if (!dev_graph) {
      DBG("[lane %d] wait flag %u\n", ln.index, iter);
      BANG_TRY(wait_flag(e, ln, iter));                                      // parents of this iteration are in h_parents :709,763
      DBG("[lane %d] got flag %u\n", ln.index, iter);
      const auto t0 = Clock::now();
      uint32_t n_par = 0;
      const uint32_t active = walk(e, ln, iter, true, &n_par);               // CPU walker :771-813
      ln.walker_ms += ms_since(t0);
      if (n_par && !e->vec_on_device) {
        if (!fp_pending) { fp_lo = iter; fp_pending = true; }
        fp_hi = iter;
      }
      if (active == 0) break;                                                // :958
      set_qmap(e->h_parents + ln.q0, active, (iter + 1) & 1u);
      ENQ_BEGIN();
      if (e->stage_mode_eff == 0)
      LANE_HIP(hipMemcpyAsync((void*)(e->d_stage + (size_t)ln.q0 * BANG_STAGE_STRIDE), e->h_stage + (size_t)ln.q0 * BANG_STAGE_STRIDE,
                              (size_t)ln.nq * BANG_STAGE_STRIDE * 4, hipMemcpyHostToDevice, ln.s_main));   // :827-833
      ENQ_END();
      if (fp_pending && fp_hi - fp_lo + 1 >= (uint32_t)e->fp_batch) { ENQ_BEGIN(); BANG_TRY(flush_fp()); ENQ_END(); }
    }
    ++iter;                                                                  // :879
    p.iter = iter; p.done_value = iter;
    if (dev_graph) p.d_active = e->d_active + iter;
    ENQ_BEGIN();
    p.d_ktime = ktime_slot(e, ln);
    BANG_TRY(bang_k_front(&p, ln.s_main));                                   // K5+K2+K4b :855-917
    ++ln.front_launches;
    ENQ_END();
    if (!dev_graph) {
      if (iter == cap_iter) {                                                // :950-956
        // CANON: the vectors of the parents chosen at the cap are still fetched for the re-rank
        BANG_TRY(wait_flag(e, ln, iter));
        uint32_t n_par = 0;
        (void)walk(e, ln, iter, false, &n_par);
        if (n_par && !e->vec_on_device) {
          if (!fp_pending) { fp_lo = iter; fp_pending = true; }
          fp_hi = iter;
        }
        break;
      }
    } else {
      if (iter == cap_iter) break;
      if ((iter % (uint32_t)e->check_every) == 0) {                          // :942-943 (amortised)
        uint32_t act = 0;
        LANE_HIP(hipMemcpyAsync(&act, e->d_active + iter, 4, hipMemcpyDeviceToHost, ln.s_main));
        LANE_HIP(hipStreamSynchronize(ln.s_main));
        if (act == 0) break;
        if (e->compact) {                                                    // refresh the slot -> query map from the parents
          if (ln.parents_tmp.size() < ln.nq) ln.parents_tmp.resize(ln.nq);
          LANE_HIP(hipMemcpy(ln.parents_tmp.data(), e->d_parents_dev + ln.q0, (size_t)ln.nq * 4, hipMemcpyDeviceToHost));
          uint32_t n_act = 0;
          for (uint32_t i = 0; i < ln.nq; ++i) n_act += ln.parents_tmp[i] != BANG_NO_PARENT;
          set_qmap(ln.parents_tmp.data(), n_act, 0);
        }
      }
    }
  }
  }
  ln.iterations = iter;
  ln.phase.store(5);
  DBG("[lane %d] loop done iter=%u\n", ln.index, iter);
  BANG_TRY(flush_fp());

  tl("search");
  // re-rank K6+K7 (:967-987)
  if (fp_any) {
    LANE_HIP(hipEventRecord(ln.ev_fp, ln.s_fp));
    LANE_HIP(hipStreamWaitEvent(ln.s_main, ln.ev_fp, 0));
  }
  {
    if (dev_graph)
      BANG_TRY(bang_k_rerank_range(e->d_graph, e->entry_len, e->d_medoid_vec, e->d_queries, e->dtype, e->d_cand_ids,
                                   nullptr, e->d_cand_cnt, e->cand_stride, ln.q0, ln.nq, (uint32_t)Q, e->D,
                                   (uint32_t)e->k, dim_adjust, e->d_ids_out, e->d_dists_out, ln.s_main));
    else if (e->vec_on_device)
      BANG_TRY(bang_k_rerank_range(e->d_vecs, vb, e->d_medoid_vec, e->d_queries, e->dtype, e->d_cand_ids, nullptr, e->d_cand_cnt,
                                   e->cand_stride, ln.q0, ln.nq, (uint32_t)Q, e->D, (uint32_t)e->k, dim_adjust, e->d_ids_out,
                                   e->d_dists_out, ln.s_main));
    else if (e->search_host)
      BANG_TRY(bang_k_rerank_byquery(e->d_fp, vb, e->d_medoid_vec, e->d_queries, e->dtype, e->d_cand_ids, e->d_cand_cnt,
                                     e->cand_stride, ln.q0, ln.nq, (uint32_t)Q, e->D, (uint32_t)e->k, dim_adjust, e->d_ids_out,
                                     e->d_dists_out, ln.s_main));
    else
      BANG_TRY(bang_k_rerank_range(e->d_fp, vb, e->d_medoid_vec, e->d_queries, e->dtype, e->d_cand_ids, e->d_cand_row,
                                   e->d_cand_cnt, e->cand_stride, ln.q0, ln.nq, (uint32_t)Q, e->D, (uint32_t)e->k,
                                   dim_adjust, e->d_ids_out, e->d_dists_out, ln.s_main));
  }
  tl("re-rank");
  // results D2H (:997-999): ids [Q][k]; dists [k][Q] (rank-major)
  // A copy into the caller's pageable arrays is staged by the runtime and costs ~20 us before the first byte moves, per copy.  The
  // results come back whole in ONE asynchronous copy into the pinned mirror and are handed out with memcpy (measured: 70 -> 17 us for
  // a 1 250-query shard, 92-107 -> 73-82 us for the 10 K batch); only a very large batch keeps the direct, runtime-pipelined copies.
  const bool whole = ln.q0 == 0 && (int)ln.nq == Q && (int)ln.nq == e->Qcur;
  static const size_t mailbox_max = getenv("BANG_MAILBOX_BYTES") ? (size_t)atoll(getenv("BANG_MAILBOX_BYTES")) : (size_t)BANG_RESULT_MAILBOX_BYTES;
  uint64_t* d_ids_user = e->pool.d_ids_user;
  float* d_dists_user = e->pool.d_dists_user;
  const bool to_device = d_ids_user != nullptr;               // bang_query_dev_e: no result leaves the device
  const bool mailbox = !to_device && whole && e->res_off_iters <= mailbox_max;
  uint32_t* h_abort = (uint32_t*)(e->h_results + e->res_bytes - BANG_MAX_LANES * 4) + ln.index;   // (one word per lane behind the results)
  *h_abort = 0;
  if (e->search_host) LANE_HIP(hipMemcpyAsync(h_abort, ln.d_pcnt + 1, 4, hipMemcpyDeviceToHost, ln.s_main));
  const bool iters = e->search_v2 || e->search_host;
  if (to_device) {
    LANE_HIP(hipMemcpyAsync(d_ids_user + (size_t)ln.q0 * e->k, e->d_ids_out + (size_t)ln.q0 * e->k, (size_t)ln.nq * e->k * sizeof(uint64_t),
                            hipMemcpyDeviceToDevice, ln.s_main));
    if (d_dists_user)
      LANE_HIP(hipMemcpy2DAsync(d_dists_user + ln.q0, (size_t)Q * 4, e->d_dists_out + ln.q0, (size_t)Q * 4, (size_t)ln.nq * 4, (size_t)e->k,
                                hipMemcpyDeviceToDevice, ln.s_main));
    if (iters) LANE_HIP(hipMemcpyAsync(e->h_results + e->res_off_iters + (size_t)ln.q0 * 4, e->d_qiters + ln.q0, (size_t)ln.nq * 4,
                                       hipMemcpyDeviceToHost, ln.s_main));
  } else if (mailbox) {
    LANE_HIP(hipMemcpyAsync(e->h_results, e->d_results, iters ? e->res_off_iters + (size_t)ln.nq * 4 : e->res_off_iters,
                            hipMemcpyDeviceToHost, ln.s_main));
  } else {
    LANE_HIP(hipMemcpyAsync(h_ids + (size_t)ln.q0 * e->k, e->d_ids_out + (size_t)ln.q0 * e->k,
                            (size_t)ln.nq * e->k * sizeof(uint64_t), hipMemcpyDeviceToHost, ln.s_main));
    LANE_HIP(hipMemcpy2DAsync(h_dists + ln.q0, (size_t)Q * 4, e->d_dists_out + ln.q0, (size_t)Q * 4, (size_t)ln.nq * 4,
                              (size_t)e->k, hipMemcpyDeviceToHost, ln.s_main));
    if (iters) LANE_HIP(hipMemcpyAsync(e->h_results + e->res_off_iters + (size_t)ln.q0 * 4, e->d_qiters + ln.q0, (size_t)ln.nq * 4,
                                       hipMemcpyDeviceToHost, ln.s_main));
  }
  ln.phase.store(6);
  LANE_HIP(hipStreamSynchronize(ln.s_main));
  ln.phase.store(7);
  if (mailbox) {
    memcpy(h_ids, e->h_results, (size_t)ln.nq * e->k * sizeof(uint64_t));
    memcpy(h_dists, e->h_results + e->res_off_dists, (size_t)ln.nq * e->k * 4);
  }
  tl("results D2H");
  pw_stats[0] = *h_abort;
  if (pw_stats[0]) { bang_set_error("search kernel gave up waiting for the host walker"); return BANG_ERR_HIP; }
  if (iters) {
    const uint32_t* hq = (const uint32_t*)(e->h_results + e->res_off_iters) + ln.q0;
    uint32_t mx = 0;
    for (uint32_t i = 0; i < ln.nq; ++i) { e->h_qiters[ln.q0 + i] = hq[i]; mx = std::max(mx, hq[i]); }
    ln.iterations = mx;
  }
  DBG("[lane %d] synced\n", ln.index);
  ln.front_ms = ln.back_ms = ln.rerank_ms = 0;   // the in-kernel stamps are reduced lazily in bang_get_stats
  return BANG_OK;
}

// one lane's work for the current query, with its walker team switched on for the duration
void lane_job(bang_engine* e, Lane& ln) {
  Pool& pool = e->pool;
  if (ln.nq == 0) { ln.rc = BANG_OK; return; }
  if (!ln.helpers.empty()) {
    { std::lock_guard<std::mutex> lk(pool.m); ln.team_active.store(true, std::memory_order_release); }
    pool.cv_team.notify_all();
  }
  ln.rc = lane_run(e, ln, pool.h_queries, pool.h_ids, pool.h_dists, pool.Q);
  if (ln.rc != BANG_OK) ln.err = bang_last_error();
  ln.team_active.store(false, std::memory_order_release);
}

void lane_thread_main(bang_engine* e, Lane* ln) {
  Pool& pool = e->pool;
  pin_walker_thread(e, ln->index * std::max(1, e->threads_eff));
  uint64_t seen = 0;
  for (;;) {
    {
      std::unique_lock<std::mutex> lk(pool.m);
      pool.cv_start.wait(lk, [&] { return pool.shutdown || pool.query_seq != seen; });
      if (pool.shutdown) return;
      seen = pool.query_seq;
    }
    lane_job(e, *ln);
    {
      std::lock_guard<std::mutex> lk(pool.m);
      ++pool.lanes_done;
    }
    pool.cv_done.notify_all();
  }
}

void start_threads(bang_engine* e) {
  Pool& pool = e->pool;
  pool.shutdown = false;
  pool.shutdown_flag.store(false);
  pool.query_seq = 0;
  const int nl = (int)e->lanes.size();
  const int T = std::max(1, e->threads_eff);
  for (int i = 0; i < nl; ++i) {
    Lane* ln = e->lanes[(size_t)i].get();
    const uint32_t epoch0 = ln->epoch.load(std::memory_order_acquire);
    for (int t = 1; t < T; ++t) ln->helpers.emplace_back(helper_main, e, ln, t, T, epoch0);
    if (i > 0) pool.lane_threads.emplace_back(lane_thread_main, e, ln);      // lane 0 runs on the caller's thread
  }
}

void stop_threads(bang_engine* e) {
  Pool& pool = e->pool;
  {
    std::lock_guard<std::mutex> lk(pool.m);
    pool.shutdown = true;
    pool.shutdown_flag.store(true);
  }
  pool.cv_start.notify_all();
  pool.cv_team.notify_all();
  for (auto& t : pool.lane_threads) t.join();
  pool.lane_threads.clear();
  for (auto& lp : e->lanes) {
    for (auto& t : lp->helpers) t.join();
    lp->helpers.clear();
  }
  pool.shutdown = false;
}

}  // namespace

// ------------------------------------------------------------------ C-ABI, engine level
extern "C" int bang_create(int dtype, bang_engine_t** out) {
  if (!out || dtype < BANG_U8 || dtype > BANG_F32) { bang_set_error("bad dtype"); return BANG_ERR_ARG; }
  bang_engine* e = new (std::nothrow) bang_engine();
  if (!e) return BANG_ERR_NOMEM;
  e->dtype = dtype;
  e->tsize = (dtype == BANG_F32) ? 4 : 1;
  // defaults from the environment so that callers of the bang.h class API (no option methods,
  // e.g. the bang_search CLI) can still choose the placement: BANG_GRAPH=host|device,
  // BANG_LANES=n, BANG_DEVICE=ordinal, BANG_PQ=0|1, BANG_TIMING=0|1
  if (const char* v = getenv("BANG_GRAPH"))
    e->graph_mode = (strcmp(v, "device") == 0 || strcmp(v, "1") == 0) ? BANG_GRAPH_DEVICE
                  : (strcmp(v, "auto") == 0 || strcmp(v, "2") == 0) ? BANG_GRAPH_AUTO : BANG_GRAPH_HOST;
  e->graph_opt = e->graph_mode;
  if (const char* v = getenv("BANG_LANES")) e->lanes_opt = std::min(BANG_MAX_LANES, std::max(0, atoi(v)));
  if (const char* v = getenv("BANG_THREADS")) e->threads_opt = std::max(0, atoi(v));
  if (const char* v = getenv("BANG_CHECK_EVERY")) e->check_every = std::max(1, atoi(v));
  if (const char* v = getenv("BANG_FRONT_WGS")) e->front_wgs_opt = atoi(v);
  if (const char* v = getenv("BANG_COMPACT")) e->compact = atoi(v) ? 1 : 0;
  if (const char* v = getenv("BANG_USE_FLAG")) e->use_flag = atoi(v) ? 1 : 0;
  if (const char* v = getenv("BANG_PERSISTENT")) e->persistent = std::min(1, std::max(-1, atoi(v)));
  if (const char* v = getenv("BANG_NUMA")) e->numa_opt = std::min(1, std::max(-1, atoi(v)));
  if (const char* v = getenv("BANG_PULL")) e->pull_opt = std::min(1, std::max(-1, atoi(v)));
  if (const char* v = getenv("BANG_SEARCH")) e->search_opt = std::min(1, std::max(-1, atoi(v)));
  if (const char* v = getenv("BANG_STAGE_ZC")) e->stage_zero_copy = std::min(2, std::max(0, atoi(v)));
  if (const char* v = getenv("BANG_STAGGER_US")) e->stagger_us = std::max(0, atoi(v));
  if (const char* v = getenv("BANG_FP_BATCH")) e->fp_batch = std::max(1, atoi(v));
  if (const char* v = getenv("BANG_DEVICE")) e->device = atoi(v);
  if (const char* v = getenv("BANG_PQ")) e->pq_mode = atoi(v);
  if (const char* v = getenv("BANG_TIMING")) e->timing = atoi(v);
  if (const char* v = getenv("BANG_PQ_RAGGED")) e->pq_ragged = atoi(v) ? 1 : 0;
  if (const char* v = getenv("BANG_VECTORS")) e->vectors_opt = std::min(1, std::max(-1, atoi(v)));
  *out = e;
  return BANG_OK;
}

extern "C" int bang_destroy(bang_engine_t* e) {
  if (!e) return BANG_OK;
  if (e->allocated) free_batch(e);
  if (e->loaded) unload_index(e);
  delete e;
  return BANG_OK;
}

extern "C" int bang_set_option(bang_engine_t* e, const char* key, long value) {
  if (!e || !key) return BANG_ERR_ARG;
  const std::string k(key);
  // placement and layout options are consumed by bang_load, loop-shape options by bang_alloc (bang_c.h): changing them
  // afterwards would leave buffers that do not match the option
  if (e->loaded && (k == "graph" || k == "device" || k == "pq" || k == "pq_ragged" || k == "vectors" || k == "pull")) {
    bang_set_error("option %s must be set before bang_load", key); return BANG_ERR_ARG;
  }
  if (e->allocated && (k == "lanes" || k == "threads" || k == "stage_zero_copy" || k == "persistent" || k == "search" || k == "numa" || k == "timing" || k == "front_wgs")) {
    bang_set_error("option %s must be set before bang_alloc", key); return BANG_ERR_ARG;
  }
  if (k == "graph") { if (value != BANG_GRAPH_HOST && value != BANG_GRAPH_DEVICE && value != BANG_GRAPH_AUTO) return BANG_ERR_ARG; e->graph_mode = e->graph_opt = (int)value; }
  else if (k == "lanes") { if (value < 0 || value > BANG_MAX_LANES) return BANG_ERR_ARG; e->lanes_opt = (int)value; }
  else if (k == "threads") { if (value < 0) return BANG_ERR_ARG; e->threads_opt = (int)value; }
  else if (k == "device") { e->device = (int)value; }
  else if (k == "pq") { e->pq_mode = (int)value; }
  else if (k == "timing") { e->timing = (int)value; }
  else if (k == "pq_ragged") { if (e->loaded) return BANG_ERR_ARG; e->pq_ragged = value ? 1 : 0; }
  else if (k == "vectors") { if (value < -1 || value > 1 || e->loaded) return BANG_ERR_ARG; e->vectors_opt = (int)value; }
  else if (k == "stage_zero_copy") { if (value < -1 || value > 2) return BANG_ERR_ARG; e->stage_zero_copy = (int)value; }
  else if (k == "stagger_us") { if (value < 0) return BANG_ERR_ARG; e->stagger_us = (int)value; }
  else if (k == "compact") { e->compact = value ? 1 : 0; }
  else if (k == "persistent") { if (value < -1 || value > 1) return BANG_ERR_ARG; e->persistent = (int)value; }
  else if (k == "numa") { if (value < -1 || value > 1) return BANG_ERR_ARG; e->numa_opt = (int)value; }
  else if (k == "pull") { if (value < -1 || value > 1) return BANG_ERR_ARG; e->pull_opt = (int)value; }
  else if (k == "search") { if (value < -1 || value > 1) return BANG_ERR_ARG; e->search_opt = (int)value; }
  else if (k == "fp_batch") { if (value < 1) return BANG_ERR_ARG; e->fp_batch = (int)value; }
  else if (k == "front_wgs") { if (value < 0) return BANG_ERR_ARG; e->front_wgs_opt = (int)value; }
  else if (k == "check_every") { if (value < 1) return BANG_ERR_ARG; e->check_every = (int)value; }
  else { bang_set_error("unknown option %s", key); return BANG_ERR_ARG; }
  return BANG_OK;
}

extern "C" int bang_load_e(bang_engine_t* e, const char* prefix) {
  if (!e || !prefix) return BANG_ERR_ARG;
  if (e->loaded) { bang_set_error("index already loaded"); return BANG_ERR_ARG; }
  BANG_TRY(ensure_device(e));
  return load_files(e, prefix);
}

extern "C" int bang_load_mem_e(bang_engine_t* e, const bang_index_desc* d) {
  if (!e || !d || !d->graph || !d->pivots || !d->centroid || !d->chunk_off || (!d->codes && !d->d_codes)) {
    bang_set_error("bad index descriptor");
    return BANG_ERR_ARG;
  }
  if (e->loaded) { bang_set_error("index already loaded"); return BANG_ERR_ARG; }
  BANG_TRY(ensure_device(e));
  e->medoid = d->medoid; e->entry_len = d->entry_len; e->D = d->D; e->R = d->R; e->N = d->N; e->m = d->m;
  e->graph = d->graph;
  e->graph_owned = nullptr;
  e->graph_path.clear();
  const int rc = upload_index(e, d->codes, d->d_codes, d->pivots, d->centroid, d->chunk_off);
  if (rc != BANG_OK) unload_index(e);
  return rc;
}

extern "C" int bang_load_stream_e(bang_engine_t* e, const bang_index_desc* d, bang_entry_source src, void* ctx) {
  if (!e || !d || !src || d->graph || !d->pivots || !d->centroid || !d->chunk_off || (!d->codes && !d->d_codes)) {
    bang_set_error("bad index descriptor (a streamed load takes an entry source and no graph pointer)");
    return BANG_ERR_ARG;
  }
  if (e->loaded) { bang_set_error("index already loaded"); return BANG_ERR_ARG; }
  BANG_TRY(ensure_device(e));
  e->medoid = d->medoid; e->entry_len = d->entry_len; e->D = d->D; e->R = d->R; e->N = d->N; e->m = d->m;
  e->graph = nullptr;
  e->graph_owned = nullptr;
  e->graph_path.clear();
  if (e->graph_mode == BANG_GRAPH_DEVICE) { bang_set_error("a streamed load keeps the adjacency lists on the host (pull mode): option graph = device does not apply"); return BANG_ERR_UNSUPPORTED; }
  e->graph_mode = BANG_GRAPH_HOST;                       // (auto included: there is no graph image to put into HBM)
  if (e->entry_len != (uint64_t)e->D * e->tsize + 4 + 4ull * e->R) {
    bang_set_error("index entry length %llu does not match D=%u x %zu B + 4 + 4 x R=%u", (unsigned long long)e->entry_len, e->D, e->tsize, e->R);
    return BANG_ERR_ARG;
  }
  e->entry_fn = src; e->entry_ctx = ctx;
  const int rc = upload_index(e, d->codes, d->d_codes, d->pivots, d->centroid, d->chunk_off);
  e->entry_fn = nullptr; e->entry_ctx = nullptr;
  if (rc != BANG_OK) unload_index(e);
  return rc;
}

extern "C" int bang_set_searchparams_e(bang_engine_t* e, int recall, int worklist_length, int distfn) {
  if (!e) return BANG_ERR_ARG;
  if (recall <= 0 || worklist_length < recall || worklist_length > BANG_MAX_L ||            // assert(2L <= 1024) :439
      (distfn != BANG_DIST_L2 && distfn != BANG_DIST_MIPS)) {
    bang_set_error("bad search params: recall=%d L=%d distfn=%d", recall, worklist_length, distfn);
    return BANG_ERR_ARG;
  }
  if (e->allocated && (recall != e->k || worklist_length != e->L)) {
    bang_set_error("bang_free must be called before changing recall / worklist length");   // sizes depend on them :370-384
    return BANG_ERR_ARG;
  }
  e->k = recall; e->L = worklist_length; e->distfn = distfn;
  e->params_set = true;
  return BANG_OK;
}


static int alloc_buffers(bang_engine* e, int Q) {
  const size_t L = (size_t)e->L, nq = (size_t)Q;
  const size_t rows = L + BANG_EXTRA_ITERS;                                  // uMAX_PARENTS_PERQUERY :370
  const size_t vb = vec_bytes(e);
  const bool dev_graph = (e->graph_mode == BANG_GRAPH_DEVICE);
  const size_t slots_cap = std::max<size_t>(nq, 8 * 16 * KT_WGS);   // rows / parent words: one per query, or one per context slot of the search kernel
  if (e->stage_zero_copy < 0) {                          // CPU-writable device memory (large BAR)?
    int large_bar = 0;
    int dev_id = 0;
    (void)hipGetDevice(&dev_id);
    if (hipDeviceGetAttribute(&large_bar, hipDeviceAttributeIsLargeBar, dev_id) != hipSuccess) large_bar = 0;
    e->stage_mode_eff = large_bar ? 2 : 1;
  } else e->stage_mode_eff = e->stage_zero_copy;
  // BAR mode: the CPU writes the staged rows into device memory while kernels read them.  That is only sound on FINE-GRAINED
  // (host-coherent) device memory: on ordinary coarse-grained memory PCIe writes do not probe the per-XCD L2, so a persistent
  // launch (no kernel boundary between two reads of a row) could be served a stale line.  No fine-grained memory -> no BAR mode.
  e->stage_local = false;
  if (!dev_graph && e->stage_mode_eff == 2) {
    if (hipExtMallocWithFlags((void**)&e->d_stage, slots_cap * BANG_STAGE_STRIDE * 4, hipDeviceMallocFinegrained) == hipSuccess) {
      e->stage_local = true;
    } else {
      (void)hipGetLastError();
      e->d_stage = nullptr;
      e->stage_mode_eff = 1;
      fprintf(stderr, "[bang] fine-grained device memory unavailable: staged rows stay in mapped host memory (stage_zero_copy=1)\n");
    }
  }
  // persistent search kernel: host graph, in-kernel completion flags, rows readable in place (BAR or zero-copy), and room in
  // LDS for the pivot table plus the merge scratch of all waves (otherwise: the launch-per-iteration loop)
  // (mapped-host rows need cache-bypassing loads, which are issued per lane: measured 2x slower than the per-iteration loop,
  // so "auto" takes the persistent kernel only in BAR mode)
  const bool persist_want = e->persistent < 0 ? (e->stage_mode_eff == 2) : (e->persistent != 0);   // host graph: one launch per batch?
  e->pq_nhi = 0;
  // graph in HBM: the query-resident search kernel, with whichever pivot table (padded / exact-size) leaves LDS for more waves
  e->search_v2 = false;
  if ((dev_graph || e->pull) && e->persistent != 0 && e->search_opt != 0 && e->psz != 0) {
    const int w_pad = bang_search_supported(e->psz, e->mp, 0, (uint32_t)e->L);
    const int w_rag = e->pq_nhi_avail ? bang_search_supported(e->psz, e->mp, e->pq_nhi_avail, (uint32_t)e->L) : 0;
    if (std::max(w_pad, w_rag) >= (e->search_opt == 1 ? 1 : 4)) {
      e->search_v2 = true;
      e->pq_nhi = (w_rag > w_pad) ? e->pq_nhi_avail : 0;
    }
  }
  // graph in host RAM: the host-paced form of the same kernel, where the walker can write device memory (BAR mode)
  e->search_host = false;
  if (!dev_graph && !e->search_v2 && persist_want && e->use_flag && e->stage_mode_eff == 2 && e->search_opt != 0 && e->psz != 0) {
    const int w_pad = bang_search_supported(e->psz, e->mp, 0, (uint32_t)e->L);
    const int w_rag = e->pq_nhi_avail ? bang_search_supported(e->psz, e->mp, e->pq_nhi_avail, (uint32_t)e->L) : 0;
    if (std::max(w_pad, w_rag) >= (e->search_opt == 1 ? 1 : 4)) {
      e->search_host = true;
      e->pq_nhi = (w_rag > w_pad) ? e->pq_nhi_avail : 0;
    }
  }
  // a walker form on an index whose graph entries only passed through at load time: map the graph file now (a streamed load
  // from an entry source has nothing to map: error)
  if (!dev_graph && !e->search_v2 && !e->graph) BANG_TRY(map_graph_file(e));
  e->fp_direct = false;
  HIP_TRY(hipMalloc(&e->d_queries, nq * e->D * e->tsize + 16));
  if (e->psz) BANG_TRY(dmalloc(&e->d_qc, nq * e->mp * e->psz));
  else BANG_TRY(dmalloc(&e->d_lut, nq * e->m * 256));                       // :380
  BANG_TRY(dmalloc(&e->d_bloom, nq * BANG_BF_WORDS));                        // :393 (bit-packed: 8x smaller)
  BANG_TRY(dmalloc(&e->d_nbrs, nq * BANG_NBR_STRIDE));
  BANG_TRY(dmalloc(&e->d_dist, nq * BANG_NBR_STRIDE));
  BANG_TRY(dmalloc(&e->d_cnt, nq));
  BANG_TRY(dmalloc(&e->d_wl_ids, nq * L));
  BANG_TRY(dmalloc(&e->d_wl_dist, nq * L));
  BANG_TRY(dmalloc(&e->d_wl_vis, nq * L));
  BANG_TRY(dmalloc(&e->d_wl_cnt, nq));
  BANG_TRY(dmalloc(&e->d_mark, nq));
  BANG_TRY(dmalloc(&e->d_cand_ids, nq * rows));
  BANG_TRY(dmalloc(&e->d_cand_cnt, nq));
  BANG_TRY(dmalloc(&e->d_qstats, nq * 2));
  BANG_TRY(dmalloc(&e->d_qskip, nq));
  {
    const size_t a64 = 63;
    e->res_off_dists = ((size_t)nq * e->k * 8 + a64) & ~a64;
    e->res_off_iters = (e->res_off_dists + (size_t)nq * e->k * 4 + a64) & ~a64;
    e->res_bytes = (e->res_off_iters + (size_t)nq * 4 + BANG_MAX_LANES * 4 + a64) & ~a64;   // + the kernel's abort word, one per lane
    BANG_TRY(dmalloc(&e->d_results, e->res_bytes));
    HIP_TRY(hipHostMalloc((void**)&e->h_results, e->res_bytes, hipHostMallocDefault));
    e->d_ids_out = (uint64_t*)e->d_results;
    e->d_dists_out = (float*)(e->d_results + e->res_off_dists);
    e->d_qiters = (uint32_t*)(e->d_results + e->res_off_iters);
    e->h_qiters.assign(nq, 0);
  }
  BANG_TRY(dmalloc(&e->d_parents_dev, nq));
  if (dev_graph) {
    BANG_TRY(dmalloc(&e->d_active, rows + 2));
  } else {
    BANG_TRY(dmalloc(&e->d_cand_row, nq * rows));
    if (e->vec_on_device) {
      e->d_fp = nullptr;                                                     // the re-rank reads d_vecs
    } else if (e->search_host && e->stage_mode_eff == 2 &&
        hipExtMallocWithFlags((void**)&e->d_fp, rows * nq * vb, hipDeviceMallocFinegrained) == hipSuccess) {
      e->fp_direct = true;                                                   // walker threads write the vector log through the BAR
    } else {
      (void)hipGetLastError();
      HIP_TRY(hipMalloc((void**)&e->d_fp, rows * nq * vb));                  // :398
    }
    HIP_TRY(hipHostMalloc((void**)&e->h_parents, slots_cap * 4, hipHostMallocMapped));       // :419
    HIP_TRY(hipHostGetDevicePointer((void**)&e->d_parents_map, e->h_parents, 0));
    if (e->search_host) {
      if (hipExtMallocWithFlags((void**)&e->d_srows, 8 * KT_WGS * 16 * 64 * 4, hipDeviceMallocFinegrained) != hipSuccess ||
          hipExtMallocWithFlags((void**)&e->d_sctl, 8 * KT_WGS * 64, hipDeviceMallocFinegrained) != hipSuccess) {
        bang_set_error("fine-grained device memory for the search kernel's staging rows: %s", hipGetErrorString(hipGetLastError()));
        return BANG_ERR_HIP;
      }
      HIP_TRY(hipMemset(e->d_srows, 0, 8 * KT_WGS * 16 * 64 * 4));
      HIP_TRY(hipMemset(e->d_sctl, 0, 8 * KT_WGS * 64));
    }
    if (e->search_host && !e->vec_on_device) {
      HIP_TRY(hipHostMalloc((void**)&e->h_pub_q, 8 * 16 * KT_WGS * 4, hipHostMallocMapped));
      HIP_TRY(hipHostGetDevicePointer((void**)&e->d_pub_q, e->h_pub_q, 0));
      HIP_TRY(hipHostMalloc((void**)&e->h_pub_c, 8 * 16 * KT_WGS * 4, hipHostMallocMapped));
      HIP_TRY(hipHostGetDevicePointer((void**)&e->d_pub_c, e->h_pub_c, 0));
    }
    HIP_TRY(hipHostMalloc((void**)&e->h_stage, nq * BANG_STAGE_STRIDE * 4, hipHostMallocMapped));      // :416
    HIP_TRY(hipHostGetDevicePointer((void**)&e->h_stage_dev, e->h_stage, 0));
    if (!e->d_stage) BANG_TRY(dmalloc(&e->d_stage, slots_cap * BANG_STAGE_STRIDE));   // stage modes 0/1: filled by H2D copies / unused
    HIP_TRY(hipMemset(e->d_stage, 0, slots_cap * BANG_STAGE_STRIDE * 4));
    memset(e->h_stage, 0, nq * BANG_STAGE_STRIDE * 4);
    if (!e->vec_on_device) HIP_TRY(hipHostMalloc((void**)&e->h_fp, rows * nq * vb, hipHostMallocDefault));          // :422
  }
  e->h_fin.assign((size_t)Q, 0);
  int nl = e->lanes_opt;
  if (nl <= 0) nl = dev_graph ? 1 : std::max(1, std::min(4, Q / 512));   // measured best on a 16-CPU-quota MI355X box
  nl = std::min(nl, Q);
  if (e->search_v2 || e->search_host) nl = 1;            // the search kernel's waves are the unit of overlap, not lanes
  if (e->search_host && e->threads_opt <= 0) e->threads_eff = std::max(1, std::min(12, usable_cpus() - 2));
  else if (e->threads_opt <= 0) e->threads_eff = (dev_graph || e->search_v2) ? 1 : std::max(1, std::min(4, (usable_cpus() - 2) / std::max(1, nl)));   // leave 2 CPUs for the caller + HIP runtime threads: a cgroup that exceeds its quota gets throttled for the rest of the period
  else e->threads_eff = e->threads_opt;
  if (!dev_graph) {
    const size_t n_flags = std::max<size_t>((size_t)nl, e->search_host ? 8 * KT_WGS : 0);
    HIP_TRY(hipHostMalloc((void**)&e->h_done, n_flags * 16 * 4, hipHostMallocMapped));
    HIP_TRY(hipHostGetDevicePointer((void**)&e->h_done_dev, e->h_done, 0));
    memset(e->h_done, 0, n_flags * 16 * 4);
    BANG_TRY(dmalloc(&e->d_done_count, (size_t)nl * 16));
    HIP_TRY(hipMemset(e->d_done_count, 0, (size_t)nl * 16 * 4));
  }
  e->lanes.clear();
  for (int i = 0; i < nl; ++i) e->lanes.emplace_back(new Lane());
  {
    int dev_id = 0, cus = 256;
    hipDeviceProp_t prop;
    if (hipGetDevice(&dev_id) == hipSuccess && hipGetDeviceProperties(&prop, dev_id) == hipSuccess) cus = prop.multiProcessorCount;
    // lanes are rarely all in their kernel phase at once: give each up to twice its fair share of the CUs
    e->front_wgs = e->front_wgs_opt >= 0 ? e->front_wgs_opt : (nl > 1 ? std::min(cus, std::max(1, 2 * cus / nl)) : 0);
  }
  for (int i = 0; i < nl; ++i) {
    Lane& ln = *e->lanes[(size_t)i];
    ln.index = i;
    HIP_TRY(hipStreamCreateWithFlags(&ln.s_main, hipStreamNonBlocking));     // :407-410
    HIP_TRY(hipStreamCreateWithFlags(&ln.s_fp, hipStreamNonBlocking));
    HIP_TRY(hipEventCreateWithFlags(&ln.ev_front, hipEventDisableTiming));
    HIP_TRY(hipEventCreateWithFlags(&ln.ev_fp, hipEventDisableTiming));
    for (int b = 0; b < 2; ++b) {                          // slot -> query maps for straggler compaction
      const size_t bytes = std::max<size_t>((size_t)Q * 4, 64);   // a lane never owns more than Q queries
      if (e->stage_mode_eff == 2 && !dev_graph &&
          hipExtMallocWithFlags((void**)&ln.qmap_host[b], bytes, hipDeviceMallocFinegrained) == hipSuccess) {
        ln.qmap_dev[b] = ln.qmap_host[b];
        ln.qmap_is_device = true;
      } else {
        (void)hipGetLastError();
        if (ln.qmap_is_device) { bang_set_error("qmap allocation failed"); return BANG_ERR_HIP; }
        HIP_TRY(hipHostMalloc((void**)&ln.qmap_host[b], bytes, hipHostMallocMapped));
        HIP_TRY(hipHostGetDevicePointer((void**)&ln.qmap_dev[b], ln.qmap_host[b], 0));
      }
    }
    if (e->search_v2 || e->search_host) {
      BANG_TRY(dmalloc(&ln.d_pcnt, 16));
      HIP_TRY(hipMemset(ln.d_pcnt, 0, 64));
    }
    if (e->search_host) ln.pw_expect.reset(new std::atomic<uint32_t>[8 * KT_WGS]);
    if (e->timing) {
      ln.kt_launches = rows + 4;
      HIP_TRY(hipMalloc((void**)&ln.d_ktime, ln.kt_launches * KT_WGS * 16));     // {start, end} stamp per workgroup and launch
      HIP_TRY(hipMemset(ln.d_ktime, 0, ln.kt_launches * KT_WGS * 16));
    }
  }
  e->numa_on = false;
  if (!dev_graph && e->numa_opt != 0) {
    // "auto" = off: on the measured box (2 x EPYC 9575F, 16-CPU cgroup quota) pinning the 12 walker threads to the GPU's node
    // was 6 % SLOWER than letting the scheduler spread them over both sockets; the option is there for hosts where it pays.
    e->numa_on = e->numa_opt == 1 && gpu_numa_cpus(e->device, &e->numa_cpus, &e->numa_node);
    e->numa_cores.clear();
    if (e->numa_on) {
      e->numa_cores = distinct_cores(e->numa_cpus);
      if ((int)e->numa_cores.size() < e->threads_eff * nl) e->numa_cores.clear();       // not enough cores: node-wide mask instead
    }
  }
  if (getenv("BANG_DEBUG")) {
    if (e->numa_on) fprintf(stderr, "[bang] walker threads pinned to NUMA node %d (%d usable CPUs)\n", e->numa_node, CPU_COUNT(&e->numa_cpus));
    else if (!dev_graph) fprintf(stderr, "[bang] walker threads not pinned\n");
  }
  if (getenv("BANG_DEBUG"))
    fprintf(stderr, "[bang] alloc Q=%d lanes=%d threads=%d stage_mode=%d search_kernel=%d/%d fp_direct=%d vec_on_device=%d\n", Q, nl,
            e->threads_eff, e->stage_mode_eff, (int)e->search_v2, (int)e->search_host, (int)e->fp_direct, (int)e->vec_on_device);
  start_threads(e);
  return BANG_OK;
}

extern "C" int bang_alloc_e(bang_engine_t* e, int Q) {
  if (!e || Q <= 0) return BANG_ERR_ARG;
  if (!e->loaded || !e->params_set) { bang_set_error("bang_alloc: load an index and set search params first"); return BANG_ERR_ARG; }
  if (e->allocated) { bang_set_error("bang_alloc: already allocated (call bang_free)"); return BANG_ERR_ARG; }
  BANG_TRY(ensure_device(e));
  e->Qcap = Q;
  e->Qcur = 0;
  e->cand_stride = (uint32_t)e->L + BANG_EXTRA_ITERS;
  const int rc = alloc_buffers(e, Q);
  if (rc != BANG_OK) { free_batch(e); return rc; }
  e->allocated = true;
  e->inited = false;
  return BANG_OK;
}

extern "C" int bang_init_e(bang_engine_t* e, int Q) {
  if (!e || !e->allocated || Q <= 0 || Q > e->Qcap) { bang_set_error("bang_init: bad state / numQueries"); return BANG_ERR_ARG; }
  BANG_TRY(ensure_device(e));
  const size_t nq = (size_t)Q;
  HIP_TRY(hipMemsetAsync(e->d_bloom, 0, nq * BANG_BF_WORDS * 4, nullptr));                  // :443
  HIP_TRY(hipMemsetAsync(e->d_qstats, 0, nq * 8, nullptr));
  HIP_TRY(hipMemsetAsync(e->d_qskip, 0, nq * 4, nullptr));
  if (e->d_active) HIP_TRY(hipMemsetAsync(e->d_active, 0, ((size_t)e->cand_stride + 2) * 4, nullptr));
  BANG_TRY(bang_k_init_state((uint32_t)Q, (uint32_t)e->medoid, e->cand_stride, e->d_cand_ids, e->d_cand_row, e->d_cand_cnt,
                             e->d_wl_cnt, e->d_mark, e->d_parents_dev, e->d_cnt, nullptr));
  HIP_TRY(hipDeviceSynchronize());
  if (e->h_parents) for (size_t i = 0; i < nq; ++i) e->h_parents[i] = BANG_NO_PARENT;
  e->inited = true;
  return BANG_OK;
}

static int query_impl(bang_engine_t* e, const void* h_queries, int Q, uint64_t* h_ids, float* h_dists, uint64_t* d_ids_user, float* d_dists_user);

extern "C" int bang_query_e(bang_engine_t* e, const void* h_queries, int Q, uint64_t* h_ids, float* h_dists) {
  if (!e || !h_queries || !h_ids || !h_dists) return BANG_ERR_ARG;
  return query_impl(e, h_queries, Q, h_ids, h_dists, nullptr, nullptr);
}

extern "C" int bang_query_dev_e(bang_engine_t* e, const void* h_queries, int Q, uint64_t* d_ids, float* d_dists) {
  if (!e || !h_queries || !d_ids) return BANG_ERR_ARG;
  return query_impl(e, h_queries, Q, nullptr, nullptr, d_ids, d_dists);
}

static int query_impl(bang_engine_t* e, const void* h_queries, int Q, uint64_t* h_ids, float* h_dists, uint64_t* d_ids_user, float* d_dists_user) {
  if (!e->allocated || !e->inited) { bang_set_error("bang_query: bang_alloc + bang_init must precede every query"); return BANG_ERR_ARG; }
  if (Q <= 0 || Q > e->Qcap) { bang_set_error("bang_query: numQueries %d exceeds allocation %d", Q, e->Qcap); return BANG_ERR_ARG; }
  e->inited = false;   // state is consumed
  e->Qcur = Q;
  // the calling thread is lane 0's walker: it joins the GPU's NUMA node for the duration of the query
  cpu_set_t caller_cpus;
  const bool repin = e->numa_on && sched_getaffinity(0, sizeof(caller_cpus), &caller_cpus) == 0;
  if (repin) pin_walker_thread(e, 0);
  const auto t0 = Clock::now();
  const int nl = (int)e->lanes.size();
  for (int i = 0; i < nl; ++i) {                       // lanes were laid out for Qcap; re-slice for this Q
    Lane& ln = *e->lanes[(size_t)i];
    ln.q0 = (uint32_t)((size_t)Q * i / nl);
    ln.nq = (uint32_t)((size_t)Q * (i + 1) / nl) - ln.q0;
  }
  std::atomic<bool> wd_stop{false};
  std::thread wd;
  if (getenv("BANG_WATCHDOG")) {
    wd = std::thread([&] {
      FILE* wf = fopen(getenv("BANG_WATCHDOG"), "a");
      if (!wf) wf = stderr;
      int ticks = 0;
      while (!wd_stop.load()) {
        std::this_thread::sleep_for(std::chrono::milliseconds(100));
        if (++ticks % 50 == 0) {
          for (auto& lp : e->lanes)
            fprintf(wf, "[watchdog] lane %d phase %d iter %u pending %u epoch %u flag %u active %d\n", lp->index, lp->phase.load(),
                    lp->phase_iter.load(), lp->pending.load(), lp->epoch.load(), e->h_done ? e->h_done[(size_t)lp->index * 16] : 0u,
                    (int)lp->team_active.load());
          fflush(wf);
        }
      }
      if (wf != stderr) fclose(wf);
    });
  }
  Pool& pool = e->pool;
  {
    std::lock_guard<std::mutex> lk(pool.m);
    pool.h_queries = h_queries; pool.h_ids = h_ids; pool.h_dists = h_dists; pool.Q = Q;
    pool.d_ids_user = d_ids_user; pool.d_dists_user = d_dists_user;
    pool.lanes_done = 0;
    ++pool.query_seq;
  }
  pool.cv_start.notify_all();
  lane_job(e, *e->lanes[0]);                           // lane 0 on the calling thread
  if (nl > 1) {
    std::unique_lock<std::mutex> lk(pool.m);
    pool.cv_done.wait(lk, [&] { return pool.lanes_done == nl - 1; });
  }
  if (wd.joinable()) { wd_stop.store(true); wd.join(); }
  if (repin) (void)sched_setaffinity(0, sizeof(caller_cpus), &caller_cpus);
  int rc = BANG_OK;
  for (auto& lp : e->lanes)
    if (lp->rc != BANG_OK) { rc = lp->rc; bang_set_error("%s", lp->err.c_str()); break; }
  bang_stats& s = e->stats;
  memset(&s, 0, sizeof(s));
  s.wall_ms = ms_since(t0);
  for (auto& lp : e->lanes) {
    Lane& ln = *lp;
    s.iterations = std::max<uint64_t>(s.iterations, ln.iterations);
    s.front_launches += ln.front_launches;
    s.front_ms += ln.front_ms; s.back_ms += ln.back_ms; s.rerank_ms += ln.rerank_ms; s.walker_ms += ln.walker_ms;
    s.sync_ms += ln.sync_ms; s.enqueue_ms += ln.enqueue_ms;
    s.h2d_bytes += ln.h2d_bytes.load();
  }
  s.persistent = (e->search_v2 || e->search_host) ? 1 : 0;
  s.vectors_on_device = e->vec_on_device ? 1 : 0;
  s.graph_mode = (uint64_t)e->graph_mode;
  s.lanes = (uint64_t)nl;
  s.walker_threads = (e->graph_mode == BANG_GRAPH_DEVICE || e->search_v2) ? 0 : (uint64_t)e->threads_eff;   // (pull mode: nothing walks)
  s.wg_queries = e->search_host ? e->sv_W * e->sv_C : 0;
  s.pacing_groups = e->search_host ? e->sv_NG : 0;
  s.graph_pull = (e->pull && e->search_v2 && e->graph_mode != BANG_GRAPH_DEVICE) ? 1 : 0;
  s.workgroups = e->search_host ? (uint64_t)e->sv_G : e->search_v2 ? (uint64_t)std::min(Q, bang_num_cus()) : 0;
  s.search_kernel = (e->search_v2 || e->search_host) ? 1 : 0;
  return rc;
}

// reduce the in-kernel stamps of a lane: per launch max(end) - min(start) over the workgroups that ran
static int reduce_ktimes(Lane& ln, std::vector<std::pair<unsigned long long, unsigned long long>>& intervals) {
  if (!ln.d_ktime || ln.kt_used == 0) return BANG_OK;
  std::vector<unsigned long long> kt(ln.kt_used * KT_WGS * 2);
  HIP_TRY(hipMemcpy(kt.data(), ln.d_ktime, kt.size() * 8, hipMemcpyDeviceToHost));
  ln.front_ms = 0;
  for (size_t l = 0; l < ln.kt_used; ++l) {
    unsigned long long lo = ~0ull, hi = 0;
    for (size_t w = 0; w < KT_WGS; ++w) {
      const unsigned long long a = kt[(l * KT_WGS + w) * 2], b = kt[(l * KT_WGS + w) * 2 + 1];
      if (a == 0 || b == 0) continue;                  // workgroup slot not used by this launch
      lo = std::min(lo, a);
      hi = std::max(hi, b);
    }
    if (hi > lo) { ln.front_ms += (double)(hi - lo) * 1e-5; intervals.emplace_back(lo, hi); }   // 100 MHz ticks -> ms
  }
  if (const char* path = getenv("BANG_KT_TRACE")) {            // raw stamps of lane 0 for offline analysis
    if (ln.index == 0) if (FILE* f = fopen(path, "wb")) {
      const uint64_t hdr[2] = {(uint64_t)ln.kt_used, (uint64_t)KT_WGS};
      fwrite(hdr, 8, 2, f);
      fwrite(kt.data(), 8, kt.size(), f);
      fclose(f);
    }
  }
  HIP_TRY(hipMemset(ln.d_ktime, 0, ln.kt_used * KT_WGS * 16));
  ln.kt_used = 0;
  return BANG_OK;
}

extern "C" int bang_get_stats(bang_engine_t* e, bang_stats* out) {
  if (!e || !out) return BANG_ERR_ARG;
  bang_stats& s = e->stats;
  if (e->allocated && e->timing && s.front_ms == 0) {
    std::vector<std::pair<unsigned long long, unsigned long long>> iv;
    for (auto& lp : e->lanes) { BANG_TRY(reduce_ktimes(*lp, iv)); s.front_ms += lp->front_ms; }
    std::sort(iv.begin(), iv.end());                   // the stamps of all lanes share one 100 MHz clock: merge the intervals
    unsigned long long cur_lo = 0, cur_hi = 0, busy = 0;
    for (auto& p : iv) {
      if (p.first > cur_hi) { busy += cur_hi - cur_lo; cur_lo = p.first; cur_hi = p.second; }
      else cur_hi = std::max(cur_hi, p.second);
    }
    busy += cur_hi - cur_lo;
    s.front_busy_ms = (double)busy * 1e-5;
  }
  if (e->allocated && e->Qcur > 0 && s.candidates == 0) {   // device-side counters are fetched lazily
    std::vector<uint32_t> qs((size_t)e->Qcur * 2);
    HIP_TRY(hipMemcpy(qs.data(), e->d_qstats, qs.size() * 4, hipMemcpyDeviceToHost));
    for (size_t i = 0; i < qs.size(); i += 2) { s.dist_evals += qs[i]; s.fetched += qs[i + 1]; }
    if (e->search_v2) {
      std::vector<uint32_t> sk((size_t)e->Qcur);
      HIP_TRY(hipMemcpy(sk.data(), e->d_qskip, sk.size() * 4, hipMemcpyDeviceToHost));
      for (uint32_t v : sk) s.filter_loads_skipped += v;
    }
    std::vector<uint32_t> cc((size_t)e->Qcur);
    HIP_TRY(hipMemcpy(cc.data(), e->d_cand_cnt, cc.size() * 4, hipMemcpyDeviceToHost));
    for (uint32_t c : cc) s.candidates += c;
    std::sort(cc.begin(), cc.end());
    s.hops_p50 = cc[cc.size() / 2];
    s.hops_p99 = cc[std::min(cc.size() - 1, (cc.size() * 99) / 100)];
    s.hops_max = cc.back();
    if (s.graph_pull) s.pulled_bytes = (s.candidates - (uint64_t)e->Qcur) * 256;      // one row per expansion (the seed list is on the device)
  }
  *out = s;
  return BANG_OK;
}

extern "C" int bang_get_query_counters(bang_engine_t* e, uint32_t* dist_evals, uint32_t* fetched, uint32_t* candidates, uint32_t* iterations) {
  if (!e) return BANG_ERR_ARG;
  if (!e->allocated || e->Qcur <= 0) { bang_set_error("bang_get_query_counters: no query has run on this allocation"); return BANG_ERR_ARG; }
  const size_t Q = (size_t)e->Qcur;
  if (dist_evals || fetched) {
    std::vector<uint32_t> qs(Q * 2);
    HIP_TRY(hipMemcpy(qs.data(), e->d_qstats, qs.size() * 4, hipMemcpyDeviceToHost));
    for (size_t i = 0; i < Q; ++i) { if (dist_evals) dist_evals[i] = qs[2 * i]; if (fetched) fetched[i] = qs[2 * i + 1]; }
  }
  if (candidates) HIP_TRY(hipMemcpy(candidates, e->d_cand_cnt, Q * 4, hipMemcpyDeviceToHost));
  if (iterations) {
    if ((e->search_v2 || e->search_host) && e->h_qiters.size() >= Q) memcpy(iterations, e->h_qiters.data(), Q * 4);
    else memset(iterations, 0, Q * 4);
  }
  return BANG_OK;
}

extern "C" int bang_free_e(bang_engine_t* e) {
  if (!e) return BANG_ERR_ARG;
  if (e->allocated) { (void)hipSetDevice(e->device); free_batch(e); }
  return BANG_OK;
}

extern "C" int bang_unload_e(bang_engine_t* e) {
  if (!e) return BANG_ERR_ARG;
  if (e->allocated) free_batch(e);
  if (e->loaded) { (void)hipSetDevice(e->device); unload_index(e); }
  return BANG_OK;
}
