// bang_load.cpp -- bang_load (bang_search.cu:138-362): index files and entry sources, graph placement, pull rows, streamed load.
// Reference line numbers: /root/reference/BANG_Base/bang_search.cu.
#include "bang_engine.h"

namespace bang {

// threads but grant 16 CPUs; spinning walker threads beyond the quota only starve each other)
int usable_cpus() {
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
size_t host_bytes_available() {
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
  for (uint32_t k = deg; k < 64; ++k) row[k] = BANG_ADJ_PAD;
}

// Pull mode: the adjacency lists of the host graph, re-laid as 256-byte rows the GPU can fetch with one PCIe read each.
// One copy per NODE when BANG_PULL_ROWS_DIR names a directory every rank can see (tmpfs): the rows live in the file
// <dir>/<index name>_pull_rows.bin, built by whichever rank loads first (write to a temporary name, rename) and mapped shared by
// the others; every rank registers the mapping with its own device.  Without the variable: private anonymous memory.
struct PullRows {
  void* m = MAP_FAILED;
  size_t bytes = 0, sig_off = 0;
  bool fill = true;                    // false: an existing rows file was mapped (its signature is checked in pull_rows_finish)
  bool keep_stale = false;             // a mapped file with another signature is left alone (shared load: it is another rank's)
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
  if (const char* dir = env_str("BANG_PULL_ROWS_DIR"))
    pr.path = std::string(dir) + "/" + (e->rows_key.empty() ? std::string("index") : e->rows_key) + "_pull_rows.bin";
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
      // (a mapping of a tmpfs file that outgrows its file system dies with SIGBUS on first touch, not with an error: check first)
      struct statvfs vfs;
      if (fstatvfs(fd, &vfs) == 0 && (unsigned long long)vfs.f_bavail * vfs.f_frsize < pr.bytes + ((size_t)1 << 30)) {
        close(fd);
        (void)unlink(pr.tmp.c_str());
        pr.tmp.clear();
        bang_set_error("pull rows: %.1f GB do not fit the file system of %s", pr.bytes / 1e9, pr.path.c_str());
        return BANG_ERR_NOMEM;
      }
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
  // ONE rows file serves every GPU of the node: its pages are spread over the NUMA nodes page by page (first touch would place
  // them wherever the builder threads happen to run), so that the GPUs of both sockets pull from both sockets' memory controllers
  // and no socket's DRAM or inter-socket link carries all eight pullers.  (One GPU alone: binding the rows to its own node measured
  // +0.5 %)  BANG_PULL_ROWS_INTERLEAVE=0 leaves the placement to first touch.
  if (!pr.tmp.empty() && env_long("BANG_PULL_ROWS_INTERLEAVE", 1) != 0) {
    unsigned long mask[16] = {0};
    int nodes = 0;
    for (int n = 0; n < 1024; ++n) {
      char pth[64];
      snprintf(pth, sizeof(pth), "/sys/devices/system/node/node%d", n);
      if (access(pth, F_OK) != 0) break;
      mask[n / 64] |= 1ul << (n % 64);
      ++nodes;
    }
    if (nodes > 1) (void)syscall(SYS_mbind, pr.m, pr.bytes, 3 /* MPOL_INTERLEAVE */, mask, (unsigned long)(nodes + 1), 0u);
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
    if (!pr.keep_stale) (void)unlink(pr.path.c_str());    // stale: rows of another graph (the caller rebuilds them)
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
  e->rows_hash = sig.adj_hash;
  e->rows_path = pr.path;
  pr.m = MAP_FAILED;
  return BANG_OK;
}

// Pull mode, at bang_alloc: the self-paced kernel reads the rows by itself, no host thread watches it -- what it is about to read
// must still be what bang_load registered.  Checked: the mapping's size (N x 256 + trailer), that the address is (still) registered
// with this device, and -- for a rows FILE, which another process may have truncated or replaced since -- the file's size and the
// signature behind the last row (the size first: touching a mapping beyond a truncated file's end is a SIGBUS, not an error code).
int validate_pull_rows(bang_engine* e) {
  if (!e->pull) return BANG_OK;
  const size_t want = (size_t)e->N * 256 + 4096;
  if (!e->h_adj || !e->d_adj || e->adj_bytes != want) {
    bang_set_error("pull rows: the mapping holds %zu bytes, the index needs %zu (N = %u rows of 256 bytes + trailer)", e->adj_bytes, want, e->N);
    return BANG_ERR_ARG;
  }
  hipPointerAttribute_t at;
  if (hipPointerGetAttributes(&at, e->h_adj) != hipSuccess || at.devicePointer == nullptr) {
    (void)hipGetLastError();
    bang_set_error("pull rows: the host mapping is not registered with the device any more");
    return BANG_ERR_HIP;
  }
  if (!e->rows_path.empty()) {
    struct stat st;
    if (stat(e->rows_path.c_str(), &st) == 0 && (size_t)st.st_size != want) {      // (an unlinked file is fine: the mapping keeps its pages)
      bang_set_error("pull rows: %s has been truncated or replaced since bang_load (%lld bytes, %zu expected): unload and load again",
                     e->rows_path.c_str(), (long long)st.st_size, want);
      return BANG_ERR_IO;
    }
  }
  PullRowsSig g;
  memcpy(&g, (const uint8_t*)e->h_adj + (size_t)e->N * 256 + 2048, sizeof(g));
  if (memcmp(g.magic, "BANGROW2", 8) != 0 || g.N != e->N || g.medoid != e->medoid || g.adj_hash != e->rows_hash) {
    bang_set_error("pull rows: the signature behind the last row no longer matches the loaded index (rows overwritten since bang_load)");
    return BANG_ERR_IO;
  }
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

// Pull mode, after everything else of the index is in place: HBM the index leaves over takes a copy of the first adjacency rows.
// The link is the next limiter of the pull mode (SIFT1B-shape: 47-49 of 57 GB/s): every row that sits in HBM is a PCIe read less.
static int cache_rows_in_hbm(bang_engine* e) {
  e->n_rows_hbm = 0;
  if (!e->pull || !e->h_adj || e->rows_hbm_opt == 0) return BANG_OK;
  size_t free_b = 0, total_b = 0;
  (void)hipMemGetInfo(&free_b, &total_b);
  const size_t keep = (size_t)6 << 30;                                    // per-batch state (filters: 50 KB per query; 0.6 GB at 10 K queries) + slack; a batch
                                                                          // that needs more gets it: bang_alloc drops this copy and retries (bang_cabi.cpp)
  size_t budget = free_b > keep ? free_b - keep : 0;
  if (e->rows_hbm_opt > 0) budget = std::min(budget, (size_t)e->rows_hbm_opt << 20);
  size_t n = std::min<size_t>(e->N, budget / 256);
  // auto: only where the rows do NOT all fit -- an index small enough to sit in HBM whole is on the host because the caller put
  // it there (graph = host), and graph = auto would have placed it in HBM anyway
  if (e->rows_hbm_opt < 0 && n >= e->N) return BANG_OK;
  n = std::min<size_t>(n, (size_t)std::max(0L, env_long("BANG_ROWS_HBM_MAX_ROWS", 1L << 40)));      // (test hook: a partial copy of a small index)
  if (n < 64) return BANG_OK;
  if (hipMalloc((void**)&e->d_rows_hbm, n * 256) != hipSuccess) { (void)hipGetLastError(); e->d_rows_hbm = nullptr; return BANG_OK; }
  const size_t step = (size_t)1 << 30;
  for (size_t off = 0; off < n * 256; off += step)
    if (hipMemcpy((uint8_t*)e->d_rows_hbm + off, (const uint8_t*)e->h_adj + off, std::min(step, n * 256 - off), hipMemcpyHostToDevice) != hipSuccess) {
      (void)hipGetLastError();                             // the copy is an optimisation: without it every row is pulled over PCIe, nothing fails
      dfree(e->d_rows_hbm);
      return BANG_OK;
    }
  e->n_rows_hbm = (uint32_t)n;
  if (env_flag("BANG_DEBUG")) fprintf(stderr, "[bang] %zu of %u adjacency rows (%.1f GB) also in HBM\n", n, e->N, n * 256 / 1e9);
  return BANG_OK;
}

// ---------------------------------------------------------------------------------------------------------------- peer rows
// (include/bang_c.h: the node's adjacency rows in the node's spare HBM, read over xGMI)
static const size_t kRowsKeepBack = (size_t)6 << 30;     // HBM left to the per-batch state, as cache_rows_in_hbm does

}  // namespace bang
using namespace bang;

extern "C" int bang_get_num_nodes(bang_engine_t* e, uint64_t* nodes_out) {
  if (!e || !nodes_out) return BANG_ERR_ARG;
  if (!e->loaded) { bang_set_error("bang_get_num_nodes: no index is loaded"); return BANG_ERR_ARG; }
  *nodes_out = e->N;
  return BANG_OK;
}

extern "C" int bang_rows_capacity_e(bang_engine_t* e, uint64_t* rows_out) {
  if (!e || !rows_out) return BANG_ERR_ARG;
  if (!e->loaded || !e->pull || !e->h_adj) { bang_set_error("peer rows: the index is not loaded in pull mode"); return BANG_ERR_ARG; }
  BANG_TRY(ensure_device(e));
  size_t free_b = 0, total_b = 0;
  HIP_TRY(hipMemGetInfo(&free_b, &total_b));
  const size_t have = (size_t)e->n_rows_hbm * 256;
  *rows_out = ((free_b > kRowsKeepBack ? free_b - kRowsKeepBack : 0) + have) / 256;
  return BANG_OK;
}

extern "C" int bang_rows_slice_e(bang_engine_t* e, uint64_t first_row, uint64_t rows) {
  if (!e) return BANG_ERR_ARG;
  if (!e->loaded || !e->pull || !e->h_adj) { bang_set_error("peer rows: the index is not loaded in pull mode"); return BANG_ERR_ARG; }
  if (e->allocated) { bang_set_error("peer rows: set the slice before bang_alloc"); return BANG_ERR_ARG; }
  if (e->rows_exported || e->n_slices) { bang_set_error("peer rows: the slice has been exported / the slice table is set: unload first"); return BANG_ERR_ARG; }
  BANG_TRY(ensure_device(e));
  if (first_row > e->N) first_row = e->N;
  if (rows > e->N - first_row) rows = e->N - first_row;
  dfree(e->d_rows_hbm);
  e->n_rows_hbm = 0; e->rows_first = 0;
  if (rows == 0) return BANG_OK;
  BANG_TRY(dmalloc(&e->d_rows_hbm, (size_t)rows * 64));
  const size_t step = (size_t)1 << 30, bytes = (size_t)rows * 256;
  const uint8_t* src = (const uint8_t*)e->h_adj + (size_t)first_row * 256;
  for (size_t off = 0; off < bytes; off += step) HIP_TRY(hipMemcpy((uint8_t*)e->d_rows_hbm + off, src + off, std::min(step, bytes - off), hipMemcpyHostToDevice));
  e->n_rows_hbm = (uint32_t)rows; e->rows_first = first_row;
  return BANG_OK;
}

extern "C" int bang_rows_export_e(bang_engine_t* e, void* handle64, uint64_t* first_row, uint64_t* rows) {
  if (!e || !handle64 || !first_row || !rows) return BANG_ERR_ARG;
  static_assert(sizeof(hipIpcMemHandle_t) == 64, "the handle travels as 64 bytes");
  memset(handle64, 0, 64);
  *first_row = e->rows_first; *rows = e->n_rows_hbm;
  if (!e->d_rows_hbm || e->n_rows_hbm == 0) return BANG_OK;             // nothing to share: the peers read these rows from the host
  BANG_TRY(ensure_device(e));
  hipIpcMemHandle_t h;
  HIP_TRY(hipIpcGetMemHandle(&h, e->d_rows_hbm));
  memcpy(handle64, &h, 64);
  e->rows_exported = true;
  return BANG_OK;
}

extern "C" int bang_rows_import_e(bang_engine_t* e, uint32_t slot, uint32_t n_slots, uint64_t slice_rows, uint64_t rows, const void* handle64) {
  if (!e) return BANG_ERR_ARG;
  if (!e->loaded || !e->pull) { bang_set_error("peer rows: the index is not loaded in pull mode"); return BANG_ERR_ARG; }
  if (e->allocated) { bang_set_error("peer rows: import before bang_alloc"); return BANG_ERR_ARG; }
  if (n_slots == 0 || n_slots > BANG_MAX_ROW_SLICES || slot >= n_slots || slice_rows == 0 || slice_rows > 0xFFFFFFFFull) { bang_set_error("peer rows: bad slot / slice size"); return BANG_ERR_ARG; }
  if (e->n_slices && (e->n_slices != n_slots || e->slice_rows != (uint32_t)slice_rows)) { bang_set_error("peer rows: every slot of one table has the same geometry"); return BANG_ERR_ARG; }
  BANG_TRY(ensure_device(e));
  const uint64_t first = (uint64_t)slot * slice_rows;
  uint64_t base = 0;
  if (!handle64) {                                                         // this engine's own slice
    if (e->d_rows_hbm && e->rows_first == first && e->n_rows_hbm >= std::min<uint64_t>(slice_rows, e->N > first ? e->N - first : 0)) base = (uint64_t)(uintptr_t)e->d_rows_hbm;
    else if (e->d_rows_hbm) { bang_set_error("peer rows: slot %u is not the slice this engine holds (rows [%llu, +%u))", slot, (unsigned long long)e->rows_first, e->n_rows_hbm); return BANG_ERR_ARG; }
    e->own_slot = slot;
  } else {
    bool any = false;
    for (int i = 0; i < 64; ++i) any |= ((const uint8_t*)handle64)[i] != 0;
    if (any) {                                                             // (an all-zero handle: that rank holds no rows -- host rows serve the slot)
      // the kernel reads base + p * 256 for EVERY p of the slot: a peer that exported fewer rows than the slot spans (a caller slicing by its own
      // capacity instead of the node minimum) would be read past its allocation, over xGMI, undetected
      const uint64_t need = std::min<uint64_t>(slice_rows, e->N > first ? e->N - first : 0);
      if (rows < need) { bang_set_error("peer rows: slot %u was exported with %llu rows, its slice spans %llu", slot, (unsigned long long)rows, (unsigned long long)need); return BANG_ERR_ARG; }
      hipIpcMemHandle_t h;
      memcpy(&h, handle64, 64);
      void* ptr = nullptr;
      HIP_TRY(hipIpcOpenMemHandle(&ptr, h, hipIpcMemLazyEnablePeerAccess));
      if (e->peer_ptr[slot]) (void)hipIpcCloseMemHandle(e->peer_ptr[slot]);
      e->peer_ptr[slot] = ptr;
      base = (uint64_t)(uintptr_t)ptr;
    }
  }
  e->n_slices = n_slots; e->slice_rows = (uint32_t)slice_rows;
  e->slice_base[slot] = base ? base - first * 256ull : 0;                  // biased: + p * 256 is node p's row
  return BANG_OK;
}

extern "C" int bang_rows_close_peers_e(bang_engine_t* e) {
  if (!e) return BANG_ERR_ARG;
  if (e->allocated) { bang_set_error("peer rows: bang_free first (a batch may still read the peers' rows)"); return BANG_ERR_ARG; }
  BANG_TRY(ensure_device(e));
  for (uint32_t s = 0; s < BANG_MAX_ROW_SLICES; ++s) {
    if (e->peer_ptr[s]) { (void)hipIpcCloseMemHandle(e->peer_ptr[s]); e->peer_ptr[s] = nullptr; e->slice_base[s] = 0; }
  }
  return BANG_OK;
}

namespace bang {

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
  if (e->ext_vecs) { e->d_vecs = e->ext_vecs; e->vecs_owned = false; }      // the caller's buffer: it can hand the vectors on (xGMI broadcast)
  else { HIP_TRY(hipMalloc((void**)&e->d_vecs, N * vb + 256)); e->vecs_owned = true; }
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
      if (e->vecs_owned) dfree(e->d_vecs);
      e->d_vecs = nullptr;
      return stage_entries_streamed(e, true);
    }
    if (frc != BANG_OK) return frc == BANG_ERR_STALE_ROWS ? BANG_ERR_IO : frc;
  }
  e->vec_on_device = true;
  BANG_TRY(stage_medoid(e, medoid_entry.data()));
  e->graph_streamed = true;
  return cache_rows_in_hbm(e);
}

// SHARED load (ranks > 0 of a multi-GPU node): the rank that read the index has written the pull rows file and handed the
// full-precision vectors on from its HBM (an RCCL broadcast over xGMI into the caller's buffer); nothing is read from the index
// here -- the rows file is mapped (its signature must carry the hash the loading rank reports), the seed list comes from the
// medoid's row, the medoid's vector from the vector buffer.
int load_shared(bang_engine* e, uint64_t expect_rows_hash) {
  const size_t vb = vec_bytes(e);
  if (!e->ext_vecs || !e->ext_vecs_ready) { bang_set_error("shared load: no filled vector buffer"); return BANG_ERR_ARG; }
  if (e->R > 64) { bang_set_error("shared load: the pull rows hold at most 64 ids (R = %u)", e->R); return BANG_ERR_UNSUPPORTED; }
  const PullRowsSig sig = sig_make(e, expect_rows_hash);
  PullRows pr;
  pr.bytes = (size_t)e->N * 256 + 4096;
  pr.sig_off = (size_t)e->N * 256 + 2048;
  const char* dir = env_str("BANG_PULL_ROWS_DIR");
  if (!dir) { bang_set_error("shared load: BANG_PULL_ROWS_DIR is not set"); return BANG_ERR_ARG; }
  pr.path = std::string(dir) + "/" + (e->rows_key.empty() ? std::string("index") : e->rows_key) + "_pull_rows.bin";
  const int fd = open(pr.path.c_str(), O_RDWR);
  if (fd < 0) { bang_set_error("shared load: cannot open %s", pr.path.c_str()); return BANG_ERR_IO; }
  struct stat st;
  if (fstat(fd, &st) != 0 || (size_t)st.st_size != pr.bytes) { close(fd); bang_set_error("shared load: %s has the wrong size", pr.path.c_str()); return BANG_ERR_IO; }
  pr.m = mmap(nullptr, pr.bytes, PROT_READ | PROT_WRITE, MAP_SHARED, fd, 0);
  close(fd);
  if (pr.m == MAP_FAILED) { bang_set_error("shared load: cannot map %s", pr.path.c_str()); return BANG_ERR_NOMEM; }
  pr.fill = false;
  pr.keep_stale = true;
  {
    const int rc = pull_rows_finish(e, pr, sig);       // compares the signature (sizes, medoid, hash of every adjacency list)
    if (rc != BANG_OK) return rc == BANG_ERR_STALE_ROWS ? BANG_ERR_IO : rc;
  }
  e->d_vecs = e->ext_vecs; e->vecs_owned = false; e->vec_on_device = true;
  std::vector<uint8_t> me(e->entry_len, 0);
  HIP_TRY(hipMemcpy(me.data(), e->d_vecs + (size_t)e->medoid * vb, vb, hipMemcpyDeviceToHost));
  const uint32_t* row = e->h_adj + (size_t)e->medoid * 64;
  uint32_t deg = 0;
  while (deg < e->R && deg < 64 && row[deg] != 0xFFFFFFFFu) ++deg;
  memcpy(me.data() + vb, &deg, 4);
  memcpy(me.data() + vb + 4, row, (size_t)deg * 4);
  BANG_TRY(stage_medoid(e, me.data()));
  e->graph_streamed = true;
  return cache_rows_in_hbm(e);
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
int map_graph_file(bang_engine* e) {
  if (e->graph) return BANG_OK;
  if (e->graph_path.empty()) { bang_set_error("the graph entries were streamed at load time and are not resident: only the pull mode can run"); return BANG_ERR_UNSUPPORTED; }
  const int fd = open(e->graph_path.c_str(), O_RDONLY);
  if (fd < 0) { printf("Error.. Could not open the Graph Index File: %s\n", e->graph_path.c_str()); bang_set_error("cannot open %s", e->graph_path.c_str()); return BANG_ERR_IO; }
  struct stat st;
  if (fstat(fd, &st) != 0 || (size_t)st.st_size < (size_t)e->N * e->entry_len) { close(fd); bang_set_error("graph file too small"); return BANG_ERR_IO; }
  const size_t gsize = (size_t)st.st_size;
  const bool want_map = env_long("BANG_GRAPH_MMAP", 1) != 0;
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
                 const float* centroid, const uint32_t* chunk_off, uint32_t desc_code_stride) {
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
  // PQ layout first: it decides how the code rows are laid out in HBM
  uint32_t psz = 0, mp = m;
  BANG_TRY(bang_pq_layout(chunk_off, D, m, &psz, &mp));
  if (e->pq_mode == 1) { psz = 0; mp = m; }
  e->psz = psz;
  e->mp = mp;
  // PQ codes (+256 B slack: the distance kernel over-reads up to 19 B past a row).
  // Row stride.  <p>_pq_compressed.bin packs the rows m bytes apart, so a 70-byte row straddles two or three 64-byte lines.  The
  // search kernel fetches a row with one request per LINE it touches (CoopFetch, bang_device.h): rows that never leave their
  // 128-byte line are fetched at 47 G rows/s against 30 G rows/s packed (tools/dev/row_fetch_bench.hip).  "code_stride" = -1 (auto)
  // pads the rows to the next power of two (64 < m <= 128 -> 128, 32 < m <= 64 -> 64) when HBM holds the padded table next to what
  // the placement needs; 0 keeps them packed; a caller's own device table (desc->d_codes) is used as it is laid out.
  const size_t hbm_reserve = (size_t)16 << 30;
  uint32_t stride = m;
  if (d_codes_ext) {
    stride = desc_code_stride ? desc_code_stride : m;
    if (stride < m) { bang_set_error("code_stride %u is smaller than a row (m = %u)", stride, m); return BANG_ERR_ARG; }
  } else if (e->code_stride_opt > 0) {
    stride = (uint32_t)e->code_stride_opt;
    if (stride < m) { bang_set_error("option code_stride = %u is smaller than a row (m = %u)", stride, m); return BANG_ERR_ARG; }
  } else if (e->code_stride_opt < 0 && psz != 0 && m > 32 && m <= 128) {
    uint32_t pad = 64;
    while (pad < m) pad <<= 1;
    if (pad != m) {
      size_t free_b = 0, total_b = 0;
      (void)hipMemGetInfo(&free_b, &total_b);
      const size_t N_ = e->N, gbytes = N_ * e->entry_len + 256, vbytes = N_ * (size_t)D * e->tsize;
      const bool dev_packed = N_ * m + gbytes + hbm_reserve <= free_b;
      bool ok;
      if (e->graph_mode == BANG_GRAPH_DEVICE || (e->graph_mode == BANG_GRAPH_AUTO && dev_packed)) ok = N_ * pad + gbytes + hbm_reserve <= free_b;   // (never flips auto to host)
      else ok = N_ * pad + (e->vectors_opt != 0 ? vbytes : 0) + hbm_reserve + ((size_t)8 << 30) <= free_b;
      if (ok) stride = pad;
    }
  }
  e->code_stride = stride;
  const size_t code_bytes = (size_t)e->N * stride;
  if (d_codes_ext) {
    e->d_codes = (uint8_t*)d_codes_ext;
    e->codes_owned = false;
  } else {
    HIP_TRY(hipMalloc((void**)&e->d_codes, code_bytes + 256));
    e->codes_owned = true;
    if (stride == m) {
      HIP_TRY(hipMemset(e->d_codes + code_bytes, 0, 256));
      const size_t step = (size_t)1 << 30;
      for (size_t off = 0; off < code_bytes; off += step)
        HIP_TRY(hipMemcpy(e->d_codes + off, h_codes + off, std::min(step, code_bytes - off), hipMemcpyHostToDevice));
    } else {
      // packed chunks go up as they are (one contiguous copy each) and are spread to the row stride ON the device (a pitched
      // host-to-device copy of 70-byte rows moves a row at a time)
      HIP_TRY(hipMemset(e->d_codes, 0, code_bytes + 256));
      const size_t rows_per = std::max<size_t>(1, ((size_t)256 << 20) / m);
      uint8_t* d_tmp = nullptr;
      HIP_TRY(hipMalloc((void**)&d_tmp, rows_per * m));
      for (size_t r0 = 0; r0 < e->N; r0 += rows_per) {
        const size_t nr = std::min(rows_per, (size_t)e->N - r0);
        if (hipMemcpy(d_tmp, h_codes + r0 * m, nr * m, hipMemcpyHostToDevice) != hipSuccess ||
            hipMemcpy2D(e->d_codes + r0 * stride, stride, d_tmp, m, m, nr, hipMemcpyDeviceToDevice) != hipSuccess) {
          (void)hipFree(d_tmp);
          bang_set_error("PQ code upload failed: %s", hipGetErrorString(hipGetLastError()));
          return BANG_ERR_HIP;
        }
      }
      (void)hipFree(d_tmp);
    }
  }
  // Placement.  HBM left after the PQ codes decides: the whole graph (adjacency + vectors) if it fits with 16 GB to spare for
  // the per-batch state -> no host in the loop at all; else the graph stays in host RAM (C++ walker) and, if THEY fit, a packed
  // copy of the full-precision vectors goes to HBM for the re-rank (128 GB for 1e9 x 128 uint8 next to 70 GB of codes).
  if (e->graph_mode == BANG_GRAPH_AUTO) {
    size_t free_b = 0, total_b = 0;
    (void)hipMemGetInfo(&free_b, &total_b);
    const size_t gbytes = (size_t)e->N * e->entry_len + 256;
    e->graph_mode = (gbytes + hbm_reserve <= free_b) ? BANG_GRAPH_DEVICE : BANG_GRAPH_HOST;
    if (env_flag("BANG_DEBUG")) fprintf(stderr, "[bang] graph=auto -> %s (graph %.1f GB, free HBM %.1f GB)\n",
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
  if (!e->graph && e->entry_fn == nullptr && e->graph_path.empty() && e->ext_vecs_ready) {
    BANG_TRY(load_shared(e, e->rows_hash));              // (rows_hash: what the caller said the loading rank reported)
    e->loaded = true;
    return BANG_OK;
  }
  if (!e->graph) {
    // no resident graph: a streamed load (the caller's entry source), or a graph FILE that has not been touched yet.  If the pull
    // mode applies the entries only pass through (vectors -> HBM, adjacency -> pull rows); else a file is mapped as before.
    std::string why;
    const bool feasible = e->graph_mode != BANG_GRAPH_DEVICE && stream_feasible(e, hbm_reserve, &why);
    const bool from_file = (e->entry_fn == nullptr);
    const bool want_stream = env_long("BANG_STREAM_LOAD", 1) != 0;
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
      if (e->vecs_owned) dfree(e->d_vecs);           // the rows do not fit this host: keep the graph resident, the walker serves it
      e->d_vecs = nullptr; e->vecs_owned = true;
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
  if (e->pull) BANG_TRY(cache_rows_in_hbm(e));
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
  if (e->vecs_owned) dfree(e->d_vecs);
  e->d_vecs = nullptr; e->vecs_owned = true; e->ext_vecs = nullptr; e->ext_vecs_ready = false; e->rows_hash = 0;
  e->vec_on_device = false;
  for (uint32_t s = 0; s < BANG_MAX_ROW_SLICES; ++s) {
    if (e->peer_ptr[s]) (void)hipIpcCloseMemHandle(e->peer_ptr[s]);
    e->peer_ptr[s] = nullptr; e->slice_base[s] = 0;
  }
  dfree(e->d_slice_tab);
  e->n_slices = 0; e->slice_rows = 0; e->own_slot = 0xFFFFFFFFu;
  dfree(e->d_rows_hbm);
  e->n_rows_hbm = 0; e->rows_first = 0; e->rows_exported = false;
  if (e->h_adj) { (void)hipHostUnregister(e->h_adj); (void)munmap(e->h_adj, e->adj_bytes); }
  e->h_adj = nullptr; e->d_adj = nullptr; e->adj_bytes = 0; e->pull = false;
  e->rows_path.clear();
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
  rc = upload_index(e, codes.data(), nullptr, pivots.data(), centroid.data(), chunk_off.data(), 0);
  if (dsrc.fd >= 0) close(dsrc.fd);
  e->entry_fn = nullptr; e->entry_ctx = nullptr; e->entry_src_rereadable = false;
  if (rc != BANG_OK) unload_index(e);
  return rc;
}


}  // namespace bang
