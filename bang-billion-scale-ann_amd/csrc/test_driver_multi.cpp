// test_driver_multi.cpp -- `bang_search_multi`: the `bang_search` harness over SEVERAL GPUs of one node, on the C-ABI alone (no Python, no RCCL): what a
// C / C++ / cgo maintainer would write to shard a query batch the way bench.py --gpus N does (the single-GPU reference
// has no counterpart: its timed region is BANG_Base/test_driver.cpp:433-439, its table :402-411/:526, its recall :43-93).
//
//   bang_search_multi <index prefix> <query.bin> <groundtruth.bin> <numQueries> <k> <uint8|int8|float> <L> <nGPUs> [share] [slice=<rows>]
//
// One PROCESS per GPU (forked before anything touches HIP).  Rank r searches the contiguous shard [r Q / W, (r + 1) Q / W) on its own
// engine; the ranks share
//   * ONE copy of the adjacency rows in host memory (BANG_PULL_ROWS_DIR: rank 0 loads first and builds the rows file, the others map it),
//   * and -- peer rows -- the node's spare HBM: every rank keeps one slice of the rows (bang_rows_slice_e), exports it as a 64-byte
//     hipIpcMemHandle (bang_rows_export_e) through a shared-memory mailbox, and maps the others' (bang_rows_import_e),
// and write their ids into one shared [Q][k] block (host memory stands in for the RCCL gather of the Python path).  Five timed runs per L
// (bang_init outside, a barrier on both sides, the slowest rank counts), then the reference's table line: L, ms, QPS, k-recall@k.
// `share`: every rank on device 0 -- a functional run of the whole exchange on a 1-GPU box (tests/test_gpu_fileflow.py); `slice=<rows>` caps
// the HBM slice per rank (a part of the graph then stays on the host: all three row sources in one run).
#include <sys/mman.h>
#include <sys/stat.h>
#include <sys/wait.h>
#include <unistd.h>

#include <algorithm>
#include <atomic>
#include <chrono>
#include <cstdint>
#include <cstdio>
#include <cstdlib>
#include <cstring>
#include <fstream>
#include <string>
#include <vector>
#include <csignal>

#include "bang_c.h"

namespace {

struct Shared {                       // one anonymous MAP_SHARED page set, created before the fork
  std::atomic<uint32_t> arrived, generation, failed, loaded;
  uint64_t capacity[BANG_MAX_ROW_SLICES];
  uint64_t slice_rows;
  uint64_t rows[BANG_MAX_ROW_SLICES];
  unsigned char handle[BANG_MAX_ROW_SLICES][64];
  double ms[BANG_MAX_ROW_SLICES];
  uint64_t from_peer[BANG_MAX_ROW_SLICES], from_own[BANG_MAX_ROW_SLICES], pulled[BANG_MAX_ROW_SLICES];
};

// sense-reversing barrier among W processes; a rank that failed releases the others with an error
bool barrier(Shared* s, uint32_t W) {
  const uint32_t gen = s->generation.load(std::memory_order_acquire);
  if (s->arrived.fetch_add(1, std::memory_order_acq_rel) + 1 == W) {
    s->arrived.store(0, std::memory_order_relaxed);
    s->generation.store(gen + 1, std::memory_order_release);
  } else {
    while (s->generation.load(std::memory_order_acquire) == gen) {
      if (s->failed.load(std::memory_order_relaxed)) return false;
      usleep(50);
    }
  }
  return s->failed.load(std::memory_order_relaxed) == 0;
}

#define CHECK(call)                                                                                   \
  do {                                                                                                \
    const int rc_ = (call);                                                                           \
    if (rc_ != BANG_OK) {                                                                             \
      fprintf(stderr, "[rank %u] %s failed (%d): %s\n", rank, #call, rc_, bang_last_error());          \
      sh->failed.store(1);                                                                            \
      return 1;                                                                                       \
    }                                                                                                 \
  } while (0)

double recall_of(const std::vector<uint32_t>& gt_ids, const std::vector<float>& gt_d, size_t gt_w, const uint64_t* found, uint32_t Q, uint32_t k) {
  // tie-aware k-recall@k (BANG_Base/test_driver.cpp:43-93): the truth set of a query = its first k entries extended over the ties of the k-th distance
  double hits = 0;
  std::vector<uint32_t> want, got;
  for (uint32_t q = 0; q < Q; ++q) {
    size_t n = k - 1;
    while (n < gt_w && gt_d[q * gt_w + n] == gt_d[q * gt_w + k - 1]) ++n;
    want.assign(gt_ids.begin() + q * gt_w, gt_ids.begin() + q * gt_w + n);
    std::sort(want.begin(), want.end());
    want.erase(std::unique(want.begin(), want.end()), want.end());
    got.resize(k);
    for (uint32_t j = 0; j < k; ++j) got[j] = (uint32_t)found[(size_t)q * k + j];
    std::sort(got.begin(), got.end());
    got.erase(std::unique(got.begin(), got.end()), got.end());
    size_t a = 0, b = 0;
    while (a < want.size() && b < got.size()) {
      if (want[a] < got[b]) ++a;
      else if (got[b] < want[a]) ++b;
      else { hits += 1; ++a; ++b; }
    }
  }
  return hits / Q * (100.0 / k);
}

int run_rank(uint32_t rank, uint32_t W, bool share, uint64_t slice_cap, Shared* sh, uint64_t* all_ids, const char* prefix, int dtype, const std::vector<uint8_t>& queries,
             size_t qbytes, uint32_t Q, uint32_t k, uint32_t L) {
  const uint32_t q0 = (uint32_t)((uint64_t)Q * rank / W), q1 = (uint32_t)((uint64_t)Q * (rank + 1) / W), nq = q1 - q0;
  bang_engine_t* e = nullptr;
  CHECK(bang_create(dtype, &e));
  CHECK(bang_set_option(e, "device", share ? 0 : (int)rank));
  CHECK(bang_set_option(e, "graph", BANG_GRAPH_HOST));           // the SIFT1B placement: rows in host memory, pulled by the kernel
  // rank 0 loads first: it builds the node's ONE rows file, the others map it
  if (rank != 0) while (sh->loaded.load(std::memory_order_acquire) == 0) { if (sh->failed.load()) return 1; usleep(200); }   // (the parent sets `failed` when a rank dies)
  CHECK(bang_load_e(e, prefix));
  if (rank == 0) sh->loaded.store(1, std::memory_order_release);
  // ---- peer rows: capacities -> slice size -> slice + export -> import
  uint64_t cap = 0, n_nodes = 0;
  CHECK(bang_rows_capacity_e(e, &cap));
  sh->capacity[rank] = cap;
  if (const char* kr = getenv("BANG_MULTI_TEST_KILL_RANK")) {       // test hook: this rank dies HERE without a word (as a GPU fault or the OOM killer would end it)
    if ((uint32_t)atoi(kr) == rank) raise(SIGKILL);
  }
  if (!barrier(sh, W)) return 1;
  CHECK(bang_get_num_nodes(e, &n_nodes));
  uint64_t n = *std::min_element(sh->capacity, sh->capacity + W);
  n = std::min<uint64_t>(n, (n_nodes + W - 1) / W);
  if (slice_cap) n = std::min<uint64_t>(n, slice_cap);
  const uint64_t first = std::min<uint64_t>((uint64_t)rank * n, n_nodes);
  if (n > 0) {
    uint64_t f = 0, r = 0;
    CHECK(bang_rows_slice_e(e, first, std::min<uint64_t>(n, n_nodes - first)));
    CHECK(bang_rows_export_e(e, sh->handle[rank], &f, &r));
    sh->rows[rank] = r;
    if (!barrier(sh, W)) return 1;
    for (uint32_t r2 = 0; r2 < W; ++r2) CHECK(bang_rows_import_e(e, r2, W, n, sh->rows[r2], r2 == rank ? nullptr : sh->handle[r2]));
  }
  if (!barrier(sh, W)) return 1;
  // ---- search: five timed runs, bang_init outside the timed region (test_driver.cpp:424-439)
  CHECK(bang_set_searchparams_e(e, (int)k, (int)L, BANG_DIST_L2));
  CHECK(bang_alloc_e(e, (int)nq));
  std::vector<uint64_t> ids((size_t)nq * k);
  std::vector<float> dists((size_t)nq * k);
  for (int run = 0; run < 5; ++run) {
    CHECK(bang_init_e(e, (int)nq));
    if (!barrier(sh, W)) return 1;
    const auto t0 = std::chrono::steady_clock::now();
    CHECK(bang_query_e(e, queries.data() + (size_t)q0 * qbytes, (int)nq, ids.data(), dists.data()));
    memcpy(all_ids + (size_t)q0 * k, ids.data(), ids.size() * 8);                       // this rank's block of the [Q][k] answer
    sh->ms[rank] = std::chrono::duration<double, std::milli>(std::chrono::steady_clock::now() - t0).count();
    if (!barrier(sh, W)) return 1;
    if (rank == 0) {
      double ms = 0;
      for (uint32_t r2 = 0; r2 < W; ++r2) ms = std::max(ms, sh->ms[r2]);
      sh->ms[BANG_MAX_ROW_SLICES - 1] = ms;                                            // (the parent prints it with the recall)
    }
    if (!barrier(sh, W)) return 1;
  }
  bang_stats st;
  CHECK(bang_get_stats(e, &st));
  sh->from_peer[rank] = st.rows_from_peer; sh->from_own[rank] = st.rows_from_own_hbm; sh->pulled[rank] = st.pulled_bytes / 256;
  // ---- tear-down in two phases: nobody frees a slice the others still map
  CHECK(bang_free_e(e));
  CHECK(bang_rows_close_peers_e(e));
  if (!barrier(sh, W)) return 1;
  CHECK(bang_unload_e(e));
  bang_destroy(e);
  return 0;
}

}  // namespace

int main(int argc, char** argv) {
  if (argc < 9) {
    fprintf(stderr, "usage: %s <index prefix> <query.bin> <groundtruth.bin> <numQueries> <k> <uint8|int8|float> <L> <nGPUs> [share] [slice=<rows>]\n", argv[0]);
    return 2;
  }
  const char* prefix = argv[1];
  const uint32_t Q = (uint32_t)atoi(argv[4]), k = (uint32_t)atoi(argv[5]), L = (uint32_t)atoi(argv[7]), W = (uint32_t)atoi(argv[8]);
  const std::string dt(argv[6]);
  const int dtype = dt == "uint8" ? BANG_U8 : dt == "int8" ? BANG_I8 : dt == "float" ? BANG_F32 : -1;
  bool share = false;
  uint64_t slice_cap = 0;
  for (int i = 9; i < argc; ++i) {
    const std::string a(argv[i]);
    if (a == "share") share = true;
    else if (a.rfind("slice=", 0) == 0) slice_cap = strtoull(a.c_str() + 6, nullptr, 10);
  }
  if (dtype < 0 || W == 0 || W > BANG_MAX_ROW_SLICES - 1 || Q < W || k == 0) { fprintf(stderr, "bad arguments\n"); return 2; }
  std::ifstream qin(argv[2], std::ios::binary);
  if (!qin.is_open()) { printf("Error.. Could not open the Query File: %s\n", argv[2]); return 1; }
  int32_t nq_file = 0, dim = 0;
  qin.read((char*)&nq_file, 4); qin.read((char*)&dim, 4);
  const size_t qbytes = (size_t)dim * (dtype == BANG_F32 ? 4 : 1);
  std::vector<uint8_t> queries((size_t)Q * qbytes);
  qin.read((char*)queries.data(), (std::streamsize)queries.size());
  std::ifstream gin(argv[3], std::ios::binary);
  if (!gin.is_open()) { printf("Groundtruth file could not be loaded:%s\n", argv[3]); return 1; }
  int32_t gn = 0, gw = 0;
  gin.read((char*)&gn, 4); gin.read((char*)&gw, 4);
  std::vector<uint32_t> gt_ids((size_t)gn * gw);
  std::vector<float> gt_d((size_t)gn * gw);
  gin.read((char*)gt_ids.data(), (std::streamsize)(gt_ids.size() * 4));
  gin.read((char*)gt_d.data(), (std::streamsize)(gt_d.size() * 4));

  // ONE rows file for the node, unless the caller has chosen a directory already
  char dir[] = "/dev/shm/bang_multi_XXXXXX";
  bool own_dir = false;
  if (!getenv("BANG_PULL_ROWS_DIR")) {
    if (!mkdtemp(dir)) { perror("mkdtemp"); return 1; }
    setenv("BANG_PULL_ROWS_DIR", dir, 1);
    own_dir = true;
  }
  const size_t sh_bytes = sizeof(Shared) + (size_t)Q * k * 8;
  void* mem = mmap(nullptr, sh_bytes, PROT_READ | PROT_WRITE, MAP_SHARED | MAP_ANONYMOUS, -1, 0);
  if (mem == MAP_FAILED) { perror("mmap"); return 1; }
  Shared* sh = new (mem) Shared();
  uint64_t* all_ids = (uint64_t*)((char*)mem + sizeof(Shared));
  std::vector<pid_t> kids;
  for (uint32_t r = 0; r < W; ++r) {              // (forked before anything has touched HIP: every child initialises its own device)
    const pid_t pid = fork();
    if (pid == 0) _exit(run_rank(r, W, share, slice_cap, sh, all_ids, prefix, dtype, queries, qbytes, Q, k, L));
    kids.push_back(pid);
  }
  // Reap in the order the ranks END, not in rank order: a rank that dies without going through CHECK() (SIGSEGV, a GPU fault abort, the OOM
  // killer) never sets `failed` itself -- the parent does it here, which releases the siblings from barrier() and the wait-for-loaded loop
  // (they used to spin forever behind a dead rank, and the parent behind them).
  int bad = 0;
  for (size_t left = kids.size(); left > 0; --left) {
    int st = 0;
    const pid_t pid = waitpid(-1, &st, 0);
    if (pid < 0) { perror("waitpid"); sh->failed.store(1); bad = 1; break; }
    if (!(WIFEXITED(st) && WEXITSTATUS(st) == 0)) {
      const size_t r = (size_t)(std::find(kids.begin(), kids.end(), pid) - kids.begin());
      if (WIFSIGNALED(st)) fprintf(stderr, "bang_search_multi: rank %zu was killed by signal %d\n", r, WTERMSIG(st));
      else fprintf(stderr, "bang_search_multi: rank %zu exited with status %d\n", r, WIFEXITED(st) ? WEXITSTATUS(st) : -1);
      sh->failed.store(1);
      bad = 1;
    }
  }
  if (own_dir) { std::string cmd = std::string("rm -rf ") + dir; (void)!system(cmd.c_str()); }
  if (bad || sh->failed.load()) { fprintf(stderr, "bang_search_multi: a rank failed\n"); return 1; }
  const double ms = sh->ms[BANG_MAX_ROW_SLICES - 1];
  const float recall = (float)recall_of(gt_ids, gt_d, (size_t)gw, all_ids, Q, k);      // (a float, as the reference harness prints it: test_driver.cpp:506)
  uint64_t peer = 0, own = 0, pulled = 0;
  for (uint32_t r = 0; r < W; ++r) { peer += sh->from_peer[r]; own += sh->from_own[r]; pulled += sh->pulled[r]; }
  printf("GPUs\tL\tTime \tQPS\t\t%u-r@%u\trows: own HBM / peer HBM / host\n", k, k);
  printf("%u\t%u\t%.2f\t%.2f\t%.2f\t%llu / %llu / %llu\n", W, L, ms, Q * 1000.0 / ms, recall, (unsigned long long)own, (unsigned long long)peer,
         (unsigned long long)pulled);
  return 0;
}
