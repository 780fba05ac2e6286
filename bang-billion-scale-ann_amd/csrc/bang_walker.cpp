// bang_walker.cpp -- the CPU graph walker (bang_search.cu:771-813) in its three forms (slice walk of the launch-per-iteration loop, the walker
// team of the host-paced search kernel), NUMA pinning and the persistent lane / helper threads.
// Reference line numbers: /root/reference/BANG_Base/bang_search.cu.
#include "bang_engine.h"

namespace bang {

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
// 64-byte lines of a staged row written with ONE 512-bit non-temporal store each (BANG_WALK_NT=0: memcpy).  The destination is the
// GPU's BAR, mapped write-combining: ordinary stores already gather in the core's WC buffers and leave as whole 64-byte writes, so
// the gain is small -- 15.5 / 13.8 ms against 15.7 / 15.0 ms per SIFT1B-shape batch in two A/B pairs.
__attribute__((target("avx512f"))) static void wc_copy_nt(void* dst, const void* src, size_t n) {
  size_t o = 0;
  for (; o + 64 <= n; o += 64) _mm512_stream_si512((__m512i*)((uint8_t*)dst + o), _mm512_loadu_si512((const uint8_t*)src + o));
  if (o < n) memcpy((uint8_t*)dst + o, (const uint8_t*)src + o, n - o);      // (degree bounds below 16 ids: a row shorter than one line)
}

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
  // vectors resident and pull rows built: the rows ARE the adjacency lists, 256-byte aligned, 64 ids, pad behind the ids -- nothing else of
  // a graph entry is needed (option walker = 1; also what lets this form run on a streamed load, which keeps no graph image)
  const uint32_t* adj_rows = e->walker_rows ? e->h_adj : nullptr;
  uint8_t* fp = ship_vec ? (e->fp_direct ? e->d_fp : e->h_fp) : nullptr;
  const uint32_t R = e->R, cstride = e->cand_stride;
  const uint64_t pf0 = ship_vec ? 0 : (vb & ~(uint64_t)63);  // resident vectors: only the adjacency part of an entry is touched
  uint64_t bytes = 0;
  auto t_last = Clock::now();
  uint32_t idle = 0;
  bool served_unfenced = false;
  const bool prof = env_flag("BANG_WALK_PROF");
  const bool nt_rows = env_long("BANG_WALK_NT", 1) != 0 && __builtin_cpu_supports("avx512f");      // diagnostic: time spent serving vs polling, per thread
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
        if (adj_rows) {
          const uint8_t* row = (const uint8_t*)(adj_rows + (uint64_t)p_ * 64u);
          for (uint64_t o = 0; o < 256; o += 64) __builtin_prefetch(row + o, 0, 0);
        } else {
          const uint8_t* ent = graph + (uint64_t)p_ * elen;
          for (uint64_t o = pf0; o < elen; o += 64) __builtin_prefetch(ent + o, 0, 0);
        }
      }
    }
    uint32_t counts[4] = {0, 0, 0, 0};
    for (uint32_t i = 0; i < W; ++i) {
      const uint32_t p_ = par[i];
      if (p_ >= BANG_IDLE_PARENT) continue;
      if (adj_rows) {                                                   // (vectors resident: every published parent wants its row)
        const uint32_t* row = adj_rows + (uint64_t)p_ * 64u;
        uint32_t deg = R < 64u ? R : 64u;
        while (deg > 0 && row[deg - 1] == BANG_ADJ_PAD) --deg;          // ids first, ascending, the pad value behind them
        uint32_t* srow = e->d_srows + ((size_t)w * 16 + i) * 64;
        const size_t nbytes = std::min<size_t>(((size_t)deg * 4 + 63) & ~(size_t)63, (size_t)R * 4);
        if (nt_rows) wc_copy_nt(srow, row, nbytes);
        else memcpy(srow, row, nbytes);
        counts[i >> 2] |= deg << (8 * (i & 3u));
        bytes += nbytes;
        continue;
      }
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
        if (nt_rows) wc_copy_nt(srow, ent + vb + 4, nbytes);            // one 512-bit non-temporal store per 64-byte line
        else memcpy(srow, ent + vb + 4, nbytes);                        // :809-810
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
      else if (ms_since(t_last) > (double)e->host_walk_timeout_ms || ln.pw_error.load(std::memory_order_relaxed)) {
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
      const double grace_ms = (double)env_long("BANG_HELPER_GRACE_US", 4000) * 1e-3;
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

}  // namespace bang
