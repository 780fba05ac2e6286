// bang_options.cpp -- EVERY option and environment switch of the host engine, in one place.
//
//  * kOptions: what bang_set_option(key, value) accepts.  Each row names the engine field, the legal range, WHEN the option may still
//    change (placement / layout options are consumed by bang_load, loop-shape options by bang_alloc) and the environment variable
//    that presets it at bang_create -- callers of the bang.h class API (no option methods, e.g. the bang_search CLI) choose that way.
//    An environment value outside the range is clamped; a bang_set_option value outside it is BANG_ERR_ARG.
//  * kSwitches: diagnostic / experiment switches that exist only as environment variables.  They are read when they are USED
//    (env_long / env_flag / env_str below), not cached at start-up: a test or a benchmark may change them between two engines.
// bang_describe_options() prints both tables (INTEGRATION.md section 4 is that text).
#include "bang_engine.h"

namespace bang {

enum Phase { ANY = 0, BEFORE_LOAD = 1, BEFORE_ALLOC = 2 };
enum Kind { INT = 0, FLAG = 1 };     // FLAG: any non-zero value means 1

struct OptionDef {
  const char* key;
  const char* env;
  int bang_engine::*field;
  long lo, hi;
  Kind kind;
  Phase phase;
  const char* help;
};

static const OptionDef kOptions[] = {
    // ---- placement and layout: consumed by bang_load
    {"graph", "BANG_GRAPH", &bang_engine::graph_mode, BANG_GRAPH_HOST, BANG_GRAPH_AUTO, INT, BEFORE_LOAD,
     "0 host | 1 device | 2 auto (default): where graph + vectors live; auto = HBM when they fit next to the PQ codes with 16 GB to spare "
     "(environment: host | device | auto)"},
    {"device", "BANG_DEVICE", &bang_engine::device, 0, 1 << 20, INT, BEFORE_LOAD, "HIP device ordinal"},
    {"pq", "BANG_PQ", &bang_engine::pq_mode, 0, 1, INT, BEFORE_LOAD, "0 = pivot table resident in LDS (default when it fits), 1 = LUT path (K1 + K2)"},
    {"pq_ragged", "BANG_PQ_RAGGED", &bang_engine::pq_ragged, 0, 1, FLAG, BEFORE_LOAD, "2-dim/1-dim PQ layouts: exact-size pivot table where a kernel instance exists (default 1)"},
    {"code_stride", "BANG_CODE_STRIDE", &bang_engine::code_stride_opt, -1, 4096, INT, BEFORE_LOAD,
     "bytes between PQ code rows in HBM: 0 = packed as in <p>_pq_compressed.bin (m), -1 = auto: padded to the next power of two (m = 70 -> 128) "
     "when HBM holds the padded table, so that a row never leaves its 128-byte line (one request per row instead of two or three)"},
    {"vectors", "BANG_VECTORS", &bang_engine::vectors_opt, -1, 1, INT, BEFORE_LOAD,
     "host graph: 1 = packed copy of the full-precision vectors in HBM for the re-rank, 0 = the walker ships every expanded node's vector "
     "(the reference's data flow), -1 = auto (1 when the copy fits with 16 GB to spare)"},
    {"pull", "BANG_PULL", &bang_engine::pull_opt, -1, 1, INT, BEFORE_LOAD,
     "host graph: 1 = the search kernel pulls 256-byte adjacency rows from pinned host memory over PCIe, 0 = C++ walker threads serve them, -1 = auto"},
    {"rows_hbm", "BANG_ROWS_HBM", &bang_engine::rows_hbm_opt, -1, 1 << 22, INT, BEFORE_LOAD,
     "pull mode: MB of HBM for a copy of the first adjacency rows, read from there instead of over PCIe (-1 = auto: what the index leaves beyond 6 GB, "
     "where the rows do not all fit; 0 = none)"},
    // ---- loop shape: consumed by bang_alloc
    {"lanes", "BANG_LANES", &bang_engine::lanes_opt, 0, BANG_MAX_LANES, INT, BEFORE_ALLOC, "launch-per-iteration loop: independent query groups pipelined against each other (0 = auto)"},
    {"threads", "BANG_THREADS", &bang_engine::threads_opt, 0, 4096, INT, BEFORE_ALLOC, "host walker threads per lane (0 = auto from the CPU quota)"},
    {"persistent", "BANG_PERSISTENT", &bang_engine::persistent, -1, 1, INT, BEFORE_ALLOC, "1 = ONE search-kernel launch per batch, 0 = a front + back launch per iteration, -1 = auto"},
    {"walker", "BANG_WALKER", &bang_engine::walker_opt, 0, 1, INT, BEFORE_ALLOC,
     "host graph loaded in pull mode: 1 = the C++ walker threads serve the adjacency rows to the host-paced search kernel (the reference's data flow, "
     "bang_search.cu:771-813) -- reading the same 256-byte pull rows the kernel would pull itself, so no resident graph image is needed; 0 = the kernel pulls (default)"},
    {"search", "BANG_SEARCH", &bang_engine::search_opt, -1, 1, INT, BEFORE_ALLOC, "1 = the query-resident search kernel (bang_k_search), 0 = the per-iteration kernels, -1 = auto"},
    {"stage_zero_copy", "BANG_STAGE_ZC", &bang_engine::stage_zero_copy, -1, 2, INT, BEFORE_ALLOC,
     "walker forms, where staged adjacency rows travel: 0 = H2D copy per iteration, 1 = kernels read mapped pinned memory, 2 = CPU stores through the PCIe BAR, -1 = auto"},
    {"numa", "BANG_NUMA", &bang_engine::numa_opt, -1, 1, INT, BEFORE_ALLOC, "1 = pin walker threads to the GPU's NUMA node, one physical core each; 0 / -1 = leave them to the scheduler"},
    {"timing", "BANG_TIMING", &bang_engine::timing, 0, 1, INT, BEFORE_ALLOC, "1 = stamp every search / front launch in-kernel (s_memrealtime) for bang_get_stats"},
    {"front_wgs", "BANG_FRONT_WGS", &bang_engine::front_wgs_opt, -1, 1 << 20, INT, BEFORE_ALLOC, "launch-per-iteration loop: workgroups per front launch (-1 = auto, 0 = all CUs)"},
    {"fuse_rerank", "BANG_FUSE_RERANK", &bang_engine::fuse_rerank, -1, 1, INT, ANY,
     "self-paced search kernel, vectors resident in HBM (8-bit: D % 16 == 0; float: D % 4 == 0; D <= 256, no MIPS): 1 / -1 (auto) = the wave that finishes a query re-ranks it on the spot (K6 + K7 inside the search launch), "
     "0 = a re-rank launch behind the search.  Same results"},
    // ---- may change between queries
    {"use_flag", "BANG_USE_FLAG", &bang_engine::use_flag, 0, 1, FLAG, BEFORE_ALLOC, "0 = wait for the front kernel with runtime calls instead of its in-kernel completion flag (ablation)"},
    {"compact", "BANG_COMPACT", &bang_engine::compact, 0, 1, FLAG, ANY, "launch-per-iteration loop: straggler compaction on / off"},
    {"stagger_us", "BANG_STAGGER_US", &bang_engine::stagger_us, 0, 1 << 30, INT, ANY, "lane i starts i x stagger_us later"},
    {"fp_batch", "BANG_FP_BATCH", &bang_engine::fp_batch, 1, 1 << 20, INT, ANY, "vector-log rows are copied to the device every fp_batch iterations"},
    {"check_every", "BANG_CHECK_EVERY", &bang_engine::check_every, 1, 1 << 20, INT, ANY, "graph in HBM, launch-per-iteration loop: poll the active counter every N iterations"},
    {"host_walk_timeout_ms", "BANG_HOST_WALK_TIMEOUT_MS", &bang_engine::host_walk_timeout_ms, 10, 3600000, INT, ANY,
     "host-paced search kernel: a walker thread that sees no progress for this long stops the kernel through its control lines (default 20000)"},
    {"kernel_go_timeout_ms", "BANG_KERNEL_GO_TIMEOUT_MS", &bang_engine::kernel_go_timeout_ms, 10, 3600000, INT, ANY,
     "host-paced search kernel: a pacing group that waits this long for its rows gives up on its own -- the host is gone (default 30000; the host gives up first)"},
    {"walker_stall_ms", nullptr, &bang_engine::walker_stall_ms, 0, 600000, INT, ANY,
     "TEST HOOK: the walker team sleeps this long at the start of the next host-paced query (one shot) -- provokes the kernel's give-up path"},
};

struct SwitchDef { const char* env; const char* help; };
static const SwitchDef kSwitches[] = {
    {"BANG_PULL_ROWS_DIR", "directory (tmpfs) every rank of a node can see: ONE pull-rows file per index there, built by the first rank to load, mapped by the others"},
    {"BANG_PULL_ROWS_INTERLEAVE", "0 = leave the pages of a shared pull-rows file where first touch puts them (default 1: interleaved over the NUMA nodes)"},
    {"BANG_ROWS_HBM_MAX_ROWS", "test hook: cap on the adjacency rows copied to HBM (a partial copy of a small index)"},
    {"BANG_STREAM_LOAD", "0 = bang_load maps <prefix>_disk.bin up front instead of streaming it through (pull mode)"},
    {"BANG_GRAPH_MMAP", "0 = private copy of the graph file (transparent huge pages) instead of a shared read-only mapping"},
    {"BANG_SEARCH_MAX_WGS", "search kernel: cap on workgroups (experiments / tests)"},
    {"BANG_SEARCH_MAX_WAVES", "search kernel: cap on waves per workgroup"},
    {"BANG_FILTER_MEM", "visited filters in 1 = uncached / 2 = fine-grained device memory instead of ordinary device memory (experiment, read at bang_alloc)"},
    {"BANG_SUMM_ITERS", "search kernel, self-paced form: the filter summary serves a query's first N iterations only (0 = auto: all, off for launches of <= 5 queries per CU; -1 = all)"},
    {"BANG_SEARCH_PRIO", "search kernel, self-paced form: raised wave priority (s_setprio) for the stretches of an iteration that end in memory requests -- 1 = both (row arrival -> probes + code-row requests; parent selection -> row request), 2 = the second only, 3 = the first only, 0 = off; unset = auto: both for launches of <= 10 queries per CU"},
    {"BANG_SPEC_ROWS", "search kernel, self-paced form, 70 / 74-chunk layouts: 1 = code rows of all ids of an adjacency row requested with their filter probes, 2 = behind the filter (survivors only), 0 = auto (1 where the rows are pulled, and for launches of <= 8 queries per CU where the graph is in HBM)"},
    {"BANG_WALKER_SELF_ROWS", "host-paced search kernel, walker-from-rows form: 0 = every row comes from the walker threads, also those this GPU holds in its HBM copy (A/B)"},
    {"BANG_SEARCH_GS", "host-paced search kernel: waves per pacing group (default 8)"},
    {"BANG_SEARCH_CTX", "host-paced search kernel: query contexts per wave (default 1; 2 measured slower)"},
    {"BANG_RESULTS_DIRECT", "0 = the fused re-rank writes a small batch's results to device memory and they are copied back behind the launch (A/B; default: straight into the pinned mirror)"},
    {"BANG_MAILBOX_BYTES", "results up to this size return through the pinned mirror in one copy (default 8 MB)"},
    {"BANG_HELPER_GRACE_US", "walker helpers spin this long for the next batch before parking (default 4000)"},
    {"BANG_WALK_NT", "0 = walker threads copy staged rows with memcpy instead of 512-bit non-temporal stores"},
    {"BANG_DEBUG", "placement / allocation decisions and lane progress on stderr"},
    {"BANG_TIMELINE", "host time of every stage of a bang_query (a stream sync behind each) on stderr"},
    {"BANG_SEARCH_PROF", "search kernel: phase times on stderr (half-rounds of the host-paced form; per-iteration phases in a -DBANG_SEARCH_PHASE_PROF build)"},
    {"BANG_WALK_PROF", "walker threads: time serving vs polling, per thread, on stderr"},
    {"BANG_WATCHDOG", "file to which a watchdog thread appends the lanes' phases every 5 s while a query runs"},
    {"BANG_KT_TRACE", "file receiving the raw in-kernel launch stamps of lane 0 (timing = 1)"},
    {"BANG_AMD_LIB", "(Python binding only) load another build of libbang.so: A/B runs"},
};

static const OptionDef* find_option(const char* key) {
  for (const OptionDef& o : kOptions)
    if (strcmp(o.key, key) == 0) return &o;
  return nullptr;
}

static void store(bang_engine* e, const OptionDef& o, long v) {
  e->*(o.field) = (int)(o.kind == FLAG ? (v ? 1 : 0) : v);
  if (o.field == &bang_engine::graph_mode) e->graph_opt = e->graph_mode;       // what the caller asked for (restored by bang_unload)
}

int set_option(bang_engine* e, const char* key, long value) {
  const OptionDef* o = find_option(key);
  if (!o) { bang_set_error("unknown option %s", key); return BANG_ERR_ARG; }
  // placement and layout options are consumed by bang_load, loop-shape options by bang_alloc: changing them afterwards would
  // leave buffers that do not match the option
  if (o->phase == BEFORE_LOAD && e->loaded) { bang_set_error("option %s must be set before bang_load", key); return BANG_ERR_ARG; }
  if (o->phase == BEFORE_ALLOC && e->allocated) { bang_set_error("option %s must be set before bang_alloc", key); return BANG_ERR_ARG; }
  if (o->kind != FLAG && (value < o->lo || value > o->hi)) { bang_set_error("option %s: %ld is outside [%ld, %ld]", key, value, o->lo, o->hi); return BANG_ERR_ARG; }
  store(e, *o, value);
  return BANG_OK;
}

void apply_env_defaults(bang_engine* e) {
  for (const OptionDef& o : kOptions) {
    const char* v = o.env ? getenv(o.env) : nullptr;
    if (!v || !*v) continue;
    long x;
    if (o.field == &bang_engine::graph_mode && !isdigit((unsigned char)*v) && *v != '-')
      x = strcmp(v, "device") == 0 ? BANG_GRAPH_DEVICE : strcmp(v, "auto") == 0 ? BANG_GRAPH_AUTO : BANG_GRAPH_HOST;
    else
      x = atol(v);
    store(e, o, std::min(o.hi, std::max(o.lo, x)));
  }
}

long env_long(const char* name, long dflt) {
  const char* v = getenv(name);
  return (v && *v) ? atol(v) : dflt;
}
bool env_flag(const char* name) { return getenv(name) != nullptr; }
const char* env_str(const char* name) {
  const char* v = getenv(name);
  return (v && *v) ? v : nullptr;
}

}  // namespace bang

extern "C" int bang_set_option(bang_engine_t* e, const char* key, long value) {
  if (!e || !key) return BANG_ERR_ARG;
  return bang::set_option(e, key, value);
}

extern "C" int bang_describe_options(char* buf, size_t cap) {
  std::string s = "options (bang_set_option key | environment preset | range | settable until):\n";
  char line[1024];
  static const char* const until[] = {"any time", "bang_load", "bang_alloc"};
  for (const bang::OptionDef& o : bang::kOptions) {
    snprintf(line, sizeof(line), "  %-22s %-26s [%ld, %ld]  %-10s  %s\n", o.key, o.env ? o.env : "-", o.lo, o.hi, until[o.phase], o.help);
    s += line;
  }
  s += "environment switches (read when used):\n";
  for (const bang::SwitchDef& w : bang::kSwitches) {
    snprintf(line, sizeof(line), "  %-26s %s\n", w.env, w.help);
    s += line;
  }
  if (buf && cap) {
    const size_t n = std::min(cap - 1, s.size());
    memcpy(buf, s.data(), n);
    buf[n] = 0;
  }
  return (int)s.size() + 1;          // bytes needed, terminator included
}
