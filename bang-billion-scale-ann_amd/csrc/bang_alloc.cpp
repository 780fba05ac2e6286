// bang_alloc.cpp -- bang_alloc / bang_free (bang_search.cu:370-425): per-batch buffers, lanes and the choice of the loop form.
// Reference line numbers: /root/reference/BANG_Base/bang_search.cu.
#include "bang_engine.h"

namespace bang {

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
  if (e->h_results) { (void)hipHostFree(e->h_results); e->h_results = nullptr; e->h_results_dev = nullptr; }
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

// an allocation that fails for want of HBM is BANG_ERR_NOMEM (bang_alloc_e then retries without the HBM row cache); anything else BANG_ERR_HIP
#define HIP_TRY_ALLOC(x)                                                  \
  do {                                                                     \
    const hipError_t _a = (x);                                             \
    if (_a == hipErrorOutOfMemory) { (void)hipGetLastError(); bang_set_error("out of device memory: %s", #x); return BANG_ERR_NOMEM; } \
    HIP_TRY(_a);                                                           \
  } while (0)

int alloc_buffers(bang_engine* e, int Q) {         // (bang_alloc_e has validated the pull rows)
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
  const bool walk_rows_want = !dev_graph && e->pull && e->walker_opt == 1;          // rows exist, but the walker team is to serve them
  if ((dev_graph || (e->pull && !walk_rows_want)) && e->persistent != 0 && e->search_opt != 0 && e->psz != 0) {
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
  // the walker team of the host-paced kernel reads the pull rows where they exist (vectors are resident then: nothing else of a graph entry
  // is needed) -- the north-star data flow without a resident graph image
  e->walker_rows = e->search_host && e->pull && e->h_adj && e->vec_on_device;
  if (walk_rows_want && !e->walker_rows) {
    bang_set_error("option walker = 1: the host-paced search kernel is not available here (needs CPU-writable device memory, persistent != 0, search != 0)");
    return BANG_ERR_UNSUPPORTED;
  }
  // a walker form on an index whose graph entries only passed through at load time: map the graph file now (a streamed load
  // from an entry source has nothing to map: error)
  if (!dev_graph && !e->search_v2 && !e->walker_rows && !e->graph) BANG_TRY(map_graph_file(e));
  e->fp_direct = false;
  HIP_TRY_ALLOC(hipMalloc(&e->d_queries, nq * e->D * e->tsize + 16));
  if (e->psz) BANG_TRY(dmalloc(&e->d_qc, nq * e->mp * e->psz));
  else BANG_TRY(dmalloc(&e->d_lut, nq * e->m * 256));                       // :380
  // :393 (bit-packed: 8x smaller).  BANG_FILTER_MEM: 1 = uncached, 2 = fine-grained device memory (experiment: does a filter probe
  // still move a whole 128-byte line when L2 does not cache the filter?  DESIGN 4.6)
  {
    const long fm = env_long("BANG_FILTER_MEM", 0);
    if (fm == 1 || fm == 2) {
      HIP_TRY_ALLOC(hipExtMallocWithFlags((void**)&e->d_bloom, std::max<size_t>(nq * BANG_BF_WORDS * 4, 16), fm == 1 ? hipDeviceMallocUncached : hipDeviceMallocFinegrained));
    } else BANG_TRY(dmalloc(&e->d_bloom, nq * BANG_BF_WORDS));
  }
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
    HIP_TRY(hipHostGetDevicePointer((void**)&e->h_results_dev, e->h_results, 0));
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
      HIP_TRY_ALLOC(hipMalloc((void**)&e->d_fp, rows * nq * vb));                  // :398
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
      HIP_TRY_ALLOC(hipMalloc((void**)&ln.d_ktime, ln.kt_launches * KT_WGS * 16));     // {start, end} stamp per workgroup and launch
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
  if (env_flag("BANG_DEBUG")) {
    if (e->numa_on) fprintf(stderr, "[bang] walker threads pinned to NUMA node %d (%d usable CPUs)\n", e->numa_node, CPU_COUNT(&e->numa_cpus));
    else if (!dev_graph) fprintf(stderr, "[bang] walker threads not pinned\n");
  }
  if (env_flag("BANG_DEBUG"))
    fprintf(stderr, "[bang] alloc Q=%d lanes=%d threads=%d stage_mode=%d search_kernel=%d/%d fp_direct=%d vec_on_device=%d\n", Q, nl,
            e->threads_eff, e->stage_mode_eff, (int)e->search_v2, (int)e->search_host, (int)e->fp_direct, (int)e->vec_on_device);
  start_threads(e);
  return BANG_OK;
}

}  // namespace bang
