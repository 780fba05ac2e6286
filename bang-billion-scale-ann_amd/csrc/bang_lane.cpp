// bang_lane.cpp -- bang_query of ONE lane (bang_search.cu:569-1068): queries H2D -> K1 -> the search loop in one of its forms -> re-rank -> results.
// Reference line numbers: /root/reference/BANG_Base/bang_search.cu.
#include "bang_engine.h"

namespace bang {

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
  p.code_stride = e->code_stride;
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

// diagnostic build (-DBANG_SEARCH_PHASE_PROF): per-iteration phase times of wave 0 of every workgroup, slots [8..15] of its record
static void print_phase_prof(const std::vector<unsigned long long>& pr, uint32_t G) {
  double a[8] = {0, 0, 0, 0, 0, 0, 0, 0};
  for (uint32_t w = 0; w < G; ++w) for (int k = 0; k < 8; ++k) a[k] += (double)pr[(size_t)w * 16 + 8 + k];
  if (a[7] <= 0) return;
  const double n = a[7];
  double x[3] = {0, 0, 0};                     // self-paced form: [0] hand-over (row request), [1] worklist head, [2] summary marks -- split off phases 6 / 0
  for (uint32_t w = 0; w < G; ++w) for (int k = 0; k < 3; ++k) x[k] += (double)pr[(size_t)w * 16 + k];
  fprintf(stderr, "[search] phases of an iteration (wave 0 of %u workgroups, %.0f iterations each): row arrival + loop %.2f us, hashes + probes %.2f, "
                  "compaction %.2f, filter update %.2f, code rows + distances %.2f, parent %.2f, publish + sort/merge %.2f (self-paced, split off: hand-over %.2f, "
                  "worklist head %.2f, summary marks %.2f)\n", G, n / G,
          a[0] * 0.01 / n, a[1] * 0.01 / n, a[2] * 0.01 / n, a[3] * 0.01 / n, a[4] * 0.01 / n, a[5] * 0.01 / n, a[6] * 0.01 / n, x[0] * 0.01 / n, x[1] * 0.01 / n,
          x[2] * 0.01 / n);
}

static int g_dbg = -1;
#define DBG(...) do { if (g_dbg < 0) g_dbg = env_flag("BANG_DEBUG") ? 1 : 0; if (g_dbg) { fprintf(stderr, __VA_ARGS__); fflush(stderr); } } while (0)

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
  const bool tl_on = env_flag("BANG_TIMELINE");
  auto tl_t = Clock::now();
  auto tl = [&](const char* what) {
    if (!tl_on) return;
    (void)hipStreamSynchronize(ln.s_main);
    fprintf(stderr, "[timeline lane %d] %-28s %8.1f us\n", ln.index, what, ms_since(tl_t) * 1000.0);
    tl_t = Clock::now();
  };
  // K6 + K7 inside the search launch (8-bit vectors, self-paced form): no launch behind it
  const bool fused_rerank = e->search_v2 && e->fuse_rerank != 0 && (dev_graph || e->vec_on_device) &&
                            bang_search_can_rerank(e->dtype, e->D, dev_graph ? e->entry_len : vb, dim_adjust) != 0;
  const bool whole = ln.q0 == 0 && (int)ln.nq == Q && (int)ln.nq == e->Qcur;
  const size_t mailbox_max = (size_t)env_long("BANG_MAILBOX_BYTES", BANG_RESULT_MAILBOX_BYTES);
  uint64_t* d_ids_user = e->pool.d_ids_user;
  float* d_dists_user = e->pool.d_dists_user;
  const bool to_device = d_ids_user != nullptr;               // bang_query_dev_e: no result leaves the device
  const bool mailbox = !to_device && whole && e->res_off_iters <= mailbox_max;
  // ... and with the re-rank fused into the search launch the kernel writes ids, distances, iteration counts and its abort word straight
  // into that pinned mirror (posted PCIe writes, a query at a time as the queries finish): nothing is copied behind the launch
  const bool results_direct = mailbox && fused_rerank && e->h_results_dev != nullptr && env_long("BANG_RESULTS_DIRECT", 1) != 0;
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
  uint32_t* h_abort = (uint32_t*)(e->h_results + e->res_bytes - BANG_MAX_LANES * 4) + ln.index;   // (one word per lane behind the results)
  *h_abort = 0;
  if (e->search_v2) {
    // graph resident in HBM: ONE launch of the query-resident search kernel; no host involvement until the re-rank
    LANE_HIP(hipMemsetAsync(ln.d_pcnt, 0, 64, ln.s_main));
    bang_search_params sp;
    memset(&sp, 0, sizeof(sp));
    sp.Q = ln.nq; sp.R = e->R; sp.m = e->m; sp.L = (uint32_t)e->L; sp.medoid = (uint32_t)e->medoid; sp.cap_iter = cap_iter;
    sp.psz = e->psz; sp.mp = e->mp; sp.pq_nhi = e->pq_nhi;
    sp.d_seed = e->d_seed; sp.d_codes = e->d_codes; sp.code_stride = e->code_stride; sp.d_pivots_packed = p.d_pivots_packed; sp.d_qc = p.d_qc;
    sp.d_graph = e->d_graph; sp.entry_len = e->entry_len; sp.vec_bytes = (uint32_t)vb;
    if (!dev_graph) {                                                        // pull mode
      sp.d_graph = (const uint8_t*)e->d_adj; sp.entry_len = 256; sp.vec_bytes = 0; sp.row_layout = 1;
      sp.d_rows_hbm = e->d_rows_hbm; sp.n_rows_hbm = e->rows_first == 0 ? e->n_rows_hbm : 0;      // (a moved slice is only reachable through the table)
      if (e->n_slices > 1 && e->d_slice_tab) { sp.d_row_slices = e->d_slice_tab; sp.n_slices = e->n_slices; sp.slice_rows = e->slice_rows; }
    }
    sp.d_bloom = p.d_bloom; sp.d_cand_ids = p.d_cand_ids; sp.d_cand_cnt = p.d_cand_cnt; sp.d_qstats = p.d_qstats;
    sp.d_qiters = e->d_qiters + ln.q0; sp.d_next_query = ln.d_pcnt; sp.d_abort = ln.d_pcnt + 1; sp.n_nodes = e->N;
    sp.d_qskip = e->d_qskip + ln.q0;
    sp.spec_rows = (uint32_t)std::min(2L, std::max(0L, env_long("BANG_SPEC_ROWS", 0)));
    e->rerank_fused = fused_rerank;
    if (fused_rerank) {                                                      // K6 + K7 by the wave that finishes the query: no launch behind this one
      sp.rr_queries = e->d_queries;
      sp.rr_dtype = (uint32_t)e->dtype; sp.rr_D = e->D; sp.rr_k = (uint32_t)e->k; sp.rr_q0 = ln.q0; sp.rr_Q_total = (uint32_t)Q;
      sp.rr_vec_base = dev_graph ? e->d_graph : e->d_vecs; sp.rr_vec_stride = dev_graph ? e->entry_len : vb;
      sp.rr_ids_out = e->d_ids_out; sp.rr_dists_out = e->d_dists_out;
      if (to_device && whole) {                                       // bang_query_dev_e: straight into the caller's device buffers
        sp.rr_ids_out = d_ids_user;
        if (d_dists_user) sp.rr_dists_out = d_dists_user;
      }
      if (results_direct) {
        sp.rr_ids_out = (uint64_t*)e->h_results_dev; sp.rr_dists_out = (float*)(e->h_results_dev + e->res_off_dists);
        sp.d_qiters = (uint32_t*)(e->h_results_dev + e->res_off_iters) + ln.q0;
        sp.d_abort = (uint32_t*)(e->h_results_dev + e->res_bytes - BANG_MAX_LANES * 4) + ln.index;
      }
    }
    { const long si = env_long("BANG_SUMM_ITERS", 0); sp.summ_iters = si < 0 ? 0xFFFFFFFFu : (uint32_t)si; }     // 0 = auto, -1 = always
    sp.d_ktime = ktime_slot(e, ln);
    sp.max_wgs = (uint32_t)std::max(0L, env_long("BANG_SEARCH_MAX_WGS", 0));          // experiment / test knobs
    sp.max_waves = (uint32_t)std::max(0L, env_long("BANG_SEARCH_MAX_WAVES", 0));
    const bool kprof_d = env_flag("BANG_SEARCH_PROF");   // diagnostic build only (-DBANG_SEARCH_PHASE_PROF)
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
      uint32_t C = (uint32_t)std::max(0L, env_long("BANG_SEARCH_CTX", 0)), GS = (uint32_t)std::max(0L, env_long("BANG_SEARCH_GS", 0));
      BANG_TRY(bang_search_geometry(e->psz, e->mp, e->pq_nhi, (uint32_t)e->L, ln.nq, (uint32_t)std::max(0L, env_long("BANG_SEARCH_MAX_WGS", 0)),
                                    (uint32_t)std::max(0L, env_long("BANG_SEARCH_MAX_WAVES", 0)), 1, &G, &W, &C, &GS));
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
    sp.d_seed = e->d_seed; sp.d_codes = e->d_codes; sp.code_stride = e->code_stride; sp.d_pivots_packed = p.d_pivots_packed; sp.d_qc = p.d_qc;
    sp.d_graph = nullptr; sp.entry_len = e->entry_len; sp.vec_bytes = (uint32_t)vb;
    sp.d_bloom = p.d_bloom; sp.d_cand_ids = p.d_cand_ids; sp.d_cand_cnt = p.d_cand_cnt; sp.d_qstats = p.d_qstats;
    sp.d_qiters = e->d_qiters + ln.q0; sp.d_next_query = ln.d_pcnt; sp.d_abort = ln.d_pcnt + 1;
    // walker-from-rows: parents whose row sits in this GPU's HBM copy of the first rows are served by the kernel itself
    e->walker_self = e->walker_rows && e->d_rows_hbm && e->rows_first == 0 && env_long("BANG_WALKER_SELF_ROWS", 1) != 0;
    if (e->walker_self) { sp.d_rows_hbm = e->d_rows_hbm; sp.n_rows_hbm = e->n_rows_hbm; }
    sp.d_ktime = ktime_slot(e, ln);
    sp.d_rows = e->d_srows; sp.d_ctl = e->d_sctl; sp.h_done = e->h_done_dev; sp.h_parents = e->d_parents_map;
    sp.h_pub_q = e->d_pub_q; sp.h_pub_c = e->d_pub_c; sp.ship_vectors = e->vec_on_device ? 0u : 1u;
    sp.go_timeout_ticks = (unsigned long long)e->kernel_go_timeout_ms * 100000ull;
    const bool kprof = env_flag("BANG_SEARCH_PROF");     // diagnostic: phase times of the half-rounds (stderr)
    unsigned long long* d_prof = nullptr;
    if (kprof) { LANE_HIP(hipMalloc((void**)&d_prof, (size_t)G * 128)); LANE_HIP(hipMemsetAsync(d_prof, 0, (size_t)G * 128, ln.s_main)); sp.d_prof = d_prof; }
    BANG_TRY(bang_k_search(&sp, ln.s_main));
    ++ln.front_launches;
    const auto t0 = Clock::now();
    const int T = 1 + (int)ln.helpers.size();
    if (e->walker_stall_ms > 0) {                // test hook: nobody serves the kernel for a while (one shot)
      std::this_thread::sleep_for(std::chrono::milliseconds(e->walker_stall_ms));
      e->walker_stall_ms = 0;
    }
    ln.job_kind = 2;
    if (T > 1) {
      ln.pending.store((uint32_t)(T - 1), std::memory_order_relaxed);
      ln.epoch.fetch_add(1, std::memory_order_release);
    }
    swalk(e, ln, 0, T);
    if (T > 1) while (ln.pending.load(std::memory_order_acquire) != 0) _mm_pause();
    ln.walker_ms += ms_since(t0);
    if (ln.pw_error.load()) {
      // one side of the hand-shake gave up: the kernel has been stopped through its control lines (or has left by itself)
      (void)hipStreamSynchronize(ln.s_main);
      uint32_t gave_up = 0;
      (void)hipMemcpy(&gave_up, ln.d_pcnt + 1, 4, hipMemcpyDeviceToHost);
      if (gave_up) bang_set_error("the search kernel gave up waiting for the host walker (kernel_go_timeout_ms = %d)", e->kernel_go_timeout_ms);
      else bang_set_error("the walker gave up waiting for the search kernel (host_walk_timeout_ms = %d)", e->host_walk_timeout_ms);
      return BANG_ERR_HIP;
    }
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
  if (!fused_rerank) {
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
  if ((e->search_host || e->search_v2) && !results_direct) LANE_HIP(hipMemcpyAsync(h_abort, ln.d_pcnt + 1, 4, hipMemcpyDeviceToHost, ln.s_main));
  const bool iters = e->search_v2 || e->search_host;
  if (to_device) {
    const bool in_place = fused_rerank && whole;                // (the fused re-rank wrote into the caller's buffers)
    if (!in_place)
      LANE_HIP(hipMemcpyAsync(d_ids_user + (size_t)ln.q0 * e->k, e->d_ids_out + (size_t)ln.q0 * e->k, (size_t)ln.nq * e->k * sizeof(uint64_t),
                              hipMemcpyDeviceToDevice, ln.s_main));
    if (d_dists_user && !in_place)
      LANE_HIP(hipMemcpy2DAsync(d_dists_user + ln.q0, (size_t)Q * 4, e->d_dists_out + ln.q0, (size_t)Q * 4, (size_t)ln.nq * 4, (size_t)e->k,
                                hipMemcpyDeviceToDevice, ln.s_main));
    if (iters) LANE_HIP(hipMemcpyAsync(e->h_results + e->res_off_iters + (size_t)ln.q0 * 4, e->d_qiters + ln.q0, (size_t)ln.nq * 4,
                                       hipMemcpyDeviceToHost, ln.s_main));
  } else if (results_direct) {                                  // (already there)
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
  if (pw_stats[0] == 2u) {
    bang_set_error("search kernel: an adjacency row named a node id out of range (N = %u): the rows were overwritten since bang_load%s%s", e->N,
                   e->rows_path.empty() ? "" : " -- ", e->rows_path.c_str());
    return BANG_ERR_HIP;
  }
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


}  // namespace bang
