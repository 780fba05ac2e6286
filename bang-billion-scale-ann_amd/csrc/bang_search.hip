// bang_search.hip -- the QUERY-RESIDENT search kernel: the whole search loop of a batch (bang_search.cu:569-1068) in ONE launch
// in which a wavefront owns ONE query at a time from its first iteration to its last and keeps that query's state on chip.
//
// What differs from the round-1 persistent kernel (front_kernel<PERSIST> in bang_kernels.hip), and why:
//  * Ownership is dynamic: a wave that finishes a query pulls the next unstarted one from a device counter.  No workgroup waits
//    for its slowest query while its other waves idle (round 1: 119 iterations of the slowest block vs a median of 75), and at
//    most (CUs x waves) queries are alive at a time, so their visited filters (50 KB each) stay within reach of the Infinity Cache.
//  * The worklist (K3b, :1605-1715), the survivors and every per-query counter live in LDS / registers for the life of the
//    query: the merge reads and writes no global memory, K4 (:1384-1521) reads the worklist head from LDS, and the only HBM
//    traffic of an iteration is what the algorithm needs -- the adjacency row, the filter words, the PQ code rows, one
//    candidate-log word.
//  * The visited filter is updated with PLAIN stores.  A query's filter is private to its wave, so atomics are only needed for
//    two lanes of the SAME wave-instruction that hit the same word.  Those are found with a 128 / 256-slot claim table in LDS (three
//    rounds with different hashes); same-word lanes merge their bits there and one lane stores old | bits.  The rare lanes that
//    lose all three rounds to other words fall back to an atomic OR after the plain stores have drained.  (Round 1 issued
//    73 M fully scattered atomic ORs per 10 K batch: ~17x below the rate of plain stores of the same shape on MI355X.)  The stores
//    are issued once the survivors' code rows have arrived: loads and stores share one counter, and a store in front of the wait
//    for the rows makes that wait cover its acknowledgement.
//  * Both filter words of an id are probed in one round trip -- except words the wave's FilterSummary (one bit per filter word,
//    six VGPRs) knows it has never stored to: those are zero without asking.
//  * Code rows of three or more 16-byte pieces are fetched cooperatively (CoopFetch: adjacent lanes ask for one row, one request
//    per line) and reduced as a two-stage LDS pipeline (pq_row_reduce_pipe).
//  * In graph-on-HBM mode the next adjacency row is requested the moment the parent is known, so its latency hides behind the
//    sort/merge.  In host-graph mode the parents of a workgroup's waves go to the host walker in one coalesced store per round,
//    and the sort/merge overlaps the walker's round trip.
//
//  * A launch of at most 5 queries per CU leaves the FilterSummary off (its LDS-crossbar work sits on the chain of every iteration and
//    a lightly loaded chip is not short of requests); an adjacency id >= N is never followed (n_nodes: the batch ends with an error
//    instead of a wild read).  Experiments that measured no faster live in git history and docs/HISTORY.md, not here.
//  * Round 5: K6 + K7 by the wave that finishes a query (wave_rerank8, bang_device.h: 8-bit vectors); the query replicated per 16-lane row
//    for the BASELINE long-row layouts (QcRow16: "pivot - query" is one DPP instruction); peer rows (a table of HBM slices of the adjacency
//    rows, this GPU's or a peer's over xGMI); arguments a query needs once read from the kernarg segment where they are used (KARG).
//  * Round 6: what an iteration needs of the arguments travels in the lanes of one vector register (IterArgs) instead of being re-read from the kernarg
//    segment on the chain; every pointer the kernel rebuilds is typed global (GAS: no flat_ instruction is left -- a flat_ load counts in lgkmcnt and
//    makes every LDS wait a wait for the loads in flight); code rows with the non-temporal hint; K6 + K7 for float vectors too (wave_rerank_f32);
//    the instances compiled in two translation units (BANG_SEARCH_PART: the BASELINE layouts under LLVM's iterative-ILP scheduling strategy).
//
// Results are bit-identical to the per-iteration kernels and to the oracle: the per-query algorithm (Appendix B of SURVEY.md,
// canonical semantics of DESIGN.md section 2) is unchanged, only where its state lives and who schedules it.
//
// Reference line numbers: /root/reference/BANG_Base/bang_search.cu.

#include <hip/hip_runtime.h>
#include <type_traits>
#include <utility>
#include <stdint.h>
#include <stdlib.h>
#include <algorithm>

#include "bang_c.h"
#include "bang_internal.h"
#include "bang_device.h"
#include "bang_worklist.h"

struct SearchArgs;
// A kernel-argument field read WHERE IT IS USED (a scalar load from the kernarg segment through a pointer the optimiser cannot see through),
// for arguments that are needed once per query (the seed list at its start, the counters and the re-rank at its end): read the ordinary way they are loop-invariant, get
// hoisted out of the search loop and occupy scalar registers the loop has none to spare of (106 of 106: they spill into VGPR lanes).
template <class T>
__device__ __forceinline__ T karg_at(size_t off) {
  const char __attribute__((address_space(4)))* ka = (const char __attribute__((address_space(4)))*)__builtin_amdgcn_kernarg_segment_ptr();
  asm volatile("" : "+s"(ka));
  return *(const T __attribute__((address_space(4)))*)(ka + off);
}
// One 64-bit word of a small device table through the SCALAR cache (uniform index).  Written as a plain load the compiler makes it a vector
// load -- it cannot prove the index uniform under the hand-over's branches -- and waits for it with vmcnt(0): behind every filter store still in
// flight, on the chain of every iteration (peer rows: the slice table).  The wait is inside the statement: nothing can be scheduled between the
// request and it.
__device__ __forceinline__ uint64_t scalar_load_u64(uint64_t tab, uint32_t idx) {
  uint64_t v;
  const uint64_t a = tab + 8ull * idx;                  // (uniform, but not provably so: made so)
  const uint64_t at = ((uint64_t)uni((uint32_t)(a >> 32)) << 32) | uni((uint32_t)a);
  asm volatile("s_load_dwordx2 %0, %1, 0x0\n\ts_waitcnt lgkmcnt(0)" : "=s"(v) : "s"(at) : "memory");
  return v;
}
#define KARG(field) karg_at<decltype(bang_search_params::field)>(offsetof(SearchArgs, p) + offsetof(bang_search_params, field))
// ... a pointer field, as a pointer to global memory (bang_device.h GAS: a pointer read this way is generic to the compiler otherwise, its accesses flat_)
#define KARGP(type, field) ((type GAS*)(uintptr_t)KARG(field))

// What an ITERATION needs of the launch's arguments (as opposed to a query: KARG) -- where the code table, the filters, the candidate log and the next
// adjacency row are -- travels in the LANES OF ONE VECTOR REGISTER of the wave (lane k = dword k of this block, loaded once at the wave's start) and is
// read with v_readlane where it is used: no memory round trip.  History of these ~24 dwords: as ordinary kernel arguments they are loop-invariant, get
// hoisted and take scalar registers the search loop has none to spare of (106 of 106: spills, or re-loads from the kernarg segment the compiler places
// where it pleases: d_bloom, summ_iters and cap_iter were re-read on the chain of every iteration); read from the kernarg segment one by one where the
// branches need them (round 5: KARG) the hand-over alone was five dependent scalar-cache round trips between "the parent is known" and "its row is
// requested" (as one block: 1 250-query SIFT1B-shape shard 1.63 -> 1.53 ms, 10 K batch 8.32 -> 8.15).  The register passes through an empty asm in front
// of each use site so that the reads stay THERE (a v_readlane of a loop-invariant register is loop-invariant too and would be hoisted back into scalar
// registers).  Pointers rebuilt from these dwords are typed GAS (bang_device.h): generic, every access through them is a flat_ instruction.
struct IterArgs {
  const uint8_t* d_codes; uint32_t n_nodes, prio;                                               // dwords 0-1, 2, 3
  uint32_t* d_cand_ids; const uint8_t* d_graph; uint64_t entry_len;                             // 4-5, 6-7, 8-9
  uint32_t vec_bytes, row_layout, n_rows_hbm, n_slices;                                         // 10, 11, 12, 13
  const uint32_t* d_rows_hbm; const uint64_t* d_row_slices;                                     // 14-15, 16-17
  uint32_t slice_rows, summ_iters;                                                              // 18, 19
  uint32_t* d_bloom; uint32_t cap_iter, pad1;                                                   // 20-21, 22, 23
};
enum { IA_CODES = 0, IA_N_NODES = 2, IA_PRIO = 3, IA_CAND_IDS = 4, IA_GRAPH = 6, IA_ENTRY_LEN = 8, IA_VEC_BYTES = 10, IA_ROW_LAYOUT = 11, IA_N_ROWS_HBM = 12,
       IA_N_SLICES = 13, IA_ROWS_HBM = 14, IA_ROW_SLICES = 16, IA_SLICE_ROWS = 18, IA_SUMM_ITERS = 19, IA_BLOOM = 20, IA_CAP_ITER = 22, IA_DWORDS = 24 };
static_assert(sizeof(IterArgs) == 4 * IA_DWORDS && offsetof(IterArgs, d_cand_ids) == 4 * IA_CAND_IDS && offsetof(IterArgs, vec_bytes) == 4 * IA_VEC_BYTES &&
              offsetof(IterArgs, d_rows_hbm) == 4 * IA_ROWS_HBM && offsetof(IterArgs, slice_rows) == 4 * IA_SLICE_ROWS && offsetof(IterArgs, d_bloom) == 4 * IA_BLOOM && offsetof(IterArgs, cap_iter) == 4 * IA_CAP_ITER,
              "IterArgs dword map");
struct SearchArgs {
  bang_search_params p;
  IterArgs iter __attribute__((aligned(16)));
  uint32_t lds_piv_floats;
  uint32_t wave_words;       // LDS words per wave: nctx worklists + 144 scratch (+ 32 parked context state when nctx == 2)
  uint32_t wl_words;         // LDS words of one worklist (2L + ceil(L/4), rounded to 4)
  uint32_t nctx;             // query contexts per wave: 1, or 2 in the host-paced form
  uint32_t gs;               // host-paced form: waves per pacing group (a workgroup's waves advance in lock-step per GROUP)
};

// Code rows are fetched cooperatively (CoopFetch, bang_device.h) from three 16-byte pieces per row on (rows of 12+ code dwords,
// m > 44): two-piece rows (m = 32) gain nothing in the kernel (2.69 vs 2.70 ms on SIFT1M-like) and would only lose LDS to the staging
// area.  The host-paced instances do so for the long-row layouts only (12 waves x 168 VGPRs, with the filter summary).
__host__ __device__ constexpr bool search_coop(int ndw, bool host_paced) { return ndw >= 12 && (!host_paced || ndw >= 16); }
// self-paced instances with the speculative row request (SPEC) exist for the long-row layouts of the BASELINE configs (70 / 74 chunks: 168-VGPR
// instances); the 128-VGPR instance of the 32-chunk layout measured 2-5 % slower with it
__host__ __device__ constexpr bool search_has_spec(int ndw) { return ndw == 18 || ndw == 19; }
// per-wave scratch: sd/ti [72] + td/compaction [72]; the filter claim table (128 slots; 256 where the scratch has them) and the
// summary's transposition area alias both, and so does the staging area of the cooperative code-row fetch (256 words: one wave
// instruction's worth of 16-byte pieces)
__host__ __device__ constexpr int search_maxt(int ndw, bool host_paced);
__host__ __device__ constexpr uint32_t search_scratch_words(int ndw, bool host_paced);

__host__ __device__ inline uint32_t search_wl_words(uint32_t L) { return (2u * L + (L + 3u) / 4u + 3u) & ~3u; }
__host__ __device__ inline uint32_t search_wave_words(uint32_t L, uint32_t nctx, int ndw, bool host_paced) {
  return nctx * search_wl_words(L) + search_scratch_words(ndw, host_paced) + (nctx == 2 ? 32u : 0u);
}

// ---------------------------------------------------------------------------------------------------------------------
// K5, second half (:1159-1160): set the two filter bits of every survivor -- plain stores, same-word lanes merged in LDS
// ---------------------------------------------------------------------------------------------------------------------
// Every lane brings up to two items (word index, bit mask, the word's value as probed in THIS iteration).  All probes of the
// iteration were issued before this point and nothing was stored in between, so two lanes that hit the same word hold the same
// old value.  tbl: 128 LDS words private to the wave, used as a claim table of 128 slots.  A round: every pending item writes its
// tag {word, lane, a|b} into the slot its word hashes to and reads the slot back -- the last writer OWNS the slot and stores
// old | bit.  An item that finds another WORD in its slot retries in the next round (another hash); one that finds its own word
// under another tag (rare: ~0.3 per iteration) has its bit merged into the owner's store through the same slot.  Nothing needs
// initialising: whoever reads a slot has just written it, so its content is this round's.  Returns with every item stored,
// merged into another lane's store, or (pa / pb still set; ~0.03 per iteration) left for the atomic fallback.
// The stores themselves are left to the caller (st_a / st_b: this lane stores va / vb to word ia / ib): issued here they would sit
// between the code-row loads already in flight and the wait for those rows, and that wait would then cover their acknowledgements too
// (one counter for loads and stores on this target).
template <int SLOTS>                                   // claim table slots (a power of two: 128, or 256 where the wave's scratch has them)
__device__ __forceinline__ void filter_commit(uint32_t* tbl, int lane, bool& pa, uint32_t ia, uint32_t ba, uint32_t wa, bool& pb,
                                              uint32_t ib, uint32_t bb, uint32_t wb, bool& st_a, uint32_t& out_a, bool& st_b,
                                              uint32_t& out_b) {
  const uint32_t tag_a = (ia << 7) | ((uint32_t)lane << 1), tag_b = (ib << 7) | ((uint32_t)lane << 1) | 1u;
#pragma unroll
  for (int r = 0; r < 3; ++r) {
    if (__ballot(pa || pb) == 0) break;                          // uniform
    const uint32_t mul = r == 0 ? 0x9E37u : r == 1 ? 0x85EBu : 0xC2B3u;
    const uint32_t sa = ((ia * mul) >> 9) & (uint32_t)(SLOTS - 1), sb = ((ib * mul) >> 9) & (uint32_t)(SLOTS - 1);
    if (pa) tbl[sa] = tag_a;
    if (pb) tbl[sb] = tag_b;                                      // a later instruction: wins over this lane's own item a
    wave_sync();
    const uint32_t ra = tbl[sa], rb = tbl[sb];
    const bool own_a = pa && ra == tag_a, own_b = pb && rb == tag_b;
    const bool same_a = pa && !own_a && (ra >> 7) == ia, same_b = pb && !own_b && (rb >> 7) == ib;
    wave_sync();
    uint32_t va = ba, vb = bb;
    if (__ballot(same_a || same_b)) {                             // uniform, rare: the slots now collect the bits of their word
      if (own_a) tbl[sa] = ba;
      if (own_b) tbl[sb] = bb;
      wave_sync();
      if (same_a) (void)__hip_atomic_fetch_or(&tbl[sa], ba, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_WAVEFRONT);   // ds_or_b32
      if (same_b) (void)__hip_atomic_fetch_or(&tbl[sb], bb, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_WAVEFRONT);
      wave_sync();
      if (own_a) va = tbl[sa];
      if (own_b) vb = tbl[sb];
      wave_sync();
    }
    if (own_a && (wa | va) != wa) { st_a = true; out_a = wa | va; }      // (a survivor has at most one of its two bits set already: that store is moot)
    if (own_b && (wb | vb) != wb) { st_b = true; out_b = wb | vb; }
    pa = pa && !(own_a || same_a);
    pb = pb && !(own_b || same_b);
  }
}

// ---------------------------------------------------------------------------------------------------------------------
// K5: exact on-chip summary "this filter word has been stored to" -- one bit per 32-bit word of the query's filter
// ---------------------------------------------------------------------------------------------------------------------
// A query's filter starts all-zero (bang_init :443) and only its own wave ever stores to it, so the wave knows which words are still
// zero without asking memory.  A probe of such a word needs no load (the bit is clear: the id passes, :1157) and a survivor's store
// to it needs no old value (old | bit == bit).  The hash positions and the snapshot semantics (:1140-1165) are untouched; only
// requests whose answer is known are dropped (with ~7 K of 400 K slots set at the end of a SIFT1M-like query, ~3/4 of all probed
// words are still zero when probed).
// Layout: 12 288 bits in SIX VGPRs of the owning wave -- word w lives in lane (w & 63), position p = w >> 6 (0..195; the 209 words
// with p >= 192 share the positions p - 192: a set bit then also covers an alias, which only costs that word its shortcut),
// register p >> 5, bit p & 31.  Read: six ds_bpermute (no LDS memory touched) + selects.  Set: the survivors' bits are transposed
// through 128 (256) words of the wave's LDS scratch, two (four) registers per pass (ds_or_b32), and OR-ed into the registers.
#ifndef BANG_FILTER_SUMMARY
#define BANG_FILTER_SUMMARY 1       // build switch (A/B libraries): 0 = every probe is loaded
#endif
template <int SUMM_REGS>                               // 6: one bit per filter word
struct FilterSummary {
  uint32_t s[SUMM_REGS];
  __device__ __forceinline__ void clear() {
#pragma unroll
    for (int r = 0; r < SUMM_REGS; ++r) s[r] = 0u;
  }
  static __device__ __forceinline__ uint32_t pos_of(uint32_t w) { const uint32_t p = w >> 6; return p >= 32u * SUMM_REGS ? p - 32u * SUMM_REGS : p; }
  // has word w been stored to?  (every lane asks about its own w; all lanes of the wave must be executing)
  __device__ __forceinline__ bool test(uint32_t w) const {
    const uint32_t p = pos_of(w);
    const int src = (int)((w & 63u) << 2);
    uint32_t v = 0;
#pragma unroll
    for (int r = 0; r < SUMM_REGS; ++r) {
      const uint32_t t = (uint32_t)__builtin_amdgcn_ds_bpermute(src, (int)s[r]);
      v = ((p >> 5) == (uint32_t)r) ? t : v;
    }
    return ((v >> (p & 31u)) & 1u) != 0u;
  }
  // mark up to three words per lane (a, b: this lane's survivor; c: the 65th id of the seed list, lane 0) as stored to.
  // RP = summary registers transposed per pass = words of LDS scratch / 64: 2 (three passes), or 4 where the wave has 256 words (two)
  template <int RP>
  __device__ __forceinline__ void set(uint32_t* tbl /* 64 RP LDS words of the wave */, int lane, bool ha, uint32_t wa, bool hb, uint32_t wb,
                                      bool hc, uint32_t wc0, uint32_t wc1) {
    const uint32_t pa = pos_of(wa), pb = pos_of(wb), pc0 = pos_of(wc0), pc1 = pos_of(wc1);
#pragma unroll
    for (int pass = 0; pass < (SUMM_REGS + RP - 1) / RP; ++pass) {
      if (RP == 2 ? 2 * pass >= SUMM_REGS : 4 * pass >= SUMM_REGS) break;
      const bool ia = ha && (pa >> 5) / RP == (uint32_t)pass, ib = hb && (pb >> 5) / RP == (uint32_t)pass;
      const bool ic0 = hc && (pc0 >> 5) / RP == (uint32_t)pass, ic1 = hc && (pc1 >> 5) / RP == (uint32_t)pass;
      if (__ballot(ia || ib || ic0 || ic1) == 0) continue;                         // uniform
      if (RP == 2) *(uint2*)(tbl + 2 * lane) = make_uint2(0u, 0u);
      else *(uint4*)(tbl + 4 * lane) = make_uint4(0u, 0u, 0u, 0u);
      wave_sync();
      if (ia) (void)__hip_atomic_fetch_or(&tbl[(wa & 63u) * RP + ((pa >> 5) % RP)], 1u << (pa & 31u), __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_WAVEFRONT);
      if (ib) (void)__hip_atomic_fetch_or(&tbl[(wb & 63u) * RP + ((pb >> 5) % RP)], 1u << (pb & 31u), __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_WAVEFRONT);
      if (ic0) (void)__hip_atomic_fetch_or(&tbl[(wc0 & 63u) * RP + ((pc0 >> 5) % RP)], 1u << (pc0 & 31u), __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_WAVEFRONT);
      if (ic1) (void)__hip_atomic_fetch_or(&tbl[(wc1 & 63u) * RP + ((pc1 >> 5) % RP)], 1u << (pc1 & 31u), __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_WAVEFRONT);
      wave_sync();
      if (RP == 2) {
        const uint2 v = *(const uint2*)(tbl + 2 * lane);
        s[2 * pass] |= v.x;
        if (2 * pass + 1 < SUMM_REGS) s[2 * pass + 1] |= v.y;
      } else {
        const uint4 v = *(const uint4*)(tbl + 4 * lane);
        s[4 * pass] |= v.x;
        if (4 * pass + 1 < SUMM_REGS) s[4 * pass + 1] |= v.y;
        if (4 * pass + 2 < SUMM_REGS) s[4 * pass + 2] |= v.z;
        if (4 * pass + 3 < SUMM_REGS) s[4 * pass + 3] |= v.w;
      }
      wave_sync();
    }
  }
};

// ---------------------------------------------------------------------------------------------------------------------
// the kernel
// ---------------------------------------------------------------------------------------------------------------------
// Barrier among the `n` waves of a pacing group (a subset of the workgroup, so s_barrier cannot be used): arrival counter +
// generation word in LDS.  Everything the group hands over at the barrier lives in LDS, and a wave's LDS operations execute in
// program order, so draining the wave's own LDS queue before it arrives is all the ordering that is needed.
__device__ __forceinline__ void group_barrier(uint32_t* bar, uint32_t n, int lane) {
  asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory");
  if (n > 1 && lane == 0) {
    const uint32_t gen = __hip_atomic_load(&bar[1], __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_WORKGROUP);
    const uint32_t arrived = __hip_atomic_fetch_add(&bar[0], 1u, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_WORKGROUP) + 1u;
    if (arrived == n) {
      __hip_atomic_store(&bar[0], 0u, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_WORKGROUP);
      asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory");
      __hip_atomic_store(&bar[1], gen + 1u, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_WORKGROUP);
    } else {
      while (__hip_atomic_load(&bar[1], __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_WORKGROUP) == gen) __builtin_amdgcn_s_sleep(1);
    }
  }
  asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory");
  wave_sync();
}

// HOST = false: graph resident in HBM.  Every wave runs on its own: no barrier, no host involvement.
// HOST = true : graph in host RAM (the north-star path).  The waves of a workgroup form PACING GROUPS of a.gs waves (default 8: two
//   groups per workgroup, which run independently of each other -- while one waits for the host the other computes).  The waves
//   of a group advance in lock-step HALF-ROUNDS.  A wave holds a.nctx (default 1; 2 measured slower) query CONTEXTS; half-round h
//   works on context c = h mod nctx of every wave of the group:
//     wait until the host walker has delivered context c's adjacency rows of its previous round (ctl[g].go, g = nctx * wg + c)
//     -> every wave runs one iteration of its context-c query: K5, K2, K4 (front), then K3 (sort/merge)
//     -> between the two the workgroup publishes the <= 16 parents in ONE coalesced store to mapped host memory, followed by the
//        round number (h_done[16 g]).
//   With two contexts the walker's round trip for context c (PCIe write of the parents, gather of 16 graph entries by a CPU
//   thread, PCIe writes of the rows through the BAR) overlaps the whole half-round of the other context, so the CU never idles
//   waiting for the host.  Workgroups do not wait for each other; a context whose query is finished starts its next one in its
//   next half-round.
#define SRCH_GO_STOP 0xFFFFFFFFu
#define SRCH_FIN 0xFFFFFFFFu        // h_done value: this context group of the workgroup has no queries left
#define SRCH_CTX_WORDS 16u          // parked per-context state of a wave, in LDS

// MAXT: threads per workgroup the instance is compiled for -- 1024 (16 waves, 128 VGPRs each) or, for the instances of the long code
// rows (>= 64 chunks), 768 (12 waves, 168 VGPRs each).  Build switch BANG_LONG_MAXT=1024: the self-paced long-row instances as 16-wave
// instances -- 120-127 VGPRs without scratch (LANE_FRESH below), 144 words of scratch per wave (the cooperative fetch hands its pieces over in
// rounds, 128-slot claim table), 16 x (2L + L/4 + 144) words beside the 128 KB pivot table up to L = 161.  Measured (round 6, docs/HISTORY.md):
// 16 waves run the 10 K SIFT1B-shape batch no faster than the 12 of the 168-VGPR build (8.31 vs 8.30 ms: the launch sits on the rate of requests
// past L2, not on waves in flight) and the 128-VGPR code loses 3-9 % where fewer waves are resident: 768 stays the default.
#ifndef BANG_LONG_MAXT
#define BANG_LONG_MAXT 768
#endif

__host__ __device__ constexpr int search_maxt(int ndw, bool host_paced) { return (search_coop(ndw, host_paced) && ndw >= 16) ? (host_paced ? 768 : BANG_LONG_MAXT) : 1024; }
// 256 words where 12 waves share the LDS beside the pivot table; the 16-wave instances hand the pieces of the cooperative fetch over in rounds (144)
__host__ __device__ constexpr uint32_t search_scratch_words(int ndw, bool host_paced) {
  return search_coop(ndw, host_paced) ? ((ndw >= 16 && search_maxt(ndw, host_paced) >= 1024) ? 144u : 256u) : 144u;
}

template <int PSZ, int NDW, bool ALIGNED, int NHI, bool HOST, bool SPEC>
__global__ __launch_bounds__(search_maxt(NDW, HOST)) void search_kernel(const SearchArgs a) {
  extern __shared__ __attribute__((aligned(16))) float lds[];
  const bang_search_params& p = a.p;
  float* piv_lds = lds;
  {
    // stage the chunk-packed pivot table (query independent) once per workgroup: 16 B per lane, 8 loads in flight per lane
    const float4* src = (const float4*)p.d_pivots_packed;
    float4* dst = (float4*)piv_lds;
    const uint32_t n4 = a.lds_piv_floats >> 2;
    for (uint32_t i0 = threadIdx.x; i0 < n4; i0 += blockDim.x * 8) {
      float4 v[8];
#pragma unroll
      for (int j = 0; j < 8; ++j) {
        const uint32_t i = i0 + (uint32_t)j * blockDim.x;
        v[j] = src[i < n4 ? i : n4 - 1];
      }
#pragma unroll
      for (int j = 0; j < 8; ++j) {
        const uint32_t i = i0 + (uint32_t)j * blockDim.x;
        if (i < n4) dst[i] = v[j];
      }
    }
  }
  const int lane0 = lane_id();
  int lane = lane0;                                // (re-derived per phase inside the loop: LANE_FRESH)
  const uint32_t wave = uni(threadIdx.x >> 6);
  const uint32_t nwaves = blockDim.x >> 6;
  // the iteration's arguments, one dword per lane (IterArgs)
  uint32_t iav;
  {
    const char __attribute__((address_space(4)))* ka = (const char __attribute__((address_space(4)))*)__builtin_amdgcn_kernarg_segment_ptr();
    iav = ((const uint32_t __attribute__((address_space(4)))*)(ka + offsetof(SearchArgs, iter)))[lane0 < IA_DWORDS ? lane0 : 0];
  }
#define IA_FRESH() asm volatile("" : "+v"(iav))
#define IA32(k) ((uint32_t)__builtin_amdgcn_readlane((int)iav, (k)))
#define IA64(k) (((uint64_t)IA32((k) + 1) << 32) | (uint64_t)IA32(k))
#define IAPTR(type, k) ((type GAS*)IA64(k))        // (global address space: bang_device.h GAS)
  const uint32_t nctx = HOST ? a.nctx : 1u;
  const uint32_t L = p.L, medoid = p.medoid;
  const unsigned long long go_timeout = p.go_timeout_ticks ? p.go_timeout_ticks : BANG_KERNEL_GO_TIMEOUT_TICKS;   // 100 MHz ticks
  // a wave's LDS region: [worklist of context 0]([worklist of context 1])[scratch 144]([parked context state 2 x 16]: nctx == 2 only)
  uint32_t* wbase = (uint32_t*)(lds + a.lds_piv_floats) + (size_t)wave * a.wave_words;
  uint32_t* scratch = wbase + (size_t)nctx * a.wl_words;
  uint32_t* park = scratch + search_scratch_words(NDW, HOST);
  // pacing group of this wave
  const uint32_t gs = HOST ? a.gs : nwaves;
  const uint32_t grp_in_wg = HOST ? wave / gs : 0u;
  const uint32_t gw0 = grp_in_wg * gs;                                   // first wave of the group
  const uint32_t gsize = (nwaves - gw0) < gs ? (nwaves - gw0) : gs;      // waves in it
  const uint32_t gslot = wave - gw0;
  const uint32_t ngrp = HOST ? (nwaves + gs - 1) / gs : 1u;
  // group-shared words behind the waves' regions (HOST only), 128 per group: [0] go value seen, [1..3] active-context counters (rotating
  // per executed half-round); per context c (a context's FIRST round has no barrier in front of it, so the two contexts must not
  // share a publish area): [4 + 48c ..] parents x 16, [20 + 48c ..] query | row wanted << 31, [36 + 48c ..] candidate index
  // [120..121] the group's barrier
  uint32_t* wg_lds = (uint32_t*)(lds + a.lds_piv_floats) + (size_t)nwaves * a.wave_words + (size_t)grp_in_wg * 128;
  if (HOST && gslot == 0) { wg_lds[lane] = 0u; wg_lds[64 + lane] = 0u; }
  if (HOST && nctx == 2 && lane < (int)(2 * SRCH_CTX_WORDS)) park[lane] = 0u;        // both contexts: inactive
  __syncthreads();
  if (p.d_ktime && threadIdx.x == 0) p.d_ktime[2 * blockIdx.x] = __builtin_amdgcn_s_memrealtime();

  float* sd = (float*)scratch;
  uint32_t* ti = scratch;
  float* td = (float*)(scratch + 72);
  uint32_t* sc = scratch + 72;                     // compaction scratch (== td: dead before the sort)
  uint32_t* tbl = scratch;                         // filter claim table, 128 words (== sd + td: dead between the stages that use them)
  const uint32_t total_waves = gridDim.x * nwaves;
  const uint32_t gw = blockIdx.x * nwaves + wave;
  const uint32_t cand_stride = L + BANG_EXTRA_ITERS;
  constexpr int SB = (NDW >= 18) ? 6 : 0;          // long rows (70 .. 128 chunks): consumed 6 code dwords (24 chunks) at a time
  constexpr bool EARLY_ROWS = !HOST;                           // code rows requested before the filter update (host-paced instances: behind it --
                                                               // 60-100 B of scratch per lane otherwise)
  constexpr bool COOP = search_coop(NDW, HOST);                // ... by P adjacent lanes per row, one 16-byte piece each
  constexpr uint32_t SCR = search_scratch_words(NDW, HOST);    // words of per-wave scratch: 144, or 256 (12-wave instances)
  // SPEC (instances of the long-row layouts, chosen per launch: spec_rows): the code rows of ALL ids of the adjacency row are requested
  // together with their filter probes, one memory latency earlier; the distances of the ids the filter then drops are computed and thrown
  // away (every lane runs the reduce anyway), the survivors' are compacted behind it.  Same distances for the same ids: same results.
  static_assert(!SPEC || !HOST, "the speculative row request belongs to the self-paced form");

  const uint32_t code_stride = p.code_stride ? p.code_stride : p.m;

  // ---- state of the context this wave is working on (registers; parked in LDS between half-rounds when there are two)
  bool active = false, exhausted = false;
  uint32_t q = 0, iter = 0, w_n = 0, cc = 0, mark = 0, evals = 0, fetched = 0;
  uint32_t cnt_in = 0, x0 = 0, x1 = 0;
  bool have_row = false;
  bool self_row = false;                           // HOST: the parent's row sits in this GPU's HBM copy of the first rows -- the wave has asked for it itself,
                                                   // the walker was told there is nothing to fetch (one context per wave only)
  WlHead head;                                     // first unvisited worklist entry + last distance, as of the last merge (uniform)
  head.found = false; head.idx = 0; head.id = 0; head.d = 0.0f; head.tail = 0.0f;
  // the centred query of the current context, in registers (lane l of qc.v[r] = element 64 r + l; read with v_readlane): loaded
  // once per query instead of streamed through scalar loads in every iteration's distance stage
  constexpr int QW = NDW * 4 * PSZ;
  // the self-paced long-row instances (168 VGPRs) keep the query replicated per 16-lane row: "pivot - query" is then one DPP instruction
  constexpr bool QC16 = !HOST && ((NDW == 18 && NHI != 0) || NDW == 19);   // (the BASELINE long-row layouts: the instances whose register budget holds it, profiles/r05_kernel_usage.md)
  constexpr int NV = QC16 ? (QW + 15) / 16 : (QW + 63) / 64;
  typedef typename std::conditional<QC16, QcRow16<NV>, QcRegs<NV>>::type Qc;
  Qc qc;
#pragma unroll
  for (int r = 0; r < NV; ++r) qc.v[r] = 0.0f;
  auto load_qc = [&](uint32_t qq) {
    const float* src = p.d_qc + (size_t)qq * QW;
#pragma unroll
    for (int r = 0; r < NV; ++r) {
      const uint32_t i = QC16 ? (uint32_t)r * 16u + ((uint32_t)lane & 15u) : (uint32_t)r * 64u + (uint32_t)lane;
      qc.v[r] = src[i < (uint32_t)QW ? i : 0u];
    }
  };
  // which words of the current query's filter have been stored to (self-paced form; the host-paced instances have no registers to spare)
  constexpr bool SUMM = (BANG_FILTER_SUMMARY != 0) && (!HOST || search_maxt(NDW, HOST) < 1024);   // (needs 6 VGPRs the 16-wave host-paced instances do not have)
  FilterSummary<6> summ;
  summ.clear();
  uint32_t probes_skipped = 0;                     // diagnostic counter (d_qskip): filter words not loaded thanks to the summary
  uint32_t started = 0;                            // bit c: context c has taken its first (statically assigned) query
  uint32_t dead_mask = 0;                          // HOST: bit c: context c of this WORKGROUP has no queries left (uniform across the workgroup)
  uint32_t rounds0 = 0, rounds1 = 0;               // HOST: rounds completed by context 0 / 1
  uint32_t tick = 0;                               // HOST: half-rounds this workgroup has EXECUTED (a finished context's are skipped)

  // diagnostic (p.d_prof != NULL, host-paced form): thread 0 of every workgroup accumulates where its half-rounds spend their time
  unsigned long long pf_poll = 0, pf_front = 0, pf_pub = 0, pf_back = 0, pf_n = 0, pf_t = 0;
  const bool prof = HOST && p.d_prof != nullptr && wave == 0;             // uniform (the accumulators stay in scalar registers): first wave of the workgroup's first group
#define PF_STAMP(acc) do { if (prof) { const unsigned long long t_ = __builtin_amdgcn_s_memrealtime(); acc += t_ - pf_t; pf_t = t_; } } while (0)
  if (prof) pf_t = __builtin_amdgcn_s_memrealtime();

#ifdef BANG_SEARCH_PHASE_PROF
  // diagnostic build (make CXXFLAGS+=-DBANG_SEARCH_PHASE_PROF): wave 0 of every workgroup stamps the phase boundaries of its
  // iterations (no draining: a phase ends where the compiler had to wait for its results anyway) into p.d_prof[wg][8 + k]
  unsigned long long ph_acc[12] = {0, 0, 0, 0, 0, 0, 0, 0, 0, 0, 0, 0}, ph_t = __builtin_amdgcn_s_memrealtime(), ph_n = 0;
#define PH(k) do { if (wave == 0) { const unsigned long long t_ = __builtin_amdgcn_s_memrealtime(); ph_acc[k] += t_ - ph_t; ph_t = t_; } } while (0)
#else
#define PH(k) do {} while (0)
#endif

  // The lane number passes through an empty asm at the top of an iteration and in front of its register-hungry phases: whatever is derived from it
  // (LDS addresses of the scratch areas, the worklist, the claim table ...) is re-derived there with one or two VALU instructions instead of living in a
  // register across the whole loop -- hoisted out of the loop these invariants cost the 128-VGPR instances ~20 registers they do not have.
#define LANE_FRESH() do { if (BANG_LONG_MAXT >= 1024) { lane = lane0; asm volatile("" : "+v"(lane)); } } while (0)
  for (uint32_t half = 0;; ++half) {
    LANE_FRESH();
    const uint32_t c = HOST ? (nctx == 2 ? (half & 1u) : 0u) : 0u;
    if (HOST) {
      if (dead_mask == (nctx == 2 ? 3u : 1u)) break;
      if ((dead_mask >> c) & 1u) continue;
    }
    const uint32_t round = HOST ? (c ? rounds1 : rounds0) + 1u : 0u;
    const uint32_t grp = HOST ? (blockIdx.x * ngrp + grp_in_wg) * nctx + c : 0u;   // pacing group of (workgroup, wave group, context)
    WaveLds s;
    s.wd = (float*)(wbase + (size_t)c * a.wl_words); s.wi = wbase + (size_t)c * a.wl_words + L; s.wv = (uint8_t*)(wbase + (size_t)c * a.wl_words + 2 * L);
    s.sd = sd; s.ti = ti; s.td = td;
    if (HOST) {
      if (nctx == 2) {                                                  // un-park context c
        const uint32_t* pk = park + c * SRCH_CTX_WORDS;
        active = uni(pk[0]) != 0u; q = uni(pk[1]); iter = uni(pk[2]); w_n = uni(pk[3]); cc = uni(pk[4]); mark = uni(pk[5]);
        evals = uni(pk[6]); fetched = uni(pk[7]);
        have_row = uni(pk[8]) != 0u;
        head.found = uni(pk[9]) != 0u; head.idx = uni(pk[10]); head.id = uni(pk[11]);
        head.d = __uint_as_float(uni(pk[12])); head.tail = __uint_as_float(uni(pk[13]));
        if (active) load_qc(q);
      }
      // ---------------- wait for the rows of this context's previous round
      if (round > 1) {
        if (gslot == 0 && lane == 0) {
          const uint32_t* go = p.d_ctl + (size_t)grp * 16;
          uint32_t v = __hip_atomic_load(go, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_SYSTEM);
          const unsigned long long t0 = __builtin_amdgcn_s_memrealtime();
          while (v < round - 1u) {                           // SRCH_GO_STOP is the largest value: it also ends the wait
            __builtin_amdgcn_s_sleep(4);
            v = __hip_atomic_load(go, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_SYSTEM);
            if (__builtin_amdgcn_s_memrealtime() - t0 > go_timeout) {                     // the host is gone (it gives up first)
              v = SRCH_GO_STOP;
              if (p.d_abort) *p.d_abort = 1u;
              break;
            }
          }
          wg_lds[0] = v;
          // rows the CPU wrote through the BAR since the last round must not be served from this CU's L1 (they live in
          // fine-grained LOCAL memory: L2 copies are invalidated by the fabric when the PCIe writes land)
          __builtin_amdgcn_fence(__ATOMIC_ACQUIRE, "agent");
          asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
        }
        group_barrier(wg_lds + 120, gsize, lane);
        PF_STAMP(pf_poll);
        if (wg_lds[0] == SRCH_GO_STOP) break;
        if (active && have_row && !self_row) {
          // control line of the group: {go, count bytes x 16, ...}; rows: 64 ids each, 256-byte aligned
          const uint32_t cw = p.d_ctl[(size_t)grp * 16 + 1 + (gslot >> 2)];
          cnt_in = (cw >> (8 * (gslot & 3u))) & 0xFFu;
          x0 = p.d_rows[((size_t)grp * 16 + gslot) * 64 + lane];
        }
      }
    }

    // ---------------- a finished context takes its next query: the first one by position, then from the hand-out counter
    if (!active && !exhausted) {
      if (!((started >> c) & 1u)) q = c * total_waves + gw;
      else {
        uint32_t t = 0;
        if (lane == 0) t = __hip_atomic_fetch_add(KARGP(uint32_t, d_next_query), 1u, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
        q = nctx * total_waves + uni(t);
      }
      started |= 1u << c;
      if (q < KARG(Q)) {
        active = true;
        w_n = 0; cc = 1; mark = 0x01010101u;           // cudaMemset(d_mark, 1, ...) :446 ; candidate log = [MEDOID] :452-464
        evals = 0; fetched = 0; iter = 1;
        if (SUMM) { summ.clear(); probes_skipped = 0; }
        if (lane == 0) KARGP(uint32_t, d_cand_ids)[(size_t)q * cand_stride] = medoid;
        load_qc(q);
        // the seed list [MEDOID, adj(MEDOID)...] (bang_init :467-489): {count, id x 65}
        { const uint32_t GAS* seed = KARGP(const uint32_t, d_seed); cnt_in = seed[0]; x0 = seed[1 + lane]; x1 = seed[65]; }
        have_row = true;
        self_row = false;
      } else exhausted = true;
    }
    if (!HOST && !active) break;

    // results of the front half, consumed by the back half below
    uint32_t n = 0, sid0 = 0, sid1 = 0, parent = 0;
    // self-paced form: the summary's transposition passes (set) run at the END of the iteration, while the wave would otherwise idle waiting
    // for the next adjacency row (nothing reads the summary in between)
    constexpr bool SET_LATE = !HOST;
    bool sl_a = false, sl_b = false;                  // summary marks of this iteration's survivors, applied behind the merge
    uint32_t sl_ua = 0, sl_ub = 0;
    float d0 = BIG_DIST, d1 = BIG_DIST;
    bool found = false;
    const bool first = (iter == 1);
    if (active) {
      // ---------------- K5: filter (neighbor_filtering_new :1140-1165) ----------------
      uint32_t ci = have_row ? uni(cnt_in) : 0u;
      if (((!HOST && p.row_layout) || (HOST && self_row)) && !first && have_row)       // a 256-byte adjacency row: ids ascending, padding behind them
        ci = (uint32_t)__popcll(__ballot(x0 != 0xFFFFFFFFu));
      PH(0);   // the adjacency row has arrived (and: query hand-out, loop overhead)
      {
        const uint32_t cap = p.R + (first ? 1u : 0u);
        if (ci > cap) ci = cap;
      }
      // a row that names a node the index does not have (rows overwritten behind the engine's back: the self-paced form has no host
      // thread that could notice) is not followed: the batch ends with BANG_ERR_HIP instead of a wild read of the code table
      IA_FRESH();
      // Wave priority (launch policy `prio`, bang_k_search): the two stretches of an iteration that END in memory requests -- hashes -> probes -> code-row
      // requests here, parent selection -> row request below -- run at raised priority, the reduce and the merge at the default: a wave that is about to
      // put requests in flight is not held up by its neighbours' arithmetic.  Pays where wave slots are free (2 500-query shard 2.24 -> 2.20 ms, 1 250:
      // 1.46 -> 1.44), costs a full chip 2 % (10 K batch 7.92 -> 8.12): on for launches of at most 10 queries per CU.
      const uint32_t prio_m = HOST ? 0u : IA32(IA_PRIO);     // bit 0: this stretch, bit 1: the hand-over's
      const bool prio = (prio_m & 1u) != 0u;
      if (prio) __builtin_amdgcn_s_setprio(3);
      uint32_t GAS* bloom = IAPTR(uint32_t, IA_BLOOM) + (size_t)q * BANG_BF_WORDS;
      const uint32_t n_nodes = HOST ? 0u : IA32(IA_N_NODES);
      if (!HOST && n_nodes != 0u) {
        if (__ballot((uint32_t)lane < ci && x0 >= n_nodes) != 0ull) {
          if (lane == 0 && KARG(d_abort)) *KARGP(uint32_t, d_abort) = 2u;
          ci = 0;
        }
      }
      fetched += ci;
      const bool v0 = (uint32_t)lane < ci;
      const bool v1 = ci > 64;                               // the 65th id exists in the seed list only (uniform)
      const uint32_t h0a = hash1(x0), h0b = hash2(x0);
      uint32_t h1a = 0, h1b = 0, w0a = 0, w0b = 0, w1a = 0, w1b = 0;
      // CANON: every id is tested against the filter state at entry (all loads before any store); both words in one round trip.
      // A word the summary knows to be untouched is zero: no request (FilterSummary).
      bool la = v0, lb = v0;                                 // load word a / b?
      const bool summ_on = SUMM && (HOST || iter <= IA32(IA_SUMM_ITERS));     // (uniform; one-way per query: once off, the registers go stale.  bang_k_search resolves 0 = auto)
      if (SUMM && summ_on) {
        la = summ.test(h0a >> 5) && v0;
        lb = summ.test(h0b >> 5) && v0;
        probes_skipped += (uint32_t)__popcll(__ballot(v0 && !la)) + (uint32_t)__popcll(__ballot(v0 && !lb));
      }
      if (la) w0a = ld_bypass_l1(&bloom[h0a >> 5]);
      if (lb) w0b = ld_bypass_l1(&bloom[h0b >> 5]);
      if (v1) {
        h1a = hash1(x1); h1b = hash2(x1);
        if (lane == 0) { w1a = ld_bypass_l1(&bloom[h1a >> 5]); w1b = ld_bypass_l1(&bloom[h1b >> 5]); }
      }
      PqRow<NDW, ALIGNED> row;
      CoopFetch<NDW, ALIGNED> cf;
      const uint8_t GAS* d_codes = IAPTR(const uint8_t, IA_CODES);
      // SPEC: the code rows of every id of the row, behind the probes just issued (older loads are waited for first: the probes' answers
      // are used while the rows still travel)
      if (SPEC && COOP) cf.issue(d_codes, code_stride, x0, ci < 64u ? ci : 64u, lane);
      else if (SPEC && v0) pq_row_load(row, d_codes, code_stride, x0);
      // The row loads stay HERE, in front of the wait for the probes: left to itself the scheduler -- this instance sits at its VGPR limit -- may sink them
      // behind that wait to shorten the 24 piece registers' lives, and the early request is gone without a trace but the time (one more live register
      // was enough: 1 250-query shard 1.51 -> 1.63 ms, ISA checked).  A compiler-level memory barrier: loads do not move across it.
      if (SPEC) asm volatile("" ::: "memory");
      if (prio) __builtin_amdgcn_s_setprio(0);
      const bool pass0 = v0 && !(((w0a >> (h0a & 31)) & 1u) && ((w0b >> (h0b & 31)) & 1u));
      const bool pass1 = v1 && (lane == 0) && !(((w1a >> (h1a & 31)) & 1u) && ((w1b >> (h1b & 31)) & 1u));
      const uint64_t m0 = __ballot(pass0);
      const uint64_t m1 = __ballot(pass1);
      const uint32_t n0 = (uint32_t)__popcll(m0);
      n = n0 + (uint32_t)__popcll(m1);
      PH(1);   // hashes + filter probes returned
      // ordered compaction through LDS: survivors keep input order (CANON; the reference emits in atomicAdd order :1161)
      if (pass0) sc[lanes_below(m0)] = x0;
      if (pass1) sc[n0] = x1;
      wave_sync();
      if ((uint32_t)lane < n) sid0 = sc[lane];
      if (lane == 0 && n > 64) sid1 = sc[64];
      wave_sync();
      evals += n;
      PH(2);   // compaction

      // the survivors' PQ code rows are requested NOW (unless they already travel: SPEC): under the filter update below, which runs on LDS
      if (!SPEC && EARLY_ROWS) {
        if (COOP) cf.issue(d_codes, code_stride, sid0, n < 64u ? n : 64u, lane);
        else if ((uint32_t)lane < n) pq_row_load(row, d_codes, code_stride, sid0);
      }

      // ---------------- K5, second half: set the slots of the survivors (:1159-1160) ----------------
      // (before the distance arithmetic: the hashes and the probed words die here instead of living through the register-hungry K2)
      // The claim rounds run now, on LDS, while the code rows travel; the stores they decide on are issued once the rows are here.
      bool pa = pass0, pb = pass0, st_a = false, st_b = false;
      uint32_t sv_a = 0, sv_b = 0;
      filter_commit<SCR >= 256 ? 256 : 128>(tbl, lane, pa, h0a >> 5, 1u << (h0a & 31), w0a, pb, h0b >> 5, 1u << (h0b & 31), w0b, st_a, sv_a, st_b, sv_b);
      // the words about to be stored to are no longer zero (only those the summary did not know yet need marking)
      if (SUMM && summ_on) {
        if (SET_LATE && !first) { sl_a = pass0 && !la; sl_b = pass0 && !lb; sl_ua = h0a >> 5; sl_ub = h0b >> 5; }
        else summ.template set<SCR >= 256 ? 4 : 2>(tbl, lane, pass0 && !la, h0a >> 5, pass0 && !lb, h0b >> 5, pass1, h1a >> 5, h1b >> 5);
      }
      auto filter_stores = [&]() {
        asm volatile("" ::: "memory");
        if (st_a) bloom[h0a >> 5] = sv_a;
        if (st_b) bloom[h0b >> 5] = sv_b;
        const uint64_t left = __ballot(pa || pb || pass1);
        if (left) {                                            // rare: lost three claim rounds; or the 65th id of the seed list
          asm volatile("s_waitcnt vmcnt(0)" ::: "memory");    // behind the plain stores (which were computed from the old words)
          if (pa) (void)__hip_atomic_fetch_or(&bloom[h0a >> 5], 1u << (h0a & 31), __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
          if (pb) (void)__hip_atomic_fetch_or(&bloom[h0b >> 5], 1u << (h0b & 31), __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
          if (pass1) {
            (void)__hip_atomic_fetch_or(&bloom[h1a >> 5], 1u << (h1a & 31), __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
            (void)__hip_atomic_fetch_or(&bloom[h1b >> 5], 1u << (h1b & 31), __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
          }
        }
      };
      if (!EARLY_ROWS) filter_stores();

      PH(3);   // filter update (claim table + stores issued)
      // ---------------- K2: PQ distances (compute_neighborDist_par :1201-1241) ----------------
      LANE_FRESH();
      {
        if (COOP && !EARLY_ROWS) cf.issue(d_codes, code_stride, sid0, n < 64u ? n : 64u, lane);
        if (COOP) cf.template collect<(int)SCR>(row, scratch, code_stride, SPEC ? x0 : sid0, lane);     // (all lanes: the pieces change hands through LDS)
        if (EARLY_ROWS) {
          // the rows have arrived (per-lane loads: every row register passes through an empty asm, which is where the compiler waits
          // for them): now the filter stores -- their acknowledgements are not waited for until the next row is needed
          if (!COOP) {
#pragma unroll
            for (int i = 0; i < PqRow<NDW, ALIGNED>::NX4 * 4; ++i) asm volatile("" : "+v"(row.w[i]));
          }
          filter_stores();
        }
        if (QC16) {
          // every lane executes the reduce (the query operand of its subtractions comes from other lanes by DPP); a lane without a survivor
          // reduces whatever row the cooperative fetch left it (row 0) and its distance is never looked at
          const float dd = pq_row_reduce_pipe<PSZ, NDW, ALIGNED, NHI>(row, piv_lds, qc);
          if (SPEC) {                                          // lane i evaluated id i of the row: the survivors' distances move up, in input order
            if ((m0 >> lane) & 1ull) sc[lanes_below(m0)] = __float_as_uint(dd);
            wave_sync();
            if ((uint32_t)lane < n0) d0 = __uint_as_float(sc[lane]);
            wave_sync();
          } else if ((uint32_t)lane < n) d0 = dd;
          if (n > 64) {                                        // survivor 64 (seed list only): every lane reduces that row, lane 0's counts
            PqRow<NDW, ALIGNED> r1;
            pq_row_load(r1, d_codes, code_stride, uni(sid1));
            const float d1v = pq_row_reduce<PSZ, NDW, ALIGNED, NHI, SB>(r1, piv_lds, qc);
            if (lane == 0) d1 = d1v;
          }
        } else {
          if (SPEC) {
            float dd = BIG_DIST;
            if (v0) dd = pq_row_reduce_pipe<PSZ, NDW, ALIGNED, NHI>(row, piv_lds, qc);
            if ((m0 >> lane) & 1ull) sc[lanes_below(m0)] = __float_as_uint(dd);
            wave_sync();
            if ((uint32_t)lane < n0) d0 = __uint_as_float(sc[lane]);
            wave_sync();
          } else if ((uint32_t)lane < n) {
            if (!COOP && !EARLY_ROWS) pq_row_load(row, d_codes, code_stride, sid0);
            d0 = !HOST ? pq_row_reduce_pipe<PSZ, NDW, ALIGNED, NHI>(row, piv_lds, qc)     // (host-paced instances: 12-24 B of scratch with it)
                       : pq_row_reduce<PSZ, NDW, ALIGNED, NHI, SB>(row, piv_lds, qc);
          }
          if (n > 64) {                                          // survivor 64 (seed list only), lane 0
            if (lane == 0) {
              PqRow<NDW, ALIGNED> r1;
              pq_row_load(r1, d_codes, code_stride, sid1);
              d1 = pq_row_reduce<PSZ, NDW, ALIGNED, NHI, SB>(r1, piv_lds, qc);
            }
          }
        }
      }

      // ---------------- K4: parent (compute_parent1 :1464-1521 / compute_parent2 :1384-1459) ----------------
      if (prio_m & 2u) __builtin_amdgcn_s_setprio(3);         // parent selection -> row request
      LANE_FRESH();
      // closest new neighbour: strict '<', first minimum wins, MEDOID skipped (:1413-1418)
      const bool elig = (uint32_t)lane < n && sid0 != medoid && d0 < BIG_DIST;
      PH(4);   // code rows returned + distances
      // (PQ distances are sums of squares: non-negative floats order like their bit patterns, so {distance bits, lane} is a key
      // whose minimum is "smallest distance, first lane wins")
      uint32_t khi = elig ? __float_as_uint(d0) : 0xFFFFFFFFu, klo = (uint32_t)lane;
      wave_min_key(khi, klo);
      uint32_t bi = (khi != 0xFFFFFFFFu) ? klo : 0xFFFFu;
      float bd = (khi != 0xFFFFFFFFu) ? __uint_as_float(khi) : BIG_DIST;
      uint32_t bid = (uint32_t)__builtin_amdgcn_readlane((int)sid0, (int)(klo & 63u));
      if (n > 64) {                                            // element 64 can only win with a strictly smaller distance
        const float e_d = __shfl(d1, 0);
        const uint32_t e_id = (uint32_t)__shfl((int)sid1, 0);
        if (e_id != medoid && e_d < BIG_DIST && (bi == 0xFFFFu || e_d < bd)) { bd = e_d; bi = 64; bid = e_id; }
      }
      const bool have_best = (bi != 0xFFFFu);
      if (!have_best) bd = BIG_DIST;
      bool from_best = false;
      if (first) {
        if (have_best) { found = true; parent = bid; from_best = true; }
      } else {
        if (head.found) {                                      // first unvisited entry :1425-1439 (worklist_head() after the last merge)
          found = true;
          if (bd < head.d) { parent = bid; from_best = true; }
          else { parent = head.id; if (lane == 0) s.wv[head.idx] = 1; }
        } else if (w_n > 0) {                                  // corner case :1442-1446
          if (bd < head.tail) { found = true; parent = bid; from_best = true; }
        }
      }
      parent = uni(parent);
      if (found) {
        if (from_best) mark = parent;
        ++cc;                                                  // (the candidate-log store :1451-1458 is issued behind the row request: it is not on the chain)
      }
    }
    IA_FRESH();                                                 // (the hand-over's arguments: IterArgs)
    const uint32_t cap_iter = IA32(IA_CAP_ITER);
    const bool want_row = active && found && iter < cap_iter;
    PH(5);     // parent selection

    // ---------------- hand the parent over
    if (!HOST) {
      // graph resident in HBM: the next adjacency row is requested NOW, straight into the registers the next iteration reads (no copy at the
      // loop's end that would wait for it); it travels while the survivors are merged
      if (want_row) {
        const uint8_t GAS* gbase = IAPTR(const uint8_t, IA_GRAPH);
        if (IA32(IA_ROW_LAYOUT)) {                                // adjacency rows (pinned host memory, pull mode): 64 ids, padded
          // the rows of the first n_rows_hbm nodes also sit in HBM (whatever HBM the index left over): no PCIe read for those.  Peer rows
          // (n_slices > 1): slice parent / slice_rows of the node's HBM-resident rows -- this GPU's HBM or a peer's over xGMI; the table
          // holds biased base addresses (0: that slice is not there), read through the scalar cache
          const uint32_t GAS* hb = nullptr;
          const uint32_t nsl = IA32(IA_N_SLICES);
          if (nsl > 1u) {
            const uint32_t sl = parent / IA32(IA_SLICE_ROWS);                      // (uniform: scalar)
            if (sl < nsl) hb = (const uint32_t GAS*)scalar_load_u64(IA64(IA_ROW_SLICES), sl);
          } else if (parent < IA32(IA_N_ROWS_HBM)) hb = IAPTR(const uint32_t, IA_ROWS_HBM);
          if (hb) x0 = hb[(uint64_t)parent * 64u + lane];        // (plain: with the non-temporal hint on the rows read from HBM the 10 K batch took 8.22 instead of 7.96 ms)
          else x0 = __builtin_nontemporal_load((const uint32_t GAS*)gbase + (uint64_t)parent * 64u + lane);
          cnt_in = 64u;                                      // counted when the row is consumed
        } else {
          const uint32_t GAS* nrow = (const uint32_t GAS*)(gbase + (uint64_t)parent * IA64(IA_ENTRY_LEN) + IA32(IA_VEC_BYTES));
          cnt_in = nrow[0];
          x0 = nrow[1 + lane];                               // in bounds: an entry holds R = 64 id slots (+ slack behind the graph)
        }
      }
    } else {
      uint32_t* cnt_act = wg_lds + 1 + (tick % 3u);
      // a parent whose row this GPU holds itself (the HBM copy of the first rows: walker-from-rows form) is not the walker's business: the wave
      // asks for the row now -- it travels during the sort/merge, as in the self-paced form -- and publishes "nothing to fetch"
      self_row = false;
      if (nctx == 1u && want_row && !p.ship_vectors) {
        if (parent < IA32(IA_N_ROWS_HBM)) { self_row = true; x0 = (IAPTR(const uint32_t, IA_ROWS_HBM))[(uint64_t)parent * 64u + lane]; cnt_in = 64u; }
      }
      if (lane == 0) {
        // parents travel to the host in one coalesced store per workgroup.  A parent whose vector the walker must ship (vectors
        // not resident) is published even when no row is needed any more (the one chosen at the iteration cap, CANON 6).
        const bool tell = active && found && (want_row || p.ship_vectors);
        wg_lds[4 + 48 * c + gslot] = tell ? (self_row ? BANG_IDLE_PARENT : parent) : BANG_NO_PARENT;
        wg_lds[20 + 48 * c + gslot] = q | (want_row ? 0x80000000u : 0u);
        wg_lds[36 + 48 * c + gslot] = cc - 1u;
        if (active) (void)__hip_atomic_fetch_add(cnt_act, 1u, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_WORKGROUP);
      }
      group_barrier(wg_lds + 120, gsize, lane);
      PF_STAMP(pf_front);
      const bool grp_alive = (*cnt_act != 0u);               // none active => every wave found the hand-out counter exhausted: final
      if (gslot == 0) {
        if (grp_alive) {
          if ((uint32_t)lane < gsize) {
            __hip_atomic_store(&p.h_parents[(size_t)grp * 16 + lane], wg_lds[4 + 48 * c + lane], __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_SYSTEM);
            if (p.ship_vectors) {
              __hip_atomic_store(&p.h_pub_q[(size_t)grp * 16 + lane], wg_lds[20 + 48 * c + lane], __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_SYSTEM);
              __hip_atomic_store(&p.h_pub_c[(size_t)grp * 16 + lane], wg_lds[36 + 48 * c + lane], __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_SYSTEM);
            }
          }
          asm volatile("s_waitcnt vmcnt(0)" ::: "memory");    // the flag must not overtake the parents (MI355X guide)
        }
        if (lane == 0) {
          __hip_atomic_store(p.h_done + (size_t)grp * 16, grp_alive ? round : SRCH_FIN, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_SYSTEM);
          // three counters in rotation: the one reset here was last read before this half-round's barrier and is next
          // incremented two executed half-rounds from now, i.e. behind the next half-round's barrier (a context's FIRST round has
          // no wait-for-rows barrier in front, so the counter of the very next half-round may already be in use)
          wg_lds[1 + ((tick + 2u) % 3u)] = 0u;
        }
      }
      ++tick;
      PF_STAMP(pf_pub);
      if (prof) ++pf_n;
      if (!grp_alive) { dead_mask |= 1u << c; continue; }
      if (c) rounds1 = round; else rounds0 = round;
    }

    PH(8);     // hand-over: the next row is requested
    if (!HOST && (IA32(IA_PRIO) & 2u) != 0u) __builtin_amdgcn_s_setprio(0);
    if (active && found && lane == 0) IAPTR(uint32_t, IA_CAND_IDS)[(size_t)q * cand_stride + cc - 1u] = parent;      // :1451-1458 (cc counts it already)
    if (active) {
      // ---------------- K3a + K3b: sort the survivors, merge them into the worklist ----------------
      LANE_FRESH();
      if (n > 0 && iter < cap_iter) {
        w_n = sort_and_merge(s, n, d0, sid0, d1, sid1, iter, w_n, L, medoid, mark, head.tail, lane);
      }
      PH(6);   // (publish +) sort/merge
#ifdef BANG_SEARCH_PHASE_PROF
      if (wave == 0) ++ph_n;
#endif

      // ---------------- next iteration of this query, or the query is finished
      // a query is active while it has a parent or unmerged survivors (CANON 4); the loop ends at the cap (:950-956)
      if ((!found && n == 0) || iter == cap_iter) {
        if (lane == 0) {
          // (the once-per-query arguments are read where they are used: KARG)
          KARGP(uint32_t, d_cand_cnt)[q] = cc;
          uint32_t GAS* qstats = KARGP(uint32_t, d_qstats);
          uint32_t GAS* qiters = KARGP(uint32_t, d_qiters);
          uint32_t GAS* qskip = KARGP(uint32_t, d_qskip);
          if (qstats) { qstats[(size_t)q * 2] = evals; qstats[(size_t)q * 2 + 1] = fetched; }
          if (qiters) qiters[q] = iter;
          if (qskip) qskip[q] = probes_skipped;
        }
        if (!HOST) {
          RerankArgs8 rr;                                                      // (read here, once per query: KARG)
          rr.queries = (const uint8_t GAS*)KARG(rr_queries);
          if (rr.queries) {                                                    // K6 + K7 on the spot (uniform)
            rr.vec_base = (const uint8_t GAS*)KARG(rr_vec_base); rr.vec_stride = KARG(rr_vec_stride); rr.ids_out = (uint64_t GAS*)KARG(rr_ids_out); rr.dists_out = (float GAS*)KARG(rr_dists_out);
            rr.D = KARG(rr_D); rr.k = KARG(rr_k); rr.q0 = KARG(rr_q0); rr.Q_total = KARG(rr_Q_total);
            rr.cand = KARGP(const uint32_t, d_cand_ids) + (size_t)q * cand_stride;
            const uint32_t nc = cc < cand_stride ? cc : cand_stride;
            const uint32_t rdt = KARG(rr_dtype);
            if (rdt == BANG_F32) wave_rerank_f32<4>(rr, q, nc, wbase, lane);
            else if (rdt == BANG_I8) wave_rerank8<true>(rr, q, nc, wbase, lane);
            else wave_rerank8<false>(rr, q, nc, wbase, lane);
          }
        }
        active = false;
      } else {
        ++iter;
        have_row = found;
        head = worklist_head(s, w_n, lane);
        PH(9);   // worklist head
        if (HOST && !self_row) { cnt_in = 0; x0 = 0; }                       // (the walker's rows are read at the top of the next round)
        // the words this iteration's survivors stored to are no longer zero: marked now, under the latency of the row just requested
        if (SUMM && SET_LATE) summ.template set<SCR >= 256 ? 4 : 2>(tbl, lane, sl_a, sl_ua, sl_b, sl_ub, false, 0u, 0u);
        PH(10);  // summary marks
      }
    }
    if (HOST && nctx == 2) {                                             // park context c
      uint32_t* pk = park + c * SRCH_CTX_WORDS;
      if (lane == 0) {
        pk[0] = active ? 1u : 0u; pk[1] = q; pk[2] = iter; pk[3] = w_n; pk[4] = cc; pk[5] = mark; pk[6] = evals; pk[7] = fetched;
        pk[8] = have_row ? 1u : 0u;
        pk[9] = head.found ? 1u : 0u; pk[10] = head.idx; pk[11] = head.id; pk[12] = __float_as_uint(head.d); pk[13] = __float_as_uint(head.tail);
      }
      wave_sync();
    }
    PF_STAMP(pf_back);
  }
  if (prof && lane == 0) {
    unsigned long long* o = p.d_prof + (size_t)blockIdx.x * 16;
    o[0] = pf_poll; o[1] = pf_front; o[2] = pf_pub; o[3] = pf_back; o[4] = pf_n;
  }
#undef PF_STAMP
#ifdef BANG_SEARCH_PHASE_PROF
  if (p.d_prof && wave == 0 && lane == 0) {
    unsigned long long* o = p.d_prof + (size_t)blockIdx.x * 16 + 8;
    for (int k = 0; k < 7; ++k) o[k] = ph_acc[k];
    o[7] = ph_n;
    if (!HOST) { unsigned long long* x = p.d_prof + (size_t)blockIdx.x * 16; x[0] = ph_acc[8]; x[1] = ph_acc[9]; x[2] = ph_acc[10]; }
  }
#endif
#undef PH
  if (KARG(d_ktime)) {
    __syncthreads();
    if (threadIdx.x == 0) KARG(d_ktime)[2 * blockIdx.x + 1] = __builtin_amdgcn_s_memrealtime();
  }
}

// ---------------------------------------------------------------------------------------------------------------------
// launcher
// ---------------------------------------------------------------------------------------------------------------------
template <int PSZ, int NDW, bool ALIGNED, int NHI, bool HOST, bool SPEC>
static int launch_inst(const SearchArgs& a, dim3 grid, dim3 block, size_t lds, hipStream_t st) {
  static bool attr_done[BANG_MAX_DEVICES] = {false};      // per kernel instance AND device
  const int dev = current_device();
  if (!attr_done[dev]) {
    HIP_TRY(hipFuncSetAttribute((const void*)search_kernel<PSZ, NDW, ALIGNED, NHI, HOST, SPEC>, hipFuncAttributeMaxDynamicSharedMemorySize,
                                160 * 1024));
    attr_done[dev] = true;
  }
  hipLaunchKernelGGL((search_kernel<PSZ, NDW, ALIGNED, NHI, HOST, SPEC>), grid, block, lds, st, a);
  HIP_TRY(hipGetLastError());
  return BANG_OK;
}

// The instances live in TWO translation units of this one file (Makefile): part 0 (BANG_SEARCH_PART = 0: bang_search.o, built with
// -mllvm -amdgpu-sched-strategy=iterative-ilp -- round 6: every BASELINE layout 1-5 % faster under it, a 1 250-query SIFT1B-shape shard 1.52 -> 1.45 ms) and part 1
// (bang_search_b.o, the default scheduler): the instances the ILP strategy costs scratch -- rows of 96+ chunks, 74-chunk rows that are not dword-aligned
// (0 -> 72-116 B per lane), and the host-paced instances of the long-row layouts (20-68 -> 116-168 B).  Both units compile the same dispatch; each instantiates only its own instances.
#ifndef BANG_SEARCH_PART
#define BANG_SEARCH_PART 0
#endif
constexpr bool search_in_part1(int ndw, bool aligned, bool host_paced) { return ndw >= 24 || (ndw == 19 && !aligned) || (host_paced && ndw >= 16); }
extern "C" int bang_search_launch_part1(const SearchArgs* a, uint32_t grid, uint32_t block, size_t lds, void* stream);

template <int PSZ, int NDW, bool ALIGNED, int NHI, bool HOST, bool SPEC>
static int launch_part(const SearchArgs& a, dim3 grid, dim3 block, size_t lds, hipStream_t st) {
  if constexpr (search_in_part1(NDW, ALIGNED, HOST) == (BANG_SEARCH_PART == 1)) return launch_inst<PSZ, NDW, ALIGNED, NHI, HOST, SPEC>(a, grid, block, lds, st);
#if BANG_SEARCH_PART == 0
  return bang_search_launch_part1(&a, grid.x, block.x, lds, (void*)st);
#else
  bang_set_error("search-kernel instance psz=%u mp=%u is not part of this translation unit", a.p.psz, a.p.mp);
  return BANG_ERR_UNSUPPORTED;
#endif
}

template <int PSZ, int NDW, bool ALIGNED, int NHI>
static int launch_hd(const SearchArgs& a, dim3 grid, dim3 block, size_t lds, hipStream_t st) {
  if constexpr (search_has_spec(NDW)) {
    if (a.p.d_graph && a.p.spec_rows == 1u) return launch_part<PSZ, NDW, ALIGNED, NHI, false, true>(a, grid, block, lds, st);
  }
  return a.p.d_graph ? launch_part<PSZ, NDW, ALIGNED, NHI, false, false>(a, grid, block, lds, st)
                     : launch_part<PSZ, NDW, ALIGNED, NHI, true, false>(a, grid, block, lds, st);
}

template <int PSZ, int NDW>
static int launch_al(const SearchArgs& a, dim3 grid, dim3 block, size_t lds, hipStream_t st) {
  const bool al = ((a.p.code_stride ? a.p.code_stride : a.p.m) % 4u) == 0;          // rows start dword-aligned
  if (a.p.pq_nhi) {
    // exact-size pivot table: compiled for the two layouts of the BASELINE configs (128 dims in 70 chunks: 58 x 2 + 12 x 1;
    // 96 dims in 74 chunks: 22 x 2 + 52 x 1), rows packed (m = 70 / 74 bytes apart) or padded to a dword-aligned stride
    constexpr int NHI = (PSZ == 2 && NDW == 18) ? 58 : (PSZ == 2 && NDW == 19) ? 22 : 0;
    if constexpr (NHI != 0) {
      if ((int)a.p.pq_nhi == NHI) return al ? launch_hd<PSZ, NDW, true, NHI>(a, grid, block, lds, st) : launch_hd<PSZ, NDW, false, NHI>(a, grid, block, lds, st);
    }
    bang_set_error("no search-kernel instance for the exact-size pivot table psz=%u mp=%u nhi=%u", a.p.psz, a.p.mp, a.p.pq_nhi);
    return BANG_ERR_UNSUPPORTED;
  }
  return al ? launch_hd<PSZ, NDW, true, 0>(a, grid, block, lds, st) : launch_hd<PSZ, NDW, false, 0>(a, grid, block, lds, st);
}

static int search_dispatch(const SearchArgs& a, dim3 grid, dim3 block, size_t lds, hipStream_t st) {
  const bang_search_params* p = &a.p;
  const uint32_t key = p->psz * 100u + p->mp / 4u;
  switch (key) {
#ifndef BANG_DEV_ONLY_218          // development builds (ISA dumps, quick A/B libraries): the SIFT1B layout only
    case 108: return launch_al<1, 8>(a, grid, block, lds, st);
    case 116: return launch_al<1, 16>(a, grid, block, lds, st);
    case 124: return launch_al<1, 24>(a, grid, block, lds, st);
    case 132: return launch_al<1, 32>(a, grid, block, lds, st);
    case 208: return launch_al<2, 8>(a, grid, block, lds, st);
    case 216: return launch_al<2, 16>(a, grid, block, lds, st);
    case 219: return launch_al<2, 19>(a, grid, block, lds, st);
#endif
    case 218: return launch_al<2, 18>(a, grid, block, lds, st);
#ifndef BANG_DEV_ONLY_218
    case 404: return launch_al<4, 4>(a, grid, block, lds, st);
    case 408: return launch_al<4, 8>(a, grid, block, lds, st);
    case 802: return launch_al<8, 2>(a, grid, block, lds, st);
    case 804: return launch_al<8, 4>(a, grid, block, lds, st);
#endif
    default: bang_set_error("no search-kernel instance for psz=%u mp=%u", p->psz, p->mp); return BANG_ERR_UNSUPPORTED;
  }
}

#if BANG_SEARCH_PART == 1
extern "C" int bang_search_launch_part1(const SearchArgs* a, uint32_t grid, uint32_t block, size_t lds, void* stream) {
  return search_dispatch(*a, dim3(grid), dim3(block), lds, (hipStream_t)stream);
}
#else

#define SRCH_WG_SHARED_BYTES 2048u     // group-shared LDS behind the waves' regions (host-paced form): 128 words per pacing group, up to 4 groups
#define SRCH_DEFAULT_GROUP_WAVES 8u

// waves per workgroup that fit beside the pivot table with nctx query contexts each (0: not even one); the host-paced form also
// keeps its pacing groups' shared words there
static uint32_t waves_that_fit(uint32_t psz, uint32_t mp, uint32_t nhi, uint32_t L, uint32_t nctx, bool host_paced) {
  const size_t piv_bytes = (size_t)pivot_table_floats(psz, mp, nhi) * 4u;
  const size_t per_wave = (size_t)search_wave_words(L, nctx, (int)(mp / 4u), host_paced) * 4u;
  const size_t cap = (size_t)160 * 1024 - (host_paced ? SRCH_WG_SHARED_BYTES : 0u);
  if (piv_bytes + per_wave > cap) return 0;
  const size_t w = (cap - piv_bytes) / per_wave;
  const size_t most = (size_t)search_maxt((int)(mp / 4u), host_paced) / WAVE;      // what the instance is compiled for
  return (uint32_t)(w > most ? most : w);
}

extern "C" int bang_search_can_rerank(int dtype, uint32_t D, uint64_t vec_stride, uint32_t dim_adjust) {
  const uint32_t G = D >> 4;
  if (dim_adjust != 0 || (vec_stride & 3u) != 0 || D > 256) return 0;
  if (dtype == BANG_F32) return D >= 4 && (D & 3u) == 0;                       // (one lane per candidate: wave_rerank_f32)
  return (dtype == BANG_U8 || dtype == BANG_I8) && D >= 16 && (D & 15u) == 0 && (G & (G - 1u)) == 0;
}

extern "C" int bang_search_supported(uint32_t psz, uint32_t mp, uint32_t nhi, uint32_t L) {
  if (psz == 0 || L == 0 || L > BANG_MAX_L) return 0;
  return (int)waves_that_fit(psz, mp, nhi, L, 1, false);            // (the self-paced form; the host-paced one may hold one wave less)
}

// One workgroup per CU at most (the pivot table takes most of the LDS).  A batch smaller than CUs x waves is spread over all
// CUs with fewer waves each: a wave's iteration is latency bound, and fewer waves per CU contend less for LDS and L1.
// Host-paced form: *nctx_io = query contexts per wave (0 = auto = 1), *group_waves_io = waves per pacing group (0 = auto = 8).
extern "C" int bang_search_geometry(uint32_t psz, uint32_t mp, uint32_t nhi, uint32_t L, uint32_t Q, uint32_t max_wgs,
                                    uint32_t max_waves, int host_paced, uint32_t* workgroups, uint32_t* waves_out, uint32_t* nctx_io,
                                    uint32_t* group_waves_io) {
  if (!workgroups || !waves_out || !nctx_io || !group_waves_io || Q == 0) return BANG_ERR_ARG;
  if (psz == 0 || L == 0 || L > BANG_MAX_L) { bang_set_error("bad pq layout / L"); return BANG_ERR_ARG; }
  uint32_t nctx = *nctx_io;
  if (!host_paced) nctx = 1;
  else if (nctx == 0) nctx = 1;     // two contexts per wave measured slower (more, emptier half-rounds): kept as an experiment knob
  else if (nctx > 2) nctx = 2;
  if (host_paced && search_maxt((int)(mp / 4u), true) < 1024) nctx = 1;       // (the filter summary lives in registers: one query per wave)
  uint32_t waves = waves_that_fit(psz, mp, nhi, L, nctx, host_paced != 0);
  if (waves == 0) { bang_set_error("pivot table + one wave's worklist do not fit LDS at L=%u", L); return BANG_ERR_UNSUPPORTED; }
  if (max_waves && max_waves < waves) waves = max_waves;
  const uint32_t cus = (uint32_t)num_cus();
  uint32_t grid_n = Q < cus ? Q : cus;
  if (max_wgs && max_wgs < grid_n) grid_n = max_wgs;
  const uint32_t per_wg = ((Q + grid_n - 1) / grid_n + nctx - 1) / nctx;       // waves a workgroup needs to hold its share at once
  if (per_wg < waves) waves = per_wg;
  // A batch of one to two wave-fulls per CU runs as two EQUAL rounds (a full one followed by a nearly empty one keeps the chip a query
  // lifetime longer for a few queries: 4 000 queries 3.75 -> 3.50 ms with 8 waves, 5 000: 4.39 -> 4.28 with 10; three rounds and more: nothing)
  else if (!host_paced && per_wg > waves && per_wg <= 2 * waves) waves = (per_wg + 1) / 2;
  // pacing groups: 8 waves by default, at least 4 (the group-shared LDS area holds 4 groups), the whole workgroup if it is small
  uint32_t gs = *group_waves_io ? *group_waves_io : SRCH_DEFAULT_GROUP_WAVES;
  if (gs > 16) gs = 16;
  if (gs < 4) gs = 4;
  if (gs > waves) gs = waves;
  *workgroups = grid_n;
  *waves_out = waves;
  *nctx_io = nctx;
  *group_waves_io = host_paced ? gs : waves;
  return BANG_OK;
}

extern "C" int bang_k_search(const bang_search_params* p, void* stream) {
  if (!p) return BANG_ERR_ARG;
  if (p->Q == 0) return BANG_OK;
  if (p->R == 0 || p->R > BANG_MAX_R || p->L == 0 || p->L > BANG_MAX_L || p->m == 0) { bang_set_error("bad R/L/m"); return BANG_ERR_ARG; }
  if (p->psz == 0 || p->mp < p->m || (p->mp & 3u)) { bang_set_error("the search kernel needs the LDS-resident pivot layout"); return BANG_ERR_UNSUPPORTED; }
  if (!p->d_codes || !p->d_pivots_packed || !p->d_qc || !p->d_seed || !p->d_bloom || !p->d_cand_ids || !p->d_cand_cnt ||
      !p->d_next_query) { bang_set_error("null buffer"); return BANG_ERR_ARG; }
  if (!p->d_graph && (!p->d_rows || !p->d_ctl || !p->h_done || !p->h_parents || (p->ship_vectors && (!p->h_pub_q || !p->h_pub_c)))) {
    bang_set_error("host-paced search kernel: null pacing buffer"); return BANG_ERR_ARG;
  }
  if (p->pq_nhi && (p->psz != 2 || p->pq_nhi > p->mp)) { bang_set_error("bad pq_nhi"); return BANG_ERR_ARG; }
  if (p->cap_iter == 0 || p->cap_iter > p->L + BANG_EXTRA_ITERS - 1) { bang_set_error("bad iteration cap"); return BANG_ERR_ARG; }
  if (p->rr_queries) {
    if (!p->d_graph) { bang_set_error("the fused re-rank belongs to the self-paced form"); return BANG_ERR_ARG; }
    if (!p->rr_vec_base || !p->rr_ids_out || !p->rr_dists_out || p->rr_k == 0 || p->rr_Q_total < p->rr_q0 + p->Q ||
        !bang_search_can_rerank((int)p->rr_dtype, p->rr_D, p->rr_vec_stride, 0) || (((uintptr_t)p->rr_vec_base) & 3u) || (((uintptr_t)p->rr_queries) & 3u)) {
      bang_set_error("fused re-rank: bad arguments / unsupported vector layout"); return BANG_ERR_ARG;
    }
  }
  SearchArgs a;
  a.p = *p;
  a.lds_piv_floats = pivot_table_floats(p->psz, p->mp, p->pq_nhi);
  __builtin_memset(&a.iter, 0, sizeof(a.iter));
  a.iter.d_codes = p->d_codes; a.iter.n_nodes = p->n_nodes; a.iter.d_cand_ids = p->d_cand_ids; a.iter.d_graph = p->d_graph; a.iter.entry_len = p->entry_len;
  a.iter.vec_bytes = p->vec_bytes; a.iter.row_layout = p->row_layout; a.iter.n_rows_hbm = p->n_rows_hbm; a.iter.n_slices = p->n_slices;
  a.iter.d_rows_hbm = p->d_rows_hbm; a.iter.d_row_slices = p->d_row_slices; a.iter.slice_rows = p->slice_rows;
  a.iter.d_bloom = p->d_bloom; a.iter.cap_iter = p->cap_iter;      // (summ_iters: below, once the policy is resolved)
  uint32_t grid_n = 0, waves = 0, nctx = p->nctx, gs = p->group_waves;
  const int rc = bang_search_geometry(p->psz, p->mp, p->pq_nhi, p->L, p->Q, p->max_wgs, p->max_waves, p->d_graph ? 0 : 1, &grid_n, &waves, &nctx, &gs);
  if (rc != BANG_OK) return rc;
  a.nctx = nctx;
  a.gs = gs;
  // The filter summary trades LDS-crossbar work on the chain of every iteration (12 ds_bpermute + two transposition passes) for
  // memory requests.  A full chip is short of requests-in-flight: with it the 10 K SIFT1B-shape batch takes 8.36 instead of 9.09 ms, 5 000 /
  // 2 500 queries 4.68 / 2.37 instead of 4.99 / 2.71.  A lightly loaded one is short of nothing but the chain: 1 250 queries (5 waves per CU)
  // 1.70 ms without it against 1.77, 625 queries 1.43 against 1.53 (profiles/r04_summary_cutoff.md).  auto then: off up to 6 queries per CU.
  // (with spec_rows on, re-measured: 1 250 queries 1.69 with / 1.63 without, 1 400: 1.73 / 1.72, 1 536 = 6 per CU: 1.74-1.78 / 1.73-1.74, 1 900: 1.94 / 1.99)
  // (round 6, final kernel: 1 536 queries = 6 per CU 1.555 with / 1.585 without, 1 250 = 5 per CU 1.493 / 1.453: the cut-off is 5)
  const bool light = (p->Q + grid_n - 1) / grid_n <= 5u;
  if (a.p.summ_iters == 0u) a.p.summ_iters = light ? 1u : 0xFFFFFFFFu;
  a.iter.summ_iters = a.p.summ_iters;
  // wave priority around the request-issuing stretches of an iteration: on where wave slots are free (<= 10 queries per CU); BANG_SEARCH_PRIO = 0 / 1 forces it
  { const char* e = getenv("BANG_SEARCH_PRIO"); a.iter.prio = (e && e[0] >= '0' && e[0] <= '3') ? (e[0] == '1' ? 3u : e[0] == '3' ? 1u : (uint32_t)(e[0] - '0')) : (((p->Q + grid_n - 1) / grid_n <= 10u) ? 3u : 0u); }   // (1 = both stretches, 2 = the hand-over's only, 3 = the top's only)
  // spec_rows: one memory latency less on the chain of every iteration, the rows of the ids the filter drops fetched in vain.  Without / with,
  // ms per batch (profiles/r05_spec_rows.md) -- rows pulled, N = 1e9 random graph: 10 000 queries 8.75 / 8.38, 5 000 4.72 / 4.61, 2 500 2.33 / 2.29, 1 250 1.74 / 1.63; N = 1e8 Vamana-style graph, pulled:
  // 6.48 / 6.38, 1.77 / 1.77, 1.29 / 1.23; the same graph in HBM: 4.68 / 4.79, 1.44 / 1.41, 1.13 / 1.07.
  // auto: on where the rows are pulled, and up to 8 queries per CU where the graph is in HBM
  // (round 6, final kernel, structured 1e8-point graph in HBM, without / with: 10 000 queries 4.40 / 4.62 ms, 2 500: 1.32 / 1.345, 2 000: 1.18 / 1.13, 1 250: 1.05 / 1.02, 625: 0.94 / 0.91;
  //  the shape-only DEEP100M index -- its filter drops next to nothing -- 7.16 / 7.05 at 10 000: the policy follows the structured graph, up to 8 queries per CU)
  const bool spec_auto = p->row_layout != 0u || (p->Q + grid_n - 1) / grid_n <= 8u;
  a.p.spec_rows = (search_has_spec((int)(p->mp / 4u)) && p->d_graph && (p->spec_rows == 1u || (p->spec_rows == 0u && spec_auto))) ? 1u : 2u;
  a.wl_words = search_wl_words(p->L);
  a.wave_words = search_wave_words(p->L, nctx, (int)(p->mp / 4u), p->d_graph == nullptr);
  const size_t lds = (size_t)a.lds_piv_floats * 4 + (size_t)waves * a.wave_words * 4 + (p->d_graph ? 0u : SRCH_WG_SHARED_BYTES);
  const dim3 grid(grid_n), block(waves * WAVE);
  hipStream_t st = (hipStream_t)stream;
  return search_dispatch(a, grid, block, lds, st);
}
#endif   // BANG_SEARCH_PART == 0
