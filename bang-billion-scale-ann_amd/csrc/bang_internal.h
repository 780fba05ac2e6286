// bang_internal.h -- declarations shared by the kernel launchers and the host engine (not public).
#ifndef BANG_INTERNAL_H_
#define BANG_INTERNAL_H_

#include <stdint.h>

#include "bang_c.h"

#ifdef __cplusplus
extern "C" {
#endif

// printf-style; stores a thread-local message returned by bang_last_error()
void bang_set_error(const char* fmt, ...) __attribute__((format(printf, 1, 2)));

// bang_k_rerank on the sub-range [q0, q0+nq) of a batch of Q_total queries (one lane's share)
int bang_k_rerank_range(const void* d_vec_base, uint64_t vec_stride, const void* d_medoid_vec, const void* d_queries,
                        int dtype, const uint32_t* d_cand_ids, const uint32_t* d_cand_row, const uint32_t* d_cand_cnt,
                        uint32_t cand_stride, uint32_t q0, uint32_t nq, uint32_t Q_total, uint32_t D, uint32_t k,
                        uint32_t dim_adjust, uint64_t* d_ids_out, float* d_dists_out, void* stream);

// the same with the vector log laid out [query][candidate index][vec_stride] (host-paced search kernel with shipped vectors)
int bang_k_rerank_byquery(const void* d_fp, uint64_t vec_stride, const void* d_medoid_vec, const void* d_queries, int dtype,
                          const uint32_t* d_cand_ids, const uint32_t* d_cand_cnt, uint32_t cand_stride, uint32_t q0, uint32_t nq,
                          uint32_t Q_total, uint32_t D, uint32_t k, uint32_t dim_adjust, uint64_t* d_ids_out, float* d_dists_out,
                          void* stream);

// device side of bang_init: candidate log = [MEDOID], empty worklists, mark = 0x01010101
int bang_k_init_state(uint32_t Q, uint32_t medoid, uint32_t cand_stride, uint32_t* d_cand_ids, uint32_t* d_cand_row,
                      uint32_t* d_cand_cnt, uint32_t* d_wl_cnt, uint32_t* d_mark, uint32_t* d_parents, uint32_t* d_cnt,
                      void* stream);

#define BANG_ADJ_PAD 0xFFFFFFFFu      /* unused slot of a 256-byte adjacency row (bang_search_params.row_layout = 1) */

// bang_init in one launch: clears the visited filters, resets the per-query state (as bang_k_init_state) and the diagnostic counters
typedef struct {
  uint32_t Q, medoid, cand_stride, n_active;
  uint32_t* d_bloom;                                   /* [Q][BANG_BF_WORDS] */
  uint32_t *d_cand_ids, *d_cand_row, *d_cand_cnt, *d_wl_cnt, *d_mark, *d_parents, *d_cnt;
  uint32_t *d_qstats, *d_qskip, *d_active;   /* may be NULL */
} bang_init_params;
int bang_k_init_all(const bang_init_params* a, void* stream);

// Hand-shake timeouts of the host-paced search kernel are run-time options (host_walk_timeout_ms = 20 000, kernel_go_timeout_ms =
// 30 000: bang_options.cpp).  The HOST gives up first -- it then stores STOP into every pacing word, which ends the kernel at once;
// a pacing group only gives up on its own (the host process is gone) well after that.  Default of a launch without the field set:
#define BANG_KERNEL_GO_TIMEOUT_TICKS 3000000000ull   /* 30 s of the 100 MHz s_memrealtime clock */
#define BANG_RESULT_MAILBOX_BYTES (8 * 1024 * 1024)   // results (ids + distances) up to this size return through the pinned mirror in one copy (BANG_MAILBOX_BYTES)

#define BANG_MAX_LANES 256          /* option "lanes" / BANG_LANES; one abort word per lane sits behind the result mirror */
#define BANG_ERR_STALE_ROWS (-100)  /* internal: a pull-rows file of another graph was found (and removed); the caller rebuilds */

int bang_num_cus(void);
// 1 if the fused kernel has an instance for the exact-size ("ragged") pivot table of this layout
int bang_ragged_supported(uint32_t psz, uint32_t mp, uint32_t nhi, uint32_t m);

#ifdef __cplusplus
}
#endif
#endif
