/*
 * bang.h -- public C++ API of the MI355X-native BANG_Base search engine.
 *
 * Drop-in for the reference's BANG_Base/bang.h (lines 20-87): same class name, method names,
 * argument meaning, typedefs, enum values and macros, so BANG_Base/test_driver.cpp and
 * big-ann-benchmarks style callers compile and link against this library unchanged
 * (`-lbang`).  The implementation behind it is new (HIP/CDNA4); see DESIGN.md.
 *
 * Call order (reference: test_driver.cpp:338-557):
 *   bang_load(prefix) -> per L { bang_set_searchparams(k, L, fn); bang_alloc(Q);
 *   n x { bang_init(Q); bang_query(queries, Q, ids, dists); } bang_free(); } -> bang_unload().
 *
 * Errors: bang_load returns false (reference bang_search.cu:153-177,227-232); the void methods
 * print to stderr and exit(code) on a HIP failure exactly like the reference's gpuErrchk
 * (utils/utils.h:28-35).  Callers that want status codes use the C-ABI in bang_c.h.
 */
#ifndef BANG_H_
#define BANG_H_

#include <cstdint>

#define MAX_L 512 // L_search upper bound                      (reference bang.h:20)

typedef unsigned long result_ann_t; // 64-bit ids out           (reference bang.h:23)

typedef enum _DistFunc {            //                          (reference bang.h:26-30)
  ENUM_DIST_L2 = 0,
  ENUM_DIST_MIPS,
} DistFunc;
#define MIPS_EXTRA_DIM (1)          //                          (reference bang.h:31)

template <typename T>
class BANGSearch {                  //                          (reference bang.h:36-84)
  void* m_pImpl;

 public:
  BANGSearch();
  virtual ~BANGSearch();

  /* Load <prefix>_pq_pivots.bin, _pq_compressed.bin, _disk.bin, _disk_metadata.bin. */
  bool bang_load(char* indexfile_path_prefix);

  /* Allocate device / pinned buffers for batches of up to numQueries, sized from the current
   * search params (call bang_set_searchparams first). */
  void bang_alloc(int numQueries);

  /* Reset per-batch state (visited filters, worklists, candidate logs); required before every
   * bang_query.  Outside the timed region in the reference harness (test_driver.cpp:432-433). */
  void bang_init(int numQueries);

  void bang_set_searchparams(int recall, int worklist_length, DistFunc nDistFunc = ENUM_DIST_L2);

  /* nearestNeighbours: [num_queries][recall] ids.  nearestNeighbours_dist: recall*num_queries
   * floats in rank-major order ([rank][query]) -- the layout the reference returns
   * (bang_search.cu:999 copies the head of the [candidate][query] matrix). */
  void bang_query(T* query_array, int num_queries, result_ann_t* nearestNeighbours,
                  float* nearestNeighbours_dist);

  void bang_free();

  void bang_unload();
};

extern template class BANGSearch<float>;
extern template class BANGSearch<uint8_t>;
extern template class BANGSearch<int8_t>;

#endif // BANG_H_
