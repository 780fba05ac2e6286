// bang_api.cpp -- the C++ drop-in surface (include/bang.h == reference BANG_Base/bang.h:36-87) and the
// reference's handle-less C mirror (bang.h:89-101), both thin forwards to the C-ABI engine
// (bang_engine.cpp), like the reference's BANGSearch<T> -> BANGSearchInner<T> forwarding
// (bang_search.cu:70-135).
//
// Error behaviour follows the reference: bang_load returns false; any other failure prints
// "GPUassert: ..." to stderr and exits (utils/utils.h:28-35) because the methods return void.

#include <cstdio>
#include <cstdlib>
#include <type_traits>

#include "bang.h"
#include "bang_c.h"

namespace {
template <typename T> constexpr int dtype_of() {
  return std::is_same<T, float>::value ? BANG_F32 : std::is_same<T, int8_t>::value ? BANG_I8 : BANG_U8;
}
void die_on(int rc, const char* what) {
  if (rc != BANG_OK) {
    fprintf(stderr, "GPUassert: %s: %s (code %d)\n", what, bang_last_error(), rc);
    exit(rc < 0 ? -rc : rc);
  }
}
}  // namespace

template <typename T>
BANGSearch<T>::BANGSearch() {
  bang_engine_t* e = nullptr;
  die_on(bang_create(dtype_of<T>(), &e), "bang_create");
  m_pImpl = e;
}

template <typename T>
BANGSearch<T>::~BANGSearch() {
  bang_destroy(static_cast<bang_engine_t*>(m_pImpl));
}

template <typename T>
bool BANGSearch<T>::bang_load(char* indexfile_path_prefix) {
  const int rc = bang_load_e(static_cast<bang_engine_t*>(m_pImpl), indexfile_path_prefix);
  if (rc != BANG_OK) fprintf(stderr, "bang_load failed: %s (code %d)\n", bang_last_error(), rc);
  return rc == BANG_OK;
}

template <typename T>
void BANGSearch<T>::bang_alloc(int numQueries) {
  die_on(bang_alloc_e(static_cast<bang_engine_t*>(m_pImpl), numQueries), "bang_alloc");
}

template <typename T>
void BANGSearch<T>::bang_init(int numQueries) {
  die_on(bang_init_e(static_cast<bang_engine_t*>(m_pImpl), numQueries), "bang_init");
}

template <typename T>
void BANGSearch<T>::bang_set_searchparams(int recall, int worklist_length, DistFunc nDistFunc) {
  die_on(bang_set_searchparams_e(static_cast<bang_engine_t*>(m_pImpl), recall, worklist_length, (int)nDistFunc),
         "bang_set_searchparams");
}

template <typename T>
void BANGSearch<T>::bang_query(T* query_array, int num_queries, result_ann_t* nearestNeighbours,
                               float* nearestNeighbours_dist) {
  static_assert(sizeof(result_ann_t) == sizeof(uint64_t), "result_ann_t must be 64-bit");
  die_on(bang_query_e(static_cast<bang_engine_t*>(m_pImpl), query_array, num_queries,
                      reinterpret_cast<uint64_t*>(nearestNeighbours), nearestNeighbours_dist),
         "bang_query");
}

template <typename T>
void BANGSearch<T>::bang_free() {
  die_on(bang_free_e(static_cast<bang_engine_t*>(m_pImpl)), "bang_free");
}

template <typename T>
void BANGSearch<T>::bang_unload() {
  printf("Bang Unload \n");   // bang_search.cu:553
  die_on(bang_unload_e(static_cast<bang_engine_t*>(m_pImpl)), "bang_unload");
}

template class BANGSearch<float>;
template class BANGSearch<uint8_t>;
template class BANGSearch<int8_t>;

// ---- the reference's C mirror (bang.h:91-100): uint8 only, one process-global engine ----
static bang_engine_t* g_engine = nullptr;

extern "C" int bang_load_c(char* indexfile_path_prefix) {
  if (!g_engine) {
    const int rc = bang_create(BANG_U8, &g_engine);
    if (rc != BANG_OK) return rc;
  }
  return bang_load_e(g_engine, indexfile_path_prefix);
}
extern "C" int bang_set_searchparams_c(int recall, int worklist_length, int nDistFunc) {
  return g_engine ? bang_set_searchparams_e(g_engine, recall, worklist_length, nDistFunc) : BANG_ERR_ARG;
}
extern "C" int bang_alloc_c(int num_queries) { return g_engine ? bang_alloc_e(g_engine, num_queries) : BANG_ERR_ARG; }
extern "C" int bang_init_c(int num_queries) { return g_engine ? bang_init_e(g_engine, num_queries) : BANG_ERR_ARG; }
extern "C" int bang_query_c(uint8_t* query_array, int num_queries, unsigned long* nearestNeighbours,
                            float* nearestNeighbours_dist) {
  return g_engine ? bang_query_e(g_engine, query_array, num_queries, reinterpret_cast<uint64_t*>(nearestNeighbours),
                                 nearestNeighbours_dist)
                  : BANG_ERR_ARG;
}
extern "C" int bang_free_c(void) { return g_engine ? bang_free_e(g_engine) : BANG_ERR_ARG; }
extern "C" int bang_unload_c(void) {
  if (!g_engine) return BANG_ERR_ARG;
  const int rc = bang_unload_e(g_engine);
  bang_destroy(g_engine);
  g_engine = nullptr;
  return rc;
}
