/* shape_fill.c -- fills a BANG graph file image ([T vec[D]][u32 deg][u32 nbr[R]] per node) with SHAPE-ONLY data for
 * throughput runs at billion scale (SURVEY 8(d) Tier B): uniform random vector bytes, degree = R, and R sorted distinct
 * neighbour ids drawn by stratified sampling (nbr_j uniform in [j*N/R, (j+1)*N/R)).  Multi-threaded, one pass, so a
 * 388 GB SIFT1B-shape image is produced at memory speed.  Benchmark tooling only (not part of libbang). */
#include <stdint.h>
#include <stdlib.h>
#include <string.h>
#include <fcntl.h>
#include <sys/mman.h>
#include <unistd.h>
#include <omp.h>

static inline uint64_t splitmix(uint64_t *s) {
  uint64_t z = (*s += 0x9E3779B97F4A7C15ull);
  z = (z ^ (z >> 30)) * 0xBF58476D1CE4E5B9ull;
  z = (z ^ (z >> 27)) * 0x94D049BB133111EBull;
  return z ^ (z >> 31);
}

/* returns a page-aligned anonymous mapping with transparent huge pages requested, or NULL */
void *shape_alloc(size_t bytes) {
  void *p = mmap(NULL, bytes, PROT_READ | PROT_WRITE, MAP_PRIVATE | MAP_ANONYMOUS | MAP_NORESERVE, -1, 0);
  if (p == MAP_FAILED) return NULL;
  madvise(p, bytes, MADV_HUGEPAGE);
  return p;
}
void shape_free(void *p, size_t bytes) { if (p) munmap(p, bytes); }

/* A mapping of the file `path` shared by all processes that map it (the ranks of a multi-GPU job walk ONE host graph).
 * create != 0: the file is created and sized (its pages are filled by the creator); otherwise it must exist.  NULL on failure. */
void *shape_map_shared(const char *path, size_t bytes, int create) {
  int fd = open(path, create ? (O_CREAT | O_RDWR | O_TRUNC) : O_RDWR, 0600);
  if (fd < 0) return NULL;
  if (create && ftruncate(fd, (off_t)bytes) != 0) { close(fd); return NULL; }
  void *p = mmap(NULL, bytes, PROT_READ | PROT_WRITE, MAP_SHARED, fd, 0);
  close(fd);
  if (p == MAP_FAILED) return NULL;
  madvise(p, bytes, MADV_HUGEPAGE);
  return p;
}

/* one node's graph entry; every node has its own generator state, so any range (or single node) can be produced on its own.
 * vec_float != 0: the vector part holds vec_bytes/4 floats, uniform in [-1, 1) (random bytes are not sane floats) */
static inline void fill_entry(uint8_t *e, uint64_t i, uint64_t N, uint32_t vec_bytes, uint32_t R, uint64_t seed, int vec_float,
                              const uint64_t *lo_, const uint64_t *w_) {
  uint64_t s = seed ^ (i * 0xD1342543DE82EF95ull);
  uint32_t k = 0;
  if (vec_float) {
    for (; k + 8 <= vec_bytes; k += 8) {
      const uint64_t r = splitmix(&s);
      const float f[2] = {(float)((int32_t)(r & 0xFFFFFF) - 0x800000) * (1.0f / 8388608.0f),
                          (float)((int32_t)((r >> 32) & 0xFFFFFF) - 0x800000) * (1.0f / 8388608.0f)};
      memcpy(e + k, f, 8);
    }
    for (; k + 4 <= vec_bytes; k += 4) {
      const float f = (float)((int32_t)(splitmix(&s) & 0xFFFFFF) - 0x800000) * (1.0f / 8388608.0f);
      memcpy(e + k, &f, 4);
    }
  }
  for (; k + 8 <= vec_bytes; k += 8) { uint64_t r = splitmix(&s); memcpy(e + k, &r, 8); }
  for (; k < vec_bytes; ++k) e[k] = (uint8_t)splitmix(&s);
  uint32_t deg = R;
  memcpy(e + vec_bytes, &deg, 4);
  uint32_t *nb = (uint32_t *)(e + vec_bytes + 4);
  for (uint32_t j = 0; j < R; ++j) {
    const uint64_t lo = lo_[j], hi = lo + w_[j];                /* strata are disjoint: ids come out sorted & distinct */
    uint64_t id = lo + (uint64_t)(((unsigned __int128)splitmix(&s) * w_[j]) >> 64);
    if (id == i) id = (id + 1 < hi) ? id + 1 : lo;             /* no self loop */
    nb[j] = (uint32_t)id;
  }
}

/* entries of the nodes [first, first + count) of an N-node graph, written to dst[0 .. count * entry) */
void shape_fill_range(uint8_t *dst, uint64_t N, uint64_t first, uint64_t count, uint32_t vec_bytes, uint32_t R, uint64_t seed,
                      int nthreads, int vec_float) {
  const uint64_t entry = (uint64_t)vec_bytes + 4 + 4ull * R;
  uint64_t lo_[64], w_[64];
  if (R > 64) return;
  for (uint32_t j = 0; j < R; ++j) { lo_[j] = N * j / R; w_[j] = N * (j + 1) / R - lo_[j]; if (w_[j] == 0) w_[j] = 1; }
  if (nthreads > 0) omp_set_num_threads(nthreads);
#pragma omp parallel for schedule(static)
  for (int64_t k = 0; k < (int64_t)count; ++k)
    fill_entry(dst + (uint64_t)k * entry, first + (uint64_t)k, N, vec_bytes, R, seed, vec_float, lo_, w_);
}

void shape_fill_graph(uint8_t *graph, uint64_t N, uint32_t vec_bytes, uint32_t R, uint64_t seed, int nthreads, int vec_float) {
  shape_fill_range(graph, N, 0, N, vec_bytes, R, seed, nthreads, vec_float);
}

/* the same generator as an ENTRY SOURCE of libbang's streamed load (bang_entry_source, include/bang_c.h): the index never exists
 * in host memory as a whole -- the engine pulls it through in chunks */
typedef struct { uint64_t N, seed; uint32_t vec_bytes, R; int32_t nthreads, vec_float; } shape_source;
int shape_entry_source(void *ctx, uint64_t first, uint64_t count, uint8_t *dst) {
  const shape_source *c = (const shape_source *)ctx;
  if (first + count > c->N) return -1;
  shape_fill_range(dst, c->N, first, count, c->vec_bytes, c->R, c->seed, c->nthreads, c->vec_float);
  return 0;
}

void shape_fill_bytes(uint8_t *dst, uint64_t n, uint64_t seed, int nthreads) {
  if (nthreads > 0) omp_set_num_threads(nthreads);
  const uint64_t blocks = (n + 4095) / 4096;
#pragma omp parallel for schedule(static)
  for (int64_t b = 0; b < (int64_t)blocks; ++b) {
    uint64_t s = seed ^ ((uint64_t)b * 0xA24BAED4963EE407ull);
    const uint64_t lo = (uint64_t)b * 4096, hi = lo + 4096 < n ? lo + 4096 : n;
    uint64_t k = lo;
    for (; k + 8 <= hi; k += 8) { uint64_t r = splitmix(&s); memcpy(dst + k, &r, 8); }
    for (; k < hi; ++k) dst[k] = (uint8_t)splitmix(&s);
  }
}
