// test_driver.cpp -- the `bang_search` command line harness.
//
// Keeps the CLI contract and the output table of the reference harness
// (BANG_Base/test_driver.cpp:564-599 usage, :402-411/:526 table, :424-439 five timed runs per L
// with bang_init outside the timed region, :43-93 tie-aware recall, :238-272 truthset format,
// :281-336 MIPS query pre-transform) on top of the bang.h API:
//
//   bang_search <index prefix> <query.bin> <groundtruth.bin> <numQueries> <k> <uint8|int8|float> <l2|mips>
//        -> interactive: asks for the worklist length L, prints 5 runs, asks whether to continue
//   bang_search <...same 7 args...> <anything>
//        -> auto sweep L = k, k+12, k+24, ... <= MAX_L
//   bang_search <query.bin> <numQueries>
//        -> writes <query.bin>_transformed: unit-normalised float queries with one zero dim appended
//
// Engine options (graph placement, lanes) are taken from the environment, see bang_create().

#include <sys/stat.h>

#include <algorithm>
#include <chrono>
#include <cmath>
#include <cstdint>
#include <cstdio>
#include <cstdlib>
#include <cstring>
#include <fstream>
#include <iostream>
#include <set>
#include <string>
#include <vector>

#include "bang.h"

namespace {

unsigned long long now_millis() {   // millisecond wall clock, as the reference (:35-41)
  using namespace std::chrono;
  return (unsigned long long)duration_cast<milliseconds>(system_clock::now().time_since_epoch()).count();
}

// k-recall@k as the reference harness defines it (test_driver.cpp:43-93): per query, the ground-truth set is the first k
// entries extended over every further entry that ties with the k-th ground-truth DISTANCE; the score is the number of
// distinct ground-truth ids found among the k returned ids (compared as 32-bit ids); the result is the mean, in percent of k.
double calculate_recall(unsigned num_queries, const unsigned* truth_ids, const float* truth_dists, unsigned truth_width,
                        const result_ann_t* found, unsigned found_width, unsigned k) {
  std::vector<unsigned> want, got;
  double hits = 0;
  for (size_t q = 0; q < num_queries; ++q) {
    const unsigned* t_ids = truth_ids + q * truth_width;
    size_t n_want = k;
    if (truth_dists) {                                  // extend over the ties of the k-th distance
      const float* t_d = truth_dists + q * truth_width;
      n_want = k - 1;
      while (n_want < truth_width && t_d[n_want] == t_d[k - 1]) ++n_want;
    }
    want.assign(t_ids, t_ids + n_want);
    std::sort(want.begin(), want.end());
    want.erase(std::unique(want.begin(), want.end()), want.end());
    got.resize(k);
    for (unsigned j = 0; j < k; ++j) got[j] = (unsigned)found[q * found_width + j];
    std::sort(got.begin(), got.end());
    got.erase(std::unique(got.begin(), got.end()), got.end());
    size_t a = 0, b = 0;                                // size of the intersection of two sorted id lists
    while (a < want.size() && b < got.size()) {
      if (want[a] < got[b]) ++a;
      else if (got[b] < want[a]) ++b;
      else { hits += 1; ++a; ++b; }
    }
  }
  return hits / num_queries * (100.0 / k);
}

bool file_exists(const std::string& name) {
  struct stat st;
  return stat(name.c_str(), &st) == 0;
}

// {i32 n, i32 K, u32 ids[n][K], f32 dists[n][K]}, size-checked (:238-272)
bool load_truthset(const std::string& path, std::vector<uint32_t>& ids, std::vector<float>& dists, size_t& npts, size_t& dim) {
  std::ifstream in(path, std::ios::binary | std::ios::ate);
  if (!in.is_open()) return false;
  const size_t fsize = (size_t)in.tellg();
  in.seekg(0);
  int32_t n = 0, k = 0;
  in.read((char*)&n, 4);
  in.read((char*)&k, 4);
  npts = (unsigned)n;
  dim = (unsigned)k;
  const size_t expect = 2 * npts * dim * sizeof(uint32_t) + 2 * sizeof(uint32_t);
  if (fsize != expect) {
    std::cout << "Error. File size mismatch. Actual size is " << fsize << " while expected size is  " << expect
              << " npts = " << npts << " dim = " << dim << std::endl;
    exit(1);
  }
  ids.resize(npts * dim);
  dists.resize(npts * dim);
  in.read((char*)ids.data(), (std::streamsize)(npts * dim * 4));
  in.read((char*)dists.data(), (std::streamsize)(npts * dim * 4));
  return true;
}

// MIPS helper mode (:281-336): normalise every query, append a zero coordinate, save as float bin
void preprocess_query_file(const std::string& path, int numQueries) {
  std::ifstream in(path, std::ios::binary);
  if (!in.is_open()) { printf("Error.. Could not open the Query File: %s\n", path.c_str()); return; }
  in.seekg(4);
  int dim = 0;
  in.read((char*)&dim, 4);
  std::vector<float> q((size_t)numQueries * dim), out((size_t)numQueries * (dim + 1));
  in.read((char*)q.data(), (std::streamsize)(q.size() * 4));
  for (int i = 0; i < numQueries; ++i) {
    float norm = 0;
    for (int j = 0; j < dim; ++j) norm += q[(size_t)i * dim + j] * q[(size_t)i * dim + j];
    norm = std::sqrt(norm);
    for (int j = 0; j < dim; ++j) out[(size_t)i * (dim + 1) + j] = q[(size_t)i * dim + j] / norm;
    out[(size_t)i * (dim + 1) + dim] = 0;
  }
  const std::string dst = path + "_transformed";
  std::ofstream w(dst, std::ios::binary);
  const int32_t n32 = numQueries, d32 = dim + 1;
  std::cout << "Writing bin: " << dst << std::endl;
  w.write((const char*)&n32, 4);
  w.write((const char*)&d32, 4);
  w.write((const char*)out.data(), (std::streamsize)(out.size() * 4));
  std::cout << "Finished writing bin." << std::endl;
}

template <typename T>
int run_anns(int argc, char** argv) {
  BANGSearch<T> bang;
  if (!bang.bang_load(argv[1])) {
    std::cout << "Error: Bang_load failed" << std::endl;
    return -1;
  }
  const int numQueries = atoi(argv[4]);
  const std::string query_file(argv[2]);
  std::ifstream qin(query_file, std::ios::binary);
  if (!qin.is_open()) { printf("Error.. Could not open the Query File: %s\n", query_file.c_str()); return -1; }
  qin.seekg(4);                                   // skip the point count, read the dimension (:360-362)
  int dim = 0;
  qin.read((char*)&dim, 4);
  std::vector<T> queries((size_t)numQueries * dim);
  qin.read((char*)queries.data(), (std::streamsize)(queries.size() * sizeof(T)));
  qin.close();

  const int k = atoi(argv[5]);
  const DistFunc fn = (std::string(argv[7]) == "mips") ? ENUM_DIST_MIPS : ENUM_DIST_L2;
  const bool interactive = (argc == 8);           // (:384-386)
  int L = k;
  const int step = 12;                            // (:377)
  for (int round = 0;; ++round) {
    if (interactive) {
      std::cout << "Enter value of WorkList Length" << std::endl;
      if (!(std::cin >> L)) break;
      if (L < k) { std::cout << " Error: WorkList Length must be at least recall_at" << std::endl; continue; }
      if (L > MAX_L) { std::cout << " Error: WorkList Length must be at most " << MAX_L << std::endl; continue; }
    } else {
      if (round > 0) L += step;
      if (L > MAX_L) break;
    }
    if (interactive || round == 0) {
      std::cout << "L\t" << "Time \t" << "QPS\t" << "\t" << k << "-r@" << k << std::endl;
      std::cout << "--\t" << "---- \t" << "---\t" << "\t------" << std::endl;
    }
    bang.bang_set_searchparams(k, L, fn);
    bang.bang_alloc(numQueries);
    for (int run = 0; run < 5; ++run) {           // (:424)
      std::vector<result_ann_t> ids((size_t)k * numQueries);
      std::vector<float> dists((size_t)k * numQueries);
      bang.bang_init(numQueries);                 // outside the timed region (:432-433)
      const auto t0 = now_millis();
      bang.bang_query(queries.data(), numQueries, ids.data(), dists.data());
      const auto t1 = now_millis();
      const double wall = (double)(t1 - t0);
      const double qps = (numQueries * 1000.0) / wall;
      std::vector<uint32_t> gt_ids;
      std::vector<float> gt_dists;
      size_t gt_num = 0, gt_dim = 0;
      if (!file_exists(argv[3]) || !load_truthset(argv[3], gt_ids, gt_dists, gt_num, gt_dim)) {
        std::cout << "Groundtruth file could not be loaded:" << argv[3] << std::endl;
        exit(1);
      }
      // (a float, as in the reference (:506): its double -> float step decides the second decimal at x.xx5 -- 88.535 prints as 88.54)
      const float recall = (float)calculate_recall((unsigned)numQueries, gt_ids.data(), gt_dists.data(), (unsigned)gt_dim,
                                                   ids.data(), (unsigned)k, (unsigned)k);
      std::cout.setf(std::ios_base::fixed, std::ios_base::floatfield);
      std::cout.precision(2);
      std::cout << L << "\t" << wall << "\t" << qps << "\t" << recall << std::endl;   // (:526)
    }
    bang.bang_free();
    if (interactive) {
      char c = 'n';
      std::cout << "Try Next run ? [y|n]" << std::endl;
      std::cin >> c;
      if (c == 'n') break;
    }
  }
  bang.bang_unload();
  return 0;
}

}  // namespace

int main(int argc, char** argv) {
  if (argc == 3) {
    preprocess_query_file(argv[1], atoi(argv[2]));
    return 0;
  }
  if (argc < 8) {
    std::cerr << "Too few parameters! " << argv[0]
              << " <path with file prefix to the director with index files > <query file> <GroundTruth File> "
                 "<NumQueries> <recall parameter k> <data type : uint8, int8 or float> <dist funct: l2 or mips>"
              << std::endl;
    exit(1);
  }
  const std::string dt(argv[6]);
  if (dt == "uint8") return run_anns<uint8_t>(argc, argv);
  if (dt == "int8") return run_anns<int8_t>(argc, argv);
  if (dt == "float") return run_anns<float>(argc, argv);
  std::cerr << "Invalid data type specified" << std::endl;
  exit(1);
}
