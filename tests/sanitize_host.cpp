// Driver of the library's HOST entry points (id-grec_amd/csrc/idg_host.cpp) for a build with AddressSanitizer and
// UndefinedBehaviorSanitizer (tests/test_host_sanitizers.py compiles and runs it; the GPU pool has no sanitizers, the host
// code is what can run under them).  Exercises the rating-file parser on good and malformed files, the adjacency builder
// (with / without self loops, duplicate pairs, isolated nodes, E = 0), the MT19937 restatement (sampler, shuffle,
// random.sample in both branches, raw bytes) at their edge sizes.  Exit status 0 = every call behaved (good input: IDG_OK;
// bad input: an error code and a message) — a sanitizer report aborts the process.
#include <algorithm>
#include <cstdio>
#include <cstdlib>
#include <cstring>
#include <string>
#include <vector>

#include "idgrec.h"

static int fails = 0;
#define EXPECT(cond, ...)                        \
  do {                                           \
    if (!(cond)) {                               \
      std::fprintf(stderr, "FAIL: " __VA_ARGS__); \
      std::fprintf(stderr, "\n");                \
      ++fails;                                   \
    }                                            \
  } while (0)

static void write_file(const std::string& path, const std::string& text) {
  FILE* f = std::fopen(path.c_str(), "wb");
  if (!f) std::abort();
  std::fwrite(text.data(), 1, text.size(), f);
  std::fclose(f);
}

static void whole_path(const char* path, uint32_t seed) {
  idg_ratings* r = nullptr;
  int64_t E = 0, L = 0, mu = 0, mi = 0;
  int rc = idg_ratings_open(path, &r, &E, &L, &mu, &mi);
  EXPECT(rc == IDG_OK, "open %s: %s", path, idg_last_error());
  if (rc != IDG_OK) return;
  std::vector<int64_t> users((size_t)E), items((size_t)E), lu((size_t)L), lc((size_t)L);
  EXPECT(idg_ratings_read(r, users.data(), items.data(), lu.data(), lc.data()) == IDG_OK, "read");
  EXPECT(idg_ratings_read(r, nullptr, nullptr, nullptr, nullptr) == IDG_OK, "read with NULL outputs");
  idg_ratings_destroy(r);
  const int64_t U = mu + 3, I = mi + 2;  // a few isolated nodes at the end of either side
  for (int self = 0; self < 2; ++self) {
    int64_t nnz = 0;
    EXPECT(idg_build_norm_adj(U, I, E, users.data(), items.data(), self, nullptr, &nnz, nullptr, nullptr, nullptr) == IDG_OK,
           "adjacency count: %s", idg_last_error());
    std::vector<int64_t> indptr((size_t)(U + I + 1));
    std::vector<int32_t> indices((size_t)std::max<int64_t>(nnz, 1));
    std::vector<float> values((size_t)std::max<int64_t>(nnz, 1));
    EXPECT(idg_build_norm_adj(U, I, E, users.data(), items.data(), self, nullptr, &nnz, indptr.data(), indices.data(),
                              values.data()) == IDG_OK, "adjacency fill: %s", idg_last_error());
    EXPECT(indptr[0] == 0 && indptr[(size_t)(U + I)] == nnz, "indptr ends");
    for (int64_t i = 0; i < U + I; ++i) EXPECT(indptr[(size_t)i] <= indptr[(size_t)i + 1], "indptr not monotone at %lld", (long long)i);
    for (int64_t j = 0; j < nnz; ++j) EXPECT(indices[(size_t)j] >= 0 && indices[(size_t)j] < U + I && values[(size_t)j] == values[(size_t)j], "entry %lld", (long long)j);
  }
  // positives per user (CSR, ascending, duplicates removed) for the sampler
  std::vector<std::pair<int64_t, int64_t>> pairs((size_t)E);
  for (int64_t e = 0; e < E; ++e) pairs[(size_t)e] = {users[(size_t)e], items[(size_t)e]};
  std::sort(pairs.begin(), pairs.end());
  pairs.erase(std::unique(pairs.begin(), pairs.end()), pairs.end());
  std::vector<int64_t> ptr((size_t)U + 1, 0);
  std::vector<int32_t> idx(pairs.size() ? pairs.size() : 1);
  for (size_t k = 0; k < pairs.size(); ++k) ptr[(size_t)pairs[k].first + 1]++, idx[k] = (int32_t)pairs[k].second;
  for (int64_t u = 0; u < U; ++u) ptr[(size_t)u + 1] += ptr[(size_t)u];
  idg_rng* rng = nullptr;
  EXPECT(idg_rng_create(seed, &rng) == IDG_OK, "rng");
  std::vector<int64_t> triples((size_t)std::max<int64_t>(3 * E, 1));
  int64_t count = -1;
  EXPECT(idg_sample_epoch(rng, users.data(), items.data(), E, ptr.data(), idx.data(), U, I, triples.data(), &count) == IDG_OK,
         "sample: %s", idg_last_error());
  EXPECT(count == E, "sampled %lld of %lld", (long long)count, (long long)E);
  for (int64_t t = 0; t < count; ++t) {
    const int64_t u = triples[(size_t)(3 * t)], neg = triples[(size_t)(3 * t + 2)];
    EXPECT(neg >= 0 && neg < I, "negative out of range");
    EXPECT(!std::binary_search(idx.begin() + ptr[(size_t)u], idx.begin() + ptr[(size_t)u + 1], (int32_t)neg), "negative is a positive");
  }
  std::vector<int64_t> perm((size_t)std::max<int64_t>(E, 1));
  EXPECT(idg_shuffle_perm(rng, E, perm.data()) == IDG_OK, "shuffle");
  std::vector<int64_t> sorted(perm.begin(), perm.begin() + E);
  std::sort(sorted.begin(), sorted.end());
  for (int64_t i = 0; i < E; ++i) EXPECT(sorted[(size_t)i] == i, "not a permutation");
  idg_rng_destroy(rng);
}

int main(int argc, char** argv) {
  if (argc < 2) return 2;
  const std::string tmp = argv[1];
  EXPECT(idg_version() == IDG_VERSION, "version");
  for (int a = 2; a < argc; ++a) whole_path(argv[a], 2024u + (uint32_t)a);

  // ---- the parser on odd and malformed files
  struct Case { const char* name; std::string text; bool ok; int64_t edges, lines; };
  const Case cases[] = {
      {"empty", "", true, 0, 0},
      {"no_newline_at_end", "0 1 2\n1 3", true, 3, 2},
      {"user_without_items", "0\n1 5\n", true, 1, 2},
      {"blanks_tabs_cr", "0  1\t2 \r\n3 4\r\n", true, 3, 2},
      {"plus_sign", "+1 +2\n", true, 1, 1},
      {"blank_line", "0 1\n\n1 2\n", false, 0, 0},
      {"only_blanks_line", "0 1\n   \n", false, 0, 0},
      {"word", "0 1 x\n", false, 0, 0},
      {"float", "0 1.5\n", false, 0, 0},
      {"lonely_sign", "0 -\n", false, 0, 0},
      {"huge_number", "0 99999999999999999999999999999999\n", false, 0, 0},
      {"binary", std::string("\x00\x01\xff\n", 4), false, 0, 0},
  };
  for (const Case& c : cases) {
    const std::string path = tmp + "/" + c.name + ".txt";
    write_file(path, c.text);
    idg_ratings* r = nullptr;
    int64_t E = -1, L = -1, mu = 0, mi = 0;
    const int rc = idg_ratings_open(path.c_str(), &r, &E, &L, &mu, &mi);
    if (c.ok) {
      EXPECT(rc == IDG_OK && E == c.edges && L == c.lines, "%s: rc %d, %lld edges, %lld lines (%s)", c.name, rc, (long long)E,
             (long long)L, idg_last_error());
      if (rc == IDG_OK) idg_ratings_destroy(r);
    } else {
      EXPECT(rc != IDG_OK && idg_last_error()[0] != 0, "%s: accepted (rc %d)", c.name, rc);
    }
  }
  {
    idg_ratings* r = nullptr;
    int64_t E, L, mu, mi;
    EXPECT(idg_ratings_open((tmp + "/does_not_exist.txt").c_str(), &r, &E, &L, &mu, &mi) != IDG_OK, "missing file accepted");
    EXPECT(idg_ratings_open(nullptr, &r, &E, &L, &mu, &mi) != IDG_OK, "NULL path accepted");
  }

  // ---- adjacency at its edges
  {
    int64_t nnz = -1;
    EXPECT(idg_build_norm_adj(3, 2, 0, nullptr, nullptr, 0, nullptr, &nnz, nullptr, nullptr, nullptr) == IDG_OK && nnz == 0, "E = 0");
    EXPECT(idg_build_norm_adj(3, 2, 0, nullptr, nullptr, 1, nullptr, &nnz, nullptr, nullptr, nullptr) == IDG_OK && nnz == 5, "E = 0 with self loops: %lld", (long long)nnz);
    const int64_t u[] = {0, 0, 0, 2}, it[] = {1, 1, 0, 1};  // a duplicate pair
    EXPECT(idg_build_norm_adj(3, 2, 4, u, it, 0, nullptr, &nnz, nullptr, nullptr, nullptr) == IDG_OK && nnz == 6, "duplicates: %lld", (long long)nnz);
    const int64_t bad_u[] = {0, 7}, bad_i[] = {1, 1};
    EXPECT(idg_build_norm_adj(3, 2, 2, bad_u, bad_i, 0, nullptr, &nnz, nullptr, nullptr, nullptr) != IDG_OK, "user id out of range accepted");
    const int64_t neg_i[] = {1, -1};
    EXPECT(idg_build_norm_adj(3, 2, 2, u, neg_i, 0, nullptr, &nnz, nullptr, nullptr, nullptr) != IDG_OK, "negative item id accepted");
  }

  // ---- the generator at its edges
  {
    idg_rng* rng = nullptr;
    EXPECT(idg_rng_create(0u, &rng) == IDG_OK, "rng");
    uint32_t key[624];
    int32_t pos = 0;
    EXPECT(idg_rng_get_state(rng, key, &pos) == IDG_OK && idg_rng_set_state(rng, key, pos) == IDG_OK, "state round trip");
    EXPECT(idg_rng_set_state(rng, key, 625) != IDG_OK, "pos 625 accepted");
    uint8_t bytes[7];
    EXPECT(idg_rng_bytes(rng, 7, bytes) == IDG_OK && idg_rng_bytes(rng, 0, nullptr) == IDG_OK, "bytes");
    int64_t v[5];
    EXPECT(idg_rng_randint(rng, 1, 5, v) == IDG_OK && v[0] == 0 && v[4] == 0, "randint(0, 1)");
    EXPECT(idg_rng_randint(rng, (int64_t)1 << 40, 5, v) == IDG_OK, "randint beyond 2^32");
    EXPECT(idg_rng_randint(rng, 0, 5, v) != IDG_OK, "high = 0 accepted");
    EXPECT(idg_shuffle_perm(rng, 0, nullptr) == IDG_OK && idg_shuffle_perm(rng, 1, v) == IDG_OK && v[0] == 0, "shuffle of 0 / 1");
    std::vector<int64_t> s(1000);
    for (int pool = 0; pool < 2; ++pool) {
      EXPECT(idg_py_random_sample(rng, 1000, 0, pool, nullptr) == IDG_OK, "sample k = 0");
      EXPECT(idg_py_random_sample(rng, 1000, 1000, pool, s.data()) == IDG_OK, "sample k = n");
      std::vector<int64_t> t(s);
      std::sort(t.begin(), t.end());
      for (int64_t i = 0; i < 1000; ++i) EXPECT(t[(size_t)i] == i, "sample k = n is not a permutation");
      EXPECT(idg_py_random_sample(rng, 1000, 37, pool, s.data()) == IDG_OK, "sample");
    }
    EXPECT(idg_py_random_sample(rng, 5, 6, 0, s.data()) != IDG_OK, "k > n accepted");
    // a user whose positives are every item but one: the negative draw must terminate on that one
    const int64_t tu[] = {0}, ti[] = {0}, ptr[] = {0, 3};
    const int32_t idx[] = {0, 1, 2};
    int64_t tri[3], cnt = 0;
    EXPECT(idg_sample_epoch(rng, tu, ti, 1, ptr, idx, 1, 4, tri, &cnt) == IDG_OK && cnt == 1 && tri[2] == 3, "last free item");
    EXPECT(idg_sample_epoch(rng, tu, ti, 1, ptr, idx, 1, 3, tri, &cnt) != IDG_OK, "a user with every item positive accepted (would never end)");
    idg_rng_destroy(rng);
  }
  if (fails) std::fprintf(stderr, "%d check(s) failed\n", fails);
  return fails ? 1 : 0;
}
