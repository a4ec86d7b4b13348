// Host half of libidgrec.so: the NumPy-legacy MT19937 stream, the BPR negative sampler,
// the epoch permutation, the rating-file parser and the normalised-adjacency builder.
// Pure C++ (no HIP calls) — these run on the host in the reference too, as Python loops.
#include <algorithm>
#include <cerrno>
#include <cmath>
#include <cstdint>
#include <cstdlib>
#include <cstring>
#include <new>
#include <string>
#include <vector>

#include "idg_common.h"

namespace idg {

static thread_local char g_err[512] = "";

void set_error(const char* fmt, ...) {
  va_list ap;
  va_start(ap, fmt);
  vsnprintf(g_err, sizeof g_err, fmt, ap);
  va_end(ap);
}
const char* get_error() { return g_err; }

}  // namespace idg

// ---------------------------------------------------------------------------------------
// MT19937 exactly as numpy.random.RandomState drives it.
// ---------------------------------------------------------------------------------------
struct idg_rng {
  uint32_t key[624];
  int pos;

  void seed(uint32_t s) {
    // np.random.seed(int) -> init_genrand (tools.py:10 in the reference).
    key[0] = s;
    for (int i = 1; i < 624; ++i) key[i] = 1812433253u * (key[i - 1] ^ (key[i - 1] >> 30)) + (uint32_t)i;
    pos = 624;
  }

  void refill() {
    const uint32_t UPPER = 0x80000000u, LOWER = 0x7fffffffu, MAG = 0x9908b0dfu;
    int i = 0;
    for (; i < 624 - 397; ++i) {
      uint32_t y = (key[i] & UPPER) | (key[i + 1] & LOWER);
      key[i] = key[i + 397] ^ (y >> 1) ^ ((y & 1u) ? MAG : 0u);
    }
    for (; i < 623; ++i) {
      uint32_t y = (key[i] & UPPER) | (key[i + 1] & LOWER);
      key[i] = key[i + (397 - 624)] ^ (y >> 1) ^ ((y & 1u) ? MAG : 0u);
    }
    uint32_t y = (key[623] & UPPER) | (key[0] & LOWER);
    key[623] = key[396] ^ (y >> 1) ^ ((y & 1u) ? MAG : 0u);
    pos = 0;
  }

  inline uint32_t next32() {
    if (pos >= 624) refill();
    uint32_t y = key[pos++];
    y ^= y >> 11;
    y ^= (y << 7) & 0x9d2c5680u;
    y ^= (y << 15) & 0xefc60000u;
    y ^= y >> 18;
    return y;
  }
  inline uint64_t next64() {
    uint64_t hi = next32();
    return (hi << 32) | next32();
  }

  // Uniform integer in [0, max] by masked rejection: the scheme behind both the legacy
  // scalar randint(0, max+1) and shuffle's random_interval(max).  max == 0 draws nothing.
  inline uint64_t bounded(uint64_t max) {
    if (max == 0) return 0;
    uint64_t mask = max;
    mask |= mask >> 1;
    mask |= mask >> 2;
    mask |= mask >> 4;
    mask |= mask >> 8;
    mask |= mask >> 16;
    mask |= mask >> 32;
    uint64_t v;
    if (max <= 0xffffffffull) {
      const uint32_t m32 = (uint32_t)mask;
      do v = next32() & m32;
      while (v > max);
    } else {
      do v = next64() & mask;
      while (v > max);
    }
    return v;
  }
};

extern "C" {

int idg_version(void) { return IDG_VERSION; }
const char* idg_last_error(void) { return idg::get_error(); }

int idg_rng_create(uint32_t seed, idg_rng** out) {
  IDG_REQUIRE(out != nullptr, "idg_rng_create: out is NULL");
  idg_rng* r = new (std::nothrow) idg_rng;
  if (!r) return idg::fail(IDG_E_NOMEM, "idg_rng_create: out of memory");
  r->seed(seed);
  *out = r;
  return IDG_OK;
}

int idg_rng_destroy(idg_rng* rng) {
  delete rng;
  return IDG_OK;
}

int idg_rng_get_state(const idg_rng* rng, uint32_t key[624], int32_t* pos) {
  IDG_REQUIRE(rng && key && pos, "idg_rng_get_state: NULL argument");
  std::memcpy(key, rng->key, sizeof rng->key);
  *pos = rng->pos;
  return IDG_OK;
}

int idg_rng_set_state(idg_rng* rng, const uint32_t key[624], int32_t pos) {
  IDG_REQUIRE(rng && key, "idg_rng_set_state: NULL argument");
  IDG_REQUIRE(pos >= 0 && pos <= 624, "idg_rng_set_state: pos %d outside [0,624]", pos);
  std::memcpy(rng->key, key, sizeof rng->key);
  rng->pos = pos;
  return IDG_OK;
}

int idg_rng_bytes(idg_rng* rng, int64_t nbytes, uint8_t* out) {
  IDG_REQUIRE(rng && (out || nbytes == 0), "idg_rng_bytes: NULL argument");
  IDG_REQUIRE(nbytes >= 0, "idg_rng_bytes: negative length");
  // RandomState.bytes draws ceil(n/4) uint32 and keeps the first n little-endian bytes.
  int64_t i = 0;
  while (i < nbytes) {
    uint32_t v = rng->next32();
    for (int b = 0; b < 4 && i < nbytes; ++b, ++i) out[i] = (uint8_t)(v >> (8 * b));
  }
  return IDG_OK;
}

int idg_rng_randint(idg_rng* rng, int64_t high, int64_t count, int64_t* out) {
  IDG_REQUIRE(rng && (out || count == 0), "idg_rng_randint: NULL argument");
  IDG_REQUIRE(high > 0, "idg_rng_randint: high must be > 0 (got %lld)", (long long)high);
  IDG_REQUIRE(count >= 0, "idg_rng_randint: negative count");
  for (int64_t i = 0; i < count; ++i) out[i] = (int64_t)rng->bounded((uint64_t)high - 1);
  return IDG_OK;
}

int idg_sample_epoch(idg_rng* rng, const int64_t* train_user, const int64_t* train_item, int64_t E,
                     const int64_t* pos_indptr, const int32_t* pos_indices, int64_t num_users,
                     int64_t num_items, int64_t* out_triples, int64_t* out_count) {
  IDG_REQUIRE(rng && pos_indptr && out_count, "idg_sample_epoch: NULL argument");
  IDG_REQUIRE(E >= 0 && num_users >= 0, "idg_sample_epoch: negative size");
  IDG_REQUIRE(E == 0 || (train_user && train_item && out_triples), "idg_sample_epoch: NULL edge array");
  IDG_REQUIRE(num_items > 0 || E == 0, "idg_sample_epoch: num_items must be > 0");
  int64_t w = 0;
  for (int64_t i = 0; i < E; ++i) {
    const int64_t u = train_user[i];
    IDG_REQUIRE(u >= 0 && u < num_users, "idg_sample_epoch: train_user[%lld]=%lld outside [0,%lld)",
                (long long)i, (long long)u, (long long)num_users);
    const int32_t* pb = pos_indices + pos_indptr[u];
    const int32_t* pe = pos_indices + pos_indptr[u + 1];
    if (pb == pe) continue;  // data_loader.py:114-115
    IDG_REQUIRE(pe - pb < num_items,
                "idg_sample_epoch: user %lld interacted with every item; no negative exists",
                (long long)u);
    int64_t neg;
    do {
      neg = (int64_t)rng->bounded((uint64_t)num_items - 1);
    } while (std::binary_search(pb, pe, (int32_t)neg));
    out_triples[3 * w + 0] = u;
    out_triples[3 * w + 1] = train_item[i];
    out_triples[3 * w + 2] = neg;
    ++w;
  }
  *out_count = w;
  return IDG_OK;
}

// Python's random.sample(range(n), k) (Lib/random.py, CPython 3.10+) on the same MT19937 stream the `random`
// module owns (tools.create_adj_mat, utility/utility_function/tools.py:80, draws SGL's kept edges with it):
//   _randbelow(m): b = m.bit_length(); r = getrandbits(b) until r < m;  getrandbits(b <= 32) = genrand_uint32() >> (32 - b)
//   use_pool (n <= setsize, decided by the caller with random.py's own expression): partial Fisher-Yates over a pool;
//   otherwise: draw until unseen, membership kept in a bitmap.
int idg_py_random_sample(idg_rng* rng, int64_t n, int64_t k, int use_pool, int64_t* out) {
  IDG_REQUIRE(rng && (out || k == 0), "idg_py_random_sample: NULL argument");
  IDG_REQUIRE(n >= 0 && k >= 0 && k <= n, "idg_py_random_sample: need 0 <= k <= n (n=%lld k=%lld)", (long long)n, (long long)k);
  IDG_REQUIRE(n < ((int64_t)1 << 32), "idg_py_random_sample: populations of 2^32 or more are not supported");
  auto bit_length = [](uint64_t m) {
    int b = 0;
    while (m) ++b, m >>= 1;
    return b;
  };
  auto randbelow = [&](uint64_t m) -> uint64_t {  // m >= 1
    const int b = bit_length(m);
    uint64_t r;
    do r = (uint64_t)(rng->next32() >> (32 - b));
    while (r >= m);
    return r;
  };
  if (use_pool) {
    std::vector<int64_t> pool((size_t)n);
    for (int64_t i = 0; i < n; ++i) pool[(size_t)i] = i;
    for (int64_t i = 0; i < k; ++i) {
      const uint64_t j = randbelow((uint64_t)(n - i));
      out[i] = pool[(size_t)j];
      pool[(size_t)j] = pool[(size_t)(n - i - 1)];
    }
  } else {
    std::vector<uint64_t> seen((size_t)((n + 63) / 64), 0);
    for (int64_t i = 0; i < k; ++i) {
      uint64_t j = randbelow((uint64_t)n);
      while (seen[(size_t)(j >> 6)] >> (j & 63) & 1) j = randbelow((uint64_t)n);
      seen[(size_t)(j >> 6)] |= (uint64_t)1 << (j & 63);
      out[i] = (int64_t)j;
    }
  }
  return IDG_OK;
}

int idg_shuffle_perm(idg_rng* rng, int64_t n, int64_t* out_perm) {
  IDG_REQUIRE(rng && (out_perm || n == 0), "idg_shuffle_perm: NULL argument");
  IDG_REQUIRE(n >= 0, "idg_shuffle_perm: negative length");
  for (int64_t i = 0; i < n; ++i) out_perm[i] = i;
  // Fisher-Yates from the top, j drawn on [0, i] — RandomState.shuffle on a 1-d array.
  for (int64_t i = n - 1; i >= 1; --i) {
    int64_t j = (int64_t)rng->bounded((uint64_t)i);
    std::swap(out_perm[i], out_perm[j]);
  }
  return IDG_OK;
}

}  // extern "C"

// ---------------------------------------------------------------------------------------
// Rating file parser
// ---------------------------------------------------------------------------------------
struct idg_ratings {
  std::vector<int64_t> users, items, line_users, line_counts;
};

extern "C" {

int idg_ratings_open(const char* path, idg_ratings** out, int64_t* n_edges, int64_t* n_lines,
                     int64_t* max_user, int64_t* max_item) {
  IDG_REQUIRE(path && out && n_edges && n_lines && max_user && max_item, "idg_ratings_open: NULL argument");
  FILE* f = std::fopen(path, "rb");
  if (!f) return idg::fail(IDG_E_IO, "idg_ratings_open: cannot open %s: %s", path, std::strerror(errno));
  std::string buf;
  {
    char chunk[1 << 16];
    size_t got;
    while ((got = std::fread(chunk, 1, sizeof chunk, f)) > 0) buf.append(chunk, got);
    std::fclose(f);
  }
  idg_ratings* r = new (std::nothrow) idg_ratings;
  if (!r) return idg::fail(IDG_E_NOMEM, "idg_ratings_open: out of memory");
  int64_t mu = -1, mi = -1;
  const char* p = buf.data();
  const char* end = p + buf.size();
  int64_t line_no = 0;
  while (p < end) {
    const char* eol = (const char*)std::memchr(p, '\n', (size_t)(end - p));
    if (!eol) eol = end;
    ++line_no;
    // tokens are separated by single spaces in the reference; accept runs of blanks/CR.
    const char* q = p;
    bool have_user = false;
    int64_t user = 0, n_items = 0, line_max = -1;
    const size_t first_item = r->items.size();
    while (q < eol) {
      while (q < eol && (*q == ' ' || *q == '\t' || *q == '\r')) ++q;
      if (q >= eol) break;
      bool negative = false;
      if (*q == '-' || *q == '+') negative = (*q++ == '-');
      if (q >= eol || *q < '0' || *q > '9') {
        delete r;
        return idg::fail(IDG_E_IO, "idg_ratings_open: %s line %lld: not an integer", path, (long long)line_no);
      }
      int64_t v = 0;
      while (q < eol && *q >= '0' && *q <= '9') {
        const int digit = *q++ - '0';
        if (v > (INT64_MAX - digit) / 10) {  // (Python's int() would take it; no id of a data set is this large: refuse, do not wrap)
          delete r;
          return idg::fail(IDG_E_IO, "idg_ratings_open: %s line %lld: number does not fit 64 bits", path, (long long)line_no);
        }
        v = v * 10 + digit;
      }
      if (negative) v = -v;
      if (!have_user) {
        user = v;
        have_user = true;
      } else {
        r->items.push_back(v);
        line_max = std::max(line_max, v);
        ++n_items;
      }
    }
    if (!have_user) {
      // int('') raises in the reference (data_loader.py:56); a blank line is malformed.
      delete r;
      return idg::fail(IDG_E_IO, "idg_ratings_open: %s line %lld is empty", path, (long long)line_no);
    }
    r->line_users.push_back(user);
    r->line_counts.push_back(n_items);
    if (n_items > 0) {
      r->users.insert(r->users.end(), (size_t)n_items, user);
      mu = std::max(mu, user);
      mi = std::max(mi, line_max);
    }
    (void)first_item;
    p = eol + 1;
  }
  *out = r;
  *n_edges = (int64_t)r->users.size();
  *n_lines = (int64_t)r->line_users.size();
  *max_user = mu;
  *max_item = mi;
  return IDG_OK;
}

int idg_ratings_read(const idg_ratings* r, int64_t* users, int64_t* items, int64_t* line_users,
                     int64_t* line_counts) {
  IDG_REQUIRE(r, "idg_ratings_read: NULL handle");
  if (users && !r->users.empty()) std::memcpy(users, r->users.data(), r->users.size() * sizeof(int64_t));
  if (items && !r->items.empty()) std::memcpy(items, r->items.data(), r->items.size() * sizeof(int64_t));
  if (line_users && !r->line_users.empty())
    std::memcpy(line_users, r->line_users.data(), r->line_users.size() * sizeof(int64_t));
  if (line_counts && !r->line_counts.empty())
    std::memcpy(line_counts, r->line_counts.data(), r->line_counts.size() * sizeof(int64_t));
  return IDG_OK;
}

int idg_ratings_destroy(idg_ratings* r) {
  delete r;
  return IDG_OK;
}

// ---------------------------------------------------------------------------------------
// Normalised bipartite adjacency
// ---------------------------------------------------------------------------------------
int idg_build_norm_adj(int64_t U, int64_t I, int64_t E, const int64_t* users, const int64_t* items,
                       int self_loops, const double* dinv, int64_t* nnz_out, int64_t* indptr,
                       int32_t* indices, float* values) {
  IDG_REQUIRE(U >= 0 && I >= 0 && E >= 0 && nnz_out, "idg_build_norm_adj: bad size / NULL nnz");
  IDG_REQUIRE(E == 0 || (users && items), "idg_build_norm_adj: NULL edge array");
  const int64_t n = U + I;
  IDG_REQUIRE(n < (int64_t)1 << 31, "idg_build_norm_adj: %lld nodes exceed int32 column ids", (long long)n);
  for (int64_t e = 0; e < E; ++e) {
    IDG_REQUIRE(users[e] >= 0 && users[e] < U, "idg_build_norm_adj: user id %lld outside [0,%lld)",
                (long long)users[e], (long long)U);
    IDG_REQUIRE(items[e] >= 0 && items[e] < I, "idg_build_norm_adj: item id %lld outside [0,%lld)",
                (long long)items[e], (long long)I);
  }
  // R as CSR with sorted columns and duplicate (u,i) pairs summed (data_loader.py:42-43).
  std::vector<int64_t> uptr((size_t)U + 1, 0);
  for (int64_t e = 0; e < E; ++e) ++uptr[(size_t)users[e] + 1];
  for (int64_t u = 0; u < U; ++u) uptr[(size_t)u + 1] += uptr[(size_t)u];
  std::vector<int32_t> ucol((size_t)E);
  {
    std::vector<int64_t> cur(uptr.begin(), uptr.end() - 1);
    for (int64_t e = 0; e < E; ++e) ucol[(size_t)cur[(size_t)users[e]]++] = (int32_t)items[e];
  }
  std::vector<int64_t> rptr((size_t)U + 1, 0);  // after de-duplication
  std::vector<int32_t> rcol;
  std::vector<float> rval;  // multiplicity, float32 like the LIL the reference assigns into
  rcol.reserve((size_t)E);
  rval.reserve((size_t)E);
  for (int64_t u = 0; u < U; ++u) {
    int32_t* b = ucol.data() + uptr[(size_t)u];
    int32_t* e = ucol.data() + uptr[(size_t)u + 1];
    std::sort(b, e);
    for (int32_t* p = b; p < e;) {
      int32_t* q = p;
      while (q < e && *q == *p) ++q;
      rcol.push_back(*p);
      rval.push_back((float)(q - p));
      p = q;
    }
    rptr[(size_t)u + 1] = (int64_t)rcol.size();
  }
  const int64_t nnzR = (int64_t)rcol.size();
  const int64_t nnz = 2 * nnzR + (self_loops ? n : 0);
  *nnz_out = nnz;
  if (!indptr) return IDG_OK;  // size query
  IDG_REQUIRE(indices && values, "idg_build_norm_adj: NULL output array");

  // Degrees = row sums of A (+I).  Sums of small integers: exact in float32 below 2^24.
  std::vector<double> deg((size_t)n, self_loops ? 1.0 : 0.0);
  for (int64_t u = 0; u < U; ++u)
    for (int64_t k = rptr[(size_t)u]; k < rptr[(size_t)u + 1]; ++k) {
      deg[(size_t)u] += rval[(size_t)k];
      deg[(size_t)(U + rcol[(size_t)k])] += rval[(size_t)k];
    }
  // d^-1/2.  Without self loops the reference stays in float32 end to end
  // (data_graph.py:46-48: np.power on a float32 row sum).  With self loops the `+ sp.eye`
  // (data_graph.py:20) promotes everything to float64 and only the final tensor
  // conversion (tools.py:101) rounds to float32.  NumPy's power kernels are SIMD routines
  // that are not correctly rounded, so a caller that wants the reference's exact bits
  // passes the np.power result in `dinv`; otherwise the correctly rounded value is used.
  std::vector<float> dinv32;
  std::vector<double> dinv64;
  if (!self_loops) {
    dinv32.resize((size_t)n);
    for (int64_t i = 0; i < n; ++i) {
      if (dinv) dinv32[(size_t)i] = (float)dinv[i];
      else dinv32[(size_t)i] = deg[(size_t)i] > 0 ? (float)(1.0 / std::sqrt(deg[(size_t)i])) : 0.0f;
    }
  } else {
    dinv64.resize((size_t)n);
    for (int64_t i = 0; i < n; ++i) dinv64[(size_t)i] = dinv ? dinv[i] : 1.0 / std::sqrt(deg[(size_t)i]);
  }
  auto norm = [&](int64_t i, int64_t j, float a) -> float {
    if (!self_loops) {
      // (D.A).D evaluated left to right in float32 (data_graph.py:51)
      float t = dinv32[(size_t)i] * a;
      return t * dinv32[(size_t)j];
    }
    double t = dinv64[(size_t)i] * (double)a;
    return (float)(t * dinv64[(size_t)j]);
  };

  // Item-side transpose structure: count then fill (ascending users per item by construction).
  std::vector<int64_t> iptr((size_t)I + 1, 0);
  for (int64_t k = 0; k < nnzR; ++k) ++iptr[(size_t)rcol[(size_t)k] + 1];
  for (int64_t i = 0; i < I; ++i) iptr[(size_t)i + 1] += iptr[(size_t)i];

  // Row pointers of the full matrix.
  indptr[0] = 0;
  for (int64_t u = 0; u < U; ++u)
    indptr[u + 1] = indptr[u] + (rptr[(size_t)u + 1] - rptr[(size_t)u]) + (self_loops ? 1 : 0);
  for (int64_t i = 0; i < I; ++i)
    indptr[U + i + 1] = indptr[U + i] + (iptr[(size_t)i + 1] - iptr[(size_t)i]) + (self_loops ? 1 : 0);

  // User rows: [self] then item columns (all > u since they are offset by U).
  for (int64_t u = 0; u < U; ++u) {
    int64_t w = indptr[u];
    if (self_loops) {
      indices[w] = (int32_t)u;
      values[w] = norm(u, u, 1.0f);
      ++w;
    }
    for (int64_t k = rptr[(size_t)u]; k < rptr[(size_t)u + 1]; ++k, ++w) {
      indices[w] = (int32_t)(U + rcol[(size_t)k]);
      values[w] = norm(u, U + rcol[(size_t)k], rval[(size_t)k]);
    }
  }
  // Item rows: user columns ascending, then [self].
  {
    std::vector<int64_t> cur((size_t)I);
    for (int64_t i = 0; i < I; ++i) cur[(size_t)i] = indptr[U + i];
    for (int64_t u = 0; u < U; ++u)
      for (int64_t k = rptr[(size_t)u]; k < rptr[(size_t)u + 1]; ++k) {
        const int64_t i = rcol[(size_t)k];
        const int64_t w = cur[(size_t)i]++;
        indices[w] = (int32_t)u;
        values[w] = norm(U + i, u, rval[(size_t)k]);
      }
    if (self_loops)
      for (int64_t i = 0; i < I; ++i) {
        const int64_t w = cur[(size_t)i];
        indices[w] = (int32_t)(U + i);
        values[w] = norm(U + i, U + i, 1.0f);
      }
  }
  return IDG_OK;
}

}  // extern "C"
