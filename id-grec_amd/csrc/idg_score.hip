// Full-rank scoring (fp32 MFMA GEMM + sigmoid), train-positive masking and top-K.
//
// Scoring is the one GEMM-shaped op on the path: rating = act(U[users] . V^T), d = 64..256.
// It runs on v_mfma_f32_32x32x2_f32 (exact fp32 products and accumulation).  Each wave owns
// a 32-user x 32-item accumulator tile; lane (i, h) feeds user i's features
// [kc+32h, kc+32h+32) as the A operand and item i's same feature range as the B operand,
// so every lane reads one contiguous 128-byte run per 64-deep k chunk and nothing is
// transposed or staged through LDS.
//
// Top-K is a wave-level streaming select over one user's scores: a 64-bit key
// (order-preserving score bits << 32 | ~item) makes "larger key" mean "better score, then
// lower item id", candidates above the running k-th key are appended to a small LDS buffer
// with ballot compaction, and every 64 candidates a 128-key bitonic network (shuffles only)
// folds them into the sorted best-64 held one per lane.
#include <hip/hip_runtime.h>

#include <cstdlib>

#include "idg_common.h"

namespace {

constexpr int WAVE = 64;
constexpr int BLOCK = 256;
constexpr int ITEMS_PER_WAVE = 256;  // 8 MFMA tiles per wave per launch row

using f32x16 = __attribute__((ext_vector_type(16))) float;

__device__ __forceinline__ float sigmoidf_(float x) { return 1.0f / (1.0f + expf(-x)); }

// grid: (ceil(I / (4*ITEMS_PER_WAVE)), ceil(Bt / 32))
template <bool SIGMOID>
__global__ __launch_bounds__(BLOCK) void score_dense_kernel(const float* __restrict__ U,
                                                            const float* __restrict__ V,
                                                            const int64_t* __restrict__ users, int64_t Bt,
                                                            int64_t I, int64_t d, float* __restrict__ rating,
                                                            int64_t ld) {
  const int lane = threadIdx.x % WAVE;
  const int wave = threadIdx.x / WAVE;
  const int i = lane & 31;
  const int h = lane >> 5;
  const int64_t b0 = (int64_t)blockIdx.y * 32;
  const int64_t bu = b0 + i < Bt ? b0 + i : Bt - 1;
  const float* urow = U + users[bu] * d;
  const int64_t j_begin = ((int64_t)blockIdx.x * (BLOCK / WAVE) + wave) * ITEMS_PER_WAVE;
  const bool d4 = (d % 4 == 0);

  for (int64_t j0 = j_begin; j0 < j_begin + ITEMS_PER_WAVE && j0 < I; j0 += 32) {
    const int64_t jv = j0 + i < I ? j0 + i : I - 1;
    const float* vrow = V + jv * d;
    f32x16 acc;
#pragma unroll
    for (int r = 0; r < 16; ++r) acc[r] = 0.f;
    for (int64_t kc = 0; kc < d; kc += 64) {
      const int64_t k0 = kc + 32 * h;
      float a[32], b[32];
      if (d4 && k0 + 32 <= d) {
#pragma unroll
        for (int q = 0; q < 8; ++q) {
          const float4 x = *reinterpret_cast<const float4*>(urow + k0 + 4 * q);
          const float4 y = *reinterpret_cast<const float4*>(vrow + k0 + 4 * q);
          a[4 * q + 0] = x.x, a[4 * q + 1] = x.y, a[4 * q + 2] = x.z, a[4 * q + 3] = x.w;
          b[4 * q + 0] = y.x, b[4 * q + 1] = y.y, b[4 * q + 2] = y.z, b[4 * q + 3] = y.w;
        }
      } else {
#pragma unroll
        for (int s = 0; s < 32; ++s) {
          const bool in = k0 + s < d;
          a[s] = in ? urow[k0 + s] : 0.f;
          b[s] = in ? vrow[k0 + s] : 0.f;
        }
      }
#pragma unroll
      for (int s = 0; s < 32; ++s) acc = __builtin_amdgcn_mfma_f32_32x32x2f32(a[s], b[s], acc, 0, 0, 0);
    }
    // C/D map: column (item) = lane & 31, row (user) = (r & 3) + 8 * (r >> 2) + 4 * (lane >> 5)
    if (j0 + i < I) {
#pragma unroll
      for (int r = 0; r < 16; ++r) {
        const int64_t b = b0 + (r & 3) + 8 * (r >> 2) + 4 * h;
        if (b < Bt) {
          float s = acc[r];
          if (SIGMOID) s = sigmoidf_(s);
          rating[b * ld + j0 + i] = s;
        }
      }
    }
  }
}

// ---- 64-bit keys ------------------------------------------------------------------------
__device__ __forceinline__ unsigned long long make_key(float s, uint32_t item) {
  uint32_t u = __float_as_uint(s);
  u = (u & 0x80000000u) ? ~u : (u | 0x80000000u);
  return ((unsigned long long)u << 32) | (uint32_t)(~item);
}
__device__ __forceinline__ float key_score(unsigned long long k) {
  uint32_t u = (uint32_t)(k >> 32);
  u = (u & 0x80000000u) ? (u & 0x7fffffffu) : ~u;
  return __uint_as_float(u);
}
__device__ __forceinline__ uint32_t key_item(unsigned long long k) { return ~(uint32_t)k; }

__device__ __forceinline__ unsigned long long shfl_xor_u64(unsigned long long v, int m) {
  uint32_t lo = (uint32_t)v, hi = (uint32_t)(v >> 32);
  lo = __shfl_xor(lo, m, WAVE);
  hi = __shfl_xor(hi, m, WAVE);
  return ((unsigned long long)hi << 32) | lo;
}
__device__ __forceinline__ unsigned long long shfl_u64(unsigned long long v, int src) {
  uint32_t lo = (uint32_t)v, hi = (uint32_t)(v >> 32);
  lo = __shfl(lo, src, WAVE);
  hi = __shfl(hi, src, WAVE);
  return ((unsigned long long)hi << 32) | lo;
}

// Bitonic sort of the 128 keys {k0 (element = lane), k1 (element = 64 + lane)}, descending.
__device__ __forceinline__ void wave_sort128_desc(unsigned long long& k0, unsigned long long& k1, int lane) {
#pragma unroll
  for (int k = 2; k <= 128; k <<= 1) {
#pragma unroll
    for (int j = k >> 1; j > 0; j >>= 1) {
      if (j == 64) {
        // partner is the other register of this lane; (e & 128) == 0 always: descending
        const unsigned long long hi = k0 > k1 ? k0 : k1, lo = k0 > k1 ? k1 : k0;
        k0 = hi;
        k1 = lo;
      } else {
        const bool lower = (lane & j) == 0;  // this element has the smaller index of the pair
        {
          const bool desc = ((lane & k) == 0);  // element index = lane
          const unsigned long long o = shfl_xor_u64(k0, j);
          const bool take_max = (lower == desc);
          k0 = take_max ? (k0 > o ? k0 : o) : (k0 > o ? o : k0);
        }
        {
          const bool desc = (((64 + lane) & k) == 0);  // element index = 64 + lane
          const unsigned long long o = shfl_xor_u64(k1, j);
          const bool take_max = (lower == desc);
          k1 = take_max ? (k1 > o ? k1 : o) : (k1 > o ? o : k1);
        }
      }
    }
  }
}

// One wave per batch row, streaming `n_items` scores of the row (item ids item_lo + 0..n_items-1).
// 16-byte loads, the next 256 scores prefetched while the current ones are filtered.  A row's
// running best-64 can be carried across launches through `state` (item-chunked evaluation keeps
// the score scratch bounded for any catalogue size).
__global__ __launch_bounds__(BLOCK) void topk_rows_kernel(const float* __restrict__ rating, int64_t ld, int64_t Bt,
                                                          int64_t item_lo, int64_t n_items,
                                                          const int64_t* __restrict__ users,
                                                          const int64_t* __restrict__ excl_indptr,
                                                          const int32_t* __restrict__ excl_items, int k,
                                                          unsigned long long* __restrict__ state, int load_state,
                                                          int final_pass, int64_t* __restrict__ out_idx,
                                                          float* __restrict__ out_val) {
  __shared__ unsigned long long s_pend[BLOCK / WAVE][128];
  const int wave = threadIdx.x / WAVE;
  const int lane = threadIdx.x % WAVE;
  const int64_t b = (int64_t)blockIdx.x * (BLOCK / WAVE) + wave;
  if (b >= Bt) return;
  unsigned long long* pend = s_pend[wave];
  const float* row = rating + b * ld;
  const int32_t* ex_b = nullptr;
  int ex_n = 0;
  if (excl_indptr) {
    const int64_t u = users[b];
    ex_b = excl_items + excl_indptr[u];
    ex_n = (int)(excl_indptr[u + 1] - excl_indptr[u]);
  }
  unsigned long long best = load_state ? state[b * WAVE + lane] : 0ull;  // sorted descending across lanes; 0 = empty
  unsigned long long tau = shfl_u64(best, k - 1);
  int n_pend = 0;

  auto flush = [&](int take) {
    __builtin_amdgcn_wave_barrier();
    unsigned long long a = lane < take ? pend[n_pend - take + lane] : 0ull;
    n_pend -= take;
    wave_sort128_desc(best, a, lane);
    tau = shfl_u64(best, k - 1);
  };
  auto offer = [&](float sc, int64_t local, bool valid) {
    bool pass = false;
    unsigned long long key = 0;
    if (valid) {
      const uint32_t item = (uint32_t)(item_lo + local);
      key = make_key(sc, item);
      pass = key > tau;
      if (pass && ex_n > 0) {
        int lo = 0, hi = ex_n;
        while (lo < hi) {
          const int mid = (lo + hi) >> 1;
          if (ex_b[mid] < (int32_t)item) lo = mid + 1;
          else hi = mid;
        }
        if (lo < ex_n && ex_b[lo] == (int32_t)item) {
          key = make_key(-1.0f, item);  // batch_test.py:65
          pass = key > tau;
        }
      }
    }
    const unsigned long long m = __ballot(pass);
    if (m) {
      if (pass) pend[n_pend + __popcll(m & ((1ull << lane) - 1ull))] = key;
      n_pend += __popcll(m);
      if (n_pend >= 64) flush(64);
    }
  };

  const bool vec = ((ld % 4) == 0) && (((uintptr_t)rating % 16) == 0);
  if (vec) {
    const int64_t n4 = n_items / 4;  // whole float4 groups
    float4 cur = make_float4(0.f, 0.f, 0.f, 0.f), nxt = cur;
    if (lane < n4) cur = reinterpret_cast<const float4*>(row)[lane];
    for (int64_t g0 = 0; g0 < n4; g0 += WAVE) {
      const int64_t gi = g0 + lane;
      if (gi + WAVE < n4) nxt = reinterpret_cast<const float4*>(row)[gi + WAVE];
      const bool valid = gi < n4;
      offer(cur.x, gi * 4 + 0, valid);
      offer(cur.y, gi * 4 + 1, valid);
      offer(cur.z, gi * 4 + 2, valid);
      offer(cur.w, gi * 4 + 3, valid);
      cur = nxt;
    }
    const int64_t tail = n4 * 4 + lane;
    offer(tail < n_items ? row[tail] : 0.f, tail, tail < n_items);
  } else {
    for (int64_t base = 0; base < n_items; base += WAVE) {
      const int64_t local = base + lane;
      offer(local < n_items ? row[local] : 0.f, local, local < n_items);
    }
  }
  if (n_pend > 0) flush(n_pend);
  if (!final_pass) {
    state[b * WAVE + lane] = best;
    return;
  }
  if (lane < k) {
    out_idx[b * k + lane] = (int64_t)key_item(best);
    if (out_val) out_val[b * k + lane] = key_score(best);
  }
}

}  // namespace

extern "C" {

static int launch_dense(const float* user_panel, const float* item_panel, const int64_t* users, int64_t Bt, int64_t n_items,
                        int64_t d, int apply_sigmoid, float* rating, int64_t ld, hipStream_t st) {
  const dim3 grid((unsigned)((n_items + (BLOCK / WAVE) * ITEMS_PER_WAVE - 1) / ((BLOCK / WAVE) * ITEMS_PER_WAVE)),
                  (unsigned)((Bt + 31) / 32));
  if (apply_sigmoid)
    hipLaunchKernelGGL(score_dense_kernel<true>, grid, dim3(BLOCK), 0, st, user_panel, item_panel, users, Bt, n_items, d,
                       rating, ld);
  else
    hipLaunchKernelGGL(score_dense_kernel<false>, grid, dim3(BLOCK), 0, st, user_panel, item_panel, users, Bt, n_items, d,
                       rating, ld);
  IDG_HIP(hipGetLastError());
  return IDG_OK;
}

int idg_score_dense_f32(const float* user_panel, const float* item_panel, const int64_t* users, int64_t Bt,
                        int64_t I, int64_t d, int apply_sigmoid, float* rating, void* stream) {
  IDG_REQUIRE(user_panel && item_panel && users && rating, "idg_score_dense_f32: NULL argument");
  IDG_REQUIRE(Bt > 0 && I > 0 && d > 0, "idg_score_dense_f32: bad sizes");
  return launch_dense(user_panel, item_panel, users, Bt, I, d, apply_sigmoid, rating, I, (hipStream_t)stream);
}

// Items are processed in windows of at most this many, so the score scratch is Bt x min(I, window)
// floats however large the catalogue is; a row's running best-64 is carried between windows.
static inline int64_t item_window() {
  if (const char* v = std::getenv("IDG_ITEM_WINDOW")) {  // testing knob: exercise the windowed path on small catalogues
    const long long w = std::atoll(v);
    if (w >= 32) return (int64_t)w / 4 * 4;
  }
  return (int64_t)1 << 16;
}

static inline size_t topk_scratch_floats(int64_t Bt, int64_t I) {
  const int64_t ITEM_WINDOW = item_window();
  const int64_t w = I < ITEM_WINDOW ? I : ITEM_WINDOW;
  return (size_t)Bt * (size_t)((w + 3) / 4 * 4);
}

size_t idg_score_topk_workspace_bytes(int64_t Bt, int64_t I, int64_t d, int k) {
  (void)d;
  (void)k;
  if (Bt <= 0 || I <= 0) return 0;
  const size_t scores = (topk_scratch_floats(Bt, I) * sizeof(float) + 255) / 256 * 256;
  return scores + (size_t)Bt * WAVE * sizeof(unsigned long long);
}

int idg_score_topk_f32(const float* user_panel, const float* item_panel, const int64_t* users, int64_t Bt,
                       int64_t I, int64_t d, const int64_t* excl_indptr, const int32_t* excl_items, int k,
                       int apply_sigmoid, int64_t* out_idx, float* out_val, void* ws, void* stream) {
  IDG_REQUIRE(user_panel && item_panel && users && out_idx && ws, "idg_score_topk_f32: NULL argument");
  IDG_REQUIRE(Bt > 0 && I > 0 && d > 0, "idg_score_topk_f32: bad sizes");
  IDG_REQUIRE(k >= 1 && k <= 64, "idg_score_topk_f32: k=%d outside [1,64]", k);
  IDG_REQUIRE(k <= I, "idg_score_topk_f32: k=%d exceeds the item count %lld", k, (long long)I);
  IDG_REQUIRE(I < ((int64_t)1 << 32), "idg_score_topk_f32: more than 2^32 items");
  IDG_REQUIRE((excl_indptr == nullptr) == (excl_items == nullptr), "idg_score_topk_f32: excl_indptr and excl_items go together");
  hipStream_t st = (hipStream_t)stream;
  float* rating = reinterpret_cast<float*>(ws);
  const size_t scores = (topk_scratch_floats(Bt, I) * sizeof(float) + 255) / 256 * 256;
  unsigned long long* state = reinterpret_cast<unsigned long long*>(reinterpret_cast<char*>(ws) + scores);
  const unsigned nb = (unsigned)((Bt + (BLOCK / WAVE) - 1) / (BLOCK / WAVE));
  const int64_t ITEM_WINDOW = item_window();
  for (int64_t lo = 0; lo < I; lo += ITEM_WINDOW) {
    const int64_t n_items = I - lo < ITEM_WINDOW ? I - lo : ITEM_WINDOW;
    const int64_t ld = (n_items + 3) / 4 * 4;  // rows start 16-byte aligned
    int rc = launch_dense(user_panel, item_panel + lo * d, users, Bt, n_items, d, apply_sigmoid, rating, ld, st);
    if (rc != IDG_OK) return rc;
    hipLaunchKernelGGL(topk_rows_kernel, dim3(nb), dim3(BLOCK), 0, st, rating, ld, Bt, lo, n_items, users, excl_indptr,
                       excl_items, k, state, lo > 0 ? 1 : 0, lo + n_items >= I ? 1 : 0, out_idx, out_val);
  }
  IDG_HIP(hipGetLastError());
  return IDG_OK;
}

}  // extern "C"
