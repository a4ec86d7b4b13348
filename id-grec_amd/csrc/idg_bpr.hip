// Fused gather + BPR + L2-regulariser loss and gradients, and the dense Adam step.
//
// One wavefront per (user, pos, neg) triple: six embedding rows are gathered with
// coalesced loads, the two dot products are reduced with wave shuffles, the loss term and
// d loss / d score are formed in registers, and the three gradient rows are scattered.
// Duplicate rows inside a batch are the norm (popular items), so the scatter has two
// forms: float atomics (fast, sum order varies run to run) and a deterministic one — the
// 3B (row, slot) pairs are sorted and one lane group per distinct row adds its
// contributions in batch order.
#include <hip/hip_runtime.h>

#include <cmath>
#include <cstdlib>

#include "idg_common.h"

namespace {

constexpr int WAVE = 64;
constexpr int BLOCK = 256;
constexpr int LDS_SORT_MAX = 8192;  // (key,slot) pairs one workgroup sorts in LDS
// IDG_SORT_SINGLE_BLOCK=1: lists of 1025..8192 pairs sorted by ONE workgroup as before round 5 (A/B runs; same result)
static const bool g_single_block_sort = [] {
  const char* e = std::getenv("IDG_SORT_SINGLE_BLOCK");
  return e && e[0] == '1';
}();

__device__ __forceinline__ float wave_sum(float v) {
#pragma unroll
  for (int o = 32; o > 0; o >>= 1) v += __shfl_xor(v, o, WAVE);
  return v;
}

struct BprArgs {
  const float* fin;
  const float* ego;
  const int64_t* users;
  const int64_t* pos;
  const int64_t* neg;
  int64_t num_users, B, d;
  int64_t de;              // width of the ego rows (== d except for idg_bpr_fused_ex_f32: NGCF's 4d-wide final rows)
  int reg_users;           // 0: the regulariser covers the two item blocks only (models/NGCF.py:125, EGCF.py:95-96)
  float inv_B, reg_scale;  // reg_scale = reg_lambda / B
  float* coef;             // [B]   d loss0 / d x_i (already / B)
  float* loss_i;           // [B]
  float* sq;               // [3,B] squared norms of the ego rows
  int32_t* keys;           // [3B]  destination row per slot (deterministic form)
  int32_t* slots;          // [3B]
  float* g_final;
  float* g_ego;
  const float* upstream;  // device [2]: d total / d loss[0], d total / d loss[1]; NULL = ones
  uint32_t* touched;      // bitmap of g_final rows written; when set, rows are STORED, not accumulated
  int touched_preset;     // the bitmap already holds exactly these rows (idg_bpr_touch_rows): store mode, nothing written to it
  int atomic;
};

__global__ __launch_bounds__(BLOCK) void bpr_triple_kernel(BprArgs a) {
  const int64_t i = (int64_t)blockIdx.x * (BLOCK / WAVE) + threadIdx.x / WAVE;
  const int lane = threadIdx.x % WAVE;
  if (i >= a.B) return;
  const int64_t ru = a.users[i];
  const int64_t rp = a.num_users + a.pos[i];
  const int64_t rn = a.num_users + a.neg[i];
  const float* fu = a.fin + ru * a.d;
  const float* fp = a.fin + rp * a.d;
  const float* fn = a.fin + rn * a.d;
  const float* eu = a.ego + ru * a.d;
  const float* epp = a.ego + rp * a.d;
  const float* en = a.ego + rn * a.d;
  float sp = 0.f, sn = 0.f, qu = 0.f, qp = 0.f, qn = 0.f;
  if (a.de == a.d && a.reg_users) {  // the LightGCN-family form: one pass over rows of one width
    for (int64_t f = lane; f < a.d; f += WAVE) {
      const float u = fu[f], p = fp[f], n = fn[f];
      sp = __builtin_fmaf(u, p, sp);
      sn = __builtin_fmaf(u, n, sn);
      const float x = eu[f], y = epp[f], z = en[f];
      qu = __builtin_fmaf(x, x, qu);
      qp = __builtin_fmaf(y, y, qp);
      qn = __builtin_fmaf(z, z, qn);
    }
  } else {
    for (int64_t f = lane; f < a.d; f += WAVE) {
      const float u = fu[f], p = fp[f], n = fn[f];
      sp = __builtin_fmaf(u, p, sp);
      sn = __builtin_fmaf(u, n, sn);
    }
    const float* eu2 = a.ego + ru * a.de;
    const float* ep2 = a.ego + rp * a.de;
    const float* en2 = a.ego + rn * a.de;
    for (int64_t f = lane; f < a.de; f += WAVE) {
      const float x = a.reg_users ? eu2[f] : 0.f, y = ep2[f], z = en2[f];
      qu = __builtin_fmaf(x, x, qu);
      qp = __builtin_fmaf(y, y, qp);
      qn = __builtin_fmaf(z, z, qn);
    }
  }
  sp = wave_sum(sp);
  sn = wave_sum(sn);
  qu = wave_sum(qu);
  qp = wave_sum(qp);
  qn = wave_sum(qn);
  const float x = sp - sn;
  const float sig = 1.0f / (1.0f + expf(-x));
  const float li = -logf(sig + 1e-7f);  // losses.py:11 (10e-8)
  const float c = -(sig * (1.0f - sig)) / (sig + 1e-7f) * a.inv_B;
  if (lane == 0) {
    a.loss_i[i] = li;
    a.coef[i] = c;
    a.sq[i] = qu;
    a.sq[a.B + i] = qp;
    a.sq[2 * a.B + i] = qn;
  }
}

// (row, slot) keys of the 3B gradient contributions of a batch: depends on the indices only.
__global__ __launch_bounds__(BLOCK) void bpr_keys_kernel(const int64_t* __restrict__ users, const int64_t* __restrict__ pos,
                                                         const int64_t* __restrict__ neg, int64_t B, int64_t num_users,
                                                         int32_t* __restrict__ keys, int32_t* __restrict__ slots,
                                                         uint32_t* __restrict__ bitmap, uint32_t* __restrict__ zero2) {
  const int64_t i = (int64_t)blockIdx.x * BLOCK + threadIdx.x;
  if (i >= B) return;
  if (zero2 && i == 0) zero2[0] = 0u, zero2[1] = 0u;  // (idg::bpr_plan_rows: the header of a unit list built right after)
  const int64_t r0 = users[i], r1 = num_users + pos[i], r2 = num_users + neg[i];
  keys[3 * i + 0] = (int32_t)r0;
  keys[3 * i + 1] = (int32_t)r1;
  keys[3 * i + 2] = (int32_t)r2;
  if (bitmap) {  // (idg_bpr_plan_rows_f32: the batch's row bitmap on the way — what bpr_touch_kernel sets)
    atomicOr(bitmap + (r0 >> 5), 1u << (r0 & 31));
    atomicOr(bitmap + (r1 >> 5), 1u << (r1 & 31));
    atomicOr(bitmap + (r2 >> 5), 1u << (r2 & 31));
  }
  slots[3 * i + 0] = (int32_t)(3 * i + 0);
  slots[3 * i + 1] = (int32_t)(3 * i + 1);
  slots[3 * i + 2] = (int32_t)(3 * i + 2);
}

// Bitmap of the <= 3B panel rows a batch reads and writes (users, num_users + pos, num_users + neg).
__global__ __launch_bounds__(BLOCK) void bpr_touch_kernel(const int64_t* __restrict__ users, const int64_t* __restrict__ pos,
                                                          const int64_t* __restrict__ neg, int64_t B, int64_t num_users,
                                                          uint32_t* __restrict__ bitmap) {
  const int64_t i = (int64_t)blockIdx.x * BLOCK + threadIdx.x;
  if (i >= B) return;
  const int64_t r0 = users[i], r1 = num_users + pos[i], r2 = num_users + neg[i];
  atomicOr(bitmap + (r0 >> 5), 1u << (r0 & 31));
  atomicOr(bitmap + (r1 >> 5), 1u << (r1 & 31));
  atomicOr(bitmap + (r2 >> 5), 1u << (r2 & 31));
}

// Backward, atomic form: one wave per triple, float atomics into the gradient rows.
__global__ __launch_bounds__(BLOCK) void bpr_atomic_kernel(BprArgs a) {
  const int64_t i = (int64_t)blockIdx.x * (BLOCK / WAVE) + threadIdx.x / WAVE;
  const int lane = threadIdx.x % WAVE;
  if (i >= a.B) return;
  const int64_t ru = a.users[i];
  const int64_t rp = a.num_users + a.pos[i];
  const int64_t rn = a.num_users + a.neg[i];
  const float up0 = a.upstream ? a.upstream[0] : 1.0f;
  const float up1 = a.upstream ? a.upstream[1] : 1.0f;
  const float c = a.coef[i] * up0;
  const float rs = a.reg_scale * up1;
  for (int64_t f = lane; f < a.d; f += WAVE) {
    if (a.g_final) {
      const float u = a.fin[ru * a.d + f], p = a.fin[rp * a.d + f], n = a.fin[rn * a.d + f];
      atomicAdd(a.g_final + ru * a.d + f, c * (p - n));
      atomicAdd(a.g_final + rp * a.d + f, c * u);
      atomicAdd(a.g_final + rn * a.d + f, -c * u);
    }
    if (a.g_ego) {
      atomicAdd(a.g_ego + ru * a.d + f, rs * a.ego[ru * a.d + f]);
      atomicAdd(a.g_ego + rp * a.d + f, rs * a.ego[rp * a.d + f]);
      atomicAdd(a.g_ego + rn * a.d + f, rs * a.ego[rn * a.d + f]);
    }
  }
}

// Sort <= LDS_SORT_MAX packed (row, slot) keys ascending, one 1024-thread workgroup.
// Element i = e * 1024 + tid lives in register e of thread tid.  Bitonic network; the partner of
// element i at distance j is i ^ j:  j >= 1024 -> another register of the same thread (no traffic),
// 64 <= j < 1024 -> another wave (one LDS exchange), j < 64 -> another lane (wave shuffle).
// Of the 78 phases of a 4096-key sort only 18 touch LDS.  KeyT is uint32 when row and slot bits
// fit in 32 (every reference configuration), else uint64.
template <typename KeyT>
__device__ __forceinline__ KeyT shfl_xor_key(KeyT v, int m);
template <>
__device__ __forceinline__ uint32_t shfl_xor_key<uint32_t>(uint32_t v, int m) {
  return __shfl_xor(v, m, WAVE);
}
template <>
__device__ __forceinline__ unsigned long long shfl_xor_key<unsigned long long>(unsigned long long v, int m) {
  uint32_t lo = (uint32_t)v, hi = (uint32_t)(v >> 32);
  lo = __shfl_xor(lo, m, WAVE);
  hi = __shfl_xor(hi, m, WAVE);
  return ((unsigned long long)hi << 32) | lo;
}

// Workgroup b sorts the keys [b * 1024 E, (b + 1) * 1024 E) of the list (one workgroup: the whole list).  packed_out != NULL:
// the sorted PACKED keys go there (a run for merge_runs_kernel) instead of being unpacked into keys_out / slots_out.
template <typename KeyT, int E>
__global__ __launch_bounds__(1024) void lds_sort_kernel(const int32_t* __restrict__ keys,
                                                        const int32_t* __restrict__ slots, int n, int p2,
                                                        int slot_bits, int32_t* __restrict__ keys_out,
                                                        int32_t* __restrict__ slots_out, KeyT* __restrict__ packed_out) {
  __shared__ KeyT s_k[1024 * E];
  const int tid = threadIdx.x;
  const int base = blockIdx.x * (1024 * E);
  keys += base, slots += base;
  n = n - base < 1024 * E ? n - base : 1024 * E;
  KeyT v[E];
#pragma unroll
  for (int e = 0; e < E; ++e) {
    const int i = e * 1024 + tid;
    v[e] = i < n ? (KeyT)(((KeyT)(uint32_t)keys[i] << slot_bits) | (KeyT)(uint32_t)slots[i]) : (KeyT)~(KeyT)0;
  }
  auto cmpx = [](KeyT a, KeyT o, bool take_min) { return take_min ? (a < o ? a : o) : (a < o ? o : a); };
  for (int k = 2; k <= p2; k <<= 1) {
    for (int j = k >> 1; j > 0; j >>= 1) {
      if (j >= 1024) {
        // partner register e ^ (j / 1024): static indexing needs the distance as a constant
#define IDG_INTHREAD(JE)                                                                  \
  if (E > JE) {                                                                           \
    _Pragma("unroll") for (int e = 0; e < E; ++e) if ((e & JE) == 0 && (e | JE) < E) {    \
      const int i = e * 1024 + tid;                                                       \
      const bool up = (i & k) == 0;                                                       \
      const KeyT a = v[e], b = v[e | JE];                                                 \
      const bool sw = (a > b) == up;                                                      \
      v[e] = sw ? b : a;                                                                  \
      v[e | JE] = sw ? a : b;                                                             \
    }                                                                                     \
  }
        const int je = j >> 10;
        if (je == 1) { IDG_INTHREAD(1) } else if (je == 2) { IDG_INTHREAD(2) } else if (je == 4) { IDG_INTHREAD(4) }
#undef IDG_INTHREAD
      } else if (j >= WAVE) {
#pragma unroll
        for (int e = 0; e < E; ++e) s_k[e * 1024 + tid] = v[e];
        __syncthreads();
#pragma unroll
        for (int e = 0; e < E; ++e) {
          const int i = e * 1024 + tid;
          const KeyT o = s_k[i ^ j];
          v[e] = cmpx(v[e], o, ((i & j) == 0) == ((i & k) == 0));
        }
        __syncthreads();
      } else {
#pragma unroll
        for (int e = 0; e < E; ++e) {
          const int i = e * 1024 + tid;
          const KeyT o = shfl_xor_key<KeyT>(v[e], j);
          v[e] = cmpx(v[e], o, ((i & j) == 0) == ((i & k) == 0));
        }
      }
    }
  }
  const KeyT slot_mask = (KeyT)(((KeyT)1 << slot_bits) - 1);
#pragma unroll
  for (int e = 0; e < E; ++e) {
    const int i = e * 1024 + tid;
    if (i < n) {
      if (packed_out) {
        packed_out[base + i] = v[e];
      } else {
        keys_out[base + i] = (int32_t)(v[e] >> slot_bits);
        slots_out[base + i] = (int32_t)(v[e] & slot_mask);
      }
    }
  }
}

// One merge pass over sorted runs of `run` packed keys (the last run may be shorter): runs 2p and 2p + 1 become one run of
// 2 * run keys.  Every thread produces MERGE_ITEMS consecutive outputs: a merge-path search (binary search along the
// output diagonal; the keys are unique, so the split is unique) finds where its outputs start in the two runs, then it
// merges sequentially.  With keys_out != NULL (the last pass) the keys are unpacked on the way out.  Batches beyond the
// LDS sort's 8192 pairs (B > 2730; the throughput-oriented batch of BASELINE configs[4] is B = 2^20) sort this way:
// LDS-sorted runs of 8192 + ceil(log2(runs)) passes over 8 B per pair — bandwidth-trivial, on the side stream.
constexpr int MERGE_ITEMS = 8;
__global__ __launch_bounds__(BLOCK) void merge_runs_kernel(const unsigned long long* __restrict__ in,
                                                           unsigned long long* __restrict__ out, int64_t n, int64_t run,
                                                           int slot_bits, int32_t* __restrict__ keys_out,
                                                           int32_t* __restrict__ slots_out) {
  const int64_t o0 = ((int64_t)blockIdx.x * BLOCK + threadIdx.x) * MERGE_ITEMS;
  if (o0 >= n) return;
  const int64_t a0 = o0 / (2 * run) * (2 * run);
  const int64_t a_len = n - a0 < run ? n - a0 : run;
  const int64_t b0 = a0 + a_len;
  const int64_t b_len = n - b0 < run ? (n - b0 > 0 ? n - b0 : 0) : run;
  const unsigned long long* A = in + a0;
  const unsigned long long* Bv = in + b0;
  const int64_t diag = o0 - a0;  // outputs of this pair that precede this thread's
  int64_t lo = diag > b_len ? diag - b_len : 0, hi = diag < a_len ? diag : a_len;
  while (lo < hi) {
    const int64_t mid = (lo + hi) >> 1;
    if (A[mid] < Bv[diag - 1 - mid]) lo = mid + 1;
    else hi = mid;
  }
  int64_t i = lo, j = diag - lo;
  const unsigned long long slot_mask = ((unsigned long long)1 << slot_bits) - 1;
#pragma unroll
  for (int t = 0; t < MERGE_ITEMS; ++t) {
    if (o0 + t >= n || (i >= a_len && j >= b_len)) break;
    const bool take_a = j >= b_len || (i < a_len && A[i] < Bv[j]);
    const unsigned long long v = take_a ? A[i] : Bv[j];
    i += take_a ? 1 : 0;
    j += take_a ? 0 : 1;
    if (keys_out) {
      keys_out[o0 + t] = (int32_t)(v >> slot_bits);
      slots_out[o0 + t] = (int32_t)(v & slot_mask);
    } else {
      out[o0 + t] = v;
    }
  }
}

template <typename KeyT>
static void launch_lds_sort(const int32_t* keys, const int32_t* slots, int n3, int p2, int slot_bits, int32_t* skeys,
                            int32_t* sslots, hipStream_t st) {
  const int E = p2 <= 1024 ? 1 : p2 / 1024;
  KeyT* none = nullptr;
  switch (E) {
    case 1: hipLaunchKernelGGL((lds_sort_kernel<KeyT, 1>), dim3(1), dim3(1024), 0, st, keys, slots, n3, p2 < 64 ? 64 : p2, slot_bits, skeys, sslots, none); break;
    case 2: hipLaunchKernelGGL((lds_sort_kernel<KeyT, 2>), dim3(1), dim3(1024), 0, st, keys, slots, n3, p2, slot_bits, skeys, sslots, none); break;
    case 4: hipLaunchKernelGGL((lds_sort_kernel<KeyT, 4>), dim3(1), dim3(1024), 0, st, keys, slots, n3, p2, slot_bits, skeys, sslots, none); break;
    default: hipLaunchKernelGGL((lds_sort_kernel<KeyT, 8>), dim3(1), dim3(1024), 0, st, keys, slots, n3, p2, slot_bits, skeys, sslots, none); break;
  }
}

// 1024 < n3 <= LDS_SORT_MAX pairs, two launches on SEVERAL compute units (round 5).  One 1024-thread workgroup sorting the
// whole list is bound by its CU's LDS pipe (every cross-lane exchange of 16 waves goes through it): 45 us for the 6144
// pairs of a B = 2048 batch — the longest thing on the side stream, and for a step without propagation (MFBPR) longer
// than the step itself.  Instead: runs of 1024 keys sorted by one workgroup each (lds_sort_kernel<KeyT, 1>, packed keys
// out), then every key finds its final position on its own: the keys are unique, so rank = position in its run + the
// number of smaller keys in every other run (a binary search per run, all runs staged in LDS).  Same sorted list.
constexpr int RANK_RUN = 1024, RANK_BLOCK = 256;
template <typename KeyT>
__global__ __launch_bounds__(RANK_BLOCK) void rank_merge_kernel(const KeyT* __restrict__ runs, int n, int slot_bits,
                                                                int32_t* __restrict__ keys_out,
                                                                int32_t* __restrict__ slots_out) {
  __shared__ KeyT s_k[LDS_SORT_MAX];
  for (int i = threadIdx.x; i < n; i += RANK_BLOCK) s_k[i] = runs[i];
  __syncthreads();
  const int g = blockIdx.x * RANK_BLOCK + threadIdx.x;
  if (g >= n) return;
  const KeyT key = s_k[g];
  const int own = g / RANK_RUN;
  int rank = g - own * RANK_RUN;
  const int n_runs = (n + RANK_RUN - 1) / RANK_RUN;
  for (int q = 0; q < n_runs; ++q) {
    if (q == own) continue;
    const KeyT* r = s_k + q * RANK_RUN;
    int lo = 0, hi = n - q * RANK_RUN < RANK_RUN ? n - q * RANK_RUN : RANK_RUN;
    while (lo < hi) {  // number of keys of run q below `key`
      const int mid = (lo + hi) >> 1;
      if (r[mid] < key) lo = mid + 1;
      else hi = mid;
    }
    rank += lo;
  }
  keys_out[rank] = (int32_t)(key >> slot_bits);
  slots_out[rank] = (int32_t)(key & (KeyT)(((KeyT)1 << slot_bits) - 1));
}

template <typename KeyT>
static void launch_rank_sort(const int32_t* keys, const int32_t* slots, int n3, int slot_bits, int32_t* skeys, int32_t* sslots,
                             void* temp, hipStream_t st) {
  KeyT* runs = reinterpret_cast<KeyT*>(temp);
  hipLaunchKernelGGL((lds_sort_kernel<KeyT, 1>), dim3((unsigned)((n3 + RANK_RUN - 1) / RANK_RUN)), dim3(1024), 0, st, keys, slots, n3,
                     RANK_RUN, slot_bits, skeys, sslots, runs);
  hipLaunchKernelGGL((rank_merge_kernel<KeyT>), dim3((unsigned)((n3 + RANK_BLOCK - 1) / RANK_BLOCK)), dim3(RANK_BLOCK), 0, st, runs,
                     n3, slot_bits, skeys, sslots);
}

// n3 > LDS_SORT_MAX pairs: runs of LDS_SORT_MAX sorted in LDS (one workgroup each), then merge passes between the two
// halves of `temp` (2 x n3 packed 64-bit keys); the last pass unpacks into skeys / sslots.
static void launch_merge_sort(const int32_t* keys, const int32_t* slots, int64_t n3, int slot_bits, int32_t* skeys,
                              int32_t* sslots, void* temp, hipStream_t st) {
  unsigned long long* buf[2] = {reinterpret_cast<unsigned long long*>(temp), reinterpret_cast<unsigned long long*>(temp) + n3};
  const unsigned runs = (unsigned)((n3 + LDS_SORT_MAX - 1) / LDS_SORT_MAX);
  hipLaunchKernelGGL((lds_sort_kernel<unsigned long long, LDS_SORT_MAX / 1024>), dim3(runs), dim3(1024), 0, st, keys, slots,
                     (int)n3, LDS_SORT_MAX, slot_bits, skeys, sslots, buf[0]);
  const unsigned nb = (unsigned)((n3 + (int64_t)BLOCK * MERGE_ITEMS - 1) / ((int64_t)BLOCK * MERGE_ITEMS));
  int cur = 0;
  for (int64_t run = LDS_SORT_MAX; run < n3; run *= 2) {
    const bool last = run * 2 >= n3;
    hipLaunchKernelGGL(merge_runs_kernel, dim3(nb), dim3(BLOCK), 0, st, buf[cur], buf[cur ^ 1], n3, run, slot_bits,
                       last ? skeys : nullptr, last ? sslots : nullptr);
    cur ^= 1;
  }
}

// One wave per sorted position that starts a run of equal rows.
template <int NT>
__device__ __forceinline__ void bpr_reduce_block(const float* __restrict__ loss_i, const float* __restrict__ sq, int64_t B,
                                                 float reg_lambda, float* __restrict__ loss, float (*s)[1024]);

// loss_out != NULL: one extra trailing workgroup reduces the per-triple loss terms while the others scatter (the
// loss is on nobody's critical path; as its own launch it cost 6 us between the forward and the backward)
__global__ __launch_bounds__(BLOCK) void bpr_scatter_kernel(BprArgs a, const int32_t* __restrict__ skeys,
                                                            const int32_t* __restrict__ sslots, float reg_lambda,
                                                            float* __restrict__ loss_out) {
  if (loss_out && blockIdx.x == gridDim.x - 1) {
    __shared__ float s_red[4][1024];
    bpr_reduce_block<BLOCK>(a.loss_i, a.sq, a.B, reg_lambda, loss_out, s_red);
    return;
  }
  const int64_t j = (int64_t)blockIdx.x * (BLOCK / WAVE) + threadIdx.x / WAVE;
  const int lane = threadIdx.x % WAVE;
  const int64_t n3 = 3 * a.B;
  if (j >= n3) return;
  const int32_t row = skeys[j];
  if (j > 0 && skeys[j - 1] == row) return;
  int64_t e = j + 1;
  while (e < n3 && skeys[e] == row) ++e;
  const float up0 = a.upstream ? a.upstream[0] : 1.0f;
  const float up1 = a.upstream ? a.upstream[1] : 1.0f;
  if (a.touched && !a.touched_preset && lane == 0) atomicOr(a.touched + (row >> 5), 1u << (row & 31));
  // The run's slots are resolved 64 at a time by the lanes in parallel (slot -> triple -> coefficient and the one or two
  // panel rows it reads): the sequential walk below then has ONE dependent load level per slot, with several slots'
  // row loads in flight, instead of three (a hub item's run is the kernel's critical path).  The additions keep
  // their order, so the result is bit-identical.
  for (int64_t f0 = 0; f0 < a.d; f0 += WAVE) {
    const int64_t f = f0 + lane;
    const bool live = f < a.d;
    float acc = 0.f;
    for (int64_t base = j; base < e; base += WAVE) {
      const int cnt = (int)(e - base < WAVE ? e - base : WAVE);
      int32_t my_a = 0, my_b = 0, my_kind = 1;
      float my_c = 0.f;
      if (lane < cnt) {
        const int32_t s = sslots[base + lane];
        const int64_t i = s / 3;
        my_kind = s - 3 * (int32_t)i;
        my_c = a.coef[i] * up0;
        if (my_kind == 0) {
          my_a = (int32_t)(a.num_users + a.pos[i]);
          my_b = (int32_t)(a.num_users + a.neg[i]);
        } else {
          my_a = (int32_t)a.users[i];
        }
      }
#pragma unroll 4
      for (int tt = 0; tt < cnt; ++tt) {
        const int32_t ra = __shfl(my_a, tt, WAVE), rb = __shfl(my_b, tt, WAVE);
        const int kind = __shfl(my_kind, tt, WAVE);
        const float c = __shfl(my_c, tt, WAVE);
        float v = 0.f;
        if (live) {
          if (kind == 0) {
            const float p = a.fin[(int64_t)ra * a.d + f];
            const float n = a.fin[(int64_t)rb * a.d + f];
            v = c * (p - n);
          } else {
            const float u = a.fin[(int64_t)ra * a.d + f];
            v = kind == 1 ? c * u : -c * u;
          }
        }
        acc = (base == j && tt == 0) ? v : acc + v;
      }
    }
    if (!live) continue;
    const int64_t o = (int64_t)row * a.d + f;
    float reg = 0.f;
    if (a.g_ego && a.de == a.d) {
      const float r1 = (a.reg_users || row >= a.num_users) ? (a.reg_scale * up1) * a.ego[o] : 0.f;
      reg = r1;
      for (int64_t t = j + 1; t < e; ++t) reg += r1;
    }
    if (a.de != a.d) {  // ego rows of another width: their gradient rows are written by the loop below
      if (a.g_final) {
        if (a.touched) a.g_final[o] = acc;
        else a.g_final[o] += acc;
      }
      continue;
    }
    if (a.g_final && a.g_final == a.g_ego) {
      // (MFBPR: one panel receives both gradients; with a bitmap the row is stored — what adding to a zero-filled row gives)
      if (a.touched) a.g_final[o] = acc + reg;
      else a.g_final[o] += acc + reg;
    } else {
      if (a.g_final) {
        if (a.touched) a.g_final[o] = acc;  // one wave owns a row: a plain store, no zero-fill needed
        else a.g_final[o] += acc;
      }
      if (a.g_ego) {
        if (a.touched) a.g_ego[o] = reg;
        else a.g_ego[o] += reg;
      }
    }
  }
  if (a.de != a.d && a.g_ego) {
    for (int64_t f = lane; f < a.de; f += WAVE) {
      const int64_t o = (int64_t)row * a.de + f;
      const float r1 = (a.reg_users || row >= a.num_users) ? (a.reg_scale * up1) * a.ego[o] : 0.f;
      float reg = r1;
      for (int64_t t = j + 1; t < e; ++t) reg += r1;
      if (a.touched) a.g_ego[o] = reg;
      else a.g_ego[o] += reg;
    }
  }
}

// loss[0] = mean(loss_i); loss[1] = reg_lambda * sum_blocks 0.5 * (sqrt(sum sq))^2 / B
// One workgroup of NT threads plays the 1024 "virtual threads" of a fixed reduction tree, so that the result does not
// depend on NT (the fused step runs this inside the scatter launch with 256 threads, the stand-alone kernel with 1024).
template <int NT>
__device__ __forceinline__ void bpr_reduce_block(const float* __restrict__ loss_i, const float* __restrict__ sq, int64_t B,
                                                 float reg_lambda, float* __restrict__ loss, float (*s)[1024]) {
  const int tid = threadIdx.x;
  for (int vt = tid; vt < 1024; vt += NT) {
    float acc[4] = {0.f, 0.f, 0.f, 0.f};
    for (int64_t i = vt; i < B; i += 1024) {
      acc[0] += loss_i[i];
      acc[1] += sq[i];
      acc[2] += sq[B + i];
      acc[3] += sq[2 * B + i];
    }
    for (int q = 0; q < 4; ++q) s[q][vt] = acc[q];
  }
  __syncthreads();
  for (int o = 512; o > 0; o >>= 1) {
    for (int vt = tid; vt < o; vt += NT)
      for (int q = 0; q < 4; ++q) s[q][vt] += s[q][vt + o];
    __syncthreads();
  }
  if (tid == 0) {
    const float fB = (float)B;
    loss[0] = s[0][0] / fB;
    float reg = 0.f;
    for (int q = 1; q < 4; ++q) {
      const float nrm = sqrtf(s[q][0]);  // embedding.norm(2)
      reg += 0.5f * (nrm * nrm) / fB;    // 1/2 * norm.pow(2) / B   (losses.py:19)
    }
    loss[1] = reg_lambda * reg;
  }
}

__global__ __launch_bounds__(1024) void bpr_reduce_kernel(const float* __restrict__ loss_i,
                                                          const float* __restrict__ sq, int64_t B, float reg_lambda,
                                                          float* __restrict__ loss) {
  __shared__ float s[4][1024];
  bpr_reduce_block<1024>(loss_i, sq, B, reg_lambda, loss, s);
}

__global__ __launch_bounds__(BLOCK) void adam_kernel(float* __restrict__ p, const float* __restrict__ g,
                                                     float* __restrict__ m, float* __restrict__ v, int64_t n4,
                                                     int64_t n, float w1, float beta2, float w2, float step_size,
                                                     float bc2_sqrt, float eps) {
  const int64_t stride = (int64_t)gridDim.x * BLOCK;
  for (int64_t i = (int64_t)blockIdx.x * BLOCK + threadIdx.x; i < n4; i += stride) {
    float4 P = reinterpret_cast<float4*>(p)[i];
    const float4 G = reinterpret_cast<const float4*>(g)[i];
    float4 M = reinterpret_cast<float4*>(m)[i];
    float4 V = reinterpret_cast<float4*>(v)[i];
#define IDG_ADAM1(c)                                                 \
  M.c = __builtin_fmaf(w1, G.c - M.c, M.c);                          \
  V.c = __builtin_fmaf(w2 * G.c, G.c, V.c * beta2);                  \
  P.c = P.c - step_size * (M.c / (sqrtf(V.c) / bc2_sqrt + eps));
    IDG_ADAM1(x) IDG_ADAM1(y) IDG_ADAM1(z) IDG_ADAM1(w)
    reinterpret_cast<float4*>(p)[i] = P;
    reinterpret_cast<float4*>(m)[i] = M;
    reinterpret_cast<float4*>(v)[i] = V;
  }
  // scalar tail
  if (blockIdx.x == 0) {
    for (int64_t i = n4 * 4 + threadIdx.x; i < n; i += BLOCK) {
      float P = p[i], M = m[i], V = v[i];
      const float G = g[i];
      M = __builtin_fmaf(w1, G - M, M);
      V = __builtin_fmaf(w2 * G, G, V * beta2);
      P = P - step_size * (M / (sqrtf(V) / bc2_sqrt + eps));
      p[i] = P;
      m[i] = M;
      v[i] = V;
    }
  }
#undef IDG_ADAM1
}

// adam_kernel with the gradient read at the rows flagged in `rows` only (zero elsewhere).  One thread per float4 column
// of a row (d4 = d / 4 columns, BLOCK / d4 rows per workgroup pass when d4 divides BLOCK; else rows are walked by the
// first d4-multiple of the block's threads): the row index costs one 32-bit division per THREAD, not one per element
__global__ __launch_bounds__(BLOCK) void adam_rows_kernel(float* __restrict__ p, const float* __restrict__ g,
                                                          const uint32_t* __restrict__ rows, float* __restrict__ m,
                                                          float* __restrict__ v, int64_t n, int d4, float w1, float beta2,
                                                          float w2, float step_size, float bc2_sqrt, float eps) {
  const int rpb = BLOCK / d4;  // rows per workgroup pass (>= 1: the caller guarantees d4 <= BLOCK)
  const int r_in = (int)threadIdx.x / d4, c = (int)threadIdx.x - r_in * d4;
  if (r_in >= rpb) return;
  for (int64_t row = (int64_t)blockIdx.x * rpb + r_in; row < n; row += (int64_t)gridDim.x * rpb) {
    const int64_t i = row * d4 + c;
    const bool live = (rows[row >> 5] >> (row & 31)) & 1u;
    float4 P = reinterpret_cast<float4*>(p)[i];
    float4 G = make_float4(0.f, 0.f, 0.f, 0.f);
    if (live) G = reinterpret_cast<const float4*>(g)[i];
    float4 M = reinterpret_cast<float4*>(m)[i];
    float4 V = reinterpret_cast<float4*>(v)[i];
#define IDG_ADAM1(c)                                                 \
  M.c = __builtin_fmaf(w1, G.c - M.c, M.c);                          \
  V.c = __builtin_fmaf(w2 * G.c, G.c, V.c * beta2);                  \
  P.c = P.c - step_size * (M.c / (sqrtf(V.c) / bc2_sqrt + eps));
    IDG_ADAM1(x) IDG_ADAM1(y) IDG_ADAM1(z) IDG_ADAM1(w)
#undef IDG_ADAM1
    reinterpret_cast<float4*>(p)[i] = P;
    reinterpret_cast<float4*>(m)[i] = M;
    reinterpret_cast<float4*>(v)[i] = V;
  }
}

// dst[t] = idx[t] >= 0 ? src[idx[t]] : 0   (rows of d floats; one wave per row)
__global__ __launch_bounds__(BLOCK) void rows_gather_kernel(float* __restrict__ dst, const float* __restrict__ src,
                                                            const int64_t* __restrict__ idx, int64_t count, int64_t d) {
  const int64_t t = (int64_t)blockIdx.x * (BLOCK / 64) + threadIdx.x / 64;
  if (t >= count) return;
  const int64_t r = idx[t];
  for (int64_t f = threadIdx.x % 64; f < d; f += 64) dst[t * d + f] = r >= 0 ? src[r * d + f] : 0.f;
}

// For every t with idx[t] >= 0 (the head of a chain): dst[idx[t]] += src[t] + src[next[t]] + src[next[next[t]]] + ...
// added in chain order (the caller links the occurrences of one destination in list order), one wave per chain: no
// atomics, run-to-run identical bits.
__global__ __launch_bounds__(BLOCK) void rows_chain_add_kernel(float* __restrict__ dst, const float* __restrict__ src,
                                                               const int64_t* __restrict__ idx,
                                                               const int64_t* __restrict__ next, int64_t count, int64_t d) {
  const int64_t t = (int64_t)blockIdx.x * (BLOCK / 64) + threadIdx.x / 64;
  if (t >= count) return;
  const int64_t r = idx[t];
  if (r < 0) return;
  for (int64_t f = threadIdx.x % 64; f < d; f += 64) {
    float acc = dst[r * d + f];
    for (int64_t j = t; j >= 0; j = next[j]) acc += src[j * d + f];
    dst[r * d + f] = acc;
  }
}

__global__ __launch_bounds__(BLOCK) void lincomb_kernel(float* __restrict__ out, const float* x, float a,
                                                        const float* y, float b, int64_t n4, int64_t n) {
  const int64_t stride = (int64_t)gridDim.x * BLOCK;
  for (int64_t i = (int64_t)blockIdx.x * BLOCK + threadIdx.x; i < n4; i += stride) {
    const float4 X = reinterpret_cast<const float4*>(x)[i];
    float4 R = make_float4(a * X.x, a * X.y, a * X.z, a * X.w);
    if (y) {
      const float4 Y = reinterpret_cast<const float4*>(y)[i];
      R.x = __builtin_fmaf(b, Y.x, R.x);
      R.y = __builtin_fmaf(b, Y.y, R.y);
      R.z = __builtin_fmaf(b, Y.z, R.z);
      R.w = __builtin_fmaf(b, Y.w, R.w);
    }
    reinterpret_cast<float4*>(out)[i] = R;
  }
  if (blockIdx.x == 0)
    for (int64_t i = n4 * 4 + threadIdx.x; i < n; i += BLOCK) out[i] = y ? __builtin_fmaf(b, y[i], a * x[i]) : a * x[i];
}

// ---- gradient rows as a message (data-parallel replicas, id-grec_amd/replicated.py).  Backward propagation is linear, so
// replicas exchange the batch's few thousand gradient rows BEFORE it instead of all-reducing the dense [n, d] result
// after it.  Message of one rank, in 4-byte words: [0] bpr loss, [1] reg loss, ... | keys[3B] (the batch's rows,
// ascending, one per (triple, role) slot: the sorted scatter plan) | rows[3B, d] (g_final row of keys[j], present at the
// first slot of each run of equal keys).
constexpr int64_t MSG_HEADER = 64;
constexpr int MSG_MAX_WORLD = 64;

struct MsgLayout {
  int64_t keys, rows, total;
};

inline MsgLayout msg_layout(int64_t B, int64_t d) {
  MsgLayout m;
  m.keys = MSG_HEADER;
  m.rows = (m.keys + 3 * B + 63) / 64 * 64;
  m.total = m.rows + 3 * B * d;
  return m;
}

__global__ __launch_bounds__(BLOCK) void bpr_pack_rows_kernel(const int32_t* __restrict__ skeys, int64_t n3, int64_t d,
                                                              const float* __restrict__ g_final,
                                                              const float* __restrict__ loss, float* __restrict__ header,
                                                              int32_t* __restrict__ keys, float* __restrict__ rows,
                                                              uint32_t* __restrict__ clear, int64_t clear_words) {
  if (blockIdx.x == 0 && threadIdx.x < MSG_HEADER) header[threadIdx.x] = threadIdx.x < 2 ? loss[threadIdx.x] : 0.f;
  // the bitmap the merge will fill (idg_bpr_unpack_rows_f32) is cleared here, on the way: a memset between the
  // all-gather and the merge would sit on the step's critical path (two fill launches, ~10 us)
  for (int64_t w = (int64_t)blockIdx.x * BLOCK + threadIdx.x; w < clear_words; w += (int64_t)gridDim.x * BLOCK) clear[w] = 0u;
  const int64_t j = (int64_t)blockIdx.x * (BLOCK / WAVE) + threadIdx.x / WAVE;
  const int lane = threadIdx.x % WAVE;
  if (j >= n3) return;
  const int32_t row = skeys[j];
  if (lane == 0) keys[j] = row;
  if (j > 0 && skeys[j - 1] == row) return;
  for (int64_t f = lane; f < d; f += WAVE) rows[j * d + f] = g_final[(int64_t)row * d + f];
}

// First position of `row` in the ascending keys[0..n), or -1: a 64-ary search, the wave samples 64 positions per round
// (3B = 3072 keys: two dependent loads).  Every lane of the wave must call it with the same arguments.
__device__ __forceinline__ int64_t wave_find_first(const int32_t* __restrict__ keys, int64_t n, int32_t row, int lane) {
  int64_t lo = 0, len = n;
  while (len > WAVE) {
    const int64_t stride = (len + WAVE - 1) / WAVE;
    const int64_t p = lo + lane * stride;
    const bool less = p < lo + len && keys[p] < row;
    const int c = __popcll(__ballot(less));  // the samples below `row` are a prefix: keys ascend
    if (c == 0) {
      len = 1;  // keys[lo] >= row: the lower bound is lo itself
    } else {
      const int64_t nlo = lo + (int64_t)(c - 1) * stride + 1;  // lower bound in (sample c-1, sample c]
      const int64_t end = lo + (int64_t)c * stride + 1 < lo + len ? lo + (int64_t)c * stride + 1 : lo + len;
      lo = nlo;
      len = end - nlo;
    }
  }
  const bool eq = lane < len && keys[lo + lane] == row;
  const unsigned long long m = __ballot(eq);
  return m ? lo + (int64_t)__builtin_ctzll(m) : -1;
}

// One wave per (rank, slot).  The wave of the LOWEST rank that names a row owns it: it adds the ranks' rows in rank
// order — every replica performs the same additions in the same order and ends with the same bits — and stores the
// result (the panels are never zero-filled; `touched` flags the stored rows).
__global__ __launch_bounds__(BLOCK) void bpr_unpack_rows_kernel(const float* __restrict__ messages, int world, MsgLayout m,
                                                                int64_t n3, int64_t d, float scale, float reg_scale,
                                                                const float* __restrict__ ego, float* __restrict__ g_final,
                                                                float* __restrict__ g_ego, uint32_t* __restrict__ touched,
                                                                float* __restrict__ loss, unsigned blocks_per_rank) {
  if (blockIdx.x == 0 && threadIdx.x < 2) {
    float s = messages[threadIdx.x] * scale;
    for (int r = 1; r < world; ++r) s += messages[(int64_t)r * m.total + threadIdx.x] * scale;
    loss[threadIdx.x] = s;
  }
  const int r = (int)(blockIdx.x / blocks_per_rank);
  const int64_t j = (int64_t)(blockIdx.x % blocks_per_rank) * (BLOCK / WAVE) + threadIdx.x / WAVE;
  const int lane = threadIdx.x % WAVE;
  if (j >= n3) return;
  const float* msg = messages + (int64_t)r * m.total;
  const int32_t* keys = reinterpret_cast<const int32_t*>(msg + m.keys);
  const int32_t row = keys[j];
  if (j > 0 && keys[j - 1] == row) return;
  for (int q = 0; q < r; ++q)
    if (wave_find_first(reinterpret_cast<const int32_t*>(messages + (int64_t)q * m.total + m.keys), n3, row, lane) >= 0) return;
  int64_t cnt = 1;
  while (j + cnt < n3 && keys[j + cnt] == row) ++cnt;
  int64_t mine = -1;  // lane q: where rank q (> r) keeps this row
  for (int q = r + 1; q < world; ++q) {
    const int32_t* kq = reinterpret_cast<const int32_t*>(messages + (int64_t)q * m.total + m.keys);
    const int64_t p = wave_find_first(kq, n3, row, lane);
    if (p >= 0) {
      int64_t e = p + 1;
      while (e < n3 && kq[e] == row) ++e;
      cnt += e - p;
    }
    if (lane == q) mine = p;
  }
  if (lane == 0) atomicOr(touched + (row >> 5), 1u << (row & 31));
  for (int64_t f0 = 0; f0 < d; f0 += WAVE) {
    const int64_t f = f0 + lane;
    const bool live = f < d;
    float acc = live ? msg[m.rows + j * d + f] * scale : 0.f;
    for (int q = r + 1; q < world; ++q) {
      const int64_t p = __shfl(mine, q, WAVE);
      if (p >= 0 && live) acc += messages[(int64_t)q * m.total + m.rows + p * d + f] * scale;
    }
    if (live) {
      const int64_t o = (int64_t)row * d + f;
      const float r1 = reg_scale * ego[o];
      float reg = r1;
      for (int64_t t = 1; t < cnt; ++t) reg += r1;
      g_final[o] = acc;
      g_ego[o] = reg;
    }
  }
}

inline size_t align256(size_t x) { return (x + 255) / 256 * 256; }

struct BprWs {
  size_t coef, loss_i, sq, keys, slots, skeys, sslots, temp, total;
};

BprWs bpr_layout(int64_t B, size_t sort_temp) {
  BprWs w;
  size_t o = 0;
  w.coef = o;
  o += align256((size_t)B * 4);
  w.loss_i = o;
  o += align256((size_t)B * 4);
  w.sq = o;
  o += align256((size_t)B * 12);
  w.keys = o;
  o += align256((size_t)B * 12);
  w.slots = o;
  o += align256((size_t)B * 12);
  w.skeys = o;
  o += align256((size_t)B * 12);
  w.sslots = o;
  o += align256((size_t)B * 12);
  w.temp = o;
  o += align256(sort_temp);
  w.total = o;
  return w;
}

// Scratch of the sort of the (row, slot) pairs: none up to one run of 1024; the packed runs (n3 keys of up to 64 bits) up
// to LDS_SORT_MAX; two buffers of n3 packed 64-bit keys for the merge sort beyond.
size_t sort_temp_bytes(int64_t n3) {
  if (n3 <= RANK_RUN) return 0;
  if (n3 <= LDS_SORT_MAX) return (size_t)n3 * 8;
  return (size_t)n3 * 16;
}

}  // namespace

static int bpr_sort_plan(const int64_t* users, const int64_t* pos, const int64_t* neg, int64_t B, int64_t num_users, int64_t n,
                         void* ws, hipStream_t st, const char* who, uint32_t* bitmap, uint32_t* zero2, bool one_launch_sort);
int idg::bpr_plan_rows(const int64_t* users, const int64_t* pos, const int64_t* neg, int64_t B, int64_t num_users, int64_t n,
                       void* ws, uint32_t* bitmap, uint32_t* zero2, bool one_launch_sort, void* stream) {
  if (!bitmap) return idg::fail(IDG_E_INVALID, "bpr_plan_rows: NULL bitmap");
  return bpr_sort_plan(users, pos, neg, B, num_users, n, ws, (hipStream_t)stream, "idg_step_run_f32 (plan)", bitmap, zero2,
                       one_launch_sort);
}

extern "C" {

size_t idg_bpr_workspace_bytes(int64_t B, int64_t d) {
  (void)d;
  if (B <= 0) return 0;
  return bpr_layout(B, sort_temp_bytes(3 * B)).total;
}

static int bpr_args(BprArgs& a, const BprWs& w, const float* final_panel, const float* ego_panel,
                    int64_t num_users, int64_t n, const int64_t* users, const int64_t* pos, const int64_t* neg,
                    int64_t B, int64_t d, float reg_lambda, void* ws, const char* who) {
  if (!(final_panel && ego_panel && users && pos && neg && ws)) return idg::fail(IDG_E_INVALID, "%s: NULL argument", who);
  if (!(B > 0 && d > 0 && num_users >= 0 && n >= num_users)) return idg::fail(IDG_E_INVALID, "%s: bad sizes", who);
  if (n >= ((int64_t)1 << 31)) return idg::fail(IDG_E_INVALID, "%s: more than 2^31 rows", who);
  if (3 * B >= ((int64_t)1 << 31)) return idg::fail(IDG_E_INVALID, "%s: batch too large", who);
  char* base = reinterpret_cast<char*>(ws);
  a.fin = final_panel;
  a.ego = ego_panel;
  a.users = users;
  a.pos = pos;
  a.neg = neg;
  a.num_users = num_users;
  a.B = B;
  a.d = d;
  a.de = d;
  a.reg_users = 1;
  a.inv_B = 1.0f / (float)B;
  a.reg_scale = reg_lambda / (float)B;
  a.coef = reinterpret_cast<float*>(base + w.coef);
  a.loss_i = reinterpret_cast<float*>(base + w.loss_i);
  a.sq = reinterpret_cast<float*>(base + w.sq);
  a.keys = reinterpret_cast<int32_t*>(base + w.keys);
  a.slots = reinterpret_cast<int32_t*>(base + w.slots);
  return IDG_OK;
}

static int bpr_forward_impl(const float* final_panel, const float* ego_panel, int64_t num_users, int64_t n,
                            const int64_t* users, const int64_t* pos, const int64_t* neg, int64_t B, int64_t d,
                            float reg_lambda, float* loss, void* ws, void* stream, bool reduce_now, int64_t de = 0,
                            int reg_users = 1) {
  IDG_REQUIRE(loss, "idg_bpr_forward_f32: loss is NULL");
  hipStream_t st = (hipStream_t)stream;
  const BprWs w = bpr_layout(B > 0 ? B : 1, sort_temp_bytes(3 * B));
  BprArgs a{};
  int rc = bpr_args(a, w, final_panel, ego_panel, num_users, n, users, pos, neg, B, d, reg_lambda, ws,
                    "idg_bpr_forward_f32");
  if (rc != IDG_OK) return rc;
  if (de > 0) a.de = de;
  a.reg_users = reg_users;
  const unsigned nb = (unsigned)((B + (BLOCK / WAVE) - 1) / (BLOCK / WAVE));
  hipLaunchKernelGGL(bpr_triple_kernel, dim3(nb), dim3(BLOCK), 0, st, a);
  if (reduce_now) hipLaunchKernelGGL(bpr_reduce_kernel, dim3(1), dim3(1024), 0, st, a.loss_i, a.sq, B, reg_lambda, loss);
  IDG_HIP(hipGetLastError());
  return IDG_OK;
}

int idg_bpr_forward_f32(const float* final_panel, const float* ego_panel, int64_t num_users, int64_t n,
                        const int64_t* users, const int64_t* pos, const int64_t* neg, int64_t B, int64_t d,
                        float reg_lambda, float* loss, void* ws, void* stream) {
  return bpr_forward_impl(final_panel, ego_panel, num_users, n, users, pos, neg, B, d, reg_lambda, loss, ws, stream, true);
}

static int bpr_sort_plan(const int64_t* users, const int64_t* pos, const int64_t* neg, int64_t B, int64_t num_users,
                         int64_t n, void* ws, hipStream_t st, const char* who, uint32_t* bitmap = nullptr,
                         uint32_t* zero2 = nullptr, bool one_launch_sort = false) {
  if (!(users && pos && neg && ws)) return idg::fail(IDG_E_INVALID, "%s: NULL argument", who);
  if (!(B > 0 && num_users >= 0 && n >= num_users)) return idg::fail(IDG_E_INVALID, "%s: bad sizes", who);
  if (n >= ((int64_t)1 << 31) || 3 * B >= ((int64_t)1 << 31)) return idg::fail(IDG_E_INVALID, "%s: sizes exceed int32 keys", who);
  const BprWs w = bpr_layout(B, sort_temp_bytes(3 * B));
  char* base = reinterpret_cast<char*>(ws);
  int32_t* keys = reinterpret_cast<int32_t*>(base + w.keys);
  int32_t* slots = reinterpret_cast<int32_t*>(base + w.slots);
  int32_t* skeys = reinterpret_cast<int32_t*>(base + w.skeys);
  int32_t* sslots = reinterpret_cast<int32_t*>(base + w.sslots);
  if (bitmap) idg::rows_changed(bitmap);
  hipLaunchKernelGGL(bpr_keys_kernel, dim3((unsigned)((B + BLOCK - 1) / BLOCK)), dim3(BLOCK), 0, st, users, pos, neg, B,
                     num_users, keys, slots, bitmap, zero2);
  const int64_t n3 = 3 * B;
  if (n3 <= LDS_SORT_MAX) {
    int p2 = 1;
    while (p2 < n3) p2 <<= 1;
    int slot_bits = 1, row_bits = 1;
    while (((int64_t)1 << slot_bits) < n3) ++slot_bits;
    while (((int64_t)1 << row_bits) < n) ++row_bits;
    const bool rank_sort = n3 > RANK_RUN && !g_single_block_sort && !(one_launch_sort && n3 <= 4096);
    if (slot_bits + row_bits <= 31) {  // top bit kept clear so the all-ones padding key sorts last
      if (rank_sort) launch_rank_sort<uint32_t>(keys, slots, (int)n3, slot_bits, skeys, sslots, base + w.temp, st);
      else launch_lds_sort<uint32_t>(keys, slots, (int)n3, p2, slot_bits, skeys, sslots, st);
    } else {
      if (rank_sort) launch_rank_sort<unsigned long long>(keys, slots, (int)n3, slot_bits, skeys, sslots, base + w.temp, st);
      else launch_lds_sort<unsigned long long>(keys, slots, (int)n3, p2, slot_bits, skeys, sslots, st);
    }
  } else {
    int slot_bits = 1;
    while (((int64_t)1 << slot_bits) < n3) ++slot_bits;
    launch_merge_sort(keys, slots, n3, slot_bits, skeys, sslots, base + w.temp, st);
  }
  IDG_HIP(hipGetLastError());
  return IDG_OK;
}

int idg_bpr_touch_rows(const int64_t* users, const int64_t* pos, const int64_t* neg, int64_t B, int64_t num_users,
                       uint32_t* bitmap, void* stream) {
  IDG_REQUIRE(users && pos && neg && bitmap, "idg_bpr_touch_rows: NULL argument");
  IDG_REQUIRE(B > 0 && num_users >= 0, "idg_bpr_touch_rows: bad sizes");
  idg::rows_changed(bitmap);
  hipLaunchKernelGGL(bpr_touch_kernel, dim3((unsigned)((B + BLOCK - 1) / BLOCK)), dim3(BLOCK), 0, (hipStream_t)stream, users,
                     pos, neg, B, num_users, bitmap);
  IDG_HIP(hipGetLastError());
  return IDG_OK;
}

int idg_bitmap_clear(uint32_t* bitmap, int64_t n_bits, void* stream) {
  IDG_REQUIRE(bitmap && n_bits >= 0, "idg_bitmap_clear: bad argument");
  idg::rows_changed(bitmap, (size_t)((n_bits + 31) / 32) * sizeof(uint32_t));
  IDG_HIP(hipMemsetAsync(bitmap, 0, (size_t)((n_bits + 31) / 32) * sizeof(uint32_t), (hipStream_t)stream));
  return IDG_OK;
}

int idg_bpr_plan_f32(const int64_t* users, const int64_t* pos, const int64_t* neg, int64_t B, int64_t num_users,
                     int64_t n, void* ws, void* stream) {
  return bpr_sort_plan(users, pos, neg, B, num_users, n, ws, (hipStream_t)stream, "idg_bpr_plan_f32");
}

int idg_bpr_plan_rows_f32(const int64_t* users, const int64_t* pos, const int64_t* neg, int64_t B, int64_t num_users,
                          int64_t n, void* ws, uint32_t* bitmap, void* stream) {
  IDG_REQUIRE(bitmap, "idg_bpr_plan_rows_f32: NULL bitmap");
  return bpr_sort_plan(users, pos, neg, B, num_users, n, ws, (hipStream_t)stream, "idg_bpr_plan_rows_f32", bitmap);
}

// loss_out != NULL (deterministic scatter only): the scatter launch also reduces the loss terms the forward left
// in the workspace (the forward was then run without its own reduction)
static int bpr_backward_impl(const float* final_panel, const float* ego_panel, int64_t num_users, int64_t n,
                             const int64_t* users, const int64_t* pos, const int64_t* neg, int64_t B, int64_t d,
                             float reg_lambda, const float* upstream, float* g_final, float* g_ego, int deterministic,
                             uint32_t* touched, void* ws, void* stream, float* loss_out, int64_t de = 0, int reg_users = 1) {
  hipStream_t st = (hipStream_t)stream;
  IDG_REQUIRE(!touched || (deterministic && (g_final != g_ego || (de <= 0 || de == d) && reg_users)),
              "idg_bpr_backward_f32: a touched-row bitmap needs a deterministic scatter (and, for ego rows of another width or an "
              "item-only regulariser, g_final distinct from g_ego)");
  const BprWs w = bpr_layout(B > 0 ? B : 1, sort_temp_bytes(3 * B));
  BprArgs a{};
  int rc = bpr_args(a, w, final_panel, ego_panel, num_users, n, users, pos, neg, B, d, reg_lambda, ws,
                    "idg_bpr_backward_f32");
  if (rc != IDG_OK) return rc;
  if (!g_final && !g_ego) return IDG_OK;
  if (de > 0) a.de = de;
  a.reg_users = reg_users;
  IDG_REQUIRE((a.de == a.d && a.reg_users) || (deterministic && g_final != g_ego),
              "idg_bpr: ego rows of another width / an item-only regulariser need the deterministic scatter and distinct panels");
  a.g_final = g_final;
  a.g_ego = g_ego;
  a.upstream = upstream;
  a.touched = touched;
  a.touched_preset = (deterministic & IDG_BPR_TOUCHED_PRESET) ? 1 : 0;
  deterministic &= ~IDG_BPR_TOUCHED_PRESET;
  // (the scatter sets bits in it — unless the caller says they are set already: lists registered for the bitmap stay valid)
  if (!a.touched_preset) idg::rows_changed(touched, (size_t)((n + 31) / 32) * sizeof(uint32_t));
  char* base = reinterpret_cast<char*>(ws);
  if (!deterministic) {
    const unsigned nb = (unsigned)((B + (BLOCK / WAVE) - 1) / (BLOCK / WAVE));
    hipLaunchKernelGGL(bpr_atomic_kernel, dim3(nb), dim3(BLOCK), 0, st, a);
  } else {
    if (deterministic != IDG_BPR_PLANNED) {
      rc = bpr_sort_plan(users, pos, neg, B, num_users, n, ws, st, "idg_bpr_backward_f32");
      if (rc != IDG_OK) return rc;
    }
    const int64_t n3 = 3 * B;
    const int32_t* skeys = reinterpret_cast<const int32_t*>(base + w.skeys);
    const int32_t* sslots = reinterpret_cast<const int32_t*>(base + w.sslots);
    const unsigned nb3 = (unsigned)((n3 + (BLOCK / WAVE) - 1) / (BLOCK / WAVE));
    hipLaunchKernelGGL(bpr_scatter_kernel, dim3(nb3 + (loss_out ? 1u : 0u)), dim3(BLOCK), 0, st, a, skeys, sslots, reg_lambda,
                       loss_out);
  }
  IDG_HIP(hipGetLastError());
  return IDG_OK;
}

int idg_bpr_backward_f32(const float* final_panel, const float* ego_panel, int64_t num_users, int64_t n,
                         const int64_t* users, const int64_t* pos, const int64_t* neg, int64_t B, int64_t d,
                         float reg_lambda, const float* upstream, float* g_final, float* g_ego, int deterministic,
                         uint32_t* touched, void* ws, void* stream) {
  return bpr_backward_impl(final_panel, ego_panel, num_users, n, users, pos, neg, B, d, reg_lambda, upstream, g_final, g_ego,
                           deterministic, touched, ws, stream, nullptr);
}

int idg_bpr_fused_f32(const float* final_panel, const float* ego_panel, int64_t num_users, int64_t n,
                      const int64_t* users, const int64_t* pos, const int64_t* neg, int64_t B, int64_t d,
                      float reg_lambda, float* loss, float* g_final, float* g_ego, int deterministic,
                      uint32_t* touched, void* ws, void* stream) {
  // with a deterministic scatter the loss reduction rides in the scatter launch (same reduction tree, same bits)
  const bool ride = (deterministic & ~IDG_BPR_TOUCHED_PRESET) != 0 && (g_final || g_ego);
  int rc = bpr_forward_impl(final_panel, ego_panel, num_users, n, users, pos, neg, B, d, reg_lambda, loss, ws, stream, !ride);
  if (rc != IDG_OK) return rc;
  return bpr_backward_impl(final_panel, ego_panel, num_users, n, users, pos, neg, B, d, reg_lambda, nullptr, g_final, g_ego,
                           deterministic, touched, ws, stream, ride ? loss : nullptr);
}

int idg_bpr_fused_ex_f32(const float* final_panel, int64_t d_final, const float* ego_panel, int64_t d_ego, int64_t num_users,
                         int64_t n, const int64_t* users, const int64_t* pos, const int64_t* neg, int64_t B, float reg_lambda,
                         int reg_users, float* loss, float* g_final, float* g_ego, int deterministic, uint32_t* touched,
                         void* ws, void* stream) {
  IDG_REQUIRE(d_final > 0 && d_ego > 0, "idg_bpr_fused_ex_f32: bad widths");
  IDG_REQUIRE(deterministic != 0, "idg_bpr_fused_ex_f32: deterministic scatter only");
  const bool ride = g_final || g_ego;
  int rc = bpr_forward_impl(final_panel, ego_panel, num_users, n, users, pos, neg, B, d_final, reg_lambda, loss, ws, stream,
                            !ride, d_ego, reg_users ? 1 : 0);
  if (rc != IDG_OK) return rc;
  return bpr_backward_impl(final_panel, ego_panel, num_users, n, users, pos, neg, B, d_final, reg_lambda, nullptr, g_final,
                           g_ego, deterministic, touched, ws, stream, ride ? loss : nullptr, d_ego, reg_users ? 1 : 0);
}

size_t idg_bpr_rows_message_floats(int64_t B, int64_t d) {
  if (B <= 0 || d <= 0) return 0;
  return (size_t)msg_layout(B, d).total;
}

int idg_bpr_pack_rows_f32(const void* ws, int64_t B, int64_t d, const float* g_final, const float* loss, float* message,
                          uint32_t* clear_bitmap, int64_t clear_bits, void* stream) {
  IDG_REQUIRE(ws && g_final && loss && message, "idg_bpr_pack_rows_f32: NULL argument");
  IDG_REQUIRE(clear_bits >= 0 && (clear_bitmap || clear_bits == 0), "idg_bpr_pack_rows_f32: bad bitmap to clear");
  IDG_REQUIRE(B > 0 && d > 0 && 3 * B < ((int64_t)1 << 31), "idg_bpr_pack_rows_f32: bad sizes");
  idg::rows_changed(clear_bitmap, (size_t)((clear_bits + 31) / 32) * sizeof(uint32_t));
  const BprWs w = bpr_layout(B, sort_temp_bytes(3 * B));
  const MsgLayout m = msg_layout(B, d);
  const int32_t* skeys = reinterpret_cast<const int32_t*>(reinterpret_cast<const char*>(ws) + w.skeys);
  const int64_t n3 = 3 * B;
  const unsigned nb = (unsigned)((n3 + (BLOCK / WAVE) - 1) / (BLOCK / WAVE));
  hipLaunchKernelGGL(bpr_pack_rows_kernel, dim3(nb), dim3(BLOCK), 0, (hipStream_t)stream, skeys, n3, d, g_final, loss, message,
                     reinterpret_cast<int32_t*>(message + m.keys), message + m.rows, clear_bitmap, (clear_bits + 31) / 32);
  IDG_HIP(hipGetLastError());
  return IDG_OK;
}

int idg_bpr_unpack_rows_f32(const float* messages, int world, int64_t B, int64_t d, int64_t n, const float* ego_panel,
                            float reg_lambda, float* g_final, float* g_ego, uint32_t* touched, int touched_is_clear,
                            float* loss, void* stream) {
  IDG_REQUIRE(messages && ego_panel && g_final && g_ego && touched && loss, "idg_bpr_unpack_rows_f32: NULL argument");
  IDG_REQUIRE(world > 0 && world <= MSG_MAX_WORLD, "idg_bpr_unpack_rows_f32: world size %d outside [1, %d]", world, MSG_MAX_WORLD);
  IDG_REQUIRE(B > 0 && d > 0 && n > 0 && g_final != g_ego, "idg_bpr_unpack_rows_f32: bad sizes / aliased panels");
  hipStream_t st = (hipStream_t)stream;
  idg::rows_changed(touched, (size_t)((n + 31) / 32) * sizeof(uint32_t));
  const MsgLayout m = msg_layout(B, d);
  const int64_t n3 = 3 * B;
  const int64_t nb = (n3 + (BLOCK / WAVE) - 1) / (BLOCK / WAVE);
  IDG_REQUIRE(nb * world < ((int64_t)1 << 31), "idg_bpr_unpack_rows_f32: batch x world too large for one launch");
  const float scale = 1.0f / (float)world;
  const float reg_scale = (reg_lambda / (float)B) * scale;
  if (!touched_is_clear) IDG_HIP(hipMemsetAsync(touched, 0, (size_t)((n + 31) / 32) * sizeof(uint32_t), st));
  hipLaunchKernelGGL(bpr_unpack_rows_kernel, dim3((unsigned)(nb * world)), dim3(BLOCK), 0, st, messages, world, m, n3, d, scale,
                     reg_scale, ego_panel, g_final, g_ego, touched, loss, (unsigned)nb);
  IDG_HIP(hipGetLastError());
  return IDG_OK;
}

int idg_lincomb_f32(float* out, const float* x, float a, const float* y, float b, int64_t count, void* stream) {
  IDG_REQUIRE(out && x && count >= 0, "idg_lincomb_f32: NULL argument / negative count");
  if (count == 0) return IDG_OK;
  IDG_REQUIRE(((uintptr_t)out | (uintptr_t)x | (uintptr_t)y) % 16 == 0, "idg_lincomb_f32: pointers must be 16-byte aligned");
  const int64_t n4 = count / 4;
  int64_t nb = std::max<int64_t>(1, std::min<int64_t>((n4 + BLOCK - 1) / BLOCK, 256 * 8));
  hipLaunchKernelGGL(lincomb_kernel, dim3((unsigned)nb), dim3(BLOCK), 0, (hipStream_t)stream, out, x, a, y, b, n4, count);
  IDG_HIP(hipGetLastError());
  return IDG_OK;
}

int idg_rows_gather_f32(float* dst, const float* src, const int64_t* idx, int64_t count, int64_t d, void* stream) {
  IDG_REQUIRE(dst && src && idx && count >= 0 && d > 0, "idg_rows_gather_f32: bad argument");
  if (count == 0) return IDG_OK;
  hipLaunchKernelGGL(rows_gather_kernel, dim3((unsigned)((count + BLOCK / 64 - 1) / (BLOCK / 64))), dim3(BLOCK), 0,
                     (hipStream_t)stream, dst, src, idx, count, d);
  IDG_HIP(hipGetLastError());
  return IDG_OK;
}

int idg_rows_chain_add_f32(float* dst, const float* src, const int64_t* idx, const int64_t* next, int64_t count, int64_t d,
                           void* stream) {
  IDG_REQUIRE(dst && src && idx && next && count >= 0 && d > 0, "idg_rows_chain_add_f32: bad argument");
  if (count == 0) return IDG_OK;
  hipLaunchKernelGGL(rows_chain_add_kernel, dim3((unsigned)((count + BLOCK / 64 - 1) / (BLOCK / 64))), dim3(BLOCK), 0,
                     (hipStream_t)stream, dst, src, idx, next, count, d);
  IDG_HIP(hipGetLastError());
  return IDG_OK;
}

int idg_adam_step_f32(float* param, const float* grad, float* exp_avg, float* exp_avg_sq, int64_t count, double lr,
                      double beta1, double beta2, double eps, int64_t step, void* stream) {
  IDG_REQUIRE(param && grad && exp_avg && exp_avg_sq, "idg_adam_step_f32: NULL argument");
  IDG_REQUIRE(count >= 0 && step >= 1, "idg_adam_step_f32: bad count/step");
  if (count == 0) return IDG_OK;
  IDG_REQUIRE(((uintptr_t)param | (uintptr_t)grad | (uintptr_t)exp_avg | (uintptr_t)exp_avg_sq) % 16 == 0,
              "idg_adam_step_f32: pointers must be 16-byte aligned");
  // scalars in double on the host, exactly as torch/optim/adam.py forms them
  const double bc1 = 1.0 - std::pow(beta1, (double)step);
  const double bc2 = 1.0 - std::pow(beta2, (double)step);
  const float step_size = (float)(lr / bc1);
  const float bc2_sqrt = (float)std::sqrt(bc2);
  const int64_t n4 = count / 4;
  int64_t nb = (n4 + BLOCK - 1) / BLOCK;
  nb = std::max<int64_t>(1, std::min<int64_t>(nb, 256 * 8));
  hipLaunchKernelGGL(adam_kernel, dim3((unsigned)nb), dim3(BLOCK), 0, (hipStream_t)stream, param, grad, exp_avg,
                     exp_avg_sq, n4, count, (float)(1.0 - beta1), (float)beta2, (float)(1.0 - beta2), step_size,
                     bc2_sqrt, (float)eps);
  IDG_HIP(hipGetLastError());
  return IDG_OK;
}

int idg_adam_rows_f32(float* param, const float* grad, const uint32_t* rows, float* exp_avg, float* exp_avg_sq, int64_t n,
                      int64_t d, double lr, double beta1, double beta2, double eps, int64_t step, void* stream) {
  IDG_REQUIRE(param && grad && rows && exp_avg && exp_avg_sq, "idg_adam_rows_f32: NULL argument");
  IDG_REQUIRE(n >= 0 && d > 0 && d % 4 == 0 && step >= 1, "idg_adam_rows_f32: bad sizes (d must be a multiple of 4) / step");
  if (n == 0) return IDG_OK;
  IDG_REQUIRE(((uintptr_t)param | (uintptr_t)grad | (uintptr_t)exp_avg | (uintptr_t)exp_avg_sq) % 16 == 0,
              "idg_adam_rows_f32: pointers must be 16-byte aligned");
  const double bc1 = 1.0 - std::pow(beta1, (double)step);
  const double bc2 = 1.0 - std::pow(beta2, (double)step);
  IDG_REQUIRE(d / 4 <= BLOCK, "idg_adam_rows_f32: rows of up to %d floats", 4 * BLOCK);
  const int d4 = (int)(d / 4), rpb = BLOCK / d4;
  int64_t nb = std::max<int64_t>(1, std::min<int64_t>((n + rpb - 1) / rpb, 256 * 8));
  hipLaunchKernelGGL(adam_rows_kernel, dim3((unsigned)nb), dim3(BLOCK), 0, (hipStream_t)stream, param, grad, rows, exp_avg,
                     exp_avg_sq, n, d4, (float)(1.0 - beta1), (float)beta2, (float)(1.0 - beta2), (float)(lr / bc1),
                     (float)std::sqrt(bc2), (float)eps);
  IDG_HIP(hipGetLastError());
  return IDG_OK;
}

}  // extern "C"
