// Full-rank scoring (fp32 MFMA GEMM + sigmoid), train-positive masking and top-K.
//
// Scoring is the one GEMM-shaped op on the path: rating = act(U[users] . V^T), d = 64..256.
// It runs on v_mfma_f32_32x32x2_f32 (exact fp32 products and accumulation).  Each wave owns
// a 32-user x 32-item accumulator tile; lane (i, h) feeds user i's features
// [kc+32h, kc+32h+32) as the A operand and item i's same feature range as the B operand,
// so every lane reads one contiguous 128-byte run per 64-deep k chunk and nothing is
// transposed or staged through LDS.
//
// Top-K is a wave-level streaming select over one user's scores, fused behind the MFMA tiles
// (scores go MFMA accumulator -> LDS slab -> select; never to global memory): a 64-bit key
// (order-preserving score bits << 32 | ~item) makes "larger key" mean "better score, then
// lower item id"; a user's running list sits in registers (lane = rank), the steady state is
// two float compares per 128 scores, and the occasional candidate above the k-th key is
// inserted with a one-lane DPP shift.  One 128-key bitonic network (shuffles only) per user
// and catalogue chunk fills the list at the start.
#include <hip/hip_runtime.h>

#include <algorithm>
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
// Order of the score word: -NaN* < -inf < ... < -0 < +0 < ... < +inf < NaN.  A NaN score (a NaN or an infinity in a row of
// either table) therefore ranks ABOVE every number, as torch.topk ranks it (batch_test.py:68); every float pre-filter in
// front of the key compares tests "not below", so that NaNs reach the keys.  (*a NaN with the sign bit set — never what
// the fmaf chain / the matrix cores make of the default NaN — ranks lowest.)
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

// ---- fused scoring + masking + top-K -----------------------------------------------------
// One 256-thread workgroup owns 64 batch users and one chunk of the catalogue, walked in slabs
// of 128 items.  Per slab: (A) every wave computes the 64 x 32 scores of its 32-item column tile
// with 2 x 32 v_mfma_f32_32x32x2_f32 per 64-deep k chunk and writes them to an LDS slab;
// (B) every wave runs the streaming select for its 16 users straight out of LDS.  A user's
// running best list lives in REGISTERS (lane L = rank L, 16 users x one 64-bit key per lane)
// and its k-th key in scalar registers, so the steady state costs two LDS reads and two float
// compares per user and slab; the rare candidate above the k-th key is inserted by a one-lane
// shift.  The LDS slab (33 KB) is the only shared memory (registers, not LDS, bound the residency:
// 3 workgroups per CU).  Scores never reach global memory.  Chunks of one user are merged by
// topk_merge_kernel.
// Selection is on RAW scores (sigmoid is monotone, it is applied to the k winners only):
// order = (raw score descending, item id ascending) — a valid tie order for torch.topk.
// Masked train positives rank as -1: below every sigmoid output, and among raw scores as the
// value -1 itself (batch_test.py:65).
constexpr int FT_USERS = 64;
constexpr int FT_SLAB = 128;
constexpr int FT_LD = FT_SLAB + 4;
constexpr int FT_UPW = FT_USERS / (BLOCK / WAVE);  // users per wave
#ifndef IDG_TOPK_FLOOR_SLABS
#define IDG_TOPK_FLOOR_SLABS 3
#endif
constexpr int FLOOR_SLABS = IDG_TOPK_FLOOR_SLABS;   // slabs of a chunk the floor phase of a many-chunk call looks at (at most
                                                    // half the chunk; full yelp2018-size evaluation in calls of 1024 users:
                                                    // 2 / 3 / 4 / 5 slabs -> profiles/r03/topk_floor_slabs.txt)
constexpr int TOPK_WGS = 1024;                      // workgroups wanted per launch (3 fit a CU; measured best of 512..1536)

__device__ __forceinline__ unsigned long long readlane_u64(unsigned long long v, int src) {
  const uint32_t lo = (uint32_t)__builtin_amdgcn_readlane((int)(uint32_t)v, src);
  const uint32_t hi = (uint32_t)__builtin_amdgcn_readlane((int)(uint32_t)(v >> 32), src);
  return ((unsigned long long)hi << 32) | lo;
}
// value of lane - 1 (lane 0 receives an unspecified value)
__device__ __forceinline__ unsigned long long from_lane_below(unsigned long long v) {
  const uint32_t lo = (uint32_t)__builtin_amdgcn_update_dpp(0, (int)(uint32_t)v, 0x138 /* wave_shr:1 */, 0xf, 0xf, false);
  const uint32_t hi = (uint32_t)__builtin_amdgcn_update_dpp(0, (int)(uint32_t)(v >> 32), 0x138, 0xf, 0xf, false);
  return ((unsigned long long)hi << 32) | lo;
}
// `list` is descending over the lanes; the wave-uniform key c (different from every key in it) takes
// its place and everything below moves down one lane (the last key falls off).
__device__ __forceinline__ void list_insert(unsigned long long& list, unsigned long long c, int lane) {
  const unsigned long long up = from_lane_below(list);
  if (list < c) list = (lane == 0 || up > c) ? c : up;
}
// Feed the keys of this wave's lanes (0 = no candidate) that beat the k-th key into the list.
__device__ __forceinline__ void list_offer(unsigned long long& list, unsigned long long& tau, unsigned long long key,
                                           int k, int lane) {
  unsigned long long m = __ballot(key > tau);
  while (m) {
    const int src = __builtin_ctzll(m);
    list_insert(list, readlane_u64(key, src), lane);
    tau = readlane_u64(list, k - 1);
    m = (m & (m - 1)) & __ballot(key > tau);
  }
}
__device__ __forceinline__ float tau_floor(unsigned long long tau) { return tau ? key_score(tau) : -__builtin_inff(); }
// The floor a consumer PUBLISHES to the producers' flagging (flag_candidates compares raw accumulators, before masking).
// On raw scores a train item ranks as the value -1 whatever its raw score is, so while a user's floor is not above -1 a
// slab may hold a candidate the producers cannot see: publish -inf (every slab of that user is looked at) until the
// floor has passed -1.  After the sigmoid a masked entry ranks below every score and never is a candidate.
template <bool SIGMOID>
__device__ __forceinline__ float floor_published(float f) {
  return (!SIGMOID && f <= -1.0f) ? -__builtin_inff() : f;
}

// `gate` (nullable): the exact kernels are also enqueued behind every call of the threshold + collect form as its
// whole-call fall-back (idg_score_collect.inc) and run only when that call's scalars say so (`gate` points at them);
// otherwise a launch returns at once.  nullptr: an ordinary call.
__device__ __forceinline__ bool fallback_taken(const uint32_t* __restrict__ scalars);  // (idg_score_bf16.inc)
__device__ __forceinline__ bool gate_closed(const uint32_t* __restrict__ gate) { return gate != nullptr && !fallback_taken(gate); }

template <bool SIGMOID>
__global__ __launch_bounds__(BLOCK, 3) void score_topk_fused_kernel(const float* __restrict__ U,
                                                                 const float* __restrict__ V,
                                                                 const int64_t* __restrict__ users, int64_t Bt,
                                                                 int64_t I, int64_t d, int64_t chunk_items,
                                                                 const int64_t* __restrict__ excl_indptr,
                                                                 const int32_t* __restrict__ excl_items, int k,
                                                                 unsigned long long* __restrict__ partial,
                                                                 const unsigned long long* __restrict__ bound,
                                                                 const uint32_t* __restrict__ gate) {
  __shared__ float s_score[FT_USERS * FT_LD];
  if (gate_closed(gate)) return;

  const int tid = threadIdx.x, lane = tid % WAVE, wave = tid / WAVE;
  const int i = lane & 31, h = lane >> 5;
  const int64_t b0 = (int64_t)blockIdx.y * FT_USERS;
  const int64_t c_lo = (int64_t)blockIdx.x * chunk_items;
  const int64_t c_hi = c_lo + chunk_items < I ? c_lo + chunk_items : I;
  const int n_chunks = gridDim.x;
  const int64_t bu0 = b0 + i < Bt ? b0 + i : Bt - 1;
  const int64_t bu1 = b0 + 32 + i < Bt ? b0 + 32 + i : Bt - 1;
  const float* urow0 = U + users[bu0] * d;
  const float* urow1 = U + users[bu1] * d;
  const bool d4 = (d % 4 == 0);
  // Train-item masking (batch_test.py:62-65).  Exclusion lists are ascending.  Lane (mu, mj) = (lane / 4, lane % 4)
  // holds entry mj of a 4-entry window into the list of user mu of this wave, starting at the user's cursor (parked at
  // the first train item >= c_lo).  Per slab ONE compare pair finds the window entries inside the slab (their LDS
  // scores are overwritten before selection), one ballot counts how many entries each user has passed, cursors advance
  // and the windows of users that advanced are re-read; a user that used up all four entries may have more inside this
  // slab: the loop then goes round again.  (Round 1's 32-entry windows per user PAIR took 8 dependent compare / ballot
  // / branch rounds per slab: 40 % of the select's cycles, measured in the producer / consumer kernel below.)
  const int mu = lane >> 2, mj = lane & 3;
  int64_t m_cur = 0, m_end = 0;
  if (excl_indptr && b0 + FT_UPW * wave + mu < Bt) {
    const int64_t uid = users[b0 + FT_UPW * wave + mu];
    int64_t lo = excl_indptr[uid], hi = excl_indptr[uid + 1];
    m_end = hi;
    while (lo < hi) {
      const int64_t mid = (lo + hi) >> 1;
      if (excl_items[mid] < (int32_t)c_lo) lo = mid + 1;
      else hi = mid;
    }
    m_cur = lo;
  }
  int32_t m_win = (excl_indptr && m_cur + mj < m_end) ? excl_items[m_cur + mj] : 0x7fffffff;
  unsigned long long best[FT_UPW], tau[FT_UPW];  // per user of this wave: list (lane = rank) and its k-th key
#pragma unroll
  for (int uu = 0; uu < FT_UPW; ++uu) best[uu] = 0ull, tau[uu] = 0ull;
  // k > 64 (idg_score_topk_f32 runs one pass per 64 ranks): only keys strictly below the user's bound — the last
  // key the previous pass emitted — are candidates.  Lane uu keeps user uu's bound; no bound = every key passes.
  unsigned long long my_bound = ~0ull;
  if (bound && lane < FT_UPW && b0 + FT_UPW * wave + lane < Bt) my_bound = bound[b0 + FT_UPW * wave + lane];

  for (int64_t slab = c_lo; slab < c_hi; slab += FT_SLAB) {
    // ---- (A) scores of this wave's 32-item column tile for both 32-user row tiles
    {
      const int64_t j = slab + 32 * wave + i;
      const float* vrow = V + (j < I ? j : I - 1) * d;
      f32x16 acc0, acc1;
#pragma unroll
      for (int r = 0; r < 16; ++r) acc0[r] = 0.f, acc1[r] = 0.f;
      // 8 features of each operand at a time (same product order as score_dense_kernel: MFMA q of a 64-deep chunk
      // pairs feature kc+q with feature kc+32+q) keeps the operand registers at 24
#pragma unroll 1
      for (int64_t kq = 0; kq < (d + 63) / 64 * 64; kq += 16) {
        const int64_t k0 = (kq / 64) * 64 + 32 * h + (kq % 64) / 2;  // kq%64 in {0,16,32,48} -> offset {0,8,16,24}
        float a0[8], a1[8], bb[8];
        if (d4 && k0 + 8 <= d) {
#pragma unroll
          for (int q = 0; q < 2; ++q) {
            const float4 x = *reinterpret_cast<const float4*>(urow0 + k0 + 4 * q);
            const float4 y = *reinterpret_cast<const float4*>(urow1 + k0 + 4 * q);
            const float4 z = *reinterpret_cast<const float4*>(vrow + k0 + 4 * q);
            a0[4 * q + 0] = x.x, a0[4 * q + 1] = x.y, a0[4 * q + 2] = x.z, a0[4 * q + 3] = x.w;
            a1[4 * q + 0] = y.x, a1[4 * q + 1] = y.y, a1[4 * q + 2] = y.z, a1[4 * q + 3] = y.w;
            bb[4 * q + 0] = z.x, bb[4 * q + 1] = z.y, bb[4 * q + 2] = z.z, bb[4 * q + 3] = z.w;
          }
        } else {
#pragma unroll
          for (int q = 0; q < 8; ++q) {
            const bool in = k0 + q < d;
            a0[q] = in ? urow0[k0 + q] : 0.f;
            a1[q] = in ? urow1[k0 + q] : 0.f;
            bb[q] = in ? vrow[k0 + q] : 0.f;
          }
        }
#pragma unroll
        for (int q = 0; q < 8; ++q) {
          acc0 = __builtin_amdgcn_mfma_f32_32x32x2f32(a0[q], bb[q], acc0, 0, 0, 0);
          acc1 = __builtin_amdgcn_mfma_f32_32x32x2f32(a1[q], bb[q], acc1, 0, 0, 0);
        }
      }
      // C/D map: column = lane & 31, row = (r & 3) + 8 * (r >> 2) + 4 * (lane >> 5)
      __syncthreads();  // every wave is done selecting from the previous slab
#pragma unroll
      for (int r = 0; r < 16; ++r) {
        const int row = (r & 3) + 8 * (r >> 2) + 4 * h;
        s_score[row * FT_LD + 32 * wave + i] = acc0[r];
        s_score[(32 + row) * FT_LD + 32 * wave + i] = acc1[r];
      }
    }
    __syncthreads();
    // ---- (B) wave w owns users 16w .. 16w+15 of the block: mask their train items, then select
    const int64_t slab_end = slab + FT_SLAB < c_hi ? slab + FT_SLAB : c_hi;
    if (excl_indptr) {
      const float masked = SIGMOID ? -__builtin_inff() : -1.0f;  // ranks as the value -1 in either domain
      while (true) {
        const bool passed = m_win < (int32_t)slab_end;
        if (passed && m_win >= (int32_t)slab) s_score[(FT_UPW * wave + mu) * FT_LD + (m_win - (int32_t)slab)] = masked;
        const unsigned long long pm = __ballot(passed);
        if (pm == 0ull) break;  // nobody has a train item up to the end of this slab
        const int adv = __popc((uint32_t)(pm >> (4 * mu)) & 0xFu);  // entries user mu has passed
        m_cur += adv;
        const bool again = __ballot(adv == 4) != 0ull;  // a whole window used up: that user may have more in this slab
        if (adv > 0) m_win = m_cur + mj < m_end ? excl_items[m_cur + mj] : 0x7fffffff;
        if (!again) break;
      }
      __builtin_amdgcn_wave_barrier();
    }
    const bool in0 = slab + lane < c_hi, in1 = slab + 64 + lane < c_hi;
    if (slab == c_lo) {
      // first slab of the chunk: one 128-key sorting network per user fills its list
      for (int uu = 0; uu < FT_UPW; ++uu) {
        const int u = FT_UPW * wave + uu;
        unsigned long long k0 = in0 ? make_key(s_score[u * FT_LD + lane], (uint32_t)(slab + lane)) : 0ull;
        unsigned long long k1 = in1 ? make_key(s_score[u * FT_LD + 64 + lane], (uint32_t)(slab + 64 + lane)) : 0ull;
        if (bound) {
          const unsigned long long bd = readlane_u64(my_bound, uu);
          k0 = k0 < bd ? k0 : 0ull;
          k1 = k1 < bd ? k1 : 0ull;
        }
        wave_sort128_desc(k0, k1, lane);
        const unsigned long long t = readlane_u64(k0, k - 1);
#pragma unroll
        for (int q = 0; q < FT_UPW; ++q) {
          best[q] = q == uu ? k0 : best[q];
          tau[q] = q == uu ? t : tau[q];
        }
      }
    } else {
#pragma unroll
      for (int uu = 0; uu < FT_UPW; ++uu) {
        const int u = FT_UPW * wave + uu;
        const float s0 = s_score[u * FT_LD + lane], s1 = s_score[u * FT_LD + 64 + lane];
        // columns past the chunk hold scores of a clamped item: the exact key test below drops them
        const float floor_ = tau_floor(tau[uu]);  // score of the k-th key (-inf while the list is short)
        if (__ballot(!(s0 < floor_)) | __ballot(!(s1 < floor_))) {  // ("not below": a NaN score is a candidate, see make_key)
          unsigned long long c0 = in0 ? make_key(s0, (uint32_t)(slab + lane)) : 0ull;
          unsigned long long c1 = in1 ? make_key(s1, (uint32_t)(slab + 64 + lane)) : 0ull;
          if (bound) {
            const unsigned long long bd = readlane_u64(my_bound, uu);
            c0 = c0 < bd ? c0 : 0ull;
            c1 = c1 < bd ? c1 : 0ull;
          }
          list_offer(best[uu], tau[uu], c0, k, lane);
          list_offer(best[uu], tau[uu], c1, k, lane);
        }
      }
    }
  }
  // ---- publish this chunk's list per user
#pragma unroll
  for (int uu = 0; uu < FT_UPW; ++uu) {
    const int64_t b = b0 + FT_UPW * wave + uu;
    if (b < Bt) partial[(b * n_chunks + blockIdx.x) * 64 + lane] = best[uu];
  }
}

// ---- the same computation with the two phases on DIFFERENT waves (round 2) ------------------------------------
// In the kernel above every wave alternates between the matrix pipe (A) and the vector pipe (B), two barriers per
// slab, and the waves of a workgroup reach each phase together: measured, the matrix phase alone takes 2.04 of the
// 2.65 ms of a yelp2018-size evaluation and the select phase adds its 0.6 ms serially.  Here a 512-thread workgroup
// has four PRODUCER waves (one per SIMD: the 64 x 32 score tile of their column tile, written to one of TWO LDS
// slabs, plus one compare per accumulator register that flags the users with a possible candidate) and four CONSUMER
// waves (one per SIMD: masking + streaming select for 16 users each out of the other slab, flagged users only).
// One barrier per slab: barrier t says "slab t is written and slab t - 1 is consumed", so producers compute slab
// t + 1 while consumers select from slab t — the two pipes of a SIMD work at the same time.  Same MFMA sequence per
// score and same selection order as above: results are identical.  LDS 2 x 33 KB: two workgroups (16 waves) per CU.
// Where the time goes (cycle counters inside the kernel, profiles/r02/topk_roles.txt): the select is a chain of
// dependent scalar-ish steps — one instruction per ~20 cycles on its wave — and the consumers, not the producers,
// set the pace (producers wait at the barrier for 64 % of their cycles; alone they would finish in 1.5 ms).
constexpr int SP_BLOCK = 2 * BLOCK;
typedef float f32x4 __attribute__((ext_vector_type(4)));

// Producer side of the select: which users of the block have, among this wave's 32 items of the slab, a score that
// reaches their floor (the score of their k-th key as last published by their consumer wave — floors only rise, so a
// stale one flags too many users, never too few).  One compare per accumulator register (64 scores); consumers then
// look at flagged users only.  (Queues of (user, column) pairs filled here with LDS atomics, so that consumers read
// the listed columns only, were measured: the same 2.07 ms — the work just moves to the producers.)
__device__ __forceinline__ void flag_candidates(const f32x16& acc0, const f32x16& acc1, const float* s_floor,
                                                uint32_t* flag, int h) {
#pragma unroll
  for (int g = 0; g < 4; ++g) {
    const float4 f0 = *reinterpret_cast<const float4*>(s_floor + 8 * g + 4 * h);
    const float4 f1 = *reinterpret_cast<const float4*>(s_floor + 32 + 8 * g + 4 * h);
    const float fa[4] = {f0.x, f0.y, f0.z, f0.w}, fb[4] = {f1.x, f1.y, f1.z, f1.w};
#pragma unroll
    for (int c = 0; c < 4; ++c) {
      // ("not below the floor" rather than ">=": a NaN score ranks ABOVE everything — torch.topk's order, make_key — and
      //  must reach the select; the same instruction count)
      if (!(acc0[4 * g + c] < fa[c])) flag[8 * g + 4 * h + c] = 1u;
      if (!(acc1[4 * g + c] < fb[c])) flag[32 + 8 * g + 4 * h + c] = 1u;
    }
  }
}

// The consumer half of the producer / consumer kernel (score_topk_spec_kernel: exact fp32 scores): wave `wave` (0..3)
// owns users 16 wave .. 16 wave + 15 of the block's 64.
template <bool SIGMOID, bool MAXONLY>
__device__ __forceinline__ void topk_consume(float (*s_buf)[FT_USERS * FT_LD], float* s_floor, uint32_t (*s_flag)[FT_USERS],
                                             const int wave, const int lane, const int64_t* __restrict__ users, int64_t Bt,
                                             const int64_t b0, const int64_t c_lo, const int64_t c_hi, const int n_chunks,
                                             const int n_slabs, const int64_t* __restrict__ excl_indptr,
                                             const int32_t* __restrict__ excl_items, int k,
                                             unsigned long long* __restrict__ partial,
                                             const unsigned long long* __restrict__ bound, float* __restrict__ chunk_max,
                                             const float* __restrict__ floor0) {
  // ================= consumers: wave w owns users 16w .. 16w+15 of the block
#ifndef IDG_TOPK_CONS_PRIO
#define IDG_TOPK_CONS_PRIO 2
#endif
  // Vector issue on a SIMD is arbitrated by priority, then age: the consumers are the younger half of the workgroup
  // and would get what the producers' MFMA stream leaves over.  Their chains of short dependent instructions are what
  // the slab time hangs on, and an MFMA that issues a few cycles late costs the paced matrix pipe nothing.
  __builtin_amdgcn_s_setprio(IDG_TOPK_CONS_PRIO);
  // Train-item masking (batch_test.py:62-65).  Exclusion lists are ascending.  Lane (mu, mj) = (lane / 4, lane % 4)
  // holds entry mj of a 4-entry window into the list of user mu of this wave, starting at the user's cursor (parked at
  // the first train item >= c_lo).  Per slab ONE compare pair finds the window entries inside the slab (their LDS
  // scores are overwritten before selection), one ballot counts how many entries each user has passed (a prefix of
  // its window: ascending), cursors advance and the windows of users that advanced are re-read — a load whose result
  // is first needed at the next slab.  A user that used up all four entries may have more inside this slab: rare, the
  // loop then goes round again.  (The alternating kernel's 32-entry windows per user PAIR cost 8 dependent
  // compare / ballot / branch rounds per slab: 6,500 of this wave's ~16,000 cycles per slab, measured.)
  const int mu = lane >> 2, mj = lane & 3;
  int64_t m_cur = 0, m_end = 0;
  if (excl_indptr && b0 + FT_UPW * wave + mu < Bt) {
    const int64_t uid = users[b0 + FT_UPW * wave + mu];  // (the exclusion lists are indexed by user id)
    int64_t lo = excl_indptr[uid], hi = excl_indptr[uid + 1];
    m_end = hi;
    while (lo < hi) {
      const int64_t mid = (lo + hi) >> 1;
      if (excl_items[mid] < (int32_t)c_lo) lo = mid + 1;
      else hi = mid;
    }
    m_cur = lo;
  }
  int32_t m_win = (excl_indptr && m_cur + mj < m_end) ? excl_items[m_cur + mj] : 0x7fffffff;
  // Per user of this wave: the list (lane = rank).  Its k-th key is wave-uniform, but it is fetched where needed with
  // ds_bpermute into VECTOR registers on purpose: an fp32 MFMA runs on the SIMD's own FMA lanes (the fp32 matrix and
  // vector peaks are the same number), so while the producers' MFMAs execute no other vector instruction of this SIMD
  // does — every vector-class instruction of the select waits for an MFMA boundary (measured: halving the MFMAs takes
  // exactly their pipe time off the kernel; ~2,800 cycles for the ~85 vector instructions of one (user, slab) pair).
  // Scalar and LDS-class instructions do not wait, so the select is written to use them: thresholds broadcast with
  // ds_bpermute instead of v_readlane pairs, no scalar-register spills (v_readlane / v_writelane are vector-class),
  // keys built in three vector instructions.
  unsigned long long best[FT_UPW];
#pragma unroll
  for (int uu = 0; uu < FT_UPW; ++uu) best[uu] = 0ull;
  unsigned long long my_bound = ~0ull;
  if (bound && lane < FT_UPW && b0 + FT_UPW * wave + lane < Bt) my_bound = bound[b0 + FT_UPW * wave + lane];
  float my_floor = -__builtin_inff();  // lane uu: the floor user uu of this wave starts every chunk with (floor0)
  if (!MAXONLY && floor0 && lane < FT_UPW && b0 + FT_UPW * wave + lane < Bt) my_floor = floor0[b0 + FT_UPW * wave + lane];
  auto key_of = [](float sc, uint32_t not_item) {  // make_key with the item id already complemented
    const uint32_t ub = __float_as_uint(sc);
    const uint32_t hi = ub ^ ((uint32_t)((int32_t)ub >> 31) | 0x80000000u);
    return ((unsigned long long)hi << 32) | not_item;
  };
  auto floor_of = [](unsigned long long tk) { return tk ? key_score(tk) : -__builtin_inff(); };
  // feed the keys of this wave's lanes (0 = none) that beat the k-th key into the list; true if the list changed
  auto offer = [&](unsigned long long& list, unsigned long long& tk, unsigned long long key) {
    unsigned long long m = __ballot(key > tk);
    const bool any = m != 0ull;
    while (m) {
      const int src = __builtin_ctzll(m);
      list_insert(list, readlane_u64(key, src), lane);
      tk = shfl_u64(list, k - 1);
      m = (m & (m - 1)) & __ballot(key > tk);
    }
    return any;
  };

  for (int t = 0; t < n_slabs; ++t) {
    const int64_t slab = c_lo + (int64_t)t * FT_SLAB;
    __syncthreads();  // barrier t: slab t is complete in buffer t & 1
    float* s_score = s_buf[t & 1];
    const int64_t slab_end = slab + FT_SLAB < c_hi ? slab + FT_SLAB : c_hi;
    if (excl_indptr) {
      const float masked = SIGMOID ? -__builtin_inff() : -1.0f;  // ranks as the value -1 in either domain
      while (true) {
        const bool passed = m_win < (int32_t)slab_end;
        if (passed && m_win >= (int32_t)slab) s_score[(FT_UPW * wave + mu) * FT_LD + (m_win - (int32_t)slab)] = masked;
        const unsigned long long pm = __ballot(passed);
        if (pm == 0ull) break;  // nobody has a train item up to the end of this slab
        const int adv = __popc((uint32_t)(pm >> (4 * mu)) & 0xFu);  // entries user mu has passed
        m_cur += adv;
        const bool again = __ballot(adv == 4) != 0ull;  // a whole window used up: that user may have more in this slab
        if (adv > 0) m_win = m_cur + mj < m_end ? excl_items[m_cur + mj] : 0x7fffffff;
        if (!again) break;
      }
      __builtin_amdgcn_wave_barrier();
    }
    const bool in0 = slab + lane < c_hi, in1 = slab + 64 + lane < c_hi;
    const uint32_t not_item0 = ~(uint32_t)(slab + lane), not_item1 = ~(uint32_t)(slab + 64 + lane);
    if (MAXONLY) {
      // phase one of a many-chunk call: the best score per user over the chunk's first FLOOR_SLABS slabs, nothing else
      // (lane uu keeps user uu's running maximum)
      for (int uu = 0; uu < FT_UPW; ++uu) {
        const int u = FT_UPW * wave + uu;
        float m = fmaxf(in0 ? s_score[u * FT_LD + lane] : -__builtin_inff(), in1 ? s_score[u * FT_LD + 64 + lane] : -__builtin_inff());
#pragma unroll
        for (int o = 32; o > 0; o >>= 1) m = fmaxf(m, __shfl_xor(m, o, WAVE));
        if (lane == uu) my_floor = fmaxf(my_floor, m);
      }
      if (t + 1 == n_slabs && lane < FT_UPW && b0 + FT_UPW * wave + lane < Bt)
        chunk_max[(b0 + FT_UPW * wave + lane) * n_chunks + blockIdx.x] = my_floor;
      continue;
    }
    if (t == 0 && !floor0) {
      // first slab of the chunk: one 128-key sorting network per user fills its list
      for (int uu = 0; uu < FT_UPW; ++uu) {
        const int u = FT_UPW * wave + uu;
        unsigned long long k0 = in0 ? key_of(s_score[u * FT_LD + lane], not_item0) : 0ull;
        unsigned long long k1 = in1 ? key_of(s_score[u * FT_LD + 64 + lane], not_item1) : 0ull;
        if (bound) {
          const unsigned long long bd = shfl_u64(my_bound, uu);
          k0 = k0 < bd ? k0 : 0ull;
          k1 = k1 < bd ? k1 : 0ull;
        }
        wave_sort128_desc(k0, k1, lane);
        const unsigned long long tk = shfl_u64(k0, k - 1);
        if (lane == 0) s_floor[u] = floor_published<SIGMOID>(floor_of(tk));
#pragma unroll
        for (int q = 0; q < FT_UPW; ++q) best[q] = q == uu ? k0 : best[q];
      }
      if (lane < FT_UPW) s_flag[0][FT_UPW * wave + lane] = 0u;  // (slab 0 flagged everybody: floors were -inf)
      continue;
    }
    // the producers have flagged the users that may have a candidate in this slab (flag_candidates): one LDS read for
    // the wave's users; the flags are cleared for the slab after next, which reuses this buffer
    uint32_t fl_ = 0u;
    if (lane < FT_UPW) {
      fl_ = s_flag[t & 1][FT_UPW * wave + lane];
      s_flag[t & 1][FT_UPW * wave + lane] = 0u;
    }
    const uint32_t cand = (uint32_t)__ballot(fl_ != 0u);
#if defined(IDG_TOPK_PROBE) && IDG_TOPK_PROBE == 1  // (timing probe 1, wrong results: the producers on their own)
    continue;
#endif
    if (cand == 0) continue;
#pragma unroll
    for (int uu = 0; uu < FT_UPW; ++uu) {
      if (!((cand >> uu) & 1u)) continue;
      const int u = FT_UPW * wave + uu;
      const float s0 = s_score[u * FT_LD + lane], s1 = s_score[u * FT_LD + 64 + lane];
      // columns past the chunk hold scores of a clamped item: in0 / in1 drop them.  No float pre-filter: the keys
      // are three vector instructions each and the offer's own first compare is the filter
      unsigned long long c0 = in0 ? key_of(s0, not_item0) : 0ull;
      unsigned long long c1 = in1 ? key_of(s1, not_item1) : 0ull;
      if (bound) {
        const unsigned long long bd = shfl_u64(my_bound, uu);
        c0 = c0 < bd ? c0 : 0ull;
        c1 = c1 < bd ? c1 : 0ull;
      }
      float fl0 = -__builtin_inff();
      if (floor0) {  // nothing below the user's starting floor can reach the final list (the chunk's list may stay short)
        fl0 = __shfl(my_floor, uu, WAVE);
        c0 = s0 < fl0 ? 0ull : c0;  // (a NaN stays: it ranks above everything)
        c1 = s1 < fl0 ? 0ull : c1;
      }
      unsigned long long tk = shfl_u64(best[uu], k - 1);
      const bool ch0 = offer(best[uu], tk, c0);
      const bool ch1 = offer(best[uu], tk, c1);
      if ((ch0 || ch1) && lane == 0) s_floor[u] = floor_published<SIGMOID>(fmaxf(fl0, floor_of(tk)));
    }
  }
#pragma unroll
  for (int uu = 0; uu < FT_UPW; ++uu) {
    const int64_t b = b0 + FT_UPW * wave + uu;
    if (b < Bt) partial[(b * n_chunks + blockIdx.x) * 64 + lane] = best[uu];
  }
}

// Calls with few user tiles cut the catalogue into many short chunks (to fill the chip), and a chunk's list starts
// empty: while it is young nearly every (user, slab) pair holds a candidate and pays the ~2,800-cycle select — at 1024
// users per call (the reference's test_batch_size shape, batch_test.py:52-68) that was 190 of a launch's 240 us.  So
// such calls run in two phases.  MAXONLY: every (tile, chunk) workgroup scores the first few slabs of its chunk and writes
// each user's maximum (train items masked) to chunk_max[b, chunk]; chunk_floor_kernel takes the k-th largest of a
// user's chunk maxima — k items score at least that, so it is a lower bound of the final k-th score — and the main
// launch starts every chunk with that floor (floor0): an item below it is never a candidate, lists start empty, and a
// (user, slab) pair is looked at only when a score reaches the floor.  Same final lists (ties at the floor are kept).
template <bool SIGMOID, bool D64, bool MAXONLY = false>
__global__ __launch_bounds__(SP_BLOCK, 4) void score_topk_spec_kernel(const float* __restrict__ U,
                                                                    const float* __restrict__ V,
                                                                    const int64_t* __restrict__ users, int64_t Bt,
                                                                    int64_t I, int64_t d, int64_t chunk_items,
                                                                    const int64_t* __restrict__ excl_indptr,
                                                                    const int32_t* __restrict__ excl_items, int k,
                                                                    unsigned long long* __restrict__ partial,
                                                                    const unsigned long long* __restrict__ bound,
                                                                    float* __restrict__ chunk_max,
                                                                    const float* __restrict__ floor0,
                                                                    const uint32_t* __restrict__ gate) {
  __shared__ float s_buf[2][FT_USERS * FT_LD];
  __shared__ __attribute__((aligned(16))) float s_floor[FT_USERS];  // per user: score of its k-th key (consumers publish)
  __shared__ uint32_t s_flag[2][FT_USERS];                          // per slab buffer and user: a candidate may exist
  if (gate_closed(gate)) return;

  const int tid = threadIdx.x, lane = tid % WAVE;
  const int wave8 = __builtin_amdgcn_readfirstlane(tid / WAVE);  // scalar: the role branch below is wave-uniform
  const int64_t b0 = (int64_t)blockIdx.y * FT_USERS;
  if (tid < FT_USERS) {
    s_floor[tid] = (!MAXONLY && floor0 && b0 + tid < Bt) ? floor_published<SIGMOID>(floor0[b0 + tid]) : -__builtin_inff();
    s_flag[0][tid] = 0u, s_flag[1][tid] = 0u;
  }
  __syncthreads();
  const int i = lane & 31, h = lane >> 5;
  const int64_t c_lo = (int64_t)blockIdx.x * chunk_items;
  const int64_t c_hi = c_lo + chunk_items < I ? c_lo + chunk_items : I;
  const int n_chunks = gridDim.x;
  int n_slabs = c_hi > c_lo ? (int)((c_hi - c_lo + FT_SLAB - 1) / FT_SLAB) : 0;
  if (MAXONLY && n_slabs > k) n_slabs = k;  // (MAXONLY launches pass the number of slabs to look at where k would be)

  if (wave8 < BLOCK / WAVE) {
    // ================= producers: wave p computes items slab + 32p .. slab + 32p + 31 for the block's 64 users
    const int wave = wave8;
    const int64_t bu0 = b0 + i < Bt ? b0 + i : Bt - 1;
    const int64_t bu1 = b0 + 32 + i < Bt ? b0 + 32 + i : Bt - 1;
    const float* urow0 = U + users[bu0] * d;
    const float* urow1 = U + users[bu1] * d;
    const bool d4 = (d % 4 == 0);
    if (D64) {
      // d = 64: this lane's 32 features of its two users stay in registers for the whole chunk, and the item operand
      // arrives 8 features at a time, one piece AHEAD of the MFMAs that use it (the piece after a slab's last is the
      // next slab's first, so it is in flight across the LDS writes and the barrier).  Per slab the wave then issues
      // 8 loads instead of 24: every float4 load here touches 64 cache lines (one row per lane), and the address
      // path of the CU, not the matrix pipe, was what bounded the three-operand form.
      float a0[32], a1[32];
#pragma unroll
      for (int q = 0; q < 8; ++q) {
        const float4 x = *reinterpret_cast<const float4*>(urow0 + 32 * h + 4 * q);
        const float4 y = *reinterpret_cast<const float4*>(urow1 + 32 * h + 4 * q);
        a0[4 * q + 0] = x.x, a0[4 * q + 1] = x.y, a0[4 * q + 2] = x.z, a0[4 * q + 3] = x.w;
        a1[4 * q + 0] = y.x, a1[4 * q + 1] = y.y, a1[4 * q + 2] = y.z, a1[4 * q + 3] = y.w;
      }
      const int64_t j0 = c_lo + 32 * wave + i;
      const float* vnext = V + (j0 < I ? j0 : I - 1) * 64 + 32 * h;
      // The item loads are issued by hand (asm) and waited for by hand: left to the compiler, each load is scheduled
      // right before the MFMAs that use it (it reuses the registers) and the wave waits out the whole latency every
      // 8 MFMAs.  Two register pairs alternate; the wait before a piece's MFMAs leaves the two newest loads in flight.
      f32x4 pa0, pa1, pb0, pb1;
#define IDG_LD(dst, ptr, off) asm volatile("global_load_dwordx4 %0, %1, off offset:" #off : "=v"(dst) : "v"(ptr) : "memory")
#define IDG_WAIT2(x, y) asm volatile("s_waitcnt vmcnt(2)" : "+v"(x), "+v"(y) : : "memory")
#define IDG_MM(A0, A1, B)                                               \
  acc0 = __builtin_amdgcn_mfma_f32_32x32x2f32(A0, B, acc0, 0, 0, 0);   \
  acc1 = __builtin_amdgcn_mfma_f32_32x32x2f32(A1, B, acc1, 0, 0, 0)
#define IDG_PIECE(pc, X0, X1)                                                                                        \
  IDG_MM(a0[8 * pc + 0], a1[8 * pc + 0], X0.x); IDG_MM(a0[8 * pc + 1], a1[8 * pc + 1], X0.y);                        \
  IDG_MM(a0[8 * pc + 2], a1[8 * pc + 2], X0.z); IDG_MM(a0[8 * pc + 3], a1[8 * pc + 3], X0.w);                        \
  IDG_MM(a0[8 * pc + 4], a1[8 * pc + 4], X1.x); IDG_MM(a0[8 * pc + 5], a1[8 * pc + 5], X1.y);                        \
  IDG_MM(a0[8 * pc + 6], a1[8 * pc + 6], X1.z); IDG_MM(a0[8 * pc + 7], a1[8 * pc + 7], X1.w)
      // the user operands have arrived before the first hand-issued load goes out: empty asm statements that "use"
      // them make the compiler wait for its own loads HERE (it does not count the asm loads; left alone it sinks the
      // operand loads to the loop entry and re-waits for them, down to vmcnt(0), in every trip)
#define IDG_PIN8(a, o) asm volatile("" : "+v"(a[o]), "+v"(a[o + 1]), "+v"(a[o + 2]), "+v"(a[o + 3]), "+v"(a[o + 4]), "+v"(a[o + 5]), "+v"(a[o + 6]), "+v"(a[o + 7]))
      IDG_PIN8(a0, 0); IDG_PIN8(a0, 8); IDG_PIN8(a0, 16); IDG_PIN8(a0, 24);
      IDG_PIN8(a1, 0); IDG_PIN8(a1, 8); IDG_PIN8(a1, 16); IDG_PIN8(a1, 24);
#undef IDG_PIN8
      IDG_LD(pa0, vnext, 0);
      IDG_LD(pa1, vnext, 16);
      for (int t = 0; t < n_slabs; ++t) {
        const float* vrow = vnext;
        {
          const int64_t jn = j0 + (int64_t)(t + 1) * FT_SLAB;
          vnext = V + (jn < I ? jn : I - 1) * 64 + 32 * h;
        }
        f32x16 acc0, acc1;
#pragma unroll
        for (int r = 0; r < 16; ++r) acc0[r] = 0.f, acc1[r] = 0.f;
        IDG_LD(pb0, vrow, 32);
        IDG_LD(pb1, vrow, 48);
        IDG_WAIT2(pa0, pa1);
        IDG_PIECE(0, pa0, pa1);
        IDG_LD(pa0, vrow, 64);
        IDG_LD(pa1, vrow, 80);
        IDG_WAIT2(pb0, pb1);
        IDG_PIECE(1, pb0, pb1);
        IDG_LD(pb0, vrow, 96);
        IDG_LD(pb1, vrow, 112);
        IDG_WAIT2(pa0, pa1);
#if !defined(IDG_TOPK_PROBE) || IDG_TOPK_PROBE != 4  // (timing probe 4, wrong results: half the MFMAs; profiles/r02/topk_roles.txt)
        IDG_PIECE(2, pa0, pa1);
#endif
        IDG_LD(pa0, vnext, 0);  // (past the last slab: a clamped, valid row nobody uses)
        IDG_LD(pa1, vnext, 16);
        IDG_WAIT2(pb0, pb1);
#if !defined(IDG_TOPK_PROBE) || IDG_TOPK_PROBE != 4
        IDG_PIECE(3, pb0, pb1);
#endif
        float* s_score = s_buf[t & 1];
#pragma unroll
        for (int r = 0; r < 16; ++r) {
          const int row = (r & 3) + 8 * (r >> 2) + 4 * h;
          s_score[row * FT_LD + 32 * wave + i] = acc0[r];
          s_score[(32 + row) * FT_LD + 32 * wave + i] = acc1[r];
        }
        flag_candidates(acc0, acc1, s_floor, s_flag[t & 1], h);
        __syncthreads();  // barrier t
      }
      asm volatile("s_waitcnt vmcnt(0)" : "+v"(pa0), "+v"(pa1) : : "memory");
#undef IDG_LD
#undef IDG_WAIT2
#undef IDG_MM
#undef IDG_PIECE
      return;
    }
    for (int t = 0; t < n_slabs; ++t) {
      const int64_t slab = c_lo + (int64_t)t * FT_SLAB;
      const int64_t j = slab + 32 * wave + i;
      const float* vrow = V + (j < I ? j : I - 1) * d;
      f32x16 acc0, acc1;
#pragma unroll
      for (int r = 0; r < 16; ++r) acc0[r] = 0.f, acc1[r] = 0.f;
#pragma unroll 1
      for (int64_t kq = 0; kq < (d + 63) / 64 * 64; kq += 16) {
        const int64_t k0 = (kq / 64) * 64 + 32 * h + (kq % 64) / 2;  // kq%64 in {0,16,32,48} -> offset {0,8,16,24}
        float a0[8], a1[8], bb[8];
        if (d4 && k0 + 8 <= d) {
#pragma unroll
          for (int q = 0; q < 2; ++q) {
            const float4 x = *reinterpret_cast<const float4*>(urow0 + k0 + 4 * q);
            const float4 y = *reinterpret_cast<const float4*>(urow1 + k0 + 4 * q);
            const float4 z = *reinterpret_cast<const float4*>(vrow + k0 + 4 * q);
            a0[4 * q + 0] = x.x, a0[4 * q + 1] = x.y, a0[4 * q + 2] = x.z, a0[4 * q + 3] = x.w;
            a1[4 * q + 0] = y.x, a1[4 * q + 1] = y.y, a1[4 * q + 2] = y.z, a1[4 * q + 3] = y.w;
            bb[4 * q + 0] = z.x, bb[4 * q + 1] = z.y, bb[4 * q + 2] = z.z, bb[4 * q + 3] = z.w;
          }
        } else {
#pragma unroll
          for (int q = 0; q < 8; ++q) {
            const bool in = k0 + q < d;
            a0[q] = in ? urow0[k0 + q] : 0.f;
            a1[q] = in ? urow1[k0 + q] : 0.f;
            bb[q] = in ? vrow[k0 + q] : 0.f;
          }
        }
#pragma unroll
        for (int q = 0; q < 8; ++q) {
          acc0 = __builtin_amdgcn_mfma_f32_32x32x2f32(a0[q], bb[q], acc0, 0, 0, 0);
          acc1 = __builtin_amdgcn_mfma_f32_32x32x2f32(a1[q], bb[q], acc1, 0, 0, 0);
        }
      }
      // slab t goes to buffer t & 1, last read (slab t - 2) before barrier t - 1, which this wave has passed
      float* s_score = s_buf[t & 1];
#pragma unroll
      for (int r = 0; r < 16; ++r) {
        const int row = (r & 3) + 8 * (r >> 2) + 4 * h;
        s_score[row * FT_LD + 32 * wave + i] = acc0[r];
        s_score[(32 + row) * FT_LD + 32 * wave + i] = acc1[r];
      }
      flag_candidates(acc0, acc1, s_floor, s_flag[t & 1], h);
      __syncthreads();  // barrier t
    }
    return;
  }

  // ================= consumers: wave w owns users 16w .. 16w+15 of the block
  topk_consume<SIGMOID, MAXONLY>(s_buf, s_floor, s_flag, wave8 - BLOCK / WAVE, lane, users, Bt, b0, c_lo, c_hi, n_chunks, n_slabs,
                                 excl_indptr, excl_items, k, partial, bound, chunk_max, floor0);
}

// one wave per batch row: fold the per-chunk lists (each holds its chunk's true top-k in lanes < k),
// emit ids and values
template <bool SIGMOID>
__global__ __launch_bounds__(BLOCK) void topk_merge_kernel(const unsigned long long* __restrict__ partial, int64_t Bt,
                                                           int n_chunks, int k, int64_t* __restrict__ out_idx,
                                                           float* __restrict__ out_val, int64_t ld_out, int64_t col0,
                                                           unsigned long long* __restrict__ bound_out,
                                                           unsigned long long* __restrict__ keys_out,
                                                           const uint32_t* __restrict__ gate) {
  const int lane = threadIdx.x % WAVE;
  const int64_t b = (int64_t)blockIdx.x * (BLOCK / WAVE) + threadIdx.x / WAVE;
  if (b >= Bt || gate_closed(gate)) return;
  unsigned long long best = partial[(b * n_chunks) * 64 + lane];
  unsigned long long tau = readlane_u64(best, k - 1);
  // the chunk lists are read EIGHT at a time (independent loads, one latency per group) and then offered in chunk order:
  // the offers' data-dependent loops keep the compiler from hoisting a load over them, and a call of 1024 users has ~30
  // lists per user (one dependent L2 round trip each: 15.6 of the call's 142 us before, round 4)
  for (int c0 = 1; c0 < n_chunks; c0 += 8) {
    unsigned long long a[8];
#pragma unroll
    for (int q = 0; q < 8; ++q) a[q] = (lane < k && c0 + q < n_chunks) ? partial[(b * n_chunks + c0 + q) * 64 + lane] : 0ull;
#pragma unroll
    for (int q = 0; q < 8; ++q) list_offer(best, tau, a[q], k, lane);
  }
  if (keys_out) {  // (the candidate list of the split-bf16 pre-filter: raw keys, 0 = no candidate)
    keys_out[b * 64 + lane] = lane < k ? best : 0ull;
    return;
  }
  if (lane < k) {
    out_idx[b * ld_out + col0 + lane] = (int64_t)key_item(best);
    if (out_val) {
      float s = key_score(best);
      if (SIGMOID) s = s == -__builtin_inff() ? -1.0f : sigmoidf_(s);
      out_val[b * ld_out + col0 + lane] = s;
    }
  }
  if (bound_out && lane == k - 1) bound_out[b] = best;  // ranks col0 + k ... of the next pass lie strictly below this key
}

// floor0[b] = the k-th largest of user b's n_chunks chunk maxima (n_chunks >= k): one wave per user, rank by counting
__global__ __launch_bounds__(BLOCK) void chunk_floor_kernel(const float* __restrict__ chunk_max, int64_t Bt, int n_chunks, int k,
                                                           float* __restrict__ floor0, const uint32_t* __restrict__ gate) {
  const int lane = threadIdx.x % WAVE;
  const int64_t b = (int64_t)blockIdx.x * (BLOCK / WAVE) + threadIdx.x / WAVE;
  if (b >= Bt || gate_closed(gate)) return;
  const float* m = chunk_max + b * n_chunks;
  // value v is the k-th largest iff fewer than k values are greater and at least k are greater or equal
  float found = -__builtin_inff();
  for (int j0 = 0; j0 < n_chunks; j0 += WAVE) {
    const int j = j0 + lane;
    const float v = j < n_chunks ? m[j] : -__builtin_inff();
    int gt = 0, ge = 0;
    for (int q = 0; q < n_chunks; ++q) {
      const float w = m[q];
      gt += w > v, ge += w >= v;
    }
    const bool is_kth = j < n_chunks && gt < k && ge >= k;
    const unsigned long long hit = __ballot(is_kth);
    if (hit) found = __shfl(v, __builtin_ctzll(hit), WAVE);
  }
  if (lane == 0) floor0[b] = found;
}

#include "idg_score_bf16.inc"
#include "idg_score_collect.inc"

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

// ---- knobs of the form choice (testing / tuning).  Read from the environment ONCE, by the first call that needs them
// (IDG_TOPK_FORM, IDG_TOPK_COLLECT, IDG_TOPK_FLOOR, IDG_TOPK_WGS, IDG_TOPK_CHUNKS, IDG_TOPK_FALLBACK_PERMILLE);
// idg_score_topk_option changes them afterwards.  (Until round 6 every call read the environment three to five times.)
struct TopkOptions {
  int64_t v[IDG_TOPK_OPT_COUNT];
};
static TopkOptions topk_defaults() {
  TopkOptions o{};
  o.v[IDG_TOPK_OPT_FORM] = -1;      // -1: by geometry; 0 / 1 / 3 forces a kernel (3: where its domain allows)
  o.v[IDG_TOPK_OPT_COLLECT] = 1;    // 0: never the threshold + collect form
  o.v[IDG_TOPK_OPT_FLOOR] = 1;      // 0: many-chunk calls of form 1 without their floor phase
  o.v[IDG_TOPK_OPT_WGS] = 0;        // > 0: workgroups wanted per launch (chunks = wgs / user tiles)
  o.v[IDG_TOPK_OPT_CHUNKS] = 0;     // > 0: catalogue chunks
  o.v[IDG_TOPK_OPT_FALLBACK_PERMILLE] = 20;  // form 3 falls back to the exact form as a whole beyond this share of unservable users; < 0: never
  return o;
}
static TopkOptions& topk_options() {
  static TopkOptions opt = [] {
    TopkOptions o = topk_defaults();
    static const char* const names[IDG_TOPK_OPT_COUNT] = {"IDG_TOPK_FORM",   "IDG_TOPK_COLLECT", "IDG_TOPK_FLOOR",
                                                          "IDG_TOPK_WGS",    "IDG_TOPK_CHUNKS",  "IDG_TOPK_FALLBACK_PERMILLE"};
    for (int i = 0; i < IDG_TOPK_OPT_COUNT; ++i)
      if (const char* v = std::getenv(names[i]))
        if (*v) o.v[i] = std::atoll(v);
    return o;
  }();
  return opt;
}

// Fused path geometry: 64 users per workgroup, the catalogue cut into n_chunks so that the grid has about TOPK_WGS
// workgroups (three are resident per CU, what the register budget allows); scratch = one best-64 list per (user, chunk).
// Which kernel, and into how many catalogue chunks.  The producer / consumer kernel pays a sort and a burst of list
// insertions at the start of every chunk, so it wants the FEWEST chunks that still fill its 512 workgroup slots (2 per
// CU) — measured at yelp2018 size, all users in one call: 1 chunk 1.95 ms, 2 chunks 2.41, 3 chunks 2.68 — and it pays
// from about 16 user tiles on (calls of 1024 users: 7.9 vs 8.4 ms for the 31,668 users; of 4096: 3.9 vs 5.2 ms);
// below that the two kernels tie at the launch floor (calls of 100 users, the reference's test_batch_size: 66 vs 69 ms
// for 317 calls) and the alternating kernel, with its ~1024 short workgroups, is kept.
static inline bool floor_phase(int form, int nc, int64_t ci, int k);

// form 3 (round 5): threshold + collect + exact finish on bf16 bound scores (idg_score_collect.inc, idg_score_bf16.inc):
// d = 64, 128 or 256, one pass, a catalogue of at least two slabs per sampled group, calls of at least COLLECT_MIN_TILES user
// tiles (more than 256 tiles: ONE catalogue chunk, every CU has its own tiles; fewer: the catalogue is cut so that the grid
// fills the chip, collect_chunks).  (VERDICT r04's split-bf16 pre-filter on the streaming select — "form 2" — was built,
// measured slower than form 1 and removed: profiles/r05/topk_bf16_probe.txt, HISTORY.md.)
constexpr int COLLECT_MIN_TILES = 8;  // (512 / 8 = 64 chunks: the finish reads one count per lane)
static inline bool collect_domain(int64_t user_tiles, int64_t I, int64_t d, int k) {
  // (k <= a third of the sampled groups: the floor is the k-th largest of 64 / 128 half-slab maxima — at k = 20 of 64 about
  //  220 items pass it; as k approaches the group count it falls to the smallest maximum and the candidate lists overflow)
  return user_tiles >= COLLECT_MIN_TILES && (d == 64 || d == 128 || d == 256) && k >= 1 && 3 * k <= COLLECT_GROUPS_MAX + 2 &&
         I >= (int64_t)2 * COLLECT_GROUPS_MAX * FT_SLAB;
}
// chunks of a form-3 call: two workgroups per CU's worth of them, a multiple of 8 where there are that many (workgroup ->
// XCD goes round robin over the linear id, chunk fastest: the tiles of one chunk then share an L2), chunks of >= 8 slabs
static inline int64_t collect_chunks(int64_t user_tiles, int64_t I) {
  int64_t nc = user_tiles > 256 ? 1 : 512 / user_tiles;
  const int64_t most = ((I + FT_SLAB - 1) / FT_SLAB) / 8;
  nc = nc > most ? most : nc;
  nc = nc > 64 ? 64 : nc;
  if (nc >= 8) nc = nc / 8 * 8;
  return nc < 1 ? 1 : nc;
}
// candidate keys a (user, chunk) segment holds: the whole list when there is one chunk, else a share with room for skew
static inline int collect_cap(int64_t I) { return I >= 1000000 ? COLLECT_CAP_BIG : COLLECT_CAP; }
static inline int collect_cap_chunk(int nc, int64_t I) {
  const int cap = collect_cap(I);
  return nc == 1 ? cap : std::max(cap / 8, 2 * cap / nc);
}

// fallback = true: the geometry of the exact form that form 3's whole-call fall-back launches for this call — no form 3,
// and no floor phase (two gated launches behind every ordinary call instead of four), hence no bump to k + 2 chunks either
static inline void fused_geometry(int64_t Bt, int64_t I, int* n_chunks, int64_t* chunk_items, int* form_out = nullptr, int k = 0,
                                  int64_t d = 0, bool fallback = false) {
  const TopkOptions& opt = topk_options();
  const int64_t user_tiles = (Bt + FT_USERS - 1) / FT_USERS;
  const int forced = (int)opt.v[IDG_TOPK_OPT_FORM];
  int form = forced >= 0 ? forced : (user_tiles >= 16 ? 1 : 0);
  if (forced < 0 && k > 0 && opt.v[IDG_TOPK_OPT_COLLECT] != 0 && collect_domain(user_tiles, I, d, k)) form = 3;
  if (form == 2) form = 1;  // (the removed pre-filter form)
  if (form == 3 && !collect_domain(user_tiles, I, d, k)) form = 1;
  if (form == 3 && fallback) form = user_tiles >= 16 ? 1 : 0;
  if (form_out) *form_out = form;
  const int64_t tuned = form == 3 ? collect_chunks(user_tiles, I) : form == 1 ? (2 * 256) / user_tiles : (TOPK_WGS + user_tiles - 1) / user_tiles;
  const int64_t max_nc = (I + 1023) / 1024;
  auto finish = [&](int64_t nc, int64_t* ci_out) {
    if (opt.v[IDG_TOPK_OPT_WGS] > 0) nc = (opt.v[IDG_TOPK_OPT_WGS] + user_tiles - 1) / user_tiles;
    if (opt.v[IDG_TOPK_OPT_CHUNKS] > 0) nc = opt.v[IDG_TOPK_OPT_CHUNKS];
    nc = nc < 1 ? 1 : (nc > max_nc ? max_nc : nc);
    // (form 3's producers walk slabs in pairs: an even number per chunk leaves only the last chunk a phantom slab)
    const int64_t unit = form == 3 ? 2 * FT_SLAB : FT_SLAB;
    const int64_t ci = ((I + nc - 1) / nc + unit - 1) / unit * unit;
    *ci_out = ci;
    return (I + ci - 1) / ci;
  };
  int64_t ci, nc = finish(tuned, &ci);
  // A many-chunk call of the producer / consumer kernel starts its chunks from a floor that needs at least k chunks
  // (floor_phase): a few more, shorter chunks cost less than doing without it — but only when the FINAL geometry (after
  // the clamps) still qualifies; otherwise the tuned count stands (a bumped count that the clamp brings back
  // below k, or chunks of fewer than four slabs, would run the many-short-chunks geometry WITHOUT a floor, and the
  // partial-list workspace grows with the chunk count: calls of 17..256 user tiles at k + 2 chunks up to ~16x).
  // Measured where it pays: calls of 1024 users (16 tiles) at yelp2018 / amazon-book size; beyond 64 user tiles the
  // tuned count is already <= 8 chunks and the start-up the floor saves is a small share of a launch.
  if (form == 1 && !fallback && tuned > 1 && tuned < k + 2 && k <= 64 && user_tiles <= 64) {
    int64_t ci2;
    const int64_t nc2 = finish(k + 2, &ci2);
    if (floor_phase(form, (int)nc2, ci2, k)) nc = nc2, ci = ci2;
  }
  *n_chunks = (int)nc;
  *chunk_items = ci;
}

// Two-phase form (chunk maxima -> per-user starting floor): the producer / consumer kernel, one pass (k <= 64), at least
// k chunks (the floor is the k-th largest of the chunk maxima) of at least two slabs.  IDG_TOPK_OPT_FLOOR = 0 turns it off.
static_assert(FLOOR_SLABS >= 1, "IDG_TOPK_FLOOR_SLABS must be >= 1: chunk_floor_kernel reads what the floor phase wrote");
static inline bool floor_phase(int form, int nc, int64_t ci, int k) {
  if (topk_options().v[IDG_TOPK_OPT_FLOOR] == 0) return false;
  return form == 1 && k <= 64 && nc >= k && ci >= 4 * FT_SLAB;
}

// scratch of the exact forms: one best-64 list per (user, chunk) + (k > 64 only) one bound key per user between the passes
// + (two-phase form) one maximum per (user, chunk) and one starting floor per user
static inline size_t exact_ws_bytes(int64_t Bt, int nc, int64_t ci, int form, int k, bool with_floor = true) {
  return (size_t)Bt * (size_t)nc * 64 * sizeof(unsigned long long) + (k > 64 ? (size_t)Bt * sizeof(unsigned long long) : 0) +
         (with_floor && floor_phase(form, nc, ci, k) ? ((size_t)Bt * (size_t)nc + (size_t)Bt) * sizeof(float) : 0);
}

// form 3's bound tables, behind its own scratch: offsets in bytes, 256-byte aligned
struct BoundWs {
  size_t vs, us, vbound, ubound, scalars, redo, total;
};
static inline BoundWs bound_layout(int64_t Bt, int64_t I, int64_t d, size_t base) {
  auto up = [](size_t x) { return (x + 255) / 256 * 256; };
  const size_t ks = (size_t)d / 16;
  BoundWs w{};
  size_t o = up(base);
  w.vs = o, o = up(o + (size_t)((I + 31) / 32) * (ks + 1) * 1024);  // (tiles of 32 items, KS + 1 blocks of 1 KiB)
  w.us = o, o = up(o + (size_t)Bt * (ks + 2) * 32);
  w.vbound = o, o = up(o + (size_t)I * 4);
  w.ubound = o, o = up(o + (size_t)Bt * 4);
  w.scalars = o, o = up(o + COLLECT_SCALARS * 4);  // the call's scalars (SC_*: idg_score_bf16.inc)
  w.redo = o, o = up(o + (size_t)Bt * 4);  // batch indices of the users handed to topk_redo_kernel
  w.total = o;
  return w;
}

struct CollectWs {
  size_t group_max, floor0, count, cand, tail;
};
static inline CollectWs collect_layout(int64_t Bt, int nc, int64_t I) {
  auto up = [](size_t x) { return (x + 255) / 256 * 256; };
  CollectWs w{};
  size_t o = 0;
  w.group_max = o, o = up(o + (size_t)Bt * COLLECT_GROUPS_MAX * 4);
  w.floor0 = o, o = up(o + (size_t)Bt * 4);
  w.count = o, o = up(o + (size_t)Bt * (size_t)nc * 4);
  w.cand = o, o = up(o + (size_t)Bt * (size_t)nc * (size_t)collect_cap_chunk(nc, I) * 8);
  w.tail = o;
  return w;
}
// A form-3 call's workspace: [collect scratch | the exact form's scratch of its fall-back] (the same bytes: the fall-back's
// launches run after the finish, when the candidate lists are no longer needed) followed by the bound tables.
static inline BoundWs collect_total_layout(int64_t Bt, int64_t I, int64_t d, int k, int nc) {
  int nc_x, form_x;
  int64_t ci_x;
  fused_geometry(Bt, I, &nc_x, &ci_x, &form_x, k, d, true);
  return bound_layout(Bt, I, d, std::max(collect_layout(Bt, nc, I).tail, exact_ws_bytes(Bt, nc_x, ci_x, form_x, k, false)));
}

int idg_score_topk_option(int which, int64_t value, int64_t* previous) {
  if (which == IDG_TOPK_OPT_RESET) {  // back to the defaults (NOT the environment: a test's override must not outlive it)
    topk_options() = topk_defaults();
    return IDG_OK;
  }
  IDG_REQUIRE(which >= 0 && which < IDG_TOPK_OPT_COUNT, "idg_score_topk_option: unknown option %d", which);
  if (previous) *previous = topk_options().v[which];
  if (value != IDG_TOPK_OPT_KEEP) topk_options().v[which] = value;
  return IDG_OK;
}

size_t idg_score_topk_workspace_bytes(int64_t Bt, int64_t I, int64_t d, int k) {
  if (Bt <= 0 || I <= 0) return 0;
  int nc, form;
  int64_t ci;
  fused_geometry(Bt, I, &nc, &ci, &form, k, d);
  if (form == 3) return collect_total_layout(Bt, I, d, k, nc).total;
  return exact_ws_bytes(Bt, nc, ci, form, k);
}

int idg_score_topk_info(int64_t Bt, int64_t I, int64_t d, int k, const void* ws, int64_t info[8], void* stream) {
  IDG_REQUIRE(info && Bt > 0 && I > 0 && d > 0 && k >= 1, "idg_score_topk_info: bad argument");
  int nc, form;
  int64_t ci;
  fused_geometry(Bt, I, &nc, &ci, &form, k, d);
  info[0] = form, info[1] = nc, info[2] = floor_phase(form, nc, ci, k) ? 1 : 0;
  info[3] = info[4] = info[5] = info[6] = -1, info[7] = 0;
  if (form == 3 && ws) {
    const BoundWs w = collect_total_layout(Bt, I, d, k, nc);
    uint32_t sc[COLLECT_SCALARS];
    IDG_HIP(hipMemcpyAsync(sc, reinterpret_cast<const char*>(ws) + w.scalars, sizeof(sc), hipMemcpyDeviceToHost, (hipStream_t)stream));
    IDG_HIP(hipStreamSynchronize((hipStream_t)stream));
    const bool fell_back = sc[SC_THRESHOLD] != SC_NEVER && (sc[SC_ITEMS_IRREGULAR] != 0u || sc[SC_REDO] > sc[SC_THRESHOLD]);
    info[3] = fell_back ? 0 : sc[SC_REDO], info[4] = sc[SC_REDO], info[5] = fell_back ? 1 : 0, info[6] = sc[SC_ITEMS_IRREGULAR];
  }
  return IDG_OK;
}

int idg_score_topk_candidate_counts(int64_t Bt, int64_t I, int64_t d, int k, const void* ws, int32_t* out_counts, void* stream) {
  IDG_REQUIRE(ws && out_counts && Bt > 0 && I > 0 && d > 0 && k >= 1, "idg_score_topk_candidate_counts: bad argument");
  int nc, form;
  int64_t ci;
  fused_geometry(Bt, I, &nc, &ci, &form, k, d);
  IDG_REQUIRE(form == 3, "idg_score_topk_candidate_counts: a call of this geometry does not take the threshold + collect form");
  const CollectWs cw = collect_layout(Bt, nc, I);
  hipLaunchKernelGGL(candidate_counts_kernel, dim3((unsigned)((Bt + BLOCK - 1) / BLOCK)), dim3(BLOCK), 0, (hipStream_t)stream,
                     reinterpret_cast<const unsigned int*>(reinterpret_cast<const char*>(ws) + cw.count), Bt, nc, out_counts);
  IDG_HIP(hipGetLastError());
  return IDG_OK;
}

// The exact forms (0: every wave alternates between scoring and selecting, 1: producer / consumer waves) in geometry
// (nc, ci), k of any size.  gate != nullptr (form 3's fall-back, k <= 64): every launch returns at once unless *gate != 0,
// and there is no floor phase.
static int launch_exact(const float* user_panel, const float* item_panel, const int64_t* users, int64_t Bt, int64_t I, int64_t d,
                        const int64_t* excl_indptr, const int32_t* excl_items, int k, int apply_sigmoid, int64_t* out_idx,
                        float* out_val, void* ws, int form, int nc, int64_t ci, const uint32_t* gate, hipStream_t st) {
  unsigned long long* partial = reinterpret_cast<unsigned long long*>(ws);
  const dim3 grid((unsigned)nc, (unsigned)((Bt + FT_USERS - 1) / FT_USERS));
  const unsigned nbm = (unsigned)((Bt + (BLOCK / WAVE) - 1) / (BLOCK / WAVE));
  // k <= 64: one pass.  Larger k (the reference's torch.topk takes any k <= I, batch_test.py:68): one pass per 64
  // ranks; pass p only admits keys strictly below the last key pass p - 1 emitted (keys are unique per item, so
  // "below the 64p-th best" is exactly "not among the best 64p"), and its winners fill columns [64p, 64p + kk).
  unsigned long long* bound = k > 64 ? partial + (size_t)Bt * (size_t)nc * 64 : nullptr;
  float* chunk_max = nullptr;
  float* floor0 = nullptr;
  if (gate == nullptr && floor_phase(form, nc, ci, k)) {
    chunk_max = reinterpret_cast<float*>(partial + (size_t)Bt * (size_t)nc * 64);
    floor0 = chunk_max + (size_t)Bt * (size_t)nc;
    const bool d64 = d == 64 && (uintptr_t)user_panel % 16 == 0 && (uintptr_t)item_panel % 16 == 0;
    const int slabs_per_chunk = (int)(ci / FT_SLAB);
    const int floor_slabs = slabs_per_chunk / 2 < FLOOR_SLABS ? slabs_per_chunk / 2 : FLOOR_SLABS;
#define IDG_MAXONLY(SIG, D64)                                                                                              \
  hipLaunchKernelGGL((score_topk_spec_kernel<SIG, D64, true>), grid, dim3(SP_BLOCK), 0, st, user_panel, item_panel, users, \
                     Bt, I, d, ci, excl_indptr, excl_items, floor_slabs, partial, (const unsigned long long*)nullptr,     \
                     chunk_max, (const float*)nullptr, gate)
    if (apply_sigmoid && d64) IDG_MAXONLY(true, true);
    else if (apply_sigmoid) IDG_MAXONLY(true, false);
    else if (d64) IDG_MAXONLY(false, true);
    else IDG_MAXONLY(false, false);
#undef IDG_MAXONLY
    hipLaunchKernelGGL(chunk_floor_kernel, dim3(nbm), dim3(BLOCK), 0, st, chunk_max, Bt, nc, k, floor0, gate);
  }
  for (int done = 0; done < k; done += 64) {
    const int kk = k - done < 64 ? k - done : 64;
    const unsigned long long* bd_in = done > 0 ? bound : nullptr;
    unsigned long long* bd_out = done + kk < k ? bound : nullptr;
    if (form == 1) {  // producer / consumer waves (many user tiles); 0: every wave alternates between the two phases
      const bool d64 = d == 64 && (uintptr_t)user_panel % 16 == 0 && (uintptr_t)item_panel % 16 == 0;
#define IDG_SPEC(SIG, D64)                                                                                           \
  hipLaunchKernelGGL((score_topk_spec_kernel<SIG, D64>), grid, dim3(SP_BLOCK), 0, st, user_panel, item_panel, users, \
                     Bt, I, d, ci, excl_indptr, excl_items, kk, partial, bd_in, (float*)nullptr, (const float*)floor0, gate)
      if (apply_sigmoid && d64) IDG_SPEC(true, true);
      else if (apply_sigmoid) IDG_SPEC(true, false);
      else if (d64) IDG_SPEC(false, true);
      else IDG_SPEC(false, false);
#undef IDG_SPEC
    } else if (apply_sigmoid)
      hipLaunchKernelGGL(score_topk_fused_kernel<true>, grid, dim3(BLOCK), 0, st, user_panel, item_panel, users,
                         Bt, I, d, ci, excl_indptr, excl_items, kk, partial, bd_in, gate);
    else
      hipLaunchKernelGGL(score_topk_fused_kernel<false>, grid, dim3(BLOCK), 0, st, user_panel, item_panel, users,
                         Bt, I, d, ci, excl_indptr, excl_items, kk, partial, bd_in, gate);
    if (apply_sigmoid)
      hipLaunchKernelGGL(topk_merge_kernel<true>, dim3(nbm), dim3(BLOCK), 0, st, partial, Bt, nc, kk, out_idx, out_val,
                         (int64_t)k, (int64_t)done, bd_out, (unsigned long long*)nullptr, gate);
    else
      hipLaunchKernelGGL(topk_merge_kernel<false>, dim3(nbm), dim3(BLOCK), 0, st, partial, Bt, nc, kk, out_idx, out_val,
                         (int64_t)k, (int64_t)done, bd_out, (unsigned long long*)nullptr, gate);
  }
  IDG_HIP(hipGetLastError());
  return IDG_OK;
}

int idg_score_topk_f32(const float* user_panel, const float* item_panel, const int64_t* users, int64_t Bt,
                       int64_t I, int64_t d, const int64_t* excl_indptr, const int32_t* excl_items, int k,
                       int apply_sigmoid, int64_t* out_idx, float* out_val, void* ws, void* stream) {
  IDG_REQUIRE(user_panel && item_panel && users && out_idx && ws, "idg_score_topk_f32: NULL argument");
  IDG_REQUIRE(Bt > 0 && I > 0 && d > 0, "idg_score_topk_f32: bad sizes");
  IDG_REQUIRE(k >= 1 && k <= 1024, "idg_score_topk_f32: k=%d outside [1,1024]", k);
  IDG_REQUIRE(k <= I, "idg_score_topk_f32: k=%d exceeds the item count %lld", k, (long long)I);
  IDG_REQUIRE(I < ((int64_t)1 << 32), "idg_score_topk_f32: more than 2^32 items");
  IDG_REQUIRE((excl_indptr == nullptr) == (excl_items == nullptr), "idg_score_topk_f32: excl_indptr and excl_items go together");
  hipStream_t st = (hipStream_t)stream;
  int nc, form;
  int64_t ci;
  fused_geometry(Bt, I, &nc, &ci, &form, k, d);
  if (form != 3)
    return launch_exact(user_panel, item_panel, users, Bt, I, d, excl_indptr, excl_items, k, apply_sigmoid, out_idx, out_val, ws, form,
                        nc, ci, nullptr, st);
  // threshold + collect + exact finish (idg_score_collect.inc)
  IDG_REQUIRE((uintptr_t)ws % 16 == 0, "idg_score_topk_f32: workspace must be 16-byte aligned");
  const unsigned nbm = (unsigned)((Bt + (BLOCK / WAVE) - 1) / (BLOCK / WAVE));
  const CollectWs cw = collect_layout(Bt, nc, I);
  const BoundWs w = collect_total_layout(Bt, I, d, k, nc);
  char* wb = reinterpret_cast<char*>(ws);
  float* group_max = reinterpret_cast<float*>(wb + cw.group_max);
  float* floor0 = reinterpret_cast<float*>(wb + cw.floor0);
  unsigned int* count = reinterpret_cast<unsigned int*>(wb + cw.count);
  unsigned long long* cand = reinterpret_cast<unsigned long long*>(wb + cw.cand);
  __bf16* Vs = reinterpret_cast<__bf16*>(wb + w.vs);
  __bf16* Us = reinterpret_cast<__bf16*>(wb + w.us);
  float* vbound = reinterpret_cast<float*>(wb + w.vbound);
  float* ubound = reinterpret_cast<float*>(wb + w.ubound);
  uint32_t* scal = reinterpret_cast<uint32_t*>(wb + w.scalars);
  uint32_t* redo_list = reinterpret_cast<uint32_t*>(wb + w.redo);
  const int64_t I_pad = (I + 31) / 32 * 32;
  const int lg = d == 64 ? 3 : d == 128 ? 4 : 5;  // 8-feature groups per row: d / 8
  const int64_t gpr = d / 8;
  // how many users the finish may hand to the one-by-one exact pass; beyond that (or with an irregular item row) the call
  // is answered by the exact form enqueued below, and the redo stands down — decided on the device (fallback_taken)
  const int64_t permille = topk_options().v[IDG_TOPK_OPT_FALLBACK_PERMILLE];
  const uint32_t fallback_threshold = permille < 0 ? SC_NEVER : (uint32_t)std::min<int64_t>(std::max<int64_t>(Bt * permille / 1000, 4), 0x7FFFFFFF);
  // (the users' launch goes first and zeroes the call's scalars: the items' launch takes its maximum into scal[SC_VMAX])
  hipLaunchKernelGGL(bound_table_kernel, dim3((unsigned)((Bt * gpr + BLOCK - 1) / BLOCK)), dim3(BLOCK), 0, st, user_panel, users, Bt,
                     Bt, d, lg, 0, bound_c(d), Us, ubound, (uint32_t*)nullptr, scal, fallback_threshold);
  const unsigned n_tiles = (unsigned)(I_pad / 32);
  if (d == 64)
    hipLaunchKernelGGL(item_table_kernel<4>, dim3(n_tiles), dim3(64 * 4), 0, st, item_panel, I, Vs, vbound, scal);
  else if (d == 128)
    hipLaunchKernelGGL(item_table_kernel<8>, dim3(n_tiles), dim3(64 * 8), 0, st, item_panel, I, Vs, vbound, scal);
  else
    hipLaunchKernelGGL(item_table_kernel<16>, dim3(n_tiles), dim3(64 * 16), 0, st, item_panel, I, Vs, vbound, scal);
  const int n_slabs_all = (int)((I + FT_SLAB - 1) / FT_SLAB);
  // the floor pass samples about a tenth of the catalogue whatever its size: `groups` / 2 x gs slabs, evenly spaced, a group =
  // the same half of gs consecutive sampled slabs (yelp2018 size: gs = 1, 32 or 64 slabs of 298)
  const int groups = collect_groups(k);
  const int gs = std::max(1, n_slabs_all / 300);
  const int walk = (groups / 2) * gs;
  const int stride = std::max(1, n_slabs_all / walk);  // (the last sampled slab lies inside the catalogue)
  // the floor pass is cut by slab groups (a power of two of them per workgroup), the collect pass by fused_geometry's chunks
  int fc = 1;
  while (2 * fc <= groups / 2 && (int64_t)2 * fc <= nc) fc *= 2;
  const int walk_c = walk / fc;
  const int64_t ci_floor = (int64_t)walk_c * stride * FT_SLAB;
  const int cap_chunk = collect_cap_chunk(nc, I);
  const unsigned tiles = (unsigned)((Bt + FT_USERS - 1) / FT_USERS);
#define IDG_COLLECT_KS(SIG, GM, KS_, GRIDX, CI, WALK, STRIDE, GS)                                                                \
  hipLaunchKernelGGL((score_topk_collect_kernel<SIG, GM, KS_>), dim3((unsigned)(GRIDX), tiles), dim3(SP_BLOCK), 0, st, Us, Vs, users, Bt, \
                     I, (int64_t)(CI), WALK, STRIDE, GS, excl_indptr, excl_items, group_max, (const float*)floor0, count, cand,   \
                     cap_chunk)
#define IDG_COLLECT(SIG, GM, GRIDX, CI, WALK, STRIDE, GS)                                                                       \
  {                                                                                                                             \
    if (d == 64) IDG_COLLECT_KS(SIG, GM, 4, GRIDX, CI, WALK, STRIDE, GS);                                                       \
    else if (d == 128) IDG_COLLECT_KS(SIG, GM, 8, GRIDX, CI, WALK, STRIDE, GS);                                                 \
    else IDG_COLLECT_KS(SIG, GM, 16, GRIDX, CI, WALK, STRIDE, GS);                                                              \
  }
  if (apply_sigmoid) IDG_COLLECT(true, true, fc, ci_floor, walk_c, stride, gs)
  else IDG_COLLECT(false, true, fc, ci_floor, walk_c, stride, gs)
  hipLaunchKernelGGL(group_floor_kernel, dim3(nbm), dim3(BLOCK), 0, st, (const float*)group_max, Bt, groups, k, floor0);
  if (apply_sigmoid) IDG_COLLECT(true, false, nc, ci, 0, 1, 1)
  else IDG_COLLECT(false, false, nc, ci, 0, 1, 1)
#undef IDG_COLLECT
#undef IDG_COLLECT_KS
#define IDG_FINISH(SIG, CAP_)                                                                                                   \
  hipLaunchKernelGGL((topk_finish_kernel<SIG, CAP_>), dim3(nbm), dim3(BLOCK), 0, st, user_panel, item_panel, users, Bt, I, d,     \
                     excl_indptr, excl_items, k, count, cand, nc, cap_chunk, ubound, scal, out_idx, out_val, redo_list)
  if (collect_cap(I) == COLLECT_CAP) {
    if (apply_sigmoid) IDG_FINISH(true, COLLECT_CAP);
    else IDG_FINISH(false, COLLECT_CAP);
  } else {
    if (apply_sigmoid) IDG_FINISH(true, COLLECT_CAP_BIG);
    else IDG_FINISH(false, COLLECT_CAP_BIG);
  }
#undef IDG_FINISH
  // whoever the finish could not serve (none, on tables that do not tie massively): the grid is fixed, the count on the device
  const unsigned redo_grid = (unsigned)std::min<int64_t>((Bt + 3) / 4, 512);
  if (apply_sigmoid)
    hipLaunchKernelGGL(topk_redo_kernel<true>, dim3(redo_grid), dim3(REDO_WAVES * WAVE), 0, st, user_panel, item_panel, users, I, d,
                       excl_indptr, excl_items, k, (const uint32_t*)redo_list, (const uint32_t*)scal, out_idx, out_val);
  else
    hipLaunchKernelGGL(topk_redo_kernel<false>, dim3(redo_grid), dim3(REDO_WAVES * WAVE), 0, st, user_panel, item_panel, users, I, d,
                       excl_indptr, excl_items, k, (const uint32_t*)redo_list, (const uint32_t*)scal, out_idx, out_val);
  IDG_HIP(hipGetLastError());
  if (permille < 0) return IDG_OK;
  // the whole-call fall-back: the exact form in ITS geometry for this call, every launch gated on the call's scalars (its
  // floor phase left out: two launches per ordinary call instead of four, and a fall-back is the rare case)
  int nc_x, form_x;
  int64_t ci_x;
  fused_geometry(Bt, I, &nc_x, &ci_x, &form_x, k, d, true);
  return launch_exact(user_panel, item_panel, users, Bt, I, d, excl_indptr, excl_items, k, apply_sigmoid, out_idx, out_val, ws, form_x,
                      nc_x, ci_x, scal, st);
}

}  // extern "C"
