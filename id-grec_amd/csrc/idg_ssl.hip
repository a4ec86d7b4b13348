// In-batch InfoNCE between two views, forward and backward in one chain of kernels.
//
// Reference: utility/utility_function/losses.py:24-35 (get_InfoNCE_loss) as called by
// models/SimGCL.py:79-84, XSimGCL.py:80-86 and SGL.py:96-101:
//     idx = unique(batch ids);  a = normalize(view1[idx]);  b = normalize(view2[idx])
//     pos_i = exp(<a_i, b_i>/t);  ttl_i = sum_k exp(<a_i, b_k>/t);  loss = mean_i -log(pos_i/ttl_i + 1e-5)
// once for the batch's users and once for its positive items.  On stock PyTorch that is ~100
// small launches per step (unique's sorts, gathers, normalisations, two GEMMs, exp/sum/log and
// the autograd mirror of all of it) — more time than the step's twelve SpMM launches.  Here:
//   rows    bitmap of the batch's rows -> ascending compact id list (== torch.unique's order)
//   norm    a, b rows normalised into compact panels (one wave per row)
//   logits  P = exp(a.b^T / t), 64x64 LDS tiles, both row sets in one launch (blockIdx.z)
//   stats   per row: ttl, loss term, the backward weight w_i = -r_i / (t m (r_i + eps))
//   grads   dL/da = w_i (b_i - sum_k Q_ik b_k),  dL/db_k = w_k a_k - sum_i w_i Q_ik a_i,  Q = P/ttl
//   final   back through normalize(): dx = (g - <g, y> y) / ||x||, stored to the views' gradient rows
// Everything is deterministic (no float atomics); sizes that depend on the data (the number of
// unique ids) stay on the device: launches are shaped by the batch size and read the counts.
#include <hip/hip_runtime.h>

#include <cstdint>
#include <cstdlib>

#include "idg_common.h"

extern "C" int idg_bpr_touch_rows(const int64_t* users, const int64_t* pos, const int64_t* neg, int64_t B,
                                  int64_t num_users, uint32_t* bitmap, void* stream);

namespace {

constexpr int BLOCK = 256;
constexpr int WAVE = 64;
constexpr int TS = 64;  // tile edge of the logits / gradient GEMMs
constexpr int KC = 16;  // reduction chunk staged in LDS (logits)
constexpr int GS = 8;   // split-K slices of the gradient products

struct SslWs {
  uint32_t* bitmap;  // [(n+31)/32]
  uint32_t* dup;     // [(n+31)/32] raw-id lists (dedup = 0): rows that occur more than once in their list
  int32_t* meta;     // [64] raw-id lists: [0] arrival ticket of ssl_mark_dups_kernel, [1 + 2 v] / [2 + 2 v] number of list
                     //      positions of view v whose id occurs more than once: in set 0 / in both sets (cleared with the bitmaps)
  int32_t* dlist;    // [2 views][2B] those positions, ascending
  int32_t* idx;      // [2B] ascending panel rows: the user set, then the item set
  int32_t* idx2;     // [2B] view 2's rows when they differ from view 1's (idg_infonce_cross_f32), else == idx
  int32_t* counts;   // [2] sizes of the two sets (+2 pad)
  float* An;         // [2 views][2B][d] normalised rows
  float* den;        // [2 views][2B]   max(||x||, 1e-12)
  float* P;          // [2 sets][B][B]
  float* coef;       // [2 sets][B] w_i / ttl_i (scales the second gradient product's operand)
  float* contrib;    // [2 views][2B][d] every list position's own gradient row (raw-id lists: summed per id afterwards)
  float* invttl;     // [2B]
  float* w;          // [2B]
  float* lossrow;    // [2B]
  float* G;          // [2 sides][GS slices][2B][d]  split-K partial sums of the two gradient products
  size_t bytes;
};

inline size_t up256(size_t x) { return (x + 255) / 256 * 256; }

SslWs ssl_layout(void* base, int64_t n, int64_t B, int64_t d) {
  SslWs w{};
  char* p = reinterpret_cast<char*>(base);
  size_t off = 0;
  auto take = [&](size_t bytes) {
    char* q = p ? p + off : nullptr;
    off += up256(bytes);
    return q;
  };
  w.bitmap = reinterpret_cast<uint32_t*>(take((size_t)((n + 31) / 32) * 4));
  w.dup = reinterpret_cast<uint32_t*>(take((size_t)((n + 31) / 32) * 4));
  w.meta = reinterpret_cast<int32_t*>(take(256));
  w.dlist = reinterpret_cast<int32_t*>(take((size_t)2 * 2 * B * 4));
  w.idx = reinterpret_cast<int32_t*>(take((size_t)2 * B * 4));
  w.idx2 = reinterpret_cast<int32_t*>(take((size_t)2 * B * 4));
  w.counts = reinterpret_cast<int32_t*>(take(16));
  w.An = reinterpret_cast<float*>(take((size_t)2 * 2 * B * d * 4));
  w.den = reinterpret_cast<float*>(take((size_t)2 * 2 * B * 4));
  w.P = reinterpret_cast<float*>(take((size_t)2 * B * B * 4));
  w.coef = reinterpret_cast<float*>(take((size_t)2 * B * 4));
  w.contrib = reinterpret_cast<float*>(take((size_t)2 * 2 * B * d * 4));
  w.invttl = reinterpret_cast<float*>(take((size_t)2 * B * 4));
  w.w = reinterpret_cast<float*>(take((size_t)2 * B * 4));
  w.lossrow = reinterpret_cast<float*>(take((size_t)2 * B * 4));
  w.G = reinterpret_cast<float*>(take((size_t)2 * GS * 2 * B * d * 4));
  w.bytes = off;
  return w;
}

__device__ __forceinline__ float wave_sum(float v) {
#pragma unroll
  for (int m = 32; m >= 1; m >>= 1) v += __shfl_xor(v, m, WAVE);
  return v;
}

// ---- bitmap -> ascending id list; ids < num_users form set 0, the rest set 1.  One 1024-thread block.
__global__ __launch_bounds__(1024) void ssl_compact_kernel(const uint32_t* __restrict__ bitmap, int64_t n,
                                                            int64_t num_users, int32_t* __restrict__ idx,
                                                            int32_t* __restrict__ counts, int64_t cap) {
  __shared__ int s_cnt[1024];
  __shared__ int s_users[1024];
  const int tid = threadIdx.x;
  const int64_t words = (n + 31) / 32;
  const int64_t per = (words + 1023) / 1024;
  const int64_t w0 = (int64_t)tid * per, w1 = w0 + per < words ? w0 + per : words;
  int c = 0, cu = 0;
  for (int64_t w = w0; w < w1; ++w) {
    const uint32_t m = bitmap[w];
    c += __popc(m);
    // bits of this word that are user rows
    const int64_t lo = w * 32;
    if (lo + 32 <= num_users) cu += __popc(m);
    else if (lo < num_users) cu += __popc(m & ((1u << (num_users - lo)) - 1u));
  }
  s_cnt[tid] = c;
  s_users[tid] = cu;
  __syncthreads();
  // inclusive scan (Hillis-Steele) over 1024 partial counts
  for (int off = 1; off < 1024; off <<= 1) {
    const int a = tid >= off ? s_cnt[tid - off] : 0;
    const int b = tid >= off ? s_users[tid - off] : 0;
    __syncthreads();
    s_cnt[tid] += a;
    s_users[tid] += b;
    __syncthreads();
  }
  int64_t at = s_cnt[tid] - c;
  for (int64_t w = w0; w < w1; ++w) {
    uint32_t m = bitmap[w];
    while (m) {
      const int b = __ffs(m) - 1;
      m &= m - 1;
      if (at < cap) idx[at] = (int32_t)(w * 32 + b);
      ++at;
    }
  }
  if (tid == 1023) {
    counts[0] = s_users[1023];
    counts[1] = s_cnt[1023] - s_users[1023];
  }
}

// ---- no de-duplication (SGL.py:85-86 indexes the views with the raw batch ids): the two lists as they are
__global__ __launch_bounds__(BLOCK) void ssl_copy_ids_kernel(const int64_t* __restrict__ users,
                                                             const int64_t* __restrict__ items, int64_t B,
                                                             int64_t num_users, int32_t* __restrict__ idx,
                                                             int32_t* __restrict__ counts) {
  const int64_t i = (int64_t)blockIdx.x * BLOCK + threadIdx.x;
  if (i < B) {
    idx[i] = (int32_t)users[i];
    idx[B + i] = (int32_t)(num_users + items[i]);
  }
  if (i == 0) counts[0] = counts[1] = (int32_t)B;
}

// ---- cross form (models/EGCF.py:103: get_InfoNCE_loss(user_embedding, pos_embedding)): ONE set of B rows whose view-1
// rows are the batch users and whose view-2 rows are the batch's positive items, raw ids in batch order
__global__ __launch_bounds__(BLOCK) void ssl_cross_ids_kernel(const int64_t* __restrict__ users,
                                                              const int64_t* __restrict__ items, int64_t B,
                                                              int64_t num_users, int32_t* __restrict__ idx,
                                                              int32_t* __restrict__ idx2, int32_t* __restrict__ counts) {
  const int64_t i = (int64_t)blockIdx.x * BLOCK + threadIdx.x;
  if (i < B) {
    idx[i] = (int32_t)users[i];
    idx2[i] = (int32_t)(num_users + items[i]);
  }
  if (i == 0) counts[0] = (int32_t)B, counts[1] = 0;
}

// ---- raw-id lists: which rows occur more than once (seen: first visit, dup: any later one).  Most ids of a batch occur
// once; the gradient kernel then skips its scan of the whole list for them (a batch of 2048 raw ids: 32 rounds of load /
// compare / ballot per row, 124 us per call at yelp2018 size — two thirds of it for rows that have no second occurrence).
__global__ __launch_bounds__(BLOCK) void ssl_mark_dups_kernel(const int32_t* __restrict__ idx, const int32_t* __restrict__ idx2,
                                                              int64_t count, const int32_t* __restrict__ counts,
                                                              uint32_t* __restrict__ seen, uint32_t* __restrict__ dup,
                                                              int32_t* __restrict__ meta, int32_t* __restrict__ dlist,
                                                              int64_t B) {
  const int64_t i = (int64_t)blockIdx.x * BLOCK + threadIdx.x;
  const int views = idx2 != idx ? 2 : 1;
  if (i < count) {
    for (int v = 0; v < views; ++v) {
      const int32_t id = (v == 0 ? idx : idx2)[i];
      const uint32_t bit = 1u << (id & 31);
      if (atomicOr(seen + (id >> 5), bit) & bit) atomicOr(dup + (id >> 5), bit);
    }
  }
  // The block that arrives last lists, per view, the POSITIONS whose id occurs more than once, ascending: the gradient
  // kernel's ordered per-id sums then walk these few hundred positions instead of the whole list (44 -> 15 us per call
  // at B = 2048 with popular items repeating, round 4).  (Device-scope atomics above; drained + ticket; the bits are
  // read back past the L1.)
  __shared__ int s_last;
  __shared__ int s_wave[BLOCK / WAVE];
  // (the idiom of the split-row combine in idg_graph.hip: agent-scope atomics above, drained, then a relaxed agent-scope
  //  ticket; the reader loads past its L1.  A __threadfence() here writes the XCD's whole L2 back: measured +7 us per call)
  asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
  __syncthreads();
  if (threadIdx.x == 0) s_last = __hip_atomic_fetch_add(meta, 1, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT) == (int)gridDim.x - 1;
  __syncthreads();
  if (!s_last) return;
  const int lane = threadIdx.x % WAVE, wave = threadIdx.x / WAVE;
  const int cu = counts[0], total = cu + counts[1];
  for (int v = 0; v < views; ++v) {
    const int32_t* ix = v == 0 ? idx : idx2;
    int32_t* out = dlist + (int64_t)v * 2 * B;
    int at = 0, at_set0 = 0;
    for (int p0 = 0; p0 < total; p0 += BLOCK) {
      const int pos = p0 + (int)threadIdx.x;
      bool f = false;
      if (pos < total) {
        const int32_t id = ix[pos];
        f = (__hip_atomic_load(dup + (id >> 5), __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT) >> (id & 31)) & 1u;
      }
      const unsigned long long m = __ballot(f);
      if (lane == 0) s_wave[wave] = __popcll(m);
      __syncthreads();
      int before = 0, all = 0;
#pragma unroll
      for (int q = 0; q < BLOCK / WAVE; ++q) {
        before += q < wave ? s_wave[q] : 0;
        all += s_wave[q];
      }
      if (f) out[at + before + __popcll(m & ((1ull << lane) - 1ull))] = pos;
      // positions of set 0 among this chunk's flagged ones (the chunk that straddles cu: count them exactly)
      const unsigned long long m0 = __ballot(f && pos < cu);
      __syncthreads();
      if (lane == 0) s_wave[wave] = __popcll(m0);
      __syncthreads();
#pragma unroll
      for (int q = 0; q < BLOCK / WAVE; ++q) at_set0 += s_wave[q];
      at += all;
      __syncthreads();
    }
    if (threadIdx.x == 0) meta[1 + 2 * v] = at_set0, meta[2 + 2 * v] = at;
  }
}

// ---- normalise: one wave per (compact row, view)
__global__ __launch_bounds__(BLOCK) void ssl_normalize_kernel(const float* __restrict__ view1,
                                                              const float* __restrict__ view2, int64_t d,
                                                              const int32_t* __restrict__ idx,
                                                              const int32_t* __restrict__ idx2,
                                                              const int32_t* __restrict__ counts, int64_t B,
                                                              float* __restrict__ An, float* __restrict__ den) {
  const int lane = threadIdx.x % WAVE;
  const int64_t r = (int64_t)blockIdx.x * (BLOCK / WAVE) + threadIdx.x / WAVE;
  const int v = blockIdx.y;
  if (r >= counts[0] + counts[1]) return;
  const float* x = (v == 0 ? view1 : view2) + (int64_t)(v == 0 ? idx : idx2)[r] * d;
  float ss = 0.f;
  for (int64_t f = lane; f < d; f += WAVE) ss += x[f] * x[f];
  ss = wave_sum(ss);
  const float nrm = fmaxf(sqrtf(ss), 1e-12f);  // torch.nn.functional.normalize: x / max(||x||, eps)
  float* y = An + ((int64_t)v * 2 * B + r) * d;
  for (int64_t f = lane; f < d; f += WAVE) y[f] = x[f] / nrm;
  if (lane == 0) den[(int64_t)v * 2 * B + r] = nrm;
}

// ---- P[set][i][k] = exp(<a_i, b_k> / t): 64 x 64 tile per block, 4 x 4 per thread
__global__ __launch_bounds__(BLOCK) void ssl_logits_kernel(const float* __restrict__ An, int64_t d, int64_t B,
                                                           const int32_t* __restrict__ counts, float inv_t,
                                                           float* __restrict__ P) {
  __shared__ float sa[KC][TS + 1];
  __shared__ float sb[KC][TS + 1];
  const int set = blockIdx.z;
  const int m = counts[set];
  const int i0 = blockIdx.y * TS, k0 = blockIdx.x * TS;
  if (i0 >= m || k0 >= m) return;
  const int64_t base = set == 0 ? 0 : counts[0];
  const float* A = An + base * d;                       // view 1 rows of this set
  const float* Bm = An + ((int64_t)2 * B + base) * d;   // view 2 rows of this set
  const int tid = threadIdx.x, tx = tid % 16, ty = tid / 16;
  float acc[4][4] = {};
  for (int64_t f0 = 0; f0 < d; f0 += KC) {
    for (int e = tid; e < TS * KC; e += BLOCK) {
      const int row = e / KC, f = e % KC;
      sa[f][row] = (i0 + row < m && f0 + f < d) ? A[(int64_t)(i0 + row) * d + f0 + f] : 0.f;
      sb[f][row] = (k0 + row < m && f0 + f < d) ? Bm[(int64_t)(k0 + row) * d + f0 + f] : 0.f;
    }
    __syncthreads();
#pragma unroll
    for (int f = 0; f < KC; ++f) {
      float a[4], b[4];
#pragma unroll
      for (int q = 0; q < 4; ++q) a[q] = sa[f][ty * 4 + q], b[q] = sb[f][tx * 4 + q];
#pragma unroll
      for (int p = 0; p < 4; ++p)
#pragma unroll
        for (int q = 0; q < 4; ++q) acc[p][q] = __builtin_fmaf(a[p], b[q], acc[p][q]);
    }
    __syncthreads();
  }
  float* Ps = P + (int64_t)set * B * B;
#pragma unroll
  for (int p = 0; p < 4; ++p)
#pragma unroll
    for (int q = 0; q < 4; ++q) {
      const int i = i0 + ty * 4 + p, k = k0 + tx * 4 + q;
      if (i < m && k < m) Ps[(int64_t)i * B + k] = expf(acc[p][q] * inv_t);
    }
}

// ---- per row: ttl, loss term, backward weight.  One wave per row.
__global__ __launch_bounds__(BLOCK) void ssl_rowstat_kernel(const float* __restrict__ P, int64_t B,
                                                            const int32_t* __restrict__ counts, float inv_t, float eps,
                                                            float* __restrict__ invttl, float* __restrict__ w,
                                                            float* __restrict__ lossrow, float* __restrict__ coef) {
  const int set = blockIdx.y;
  const int m = counts[set];
  const int lane = threadIdx.x % WAVE;
  const int64_t i = (int64_t)blockIdx.x * (BLOCK / WAVE) + threadIdx.x / WAVE;
  if (i >= m) return;
  const float* row = P + (int64_t)set * B * B + i * B;
  float s = 0.f;
  for (int k0 = lane; k0 < m; k0 += 8 * WAVE) {  // eight loads in flight, added in the order of the plain loop
    float v[8];
#pragma unroll
    for (int q = 0; q < 8; ++q) v[q] = k0 + q * WAVE < m ? row[k0 + q * WAVE] : 0.f;
#pragma unroll
    for (int q = 0; q < 8; ++q) s += v[q];
  }
  s = wave_sum(s);
  if (lane == 0) {
    const float r = row[i] / s;
    const int64_t o = (set == 0 ? 0 : counts[0]) + i;
    invttl[o] = 1.0f / s;
    lossrow[o] = -logf(r + eps);
    const float wv = -r * inv_t / ((float)m * (r + eps));
    w[o] = wv;
    if (coef) coef[(int64_t)set * B + i] = wv * (1.0f / s);  // == w[o] * invttl[o], as the SIMT gradient kernel forms it
  }
}

// ---- loss[set] = mean of the row terms (one block per set, fixed tree)
__global__ __launch_bounds__(BLOCK) void ssl_loss_kernel(const float* __restrict__ lossrow,
                                                         const int32_t* __restrict__ counts, float* __restrict__ loss) {
  __shared__ float s[BLOCK];
  const int set = blockIdx.x;
  const int m = counts[set];
  const float* x = lossrow + (set == 0 ? 0 : counts[0]);
  float a = 0.f;
  for (int i = threadIdx.x; i < m; i += BLOCK) a += x[i];
  s[threadIdx.x] = a;
  __syncthreads();
  for (int off = BLOCK / 2; off > 0; off >>= 1) {
    if ((int)threadIdx.x < off) s[threadIdx.x] += s[threadIdx.x + off];
    __syncthreads();
  }
  if (threadIdx.x == 0) loss[set] = m > 0 ? s[0] / (float)m : 0.f;  // (the cross form has one set)
}

// ---- gradients with respect to the normalised rows: the two products  sum_k P_ik b_k  (side 0, rows i) and
//   sum_i (w_i invttl_i) P_ik a_i  (side 1, rows k), [m x m] . [m x d] each.  With m ~ 10^3 and d = 64 these are
//   small GEMMs with a long reduction: the reduction index is cut into GS slices (split-K) so that ~10^3
//   workgroups run at once; every block owns a 64-row x 64-feature tile of one slice and writes its raw sums
//   to Gp[side][slice][row][d].  ssl_final_kernel adds the slices in slice order (deterministic).
//   grid: (feature tiles * GS, row tiles, set * 2 + side)
__global__ __launch_bounds__(BLOCK) void ssl_grad_kernel(const float* __restrict__ An, const float* __restrict__ P,
                                                         int64_t d, int64_t B, const int32_t* __restrict__ counts,
                                                         const float* __restrict__ invttl, const float* __restrict__ w,
                                                         float* __restrict__ Gp) {
  __shared__ float sp[TS][TS + 1];                            // P chunk: [reduction index][output row]
  __shared__ __attribute__((aligned(16))) float sx[TS][TS];   // the other view's rows: [reduction index][feature]
  const int set = blockIdx.z >> 1, side = blockIdx.z & 1;
  const int m = counts[set];
  const int slice = blockIdx.x % GS;
  const int64_t f0 = (int64_t)(blockIdx.x / GS) * TS;
  const int r0 = blockIdx.y * TS;
  if (r0 >= m) return;
  const int64_t base = set == 0 ? 0 : counts[0];
  const float* A = An + base * d;
  const float* Bm = An + ((int64_t)2 * B + base) * d;
  const float* X = side == 0 ? Bm : A;  // rows being mixed
  const float* Ps = P + (int64_t)set * B * B;
  const float* it = invttl + base;
  const float* ww = w + base;
  const int per = ((m + GS - 1) / GS + TS - 1) / TS * TS;  // reduction indices per slice
  const int c_lo = slice * per, c_hi = c_lo + per < m ? c_lo + per : m;
  const int tid = threadIdx.x, tx = tid % 16, ty = tid / 16;
  float acc[4][4] = {};
  for (int c0 = c_lo; c0 < c_hi; c0 += TS) {
    for (int e = tid; e < TS * TS; e += BLOCK) {
      // P element (output row r0+row, reduction index c0+c): side 0 reads P[row][c] (contiguous in c),
      // side 1 reads P[c][row] * w_c invttl_c (contiguous in row)
      const int row = side == 0 ? e / TS : e % TS, c = side == 0 ? e % TS : e / TS;
      float v = 0.f;
      if (r0 + row < m && c0 + c < c_hi)
        v = side == 0 ? Ps[(int64_t)(r0 + row) * B + c0 + c] : Ps[(int64_t)(c0 + c) * B + r0 + row] * (ww[c0 + c] * it[c0 + c]);
      sp[c][row] = v;
    }
    for (int e = tid; e < TS * TS; e += BLOCK) {
      const int c = e / TS, f = e % TS;
      sx[c][f] = (c0 + c < c_hi && f0 + f < d) ? X[(int64_t)(c0 + c) * d + f0 + f] : 0.f;
    }
    __syncthreads();
#pragma unroll 8
    for (int c = 0; c < TS; ++c) {
      float p[4];
#pragma unroll
      for (int q = 0; q < 4; ++q) p[q] = sp[c][ty * 4 + q];
      const float4 x = *reinterpret_cast<const float4*>(&sx[c][tx * 4]);
#pragma unroll
      for (int a = 0; a < 4; ++a) {
        acc[a][0] = __builtin_fmaf(p[a], x.x, acc[a][0]);
        acc[a][1] = __builtin_fmaf(p[a], x.y, acc[a][1]);
        acc[a][2] = __builtin_fmaf(p[a], x.z, acc[a][2]);
        acc[a][3] = __builtin_fmaf(p[a], x.w, acc[a][3]);
      }
    }
    __syncthreads();
  }
  float* out = Gp + (((int64_t)side * GS + slice) * 2 * B + base) * d;
#pragma unroll
  for (int a = 0; a < 4; ++a) {
    const int r = r0 + ty * 4 + a;
#pragma unroll
    for (int q = 0; q < 4; ++q) {
      const int64_t f = f0 + tx * 4 + q;
      if (r < m && f < d) out[(int64_t)r * d + f] = acc[a][q];
    }
  }
}

// ---- the two GEMM-shaped stages on the fp32 matrix cores (round 4; d % 64 == 0 and B % 4 == 0, every shipped
// configuration; otherwise the SIMT kernels above) -------------------------------------------------------------------
// v_mfma_f32_32x32x2_f32: exact fp32 products and accumulation.  Lane (i, h) feeds row i's K-values [kc + 32h, kc + 32h + 32)
// as the A operand and column i's as the B operand — one contiguous 128-byte run per lane and 64-deep chunk, nothing
// staged through LDS; C/D map: column = lane & 31, row = (r & 3) + 8 (r >> 2) + 4 (lane >> 5).
using f32x16 = __attribute__((ext_vector_type(16))) float;

__device__ __forceinline__ void load32(const float* __restrict__ p, float (&v)[32]) {
#pragma unroll
  for (int q = 0; q < 8; ++q) {
    const float4 x = *reinterpret_cast<const float4*>(p + 4 * q);
    v[4 * q + 0] = x.x, v[4 * q + 1] = x.y, v[4 * q + 2] = x.z, v[4 * q + 3] = x.w;
  }
}

// P[set][i][k] = exp(<a_i, b_k> / t): a 128 x 128 (or 64 x 128) tile per workgroup, 2 x 2 (1 x 2) MFMA tiles per wave.  The 128 rows of
// either view are read once per 64-deep chunk, coalesced, into padded LDS tiles and the MFMA operands come out of those
// (round 4; with one 32 x 32 tile per wave fed by row-per-lane global loads — 64 cache lines per load instruction, every
// operand row fetched by two waves of 32 workgroups — the kernel took 28 us for 7 us of MFMA work).  Same K order per
// element as before: same values.
constexpr int LT = 128;        // tile columns (rows of view 2); tile rows (of view 1) = 64 RT
constexpr int LLD = TS + 4;    // LDS row stride (floats)

// RT = 2: 128 x 128 per workgroup.  RT = 1: 64 x 128 (32 x 64 per wave) — twice the workgroups, for calls whose grid would
// otherwise leave half the chip idle (the cross form's single set at B = 2048: 256 tiles of 128 x 128).
template <int RT>
__global__ __launch_bounds__(BLOCK, 2) void ssl_logits_mfma_kernel(const float* __restrict__ An, int64_t d, int64_t B,
                                                                   const int32_t* __restrict__ counts, float inv_t,
                                                                   float* __restrict__ P) {
  constexpr int LTR = 64 * RT;
  __shared__ __attribute__((aligned(16))) float s_a[LTR * LLD];
  __shared__ __attribute__((aligned(16))) float s_b[LT * LLD];
  const int set = blockIdx.z;
  const int m = counts[set];
  const int i0 = blockIdx.y * LTR, k0 = blockIdx.x * LT;
  if (i0 >= m || k0 >= m) return;  // block-uniform
  const int64_t base = set == 0 ? 0 : counts[0];
  const float* A = An + base * d;
  const float* Bm = An + ((int64_t)2 * B + base) * d;
  const int tid = threadIdx.x, lane = tid % WAVE, wave = tid / WAVE, i = lane & 31, h = lane >> 5;
  const int wr = 32 * RT * (wave >> 1), wc = 64 * (wave & 1);  // this wave's (32 RT) x 64 part of the tile
  f32x16 acc[RT][2];
#pragma unroll
  for (int rt = 0; rt < RT; ++rt)
#pragma unroll
    for (int r = 0; r < 16; ++r) acc[rt][0][r] = 0.f, acc[rt][1][r] = 0.f;
  for (int64_t kc = 0; kc < d; kc += 64) {
#pragma unroll
    for (int j = 0; j < 8; ++j) {
      const int e = tid + BLOCK * j, rr = e >> 4, c4 = (e & 15) * 4;
      // (rows past the set: a valid row, never stored)
      if (rr < LTR) {
        const int ra = i0 + rr < m ? i0 + rr : m - 1;
        *reinterpret_cast<float4*>(s_a + rr * LLD + c4) = *reinterpret_cast<const float4*>(A + (int64_t)ra * d + kc + c4);
      }
      const int rb = k0 + rr < m ? k0 + rr : m - 1;
      *reinterpret_cast<float4*>(s_b + rr * LLD + c4) = *reinterpret_cast<const float4*>(Bm + (int64_t)rb * d + kc + c4);
    }
    __syncthreads();
    float a[RT][32];
#pragma unroll
    for (int rt = 0; rt < RT; ++rt) {
      const float* pa = s_a + (wr + 32 * rt + i) * LLD + 32 * h;
#pragma unroll
      for (int q = 0; q < 8; ++q) {
        const float4 x = *reinterpret_cast<const float4*>(pa + 4 * q);
        a[rt][4 * q + 0] = x.x, a[rt][4 * q + 1] = x.y, a[rt][4 * q + 2] = x.z, a[rt][4 * q + 3] = x.w;
      }
    }
#pragma unroll
    for (int ct = 0; ct < 2; ++ct) {
      const float* pb = s_b + (wc + 32 * ct + i) * LLD + 32 * h;
#pragma unroll
      for (int q = 0; q < 8; ++q) {
        const float4 y = *reinterpret_cast<const float4*>(pb + 4 * q);
        const float ys[4] = {y.x, y.y, y.z, y.w};
#pragma unroll
        for (int c = 0; c < 4; ++c)
#pragma unroll
          for (int rt = 0; rt < RT; ++rt)
            acc[rt][ct] = __builtin_amdgcn_mfma_f32_32x32x2f32(a[rt][4 * q + c], ys[c], acc[rt][ct], 0, 0, 0);
      }
    }
    __syncthreads();
  }
  float* Ps = P + (int64_t)set * B * B;
#pragma unroll
  for (int rt = 0; rt < RT; ++rt)
#pragma unroll
    for (int ct = 0; ct < 2; ++ct)
#pragma unroll
      for (int r = 0; r < 16; ++r) {
        const int row = i0 + wr + 32 * rt + (r & 3) + 8 * (r >> 2) + 4 * h, col = k0 + wc + 32 * ct + i;
        if (row < m && col < m) Ps[(int64_t)row * B + col] = expf(acc[rt][ct][r] * inv_t);
      }
}

// The two gradient products on the matrix cores: side 0  G[r][f] = sum_c P[r][c] b_c[f],  side 1  G[k][f] = sum_i P[i][k]
// (w_i / ttl_i) a_i[f].  64 output rows x 64 features per workgroup, the reduction index cut into GS slices as in the SIMT
// form; raw slice sums go to Gp[side][slice][row][f] (ssl_final_kernel adds them in slice order).
// Operands through LDS (round 4): per 64-deep chunk the workgroup reads the 64 x 64 block of P it needs — rows r, columns c
// for side 0; rows i, columns k for side 1, scaled by w_i / ttl_i on the way in — and the 64 rows of the other view, both
// as whole 256-byte runs (16 lanes x 16 bytes), into padded tiles; the MFMA operands come out of the tiles, side 1's P
// operand by columns.  Before, every lane read its own row of P, of a transposed copy PT, and of transposed copies of the
// normalised rows (64 cache lines per load instruction; the copies written by the logits / normalise kernels):
// 47 -> 28 us per call at B = 2048, and the logits kernel no longer writes PT (30 -> 20 us).  Same K order: same sums.
//   grid: (feature tiles * GS, row tiles, set * 2 + side)
constexpr int GLD = TS + 4;  // LDS row stride (floats): rows 16-byte aligned, 16 rows of b128 reads cover all banks

__global__ __launch_bounds__(BLOCK) void ssl_grad_mfma_kernel(const float* __restrict__ P, const float* __restrict__ An,
                                                              const float* __restrict__ coef, int64_t d, int64_t B,
                                                              const int32_t* __restrict__ counts, float* __restrict__ Gp) {
  __shared__ __attribute__((aligned(16))) float s_p[TS * GLD];  // side 0: [output row][reduction index]; side 1: [reduction index][output row]
  __shared__ __attribute__((aligned(16))) float s_x[TS * GLD];  // [reduction index][feature]
  const int set = blockIdx.z >> 1, side = blockIdx.z & 1;
  const int m = counts[set];
  const int slice = blockIdx.x % GS;
  const int64_t f0 = (int64_t)(blockIdx.x / GS) * TS;
  const int r0 = blockIdx.y * TS;
  if (r0 >= m) return;
  const int64_t base = set == 0 ? 0 : counts[0];
  const int tid = threadIdx.x, lane = tid % WAVE, wave = tid / WAVE, i = lane & 31, h = lane >> 5;
  const int tr = 32 * (wave >> 1), tf = 32 * (wave & 1);  // this wave's 32 x 32 tile inside the block's 64 x 64
  const float* Ps = P + (int64_t)set * B * B;
  // the OTHER view's rows: side 0 mixes b (view 2), side 1 mixes a (view 1)
  const float* X = An + ((int64_t)(side == 0 ? 2 * B : 0) + base) * d + f0;
  const float* cf = coef + (int64_t)set * B;
  const int per = ((m + GS - 1) / GS + TS - 1) / TS * TS;  // reduction indices per slice (a multiple of 64)
  const int c_lo = slice * per, c_hi = c_lo + per < m ? c_lo + per : m;
  f32x16 acc;
#pragma unroll
  for (int r = 0; r < 16; ++r) acc[r] = 0.f;
  for (int c0 = c_lo; c0 < c_hi; c0 += 64) {
#pragma unroll
    for (int j = 0; j < 4; ++j) {
      const int e = tid + BLOCK * j, rr = e >> 4, c4 = (e & 15) * 4;
      // block of P: LDS row rr = P row (side 0: r0 + rr; side 1: c0 + rr), columns (side 0: c0 ...; side 1: r0 ...).
      // Whatever lies beyond the set or the slice is staged as 0 (never read from memory: it was never written)
      const int prow = side == 0 ? r0 + rr : c0 + rr, pcol = (side == 0 ? c0 : r0) + c4;
      const int row_end = side == 0 ? m : c_hi, col_end = side == 0 ? c_hi : m;
      float4 v = make_float4(0.f, 0.f, 0.f, 0.f);
      if (prow < row_end) {
        const float* src = Ps + (int64_t)prow * B + pcol;
        if (pcol + 4 <= col_end) v = *reinterpret_cast<const float4*>(src);
        else {
          if (pcol + 0 < col_end) v.x = src[0];
          if (pcol + 1 < col_end) v.y = src[1];
          if (pcol + 2 < col_end) v.z = src[2];
        }
        if (side == 1) {
          const float w = cf[prow];
          v.x *= w, v.y *= w, v.z *= w, v.w *= w;
        }
      }
      *reinterpret_cast<float4*>(s_p + rr * GLD + c4) = v;
      float4 x = make_float4(0.f, 0.f, 0.f, 0.f);
      if (c0 + rr < c_hi) x = *reinterpret_cast<const float4*>(X + (int64_t)(c0 + rr) * d + c4);
      *reinterpret_cast<float4*>(s_x + rr * GLD + c4) = x;
    }
    __syncthreads();
    float a[32];
    if (side == 0) {
      const float* pa = s_p + (tr + i) * GLD + 32 * h;
#pragma unroll
      for (int q = 0; q < 8; ++q) {
        const float4 t = *reinterpret_cast<const float4*>(pa + 4 * q);
        a[4 * q + 0] = t.x, a[4 * q + 1] = t.y, a[4 * q + 2] = t.z, a[4 * q + 3] = t.w;
      }
    } else {
#pragma unroll
      for (int q = 0; q < 32; ++q) a[q] = s_p[(32 * h + q) * GLD + tr + i];
    }
#pragma unroll
    for (int q = 0; q < 32; ++q)
      acc = __builtin_amdgcn_mfma_f32_32x32x2f32(a[q], s_x[(32 * h + q) * GLD + tf + i], acc, 0, 0, 0);
    __syncthreads();
  }
  float* out = Gp + (((int64_t)side * GS + slice) * 2 * B + base) * d;
#pragma unroll
  for (int r = 0; r < 16; ++r) {
    const int row = r0 + tr + (r & 3) + 8 * (r >> 2) + 4 * h;
    if (row < m) out[(int64_t)row * d + f0 + tf + i] = acc[r];
  }
}

// ---- slices -> dL/d(normalised row) -> back through normalize() -> the views' gradient rows.
//   One wave per (compact row, view): view 1 rows take side 0 (Ga), view 2 rows side 1 (Gb).
//   With duplicate ids in a set (dedup == 0) the occurrences of one id all feed the same panel row: the wave of
//   the FIRST occurrence adds them up in list order (deterministic), the others leave.
// Stage 1 (every list position on its own wave): position j's gradient row for view v — slices added in slice order, back
// through normalize() — into contrib[v][j].  Fully parallel: a hub item that occurs 40 times in a raw list used to have
// all 40 rows computed one after the other by the wave of its first occurrence (72 us per call at yelp2018 size).
__global__ __launch_bounds__(BLOCK) void ssl_contrib_kernel(const float* __restrict__ An, const float* __restrict__ den,
                                                            const float* __restrict__ Gp, const float* __restrict__ invttl,
                                                            const float* __restrict__ w, int64_t d, int64_t B,
                                                            const int32_t* __restrict__ counts, float scale,
                                                            float* __restrict__ contrib) {
  const int lane = threadIdx.x % WAVE;
  const int64_t j = (int64_t)blockIdx.x * (BLOCK / WAVE) + threadIdx.x / WAVE;
  const int v = blockIdx.y;
  if (j >= counts[0] + counts[1]) return;
  const float* y = An + ((int64_t)v * 2 * B + j) * d;            // this view's normalised row
  const float* other = An + ((int64_t)(1 - v) * 2 * B + j) * d;  // b_i for view 1, a_k for view 2
  const float nrm = den[(int64_t)v * 2 * B + j];
  const float wr = w[j], itr = invttl[j];
  const bool clamped = nrm <= 1e-12f;  // normalize() divided by the constant eps: Jacobian 1/eps, no projection
  auto grad = [&](int64_t f) {
    float acc = 0.f;
    for (int sl = 0; sl < GS; ++sl) acc += Gp[(((int64_t)v * GS + sl) * 2 * B + j) * d + f];
    return v == 0 ? wr * (other[f] - itr * acc) : wr * other[f] - acc;
  };
  float dot = 0.f;  // the projection needs <g, y> over the whole row first
  for (int64_t f = lane; f < d; f += WAVE) dot += grad(f) * y[f];
  dot = wave_sum(dot);
  float* c = contrib + ((int64_t)v * 2 * B + j) * d;
  for (int64_t f = lane; f < d; f += WAVE) {
    const float g = grad(f);
    c[f] = scale * (clamped ? g / nrm : (g - dot * y[f]) / nrm);
  }
}

// Stage 2: the views' gradient rows.  One wave per (compact row, view): view 1 rows take contrib[0], view 2 rows contrib[1].
//   With duplicate ids in a set (dedup == 0) the occurrences of one id all feed the same panel row: the wave of
//   the FIRST occurrence adds them up in list order (deterministic), the others leave.
__global__ __launch_bounds__(BLOCK) void ssl_final_kernel(const float* __restrict__ contrib, int64_t d, int64_t B,
                                                          const int32_t* __restrict__ idx,
                                                          const int32_t* __restrict__ idx2,
                                                          const int32_t* __restrict__ counts, int dedup,
                                                          int accumulate, int both_views, float* g1, float* g2,
                                                          const uint32_t* __restrict__ dup, const int32_t* __restrict__ meta,
                                                          const int32_t* __restrict__ dlist) {
  const int lane = threadIdx.x % WAVE;
  const int64_t r = (int64_t)blockIdx.x * (BLOCK / WAVE) + threadIdx.x / WAVE;
  const int cu = counts[0], total = cu + counts[1];
  if (r >= total) return;
  // g1 == g2 (both_views): one wave adds both views' gradients into the shared panel, view 1 first.  The two views'
  // rows are the same panel row (idx2 == idx) or different ones (the cross form, which always accumulates)
  const int v_lo = both_views ? 0 : blockIdx.y, v_hi = both_views ? 2 : blockIdx.y + 1;
  bool fresh = !accumulate;  // the first value written to a row that is not accumulated into replaces its content
  for (int v = v_lo; v < v_hi; ++v) {
    float* out = v == 0 ? g1 : g2;
    if (!out) continue;
    const int32_t* ix = v == 0 ? idx : idx2;
    const int32_t id = ix[r];
    float* o = out + (int64_t)id * d;
    // a de-duplicated set has exactly one occurrence of `id`: r itself — and so has a raw list whose dup bit for the id
    // is clear (ssl_mark_dups_kernel)
    const bool single = dedup || (dup && !((dup[id >> 5] >> (id & 31)) & 1u));
    // occurrences of `id` in this view's list of the set, ascending: r itself, or — an id that repeats — found among the
    // listed positions of repeating ids (ssl_mark_dups_kernel; idx2 == idx: one list serves both views), this set's part
    const int lv = idx2 != idx ? v : 0;
    const int32_t* dl = dlist + (int64_t)lv * 2 * B;
    const int d_lo = single ? (int)r : (r < cu ? 0 : meta[1 + 2 * lv]);
    const int d_hi = single ? (int)r + 1 : (r < cu ? meta[1 + 2 * lv] : meta[2 + 2 * lv]);
    bool mine = true;  // the wave of the FIRST occurrence owns the panel row (for this view)
    if (!single) {
      mine = false;
      for (int c0 = d_lo; c0 < d_hi; c0 += WAVE) {
        const bool in = c0 + lane < d_hi;
        const int32_t pos = in ? dl[c0 + lane] : 0;
        const unsigned long long match = __ballot(in && ix[pos] == id);
        if (match) {
          mine = __shfl(pos, __builtin_ctzll(match), WAVE) == (int32_t)r;
          break;
        }
      }
    }
    if (mine) {
      // The row's sum is carried in a register and stored once: as a read-modify-write of the panel row per occurrence a
      // popular item's 40 occurrences were 40 dependent store -> load round trips (the kernel's critical path: 44 us per
      // call).  Up to four occurrences' rows are loaded together; the additions keep list order (same sums as before).
      for (int64_t f0 = 0; f0 < d; f0 += WAVE) {
        const int64_t f = f0 + lane;
        const bool live = f < d;
        bool have = !fresh;
        float acc = (have && live) ? o[f] : 0.f;
        for (int c0 = d_lo; c0 < d_hi; c0 += WAVE) {
          int32_t pos_l = (int32_t)r;
          unsigned long long match = 1ull;
          if (!single) {
            const bool in = c0 + lane < d_hi;
            pos_l = in ? dl[c0 + lane] : 0;
            match = __ballot(in && ix[pos_l] == id);
          }
          while (match) {
            int64_t js[4];
            int cnt = 0;
#pragma unroll
            for (int q = 0; q < 4; ++q) {
              js[q] = 0;
              if (match) {
                js[q] = single ? r : (int64_t)__shfl(pos_l, __builtin_ctzll(match), WAVE);
                match &= match - 1;
                cnt = q + 1;
              }
            }
            float cv[4];
#pragma unroll
            for (int q = 0; q < 4; ++q) cv[q] = (q < cnt && live) ? contrib[((int64_t)v * 2 * B + js[q]) * d + f] : 0.f;
#pragma unroll
            for (int q = 0; q < 4; ++q)
              if (q < cnt) {
                acc = have ? acc + cv[q] : cv[q];
                have = true;
              }
          }
        }
        if (live) o[f] = acc;
      }
      fresh = false;
    }
    if (!both_views || idx2 != idx) fresh = !accumulate;
  }
}

}  // namespace

extern "C" {

size_t idg_infonce_workspace_bytes(int64_t n, int64_t B, int64_t d) {
  if (n <= 0 || B <= 0 || d <= 0) return 0;
  return ssl_layout(nullptr, n, B, d).bytes;
}

// The id-list stage of a call — index-only work: the ascending unique rows of the batch (mode 0), its raw lists (1) or the
// cross form's two lists (2), and for raw lists the repeat flags and the listed positions of repeating ids.
static int infonce_ids(const int64_t* users, const int64_t* items, int64_t B, int64_t num_users, int64_t n, const SslWs& w,
                       int mode, hipStream_t st) {
  const bool cross = mode == 2, dedup = mode == 0;
  if (cross) {
    hipLaunchKernelGGL(ssl_cross_ids_kernel, dim3((unsigned)((B + BLOCK - 1) / BLOCK)), dim3(BLOCK), 0, st, users, items, B,
                       num_users, w.idx, w.idx2, w.counts);
  } else if (dedup) {
    IDG_HIP(hipMemsetAsync(w.bitmap, 0, (size_t)((n + 31) / 32) * 4, st));
    // rows of the batch's users and (positive) items; the third id list is not used here: pass the items twice
    int rc = idg_bpr_touch_rows(users, items, items, B, num_users, w.bitmap, (void*)st);
    if (rc != IDG_OK) return rc;
    hipLaunchKernelGGL(ssl_compact_kernel, dim3(1), dim3(1024), 0, st, w.bitmap, n, num_users, w.idx, w.counts, 2 * B);
  } else {
    hipLaunchKernelGGL(ssl_copy_ids_kernel, dim3((unsigned)((B + BLOCK - 1) / BLOCK)), dim3(BLOCK), 0, st, users, items, B,
                       num_users, w.idx, w.counts);
  }
  if (!dedup) {  // raw lists: flag the rows that occur more than once (bitmap, dup and the ticket / counts block are adjacent: one fill)
    const int64_t rows_max = cross ? B : 2 * B;
    IDG_HIP(hipMemsetAsync(w.bitmap, 0, (size_t)(reinterpret_cast<char*>(w.meta) - reinterpret_cast<char*>(w.bitmap)) + 256, st));
    hipLaunchKernelGGL(ssl_mark_dups_kernel, dim3((unsigned)((rows_max + BLOCK - 1) / BLOCK)), dim3(BLOCK), 0, st, w.idx,
                       cross ? w.idx2 : w.idx, rows_max, w.counts, w.bitmap, w.dup, w.meta, w.dlist, B);
  }
  IDG_HIP(hipGetLastError());
  return IDG_OK;
}

static int infonce_impl(const float* view1, const float* view2, int64_t n, int64_t d, const int64_t* users,
                        const int64_t* items, int64_t B, int64_t num_users, int dedup, int cross, int planned, float temperature,
                        float* loss, float* g1, float* g2, float grad_scale, int accumulate, void* ws, void* stream, const char* who) {
  IDG_REQUIRE(view1 && view2 && users && items && loss && ws, "%s: NULL argument", who);
  IDG_REQUIRE(n > 0 && d > 0 && B > 0 && num_users >= 0 && num_users <= n, "%s: bad sizes", who);
  IDG_REQUIRE(B <= 46340, "%s: batch of %lld ids is too large for the in-batch logits matrix", who, (long long)B);
  IDG_REQUIRE(temperature > 0.f, "%s: temperature must be positive", who);
  hipStream_t st = (hipStream_t)stream;
  const SslWs w = ssl_layout(ws, n, B, d);
  const int32_t* idx2 = cross ? w.idx2 : w.idx;
  const unsigned sets = cross ? 1u : 2u;
  if (!planned) {  // (planned: idg_infonce_plan has left the lists in this workspace, typically on another stream, a batch ahead)
    int rc = infonce_ids(users, items, B, num_users, n, w, cross ? 2 : (dedup ? 0 : 1), st);
    if (rc != IDG_OK) return rc;
  }
  const uint32_t* dup = (cross || !dedup) ? w.dup : nullptr;
  const int64_t rows_max = cross ? B : 2 * B;
  const unsigned row_blocks = (unsigned)((rows_max + (BLOCK / WAVE) - 1) / (BLOCK / WAVE));
  // the GEMM-shaped stages run on the fp32 matrix cores when every operand run is 16-byte aligned (IDG_SSL_MFMA=0: the SIMT
  // kernels, A/B timing)
  static const bool mfma_off = [] { const char* v = std::getenv("IDG_SSL_MFMA"); return v && *v && std::atoi(v) == 0; }();
  const bool mfma = !mfma_off && d % 64 == 0 && B % 4 == 0;
  const bool grads = g1 || g2;
  hipLaunchKernelGGL(ssl_normalize_kernel, dim3(row_blocks, 2), dim3(BLOCK), 0, st, view1, view2, d, w.idx, idx2, w.counts, B,
                     w.An, w.den);
  const unsigned tb = (unsigned)((B + TS - 1) / TS);
  const float inv_t = 1.0f / temperature;
  if (mfma) {
    const unsigned tc = (unsigned)((B + LT - 1) / LT);
    if ((int64_t)tc * tc * sets >= 512)  // enough 128 x 128 tiles for two workgroups on every CU
      hipLaunchKernelGGL(ssl_logits_mfma_kernel<2>, dim3(tc, tc, sets), dim3(BLOCK), 0, st, w.An, d, B, w.counts, inv_t, w.P);
    else
      hipLaunchKernelGGL(ssl_logits_mfma_kernel<1>, dim3(tc, (unsigned)((B + 63) / 64), sets), dim3(BLOCK), 0, st, w.An, d, B, w.counts,
                         inv_t, w.P);
  }
  else
    hipLaunchKernelGGL(ssl_logits_kernel, dim3(tb, tb, sets), dim3(BLOCK), 0, st, w.An, d, B, w.counts, inv_t, w.P);
  hipLaunchKernelGGL(ssl_rowstat_kernel, dim3((unsigned)((B + (BLOCK / WAVE) - 1) / (BLOCK / WAVE)), sets), dim3(BLOCK), 0, st,
                     w.P, B, w.counts, inv_t, 10e-6f, w.invttl, w.w, w.lossrow, (mfma && grads) ? w.coef : nullptr);
  hipLaunchKernelGGL(ssl_loss_kernel, dim3(sets), dim3(BLOCK), 0, st, w.lossrow, w.counts, loss);
  if (g1 || g2) {
    if (mfma)
      hipLaunchKernelGGL(ssl_grad_mfma_kernel, dim3((unsigned)((d + TS - 1) / TS) * GS, tb, 2 * sets), dim3(BLOCK), 0, st, w.P, w.An,
                         w.coef, d, B, w.counts, w.G);
    else
      hipLaunchKernelGGL(ssl_grad_kernel, dim3((unsigned)((d + TS - 1) / TS) * GS, tb, 2 * sets), dim3(BLOCK), 0, st, w.An, w.P, d,
                         B, w.counts, w.invttl, w.w, w.G);
    const int both = (g1 && g1 == g2) ? 1 : 0;  // one panel for both views: a single wave per row adds them in turn
    hipLaunchKernelGGL(ssl_contrib_kernel, dim3(row_blocks, 2), dim3(BLOCK), 0, st, w.An, w.den, w.G, w.invttl, w.w, d, B, w.counts,
                       grad_scale, w.contrib);
    hipLaunchKernelGGL(ssl_final_kernel, dim3(row_blocks, both ? 1 : 2), dim3(BLOCK), 0, st, w.contrib, d, B, w.idx, idx2, w.counts,
                       (dedup && !cross) ? 1 : 0, accumulate ? 1 : 0, both, g1, g2, dup, w.meta, w.dlist);
  }
  IDG_HIP(hipGetLastError());
  return IDG_OK;
}

int idg_infonce_plan(const int64_t* users, const int64_t* items, int64_t B, int64_t num_users, int64_t n, int64_t d, int mode,
                     void* ws, void* stream) {
  IDG_REQUIRE(users && items && ws, "idg_infonce_plan: NULL argument");
  IDG_REQUIRE(n > 0 && d > 0 && B > 0 && B <= 46340 && num_users >= 0 && num_users <= n, "idg_infonce_plan: bad sizes");
  IDG_REQUIRE(mode >= 0 && mode <= 2, "idg_infonce_plan: mode %d (0 unique ids, 1 raw lists, 2 cross form)", mode);
  return infonce_ids(users, items, B, num_users, n, ssl_layout(ws, n, B, d), mode, (hipStream_t)stream);
}

int idg_infonce_pair_f32(const float* view1, const float* view2, int64_t n, int64_t d, const int64_t* users,
                         const int64_t* items, int64_t B, int64_t num_users, int dedup, float temperature, float* loss,
                         float* g1, float* g2, float grad_scale, int accumulate, void* ws, void* stream) {
  return infonce_impl(view1, view2, n, d, users, items, B, num_users, dedup & 1, 0, (dedup & IDG_SSL_PLANNED) ? 1 : 0, temperature,
                      loss, g1, g2, grad_scale, accumulate, ws, stream, "idg_infonce_pair_f32");
}

int idg_infonce_cross_ex_f32(const float* view, int64_t n, int64_t d, const int64_t* users, const int64_t* items, int64_t B,
                             int64_t num_users, float temperature, float* loss, float* g, float grad_scale, int planned, void* ws,
                             void* stream) {
  // a_i = normalize(view[users[i]]), b_i = normalize(view[num_users + items[i]]), raw ids in batch order; the gradients
  // of both sides are ADDED into g's rows (users' and items' rows are disjoint ranges of the panel)
  return infonce_impl(view, view, n, d, users, items, B, num_users, 0, 1, planned ? 1 : 0, temperature, loss, g, g, grad_scale, 1,
                      ws, stream, "idg_infonce_cross_f32");
}

int idg_infonce_cross_f32(const float* view, int64_t n, int64_t d, const int64_t* users, const int64_t* items, int64_t B,
                          int64_t num_users, float temperature, float* loss, float* g, float grad_scale, void* ws,
                          void* stream) {
  return idg_infonce_cross_ex_f32(view, n, d, users, items, B, num_users, temperature, loss, g, grad_scale, 0, ws, stream);
}

}  // extern "C"
