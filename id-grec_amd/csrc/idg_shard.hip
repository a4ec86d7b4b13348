// Row movers and the item-row tail of the user-row-sharded training step (id-grec_amd/sharded.py).
// The reference trains on one device (utility/utility_train/trainer.py:36-56); these kernels are what sits between its
// step's products when the user rows are cut across GPUs: the batch's user rows travel through "guest" rows, the item
// rows a batch touches travel as compact row sets, and each rank finishes the gradient and applies Adam for the 1/N of
// the item rows it owns.  All of them are HBM / latency bound row copies: one wave per row, 16-byte lanes, no atomics.
#include <hip/hip_runtime.h>

#include <cmath>
#include <cstdint>

#include "idg_common.h"

namespace {

constexpr int BLOCK = 256;
constexpr int WAVE = 64;

__device__ __forceinline__ bool bit_at(const uint32_t* __restrict__ bits, int64_t r) { return (bits[r >> 5] >> (r & 31)) & 1u; }

// dstP[t] = idx[t] >= 0 ? srcP[idx[t]] : 0 for up to two (dst, src) panel pairs sharing one index list
__global__ __launch_bounds__(BLOCK) void rows_gather2_kernel(float* __restrict__ dst0, const float* __restrict__ src0,
                                                             float* __restrict__ dst1, const float* __restrict__ src1,
                                                             const int64_t* __restrict__ idx, int64_t count, int64_t d) {
  const int64_t w = (int64_t)blockIdx.x * (BLOCK / WAVE) + threadIdx.x / WAVE;
  const int64_t t = w >> 1;
  if (t >= count) return;
  float* dst = (w & 1) ? dst1 : dst0;
  const float* src = (w & 1) ? src1 : src0;
  if (!dst) return;
  const int64_t r = idx[t];
  for (int64_t f = (threadIdx.x % WAVE) * 4; f < d; f += WAVE * 4) {
    float4 v = make_float4(0.f, 0.f, 0.f, 0.f);
    if (r >= 0) v = *reinterpret_cast<const float4*>(src + r * d + f);
    *reinterpret_cast<float4*>(dst + t * d + f) = v;
  }
}

// dst[idx[j]] = src[j]  (idx distinct, >= 0)
__global__ __launch_bounds__(BLOCK) void rows_scatter_kernel(float* __restrict__ dst, const int64_t* __restrict__ idx,
                                                             const float* __restrict__ src, int64_t count, int64_t d) {
  const int64_t j = (int64_t)blockIdx.x * (BLOCK / WAVE) + threadIdx.x / WAVE;
  if (j >= count) return;
  const int64_t r = idx[j];
  for (int64_t f = (threadIdx.x % WAVE) * 4; f < d; f += WAVE * 4)
    *reinterpret_cast<float4*>(dst + r * d + f) = *reinterpret_cast<const float4*>(src + j * d + f);
}

// For every head t (idx[t] >= 0): dstP[idx[t]] = srcP[t] + srcP[next[t]] + ... in chain order — STORED, so the
// destination panel needs no zero-fill (its other rows are never read: the consumers go by the row bitmap).
__global__ __launch_bounds__(BLOCK) void rows_chain_store2_kernel(float* __restrict__ dst0, const float* __restrict__ src0,
                                                                  float* __restrict__ dst1, const float* __restrict__ src1,
                                                                  const int64_t* __restrict__ idx,
                                                                  const int64_t* __restrict__ next, int64_t count, int64_t d) {
  const int64_t w = (int64_t)blockIdx.x * (BLOCK / WAVE) + threadIdx.x / WAVE;
  const int64_t t = w >> 1;
  if (t >= count) return;
  float* dst = (w & 1) ? dst1 : dst0;
  const float* src = (w & 1) ? src1 : src0;
  if (!dst) return;
  const int64_t r = idx[t];
  if (r < 0) return;
  for (int64_t f = (threadIdx.x % WAVE) * 4; f < d; f += WAVE * 4) {
    float4 acc = *reinterpret_cast<const float4*>(src + t * d + f);
    for (int64_t j = next[t]; j >= 0; j = next[j]) {
      const float4 x = *reinterpret_cast<const float4*>(src + j * d + f);
      acc.x += x.x, acc.y += x.y, acc.z += x.z, acc.w += x.w;
    }
    *reinterpret_cast<float4*>(dst + r * d + f) = acc;
  }
}

// out[ids[j]] = (((a[ids[j]] + b[ids[j]]) + c[ids[j]]) + last[j]) / div — the layer mean (models/LightGCN.py:47-48, in
// torch.mean(torch.stack(...))'s left-to-right order) at the few item rows a training step reads; absent terms skipped
__global__ __launch_bounds__(BLOCK) void rows_layer_mean_kernel(float* __restrict__ out, const int64_t* __restrict__ ids,
                                                                int64_t count, const float* __restrict__ a,
                                                                const float* __restrict__ b, const float* __restrict__ c,
                                                                const float* __restrict__ last, float div, int64_t d) {
  const int64_t j = (int64_t)blockIdx.x * (BLOCK / WAVE) + threadIdx.x / WAVE;
  if (j >= count) return;
  const int64_t r = ids[j];
  for (int64_t f = (threadIdx.x % WAVE) * 4; f < d; f += WAVE * 4) {
    float4 s = *reinterpret_cast<const float4*>(last + j * d + f);
    const float* terms[3] = {a, b, c};
    bool have = false;
    float4 t = make_float4(0.f, 0.f, 0.f, 0.f);
#pragma unroll
    for (int k = 0; k < 3; ++k)
      if (terms[k]) {
        const float4 x = *reinterpret_cast<const float4*>(terms[k] + r * d + f);
        if (have) t.x += x.x, t.y += x.y, t.z += x.z, t.w += x.w;
        else t = x, have = true;
      }
    if (have) s.x = t.x + s.x, s.y = t.y + s.y, s.z = t.z + s.z, s.w = t.w + s.w;
    if (div != 1.0f) s.x = s.x / div, s.y = s.y / div, s.z = s.z / div, s.w = s.w / div;
    *reinterpret_cast<float4*>(out + r * d + f) = s;
  }
}

// The item-row tail of a sharded step, for a block of rows this rank owns: finish the gradient from the reduced
// partial sums t of the last backward product and apply Adam —
//   s = (c0 && live ? g + t : t) / cnt;  s = live ? G + s : s;  [G = s];  Adam(p, m, v; s)
// live = the row is one of the batch's items (bit row0 + r of `bits`): only there do g (d loss / d final) and G (the
// regulariser's gradient) hold anything.  Operation for operation what the last backward epilogue of the single-device
// step computes (idg_graph.hip: sum_in + acc, / div, sum_out + s, then EPI_ADAM).
__global__ __launch_bounds__(BLOCK) void grad_tail_adam_kernel(const float* __restrict__ t, const float* __restrict__ g,
                                                               float* __restrict__ G, const uint32_t* __restrict__ bits,
                                                               int64_t row0, int64_t rows, int64_t d, int c0, float cnt,
                                                               int store_grad, float* __restrict__ p, float* __restrict__ m,
                                                               float* __restrict__ v, float w1, float beta2, float w2,
                                                               float step_size, float bc2_sqrt, float eps) {
  const int64_t d4 = d / 4;
  const int64_t n4 = rows * d4;
  const int64_t stride = (int64_t)gridDim.x * BLOCK;
  for (int64_t i = (int64_t)blockIdx.x * BLOCK + threadIdx.x; i < n4; i += stride) {
    const int64_t r = i / d4;
    const bool live = bits != nullptr && bit_at(bits, row0 + r);
    float4 s = reinterpret_cast<const float4*>(t)[i];
    if (c0 && live) {
      const float4 x = reinterpret_cast<const float4*>(g)[i];
      s.x = x.x + s.x, s.y = x.y + s.y, s.z = x.z + s.z, s.w = x.w + s.w;
    }
    s.x = s.x / cnt, s.y = s.y / cnt, s.z = s.z / cnt, s.w = s.w / cnt;
    if (live) {
      const float4 x = reinterpret_cast<const float4*>(G)[i];
      s.x = x.x + s.x, s.y = x.y + s.y, s.z = x.z + s.z, s.w = x.w + s.w;
    }
    if (store_grad) reinterpret_cast<float4*>(G)[i] = s;
    float4 P = reinterpret_cast<float4*>(p)[i];
    float4 M = reinterpret_cast<float4*>(m)[i];
    float4 V = reinterpret_cast<float4*>(v)[i];
#define IDG_ADAM1(c)                                                 \
  M.c = __builtin_fmaf(w1, s.c - M.c, M.c);                          \
  V.c = __builtin_fmaf(w2 * s.c, s.c, V.c * beta2);                  \
  P.c = P.c - step_size * (M.c / (sqrtf(V.c) / bc2_sqrt + eps));
    IDG_ADAM1(x) IDG_ADAM1(y) IDG_ADAM1(z) IDG_ADAM1(w)
#undef IDG_ADAM1
    reinterpret_cast<float4*>(p)[i] = P;
    reinterpret_cast<float4*>(m)[i] = M;
    reinterpret_cast<float4*>(v)[i] = V;
  }
}

// The same with any number of earlier terms (K > 3 propagation layers): terms[0..n_terms) added left to right, then `last`.
constexpr int MEAN_TERMS_MAX = 15;
struct MeanTerms {
  const float* p[MEAN_TERMS_MAX];
  int n;
};
__global__ __launch_bounds__(BLOCK) void rows_layer_mean_n_kernel(float* __restrict__ out, const int64_t* __restrict__ ids,
                                                                  int64_t count, MeanTerms terms,
                                                                  const float* __restrict__ last, float div, int64_t d) {
  const int64_t j = (int64_t)blockIdx.x * (BLOCK / WAVE) + threadIdx.x / WAVE;
  if (j >= count) return;
  const int64_t r = ids[j];
  for (int64_t f = (threadIdx.x % WAVE) * 4; f < d; f += WAVE * 4) {
    float4 s = *reinterpret_cast<const float4*>(last + j * d + f);
    if (terms.n > 0) {
      float4 t = *reinterpret_cast<const float4*>(terms.p[0] + r * d + f);
      for (int k = 1; k < terms.n; ++k) {
        const float4 x = *reinterpret_cast<const float4*>(terms.p[k] + r * d + f);
        t.x += x.x, t.y += x.y, t.z += x.z, t.w += x.w;
      }
      s.x = t.x + s.x, s.y = t.y + s.y, s.z = t.z + s.z, s.w = t.w + s.w;
    }
    if (div != 1.0f) s.x = s.x / div, s.y = s.y / div, s.z = s.z / div, s.w = s.w / div;
    *reinterpret_cast<float4*>(out + r * d + f) = s;
  }
}

// ---- ascending ids of the non-zero entries of a flag vector, into a FIXED-capacity list (no host read-back): the
// sharded step's touched-item agreement.  Three launches: per-block counts (COMPACT_SPAN flags per block), one block
// scanning the counts, per-block ordered write; slots past the last id repeat it (a consumer that gathers, reduces and
// scatters rows through the list then moves that row more than once — the same value every time).
constexpr int COMPACT_PER_THREAD = 16;
constexpr int COMPACT_SPAN = BLOCK * COMPACT_PER_THREAD;

__device__ __forceinline__ int block_exclusive_scan(int v, int* lds, int* total) {
  // BLOCK threads; returns the exclusive prefix of v in thread order
  const int lane = threadIdx.x % WAVE, wave = threadIdx.x / WAVE;
  int inc = v;
#pragma unroll
  for (int off = 1; off < WAVE; off <<= 1) {
    const int o = __shfl_up(inc, off, WAVE);
    if (lane >= off) inc += o;
  }
  if (lane == WAVE - 1) lds[wave] = inc;
  __syncthreads();
  int base = 0, sum = 0;
  for (int w = 0; w < BLOCK / WAVE; ++w) {
    if (w < wave) base += lds[w];
    sum += lds[w];
  }
  if (total) *total = sum;
  __syncthreads();
  return base + inc - v;
}

__global__ __launch_bounds__(BLOCK) void flags_count_kernel(const float* __restrict__ flags, int64_t n, int32_t* __restrict__ counts) {
  __shared__ int lds[BLOCK / WAVE];
  const int64_t i0 = (int64_t)blockIdx.x * COMPACT_SPAN + (int64_t)threadIdx.x * COMPACT_PER_THREAD;
  int c = 0;
#pragma unroll
  for (int k = 0; k < COMPACT_PER_THREAD; ++k)
    if (i0 + k < n && flags[i0 + k] != 0.0f) ++c;
  int total;
  block_exclusive_scan(c, lds, &total);
  if (threadIdx.x == 0) counts[blockIdx.x] = total;
}

// counts[0..nb) -> exclusive offsets in place (int64 running total kept in a register); counts[nb] = min(total, INT32_MAX)
__global__ __launch_bounds__(BLOCK) void flags_scan_kernel(int32_t* __restrict__ counts, int64_t nb, int64_t* __restrict__ count_out) {
  __shared__ int lds[BLOCK / WAVE];
  int64_t run = 0;
  for (int64_t b0 = 0; b0 < nb; b0 += BLOCK) {
    const int64_t b = b0 + threadIdx.x;
    const int v = b < nb ? counts[b] : 0;
    int total;
    const int ex = block_exclusive_scan(v, lds, &total);
    // offsets beyond int32 cannot be addressed by the list anyway (its capacity is far below): saturate
    const int64_t off = run + ex;
    if (b < nb) counts[b] = off > 0x7fffffff ? 0x7fffffff : (int32_t)off;
    run += total;
  }
  if (threadIdx.x == 0) {
    counts[nb] = run > 0x7fffffff ? 0x7fffffff : (int32_t)run;
    if (count_out) *count_out = run;
  }
}

__global__ __launch_bounds__(BLOCK) void flags_write_kernel(const float* __restrict__ flags, int64_t n,
                                                            const int32_t* __restrict__ offsets, int64_t* __restrict__ ids,
                                                            int64_t cap) {
  __shared__ int lds[BLOCK / WAVE];
  const int64_t i0 = (int64_t)blockIdx.x * COMPACT_SPAN + (int64_t)threadIdx.x * COMPACT_PER_THREAD;
  int c = 0;
  unsigned m = 0;
#pragma unroll
  for (int k = 0; k < COMPACT_PER_THREAD; ++k)
    if (i0 + k < n && flags[i0 + k] != 0.0f) ++c, m |= 1u << k;
  int64_t at = (int64_t)offsets[blockIdx.x] + block_exclusive_scan(c, lds, nullptr);
#pragma unroll
  for (int k = 0; k < COMPACT_PER_THREAD; ++k)
    if ((m >> k) & 1u) {
      if (at < cap) ids[at] = i0 + k;
      ++at;
    }
}

// slots [total, cap) repeat the last id (total >= 1); an empty list is filled with row 0
__global__ __launch_bounds__(BLOCK) void flags_pad_kernel(const int32_t* __restrict__ offsets, int64_t nb, int64_t* __restrict__ ids,
                                                          int64_t cap) {
  const int64_t total = offsets[nb];
  const int64_t j = (int64_t)blockIdx.x * BLOCK + threadIdx.x;
  if (j >= cap || j < total) return;
  ids[j] = total > 0 ? ids[(total < cap ? total : cap) - 1] : 0;
}

// ---- 24-bit panels for the sharded step's exchanges (opt-in: sharded.Packed24Comm) --------------------------------
// An fp32 word keeps its sign, its 8 exponent bits and the upper 15 of its 23 mantissa bits, the dropped byte rounded to
// nearest (ties to even): 2^-16 = 1.5e-5 relative, inside the 1e-4 the north star allows, for 3/4 of the bytes on the
// links.  Four values A B C D (their upper 24 bits) travel as three words: A | B << 24,  B >> 8 | C << 16,  C >> 16 | D << 8.
// The SUM over the ranks is taken by reduce24_kernel in RANK ORDER ((q0 + q1) + q2 ...), one sequence of fp32 adds per
// element: a k-GPU run is bit-reproducible run to run whatever algorithm RCCL would have picked.
__device__ __forceinline__ uint32_t top24(float x) {
  const uint32_t b = __float_as_uint(x);
  return (b + 0x7Fu + ((b >> 8) & 1u)) >> 8;  // (a carry out of the mantissa moves the exponent up: still the nearest)
}
__device__ __forceinline__ void pack4(const float4 v, uint32_t& w0, uint32_t& w1, uint32_t& w2) {
  const uint32_t a = top24(v.x), b = top24(v.y), c = top24(v.z), d = top24(v.w);
  w0 = a | (b << 24), w1 = (b >> 8) | (c << 16), w2 = (c >> 16) | (d << 8);
}
__device__ __forceinline__ float4 unpack4(uint32_t w0, uint32_t w1, uint32_t w2) {
  return make_float4(__uint_as_float((w0 & 0xFFFFFFu) << 8), __uint_as_float(((w0 >> 24) | ((w1 & 0xFFFFu) << 8)) << 8),
                     __uint_as_float(((w1 >> 16) | ((w2 & 0xFFu) << 16)) << 8), __uint_as_float((w2 >> 8) << 8));
}

// one thread = 16 values = 64 B of fp32 <-> 48 B packed (whole 16-byte accesses on both sides); the < 16 values at the end
// of an array go four at a time
__global__ __launch_bounds__(BLOCK) void pack24_kernel(const float* __restrict__ src, uint32_t* __restrict__ dst, int64_t n) {
  const int64_t t = (int64_t)blockIdx.x * BLOCK + threadIdx.x;
  const int64_t n16 = n >> 4;
  if (t < n16) {
    const float4* in = reinterpret_cast<const float4*>(src) + 4 * t;
    uint32_t w[12];
#pragma unroll
    for (int q = 0; q < 4; ++q) pack4(in[q], w[3 * q], w[3 * q + 1], w[3 * q + 2]);
    uint4* out = reinterpret_cast<uint4*>(dst) + 3 * t;
    out[0] = make_uint4(w[0], w[1], w[2], w[3]), out[1] = make_uint4(w[4], w[5], w[6], w[7]), out[2] = make_uint4(w[8], w[9], w[10], w[11]);
  } else {
    const int64_t g = 4 * n16 + (t - n16);  // group of four values
    if (4 * g < n) pack4(reinterpret_cast<const float4*>(src)[g], dst[3 * g], dst[3 * g + 1], dst[3 * g + 2]);
  }
}

__global__ __launch_bounds__(BLOCK) void unpack24_kernel(const uint32_t* __restrict__ src, float* __restrict__ dst, int64_t n) {
  const int64_t t = (int64_t)blockIdx.x * BLOCK + threadIdx.x;
  const int64_t n16 = n >> 4;
  if (t < n16) {
    const uint4* in = reinterpret_cast<const uint4*>(src) + 3 * t;
    const uint4 a = in[0], b = in[1], c = in[2];
    const uint32_t w[12] = {a.x, a.y, a.z, a.w, b.x, b.y, b.z, b.w, c.x, c.y, c.z, c.w};
    float4* out = reinterpret_cast<float4*>(dst) + 4 * t;
#pragma unroll
    for (int q = 0; q < 4; ++q) out[q] = unpack4(w[3 * q], w[3 * q + 1], w[3 * q + 2]);
  } else {
    const int64_t g = 4 * n16 + (t - n16);
    if (4 * g < n) reinterpret_cast<float4*>(dst)[g] = unpack4(src[3 * g], src[3 * g + 1], src[3 * g + 2]);
  }
}

// out = sum over b = 0 .. n_blocks - 1, IN THAT ORDER, of the packed blocks (block b = the 3n/4 words at blocks + b * 3n/4):
// as fp32 (out_f32: a reduce-scatter's result) and / or packed again (out_packed: what an all-gather sends on)
__global__ __launch_bounds__(BLOCK) void reduce24_kernel(const uint32_t* __restrict__ blocks, int n_blocks, int64_t n,
                                                        uint32_t* __restrict__ out_packed, float* __restrict__ out_f32) {
  const int64_t g = (int64_t)blockIdx.x * BLOCK + threadIdx.x;  // group of four values
  if (4 * g >= n) return;
  const int64_t words = n / 4 * 3;
  const uint32_t* p = blocks + 3 * g;
  float4 acc = unpack4(p[0], p[1], p[2]);
  for (int b = 1; b < n_blocks; ++b) {
    p += words;
    const float4 v = unpack4(p[0], p[1], p[2]);
    acc.x += v.x, acc.y += v.y, acc.z += v.z, acc.w += v.w;
  }
  if (out_f32) reinterpret_cast<float4*>(out_f32)[g] = acc;
  if (out_packed) pack4(acc, out_packed[3 * g], out_packed[3 * g + 1], out_packed[3 * g + 2]);
}

// out = blocks[0] + blocks[1] + ... IN THAT ORDER, fp32 blocks of n values (the rank-ordered sum without the 24-bit packing:
// sharded.RankOrderComm with 32 bits — the same bytes on the links as RCCL's all-reduce, but one fixed sequence of adds)
__global__ __launch_bounds__(BLOCK) void reduce_blocks_kernel(const float* __restrict__ blocks, int n_blocks, int64_t n,
                                                             float* __restrict__ out) {
  const int64_t g = (int64_t)blockIdx.x * BLOCK + threadIdx.x;
  if (4 * g >= n) return;
  const float4* p = reinterpret_cast<const float4*>(blocks) + g;
  float4 acc = *p;
  for (int b = 1; b < n_blocks; ++b) {
    p += n / 4;
    const float4 v = *p;
    acc.x += v.x, acc.y += v.y, acc.z += v.z, acc.w += v.w;
  }
  reinterpret_cast<float4*>(out)[g] = acc;
}

}  // namespace

extern "C" {

int idg_reduce_blocks_f32(const float* blocks, int n_blocks, int64_t n, float* out, void* stream) {
  IDG_REQUIRE(blocks && out && n_blocks >= 1 && n >= 0 && n % 4 == 0, "idg_reduce_blocks_f32: NULL buffer, no block, or a count that is not a multiple of 4");
  IDG_REQUIRE(((uintptr_t)blocks | (uintptr_t)out) % 16 == 0, "idg_reduce_blocks_f32: buffers must be 16-byte aligned");
  if (n == 0) return IDG_OK;
  hipLaunchKernelGGL(reduce_blocks_kernel, dim3((unsigned)((n / 4 + BLOCK - 1) / BLOCK)), dim3(BLOCK), 0, (hipStream_t)stream, blocks,
                     n_blocks, n, out);
  IDG_HIP(hipGetLastError());
  return IDG_OK;
}

int idg_pack24_f32(const float* src, uint32_t* dst, int64_t n, void* stream) {
  IDG_REQUIRE(src && dst && n >= 0 && n % 4 == 0, "idg_pack24_f32: NULL buffer, or a count that is not a multiple of 4");
  IDG_REQUIRE((uintptr_t)src % 16 == 0 && (uintptr_t)dst % 16 == 0, "idg_pack24_f32: buffers must be 16-byte aligned");
  if (n == 0) return IDG_OK;
  const int64_t threads = (n >> 4) + ((n & 15) >> 2);
  hipLaunchKernelGGL(pack24_kernel, dim3((unsigned)((threads + BLOCK - 1) / BLOCK)), dim3(BLOCK), 0, (hipStream_t)stream, src, dst, n);
  IDG_HIP(hipGetLastError());
  return IDG_OK;
}

int idg_unpack24_f32(const uint32_t* src, float* dst, int64_t n, void* stream) {
  IDG_REQUIRE(src && dst && n >= 0 && n % 4 == 0, "idg_unpack24_f32: NULL buffer, or a count that is not a multiple of 4");
  IDG_REQUIRE((uintptr_t)src % 16 == 0 && (uintptr_t)dst % 16 == 0, "idg_unpack24_f32: buffers must be 16-byte aligned");
  if (n == 0) return IDG_OK;
  const int64_t threads = (n >> 4) + ((n & 15) >> 2);
  hipLaunchKernelGGL(unpack24_kernel, dim3((unsigned)((threads + BLOCK - 1) / BLOCK)), dim3(BLOCK), 0, (hipStream_t)stream, src, dst, n);
  IDG_HIP(hipGetLastError());
  return IDG_OK;
}

int idg_reduce24_f32(const uint32_t* blocks, int n_blocks, int64_t n, uint32_t* out_packed, float* out_f32, void* stream) {
  IDG_REQUIRE(blocks && n_blocks >= 1 && n >= 0 && n % 4 == 0, "idg_reduce24_f32: NULL buffer, no block, or a count that is not a multiple of 4");
  IDG_REQUIRE(out_packed || out_f32, "idg_reduce24_f32: nowhere to put the sum");
  IDG_REQUIRE(!out_f32 || (uintptr_t)out_f32 % 16 == 0, "idg_reduce24_f32: out_f32 must be 16-byte aligned");
  if (n == 0) return IDG_OK;
  hipLaunchKernelGGL(reduce24_kernel, dim3((unsigned)((n / 4 + BLOCK - 1) / BLOCK)), dim3(BLOCK), 0, (hipStream_t)stream, blocks,
                     n_blocks, n, out_packed, out_f32);
  IDG_HIP(hipGetLastError());
  return IDG_OK;
}


int idg_rows_gather2_f32(float* dst0, const float* src0, float* dst1, const float* src1, const int64_t* idx, int64_t count,
                         int64_t d, void* stream) {
  IDG_REQUIRE(dst0 && src0 && idx && count >= 0 && d > 0 && d % 4 == 0, "idg_rows_gather2_f32: bad argument");
  IDG_REQUIRE((dst1 == nullptr) == (src1 == nullptr), "idg_rows_gather2_f32: the second pair is both or neither");
  IDG_REQUIRE(((uintptr_t)dst0 | (uintptr_t)src0 | (uintptr_t)dst1 | (uintptr_t)src1) % 16 == 0,
              "idg_rows_gather2_f32: panels must be 16-byte aligned");
  if (count == 0) return IDG_OK;
  const int64_t waves = 2 * count;
  hipLaunchKernelGGL(rows_gather2_kernel, dim3((unsigned)((waves + BLOCK / WAVE - 1) / (BLOCK / WAVE))), dim3(BLOCK), 0,
                     (hipStream_t)stream, dst0, src0, dst1, src1, idx, count, d);
  IDG_HIP(hipGetLastError());
  return IDG_OK;
}

int idg_rows_scatter_f32(float* dst, const int64_t* idx, const float* src, int64_t count, int64_t d, void* stream) {
  IDG_REQUIRE(dst && src && idx && count >= 0 && d > 0 && d % 4 == 0, "idg_rows_scatter_f32: bad argument");
  IDG_REQUIRE(((uintptr_t)dst | (uintptr_t)src) % 16 == 0, "idg_rows_scatter_f32: panels must be 16-byte aligned");
  if (count == 0) return IDG_OK;
  hipLaunchKernelGGL(rows_scatter_kernel, dim3((unsigned)((count + BLOCK / WAVE - 1) / (BLOCK / WAVE))), dim3(BLOCK), 0,
                     (hipStream_t)stream, dst, idx, src, count, d);
  IDG_HIP(hipGetLastError());
  return IDG_OK;
}

int idg_rows_chain_store2_f32(float* dst0, const float* src0, float* dst1, const float* src1, const int64_t* idx,
                              const int64_t* next, int64_t count, int64_t d, void* stream) {
  IDG_REQUIRE(dst0 && src0 && idx && next && count >= 0 && d > 0 && d % 4 == 0, "idg_rows_chain_store2_f32: bad argument");
  IDG_REQUIRE((dst1 == nullptr) == (src1 == nullptr), "idg_rows_chain_store2_f32: the second pair is both or neither");
  IDG_REQUIRE(((uintptr_t)dst0 | (uintptr_t)src0 | (uintptr_t)dst1 | (uintptr_t)src1) % 16 == 0,
              "idg_rows_chain_store2_f32: panels must be 16-byte aligned");
  if (count == 0) return IDG_OK;
  const int64_t waves = 2 * count;
  hipLaunchKernelGGL(rows_chain_store2_kernel, dim3((unsigned)((waves + BLOCK / WAVE - 1) / (BLOCK / WAVE))), dim3(BLOCK), 0,
                     (hipStream_t)stream, dst0, src0, dst1, src1, idx, next, count, d);
  IDG_HIP(hipGetLastError());
  return IDG_OK;
}

int idg_rows_layer_mean_f32(float* out, const int64_t* ids, int64_t count, const float* a, const float* b, const float* c,
                            const float* last, float div, int64_t d, void* stream) {
  IDG_REQUIRE(out && ids && last && count >= 0 && d > 0 && d % 4 == 0 && div != 0.f, "idg_rows_layer_mean_f32: bad argument");
  IDG_REQUIRE(((uintptr_t)out | (uintptr_t)a | (uintptr_t)b | (uintptr_t)c | (uintptr_t)last) % 16 == 0,
              "idg_rows_layer_mean_f32: panels must be 16-byte aligned");
  if (count == 0) return IDG_OK;
  hipLaunchKernelGGL(rows_layer_mean_kernel, dim3((unsigned)((count + BLOCK / WAVE - 1) / (BLOCK / WAVE))), dim3(BLOCK), 0,
                     (hipStream_t)stream, out, ids, count, a, b, c, last, div, d);
  IDG_HIP(hipGetLastError());
  return IDG_OK;
}

int idg_rows_layer_mean_n_f32(float* out, const int64_t* ids, int64_t count, const float* const* terms, int n_terms,
                              const float* last, float div, int64_t d, void* stream) {
  IDG_REQUIRE(out && ids && last && count >= 0 && d > 0 && d % 4 == 0 && div != 0.f, "idg_rows_layer_mean_n_f32: bad argument");
  IDG_REQUIRE(n_terms >= 0 && n_terms <= MEAN_TERMS_MAX && (n_terms == 0 || terms),
              "idg_rows_layer_mean_n_f32: 0 <= n_terms <= 15 term panels");
  MeanTerms t;
  t.n = n_terms;
  uintptr_t align = (uintptr_t)out | (uintptr_t)last;
  for (int k = 0; k < MEAN_TERMS_MAX; ++k) {
    t.p[k] = k < n_terms ? terms[k] : nullptr;
    IDG_REQUIRE(k >= n_terms || terms[k], "idg_rows_layer_mean_n_f32: NULL term panel");
    align |= (uintptr_t)t.p[k];
  }
  IDG_REQUIRE(align % 16 == 0, "idg_rows_layer_mean_n_f32: panels must be 16-byte aligned");
  if (count == 0) return IDG_OK;
  hipLaunchKernelGGL(rows_layer_mean_n_kernel, dim3((unsigned)((count + BLOCK / WAVE - 1) / (BLOCK / WAVE))), dim3(BLOCK), 0,
                     (hipStream_t)stream, out, ids, count, t, last, div, d);
  IDG_HIP(hipGetLastError());
  return IDG_OK;
}

size_t idg_flags_compact_workspace_bytes(int64_t n) {
  const int64_t nb = n > 0 ? (n + COMPACT_SPAN - 1) / COMPACT_SPAN : 0;
  return (size_t)(nb + 1) * sizeof(int32_t);
}

int idg_flags_compact_f32(const float* flags, int64_t n, int64_t* ids, int64_t cap, int64_t* count, void* ws, void* stream) {
  IDG_REQUIRE(flags && ids && ws && n > 0 && cap > 0, "idg_flags_compact_f32: bad argument");
  IDG_REQUIRE(n <= (int64_t)0x7fffffff * COMPACT_SPAN, "idg_flags_compact_f32: flag vector too long");
  const int64_t nb = (n + COMPACT_SPAN - 1) / COMPACT_SPAN;
  int32_t* counts = reinterpret_cast<int32_t*>(ws);
  hipStream_t st = (hipStream_t)stream;
  hipLaunchKernelGGL(flags_count_kernel, dim3((unsigned)nb), dim3(BLOCK), 0, st, flags, n, counts);
  hipLaunchKernelGGL(flags_scan_kernel, dim3(1), dim3(BLOCK), 0, st, counts, nb, count);
  hipLaunchKernelGGL(flags_write_kernel, dim3((unsigned)nb), dim3(BLOCK), 0, st, flags, n, counts, ids, cap);
  hipLaunchKernelGGL(flags_pad_kernel, dim3((unsigned)((cap + BLOCK - 1) / BLOCK)), dim3(BLOCK), 0, st, counts, nb, ids, cap);
  IDG_HIP(hipGetLastError());
  return IDG_OK;
}

int idg_grad_tail_adam_f32(const float* t, const float* g, float* G, const uint32_t* live_bits, int64_t row0, int64_t rows,
                           int64_t d, int include_layer0, float cnt, int store_grad, float* param, float* exp_avg,
                           float* exp_avg_sq, double lr, double beta1, double beta2, double eps, int64_t step, void* stream) {
  IDG_REQUIRE(t && param && exp_avg && exp_avg_sq && rows >= 0 && d > 0 && d % 4 == 0 && row0 >= 0,
              "idg_grad_tail_adam_f32: bad argument");
  IDG_REQUIRE(cnt != 0.f && step >= 1, "idg_grad_tail_adam_f32: cnt must be non-zero, step 1-based");
  IDG_REQUIRE(!live_bits || (g && G), "idg_grad_tail_adam_f32: live rows need the g and G panels");
  IDG_REQUIRE(!store_grad || G, "idg_grad_tail_adam_f32: store_grad needs G");
  IDG_REQUIRE(((uintptr_t)t | (uintptr_t)g | (uintptr_t)G | (uintptr_t)param | (uintptr_t)exp_avg | (uintptr_t)exp_avg_sq) % 16 == 0,
              "idg_grad_tail_adam_f32: panels must be 16-byte aligned");
  if (rows == 0) return IDG_OK;
  const double bc1 = 1.0 - std::pow(beta1, (double)step);
  const double bc2 = 1.0 - std::pow(beta2, (double)step);
  const int64_t n4 = rows * (d / 4);
  int64_t nb = (n4 + BLOCK - 1) / BLOCK;
  nb = nb < 1 ? 1 : (nb > 256 * 8 ? 256 * 8 : nb);
  hipLaunchKernelGGL(grad_tail_adam_kernel, dim3((unsigned)nb), dim3(BLOCK), 0, (hipStream_t)stream, t, g, G, live_bits, row0,
                     rows, d, include_layer0 ? 1 : 0, cnt, store_grad ? 1 : 0, param, exp_avg, exp_avg_sq, (float)(1.0 - beta1),
                     (float)beta2, (float)(1.0 - beta2), (float)(lr / bc1), (float)std::sqrt(bc2), (float)eps);
  IDG_HIP(hipGetLastError());
  return IDG_OK;
}

int idg_shard_prepare(const idg_shard_prep* p) {
  IDG_REQUIRE(p, "idg_shard_prepare: NULL argument");
  IDG_REQUIRE(p->pos && p->neg && p->guest_ids && p->users_bits && p->items_bits && p->scatter_bits && p->plan_ws && p->B > 0,
              "idg_shard_prepare: NULL bitmap / id array");
  IDG_REQUIRE(p->n_own == 0 || p->own_users, "idg_shard_prepare: owned users without their id array");
  IDG_REQUIRE(p->n_slices >= 0 && (p->n_slices == 0 || (p->slice_graphs && p->slice_row0 && p->slice_units)),
              "idg_shard_prepare: slice tables missing");
  // what the bitmaps and unit lists were SIZED for (a prepared object handed to another engine's batch would
  // write outside them)
  IDG_REQUIRE(p->B <= p->B_cap && p->n_own >= 0 && p->n_own <= p->B_cap,
              "idg_shard_prepare: batch (B, n_own) larger than the capacity B_cap the buffers were sized for");
  IDG_REQUIRE(p->n_local_users >= 0 && p->n_items_padded >= 0 && p->n_panel_rows >= p->n_local_users + p->B_cap,
              "idg_shard_prepare: inconsistent panel geometry");
  for (int j = 0; j < p->n_slices; ++j)
    IDG_REQUIRE(p->slice_row0[j] >= 0 && p->slice_row0[j] % 32 == 0 && p->slice_row0[j] <= p->n_items_padded && p->slice_graphs[j],
                "idg_shard_prepare: slice_row0 must be a multiple of 32 inside the item bitmap");
  void* side = p->side_stream;
  int rc;
#define IDG_TRY(call) \
  if ((rc = (call)) != IDG_OK) return rc
  // the id arrays may just have been produced on the step's stream, and the step that last used these buffers runs there
  if (p->ev_fork) {
    IDG_TRY(idg_event_record(p->ev_fork, p->main_stream));
    IDG_TRY(idg_stream_wait_event(side, p->ev_fork));
  }
  IDG_TRY(idg_bitmap_clear(p->users_bits, p->n_local_users, side));
  if (p->n_own > 0) IDG_TRY(idg_bpr_touch_rows(p->own_users, p->own_users, p->own_users, p->n_own, 0, p->users_bits, side));
  IDG_TRY(idg_bitmap_clear(p->items_bits, p->n_items_padded, side));
  IDG_TRY(idg_bpr_touch_rows(p->pos, p->pos, p->neg, p->B, 0, p->items_bits, side));
  IDG_TRY(idg_bitmap_clear(p->scatter_bits, p->n_panel_rows, side));
  if (p->user_graph && p->user_units)
    IDG_TRY(idg_graph_live_units(p->user_graph, p->users_bits, p->user_units, p->B_cap, side));
  for (int j = 0; j < p->n_slices; ++j)
    if (p->slice_units[j])
      IDG_TRY(idg_graph_live_units(p->slice_graphs[j], p->items_bits + p->slice_row0[j] / 32, p->slice_units[j], 2 * p->B_cap, side));
  if (p->ev_rows) IDG_TRY(idg_event_record(p->ev_rows, side));
  IDG_TRY(idg_bpr_plan_f32(p->guest_ids, p->pos, p->neg, p->B, p->n_local_users + p->B_cap, p->n_panel_rows, p->plan_ws, side));
  if (p->ev_plan) IDG_TRY(idg_event_record(p->ev_plan, side));
#undef IDG_TRY
  return IDG_OK;
}

}  // extern "C"
