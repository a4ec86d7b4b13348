// Weight gradient of a thin dense layer: W_grad[d1, d2] = X^T . G with X [n, d1], G [n, d2], n ~ 10^5..10^7 rows
// and d1, d2 ~ 64 (NGCF's per-layer d x d transforms, models/NGCF.py:91-99: autograd of torch.matmul(side, W)).
// The product is all reduction (K = n) and no output: library GEMMs pick a tile shape for it that leaves most of
// the chip idle (202 us at n = 69,716, d = 64 on hipBLASLt).  Here the rows are cut into slices, every workgroup
// accumulates a 64 x 64 tile over its slice from LDS-staged row chunks, and a second kernel adds the slices in
// slice order — deterministic, HBM-bound (reads X and G once).
#include <hip/hip_runtime.h>

#include <algorithm>
#include <cstdint>

#include "idg_common.h"
#include "idg_dropout.h"

namespace {

constexpr int BLOCK = 256;
constexpr int T = 64;    // output tile edge
constexpr int RC = 32;   // rows staged per round
constexpr int MAX_SLICES = 1024;

inline int64_t n_slices(int64_t n) {
  int64_t s = (n + 63) / 64;  // >= 64 rows per slice: ~10^3 workgroups at 10^5 rows
  return s < 1 ? 1 : (s > MAX_SLICES ? MAX_SLICES : s);
}

// grid: (slices, ceil(d2/64), ceil(d1/64))
__global__ __launch_bounds__(BLOCK) void wgrad_partial_kernel(const float* __restrict__ X, int64_t ldx,
                                                              const float* __restrict__ G, int64_t ldg, int64_t n,
                                                              int64_t d1, int64_t d2, float* __restrict__ part) {
  __shared__ float sx[RC][T + 1];
  __shared__ float sg[RC][T + 1];
  const int64_t slices = gridDim.x;
  const int64_t per = (n + slices - 1) / slices;
  const int64_t r_lo = (int64_t)blockIdx.x * per, r_hi = r_lo + per < n ? r_lo + per : n;
  const int64_t i0 = (int64_t)blockIdx.z * T, j0 = (int64_t)blockIdx.y * T;
  const int tid = threadIdx.x, tx = tid % 16, ty = tid / 16;
  float acc[4][4] = {};
  const bool vec = (ldx % 4 == 0) && (ldg % 4 == 0) && (((uintptr_t)X | (uintptr_t)G) % 16 == 0);
  for (int64_t r0 = r_lo; r0 < r_hi; r0 += RC) {
    if (vec && i0 + T <= d1 && j0 + T <= d2) {
      // 16-byte loads: RC rows x 16 float4 per operand, two per thread
      for (int e = tid; e < RC * (T / 4); e += BLOCK) {
        const int rr = e / (T / 4), c4 = (e % (T / 4)) * 4;
        float4 x = make_float4(0.f, 0.f, 0.f, 0.f), g = x;
        if (r0 + rr < r_hi) {
          x = *reinterpret_cast<const float4*>(X + (r0 + rr) * ldx + i0 + c4);
          g = *reinterpret_cast<const float4*>(G + (r0 + rr) * ldg + j0 + c4);
        }
        sx[rr][c4] = x.x, sx[rr][c4 + 1] = x.y, sx[rr][c4 + 2] = x.z, sx[rr][c4 + 3] = x.w;
        sg[rr][c4] = g.x, sg[rr][c4 + 1] = g.y, sg[rr][c4 + 2] = g.z, sg[rr][c4 + 3] = g.w;
      }
    } else {
      for (int e = tid; e < RC * T; e += BLOCK) {
        const int rr = e / T, c = e % T;
        const bool row_ok = r0 + rr < r_hi;
        sx[rr][c] = (row_ok && i0 + c < d1) ? X[(r0 + rr) * ldx + i0 + c] : 0.f;
        sg[rr][c] = (row_ok && j0 + c < d2) ? G[(r0 + rr) * ldg + j0 + c] : 0.f;
      }
    }
    __syncthreads();
#pragma unroll
    for (int rr = 0; rr < RC; ++rr) {
      float a[4], b[4];
#pragma unroll
      for (int q = 0; q < 4; ++q) a[q] = sx[rr][ty * 4 + q], b[q] = sg[rr][tx * 4 + q];
#pragma unroll
      for (int p = 0; p < 4; ++p)
#pragma unroll
        for (int q = 0; q < 4; ++q) acc[p][q] = __builtin_fmaf(a[p], b[q], acc[p][q]);
    }
    __syncthreads();
  }
  float* out = part + (int64_t)blockIdx.x * d1 * d2;
#pragma unroll
  for (int p = 0; p < 4; ++p)
#pragma unroll
    for (int q = 0; q < 4; ++q) {
      const int64_t i = i0 + ty * 4 + p, j = j0 + tx * 4 + q;
      if (i < d1 && j < d2) out[i * d2 + j] = acc[p][q];
    }
}

// 64 consecutive output elements per workgroup, 16 lane groups share the slices (group g adds slices g, g+16, ...
// in that order; the 16 group sums are then added g = 0..15): a fixed order, ~slices/16 loads per thread.
constexpr int RG = 16;

__global__ __launch_bounds__(64 * RG) void wgrad_reduce_kernel(const float* __restrict__ part, int64_t slices, int64_t count,
                                                               float* __restrict__ out, int accumulate) {
  __shared__ float s_sum[RG][64];
  const int lane = threadIdx.x % 64, g = threadIdx.x / 64;
  const int64_t e = (int64_t)blockIdx.x * 64 + lane;
  float s = 0.f;
  if (e < count) {
    // eight independent loads in flight, then added in slice order (a load per dependent add was one L2 round trip per
    // slice: 10.5 us for 512 slices of an NGCF layer's gradients, round 4)
    for (int64_t k0 = g; k0 < slices; k0 += 8 * RG) {
      float v[8];
#pragma unroll
      for (int q = 0; q < 8; ++q) v[q] = k0 + q * RG < slices ? part[(k0 + q * RG) * count + e] : 0.f;
#pragma unroll
      for (int q = 0; q < 8; ++q) s += v[q];
    }
  }
  s_sum[g][lane] = s;
  __syncthreads();
  if (g == 0 && e < count) {
    float t = s_sum[0][lane];
#pragma unroll
    for (int q = 1; q < RG; ++q) t += s_sum[q][lane];
    out[e] = accumulate ? out[e] + t : t;
  }
}

// ---- NGCF: the four parameter gradients of a layer in ONE pass over the rows (round 4).  g W_gcn = side^T gT,
// g W_bi = (ego * side)^T gT, g b_gcn = g b_bi = column sums of gT (models/NGCF.py:91-99 under autograd) share the right
// operand: staged once per row chunk.  Row slices on workgroups, slice sums added in slice order by wgrad_reduce_kernel
// (deterministic), straight into the layout [W_gcn | b_gcn | W_bi | b_bi] of the step's flat gradient buffer.  As four
// calls (two idg_linear_wgrad_f32 of 1024 slices + a column sum) this was 113 us per layer at yelp2018 size, a third of
// the fused step.
constexpr int NG_SLICES = 512;

// grid: (slices, ceil(d2/64), ceil(d1/64))
__global__ __launch_bounds__(BLOCK) void ngcf_wgrad_kernel(const float* __restrict__ X1, const float* __restrict__ X2,
                                                           const float* __restrict__ G, int64_t n, int64_t d1, int64_t d2,
                                                           float* __restrict__ part) {
  __shared__ float s1[RC][T + 1];
  __shared__ float s2[RC][T + 1];
  __shared__ float sg[RC][T + 1];
  const int64_t slices = gridDim.x;
  const int64_t per = (n + slices - 1) / slices;
  const int64_t r_lo = (int64_t)blockIdx.x * per, r_hi = r_lo + per < n ? r_lo + per : n;
  const int64_t i0 = (int64_t)blockIdx.z * T, j0 = (int64_t)blockIdx.y * T;
  const int tid = threadIdx.x, tx = tid % 16, ty = tid / 16;
  float a1[4][4] = {}, a2[4][4] = {}, cs[4] = {};
  for (int64_t r0 = r_lo; r0 < r_hi; r0 += RC) {
    for (int e = tid; e < RC * (T / 4); e += BLOCK) {  // d1, d2 multiples of 64, panels 16-byte aligned (checked by the caller)
      const int rr = e / (T / 4), c4 = (e % (T / 4)) * 4;
      float4 x = make_float4(0.f, 0.f, 0.f, 0.f), y = x, g = x;
      if (r0 + rr < r_hi) {
        x = *reinterpret_cast<const float4*>(X1 + (r0 + rr) * d1 + i0 + c4);
        y = *reinterpret_cast<const float4*>(X2 + (r0 + rr) * d1 + i0 + c4);
        g = *reinterpret_cast<const float4*>(G + (r0 + rr) * d2 + j0 + c4);
      }
      s1[rr][c4] = x.x, s1[rr][c4 + 1] = x.y, s1[rr][c4 + 2] = x.z, s1[rr][c4 + 3] = x.w;
      s2[rr][c4] = y.x, s2[rr][c4 + 1] = y.y, s2[rr][c4 + 2] = y.z, s2[rr][c4 + 3] = y.w;
      sg[rr][c4] = g.x, sg[rr][c4 + 1] = g.y, sg[rr][c4 + 2] = g.z, sg[rr][c4 + 3] = g.w;
    }
    __syncthreads();
#pragma unroll
    for (int rr = 0; rr < RC; ++rr) {
      float a[4], c[4], b[4];
#pragma unroll
      for (int q = 0; q < 4; ++q) a[q] = s1[rr][ty * 4 + q], c[q] = s2[rr][ty * 4 + q], b[q] = sg[rr][tx * 4 + q];
#pragma unroll
      for (int p = 0; p < 4; ++p)
#pragma unroll
        for (int q = 0; q < 4; ++q) {
          a1[p][q] = __builtin_fmaf(a[p], b[q], a1[p][q]);
          a2[p][q] = __builtin_fmaf(c[p], b[q], a2[p][q]);
        }
      if (ty == 0)
#pragma unroll
        for (int q = 0; q < 4; ++q) cs[q] += b[q];
    }
    __syncthreads();
  }
  const int64_t ww = d1 * d2;
  float* out = part + (int64_t)blockIdx.x * (2 * ww + 2 * d2);
#pragma unroll
  for (int p = 0; p < 4; ++p)
#pragma unroll
    for (int q = 0; q < 4; ++q) {
      const int64_t i = i0 + ty * 4 + p, j = j0 + tx * 4 + q;
      out[i * d2 + j] = a1[p][q];
      out[ww + d2 + i * d2 + j] = a2[p][q];
    }
  if (ty == 0 && blockIdx.z == 0)
#pragma unroll
    for (int q = 0; q < 4; ++q) {
      out[ww + j0 + tx * 4 + q] = cs[q];
      out[2 * ww + d2 + j0 + tx * 4 + q] = cs[q];
    }
}

}  // namespace

extern "C" {

size_t idg_ngcf_wgrad_workspace_bytes(int64_t d1, int64_t d2) {
  if (d1 <= 0 || d2 <= 0) return 0;
  return (size_t)NG_SLICES * (size_t)(2 * d1 * d2 + 2 * d2) * sizeof(float);
}

int idg_ngcf_wgrad_f32(const float* side, const float* bi, const float* gT, int64_t n, int64_t d1, int64_t d2, float* out,
                       void* ws, void* stream) {
  IDG_REQUIRE(side && bi && gT && out && ws, "idg_ngcf_wgrad_f32: NULL argument");
  IDG_REQUIRE(n > 0 && d1 > 0 && d2 > 0 && d1 % 64 == 0 && d2 % 64 == 0, "idg_ngcf_wgrad_f32: widths must be multiples of 64");
  IDG_REQUIRE(((uintptr_t)side | (uintptr_t)bi | (uintptr_t)gT) % 16 == 0, "idg_ngcf_wgrad_f32: panels must be 16-byte aligned");
  hipStream_t st = (hipStream_t)stream;
  const int64_t slices = n < NG_SLICES ? n : NG_SLICES;
  float* part = reinterpret_cast<float*>(ws);
  hipLaunchKernelGGL(ngcf_wgrad_kernel, dim3((unsigned)slices, (unsigned)(d2 / T), (unsigned)(d1 / T)), dim3(BLOCK), 0, st, side,
                     bi, gT, n, d1, d2, part);
  const int64_t count = 2 * d1 * d2 + 2 * d2;
  hipLaunchKernelGGL(wgrad_reduce_kernel, dim3((unsigned)((count + 63) / 64)), dim3(64 * RG), 0, st, part, slices, count, out, 0);
  IDG_HIP(hipGetLastError());
  return IDG_OK;
}

}  // extern "C"

namespace {
}  // namespace

extern "C" {

size_t idg_linear_wgrad_workspace_bytes(int64_t n, int64_t d1, int64_t d2) {
  if (n <= 0 || d1 <= 0 || d2 <= 0) return 0;
  return (size_t)n_slices(n) * (size_t)d1 * (size_t)d2 * sizeof(float);
}

int idg_linear_wgrad_f32(const float* X, int64_t ldx, const float* G, int64_t ldg, int64_t n, int64_t d1, int64_t d2,
                         float* w_grad, int accumulate, void* ws, void* stream) {
  IDG_REQUIRE(X && G && w_grad && ws, "idg_linear_wgrad_f32: NULL argument");
  IDG_REQUIRE(n > 0 && d1 > 0 && d2 > 0 && ldx >= d1 && ldg >= d2, "idg_linear_wgrad_f32: bad sizes");
  hipStream_t st = (hipStream_t)stream;
  const int64_t s = n_slices(n);
  float* part = reinterpret_cast<float*>(ws);
  hipLaunchKernelGGL(wgrad_partial_kernel, dim3((unsigned)s, (unsigned)((d2 + T - 1) / T), (unsigned)((d1 + T - 1) / T)),
                     dim3(BLOCK), 0, st, X, ldx, G, ldg, n, d1, d2, part);
  const int64_t count = d1 * d2;
  hipLaunchKernelGGL(wgrad_reduce_kernel, dim3((unsigned)((count + 63) / 64)), dim3(64 * RG), 0, st, part, s, count, w_grad,
                     accumulate);
  IDG_HIP(hipGetLastError());
  return IDG_OK;
}

}  // extern "C"

// ---------------------------------------------------------------------------------------------------------------
// NGCF's per-layer tail (models/NGCF.py:95-108), everything after the two thin GEMMs, as one pass over the rows:
//     t = (S1 + b1) + (S2 + b2);  a = leaky_relu(t, 0.2);  e = dropout(a, p) (ALWAYS on: the reference builds
//     nn.Dropout inside aggregate(), so it is in training mode during evaluation too);  nrm = normalize(e, dim=1)
// and its backward.  The dropout mask is regenerated from (seed, stream, row, feature) in the backward pass; the sign of
// t is recovered from e (where the mask kept the element; where it did not, the gradient is zero anyway), so the only
// tensor kept for backward is e itself.  One wave per row, any width.
namespace {

using idg::keep_scale;
using idg::keep_scale4;
__device__ __forceinline__ float wsum(float v) {
#pragma unroll
  for (int m = 32; m >= 1; m >>= 1) v += __shfl_xor(v, m, 64);
  return v;
}

__global__ __launch_bounds__(BLOCK) void ngcf_tail_fwd_kernel(const float* __restrict__ S1, const float* __restrict__ S2,
                                                              const float* __restrict__ b1, const float* __restrict__ b2,
                                                              int64_t n, int64_t d, float slope, float p, uint64_t seed,
                                                              uint64_t stream, float* __restrict__ E,
                                                              float* __restrict__ N, int64_t ldn) {
  const int lane = threadIdx.x % 64;
  const int64_t r = (int64_t)blockIdx.x * (BLOCK / 64) + threadIdx.x / 64;
  if (r >= n) return;
  float ss = 0.f;
  for (int64_t f = lane; f < d; f += 64) {
    const float t = (S1[r * d + f] + b1[f]) + ((S2 ? S2[r * d + f] : 0.f) + b2[f]);
    const float a = t > 0.f ? t : t * slope;
    const float e = a * keep_scale(p, seed, stream, r, f);
    E[r * d + f] = e;
    ss += e * e;
  }
  ss = wsum(ss);
  const float den = fmaxf(sqrtf(ss), 1e-12f);
  for (int64_t f = lane; f < d; f += 64) N[r * ldn + f] = E[r * d + f] / den;
}

// gT = d loss / d t given gE (through the layer's output as next ego) and gN (through its normalised copy)
__global__ __launch_bounds__(BLOCK) void ngcf_tail_bwd_kernel(const float* __restrict__ E, const float* __restrict__ gE,
                                                              const float* __restrict__ gN, int64_t n, int64_t d,
                                                              float slope, float p, uint64_t seed, uint64_t stream,
                                                              float* __restrict__ gT, int64_t ldgn,
                                                              const uint32_t* __restrict__ gn_rows) {
  const int lane = threadIdx.x % 64;
  const int64_t r = (int64_t)blockIdx.x * (BLOCK / 64) + threadIdx.x / 64;
  if (r >= n) return;
  // gn_rows: bitmap of the rows where gN holds anything (a fused step's d loss / d final is stored at the batch's rows only)
  if (gN && gn_rows && !((gn_rows[r >> 5] >> (r & 31)) & 1u)) gN = nullptr;
  if (!gN && !gE) {  // nothing flows into this row
    for (int64_t f = lane; f < d; f += 64) gT[r * d + f] = 0.f;
    return;
  }
  float ss = 0.f, dot = 0.f;
  for (int64_t f = lane; f < d; f += 64) {
    const float e = E[r * d + f];
    ss += e * e;
    if (gN) dot += gN[r * ldgn + f] * e;
  }
  ss = wsum(ss);
  dot = wsum(dot);
  const float nrm = sqrtf(ss);
  const float den = fmaxf(nrm, 1e-12f);
  for (int64_t f = lane; f < d; f += 64) {
    const float e = E[r * d + f];
    float g = gE ? gE[r * d + f] : 0.f;
    if (gN) {
      // y = e / max(||e||, eps): dy/de = (I - y y^T) / ||e|| above the clamp, I / eps below it
      const float gn = gN[r * ldgn + f];
      g += nrm > 1e-12f ? (gn - dot * e / (den * den)) / den : gn / den;
    }
    const float k = keep_scale(p, seed, stream, r, f);
    // t > 0  <=>  e > 0 wherever the element was kept (k > 0); dropped elements carry no gradient
    gT[r * d + f] = g * k * (e > 0.f ? 1.0f : (e < 0.f ? slope : (k > 0.f ? slope : 0.f)));
  }
}

// The same two kernels for d % 4 == 0, d <= 256 (every shipped width): d / 4 lanes per row, 16 bytes per lane, 64 / LPR rows
// per wave — the one-wave-per-row form above moves 256 B per memory instruction at d = 64 and ran at a third of the
// streaming rate (round 4).  Same per-element arithmetic; the row's sum of squares is formed 4 elements per lane, then
// across the row's lanes.
template <int LPR>
__device__ __forceinline__ float row_sum(float v) {
#pragma unroll
  for (int m = LPR / 2; m >= 1; m >>= 1) v += __shfl_xor(v, m, LPR);
  return v;
}

template <int LPR>
__global__ __launch_bounds__(BLOCK) void ngcf_tail_fwd_vec_kernel(const float* __restrict__ S1, const float* __restrict__ S2,
                                                                  const float* __restrict__ b1, const float* __restrict__ b2,
                                                                  int64_t n, float slope, float p, uint64_t seed, uint64_t stream,
                                                                  float* __restrict__ E, float* __restrict__ N, int64_t ldn) {
  constexpr int d = LPR * 4;
  const int l = threadIdx.x % LPR;
  const int64_t r = (int64_t)blockIdx.x * (BLOCK / LPR) + threadIdx.x / LPR;
  if (r >= n) return;  // whole lane groups leave together
  const int64_t f = 4 * l;
  const float4 s1 = *reinterpret_cast<const float4*>(S1 + r * d + f);
  const float4 s2 = S2 ? *reinterpret_cast<const float4*>(S2 + r * d + f) : make_float4(0.f, 0.f, 0.f, 0.f);
  const float4 c1 = *reinterpret_cast<const float4*>(b1 + f), c2 = *reinterpret_cast<const float4*>(b2 + f);
  const float sv[4] = {s1.x, s1.y, s1.z, s1.w}, tv[4] = {s2.x, s2.y, s2.z, s2.w};
  const float bv[4] = {c1.x, c1.y, c1.z, c1.w}, dv[4] = {c2.x, c2.y, c2.z, c2.w};
  float e[4], kp[4], ss = 0.f;
  keep_scale4(p, seed, stream, r, f, kp);
#pragma unroll
  for (int c = 0; c < 4; ++c) {
    const float t = (sv[c] + bv[c]) + (tv[c] + dv[c]);
    const float a = t > 0.f ? t : t * slope;
    e[c] = a * kp[c];
    ss += e[c] * e[c];
  }
  ss = row_sum<LPR>(ss);
  const float den = fmaxf(sqrtf(ss), 1e-12f);
  *reinterpret_cast<float4*>(E + r * d + f) = make_float4(e[0], e[1], e[2], e[3]);
  *reinterpret_cast<float4*>(N + r * ldn + f) = make_float4(e[0] / den, e[1] / den, e[2] / den, e[3] / den);
}

template <int LPR>
__global__ __launch_bounds__(BLOCK) void ngcf_tail_bwd_vec_kernel(const float* __restrict__ E, const float* __restrict__ gE,
                                                                  const float* __restrict__ gN, int64_t n, float slope, float p,
                                                                  uint64_t seed, uint64_t stream, float* __restrict__ gT,
                                                                  int64_t ldgn, const uint32_t* __restrict__ gn_rows) {
  constexpr int d = LPR * 4;
  const int l = threadIdx.x % LPR;
  const int64_t r = (int64_t)blockIdx.x * (BLOCK / LPR) + threadIdx.x / LPR;
  if (r >= n) return;
  const int64_t f = 4 * l;
  if (gN && gn_rows && !((gn_rows[r >> 5] >> (r & 31)) & 1u)) gN = nullptr;
  if (!gN && !gE) {
    *reinterpret_cast<float4*>(gT + r * d + f) = make_float4(0.f, 0.f, 0.f, 0.f);
    return;
  }
  const float4 e4 = *reinterpret_cast<const float4*>(E + r * d + f);
  const float ev[4] = {e4.x, e4.y, e4.z, e4.w};
  float gn[4] = {0.f, 0.f, 0.f, 0.f}, ge[4] = {0.f, 0.f, 0.f, 0.f};
  if (gN) {
    const float4 x = *reinterpret_cast<const float4*>(gN + r * ldgn + f);
    gn[0] = x.x, gn[1] = x.y, gn[2] = x.z, gn[3] = x.w;
  }
  if (gE) {
    const float4 x = *reinterpret_cast<const float4*>(gE + r * d + f);
    ge[0] = x.x, ge[1] = x.y, ge[2] = x.z, ge[3] = x.w;
  }
  float ss = 0.f, dot = 0.f;
#pragma unroll
  for (int c = 0; c < 4; ++c) ss += ev[c] * ev[c], dot += gn[c] * ev[c];
  ss = row_sum<LPR>(ss);
  dot = row_sum<LPR>(dot);
  const float nrm = sqrtf(ss);
  const float den = fmaxf(nrm, 1e-12f);
  float out[4], kp[4];
  keep_scale4(p, seed, stream, r, f, kp);
#pragma unroll
  for (int c = 0; c < 4; ++c) {
    float g = ge[c];
    if (gN) g += nrm > 1e-12f ? (gn[c] - dot * ev[c] / (den * den)) / den : gn[c] / den;
    const float k = kp[c];
    out[c] = g * k * (ev[c] > 0.f ? 1.0f : (ev[c] < 0.f ? slope : (k > 0.f ? slope : 0.f)));
  }
  *reinterpret_cast<float4*>(gT + r * d + f) = make_float4(out[0], out[1], out[2], out[3]);
}

}  // namespace

extern "C" {

int idg_ngcf_tail_ex_f32(const float* S1, const float* S2, const float* b1, const float* b2, int64_t n, int64_t d,
                         float negative_slope, float p, uint64_t seed, uint64_t stream_id, float* E, float* N, int64_t ldn,
                         void* stream) {
  IDG_REQUIRE(S1 && b1 && b2 && E && N, "idg_ngcf_tail_f32: NULL argument");
  IDG_REQUIRE(n >= 0 && d > 0 && ldn >= d && p >= 0.f && p < 1.f, "idg_ngcf_tail_f32: bad sizes / drop probability");
  if (n == 0) return IDG_OK;
  const bool vec = ldn % 4 == 0 && (((uintptr_t)S1 | (uintptr_t)S2 | (uintptr_t)b1 | (uintptr_t)b2 | (uintptr_t)E | (uintptr_t)N) % 16 == 0);
#define IDG_TAILF(LPR)                                                                                                       \
  hipLaunchKernelGGL((ngcf_tail_fwd_vec_kernel<LPR>), dim3((unsigned)((n + BLOCK / LPR - 1) / (BLOCK / LPR))), dim3(BLOCK), 0,   \
                     (hipStream_t)stream, S1, S2, b1, b2, n, negative_slope, p, seed, stream_id, E, N, ldn)
  if (vec && d == 64) IDG_TAILF(16);
  else if (vec && d == 32) IDG_TAILF(8);
  else if (vec && d == 128) IDG_TAILF(32);
  else if (vec && d == 256) IDG_TAILF(64);
  else
  hipLaunchKernelGGL(ngcf_tail_fwd_kernel, dim3((unsigned)((n + BLOCK / 64 - 1) / (BLOCK / 64))), dim3(BLOCK), 0,
                     (hipStream_t)stream, S1, S2, b1, b2, n, d, negative_slope, p, seed, stream_id, E, N, ldn);
#undef IDG_TAILF
  IDG_HIP(hipGetLastError());
  return IDG_OK;
}

int idg_ngcf_tail_f32(const float* S1, const float* S2, const float* b1, const float* b2, int64_t n, int64_t d,
                      float negative_slope, float p, uint64_t seed, uint64_t stream_id, float* E, float* N, void* stream) {
  return idg_ngcf_tail_ex_f32(S1, S2, b1, b2, n, d, negative_slope, p, seed, stream_id, E, N, d, stream);
}

int idg_ngcf_tail_bwd_ex_f32(const float* E, const float* gE, const float* gN, int64_t ldgn, const uint32_t* gn_rows, int64_t n,
                             int64_t d, float negative_slope, float p, uint64_t seed, uint64_t stream_id, float* gT,
                             void* stream) {
  IDG_REQUIRE(E && gT && (gE || gN), "idg_ngcf_tail_bwd_f32: NULL argument");
  IDG_REQUIRE(n >= 0 && d > 0 && (!gN || ldgn >= d) && p >= 0.f && p < 1.f, "idg_ngcf_tail_bwd_f32: bad sizes / drop probability");
  if (n == 0) return IDG_OK;
  const bool vec = (!gN || ldgn % 4 == 0) && (((uintptr_t)E | (uintptr_t)gE | (uintptr_t)gN | (uintptr_t)gT) % 16 == 0);
#define IDG_TAILB(LPR)                                                                                                       \
  hipLaunchKernelGGL((ngcf_tail_bwd_vec_kernel<LPR>), dim3((unsigned)((n + BLOCK / LPR - 1) / (BLOCK / LPR))), dim3(BLOCK), 0,   \
                     (hipStream_t)stream, E, gE, gN, n, negative_slope, p, seed, stream_id, gT, ldgn, gn_rows)
  if (vec && d == 64) IDG_TAILB(16);
  else if (vec && d == 32) IDG_TAILB(8);
  else if (vec && d == 128) IDG_TAILB(32);
  else if (vec && d == 256) IDG_TAILB(64);
  else
  hipLaunchKernelGGL(ngcf_tail_bwd_kernel, dim3((unsigned)((n + BLOCK / 64 - 1) / (BLOCK / 64))), dim3(BLOCK), 0,
                     (hipStream_t)stream, E, gE, gN, n, d, negative_slope, p, seed, stream_id, gT, ldgn, gn_rows);
#undef IDG_TAILB
  IDG_HIP(hipGetLastError());
  return IDG_OK;
}

int idg_ngcf_tail_bwd_f32(const float* E, const float* gE, const float* gN, int64_t n, int64_t d, float negative_slope,
                          float p, uint64_t seed, uint64_t stream_id, float* gT, void* stream) {
  return idg_ngcf_tail_bwd_ex_f32(E, gE, gN, d, nullptr, n, d, negative_slope, p, seed, stream_id, gT, stream);
}

}  // extern "C"

// ---------------------------------------------------------------------------------------------------------------
// Glue of NGCF's fused step (id-grec_amd/ngcf.py): column sums (the bias gradients of models/NGCF.py:91-99: gradient of
// a [1, d] row broadcast over n rows), a strided panel copy (layer 0's slot of the concatenated final rows, NGCF.py:108)
// and the gathering of three gradient contributions at the batch's rows.  Deterministic (fixed slice and tree orders).
namespace {

constexpr int CS_SLICES = 256;

// part[s][f] = sum of X[r][f] over the rows of slice s (rows in order); grid (CS_SLICES), BLOCK threads: thread t owns
// column t % d' of row group t / d' ... kept simple: each thread strides over rows for its column(s), partial sums are
// combined across the block's row groups in LDS in group order.
__global__ __launch_bounds__(BLOCK) void colsum_partial_kernel(const float* __restrict__ X, int64_t ldx, int64_t n, int64_t d,
                                                               float* __restrict__ part) {
  __shared__ float s_acc[BLOCK];
  const int64_t rows_per = (n + CS_SLICES - 1) / CS_SLICES;
  const int64_t r0 = (int64_t)blockIdx.x * rows_per, r1 = r0 + rows_per < n ? r0 + rows_per : n;
  for (int64_t f0 = 0; f0 < d; f0 += BLOCK) {
    // d <= BLOCK: groups = BLOCK / d rows in flight; d > BLOCK: one row at a time
    const int64_t w = d - f0 < BLOCK ? d - f0 : BLOCK;
    const int groups = (int)(BLOCK / w);
    const int g = (int)(threadIdx.x / w), c = (int)(threadIdx.x % w);
    float acc = 0.f;
    if (g < groups)
      for (int64_t r = r0 + g; r < r1; r += groups) acc += X[r * ldx + f0 + c];
    s_acc[threadIdx.x] = acc;
    __syncthreads();
    if (g == 0) {
      float t = acc;
      for (int q = 1; q < groups; ++q) t += s_acc[q * w + c];
      part[(int64_t)blockIdx.x * d + f0 + c] = t;
    }
    __syncthreads();
  }
}

__global__ __launch_bounds__(BLOCK) void copy_cols_kernel(float* __restrict__ dst, int64_t ldd, const float* __restrict__ src,
                                                          int64_t lds, int64_t n, int64_t d4) {
  const int64_t i = (int64_t)blockIdx.x * BLOCK + threadIdx.x;
  if (i >= n * d4) return;
  const int64_t r = i / d4, c = i % d4;
  reinterpret_cast<float4*>(dst + r * ldd)[c] = reinterpret_cast<const float4*>(src + r * lds)[c];
}

// dst[r] = dst[r] + a[r] (+ b[r]) at the rows flagged in `rows`
__global__ __launch_bounds__(BLOCK) void rows_add2_kernel(float* __restrict__ dst, int64_t ldd, const float* __restrict__ a,
                                                          int64_t lda, const float* __restrict__ b, int64_t ldb,
                                                          const uint32_t* __restrict__ rows, int64_t n, int64_t d4) {
  const int64_t i = (int64_t)blockIdx.x * BLOCK + threadIdx.x;
  if (i >= n * d4) return;
  const int64_t r = i / d4, c = i % d4;
  if (!((rows[r >> 5] >> (r & 31)) & 1u)) return;
  float4 x = reinterpret_cast<float4*>(dst + r * ldd)[c];
  const float4 y = reinterpret_cast<const float4*>(a + r * lda)[c];
  x.x += y.x, x.y += y.y, x.z += y.z, x.w += y.w;
  if (b) {
    const float4 z = reinterpret_cast<const float4*>(b + r * ldb)[c];
    x.x += z.x, x.y += z.y, x.z += z.z, x.w += z.w;
  }
  reinterpret_cast<float4*>(dst + r * ldd)[c] = x;
}

}  // namespace

extern "C" {

size_t idg_colsum_workspace_bytes(int64_t d) { return d > 0 ? (size_t)CS_SLICES * (size_t)d * sizeof(float) : 0; }

int idg_colsum_f32(const float* X, int64_t ldx, int64_t n, int64_t d, float* out, int accumulate, void* ws, void* stream) {
  IDG_REQUIRE(X && out && ws && n >= 0 && d > 0 && ldx >= d, "idg_colsum_f32: bad argument");
  hipStream_t st = (hipStream_t)stream;
  float* part = reinterpret_cast<float*>(ws);
  hipLaunchKernelGGL(colsum_partial_kernel, dim3(CS_SLICES), dim3(BLOCK), 0, st, X, ldx, n, d, part);
  // the slices are added in slice order, 16 lane groups per 64 columns (wgrad_reduce_kernel's fixed tree)
  hipLaunchKernelGGL(wgrad_reduce_kernel, dim3((unsigned)((d + 63) / 64)), dim3(64 * RG), 0, st, part, (int64_t)CS_SLICES, d, out,
                     accumulate ? 1 : 0);
  IDG_HIP(hipGetLastError());
  return IDG_OK;
}

int idg_copy_cols_f32(float* dst, int64_t ldd, const float* src, int64_t lds, int64_t n, int64_t d, void* stream) {
  IDG_REQUIRE(dst && src && n >= 0 && d > 0 && d % 4 == 0 && ldd >= d && lds >= d && ldd % 4 == 0 && lds % 4 == 0,
              "idg_copy_cols_f32: bad argument (widths and leading dimensions multiples of 4)");
  IDG_REQUIRE(((uintptr_t)dst | (uintptr_t)src) % 16 == 0, "idg_copy_cols_f32: panels must be 16-byte aligned");
  if (n == 0) return IDG_OK;
  const int64_t total = n * (d / 4);
  hipLaunchKernelGGL(copy_cols_kernel, dim3((unsigned)((total + BLOCK - 1) / BLOCK)), dim3(BLOCK), 0, (hipStream_t)stream, dst, ldd,
                     src, lds, n, d / 4);
  IDG_HIP(hipGetLastError());
  return IDG_OK;
}

int idg_rows_add2_f32(float* dst, int64_t ldd, const float* a, int64_t lda, const float* b, int64_t ldb, const uint32_t* rows,
                      int64_t n, int64_t d, void* stream) {
  IDG_REQUIRE(dst && a && rows && n >= 0 && d > 0 && d % 4 == 0 && ldd % 4 == 0 && lda % 4 == 0 && (!b || ldb % 4 == 0),
              "idg_rows_add2_f32: bad argument");
  IDG_REQUIRE(((uintptr_t)dst | (uintptr_t)a | (uintptr_t)b) % 16 == 0, "idg_rows_add2_f32: panels must be 16-byte aligned");
  if (n == 0) return IDG_OK;
  const int64_t total = n * (d / 4);
  hipLaunchKernelGGL(rows_add2_kernel, dim3((unsigned)((total + BLOCK - 1) / BLOCK)), dim3(BLOCK), 0, (hipStream_t)stream, dst, ldd,
                     a, lda, b, ldb, rows, n, d / 4);
  IDG_HIP(hipGetLastError());
  return IDG_OK;
}

}  // extern "C"

// ---------------------------------------------------------------------------------------------------------------
// NGCF's two thin transforms of a layer (models/NGCF.py:88-99) on the fp32 matrix cores, as ONE pass over the rows:
//     S = side . W1 + (ego * side) . W2            (side = A.ego from the SpMM, W1 = W_gcn, W2 = W_bi, [d1, d2])
// and its input gradients
//     gSide = gS . W1^T + (gS . W2^T) * ego,   gEgo = (gS . W2^T) * side.
// v_mfma_f32_32x32x2_f32: a wave owns a 32-row x 32-column output tile; lane (i, h) feeds row i's features
// [kc + 32h, kc + 32h + 32) as the A operand (one contiguous 128-byte run per 64-deep chunk, nothing transposed or
// staged in LDS), and the matching 32 weights of column i as the B operand.  Both products of the forward share one
// accumulator, both products of the backward share one A operand.  Exact fp32 products and accumulation.
namespace {

using f32x16 = __attribute__((ext_vector_type(16))) float;

// Persistent waves: wave w of the launch owns output column tile w % nct (forward) / input feature tile w % nkt
// (backward) and keeps that tile's weights — 32 of W1 and 32 of W2 per lane — in registers while it walks the row tiles
// w / nct, w / nct + stride, ...: per row tile it loads only the row operands (one contiguous 128-byte run per lane).
constexpr int NGCF_WGS = 256 * 3;  // workgroups per launch (4 waves each; ~110 VGPRs: 4 waves per SIMD fit)

__global__ __launch_bounds__(BLOCK) void ngcf_transform_fwd_kernel(const float* __restrict__ side, const float* __restrict__ ego,
                                                                   const float* __restrict__ W1, const float* __restrict__ W2,
                                                                   int64_t n, int64_t d1, int64_t d2, float* __restrict__ S,
                                                                   float* __restrict__ BI) {
  const int lane = threadIdx.x % 64, i = lane & 31, h = lane >> 5;
  const int64_t nct = d2 / 32, n_rt = (n + 31) / 32;
  const int64_t wave = (int64_t)blockIdx.x * (BLOCK / 64) + threadIdx.x / 64;
  const int64_t n_waves = (int64_t)gridDim.x * (BLOCK / 64);
  const int64_t ct = wave % nct, c0 = ct * 32;
  const int64_t rt_stride = n_waves / nct;  // (n_waves is a multiple of nct: see the launch)
  for (int64_t kc = 0; kc < d1; kc += 64) {  // d1 = 64: one pass, weights loaded once; wider layers re-walk the rows per chunk
    const int64_t k0 = kc + 32 * h;
    float w1[32], w2[32];
#pragma unroll
    for (int s = 0; s < 32; ++s) w1[s] = W1[(k0 + s) * d2 + c0 + i], w2[s] = W2[(k0 + s) * d2 + c0 + i];
    for (int64_t rt = wave / nct; rt < n_rt; rt += rt_stride) {
      const int64_t r0 = rt * 32;
      const int64_t row = r0 + i < n ? r0 + i : n - 1;
      float a[32], b[32];
#pragma unroll
      for (int q = 0; q < 8; ++q) {
        const float4 x = *reinterpret_cast<const float4*>(side + row * d1 + k0 + 4 * q);
        const float4 y = *reinterpret_cast<const float4*>(ego + row * d1 + k0 + 4 * q);
        a[4 * q + 0] = x.x, a[4 * q + 1] = x.y, a[4 * q + 2] = x.z, a[4 * q + 3] = x.w;
        b[4 * q + 0] = x.x * y.x, b[4 * q + 1] = x.y * y.y, b[4 * q + 2] = x.z * y.z, b[4 * q + 3] = x.w * y.w;
      }
      if (BI && ct == 0 && r0 + i < n) {
#pragma unroll
        for (int q = 0; q < 8; ++q)
          *reinterpret_cast<float4*>(BI + row * d1 + k0 + 4 * q) = make_float4(b[4 * q], b[4 * q + 1], b[4 * q + 2], b[4 * q + 3]);
      }
      f32x16 acc;
#pragma unroll
      for (int r = 0; r < 16; ++r) acc[r] = 0.f;
#pragma unroll
      for (int s = 0; s < 32; ++s) {
        acc = __builtin_amdgcn_mfma_f32_32x32x2f32(a[s], w1[s], acc, 0, 0, 0);
        acc = __builtin_amdgcn_mfma_f32_32x32x2f32(b[s], w2[s], acc, 0, 0, 0);
      }
      // C/D map: column = lane & 31, row = (r & 3) + 8 * (r >> 2) + 4 * (lane >> 5)
#pragma unroll
      for (int r = 0; r < 16; ++r) {
        const int64_t rr = r0 + (r & 3) + 8 * (r >> 2) + 4 * h;
        if (rr < n) {
          float* o = S + rr * d2 + c0 + i;
          *o = kc == 0 ? acc[r] : *o + acc[r];  // further 64-deep chunks of a wide layer add to the first one's sums
        }
      }
    }
  }
}

// A1 = gS . W1^T, A2 = gS . W2^T over d2; gSide = A1 + A2 * ego, gEgo = A2 * side
__global__ __launch_bounds__(BLOCK) void ngcf_transform_bwd_kernel(const float* __restrict__ gS, const float* __restrict__ side,
                                                                   const float* __restrict__ ego, const float* __restrict__ W1,
                                                                   const float* __restrict__ W2, int64_t n, int64_t d1,
                                                                   int64_t d2, float* __restrict__ gSide,
                                                                   float* __restrict__ gEgo) {
  const int lane = threadIdx.x % 64, i = lane & 31, h = lane >> 5;
  const int64_t nkt = d1 / 32, n_rt = (n + 31) / 32;
  const int64_t wave = (int64_t)blockIdx.x * (BLOCK / 64) + threadIdx.x / 64;
  const int64_t n_waves = (int64_t)gridDim.x * (BLOCK / 64);
  const int64_t k1 = (wave % nkt) * 32;  // this wave's input features; lane i supplies the weights of feature k1 + i
  const int64_t rt_stride = n_waves / nkt;
  // d2 = 64 (every shipped configuration): this wave's 2 x 32 weights per lane are the same for every row tile it walks —
  // loaded ONCE into registers, as the forward kernel does (round 4: they were re-read per tile, two thirds of the
  // kernel's load instructions)
  const bool keep = d2 == 64;
  float kw1[32], kw2[32];
  if (keep) {
#pragma unroll
    for (int q = 0; q < 8; ++q) {
      const float4 u = *reinterpret_cast<const float4*>(W1 + (k1 + i) * d2 + 32 * h + 4 * q);
      const float4 v = *reinterpret_cast<const float4*>(W2 + (k1 + i) * d2 + 32 * h + 4 * q);
      kw1[4 * q + 0] = u.x, kw1[4 * q + 1] = u.y, kw1[4 * q + 2] = u.z, kw1[4 * q + 3] = u.w;
      kw2[4 * q + 0] = v.x, kw2[4 * q + 1] = v.y, kw2[4 * q + 2] = v.z, kw2[4 * q + 3] = v.w;
    }
  }
  for (int64_t rt = wave / nkt; rt < n_rt; rt += rt_stride) {
    const int64_t r0 = rt * 32;
    const int64_t row = r0 + i < n ? r0 + i : n - 1;
    f32x16 acc1, acc2;
#pragma unroll
    for (int r = 0; r < 16; ++r) acc1[r] = 0.f, acc2[r] = 0.f;
    if (keep) {
      float a[32];
#pragma unroll
      for (int q = 0; q < 8; ++q) {
        const float4 x = *reinterpret_cast<const float4*>(gS + row * d2 + 32 * h + 4 * q);
        a[4 * q + 0] = x.x, a[4 * q + 1] = x.y, a[4 * q + 2] = x.z, a[4 * q + 3] = x.w;
      }
#pragma unroll
      for (int s = 0; s < 32; ++s) {
        acc1 = __builtin_amdgcn_mfma_f32_32x32x2f32(a[s], kw1[s], acc1, 0, 0, 0);
        acc2 = __builtin_amdgcn_mfma_f32_32x32x2f32(a[s], kw2[s], acc2, 0, 0, 0);
      }
    } else
    for (int64_t cc = 0; cc < d2; cc += 64) {  // (the weight runs are L1 / L2 hits after the first tile)
      const int64_t c0 = cc + 32 * h;
      float a[32], w1[32], w2[32];
#pragma unroll
      for (int q = 0; q < 8; ++q) {
        const float4 x = *reinterpret_cast<const float4*>(gS + row * d2 + c0 + 4 * q);
        const float4 u = *reinterpret_cast<const float4*>(W1 + (k1 + i) * d2 + c0 + 4 * q);
        const float4 v = *reinterpret_cast<const float4*>(W2 + (k1 + i) * d2 + c0 + 4 * q);
        a[4 * q + 0] = x.x, a[4 * q + 1] = x.y, a[4 * q + 2] = x.z, a[4 * q + 3] = x.w;
        w1[4 * q + 0] = u.x, w1[4 * q + 1] = u.y, w1[4 * q + 2] = u.z, w1[4 * q + 3] = u.w;
        w2[4 * q + 0] = v.x, w2[4 * q + 1] = v.y, w2[4 * q + 2] = v.z, w2[4 * q + 3] = v.w;
      }
#pragma unroll
      for (int s = 0; s < 32; ++s) {
        acc1 = __builtin_amdgcn_mfma_f32_32x32x2f32(a[s], w1[s], acc1, 0, 0, 0);
        acc2 = __builtin_amdgcn_mfma_f32_32x32x2f32(a[s], w2[s], acc2, 0, 0, 0);
      }
    }
#pragma unroll
    for (int r = 0; r < 16; ++r) {
      const int64_t rr = r0 + (r & 3) + 8 * (r >> 2) + 4 * h;
      if (rr < n) {
        const int64_t o = rr * d1 + k1 + i;
        const float e = ego[o], sd = side[o];
        gSide[o] = __builtin_fmaf(acc2[r], e, acc1[r]);
        gEgo[o] = acc2[r] * sd;
      }
    }
  }
}

}  // namespace

extern "C" {

int idg_ngcf_transform_f32(const float* side, const float* ego, const float* W1, const float* W2, int64_t n, int64_t d1,
                           int64_t d2, float* S, float* BI, void* stream) {
  IDG_REQUIRE(side && ego && W1 && W2 && S, "idg_ngcf_transform_f32: NULL argument");
  IDG_REQUIRE(n >= 0 && d1 > 0 && d2 > 0 && d1 % 64 == 0 && d2 % 32 == 0,
              "idg_ngcf_transform_f32: needs d1 %% 64 == 0 and d2 %% 32 == 0 (got %lld, %lld)", (long long)d1, (long long)d2);
  IDG_REQUIRE(((uintptr_t)side | (uintptr_t)ego | (uintptr_t)BI) % 16 == 0, "idg_ngcf_transform_f32: panels must be 16-byte aligned");
  if (n == 0) return IDG_OK;
  // persistent waves: a multiple of the column-tile count, no more than the work needs
  const int64_t nct = d2 / 32, tasks = (n + 31) / 32 * nct;
  int64_t waves = std::min<int64_t>((int64_t)NGCF_WGS * (BLOCK / 64), tasks);
  waves = std::max<int64_t>(nct * (BLOCK / 64), waves / (nct * (BLOCK / 64)) * (nct * (BLOCK / 64)));
  hipLaunchKernelGGL(ngcf_transform_fwd_kernel, dim3((unsigned)(waves / (BLOCK / 64))), dim3(BLOCK), 0, (hipStream_t)stream, side, ego,
                     W1, W2, n, d1, d2, S, BI);
  IDG_HIP(hipGetLastError());
  return IDG_OK;
}

int idg_ngcf_transform_bwd_f32(const float* gS, const float* side, const float* ego, const float* W1, const float* W2, int64_t n,
                               int64_t d1, int64_t d2, float* g_side, float* g_ego, void* stream) {
  IDG_REQUIRE(gS && side && ego && W1 && W2 && g_side && g_ego, "idg_ngcf_transform_bwd_f32: NULL argument");
  IDG_REQUIRE(n >= 0 && d1 > 0 && d2 > 0 && d2 % 64 == 0 && d1 % 32 == 0,
              "idg_ngcf_transform_bwd_f32: needs d2 %% 64 == 0 and d1 %% 32 == 0 (got %lld, %lld)", (long long)d2, (long long)d1);
  IDG_REQUIRE(((uintptr_t)gS | (uintptr_t)W1 | (uintptr_t)W2) % 16 == 0, "idg_ngcf_transform_bwd_f32: panels must be 16-byte aligned");
  if (n == 0) return IDG_OK;
  const int64_t nkt = d1 / 32, tasks = (n + 31) / 32 * nkt;
  int64_t waves = std::min<int64_t>((int64_t)NGCF_WGS * (BLOCK / 64), tasks);
  waves = std::max<int64_t>(nkt * (BLOCK / 64), waves / (nkt * (BLOCK / 64)) * (nkt * (BLOCK / 64)));
  hipLaunchKernelGGL(ngcf_transform_bwd_kernel, dim3((unsigned)(waves / (BLOCK / 64))), dim3(BLOCK), 0, (hipStream_t)stream, gS, side,
                     ego, W1, W2, n, d1, d2, g_side, g_ego);
  IDG_HIP(hipGetLastError());
  return IDG_OK;
}

}  // extern "C"
