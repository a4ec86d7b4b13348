// One NGCF layer (models/NGCF.py:88-108) as ONE kernel per direction, d = 64 (round 4).
//
// The chain of round 3/4 — transform (fp32 MFMA) -> tail, and backwards tail -> parameter gradients -> transform — is five
// launches per layer that each stream the [n, 64] panels again: 47.7 us forward and 87 us backward per layer at yelp2018 size
// (n = 69,716), a third of that spent on the address path (the MFMA operand layout reads one ROW per lane: 64 cache lines
// per load instruction) and on launches too short to overlap their own phases.  Here a 256-thread workgroup owns 64 rows:
// their panels are read ONCE, coalesced (16 lanes x 16 bytes per row), into padded LDS tiles; the MFMA operands come out of
// LDS (ds_read_b128, conflict-free at a row stride of 68 floats); everything between the products — bias, LeakyReLU,
// dropout, the row norm, their derivatives — happens on the tile; results leave coalesced.
//   forward : (side, ego) -> E = dropout(leaky(side.W1 + (ego*side).W2 + b1 + b2)), N = E / max(||E||, eps)   [S, BI never stored]
//   backward: (E, gE, gN, side, ego) -> g_side, g_ego, and the layer's four parameter gradients                [gT never stored]
// Same MFMA sequences and the same element arithmetic as idg_ngcf_transform_f32 / idg_ngcf_tail_ex_f32 / their backward
// forms (idg_dense.hip): E, N, g_side and g_ego are bit-identical to the chain's; the parameter gradients are sums over the
// rows in another (fixed) order — slices of rows on persistent workgroups, added in slice order.
#include <hip/hip_runtime.h>

#include <algorithm>
#include <cstdint>

#include "idg_common.h"
#include "idg_dropout.h"

namespace {

using f32x16 = __attribute__((ext_vector_type(16))) float;

constexpr int BLOCK = 256;
constexpr int D = 64;
constexpr int RB = 64;   // rows per workgroup tile
constexpr int LDT = 68;  // LDS row stride in floats: rows 16-byte aligned, b128 operand reads of 16 rows hit 64 distinct banks
constexpr int BWD_WGS = 512;  // resident workgroups of the backward kernel (2 x 512 threads per CU) = most slices of its parameter-gradient sums
constexpr int RG = 16;

template <int LPR>
__device__ __forceinline__ float row_sum(float v) {
#pragma unroll
  for (int m = LPR / 2; m >= 1; m >>= 1) v += __shfl_xor(v, m, LPR);
  return v;
}

// C/D map of v_mfma_f32_32x32x2_f32: register r of lane (i, h) holds row (r & 3) + 8 (r >> 2) + 4 h, column i
__device__ __forceinline__ int c_row(int r, int h) { return (r & 3) + 8 * (r >> 2) + 4 * h; }

__global__ __launch_bounds__(BLOCK, 4) void ngcf_layer_fwd64_kernel(const float* __restrict__ side, const float* __restrict__ ego,
                                                                 const float* __restrict__ W1, const float* __restrict__ W2,
                                                                 const float* __restrict__ b1, const float* __restrict__ b2,
                                                                 int64_t n, float slope, float p, uint64_t seed, uint64_t stream,
                                                                 float* __restrict__ E, float* __restrict__ N, int64_t ldn) {
  __shared__ __attribute__((aligned(16))) float s_side[RB * LDT];
  __shared__ __attribute__((aligned(16))) float s_ego[RB * LDT];
  const int tid = threadIdx.x, lane = tid & 63, w = tid >> 6, i = lane & 31, h = lane >> 5;
  const int rt = w >> 1, ct = w & 1;  // this wave's 32 x 32 tile of the 64 x 64 block
  const int64_t r0 = (int64_t)blockIdx.x * RB;
  // the wave's 2 x 32 weights per lane (column 32 ct + i, K-values 32 h ...): issued first, in flight across the staging
  float w1[32], w2[32];
#pragma unroll
  for (int s = 0; s < 32; ++s) w1[s] = W1[(32 * h + s) * D + 32 * ct + i], w2[s] = W2[(32 * h + s) * D + 32 * ct + i];
#pragma unroll
  for (int j = 0; j < 4; ++j) {
    const int e = tid + BLOCK * j, rr = e >> 4, c4 = (e & 15) * 4;
    const int64_t row = r0 + rr < n ? r0 + rr : n - 1;
    *reinterpret_cast<float4*>(s_side + rr * LDT + c4) = *reinterpret_cast<const float4*>(side + row * D + c4);
    *reinterpret_cast<float4*>(s_ego + rr * LDT + c4) = *reinterpret_cast<const float4*>(ego + row * D + c4);
  }
  __syncthreads();
  f32x16 acc;
#pragma unroll
  for (int r = 0; r < 16; ++r) acc[r] = 0.f;
  {
    const float* ps = s_side + (32 * rt + i) * LDT + 32 * h;
    const float* pe = s_ego + (32 * rt + i) * LDT + 32 * h;
#pragma unroll
    for (int q = 0; q < 8; ++q) {  // (K-values in the order idg_ngcf_transform_f32 adds them: s = 4 q + c, W1's product then W2's)
      const float4 x = *reinterpret_cast<const float4*>(ps + 4 * q);
      const float4 y = *reinterpret_cast<const float4*>(pe + 4 * q);
      const float xs[4] = {x.x, x.y, x.z, x.w}, ys[4] = {y.x, y.y, y.z, y.w};
#pragma unroll
      for (int c = 0; c < 4; ++c) {
        acc = __builtin_amdgcn_mfma_f32_32x32x2f32(xs[c], w1[4 * q + c], acc, 0, 0, 0);
        acc = __builtin_amdgcn_mfma_f32_32x32x2f32(xs[c] * ys[c], w2[4 * q + c], acc, 0, 0, 0);
      }
    }
  }
  __syncthreads();  // every operand has left the tiles: s_side becomes the tile of E
  {
    const int col = 32 * ct + i;
    const float bb1 = b1[col], bb2 = b2[col];
#pragma unroll
    for (int r = 0; r < 16; ++r) {
      const int rr = 32 * rt + c_row(r, h);
      const float t = (acc[r] + bb1) + (0.f + bb2);  // idg_ngcf_tail_ex_f32 with S2 = NULL
      s_side[rr * LDT + col] = t > 0.f ? t : t * slope;
    }
  }
  __syncthreads();
#pragma unroll
  for (int j = 0; j < 4; ++j) {
    const int e = tid + BLOCK * j, rr = e >> 4, c4 = (e & 15) * 4;
    float4 v = *reinterpret_cast<const float4*>(s_side + rr * LDT + c4);
    float kp[4];
    idg::keep_scale4(p, seed, stream, r0 + rr, c4, kp);  // (one mix per lane here; per element in the accumulator layout)
    v.x *= kp[0], v.y *= kp[1], v.z *= kp[2], v.w *= kp[3];
    float ss = 0.f;
    ss += v.x * v.x, ss += v.y * v.y, ss += v.z * v.z, ss += v.w * v.w;
    ss = row_sum<16>(ss);
    const float den = fmaxf(sqrtf(ss), 1e-12f);
    if (r0 + rr < n) {
      *reinterpret_cast<float4*>(E + (r0 + rr) * D + c4) = v;
      *reinterpret_cast<float4*>(N + (r0 + rr) * ldn + c4) = make_float4(v.x / den, v.y / den, v.z / den, v.w / den);
    }
  }
}

// Persistent 512-thread workgroups: block b, b + grid, ... of 64 rows.  All eight waves stage the tiles; then waves 0-3 form
// the input gradients (one 32 x 32 tile of the 64 rows x 64 input features each) while waves 4-7 add the block's share of
// the parameter gradients to accumulators that live in their registers across the blocks — the two MFMA phases of a block
// run side by side on the four SIMDs, and neither half of the workgroup carries the other's registers.
constexpr int BWD_BLOCK = 512;

__global__ __launch_bounds__(BWD_BLOCK, 4) void ngcf_layer_bwd64_kernel(
    const float* __restrict__ E, const float* __restrict__ gE, const float* __restrict__ gN, int64_t ldgn,
    const uint32_t* __restrict__ gn_rows, const float* __restrict__ side, const float* __restrict__ ego,
    const float* __restrict__ W1, const float* __restrict__ W2, int64_t n, float slope, float p, uint64_t seed, uint64_t stream,
    float* __restrict__ gSide, float* __restrict__ gEgo, float* __restrict__ part) {
  __shared__ __attribute__((aligned(16))) float s_g[RB * LDT];
  __shared__ __attribute__((aligned(16))) float s_side[RB * LDT];
  __shared__ __attribute__((aligned(16))) float s_ego[RB * LDT];
  const int tid = threadIdx.x, lane = tid & 63, i = lane & 31, h = lane >> 5;
  const int w8 = __builtin_amdgcn_readfirstlane(tid >> 6);  // scalar: the role branches below are wave-uniform
  const bool input_role = w8 < 4;
  const int w = w8 & 3;
  const int rt = w >> 1, kt = w & 1;  // input gradients: rows 32 rt ..., input features 32 kt ...
  const int mt = w >> 1, nt = w & 1;  // parameter gradients: features 32 mt ... x output columns 32 nt ...
  const int k1 = 32 * kt;
  // One set of 64 registers, two uses (the compiler keeps two variables that are live across the block loop apart even
  // when no wave uses both): input waves hold their W^T operand in it — R[0..1] = the 32 weights of W1 for input feature
  // k1 + i, output columns 32 h ..., R[2..3] those of W2 — parameter waves their two accumulators, R[0] and R[1].
  f32x16 R[4];
  float cs = 0.f;
  if (input_role) {
#pragma unroll
    for (int q = 0; q < 8; ++q) {
      const float4 u = *reinterpret_cast<const float4*>(W1 + (k1 + i) * D + 32 * h + 4 * q);
      const float4 v = *reinterpret_cast<const float4*>(W2 + (k1 + i) * D + 32 * h + 4 * q);
      R[q >> 2][4 * (q & 3) + 0] = u.x, R[q >> 2][4 * (q & 3) + 1] = u.y, R[q >> 2][4 * (q & 3) + 2] = u.z, R[q >> 2][4 * (q & 3) + 3] = u.w;
      R[2 + (q >> 2)][4 * (q & 3) + 0] = v.x, R[2 + (q >> 2)][4 * (q & 3) + 1] = v.y, R[2 + (q >> 2)][4 * (q & 3) + 2] = v.z,
                   R[2 + (q >> 2)][4 * (q & 3) + 3] = v.w;
    }
  } else {
#pragma unroll
    for (int r = 0; r < 16; ++r) R[0][r] = 0.f, R[1][r] = 0.f, R[2][r] = 0.f, R[3][r] = 0.f;
  }
  const int64_t n_blocks = (n + RB - 1) / RB;
  for (int64_t blk = blockIdx.x; blk < n_blocks; blk += gridDim.x) {
    const int64_t r0 = blk * RB;
    // lane ids re-defined per block behind an empty asm: left to itself the compiler hoists every lane-dependent offset of
    // the block loop (LDS and global, ~50 registers' worth) out of it and then spills them around the loop
    int tid_l = tid, i_l = i, h_l = h;
    asm volatile("" : "+v"(tid_l), "+v"(i_l), "+v"(h_l));
    // ---- stage side / ego, and gT = d loss / d t from (E, gE, gN) exactly as idg_ngcf_tail_bwd_ex_f32 forms it
    // (addresses as a block base in scalar registers + a 32-bit lane offset: sixteen 64-bit lane addresses per phase were
    //  what pushed the kernel over its 128 registers)
    const float* side_b = side + r0 * D;
    const float* ego_b = ego + r0 * D;
    const float* E_b = E + r0 * D;
    const float* gE_b = gE ? gE + r0 * D : nullptr;
    const float* gN_b = gN ? gN + r0 * ldgn : nullptr;
    const int last = (int)(n - 1 - r0);  // rows past the end read the last row (their gT is 0)
#pragma unroll
    for (int j = 0; j < 2; ++j) {
      const int e = tid_l + BWD_BLOCK * j, rr = e >> 4, c4 = (e & 15) * 4;
      const bool live = rr <= last;
      const int rc = live ? rr : last;
      const int64_t row = r0 + rc;
      *reinterpret_cast<float4*>(s_side + rr * LDT + c4) = *reinterpret_cast<const float4*>(side_b + (rc * D + c4));
      *reinterpret_cast<float4*>(s_ego + rr * LDT + c4) = *reinterpret_cast<const float4*>(ego_b + (rc * D + c4));
      const bool has_n = gN && (!gn_rows || ((gn_rows[row >> 5] >> (row & 31)) & 1u));
      float out[4] = {0.f, 0.f, 0.f, 0.f};
      if (live && (has_n || gE)) {  // (the 16 lanes of a row decide alike)
        const float4 e4 = *reinterpret_cast<const float4*>(E_b + (rc * D + c4));
        const float ev[4] = {e4.x, e4.y, e4.z, e4.w};
        float gn[4] = {0.f, 0.f, 0.f, 0.f}, ge[4] = {0.f, 0.f, 0.f, 0.f};
        if (has_n) {
          const float4 x = *reinterpret_cast<const float4*>(gN_b + (rc * (int)ldgn + c4));
          gn[0] = x.x, gn[1] = x.y, gn[2] = x.z, gn[3] = x.w;
        }
        if (gE) {
          const float4 x = *reinterpret_cast<const float4*>(gE_b + (rc * D + c4));
          ge[0] = x.x, ge[1] = x.y, ge[2] = x.z, ge[3] = x.w;
        }
        float ss = 0.f, dot = 0.f;
#pragma unroll
        for (int c = 0; c < 4; ++c) ss += ev[c] * ev[c], dot += gn[c] * ev[c];
        ss = row_sum<16>(ss);
        dot = row_sum<16>(dot);
        const float nrm = sqrtf(ss);
        const float den = fmaxf(nrm, 1e-12f);
        float kp[4];
        idg::keep_scale4(p, seed, stream, row, c4, kp);
#pragma unroll
        for (int c = 0; c < 4; ++c) {
          float g = ge[c];
          if (has_n) g += nrm > 1e-12f ? (gn[c] - dot * ev[c] / (den * den)) / den : gn[c] / den;
          const float k = kp[c];
          out[c] = g * k * (ev[c] > 0.f ? 1.0f : (ev[c] < 0.f ? slope : (k > 0.f ? slope : 0.f)));
        }
      }
      *reinterpret_cast<float4*>(s_g + rr * LDT + c4) = make_float4(out[0], out[1], out[2], out[3]);
    }
    __syncthreads();
    if (input_role) {
      // ---- input gradients: A1 = gT . W1^T, A2 = gT . W2^T; g_side = A1 + A2 * ego, g_ego = A2 * side
      const float* pg = s_g + (32 * rt + i_l) * LDT + 32 * h_l;
      f32x16 acc1, acc2;
#pragma unroll
      for (int r = 0; r < 16; ++r) acc1[r] = 0.f, acc2[r] = 0.f;
#pragma unroll
      for (int q = 0; q < 8; ++q) {  // (the order idg_ngcf_transform_bwd_f32 adds in)
        const float4 x = *reinterpret_cast<const float4*>(pg + 4 * q);
        const float xs[4] = {x.x, x.y, x.z, x.w};
#pragma unroll
        for (int c = 0; c < 4; ++c) {
          acc1 = __builtin_amdgcn_mfma_f32_32x32x2f32(xs[c], R[q >> 2][4 * (q & 3) + c], acc1, 0, 0, 0);
          acc2 = __builtin_amdgcn_mfma_f32_32x32x2f32(xs[c], R[2 + (q >> 2)][4 * (q & 3) + c], acc2, 0, 0, 0);
        }
      }
      float* gs_b = gSide + r0 * D;
      float* ge_b = gEgo + r0 * D;
#pragma unroll
      for (int r = 0; r < 16; ++r) {
        const int rr = 32 * rt + c_row(r, h_l);
        if (rr <= last) {
          const float e = s_ego[rr * LDT + k1 + i_l], sd = s_side[rr * LDT + k1 + i_l];
          const int o = rr * D + k1 + i_l;
          gs_b[o] = __builtin_fmaf(acc2[r], e, acc1[r]);
          ge_b[o] = acc2[r] * sd;
        }
      }
    } else {
      // ---- parameter gradients: g W1 += side^T gT, g W2 += (side * ego)^T gT, bias gradients += column sums of gT.  K
      // runs over the tile's rows: step s takes rows 2 s + h (rows past the end carry gT = 0)
#pragma unroll 8
      for (int s = 0; s < 32; ++s) {
        const int rr = 2 * s + h_l;
        const float sv = s_side[rr * LDT + 32 * mt + i_l], ev = s_ego[rr * LDT + 32 * mt + i_l], gv = s_g[rr * LDT + 32 * nt + i_l];
        R[0] = __builtin_amdgcn_mfma_f32_32x32x2f32(sv, gv, R[0], 0, 0, 0);
        R[1] = __builtin_amdgcn_mfma_f32_32x32x2f32(sv * ev, gv, R[1], 0, 0, 0);
        cs += gv;
      }
    }
    __syncthreads();  // the tiles are free for the next block
  }
  if (input_role) return;
  // this workgroup's slice of the sums, in the layout [g W1 | g b1 | g W2 | g b2]
  constexpr int64_t ww = D * D;
  float* out = part + (int64_t)blockIdx.x * (2 * ww + 2 * D);
#pragma unroll
  for (int r = 0; r < 16; ++r) {
    const int m = 32 * mt + c_row(r, h), col = 32 * nt + i;
    out[m * D + col] = R[0][r];
    out[ww + D + m * D + col] = R[1][r];
  }
  cs += __shfl_xor(cs, 32, 64);
  if (mt == 0 && h == 0) {
    out[ww + 32 * nt + i] = cs;
    out[2 * ww + D + 32 * nt + i] = cs;
  }
}

// out[e] = sum over the slices, in slice order (RG partial sums of interleaved slices, then those in order)
__global__ __launch_bounds__(64 * RG) void slices_reduce_kernel(const float* __restrict__ part, int64_t slices, int64_t count,
                                                                float* __restrict__ out) {
  __shared__ float s_sum[RG][64];
  const int lane = threadIdx.x % 64, g = threadIdx.x / 64;
  const int64_t e = (int64_t)blockIdx.x * 64 + lane;
  float t = 0.f;
  if (e < count) {
    for (int64_t s0 = g; s0 < slices; s0 += 8 * RG) {  // eight independent loads in flight, added in slice order
      float v[8];
#pragma unroll
      for (int q = 0; q < 8; ++q) v[q] = s0 + q * RG < slices ? part[(s0 + q * RG) * count + e] : 0.f;
#pragma unroll
      for (int q = 0; q < 8; ++q) t += v[q];
    }
  }
  s_sum[g][lane] = t;
  __syncthreads();
  if (g == 0 && e < count) {
    t = s_sum[0][lane];
    for (int q = 1; q < RG; ++q) t += s_sum[q][lane];
    out[e] = t;
  }
}

}  // namespace

extern "C" {

int idg_ngcf_layer_fwd_f32(const float* side, const float* ego, const float* W1, const float* W2, const float* b1, const float* b2,
                           int64_t n, int64_t d, float negative_slope, float p, uint64_t seed, uint64_t stream_id, float* E,
                           float* N, int64_t ldn, void* stream) {
  IDG_REQUIRE(side && ego && W1 && W2 && b1 && b2 && E && N, "idg_ngcf_layer_fwd_f32: NULL argument");
  IDG_REQUIRE(d == D, "idg_ngcf_layer_fwd_f32: d = 64 only (got %lld); other widths: idg_ngcf_transform_f32 + idg_ngcf_tail_ex_f32",
              (long long)d);
  IDG_REQUIRE(n >= 0 && ldn >= d && ldn % 4 == 0 && p >= 0.f && p < 1.f, "idg_ngcf_layer_fwd_f32: bad sizes / drop probability");
  IDG_REQUIRE(((uintptr_t)side | (uintptr_t)ego | (uintptr_t)E | (uintptr_t)N) % 16 == 0,
              "idg_ngcf_layer_fwd_f32: panels must be 16-byte aligned");
  if (n == 0) return IDG_OK;
  hipLaunchKernelGGL(ngcf_layer_fwd64_kernel, dim3((unsigned)((n + RB - 1) / RB)), dim3(BLOCK), 0, (hipStream_t)stream, side, ego, W1,
                     W2, b1, b2, n, negative_slope, p, seed, stream_id, E, N, ldn);
  IDG_HIP(hipGetLastError());
  return IDG_OK;
}

size_t idg_ngcf_layer_bwd_workspace_bytes(int64_t d) {
  if (d != D) return 0;
  return (size_t)BWD_WGS * (size_t)(2 * D * D + 2 * D) * sizeof(float);
}

int idg_ngcf_layer_bwd_f32(const float* E, const float* gE, const float* gN, int64_t ldgn, const uint32_t* gn_rows, const float* side,
                           const float* ego, const float* W1, const float* W2, int64_t n, int64_t d, float negative_slope, float p,
                           uint64_t seed, uint64_t stream_id, float* g_side, float* g_ego, float* w_grads, void* ws, void* stream) {
  IDG_REQUIRE(E && (gE || gN) && side && ego && W1 && W2 && g_side && g_ego && w_grads && ws, "idg_ngcf_layer_bwd_f32: NULL argument");
  IDG_REQUIRE(d == D, "idg_ngcf_layer_bwd_f32: d = 64 only (got %lld); other widths: the tail / wgrad / transform chain", (long long)d);
  IDG_REQUIRE(n > 0 && (!gN || (ldgn >= d && ldgn % 4 == 0 && ldgn < (1 << 20))) && p >= 0.f && p < 1.f, "idg_ngcf_layer_bwd_f32: bad sizes / drop probability");
  IDG_REQUIRE(((uintptr_t)E | (uintptr_t)gE | (uintptr_t)gN | (uintptr_t)side | (uintptr_t)ego | (uintptr_t)W1 | (uintptr_t)W2) % 16 == 0,
              "idg_ngcf_layer_bwd_f32: panels and weights must be 16-byte aligned");
  hipStream_t st = (hipStream_t)stream;
  const int64_t n_blocks = (n + RB - 1) / RB;
  // two resident workgroups on EVERY CU (a count in between leaves some CUs with one workgroup and some with two: the
  // launch then lasts as long as the doubly loaded ones)
  const int64_t wgs = n_blocks < BWD_WGS ? n_blocks : BWD_WGS;
  float* part = reinterpret_cast<float*>(ws);
  hipLaunchKernelGGL(ngcf_layer_bwd64_kernel, dim3((unsigned)wgs), dim3(BWD_BLOCK), 0, st, E, gE, gN, ldgn, gn_rows, side, ego, W1, W2,
                     n, negative_slope, p, seed, stream_id, g_side, g_ego, part);
  const int64_t count = 2 * D * D + 2 * D;
  hipLaunchKernelGGL(slices_reduce_kernel, dim3((unsigned)((count + 63) / 64)), dim3(64 * RG), 0, st, part, wgs, count, w_grads);
  IDG_HIP(hipGetLastError());
  return IDG_OK;
}

}  // extern "C"
