// NGCF's message dropout (models/NGCF.py:104): a counter-based keep mask, a function of (seed, stream, row, feature) only,
// so the backward kernels regenerate it instead of reading a stored mask.  Shared by idg_dense.hip and idg_ngcf.hip.
// Four neighbouring features share ONE 64-bit mix (16 bits each): the kernels hold four features of a row per lane, and
// the mix — two 64-bit multiplies — was as much vector work as the layer's matrix products (round 4).
#pragma once
#include <cstdint>

namespace idg {

__device__ __forceinline__ uint64_t mix64(uint64_t seed, uint64_t stream, int64_t row, int64_t f4) {
  // splitmix64 finaliser of the coordinates (row, feature / 4)
  uint64_t z = seed + 0x9E3779B97F4A7C15ull * (stream + 1) + (uint64_t)row * 0xBF58476D1CE4E5B9ull + (uint64_t)f4 * 0x94D049BB133111EBull;
  z = (z ^ (z >> 30)) * 0xBF58476D1CE4E5B9ull;
  z = (z ^ (z >> 27)) * 0x94D049BB133111EBull;
  return z ^ (z >> 31);
}
__device__ __forceinline__ float keep_of_bits(float p, uint32_t bits16) {
  const float u = (float)bits16 * (1.0f / 65536.0f);  // [0, 1)
  return u >= p ? 1.0f / (1.0f - p) : 0.0f;
}
// 1 / (1 - p) where the element is kept, 0 where it is dropped (p <= 0: always 1)
__device__ __forceinline__ float keep_scale(float p, uint64_t seed, uint64_t stream, int64_t row, int64_t f) {
  if (p <= 0.f) return 1.0f;
  return keep_of_bits(p, (uint32_t)(mix64(seed, stream, row, f >> 2) >> (16 * (f & 3))) & 0xFFFFu);
}
// the same for features f .. f + 3 of one row (f % 4 == 0): one mix
__device__ __forceinline__ void keep_scale4(float p, uint64_t seed, uint64_t stream, int64_t row, int64_t f, float k[4]) {
  if (p <= 0.f) {
    k[0] = k[1] = k[2] = k[3] = 1.0f;
    return;
  }
  const uint64_t z = mix64(seed, stream, row, f >> 2);
#pragma unroll
  for (int c = 0; c < 4; ++c) k[c] = keep_of_bits(p, (uint32_t)(z >> (16 * c)) & 0xFFFFu);
}

}  // namespace idg
