// Sparse graph handle and the CSR row-block SpMM for gfx950.
//
// Work decomposition (all fixed at idg_graph_create, so results are run-to-run identical):
//   * every CSR row is one "virtual row" (vrow); a row with more than `split_threshold`
//     stored entries is cut into segments of seg_len(row) entries, each its own vrow that
//     produces a partial sum; the partials are added in a fixed 4-way strided order
//     (partials q, q+4, q+8, ... summed in order for q = 0..3, the four sums then added left
//     to right).  All segments of a row of <= CHUNK_NNZ entries sit in ONE tile: their
//     partials live in LDS and the workgroup combines them itself ("local" rows).  A longer
//     row is first cut into chunks of CHUNK_NNZ entries; each chunk is such a local row whose
//     sum goes to a global partial slot, and the chunk sums are combined in the same 4-way
//     order by the last workgroup to arrive (IDG_FUSED_FIX=0: by a separate fix-up kernel).
//   * consecutive vrows are packed into tiles of <= tile_cap entries and <= TILE_VROWS
//     vrows.  One 256-thread workgroup per tile stages the tile's (column,value) pairs and
//     vrow pointers in LDS with coalesced loads, then LPR = d/4 lanes walk one vrow each:
//     16-byte loads of the dense panel row (one 4*d-byte row per LPR lanes, fully
//     coalesced), 8 rows in flight per lane group, a strictly sequential fmaf chain per
//     output element — the order torch's CPU sparse.mm uses, hence bit-identical to it.
//   * HBM/L2-bound: no MFMA here by design.
#include <hip/hip_runtime.h>

#include <algorithm>
#include <climits>
#include <cmath>
#include <cstdint>
#include <cstdlib>
#include <functional>
#include <memory>
#include <mutex>
#include <atomic>
#include <new>
#include <thread>
#include <vector>

#include "idg_common.h"

#ifndef IDG_ROWS_UNROLL
#define IDG_ROWS_UNROLL 8  // panel rows in flight per lane group in the row-restricted / sparse-input kernels (few lane groups
#endif                     // are busy there: the walk of a long row is a chain of dependent rounds, wider rounds shorten it)
#ifndef IDG_FUSED_MINW
#define IDG_FUSED_MINW 1   // __launch_bounds__ minimum waves per SIMD of the dense kernel with in-kernel split-row combine
#endif
#ifndef IDG_UNITS_UNROLL
#define IDG_UNITS_UNROLL 8  // panel rows in flight per lane group in the one-wave-per-unit kernel (16 / 32 measured: no gain)
#endif
#ifndef IDG_UNITS_BLOCK
#define IDG_UNITS_BLOCK 64  // threads per workgroup of the one-wave-per-unit kernel.  One wave: the chunks of a hub row are
#endif                      // consecutive units, and a workgroup's waves share a CU — whose miss path, not the chip's, then
                            // bounds the row (measured: 256 -> 64 threads 24.4 -> 20.4 us for a yelp2018-size batch)
#ifndef IDG_WALK_TAIL
#define IDG_WALK_TAIL 1  // 0: the remainder of a row as a 4-batch + single loads (the round-1 form; kept for A/B timing)
#endif

namespace {

constexpr int TILE_NNZ = 1024;    // LDS capacity: entries staged per workgroup (8 KiB)
constexpr int CHUNK_NNZ = 512;    // longest row (or piece of a row) whose segments are combined inside one workgroup
constexpr int LSLOTS = 8;         // LDS partial slots per tile = segments of local rows a tile may hold
constexpr int32_t LOCAL_CODE = INT32_MIN;  // vtgt of a local segment: LOCAL_CODE + its LDS slot
constexpr int TILE_VROWS = 256;   // vrows per workgroup
constexpr int BLOCK = 256;
constexpr int64_t DEFAULT_SPLIT = 128;
constexpr int DEFAULT_TILE_CAP = 512;  // entries per tile: small tiles even out work per CU (measured, profiles/r01)
constexpr int64_t BAND_PANEL_BYTES = (int64_t)128 << 20;  // use XCD band placement up to this gathered-panel size
constexpr int FIX_WAYS = 4;            // lane groups that share one split row in the fix-up pass
constexpr int32_t UNIT_LONG = INT32_MIN;  // row_unit code of a chunked row: UNIT_LONG + its LongRow index
constexpr int UNITS_HEADER = 2;           // words in front of a live-unit list: [count, overflow flag]

struct __attribute__((aligned(16))) Tile {
  int64_t nnz_begin;
  int32_t vrow_begin;
  int16_t n_vrows;  // 1..TILE_VROWS
  int16_t n_local;  // local rows (LocalRow entries) of this tile
  int32_t local_begin;
  int32_t row_first, row_last;  // the tile's vrows produce (pieces of) the consecutive rows row_first..row_last
  int32_t nnz_count;  // entries of the tile: lets a kernel stage them without first reading the tile's last vrow pointer
};

struct LocalRow {
  int32_t tgt;    // >= 0: output row; < 0: ~global partial slot (a chunk of a longer row)
  int32_t vrow;   // first segment (global vrow index); segments are consecutive vrows
  int16_t n_seg;  // 2..LSLOTS
  int16_t lslot;  // first LDS partial slot
};

struct ColVal {
  int32_t col;
  float val;
};

struct LongRow {
  int64_t slot_begin;  // first partial slot
  int32_t row;
  int32_t n_seg;
};

// what follows the product in the epilogue, beyond addend / running sum / divide
constexpr int EPI_PLAIN = 0;
constexpr int EPI_NOISE = 1;  // SimGCL / XSimGCL perturbation of t = A.X
constexpr int EPI_ADAM = 2;   // the value stored to sum_out is a finished gradient: apply the Adam update to its row
constexpr int EPI_ACT = 3;    // t = tanh(t) (act 1) or t *= 1 - act_src[r]^2 (act 2: tanh's derivative) before anything is stored:
//                               EGCF's layers (models/EGCF.py:46-84: activation_layer(torch.sparse.mm(...))) and their backward

struct Epilogue {
  float* Y;             // [n_rows, ldy]   (nullable)
  const float* addend;  // acc += addend[r] (nullable)
  const float* sum_in;  // s = sum_in[r] + acc (nullable -> s = acc)
  const float* sum_in2; // with sum_in: s = ((sum_in[r] + sum_in2[r]) + sum_in3[r]) + acc, left to right (each nullable):
  const float* sum_in3; // the layer mean's additions in torch's order when the earlier layers are summed only at the end
  float* sum_out;       // sum_out[r] = s / div (nullable)
  int64_t ldy;          // leading dimension of Y/addend/sum_in/sum_out
  float div;            // 1 = no division
  int accumulate;       // sum_out[r] += instead of =
  const uint32_t* mask; // row bitmap of addend / sum_in: rows with a 0 bit are zero and are not read (nullable)
  // SimGCL / XSimGCL perturbation applied to t = A.X before anything else (models/SimGCL.py:50-51):
  //   t += sign(t) * normalize(u, dim=-1) * noise_eps,  u ~ U[0,1)^d from Philox4x32-10(seed; row, block, stream)
  float noise_eps;      // 0 = off
  uint64_t noise_seed;
  uint64_t noise_stream;
  // EPI_ADAM (last backward product of a training step): rows of the parameter / moment panels, same layout
  // as sum_out; constants as idg_adam_step_f32 derives them (torch.optim.Adam defaults, trainer.py:11)
  float* adam_p;
  float* adam_m;
  float* adam_v;
  float adam_w1, adam_beta2, adam_w2, adam_step_size, adam_bc2_sqrt, adam_eps;
  int adam_discard;     // the finished gradient feeds the update and is NOT written to sum_out (4 B / element less)
  // EPI_ACT: applied to t = A.X (+ addend) of rows r < act_rows (0: every row), before Y / sum_out see it
  int act;              // 0 none, 1 t = tanh(t), 2 t = t * (1 - act_src[r]^2), 3 none — but the EPI_ACT instantiation (y24)
  const float* act_src; // act 2: the SAVED tanh output of the forward layer (same layout as Y)
  int64_t act_rows;
  // the finished row ALSO (or only: Y may be NULL) as 24-bit values, three words per four values at y24 + (r ldy + off) / 4 * 3
  // (idg_pack24_f32's format: what the sharded step's packed exchange sends — the product writes its partial straight into
  // the send buffer).  Lives in the EPI_ACT instantiations (act = 3 when no activation is asked for): the plain kernels'
  // code and register budget are untouched.
  uint32_t* y24;
};

}  // namespace

struct idg_graph {
  int device = -1;
  int64_t n_rows = 0, n_cols = 0, nnz = 0;
  uint32_t flags = 0;
  int64_t split_threshold = 0;
  int64_t n_vrows = 0, n_tiles = 0, n_long = 0, n_slots = 0, n_xl = 0, n_local = 0, n_split = 0;
  int variant = 5;  // tuning knob (IDG_SPMM_VARIANT), see launch_fast
  // Rows of more than CHUNK_NNZ entries leave one global partial per chunk.  By default the workgroup that
  // delivers a row's last chunk adds them up (last-arriver form, below): no second launch.  IDG_FUSED_FIX=0
  // selects the separate fix-up launch instead (same bits; measured 6 us per product slower on the benchmark
  // graphs now that only ~1 % of the rows and a few hundred partials go through the protocol).
  bool no_fused_fix = false;
  bool no_units = false;  // IDG_LIVE_UNITS=0: ignore registered live-unit lists (restricted launches visit the tiles)
  int64_t tile_cap = DEFAULT_TILE_CAP;  // entries per tile (IDG_TILE_NNZ, <= TILE_NNZ)
  // device
  ColVal* d_cv = nullptr;
  int64_t* d_vptr = nullptr;   // [n_vrows+1]
  int32_t* d_vtgt = nullptr;   // [n_vrows] >=0 row id, <0 ~partial slot, < LOCAL_CODE + LSLOTS: LDS slot of a local segment
  LocalRow* d_local = nullptr; // local rows, grouped by tile
  Tile* d_tiles = nullptr;         // heaviest-first order
  Tile* d_tiles_banded = nullptr;  // XCD column-band placement (used when the gathered panel is cache resident)
  Tile* d_tiles_seq = nullptr;     // the bands one after the other (used when it is not): at any moment the chip gathers
                                   // from one band's share of the panel — for the bipartite adjacency, first all user
                                   // rows (item panel), then all item rows (user panel)
  LongRow* d_long = nullptr;
  int32_t* d_slot_row = nullptr;   // partial slot -> the row it belongs to
  int32_t* d_slot_long = nullptr;  // partial slot -> index into d_long
  int* d_long_cnt = nullptr;       // arrival tickets of the in-kernel split-row combine (zero between launches)
  int32_t* d_xl = nullptr;     // vrows too long for one tile (EXACT_ORDER only)
  int32_t* d_vrow_row = nullptr; // vrow -> the CSR row it is (a piece of) (entry -> row, for idg_graph_masked_copy)
  bool borrowed = false;       // a masked copy: everything but d_cv / d_long_cnt belongs to the handle it was made from
  // row -> work unit (idg_graph_live_units): >= 0 the row's vrow; ~li a split row combined in LDS (LocalRow li);
  // UNIT_LONG + i a chunked row (LongRow i; one unit per chunk: d_slot_unit[slot], a vrow or ~LocalRow whose target is
  // the chunk's global partial slot)
  int32_t* d_row_unit = nullptr;
  int32_t* d_slot_unit = nullptr;
  // host copies for the checker
  std::vector<int64_t> h_long_rows, h_seg_len, h_chunk_len;
};

// ---- registry of live-unit lists (idg_graph_live_units) ---------------------------------------------------------
// (schedule, row bitmap) -> the bitmap's list of work units.  Process-wide and keyed by the SCHEDULE a handle runs on
// (its vrow pointer table, which masked / revalued copies share with their base), so a copy finds its base's lists
// without a second registration, and nothing is copied with a handle.  Every library entry point that WRITES a row
// bitmap (idg_bpr_touch_rows, idg_bitmap_clear, the `touched` argument of the BPR backward, idg_graph_expand_rows,
// idg_graph_mark_cols, pack / unpack of gradient rows) drops the lists of that bitmap first (idg::rows_changed): a
// restricted launch that finds no list visits the tiles, which is always correct.  A buffer freed and handed out again
// by the caller's allocator is therefore harmless as soon as its new owner fills it through the library; callers that
// write a registered bitmap by other means must call idg_graph_forget_live_units (include/idgrec.h).
namespace {
struct UnitList {
  const void* sched = nullptr;
  const uint32_t* bitmap = nullptr;
  const int32_t* units = nullptr;
  int64_t cap = 0;
  int kind = 0;  // 0: live work units of the bitmap's OUTPUT rows; 1: the tiles' entry lists compacted to the bitmap's INPUT rows
};
constexpr int MAX_UNIT_LISTS = 64;
UnitList g_unit_lists[MAX_UNIT_LISTS];
int g_unit_next = 0;
std::mutex g_unit_mutex;

void units_register(const void* sched, const uint32_t* bitmap, const int32_t* units, int64_t cap, int kind = 0) {
  std::lock_guard<std::mutex> lock(g_unit_mutex);
  int at = -1;
  for (int i = 0; i < MAX_UNIT_LISTS; ++i)
    if (g_unit_lists[i].sched == sched && g_unit_lists[i].bitmap == bitmap && g_unit_lists[i].kind == kind) at = i;
  if (at < 0)
    for (int i = 0; i < MAX_UNIT_LISTS && at < 0; ++i)
      if (g_unit_lists[i].bitmap == nullptr) at = i;
  if (at < 0) at = g_unit_next, g_unit_next = (g_unit_next + 1) % MAX_UNIT_LISTS;  // full: the oldest slot is replaced
  g_unit_lists[at] = UnitList{sched, bitmap, units, cap, kind};
}

bool units_find(const void* sched, const uint32_t* bitmap, const int32_t** units, int64_t* cap, int kind = 0) {
  std::lock_guard<std::mutex> lock(g_unit_mutex);
  for (int i = 0; i < MAX_UNIT_LISTS; ++i)
    if (g_unit_lists[i].sched == sched && g_unit_lists[i].bitmap == bitmap && bitmap != nullptr && g_unit_lists[i].kind == kind) {
      if (units) *units = g_unit_lists[i].units;
      if (cap) *cap = g_unit_lists[i].cap;
      return true;
    }
  return false;
}

// sched == nullptr: every schedule; bitmap == nullptr: every bitmap, else the lists whose bitmap STARTS inside
// [bitmap, bitmap + bytes) (bytes = 0: at `bitmap` itself) — a list may be registered for a sub-range of a larger bitmap
// (a row slice of the sharded step's item bitmap)
void units_forget(const void* sched, const void* bitmap, size_t bytes = 0) {
  std::lock_guard<std::mutex> lock(g_unit_mutex);
  const char* lo = reinterpret_cast<const char*>(bitmap);
  const char* hi = lo + (bytes > 0 ? bytes : 1);
  for (int i = 0; i < MAX_UNIT_LISTS; ++i) {
    const char* b = reinterpret_cast<const char*>(g_unit_lists[i].bitmap);
    if ((sched == nullptr || g_unit_lists[i].sched == sched) && b != nullptr && (bitmap == nullptr || (b >= lo && b < hi)))
      g_unit_lists[i] = UnitList{};
  }
}
// every list (of either kind, on any schedule) that lives in the caller's buffer `ws`
void units_forget_ws(const void* ws) {
  std::lock_guard<std::mutex> lock(g_unit_mutex);
  for (int i = 0; i < MAX_UNIT_LISTS; ++i)
    if (ws != nullptr && g_unit_lists[i].units == ws) g_unit_lists[i] = UnitList{};
}
}  // namespace

namespace idg {
void rows_changed(const void* bitmap, size_t bytes) {
  if (bitmap) units_forget(nullptr, bitmap, bytes);
}
}  // namespace idg

namespace {

__device__ __forceinline__ float4 fma4(float a, float4 x, float4 acc) {
  acc.x = __builtin_fmaf(a, x.x, acc.x);
  acc.y = __builtin_fmaf(a, x.y, acc.y);
  acc.z = __builtin_fmaf(a, x.z, acc.z);
  acc.w = __builtin_fmaf(a, x.w, acc.w);
  return acc;
}

__device__ __forceinline__ float4 add4(float4 a, float4 b) {
  return make_float4(a.x + b.x, a.y + b.y, a.z + b.z, a.w + b.w);
}

// Philox4x32-10 (Salmon et al. 2011): counter-based, so a row's noise depends only on
// (seed, stream, row, feature block) — independent of tiling, launch order and split schedule.
__device__ __forceinline__ uint4 philox4x32_10(uint4 ctr, uint2 key) {
#pragma unroll
  for (int r = 0; r < 10; ++r) {
    const uint32_t hi0 = __umulhi(0xD2511F53u, ctr.x), lo0 = 0xD2511F53u * ctr.x;
    const uint32_t hi1 = __umulhi(0xCD9E8D57u, ctr.z), lo1 = 0xCD9E8D57u * ctr.z;
    ctr = make_uint4(hi1 ^ ctr.y ^ key.x, lo1, hi0 ^ ctr.w ^ key.y, lo0);
    key.x += 0x9E3779B9u;
    key.y += 0xBB67AE85u;
  }
  return ctr;
}

// 4 uniforms in [0,1) for features [4*fblock, 4*fblock+4) of row r
__device__ __forceinline__ float4 noise4(const Epilogue& ep, int64_t r, int fblock);

__device__ __forceinline__ bool mask_bit(const uint32_t* __restrict__ mask, int64_t r) {
  return (mask[r >> 5] >> (r & 31)) & 1u;
}

__device__ __forceinline__ float4 noise4(const Epilogue& ep, int64_t r, int fblock) {
  const uint4 c = make_uint4((uint32_t)r, (uint32_t)((uint64_t)r >> 32), (uint32_t)fblock, (uint32_t)ep.noise_stream);
  const uint2 k = make_uint2((uint32_t)ep.noise_seed, (uint32_t)(ep.noise_seed >> 32) ^ (uint32_t)(ep.noise_stream >> 32));
  const uint4 x = philox4x32_10(c, k);
  const float sc = 1.0f / 16777216.0f;  // 24 random bits -> [0, 1), like torch.rand
  return make_float4((x.x >> 8) * sc, (x.y >> 8) * sc, (x.z >> 8) * sc, (x.w >> 8) * sc);
}

// t += sign(t) * u / max(||u||_2, 1e-12) * eps over the WHOLE row: the LPR lanes of the group hold
// NB feature blocks each; the row norm is reduced across them with shuffles.
template <int LPR, int NB>
__device__ __forceinline__ float noise_row_scale(const Epilogue& ep, int64_t r, int l) {
  float ss = 0.f;
#pragma unroll
  for (int b = 0; b < NB; ++b) {
    const float4 u = noise4(ep, r, b * LPR + l);
    ss += u.x * u.x + u.y * u.y + u.z * u.z + u.w * u.w;
  }
#pragma unroll
  for (int o = LPR / 2; o > 0; o >>= 1) ss += __shfl_xor(ss, o, LPR);
  return ep.noise_eps / fmaxf(sqrtf(ss), 1e-12f);
}

__device__ __forceinline__ float sgn(float x) { return (float)(x > 0.f) - (float)(x < 0.f); }

__device__ __forceinline__ float4 perturb(const Epilogue& ep, int64_t r, int fblock, float scale, float4 t) {
  const float4 u = noise4(ep, r, fblock);
  // explicit fused multiply-adds: every instantiation that perturbs (epilogues, the stand-alone kernel, the multi-panel
  // kernel) then rounds the same way, whatever the compiler would have contracted on its own
  t.x = __builtin_fmaf(sgn(t.x) * u.x, scale, t.x);
  t.y = __builtin_fmaf(sgn(t.y) * u.y, scale, t.y);
  t.z = __builtin_fmaf(sgn(t.z) * u.z, scale, t.z);
  t.w = __builtin_fmaf(sgn(t.w) * u.w, scale, t.w);
  return t;
}

template <int EPI = EPI_PLAIN>
__device__ __forceinline__ void epilogue_store(const Epilogue& ep, int64_t r, int off, float4 acc) {
  const int64_t o = r * ep.ldy + off;
  const bool live = ep.mask == nullptr || mask_bit(ep.mask, r);  // x + 0 == x: skipping a zero row is exact
  if (ep.addend && live) acc = add4(acc, *reinterpret_cast<const float4*>(ep.addend + o));
  if (EPI == EPI_ACT && (ep.act_rows == 0 || r < ep.act_rows)) {
    if (ep.act == 1) {
      acc.x = tanhf(acc.x), acc.y = tanhf(acc.y), acc.z = tanhf(acc.z), acc.w = tanhf(acc.w);
    } else if (ep.act == 2) {  // d tanh(z) / dz = 1 - tanh(z)^2, with tanh(z) read back (torch's tanh_backward: g * (1 - y*y))
      const float4 y = *reinterpret_cast<const float4*>(ep.act_src + o);
      acc.x = acc.x * (1.0f - y.x * y.x), acc.y = acc.y * (1.0f - y.y * y.y);
      acc.z = acc.z * (1.0f - y.z * y.z), acc.w = acc.w * (1.0f - y.w * y.w);
    }
  }
  if (ep.Y) *reinterpret_cast<float4*>(ep.Y + o) = acc;
  if (EPI == EPI_ACT && ep.y24) {
    auto top24 = [](float x) {
      const uint32_t b = __float_as_uint(x);
      return (b + 0x7Fu + ((b >> 8) & 1u)) >> 8;  // (round to nearest even: idg_shard.hip top24)
    };
    const uint32_t a = top24(acc.x), b = top24(acc.y), c = top24(acc.z), e = top24(acc.w);
    uint32_t* q = ep.y24 + (o >> 2) * 3;
    q[0] = a | (b << 24), q[1] = (b >> 8) | (c << 16), q[2] = (c >> 16) | (e << 8);
  }
  if (ep.sum_out) {
    float4 s = acc;
    if (ep.sum_in && live) {
      float4 t = *reinterpret_cast<const float4*>(ep.sum_in + o);
      if (ep.sum_in2) t = add4(t, *reinterpret_cast<const float4*>(ep.sum_in2 + o));
      if (ep.sum_in3) t = add4(t, *reinterpret_cast<const float4*>(ep.sum_in3 + o));
      s = add4(t, acc);
    }
    if (ep.div != 1.0f) {
      s.x = s.x / ep.div;
      s.y = s.y / ep.div;
      s.z = s.z / ep.div;
      s.w = s.w / ep.div;
    }
    if (ep.accumulate && live) s = add4(*reinterpret_cast<const float4*>(ep.sum_out + o), s);
    if (EPI != EPI_ADAM || !ep.adam_discard) *reinterpret_cast<float4*>(ep.sum_out + o) = s;
    if (EPI == EPI_ADAM) {  // same arithmetic, operation for operation, as adam_kernel (idg_bpr.hip)
      float4 P = *reinterpret_cast<const float4*>(ep.adam_p + o);
      float4 M = *reinterpret_cast<const float4*>(ep.adam_m + o);
      float4 V = *reinterpret_cast<const float4*>(ep.adam_v + o);
#define IDG_ADAM1(c)                                                       \
  M.c = __builtin_fmaf(ep.adam_w1, s.c - M.c, M.c);                        \
  V.c = __builtin_fmaf(ep.adam_w2 * s.c, s.c, V.c * ep.adam_beta2);        \
  P.c = P.c - ep.adam_step_size * (M.c / (sqrtf(V.c) / ep.adam_bc2_sqrt + ep.adam_eps));
      IDG_ADAM1(x) IDG_ADAM1(y) IDG_ADAM1(z) IDG_ADAM1(w)
#undef IDG_ADAM1
      *reinterpret_cast<float4*>(ep.adam_p + o) = P;
      *reinterpret_cast<float4*>(ep.adam_m + o) = M;
      *reinterpret_cast<float4*>(ep.adam_v + o) = V;
    }
  }
}

// Sequential walk of entries [s, e) of a (col,val) list; CV may be LDS or global.
// UNROLL panel rows are in flight per lane group; the fmaf chain stays strictly in order.
template <int UNROLL, typename CVPtr>
__device__ __forceinline__ float4 walk(CVPtr cv, int s, int e, const float* __restrict__ Xl,
                                       int64_t ldx, float4 acc) {
  int j = s;
  for (; j + UNROLL <= e; j += UNROLL) {
    ColVal p[UNROLL];
    float4 x[UNROLL];
#pragma unroll
    for (int u = 0; u < UNROLL; ++u) p[u] = cv[j + u];
#pragma unroll
    for (int u = 0; u < UNROLL; ++u) x[u] = *reinterpret_cast<const float4*>(Xl + (int64_t)p[u].col * ldx);
#pragma unroll
    for (int u = 0; u < UNROLL; ++u) acc = fma4(p[u].val, x[u], acc);
  }
#if IDG_WALK_TAIL
  // The last (e - j) < UNROLL entries of the row as ONE more round instead of a 4-batch plus up to three dependent
  // single loads: every slot loads a valid entry of this row (slots past the end re-read the row's last entry: an L1
  // hit), and a slot past the end leaves the accumulator as it is — a select on the fmaf's result, so the chain is the
  // same sequence of operations on the same operands (no 0 * x terms: exact for any panel contents).
  if (j < e) {
    ColVal p[UNROLL];
    float4 x[UNROLL];
#pragma unroll
    for (int u = 0; u < UNROLL; ++u) p[u] = cv[j + u < e ? j + u : e - 1];
#pragma unroll
    for (int u = 0; u < UNROLL; ++u) x[u] = *reinterpret_cast<const float4*>(Xl + (int64_t)p[u].col * ldx);
#pragma unroll
    for (int u = 0; u < UNROLL; ++u) {
      const float4 t = fma4(p[u].val, x[u], acc);
      const bool in = j + u < e;
      acc.x = in ? t.x : acc.x;
      acc.y = in ? t.y : acc.y;
      acc.z = in ? t.z : acc.z;
      acc.w = in ? t.w : acc.w;
    }
  }
#else
  if (UNROLL > 8 && j + 8 <= e) {
    ColVal p[8];
    float4 x[8];
#pragma unroll
    for (int u = 0; u < 8; ++u) p[u] = cv[j + u];
#pragma unroll
    for (int u = 0; u < 8; ++u) x[u] = *reinterpret_cast<const float4*>(Xl + (int64_t)p[u].col * ldx);
#pragma unroll
    for (int u = 0; u < 8; ++u) acc = fma4(p[u].val, x[u], acc);
    j += 8;
  }
  if (j + 4 <= e) {
    ColVal p[4];
    float4 x[4];
#pragma unroll
    for (int u = 0; u < 4; ++u) p[u] = cv[j + u];
#pragma unroll
    for (int u = 0; u < 4; ++u) x[u] = *reinterpret_cast<const float4*>(Xl + (int64_t)p[u].col * ldx);
#pragma unroll
    for (int u = 0; u < 4; ++u) acc = fma4(p[u].val, x[u], acc);
    j += 4;
  }
  for (; j < e; ++j) {
    ColVal p = cv[j];
    float4 x = *reinterpret_cast<const float4*>(Xl + (int64_t)p.col * ldx);
    acc = fma4(p.val, x, acc);
  }
#endif
  return acc;
}

// walk() for an entry list in GLOBAL memory (the one-wave-per-unit kernel reads the CSR entries where they lie): a round of
// walk() is two dependent memory round trips — the entries, then the panel rows they name — and the row's chain of rounds
// is what the launch lasts (a 512-entry chunk: 64 rounds).  Here the NEXT round's entries are loaded before this round's
// rows are waited for, so a round costs the longer of the two trips instead of their sum.  Same entries, same order of
// fmaf: bit-identical.
#ifndef IDG_WALK_PREFETCH
#define IDG_WALK_PREFETCH 1
#endif
template <int UNROLL>
__device__ __forceinline__ float4 walk_global(const ColVal* __restrict__ cv, int e, const float* __restrict__ Xl, int64_t ldx,
                                              float4 acc) {
#if IDG_WALK_PREFETCH
  if (e < 2 * UNROLL) return walk<UNROLL>(cv, 0, e, Xl, ldx, acc);
  ColVal p[UNROLL];
#pragma unroll
  for (int u = 0; u < UNROLL; ++u) p[u] = cv[u];
  int j = 0;
  for (; j + 2 * UNROLL <= e; j += UNROLL) {
    float4 x[UNROLL];
    ColVal pn[UNROLL];
#pragma unroll
    for (int u = 0; u < UNROLL; ++u) x[u] = *reinterpret_cast<const float4*>(Xl + (int64_t)p[u].col * ldx);
#pragma unroll
    for (int u = 0; u < UNROLL; ++u) pn[u] = cv[j + UNROLL + u];
#pragma unroll
    for (int u = 0; u < UNROLL; ++u) acc = fma4(p[u].val, x[u], acc);
#pragma unroll
    for (int u = 0; u < UNROLL; ++u) p[u] = pn[u];
  }
  {  // the last full round (its entries are in p already)
    float4 x[UNROLL];
#pragma unroll
    for (int u = 0; u < UNROLL; ++u) x[u] = *reinterpret_cast<const float4*>(Xl + (int64_t)p[u].col * ldx);
#pragma unroll
    for (int u = 0; u < UNROLL; ++u) acc = fma4(p[u].val, x[u], acc);
    j += UNROLL;
  }
  return walk<UNROLL>(cv, j, e, Xl, ldx, acc);  // the remainder (< UNROLL entries), as walk() finishes a row
#else
  return walk<UNROLL>(cv, 0, e, Xl, ldx, acc);
#endif
}

// The same walk over a panel whose all-zero rows are flagged in a bitmap: entries that point at a
// zero row are skipped (fmaf(v, +0, acc) == acc for the finite, non-negative-zero operands here),
// so only live rows are fetched.  Used by the first backward layer, whose input has <= 3B live rows.
template <int UNROLL, typename CVPtr>
__device__ __forceinline__ float4 walk_masked(CVPtr cv, int s, int e, const float* __restrict__ Xl, int64_t ldx,
                                              const uint32_t* __restrict__ mask, float4 acc) {
  int j = s;
  for (; j + UNROLL <= e; j += UNROLL) {
    ColVal p[UNROLL];
    bool live[UNROLL];
    float4 x[UNROLL];
#pragma unroll
    for (int u = 0; u < UNROLL; ++u) p[u] = cv[j + u];
#pragma unroll
    for (int u = 0; u < UNROLL; ++u) live[u] = mask_bit(mask, p[u].col);
#pragma unroll
    for (int u = 0; u < UNROLL; ++u)
      if (live[u]) x[u] = *reinterpret_cast<const float4*>(Xl + (int64_t)p[u].col * ldx);
#pragma unroll
    for (int u = 0; u < UNROLL; ++u)
      if (live[u]) acc = fma4(p[u].val, x[u], acc);
  }
  for (; j < e; ++j) {
    ColVal p = cv[j];
    if (mask_bit(mask, p.col)) acc = fma4(p.val, *reinterpret_cast<const float4*>(Xl + (int64_t)p.col * ldx), acc);
  }
  return acc;
}

// ---- split rows combined inside the tile kernel (no separate fix-up launch) ---------------
// A lane group that finishes a segment stores its partial WRITE-THROUGH (sc1), its wave drains
// vmcnt(0), the group leader draws a ticket from the row's arrival counter (relaxed, agent
// scope); the group that draws the last ticket acquires (agent scope: drops this CU's stale L1
// lines) and adds all the row's partials in the published 4-way strided order, then runs the
// epilogue.  This is the split-K "last arriver reduces" form of the hardware guide (write-through
// slabs + drained ticket; the reducer reads the slabs with sc1 loads — an acquire fence instead
// invalidates the CU's L1, where the hot panel rows live, once per split row: measured slower);
// results do not depend on which group arrives last, so
// they are bit-identical to the separate fix-up pass.  The reducer resets the counter for the
// next launch (launches on one stream are serial; counters are zero at graph creation).
struct FixCtx {
  const LongRow* rows;
  const int32_t* slot_long;
  int* cnt;
  uint32_t part_bytes;  // 0 = not fused: plain partial stores + spmm_fixup_kernel
};

using u32x4 = __attribute__((ext_vector_type(4))) unsigned int;

template <int LPR, int NB, int EPI>
__device__ __forceinline__ void combine_if_last(const Epilogue& ep, const float* __restrict__ partials, int64_t d,
                                                const FixCtx& fx, int slot, int l) {
  asm volatile("s_waitcnt vmcnt(0)" ::: "memory");  // this wave's write-through partial stores have completed
  const int li = fx.slot_long[slot];
  int old = 0;
  if (l == 0) old = __hip_atomic_fetch_add(fx.cnt + li, 1, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
  old = __shfl(old, (int)(threadIdx.x % 64) / LPR * LPR, 64);
  const LongRow lr = fx.rows[li];
  if (old != lr.n_seg - 1) return;
  if (l == 0) __hip_atomic_store(fx.cnt + li, 0, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
  // every load of the handed-off partials is an sc1 (agent-scope, L1-bypassing) buffer load
  __builtin_amdgcn_fence(__ATOMIC_ACQUIRE, "wavefront");  // no instruction: keeps the loads below the ticket
  const __amdgpu_buffer_rsrc_t prs = __builtin_amdgcn_make_buffer_rsrc(const_cast<float*>(partials), 0, (int)fx.part_bytes, 0x00020000);
  auto ld = [&](int64_t elem) {
    const u32x4 r = __builtin_amdgcn_raw_buffer_load_b128(prs, (unsigned)(elem * 4), 0, 16);  // aux 16 = sc1
    return make_float4(__uint_as_float(r.x), __uint_as_float(r.y), __uint_as_float(r.z), __uint_as_float(r.w));
  };
  float nscale = 0.f;
  if (EPI == EPI_NOISE) nscale = noise_row_scale<LPR, NB>(ep, lr.row, l);
#pragma unroll
  for (int b = 0; b < NB; ++b) {
    const int off = (b * LPR + l) * 4;
    const int64_t p = lr.slot_begin * d + off;
    float4 sq[FIX_WAYS];
#pragma unroll
    for (int q = 0; q < FIX_WAYS; ++q) sq[q] = q < lr.n_seg ? ld(p + (int64_t)q * d) : make_float4(0.f, 0.f, 0.f, 0.f);
    int j = FIX_WAYS;
    for (; j + FIX_WAYS <= lr.n_seg; j += FIX_WAYS) {
      float4 x[FIX_WAYS];
#pragma unroll
      for (int q = 0; q < FIX_WAYS; ++q) x[q] = ld(p + (int64_t)(j + q) * d);
#pragma unroll
      for (int q = 0; q < FIX_WAYS; ++q) sq[q] = add4(sq[q], x[q]);
    }
#pragma unroll
    for (int q = 0; q < FIX_WAYS; ++q)
      if (j + q < lr.n_seg) sq[q] = add4(sq[q], ld(p + (int64_t)(j + q) * d));
    float4 row = sq[0];
#pragma unroll
    for (int q = 1; q < FIX_WAYS; ++q)
      if (q < lr.n_seg) row = add4(row, sq[q]);
    if (EPI == EPI_NOISE) row = perturb(ep, lr.row, b * LPR + l, nscale, row);
    epilogue_store<EPI>(ep, lr.row, off, row);
  }
}

// One virtual row: the sequential walk over its staged entries, then either the epilogue (whole
// row) or the partial store + combine protocol (segment of a split row).
__device__ __forceinline__ bool is_local(int tgt) { return tgt < LOCAL_CODE + LSLOTS; }

// Wave lane of this lane group's first lane, rebuilt from the hardware lane counter where it is used (two
// VALU ops per vrow) instead of living in a register across the gather loop.
template <int LPR>
__device__ __forceinline__ int group_leader() {
  int lane;
  asm volatile("v_mbcnt_lo_u32_b32 %0, -1, 0\n\tv_mbcnt_hi_u32_b32 %0, -1, %0" : "=v"(lane));
  return lane & ~(LPR - 1);
}

template <int LPR, int NB, int UNROLL, int EPI, bool FUSED, bool XM = false>
__device__ __forceinline__ void do_vrow(const ColVal* s_cv, int s, int e, int tgt, int l, const float* __restrict__ X,
                                        int64_t ldx, float* __restrict__ partials, int64_t d, const Epilogue& ep,
                                        const FixCtx& fx, float4* s_part, const uint32_t* __restrict__ xm = nullptr) {
  float nscale = 0.f;
  if (EPI == EPI_NOISE && tgt >= 0) nscale = noise_row_scale<LPR, NB>(ep, tgt, l);
#pragma unroll
  for (int b = 0; b < NB; ++b) {
    float4 acc = make_float4(0.f, 0.f, 0.f, 0.f);
    if (XM) acc = walk_masked<UNROLL>(s_cv, s, e, X + (b * LPR + l) * 4, ldx, xm, acc);  // (XM: rows of X outside xm are zero and not read)
    else acc = walk<UNROLL>(s_cv, s, e, X + (b * LPR + l) * 4, ldx, acc);
    // the store addresses are rebuilt from the lane id once per vrow: hoisting them out of the vrow loop as
    // 64-bit per-lane pairs costs the registers that keep the kernel at 64 VGPRs (8 waves/SIMD)
    int lo = l;
    asm volatile("" : "+v"(lo));
    const int off = (b * LPR + lo) * 4;
    if (tgt >= 0) {
      if (EPI == EPI_NOISE) acc = perturb(ep, tgt, b * LPR + l, nscale, acc);
      epilogue_store<EPI>(ep, tgt, off, acc);
    } else if (is_local(tgt)) {
      s_part[(tgt - LOCAL_CODE) * (NB * LPR) + b * LPR + lo] = acc;
    } else if (FUSED) {
      const __amdgpu_buffer_rsrc_t rsrc = __builtin_amdgcn_make_buffer_rsrc(partials, 0, (int)fx.part_bytes, 0x00020000);
      u32x4 u;
      u.x = __float_as_uint(acc.x), u.y = __float_as_uint(acc.y), u.z = __float_as_uint(acc.z), u.w = __float_as_uint(acc.w);
      __builtin_amdgcn_raw_buffer_store_b128(u, rsrc, (unsigned)(((int64_t)(~tgt) * d + off) * 4), 0, 16);  // aux 16 = sc1
    } else {
      *reinterpret_cast<float4*>(partials + (int64_t)(~tgt) * d + off) = acc;
    }
  }
  if (FUSED && tgt < 0 && !is_local(tgt)) combine_if_last<LPR, NB, EPI>(ep, partials, d, fx, ~tgt, l);
}

// After a tile's vrows are done (and a barrier): lane group j adds the LDS partials of the tile's
// j-th local row in the published 4-way order (n_seg <= 8: s_q = p_q [+ p_{q+4}], then s_0 + s_1 + s_2 + s_3)
// and either runs the epilogue (whole row) or hands the sum on as the global partial of a chunk.
template <int LPR, int NB, int EPI, bool FUSED>
__device__ __forceinline__ void combine_local(const Tile& t, const LocalRow* __restrict__ locals, const float4* s_part,
                                              int g, int l, float* __restrict__ partials, int64_t d, const Epilogue& ep,
                                              const FixCtx& fx, const uint32_t* __restrict__ out_mask,
                                              const int32_t* __restrict__ slot_row) {
  constexpr int GROUPS = BLOCK / LPR;
  constexpr int W = NB * LPR;
  for (int j = g; j < t.n_local; j += GROUPS) {
    const LocalRow lr = locals[t.local_begin + j];
    if (out_mask && !mask_bit(out_mask, lr.tgt >= 0 ? lr.tgt : slot_row[~lr.tgt])) continue;  // row not requested
    float nscale = 0.f;
    if (EPI == EPI_NOISE && lr.tgt >= 0) nscale = noise_row_scale<LPR, NB>(ep, lr.tgt, l);
#pragma unroll
    for (int b = 0; b < NB; ++b) {
      const int off = (b * LPR + l) * 4;
      const float4* p = s_part + lr.lslot * W + b * LPR + l;
      float4 sq[FIX_WAYS];
#pragma unroll
      for (int q = 0; q < FIX_WAYS; ++q) sq[q] = q < lr.n_seg ? p[q * W] : make_float4(0.f, 0.f, 0.f, 0.f);
#pragma unroll
      for (int q = 0; q < FIX_WAYS; ++q)
        if (q + FIX_WAYS < lr.n_seg) sq[q] = add4(sq[q], p[(q + FIX_WAYS) * W]);
      float4 row = sq[0];
#pragma unroll
      for (int q = 1; q < FIX_WAYS; ++q)
        if (q < lr.n_seg) row = add4(row, sq[q]);
      if (lr.tgt >= 0) {
        if (EPI == EPI_NOISE) row = perturb(ep, lr.tgt, b * LPR + l, nscale, row);
        epilogue_store<EPI>(ep, lr.tgt, off, row);
      } else if (FUSED) {
        const __amdgpu_buffer_rsrc_t rsrc = __builtin_amdgcn_make_buffer_rsrc(partials, 0, (int)fx.part_bytes, 0x00020000);
        u32x4 u;
        u.x = __float_as_uint(row.x), u.y = __float_as_uint(row.y), u.z = __float_as_uint(row.z), u.w = __float_as_uint(row.w);
        __builtin_amdgcn_raw_buffer_store_b128(u, rsrc, (unsigned)(((int64_t)(~lr.tgt) * d + off) * 4), 0, 16);  // aux 16 = sc1
      } else {
        *reinterpret_cast<float4*>(partials + (int64_t)(~lr.tgt) * d + off) = row;
      }
    }
    if (FUSED && lr.tgt < 0) combine_if_last<LPR, NB, EPI>(ep, partials, d, fx, ~lr.tgt, l);
  }
}

// One workgroup per tile.  LPR lanes per vrow, each lane owns 4 consecutive features of
// every feature block of width 4*LPR (d = NB * 4 * LPR).  DYNAMIC: lane groups draw the next
// vrow from an LDS counter instead of a fixed stride (evens out skewed row lengths).
template <int LPR, int NB, int UNROLL, bool DYNAMIC, int MINW = 1, int EPI = EPI_PLAIN, bool FUSED = false>
__global__ __launch_bounds__(BLOCK, MINW) void spmm_tile_kernel(const Tile* __restrict__ tiles,
                                                          const int64_t* __restrict__ vptr,
                                                          const int32_t* __restrict__ vtgt,
                                                          const ColVal* __restrict__ cv,
                                                          const float* __restrict__ X, int64_t ldx,
                                                          float* __restrict__ partials, int64_t d,
                                                          Epilogue ep, FixCtx fx,
                                                          const LocalRow* __restrict__ locals) {
  __shared__ ColVal s_cv[TILE_NNZ];
  __shared__ float4 s_part[LSLOTS * NB * LPR];
  __shared__ int s_ptr[TILE_VROWS + 1];
  __shared__ int s_tgt[TILE_VROWS];
  __shared__ int s_next;

  const Tile t = tiles[blockIdx.x];
  const int tid = threadIdx.x;
  const int nv = t.n_vrows;
  const int64_t nz0 = t.nnz_begin;
  constexpr int GROUPS = BLOCK / LPR;
  if (DYNAMIC && tid == 0) s_next = GROUPS;
  for (int i = tid; i <= nv; i += BLOCK) s_ptr[i] = (int)(vptr[t.vrow_begin + i] - nz0);
  for (int i = tid; i < nv; i += BLOCK) s_tgt[i] = vtgt[t.vrow_begin + i];
  const int cnt = t.nnz_count;
  {
    const ColVal* src = cv + nz0;
    for (int i = tid; i < cnt; i += BLOCK) s_cv[i] = src[i];
  }
  __syncthreads();

  const int g = tid / LPR;
  const int l = tid % LPR;
  int v = g;
  while (v < nv) {
    do_vrow<LPR, NB, UNROLL, EPI, FUSED>(s_cv, s_ptr[v], s_ptr[v + 1], s_tgt[v], l, X, ldx, partials, d, ep, fx, s_part);
    if (DYNAMIC) {
      int nxt = 0;
      if (l == 0) nxt = atomicAdd(&s_next, 1);
      v = __builtin_amdgcn_ds_bpermute(group_leader<LPR>() << 2, nxt);
    } else {
      v += GROUPS;
    }
  }
  if (t.n_local > 0) {  // block-uniform
    __syncthreads();
    combine_local<LPR, NB, EPI, FUSED>(t, locals, s_part, g, l, partials, d, ep, fx, nullptr, nullptr);
  }
}

// Sparse-input form of the tile kernel: the gathered panel has few live rows (bitmap x_mask).
// The staged entry list is flagged in parallel (one bitmap probe per entry, no dependent chains),
// prefix-scanned and compacted in LDS, the vrow pointers are remapped, and the ordinary sequential
// walk then runs over the live entries only.  Dropping a dead entry is exact (fmaf(v, +0, acc) ==
// acc), the survivors keep their order, so results are bit-identical to the dense form.
template <int LPR, int NB, bool FUSED, int EPI = EPI_PLAIN>
__global__ __launch_bounds__(BLOCK) void spmm_tile_sparse_kernel(const Tile* __restrict__ tiles,
                                                                 const int64_t* __restrict__ vptr,
                                                                 const int32_t* __restrict__ vtgt,
                                                                 const ColVal* __restrict__ cv,
                                                                 const float* __restrict__ X, int64_t ldx,
                                                                 float* __restrict__ partials, int64_t d, Epilogue ep,
                                                                 FixCtx fx, const uint32_t* __restrict__ x_mask,
                                                                 const LocalRow* __restrict__ locals) {
  __shared__ ColVal s_cv[TILE_NNZ];
  __shared__ float4 s_part[LSLOTS * NB * LPR];
  __shared__ int s_pre[TILE_NNZ + 1];  // s_pre[i] = live entries among [0, i)
  __shared__ int s_ptr[TILE_VROWS + 1];
  __shared__ int s_tgt[TILE_VROWS];
  __shared__ int s_wave[(TILE_NNZ / BLOCK) * (BLOCK / 64)];
  __shared__ int s_next;

  const Tile t = tiles[blockIdx.x];
  const int tid = threadIdx.x;
  const int lane = tid % 64, wave = tid / 64;
  const int nv = t.n_vrows;
  const int64_t nz0 = t.nnz_begin;
  constexpr int GROUPS = BLOCK / LPR;
  if (tid == 0) s_next = GROUPS;
  for (int i = tid; i <= nv; i += BLOCK) s_ptr[i] = (int)(vptr[t.vrow_begin + i] - nz0);
  for (int i = tid; i < nv; i += BLOCK) s_tgt[i] = vtgt[t.vrow_begin + i];
  const int cnt = t.nnz_count;
  const ColVal* src = cv + nz0;
  // flag + ordered compaction.  Every thread takes its (up to PER) entries i = k * BLOCK + tid at once: all entry loads
  // go out together, then all bitmap probes, and ONE barrier pair serves the whole tile (a round per BLOCK entries
  // cost two dependent loads and two barriers each).  Position of a live entry = live entries with a smaller index.
  constexpr int PER = TILE_NNZ / BLOCK;
  ColVal e[PER];
  bool live[PER];
#pragma unroll
  for (int k = 0; k < PER; ++k) {
    const int i = k * BLOCK + tid;
    e[k] = ColVal{};
    if (i < cnt) e[k] = src[i];
  }
  int in_wave[PER];
#pragma unroll
  for (int k = 0; k < PER; ++k) {
    const int i = k * BLOCK + tid;
    live[k] = i < cnt && mask_bit(x_mask, e[k].col);
    const unsigned long long m = __ballot(live[k]);
    in_wave[k] = __popcll(m & ((1ull << lane) - 1ull));
    if (lane == 0) s_wave[k * (BLOCK / 64) + wave] = __popcll(m);
  }
  __syncthreads();
  {
    int before = 0;
#pragma unroll
    for (int k = 0; k < PER; ++k) {
      const int i = k * BLOCK + tid;
      int mine = before;
      for (int w = 0; w < BLOCK / 64; ++w) {
        const int c = s_wave[k * (BLOCK / 64) + w];
        if (w < wave) mine += c;
        before += c;
      }
      if (i < cnt) s_pre[i] = mine + in_wave[k];
      if (live[k]) s_cv[mine + in_wave[k]] = e[k];
    }
    if (tid == 0) s_pre[cnt] = before;
  }
  __syncthreads();

  const int g = tid / LPR;
  const int l = tid % LPR;
  int v = g;
  while (v < nv) {
    do_vrow<LPR, NB, IDG_ROWS_UNROLL, EPI, FUSED>(s_cv, s_pre[s_ptr[v]], s_pre[s_ptr[v + 1]], s_tgt[v], l, X, ldx, partials, d, ep, fx,
                                      s_part);
    int nxt = 0;
    if (l == 0) nxt = atomicAdd(&s_next, 1);
    v = __builtin_amdgcn_ds_bpermute(group_leader<LPR>() << 2, nxt);
  }
  if (t.n_local > 0) {
    __syncthreads();
    combine_local<LPR, NB, EPI, FUSED>(t, locals, s_part, g, l, partials, d, ep, fx, nullptr, nullptr);
  }
}

// ---- the sparse-input form with its index-only half done AHEAD of time (round 4) -------------------------------------
// Which entries of a tile point at live rows of the gathered panel depends on the bitmap only, not on the panel: the
// flag / scan / compaction above is index-only work.  compact_inputs_kernel does it once per bitmap (on the caller's
// side stream, with the batch's other index-only work, while the previous step is still running) and leaves, per tile,
// the live entries in their original order at the tile's own offset of a second entry list, the compacted start of
// every vrow and the tile's live count.  The product then stages those lists and runs the ORDINARY walk over them
// (spmm_tile_compact_kernel): no dead entry is loaded, flagged or scanned on the critical path.  Dropping a dead entry
// is exact and the survivors keep their order: bit-identical to the dense form and to spmm_tile_sparse_kernel.
__global__ __launch_bounds__(BLOCK) void compact_inputs_kernel(const Tile* __restrict__ tiles, const int64_t* __restrict__ vptr,
                                                               const ColVal* __restrict__ cv,
                                                               const uint32_t* __restrict__ x_mask, ColVal* __restrict__ ccv,
                                                               int32_t* __restrict__ cptr, int32_t* __restrict__ clive) {
  __shared__ int s_pre[TILE_NNZ + 1];
  __shared__ int s_wave[(TILE_NNZ / BLOCK) * (BLOCK / 64)];
  const Tile t = tiles[blockIdx.x];
  const int tid = threadIdx.x;
  const int lane = tid % 64, wave = tid / 64;
  const int nv = t.n_vrows;
  const int64_t nz0 = t.nnz_begin;
  const int cnt = t.nnz_count;
  const ColVal* src = cv + nz0;
  constexpr int PER = TILE_NNZ / BLOCK;
  ColVal e[PER];
  bool live[PER];
  int in_wave[PER];
#pragma unroll
  for (int k = 0; k < PER; ++k) {
    const int i = k * BLOCK + tid;
    e[k] = ColVal{};
    if (i < cnt) e[k] = src[i];
  }
#pragma unroll
  for (int k = 0; k < PER; ++k) {
    const int i = k * BLOCK + tid;
    live[k] = i < cnt && mask_bit(x_mask, e[k].col);
    const unsigned long long m = __ballot(live[k]);
    in_wave[k] = __popcll(m & ((1ull << lane) - 1ull));
    if (lane == 0) s_wave[k * (BLOCK / 64) + wave] = __popcll(m);
  }
  __syncthreads();
  int before = 0;
#pragma unroll
  for (int k = 0; k < PER; ++k) {
    const int i = k * BLOCK + tid;
    int mine = before;
    for (int w = 0; w < BLOCK / 64; ++w) {
      const int c = s_wave[k * (BLOCK / 64) + w];
      if (w < wave) mine += c;
      before += c;
    }
    if (i < cnt) s_pre[i] = mine + in_wave[k];
    if (live[k]) ccv[nz0 + mine + in_wave[k]] = e[k];
  }
  if (tid == 0) {
    s_pre[cnt] = before;
    clive[t.vrow_begin] = before;
  }
  __syncthreads();
  for (int i = tid; i < nv; i += BLOCK) cptr[t.vrow_begin + i] = s_pre[(int)(vptr[t.vrow_begin + i] - nz0)];
}

template <int LPR, int NB, bool FUSED>
__global__ __launch_bounds__(BLOCK) void spmm_tile_compact_kernel(const Tile* __restrict__ tiles,
                                                                  const int32_t* __restrict__ vtgt,
                                                                  const ColVal* __restrict__ ccv,
                                                                  const int32_t* __restrict__ cptr,
                                                                  const int32_t* __restrict__ clive,
                                                                  const float* __restrict__ X, int64_t ldx,
                                                                  float* __restrict__ partials, int64_t d, Epilogue ep,
                                                                  FixCtx fx, const LocalRow* __restrict__ locals) {
  __shared__ ColVal s_cv[TILE_NNZ];
  __shared__ float4 s_part[LSLOTS * NB * LPR];
  __shared__ int s_ptr[TILE_VROWS + 1];
  __shared__ int s_tgt[TILE_VROWS];
  __shared__ int s_next;
  const Tile t = tiles[blockIdx.x];
  const int tid = threadIdx.x;
  const int nv = t.n_vrows;
  constexpr int GROUPS = BLOCK / LPR;
  if (tid == 0) s_next = GROUPS;
  const int cnt = clive[t.vrow_begin];
  for (int i = tid; i < nv; i += BLOCK) s_ptr[i] = cptr[t.vrow_begin + i];
  if (tid == 0) s_ptr[nv] = cnt;
  for (int i = tid; i < nv; i += BLOCK) s_tgt[i] = vtgt[t.vrow_begin + i];
  {
    const ColVal* src = ccv + t.nnz_begin;
    for (int i = tid; i < cnt; i += BLOCK) s_cv[i] = src[i];
  }
  __syncthreads();
  const int g = tid / LPR;
  const int l = tid % LPR;
  int v = g;
  while (v < nv) {
    // (a vrow without a live entry still runs its epilogue: it owns an output row, or a partial of one)
    do_vrow<LPR, NB, IDG_ROWS_UNROLL, EPI_PLAIN, FUSED>(s_cv, s_ptr[v], s_ptr[v + 1], s_tgt[v], l, X, ldx, partials, d, ep, fx, s_part);
    int nxt = 0;
    if (l == 0) nxt = atomicAdd(&s_next, 1);
    v = __builtin_amdgcn_ds_bpermute(group_leader<LPR>() << 2, nxt);
  }
  if (t.n_local > 0) {
    __syncthreads();
    combine_local<LPR, NB, EPI_PLAIN, FUSED>(t, locals, s_part, g, l, partials, d, ep, fx, nullptr, nullptr);
  }
}

// Output-row-restricted form: only rows flagged in `out_mask` are produced (the last forward layer
// of a training step feeds nothing but the layer mean at the <= 3B rows of the batch).  A tile
// without a flagged row exits before staging its entries; otherwise the flagged vrows run the
// ordinary walk, so the produced rows are bit-identical to the full product.
template <int LPR, int NB, bool FUSED, int EPI = EPI_PLAIN, bool XM = false>
__global__ __launch_bounds__(BLOCK) void spmm_tile_rows_kernel(const Tile* __restrict__ tiles,
                                                               const int64_t* __restrict__ vptr,
                                                               const int32_t* __restrict__ vtgt,
                                                               const int32_t* __restrict__ slot_row,
                                                               const ColVal* __restrict__ cv,
                                                               const float* __restrict__ X, int64_t ldx,
                                                               float* __restrict__ partials, int64_t d, Epilogue ep,
                                                               FixCtx fx, const uint32_t* __restrict__ out_mask,
                                                               const LocalRow* __restrict__ locals,
                                                               const uint32_t* __restrict__ x_mask) {
  __shared__ ColVal s_cv[TILE_NNZ];
  __shared__ float4 s_part[LSLOTS * NB * LPR];
  __shared__ int s_ptr[TILE_VROWS + 1];
  __shared__ int s_tgt[TILE_VROWS];
  __shared__ int s_live[TILE_VROWS];  // compacted list of flagged vrows
  __shared__ int s_nlive;
  __shared__ int s_next;

  const Tile t = tiles[blockIdx.x];
  {
    // a tile covers consecutive rows: a few (uniform) bitmap words decide whether anything here is requested
    uint32_t any = 0;
    for (int w = t.row_first >> 5; w <= (t.row_last >> 5); ++w) {
      uint32_t m = out_mask[w];
      if (w == (t.row_first >> 5)) m &= ~0u << (t.row_first & 31);
      if (w == (t.row_last >> 5)) m &= ~0u >> (31 - (t.row_last & 31));
      any |= m;
    }
    if (any == 0) return;
  }
  const int tid = threadIdx.x;
  const int nv = t.n_vrows;
  const int64_t nz0 = t.nnz_begin;
  constexpr int GROUPS = BLOCK / LPR;
  if (tid == 0) {
    s_nlive = 0;
    s_next = GROUPS;
  }
  __syncthreads();
  for (int i = tid; i <= nv; i += BLOCK) s_ptr[i] = (int)(vptr[t.vrow_begin + i] - nz0);
  for (int i = tid; i < nv; i += BLOCK) {
    const int tg = vtgt[t.vrow_begin + i];
    s_tgt[i] = tg;
    if (is_local(tg)) continue;  // segments of a local row are flagged through the row, below
    const int row = tg >= 0 ? tg : slot_row[~tg];
    if (mask_bit(out_mask, row)) s_live[atomicAdd(&s_nlive, 1)] = i;  // order inside a tile is irrelevant
  }
  for (int j = tid; j < t.n_local; j += BLOCK) {
    const LocalRow lr = locals[t.local_begin + j];
    if (!mask_bit(out_mask, lr.tgt >= 0 ? lr.tgt : slot_row[~lr.tgt])) continue;
    const int at = atomicAdd(&s_nlive, (int)lr.n_seg);
    for (int q = 0; q < lr.n_seg; ++q) s_live[at + q] = lr.vrow - t.vrow_begin + q;
  }
  {  // the tile's entries are fetched beside the vrow tables (Tile::nnz_count): one dependent level and one barrier less
    const ColVal* src = cv + nz0;
    const int cnt = t.nnz_count;
    for (int i = tid; i < cnt; i += BLOCK) s_cv[i] = src[i];
  }
  __syncthreads();
  const int nlive = s_nlive;
  if (nlive == 0) return;

  const int g = tid / LPR;
  const int l = tid % LPR;
  int q = g;
  while (q < nlive) {
    const int v = s_live[q];
    do_vrow<LPR, NB, IDG_ROWS_UNROLL, EPI, FUSED, XM>(s_cv, s_ptr[v], s_ptr[v + 1], s_tgt[v], l, X, ldx, partials, d, ep, fx, s_part,
                                                      x_mask);
    int nxt = 0;
    if (l == 0) nxt = atomicAdd(&s_next, 1);
    q = __builtin_amdgcn_ds_bpermute(group_leader<LPR>() << 2, nxt);
  }
  if (t.n_local > 0) {
    __syncthreads();
    combine_local<LPR, NB, EPI, FUSED>(t, locals, s_part, g, l, partials, d, ep, fx, out_mask, slot_row);
  }
}

// The row-restricted product for several panels at once (SimGCL's last forward layer: the clean pass and its
// perturbed views all want the same rows of A.X for their own X).  The restricted kernel is bound by what surrounds
// the walks — tile / pointer / bitmap loads, compaction, staging, barriers — and all of that is shared here: only the
// walks (and the epilogues) run once per panel.  Panel 0 has epilogue EPI0, the others perturb (EPI_NOISE).
// Chunked rows use the last-arriver combine with one partial region and one ticket array per panel.
constexpr int MAX_PANELS = 3;
struct MultiPanel {
  const float* X[MAX_PANELS];
  float* partials[MAX_PANELS];
  int* cnt[MAX_PANELS];
  Epilogue ep[MAX_PANELS];
};

template <int LPR, int NB, int NP, int EPI0>
__global__ __launch_bounds__(BLOCK) void spmm_tile_rows_multi_kernel(const Tile* __restrict__ tiles,
                                                                     const int64_t* __restrict__ vptr,
                                                                     const int32_t* __restrict__ vtgt,
                                                                     const int32_t* __restrict__ slot_row,
                                                                     const ColVal* __restrict__ cv, MultiPanel mp,
                                                                     int64_t ldx, int64_t d, FixCtx fx,
                                                                     const uint32_t* __restrict__ out_mask,
                                                                     const LocalRow* __restrict__ locals) {
  constexpr int SLAB = LSLOTS * NB * LPR;
  __shared__ ColVal s_cv[TILE_NNZ];
  __shared__ float4 s_part[NP * SLAB];
  __shared__ int s_ptr[TILE_VROWS + 1];
  __shared__ int s_tgt[TILE_VROWS];
  __shared__ int s_live[TILE_VROWS];
  __shared__ int s_nlive;
  __shared__ int s_next;

  const Tile t = tiles[blockIdx.x];
  {
    uint32_t any = 0;
    for (int w = t.row_first >> 5; w <= (t.row_last >> 5); ++w) {
      uint32_t m = out_mask[w];
      if (w == (t.row_first >> 5)) m &= ~0u << (t.row_first & 31);
      if (w == (t.row_last >> 5)) m &= ~0u >> (31 - (t.row_last & 31));
      any |= m;
    }
    if (any == 0) return;
  }
  const int tid = threadIdx.x;
  const int nv = t.n_vrows;
  const int64_t nz0 = t.nnz_begin;
  constexpr int GROUPS = BLOCK / LPR;
  if (tid == 0) {
    s_nlive = 0;
    s_next = GROUPS;
  }
  __syncthreads();
  for (int i = tid; i <= nv; i += BLOCK) s_ptr[i] = (int)(vptr[t.vrow_begin + i] - nz0);
  for (int i = tid; i < nv; i += BLOCK) {
    const int tg = vtgt[t.vrow_begin + i];
    s_tgt[i] = tg;
    if (is_local(tg)) continue;
    const int row = tg >= 0 ? tg : slot_row[~tg];
    if (mask_bit(out_mask, row)) s_live[atomicAdd(&s_nlive, 1)] = i;
  }
  for (int j = tid; j < t.n_local; j += BLOCK) {
    const LocalRow lr = locals[t.local_begin + j];
    if (!mask_bit(out_mask, lr.tgt >= 0 ? lr.tgt : slot_row[~lr.tgt])) continue;
    const int at = atomicAdd(&s_nlive, (int)lr.n_seg);
    for (int q = 0; q < lr.n_seg; ++q) s_live[at + q] = lr.vrow - t.vrow_begin + q;
  }
  {  // as in spmm_tile_rows_kernel: entries fetched beside the vrow tables
    const ColVal* src = cv + nz0;
    const int cnt = t.nnz_count;
    for (int i = tid; i < cnt; i += BLOCK) s_cv[i] = src[i];
  }
  __syncthreads();
  const int nlive = s_nlive;
  if (nlive == 0) return;

  const int g = tid / LPR;
  const int l = tid % LPR;
  int q = g;
  while (q < nlive) {
    const int v = s_live[q];
    const int vs = s_ptr[v], ve = s_ptr[v + 1], tg = s_tgt[v];
#pragma unroll
    for (int p = 0; p < NP; ++p) {
      FixCtx f = fx;
      f.cnt = mp.cnt[p];
      if (p == 0)
        do_vrow<LPR, NB, IDG_ROWS_UNROLL, EPI0, true>(s_cv, vs, ve, tg, l, mp.X[p], ldx, mp.partials[p], d, mp.ep[p], f, s_part + p * SLAB);
      else
        do_vrow<LPR, NB, IDG_ROWS_UNROLL, EPI_NOISE, true>(s_cv, vs, ve, tg, l, mp.X[p], ldx, mp.partials[p], d, mp.ep[p], f,
                                             s_part + p * SLAB);
    }
    int nxt = 0;
    if (l == 0) nxt = atomicAdd(&s_next, 1);
    q = __builtin_amdgcn_ds_bpermute(group_leader<LPR>() << 2, nxt);
  }
  if (t.n_local > 0) {
    __syncthreads();
#pragma unroll
    for (int p = 0; p < NP; ++p) {
      FixCtx f = fx;
      f.cnt = mp.cnt[p];
      if (p == 0)
        combine_local<LPR, NB, EPI0, true>(t, locals, s_part + p * SLAB, g, l, mp.partials[p], d, mp.ep[p], f, out_mask, slot_row);
      else
        combine_local<LPR, NB, EPI_NOISE, true>(t, locals, s_part + p * SLAB, g, l, mp.partials[p], d, mp.ep[p], f, out_mask,
                                                slot_row);
    }
  }
}

// ---- row-restricted product over a LIST of live work units (round 2) ------------------------------------------
// The tile form above visits every tile to find the few vrows of the batch's rows and stages a whole tile for one or
// two of them.  With the batch's row bitmap turned into a list of work units ahead of time (idg_graph_live_units, on
// the side stream with the rest of the batch's index-only work) the launch has one WAVE per unit and nothing else:
//   * a plain vrow: lane group 0 walks its entries straight from global memory;
//   * a split row whose segments the tile form combines in LDS (a LocalRow; also one chunk of a chunked row): the
//     wave's lane groups take the segments round-robin, partials go to a wave-private LDS slab and lane group 0 adds
//     them in the published 4-way order — the same operations on the same operands as combine_local;
//   * chunks hand their sums to the last-arriver combine exactly as in the tile form.
// Same fmaf chains, same combine orders: bit-identical to the tile form (and to the full product on those rows).
__global__ __launch_bounds__(BLOCK) void live_units_kernel(const uint32_t* __restrict__ bitmap, int64_t n_rows,
                                                           const int32_t* __restrict__ row_unit,
                                                           const LongRow* __restrict__ longs,
                                                           const int32_t* __restrict__ slot_unit, int32_t* __restrict__ out,
                                                           int64_t cap) {
  // out[0] = number of units, out[1] = 1 when more units were found than the list holds (the caller's bound on the set
  // bits was wrong: the consumer poisons its output and idg_graph_live_units_check reports it), out[2..] = the units
  // (order irrelevant).  One bitmap word per thread; a wave reserves the room for all its units with ONE atomic (a batch's
  // ~3,000 rows used to be ~3,000 serialised atomics on out[0]: 18 us at yelp2018 size).
  const int64_t w = (int64_t)blockIdx.x * BLOCK + threadIdx.x;
  const int lane = threadIdx.x % 64;
  uint32_t m = w * 32 < n_rows ? bitmap[w] : 0u;
  if (w * 32 + 32 > n_rows && w * 32 < n_rows) m &= ~0u >> (32 - (n_rows - w * 32));  // bits past the last row
  int mine = 0;
  for (uint32_t mm = m; mm; mm &= mm - 1) {
    const int32_t u = row_unit[w * 32 + __builtin_ctz(mm)];
    mine += u > UNIT_LONG / 2 ? 1 : longs[u - UNIT_LONG].n_seg;
  }
  int before = mine;  // inclusive prefix over the wave
#pragma unroll
  for (int o = 1; o < 64; o <<= 1) {
    const int v = __shfl_up(before, o, 64);
    if (lane >= o) before += v;
  }
  const int total = __shfl(before, 63, 64);
  if (total == 0) return;
  int base = 0;
  if (lane == 63) base = atomicAdd(out, total);
  base = __shfl(base, 63, 64);
  int pos = base + before - mine;
  for (; m; m &= m - 1) {
    const int32_t u = row_unit[w * 32 + __builtin_ctz(m)];
    if (u > UNIT_LONG / 2) {  // a vrow (>= 0) or ~LocalRow
      if (pos < cap) out[UNITS_HEADER + pos] = u;
      else out[1] = 1;
      ++pos;
    } else {
      const LongRow lr = longs[u - UNIT_LONG];
      for (int k = 0; k < lr.n_seg; ++k, ++pos)
        if (pos < cap) out[UNITS_HEADER + pos] = slot_unit[lr.slot_begin + k];
        else out[1] = 1;
    }
  }
}

template <int LPR, int NB, int EPI, bool XM = false>
__global__ __launch_bounds__(IDG_UNITS_BLOCK) void spmm_units_kernel(const int32_t* __restrict__ units, int64_t cap,
                                                           const int64_t* __restrict__ vptr,
                                                           const int32_t* __restrict__ vtgt,
                                                           const ColVal* __restrict__ cv, const float* __restrict__ X,
                                                           int64_t ldx, float* __restrict__ partials, int64_t d, Epilogue ep,
                                                           FixCtx fx, const LocalRow* __restrict__ locals,
                                                           const uint32_t* __restrict__ x_mask) {
  constexpr int GPW = 64 / LPR;  // lane groups per wave
  constexpr int W = NB * LPR;
  __shared__ float4 s_part[IDG_UNITS_BLOCK / 64][LSLOTS * W];
  const int wave = threadIdx.x / 64, lane = threadIdx.x % 64;
  const int64_t u_idx = (int64_t)blockIdx.x * (IDG_UNITS_BLOCK / 64) + wave;
  int64_t count = units[0];
  if (count > cap) count = cap;
  if (u_idx >= count) return;  // whole waves leave together
  const int32_t unit = units[UNITS_HEADER + u_idx];
  // the list overflowed when it was built (more set bits than the caller's bound): rows are missing from it, so what
  // this launch can produce is incomplete — every row it does produce is poisoned instead of passing for a result
  const bool poison = units[1] != 0;
  const int g = lane / LPR, l = lane % LPR;
  if (unit >= 0) {  // a plain vrow: one lane group.  Its target is a row, or (a short last chunk) a global partial slot
    if (g != 0) return;
    const int64_t s = vptr[unit], e = vptr[unit + 1];
    const int tgt = vtgt[unit];
    float nscale = 0.f;
    if (EPI == EPI_NOISE && tgt >= 0) nscale = noise_row_scale<LPR, NB>(ep, tgt, l);
#pragma unroll
    for (int b = 0; b < NB; ++b) {
      const int off = (b * LPR + l) * 4;
      float4 acc = make_float4(0.f, 0.f, 0.f, 0.f);
      for (int64_t c = s; c < e; c += (1 << 20)) {
        const int len = (int)((e - c) < (1 << 20) ? (e - c) : (1 << 20));
        acc = XM ? walk_masked<IDG_UNITS_UNROLL>(cv + c, 0, len, X + off, ldx, x_mask, acc)
                 : walk_global<IDG_UNITS_UNROLL>(cv + c, len, X + off, ldx, acc);
      }
      if (poison) acc = make_float4(__builtin_nanf(""), __builtin_nanf(""), __builtin_nanf(""), __builtin_nanf(""));
      if (tgt >= 0) {
        if (EPI == EPI_NOISE) acc = perturb(ep, tgt, b * LPR + l, nscale, acc);
        epilogue_store<EPI>(ep, tgt, off, acc);
      } else {
        const __amdgpu_buffer_rsrc_t rsrc = __builtin_amdgcn_make_buffer_rsrc(partials, 0, (int)fx.part_bytes, 0x00020000);
        u32x4 u;
        u.x = __float_as_uint(acc.x), u.y = __float_as_uint(acc.y), u.z = __float_as_uint(acc.z), u.w = __float_as_uint(acc.w);
        __builtin_amdgcn_raw_buffer_store_b128(u, rsrc, (unsigned)(((int64_t)(~tgt) * d + off) * 4), 0, 16);  // aux 16 = sc1
      }
    }
    if (tgt < 0) combine_if_last<LPR, NB, EPI>(ep, partials, d, fx, ~tgt, l);
    return;
  }
  const LocalRow lr = locals[~unit];
  float4* part = s_part[wave];
  for (int q = g; q < lr.n_seg; q += GPW) {  // segments round-robin over the wave's lane groups
    const int64_t s = vptr[lr.vrow + q], e = vptr[lr.vrow + q + 1];
#pragma unroll
    for (int b = 0; b < NB; ++b) {
      float4 acc = make_float4(0.f, 0.f, 0.f, 0.f);
      acc = XM ? walk_masked<IDG_UNITS_UNROLL>(cv + s, 0, (int)(e - s), X + (b * LPR + l) * 4, ldx, x_mask, acc)
               : walk_global<IDG_UNITS_UNROLL>(cv + s, (int)(e - s), X + (b * LPR + l) * 4, ldx, acc);
      part[q * W + b * LPR + l] = acc;
    }
  }
  __builtin_amdgcn_fence(__ATOMIC_RELEASE, "wavefront");
  __builtin_amdgcn_wave_barrier();
  __builtin_amdgcn_fence(__ATOMIC_ACQUIRE, "wavefront");
  if (g != 0) return;
  float nscale = 0.f;
  if (EPI == EPI_NOISE && lr.tgt >= 0) nscale = noise_row_scale<LPR, NB>(ep, lr.tgt, l);
#pragma unroll
  for (int b = 0; b < NB; ++b) {
    const int off = (b * LPR + l) * 4;
    const float4* p = part + b * LPR + l;
    float4 sq[FIX_WAYS];
#pragma unroll
    for (int q = 0; q < FIX_WAYS; ++q) sq[q] = q < lr.n_seg ? p[q * W] : make_float4(0.f, 0.f, 0.f, 0.f);
#pragma unroll
    for (int q = 0; q < FIX_WAYS; ++q)
      if (q + FIX_WAYS < lr.n_seg) sq[q] = add4(sq[q], p[(q + FIX_WAYS) * W]);
    float4 row = sq[0];
#pragma unroll
    for (int q = 1; q < FIX_WAYS; ++q)
      if (q < lr.n_seg) row = add4(row, sq[q]);
    if (poison) row = make_float4(__builtin_nanf(""), __builtin_nanf(""), __builtin_nanf(""), __builtin_nanf(""));
    if (lr.tgt >= 0) {
      if (EPI == EPI_NOISE) row = perturb(ep, lr.tgt, b * LPR + l, nscale, row);
      epilogue_store<EPI>(ep, lr.tgt, off, row);
    } else {
      const __amdgpu_buffer_rsrc_t rsrc = __builtin_amdgcn_make_buffer_rsrc(partials, 0, (int)fx.part_bytes, 0x00020000);
      u32x4 u;
      u.x = __float_as_uint(row.x), u.y = __float_as_uint(row.y), u.z = __float_as_uint(row.z), u.w = __float_as_uint(row.w);
      __builtin_amdgcn_raw_buffer_store_b128(u, rsrc, (unsigned)(((int64_t)(~lr.tgt) * d + off) * 4), 0, 16);  // aux 16 = sc1
    }
  }
  if (lr.tgt < 0) combine_if_last<LPR, NB, EPI>(ep, partials, d, fx, ~lr.tgt, l);
}

// EXACT_ORDER rows longer than a tile: one lane group streams the row from global memory.
template <int LPR, int NB>
__global__ __launch_bounds__(64) void spmm_xl_kernel(const int32_t* __restrict__ xl,
                                                     const int64_t* __restrict__ vptr,
                                                     const int32_t* __restrict__ vtgt,
                                                     const ColVal* __restrict__ cv,
                                                     const float* __restrict__ X, int64_t ldx,
                                                     Epilogue ep, const uint32_t* __restrict__ x_mask) {
  const int v = xl[blockIdx.x];
  const int l = threadIdx.x;
  if (l >= LPR) return;
  const int64_t s = vptr[v], e = vptr[v + 1];
  const int tgt = vtgt[v];
#pragma unroll
  for (int b = 0; b < NB; ++b) {
    const int off = (b * LPR + l) * 4;
    float4 acc = make_float4(0.f, 0.f, 0.f, 0.f);
    for (int64_t c = s; c < e; c += (1 << 20)) {
      const int len = (int)std::min<int64_t>(e - c, 1 << 20);
      acc = x_mask ? walk_masked<8>(cv + c, 0, len, X + off, ldx, x_mask, acc) : walk<8>(cv + c, 0, len, X + off, ldx, acc);
    }
    if (ep.noise_eps != 0.f) acc = perturb(ep, tgt, b * LPR + l, noise_row_scale<LPR, NB>(ep, tgt, l), acc);
    if (ep.act) epilogue_store<EPI_ACT>(ep, tgt, off, acc);
    else epilogue_store(ep, tgt, off, acc);
  }
}

// Fix-up: FIX_WAYS lane groups per split row.  Group q adds partials q, q+W, q+2W, ... in that
// order; the W group sums are then added left to right (q = 0 first).  Fixed, published order.
template <int LPR, int NB>
__global__ __launch_bounds__(FIX_WAYS * LPR) void spmm_fixup_kernel(const LongRow* __restrict__ rows, int n_long,
                                                                    const float* __restrict__ partials, int64_t d,
                                                                    Epilogue ep, const uint32_t* __restrict__ out_mask) {
  __shared__ float4 s_part[FIX_WAYS][LPR];
  const int q = threadIdx.x / LPR;
  const int l = threadIdx.x % LPR;
  const LongRow lr = rows[blockIdx.x];
  if (out_mask && !mask_bit(out_mask, lr.row)) return;  // row not requested (block-uniform)
#pragma unroll
  for (int b = 0; b < NB; ++b) {
    const int off = (b * LPR + l) * 4;
    const float* p = partials + lr.slot_begin * d + off;
    float4 acc = make_float4(0.f, 0.f, 0.f, 0.f);
    if (q < lr.n_seg) {
      acc = *reinterpret_cast<const float4*>(p + (int64_t)q * d);
      int sgm = q + FIX_WAYS;
      for (; sgm + 7 * FIX_WAYS < lr.n_seg; sgm += 8 * FIX_WAYS) {
        float4 x[8];
#pragma unroll
        for (int u = 0; u < 8; ++u) x[u] = *reinterpret_cast<const float4*>(p + (int64_t)(sgm + u * FIX_WAYS) * d);
#pragma unroll
        for (int u = 0; u < 8; ++u) acc = add4(acc, x[u]);
      }
      for (; sgm < lr.n_seg; sgm += FIX_WAYS) acc = add4(acc, *reinterpret_cast<const float4*>(p + (int64_t)sgm * d));
    }
    if (b > 0) __syncthreads();
    s_part[q][l] = acc;
    __syncthreads();
    if (q == 0) {
      const int ways = lr.n_seg < FIX_WAYS ? lr.n_seg : FIX_WAYS;
      for (int w = 1; w < ways; ++w) acc = add4(acc, s_part[w][l]);
      if (ep.noise_eps != 0.f) acc = perturb(ep, lr.row, b * LPR + l, noise_row_scale<LPR, NB>(ep, lr.row, l), acc);
      if (ep.adam_p)
        epilogue_store<EPI_ADAM>(ep, lr.row, off, acc);
      else if (ep.act)
        epilogue_store<EPI_ACT>(ep, lr.row, off, acc);
      else
        epilogue_store<EPI_PLAIN>(ep, lr.row, off, acc);
    }
  }
}

// Any-width form of the perturbation (one wave per row, lane l owns features l, l + 64, ...): the row norm of the
// uniforms depends on (seed, stream, row, d) only, so every wave forms it up front; all 64 lanes must call.
__device__ __forceinline__ float generic_noise_scale(const Epilogue& ep, int64_t r, int64_t d, int lane) {
  float ss = 0.f;
  for (int64_t fb = lane; fb * 4 < d; fb += 64) {
    const float4 u = noise4(ep, r, (int)fb);
    const float uu[4] = {u.x, u.y, u.z, u.w};
    for (int c = 0; c < 4; ++c)
      if (fb * 4 + c < d) ss += uu[c] * uu[c];
  }
  for (int o = 32; o > 0; o >>= 1) ss += __shfl_xor(ss, o, 64);
  return ep.noise_eps / fmaxf(sqrtf(ss), 1e-12f);
}

__device__ __forceinline__ float generic_perturb(const Epilogue& ep, int64_t r, int64_t f, float scale, float t) {
  const float4 u = noise4(ep, r, (int)(f >> 2));
  const int c = (int)(f & 3);
  const float uf = c == 0 ? u.x : (c == 1 ? u.y : (c == 2 ? u.z : u.w));
  return __builtin_fmaf(sgn(t) * uf, scale, t);
}

__device__ __forceinline__ void generic_epilogue(const Epilogue& ep, int64_t r, int64_t f, float acc) {
  const int64_t o = r * ep.ldy + f;
  const bool live = ep.mask == nullptr || mask_bit(ep.mask, r);
  if (ep.addend && live) acc += ep.addend[o];
  if (ep.act && (ep.act_rows == 0 || r < ep.act_rows)) {
    if (ep.act == 1) acc = tanhf(acc);
    else acc = acc * (1.0f - ep.act_src[o] * ep.act_src[o]);
  }
  if (ep.Y) ep.Y[o] = acc;
  if (ep.sum_out) {
    float sres = acc;
    if (ep.sum_in && live) {
      float t = ep.sum_in[o];
      if (ep.sum_in2) t = t + ep.sum_in2[o];
      if (ep.sum_in3) t = t + ep.sum_in3[o];
      sres = t + acc;
    }
    if (ep.div != 1.0f) sres = sres / ep.div;
    if (ep.accumulate && live) sres = ep.sum_out[o] + sres;
    ep.sum_out[o] = sres;
  }
}

// Any d (slow path): 64 lanes stride over the features one float at a time.
__global__ __launch_bounds__(BLOCK) void spmm_generic_kernel(const int64_t* __restrict__ vptr,
                                                             const int32_t* __restrict__ vtgt,
                                                             int64_t n_vrows, const ColVal* __restrict__ cv,
                                                             const float* __restrict__ X, int64_t ldx,
                                                             float* __restrict__ partials, int64_t d,
                                                             Epilogue ep, const uint32_t* __restrict__ x_mask) {
  const int64_t v = (int64_t)blockIdx.x * (BLOCK / 64) + threadIdx.x / 64;
  const int l = threadIdx.x % 64;
  if (v >= n_vrows) return;
  const int64_t s = vptr[v], e = vptr[v + 1];
  const int tgt = vtgt[v];
  if (is_local(tgt)) return;  // spmm_generic_local_kernel
  const float nscale = (ep.noise_eps != 0.f && tgt >= 0) ? generic_noise_scale(ep, tgt, d, l) : 0.f;
  for (int64_t f = l; f < d; f += 64) {
    float acc = 0.f;
    for (int64_t j = s; j < e; ++j)
      if (!x_mask || mask_bit(x_mask, cv[j].col)) acc = __builtin_fmaf(cv[j].val, X[(int64_t)cv[j].col * ldx + f], acc);
    if (tgt < 0) {
      partials[(int64_t)(~tgt) * d + f] = acc;
      continue;
    }
    if (ep.noise_eps != 0.f) acc = generic_perturb(ep, tgt, f, nscale, acc);
    generic_epilogue(ep, tgt, f, acc);
  }
}

// The perturbation on its own: Y[r] = X[r] + sign(X[r]) * normalize(u_r) * eps for every row (or the rows of a bitmap) —
// the same arithmetic as the EPI_NOISE epilogue, so a layer whose product is shared between several perturbed passes
// (SimGCL: A.E0 feeds the clean pass and both views) is perturbed once per view without being multiplied again.
template <int LPR, int NB>
__global__ __launch_bounds__(BLOCK) void perturb_rows_kernel(const float* __restrict__ X, float* __restrict__ Y, int64_t n,
                                                             int64_t ld, Epilogue ep, const uint32_t* __restrict__ rows) {
  constexpr int GROUPS = BLOCK / LPR;
  const int64_t r = (int64_t)blockIdx.x * GROUPS + threadIdx.x / LPR;
  const int l = threadIdx.x % LPR;
  if (r >= n) return;                        // whole lane groups leave together
  if (rows && !mask_bit(rows, r)) return;
  const float scale = noise_row_scale<LPR, NB>(ep, r, l);
#pragma unroll
  for (int b = 0; b < NB; ++b) {
    const int64_t o = r * ld + (b * LPR + l) * 4;
    const float4 t = *reinterpret_cast<const float4*>(X + o);
    *reinterpret_cast<float4*>(Y + o) = perturb(ep, r, b * LPR + l, scale, t);
  }
}

// out[r] = g[r] * (1 - y[r]^2) at the rows flagged in `rows` (torch's tanh_backward at the batch's rows: where EGCF's
// backward starts, models/EGCF.py:60 under autograd); other rows untouched.  One thread per 4 features.
__global__ __launch_bounds__(BLOCK) void rows_tanh_bwd_kernel(const float* __restrict__ g, const float* __restrict__ y,
                                                              const uint32_t* __restrict__ rows, int64_t n, int64_t d4,
                                                              float* __restrict__ out) {
  const int64_t i = (int64_t)blockIdx.x * BLOCK + threadIdx.x;
  if (i >= n * d4) return;
  const int64_t r = i / d4;
  if (rows && !mask_bit(rows, r)) return;
  const float4 a = reinterpret_cast<const float4*>(g)[i], b = reinterpret_cast<const float4*>(y)[i];
  reinterpret_cast<float4*>(out)[i] = make_float4(a.x * (1.0f - b.x * b.x), a.y * (1.0f - b.y * b.y), a.z * (1.0f - b.z * b.z),
                                                  a.w * (1.0f - b.w * b.w));
}

// The stand-alone perturbation for any width: one wave per row.
__global__ __launch_bounds__(BLOCK) void perturb_rows_generic_kernel(const float* __restrict__ X, float* __restrict__ Y,
                                                                     int64_t n, int64_t d, Epilogue ep,
                                                                     const uint32_t* __restrict__ rows) {
  const int64_t r = (int64_t)blockIdx.x * (BLOCK / 64) + threadIdx.x / 64;
  const int l = threadIdx.x % 64;
  if (r >= n) return;
  if (rows && !mask_bit(rows, r)) return;
  const float scale = generic_noise_scale(ep, r, d, l);
  for (int64_t f = l; f < d; f += 64) Y[r * d + f] = generic_perturb(ep, r, f, scale, X[r * d + f]);
}

// Any d: one wave per local row, its segments walked one after the other, combined in the published order.
__global__ __launch_bounds__(BLOCK) void spmm_generic_local_kernel(const LocalRow* __restrict__ locals, int64_t n_local,
                                                                   const int64_t* __restrict__ vptr,
                                                                   const ColVal* __restrict__ cv,
                                                                   const float* __restrict__ X, int64_t ldx,
                                                                   float* __restrict__ partials, int64_t d, Epilogue ep,
                                                                   const uint32_t* __restrict__ x_mask) {
  const int64_t j = (int64_t)blockIdx.x * (BLOCK / 64) + threadIdx.x / 64;
  const int l = threadIdx.x % 64;
  if (j >= n_local) return;
  const LocalRow lr = locals[j];
  const float nscale = (ep.noise_eps != 0.f && lr.tgt >= 0) ? generic_noise_scale(ep, lr.tgt, d, l) : 0.f;
  for (int64_t f = l; f < d; f += 64) {
    float way[FIX_WAYS] = {0.f, 0.f, 0.f, 0.f};
    for (int sg = 0; sg < lr.n_seg; ++sg) {
      float acc = 0.f;
      for (int64_t k = vptr[lr.vrow + sg]; k < vptr[lr.vrow + sg + 1]; ++k)
        if (!x_mask || mask_bit(x_mask, cv[k].col)) acc = __builtin_fmaf(cv[k].val, X[(int64_t)cv[k].col * ldx + f], acc);
      way[sg % FIX_WAYS] = sg < FIX_WAYS ? acc : way[sg % FIX_WAYS] + acc;
    }
    float row = way[0];
    for (int q = 1; q < FIX_WAYS && q < lr.n_seg; ++q) row = row + way[q];
    if (lr.tgt < 0) {
      partials[(int64_t)(~lr.tgt) * d + f] = row;
    } else {
      if (ep.noise_eps != 0.f) row = generic_perturb(ep, lr.tgt, f, nscale, row);
      generic_epilogue(ep, lr.tgt, f, row);
    }
  }
}

__global__ __launch_bounds__(BLOCK) void fixup_generic_kernel(const LongRow* __restrict__ rows, int n_long,
                                                              const float* __restrict__ partials, int64_t d,
                                                              Epilogue ep) {
  const int g = blockIdx.x * (BLOCK / 64) + threadIdx.x / 64;
  const int l = threadIdx.x % 64;
  if (g >= n_long) return;
  const LongRow lr = rows[g];
  const float nscale = ep.noise_eps != 0.f ? generic_noise_scale(ep, lr.row, d, l) : 0.f;
  for (int64_t f = l; f < d; f += 64) {
    const float* p = partials + lr.slot_begin * d + f;
    float acc = 0.f;
    for (int q = 0; q < FIX_WAYS && q < lr.n_seg; ++q) {  // same order as spmm_fixup_kernel
      float a = p[(int64_t)q * d];
      for (int sgm = q + FIX_WAYS; sgm < lr.n_seg; sgm += FIX_WAYS) a += p[(int64_t)sgm * d];
      acc = q == 0 ? a : acc + a;
    }
    if (ep.noise_eps != 0.f) acc = generic_perturb(ep, lr.row, f, nscale, acc);
    generic_epilogue(ep, lr.row, f, acc);
  }
}

// out |= columns of the stored entries of every row flagged in `in` (one thread per virtual row: a row's pieces are its
// vrows).  The receptive field of a batch, one hop at a time (idg_graph_expand_rows).
__global__ __launch_bounds__(BLOCK) void expand_rows_kernel(int64_t n_vrows, const int64_t* __restrict__ vptr,
                                                            const int32_t* __restrict__ vrow_row,
                                                            const ColVal* __restrict__ cv, const uint32_t* __restrict__ in,
                                                            uint32_t* __restrict__ out) {
  const int64_t v = (int64_t)blockIdx.x * BLOCK + threadIdx.x;
  if (v >= n_vrows) return;
  const int32_t r = vrow_row[v];
  if (!((in[r >> 5] >> (r & 31)) & 1u)) return;
  for (int64_t j = vptr[v]; j < vptr[v + 1]; ++j) {
    const int32_t c = cv[j].col;
    const uint32_t bit = 1u << (c & 31);
    if (!(out[c >> 5] & bit)) atomicOr(&out[c >> 5], bit);
  }
}

// col_flags[c] = 1.0f for every column c of a stored entry of a row flagged in `in` (any graph shape; the flags are
// floats because the ranks of a sharded step SUM them: idg_graph_flag_cols).
__global__ __launch_bounds__(BLOCK) void flag_cols_kernel(int64_t n_vrows, const int64_t* __restrict__ vptr,
                                                          const int32_t* __restrict__ vrow_row, const ColVal* __restrict__ cv,
                                                          const uint32_t* __restrict__ in, float* __restrict__ col_flags) {
  const int64_t v = (int64_t)blockIdx.x * BLOCK + threadIdx.x;
  if (v >= n_vrows) return;
  const int32_t r = vrow_row[v];
  if (!((in[r >> 5] >> (r & 31)) & 1u)) return;
  for (int64_t j = vptr[v]; j < vptr[v + 1]; ++j) col_flags[cv[j].col] = 1.0f;  // (same value from every writer)
}

template <int LPR, int NB>
int launch_fast(const idg_graph* g, const float* X, int64_t ldx, float* partials, int64_t d,
                const Epilogue& ep, const uint32_t* x_mask, const uint32_t* out_mask, hipStream_t st) {
  // split rows: combined inside the tile kernels when the partial buffer is addressable through one
  // buffer descriptor (< 2 GiB); otherwise plain partial stores + the separate fix-up launch
  FixCtx fx{g->d_long, g->d_slot_long, g->d_long_cnt, 0u};
  const int64_t part_bytes = g->n_slots * d * (int64_t)sizeof(float);
  const bool fused_fix = g->n_long > 0 && part_bytes < ((int64_t)1 << 31) && !g->no_fused_fix;
  if (fused_fix) fx.part_bytes = (uint32_t)part_bytes;
  if (g->n_tiles > 0) {
    const dim3 grid((unsigned)g->n_tiles), block(BLOCK);
    // Band placement pays while the gathered panel lives in L2 / Infinity Cache (measured: 17.8 and
    // 36.9 MB panels -16 %); beyond it the two bands differ too much in miss cost (384 MB panel +9 %).
    const bool cache_resident = (int64_t)g->n_cols * d * 4 <= BAND_PANEL_BYTES;
    const Tile* tile_order = (g->d_tiles_banded && cache_resident) ? g->d_tiles_banded : (g->d_tiles_seq ? g->d_tiles_seq : g->d_tiles);
#define IDG_TILE(U, DYN, MINW, EPI)                                                                            \
  do {                                                                                                         \
    if (fused_fix)                                                                                             \
      hipLaunchKernelGGL((spmm_tile_kernel<LPR, NB, U, DYN, IDG_FUSED_MINW, EPI, true>  ), grid, block, 0, st, tile_order,  \
                         g->d_vptr, g->d_vtgt, g->d_cv, X, ldx, partials, d, ep, fx, g->d_local);              \
    else                                                                                                       \
      hipLaunchKernelGGL((spmm_tile_kernel<LPR, NB, U, DYN, MINW, EPI, false>  ), grid, block, 0, st,          \
                         tile_order, g->d_vptr, g->d_vtgt, g->d_cv, X, ldx, partials, d, ep, fx, g->d_local);  \
  } while (0)
    const int32_t* units = nullptr;
    int64_t ucap = 0;
    if (out_mask && !g->no_units) units_find(g->d_vptr, out_mask, &units, &ucap);
    if (units && ucap > 0 && (g->n_long == 0 || fused_fix) && !g->no_units) {
      // the bitmap's live work units are listed: one wave per unit, no tile is visited (spmm_units_kernel)
      const dim3 ugrid((unsigned)((ucap + IDG_UNITS_BLOCK / 64 - 1) / (IDG_UNITS_BLOCK / 64))), ublock(IDG_UNITS_BLOCK);
      if (ep.noise_eps != 0.f)
        hipLaunchKernelGGL((spmm_units_kernel<LPR, NB, EPI_NOISE>), ugrid, ublock, 0, st, units, ucap, g->d_vptr, g->d_vtgt, g->d_cv,
                           X, ldx, partials, d, ep, fx, g->d_local, nullptr);
      else if (ep.act && !x_mask)
        hipLaunchKernelGGL((spmm_units_kernel<LPR, NB, EPI_ACT>), ugrid, ublock, 0, st, units, ucap, g->d_vptr, g->d_vtgt, g->d_cv,
                           X, ldx, partials, d, ep, fx, g->d_local, nullptr);
      else if (x_mask)  // ... of a panel whose live rows are flagged too (a backward product between two small row sets)
        hipLaunchKernelGGL((spmm_units_kernel<LPR, NB, EPI_PLAIN, true>), ugrid, ublock, 0, st, units, ucap, g->d_vptr, g->d_vtgt,
                           g->d_cv, X, ldx, partials, d, ep, fx, g->d_local, x_mask);
      else
        hipLaunchKernelGGL((spmm_units_kernel<LPR, NB, EPI_PLAIN>), ugrid, ublock, 0, st, units, ucap, g->d_vptr, g->d_vtgt, g->d_cv,
                           X, ldx, partials, d, ep, fx, g->d_local, nullptr);
      IDG_HIP(hipGetLastError());
      return IDG_OK;
    }
    if (out_mask && ep.act && !x_mask) {  // flagged rows of an activated layer (EGCF's last forward layer)
      if (fused_fix)
        hipLaunchKernelGGL((spmm_tile_rows_kernel<LPR, NB, true, EPI_ACT>), grid, block, 0, st, tile_order, g->d_vptr,
                           g->d_vtgt, g->d_slot_row, g->d_cv, X, ldx, partials, d, ep, fx, out_mask, g->d_local, nullptr);
      else
        hipLaunchKernelGGL((spmm_tile_rows_kernel<LPR, NB, false, EPI_ACT>), grid, block, 0, st, tile_order, g->d_vptr,
                           g->d_vtgt, g->d_slot_row, g->d_cv, X, ldx, partials, d, ep, fx, out_mask, g->d_local, nullptr);
    } else if (out_mask && ep.noise_eps != 0.f) {  // flagged rows of a perturbed layer (the noise of a row depends on that row only)
      if (fused_fix)
        hipLaunchKernelGGL((spmm_tile_rows_kernel<LPR, NB, true, EPI_NOISE>), grid, block, 0, st, tile_order, g->d_vptr,
                           g->d_vtgt, g->d_slot_row, g->d_cv, X, ldx, partials, d, ep, fx, out_mask, g->d_local, nullptr);
      else
        hipLaunchKernelGGL((spmm_tile_rows_kernel<LPR, NB, false, EPI_NOISE>), grid, block, 0, st, tile_order, g->d_vptr,
                           g->d_vtgt, g->d_slot_row, g->d_cv, X, ldx, partials, d, ep, fx, out_mask, g->d_local, nullptr);
    } else if (out_mask && x_mask) {  // flagged output rows of a panel with flagged live rows
      if (fused_fix)
        hipLaunchKernelGGL((spmm_tile_rows_kernel<LPR, NB, true, EPI_PLAIN, true>), grid, block, 0, st, tile_order, g->d_vptr,
                           g->d_vtgt, g->d_slot_row, g->d_cv, X, ldx, partials, d, ep, fx, out_mask, g->d_local, x_mask);
      else
        hipLaunchKernelGGL((spmm_tile_rows_kernel<LPR, NB, false, EPI_PLAIN, true>), grid, block, 0, st, tile_order, g->d_vptr,
                           g->d_vtgt, g->d_slot_row, g->d_cv, X, ldx, partials, d, ep, fx, out_mask, g->d_local, x_mask);
    } else if (out_mask) {  // only flagged output rows (last forward layer of a training step)
      if (fused_fix)
        hipLaunchKernelGGL((spmm_tile_rows_kernel<LPR, NB, true>), grid, block, 0, st, tile_order, g->d_vptr, g->d_vtgt,
                           g->d_slot_row, g->d_cv, X, ldx, partials, d, ep, fx, out_mask, g->d_local, nullptr);
      else
        hipLaunchKernelGGL((spmm_tile_rows_kernel<LPR, NB, false>), grid, block, 0, st, tile_order, g->d_vptr, g->d_vtgt,
                           g->d_slot_row, g->d_cv, X, ldx, partials, d, ep, fx, out_mask, g->d_local, nullptr);
    } else if (x_mask && ep.act) {  // sparse-input form with the activation's derivative in the epilogue
      if (fused_fix)
        hipLaunchKernelGGL((spmm_tile_sparse_kernel<LPR, NB, true, EPI_ACT>), grid, block, 0, st, tile_order, g->d_vptr, g->d_vtgt,
                           g->d_cv, X, ldx, partials, d, ep, fx, x_mask, g->d_local);
      else
        hipLaunchKernelGGL((spmm_tile_sparse_kernel<LPR, NB, false, EPI_ACT>), grid, block, 0, st, tile_order, g->d_vptr, g->d_vtgt,
                           g->d_cv, X, ldx, partials, d, ep, fx, x_mask, g->d_local);
    } else if (x_mask && !g->no_units && g->n_xl == 0 && units_find(g->d_cv, x_mask, &units, &ucap, 1)) {
      // sparse-input form whose index-only half was done ahead of time (idg_graph_compact_inputs): the ordinary walk over
      // the live entries of every tile
      const ColVal* ccv = reinterpret_cast<const ColVal*>(units);
      const int32_t* cptr = reinterpret_cast<const int32_t*>(ccv + g->nnz);
      const int32_t* clive = cptr + g->n_vrows;
      if (fused_fix)
        hipLaunchKernelGGL((spmm_tile_compact_kernel<LPR, NB, true>), grid, block, 0, st, tile_order, g->d_vtgt, ccv, cptr, clive, X,
                           ldx, partials, d, ep, fx, g->d_local);
      else
        hipLaunchKernelGGL((spmm_tile_compact_kernel<LPR, NB, false>), grid, block, 0, st, tile_order, g->d_vtgt, ccv, cptr, clive, X,
                           ldx, partials, d, ep, fx, g->d_local);
    } else if (x_mask) {  // sparse-input form (first backward layer)
      if (fused_fix)
        hipLaunchKernelGGL((spmm_tile_sparse_kernel<LPR, NB, true>), grid, block, 0, st, tile_order, g->d_vptr, g->d_vtgt,
                           g->d_cv, X, ldx, partials, d, ep, fx, x_mask, g->d_local);
      else
        hipLaunchKernelGGL((spmm_tile_sparse_kernel<LPR, NB, false>), grid, block, 0, st, tile_order, g->d_vptr, g->d_vtgt,
                           g->d_cv, X, ldx, partials, d, ep, fx, x_mask, g->d_local);
    } else if (ep.noise_eps != 0.f) {  // perturbed layers: own instantiation (Philox + row-norm shuffles)
      IDG_TILE(8, true, 1, EPI_NOISE);
    } else if (ep.adam_p) {  // last backward product of a training step: Adam applied to each finished gradient row
      IDG_TILE(8, true, 1, EPI_ADAM);
    } else if (ep.act) {     // EGCF: tanh / its derivative applied to each finished row
      IDG_TILE(8, true, 1, EPI_ACT);
    } else switch (g->variant) {
      case 0: IDG_TILE(8, false, 1, EPI_PLAIN); break;
      case 1: IDG_TILE(8, true, 1, EPI_PLAIN); break;
      default: IDG_TILE(8, true, 8, EPI_PLAIN); break;
    }
#undef IDG_TILE
  }
  if (g->n_xl > 0)
    hipLaunchKernelGGL((spmm_xl_kernel<LPR, NB>), dim3((unsigned)g->n_xl), dim3(64), 0, st, g->d_xl, g->d_vptr,
                       g->d_vtgt, g->d_cv, X, ldx, ep, x_mask);
  if (g->n_long > 0 && !fused_fix) {
    hipLaunchKernelGGL((spmm_fixup_kernel<LPR, NB>), dim3((unsigned)g->n_long), dim3(FIX_WAYS * LPR), 0, st,
                       g->d_long, (int)g->n_long, partials, d, ep, out_mask);
  }
  IDG_HIP(hipGetLastError());
  return IDG_OK;
}

int spmm_dispatch(const idg_graph* g, const float* X, int64_t ldx, int64_t d, void* ws, const Epilogue& ep,
                  hipStream_t st, const uint32_t* x_mask = nullptr, const uint32_t* out_mask = nullptr) {
  float* partials = reinterpret_cast<float*>(ws);
  if (g->n_slots > 0 && !partials) return idg::fail(IDG_E_INVALID, "idg_spmm: workspace is NULL but the graph has split rows");
  const bool aligned = (ldx % 4 == 0) && (ep.ldy % 4 == 0) && ((uintptr_t)X % 16 == 0) &&
                       ((uintptr_t)ep.Y % 16 == 0) && ((uintptr_t)ep.addend % 16 == 0) &&
                       ((uintptr_t)ep.sum_in % 16 == 0) && ((uintptr_t)ep.sum_in2 % 16 == 0) &&
                       ((uintptr_t)ep.sum_in3 % 16 == 0) && ((uintptr_t)ep.sum_out % 16 == 0);
  if (aligned) {
    switch (d) {
      case 32: return launch_fast<8, 1>(g, X, ldx, partials, d, ep, x_mask, out_mask, st);
      case 64: return launch_fast<16, 1>(g, X, ldx, partials, d, ep, x_mask, out_mask, st);
      case 128: return launch_fast<32, 1>(g, X, ldx, partials, d, ep, x_mask, out_mask, st);
      case 256: return launch_fast<64, 1>(g, X, ldx, partials, d, ep, x_mask, out_mask, st);
      case 512: return launch_fast<64, 2>(g, X, ldx, partials, d, ep, x_mask, out_mask, st);
      default: break;
    }
  }
  if (ep.adam_p) return idg::fail(IDG_E_UNSUPPORTED, "idg_propagate: the fused Adam epilogue needs the tiled kernels");
  if (out_mask)
    return idg::fail(IDG_E_UNSUPPORTED, "idg_propagate: row-restricted output needs d in {32,64,128,256,512} and 16-byte aligned panels (d=%lld)",
                     (long long)d);
  if (g->n_vrows > 0) {
    const unsigned nb = (unsigned)((g->n_vrows + (BLOCK / 64) - 1) / (BLOCK / 64));
    hipLaunchKernelGGL(spmm_generic_kernel, dim3(nb), dim3(BLOCK), 0, st, g->d_vptr, g->d_vtgt, g->n_vrows,
                       g->d_cv, X, ldx, partials, d, ep, x_mask);
  }
  if (g->n_local > 0) {
    const unsigned nb = (unsigned)((g->n_local + (BLOCK / 64) - 1) / (BLOCK / 64));
    hipLaunchKernelGGL(spmm_generic_local_kernel, dim3(nb), dim3(BLOCK), 0, st, g->d_local, g->n_local, g->d_vptr, g->d_cv,
                       X, ldx, partials, d, ep, x_mask);
  }
  if (g->n_long > 0) {
    const unsigned nb = (unsigned)((g->n_long + (BLOCK / 64) - 1) / (BLOCK / 64));
    hipLaunchKernelGGL(fixup_generic_kernel, dim3(nb), dim3(BLOCK), 0, st, g->d_long, (int)g->n_long, partials,
                       d, ep);
  }
  IDG_HIP(hipGetLastError());
  return IDG_OK;
}

template <int LPR, int NB>
static int launch_rows_multi(const idg_graph* g, int np, const MultiPanel& mp, int64_t ldx, int64_t d,
                             const uint32_t* out_rows, bool first_is_noise, hipStream_t st) {
  FixCtx fx{g->d_long, g->d_slot_long, g->d_long_cnt, (uint32_t)(g->n_slots * d * (int64_t)sizeof(float))};
  const bool cache_resident = (int64_t)g->n_cols * d * 4 <= BAND_PANEL_BYTES;
  const Tile* tile_order = (g->d_tiles_banded && cache_resident) ? g->d_tiles_banded : (g->d_tiles_seq ? g->d_tiles_seq : g->d_tiles);
  const dim3 grid((unsigned)g->n_tiles), block(BLOCK);
#define IDG_MULTI(NP, E0)                                                                                              \
  hipLaunchKernelGGL((spmm_tile_rows_multi_kernel<LPR, NB, NP, E0>), grid, block, 0, st, tile_order, g->d_vptr, g->d_vtgt, \
                     g->d_slot_row, g->d_cv, mp, ldx, d, fx, out_rows, g->d_local)
  if (np == 3 && !first_is_noise) IDG_MULTI(3, EPI_PLAIN);
  else if (np == 2 && !first_is_noise) IDG_MULTI(2, EPI_PLAIN);
  else if (np == 2 && first_is_noise) IDG_MULTI(2, EPI_NOISE);
  else return idg::fail(IDG_E_UNSUPPORTED, "idg_propagate_views: unsupported panel combination");
#undef IDG_MULTI
  IDG_HIP(hipGetLastError());
  return IDG_OK;
}

int64_t seg_len_for(int64_t len) {
  // ~sqrt(len) balances the parallel segment pass against the sequential fix-up pass.
  int64_t s = (int64_t)std::ceil(std::sqrt((double)len));
  s = (s + 7) / 8 * 8;
  return std::min<int64_t>(std::max<int64_t>(s, 64), TILE_NNZ);
}

// [0, n) cut into contiguous pieces, one per worker thread (the schedule build's passes over the stored entries: at
// configs[4] size — 4e8 entries — each sequential pass is ~0.5 s of the handle's creation).  fn(begin, end, worker).
template <typename F>
void parallel_ranges(int64_t n, F fn, int64_t min_per_thread = 1 << 20) {
  int nt = (int)std::min<int64_t>(std::min<unsigned>(std::max(1u, std::thread::hardware_concurrency()), 32u),
                                  std::max<int64_t>(1, n / std::max<int64_t>(min_per_thread, 1)));
  if (const char* v = std::getenv("IDG_BUILD_THREADS")) nt = std::max(1, std::min(nt, std::atoi(v)));
  if (nt <= 1) {
    fn((int64_t)0, n, 0);
    return;
  }
  std::vector<std::thread> th;
  th.reserve((size_t)nt);
  for (int t = 0; t < nt; ++t) {
    const int64_t b = n * t / nt, e = n * (t + 1) / nt;
    th.emplace_back([=] { fn(b, e, t); });
  }
  for (auto& x : th) x.join();
}

struct DeviceGuard {
  int prev = -1;
  bool switched = false;
  int enter(int dev) {
    IDG_HIP(hipGetDevice(&prev));
    if (prev != dev) {
      IDG_HIP(hipSetDevice(dev));
      switched = true;
    }
    return IDG_OK;
  }
  ~DeviceGuard() {
    if (switched) (void)hipSetDevice(prev);
  }
};

// ---- the three looks at the stored entries when they live on the device (graph_build, idg_graph_create_from_device) ----
__global__ __launch_bounds__(BLOCK) void check_columns_kernel(const int32_t* __restrict__ indices, int64_t nnz, int64_t n_cols,
                                                              int64_t* __restrict__ first_bad) {
  const int64_t stride = (int64_t)gridDim.x * BLOCK;
  for (int64_t k = (int64_t)blockIdx.x * BLOCK + threadIdx.x; k < nnz; k += stride) {
    const int32_t c = indices[k];
    if (c < 0 || c >= n_cols) {  // the lowest offending position wins (reported in the error message)
      unsigned long long* p = reinterpret_cast<unsigned long long*>(first_bad);
      unsigned long long cur = *p;
      while ((cur == ~0ull || (unsigned long long)k < cur) && atomicCAS(p, cur, (unsigned long long)k) != cur) cur = *p;
    }
  }
}

__global__ __launch_bounds__(BLOCK) void gather_columns_kernel(const int32_t* __restrict__ indices, const int64_t* __restrict__ where,
                                                               int64_t count, int32_t* __restrict__ out) {
  const int64_t i = (int64_t)blockIdx.x * BLOCK + threadIdx.x;
  if (i < count) out[i] = where[i] >= 0 ? indices[where[i]] : 0;
}

// 16 lanes per virtual row: cv[vptr[v] + j] = (indices[src[v] + j], values[src[v] + j])
__global__ __launch_bounds__(BLOCK) void fill_entries_kernel(const int32_t* __restrict__ indices, const float* __restrict__ values,
                                                             const int64_t* __restrict__ vptr, const int64_t* __restrict__ src,
                                                             int64_t n_vrows, ColVal* __restrict__ cv) {
  const int64_t v = (int64_t)blockIdx.x * (BLOCK / 16) + threadIdx.x / 16;
  if (v >= n_vrows) return;
  const int64_t b = vptr[v], e = vptr[v + 1], s = src[v];
  for (int64_t j = b + threadIdx.x % 16; j < e; j += 16) cv[j] = ColVal{indices[s + (j - b)], values[s + (j - b)]};
}

template <typename T>
int upload(T** dst, const std::vector<T>& src) {
  *dst = nullptr;
  if (src.empty()) return IDG_OK;
  IDG_HIP(hipMalloc(reinterpret_cast<void**>(dst), src.size() * sizeof(T)));
  IDG_HIP(hipMemcpy(*dst, src.data(), src.size() * sizeof(T), hipMemcpyHostToDevice));
  return IDG_OK;
}

}  // namespace

extern "C" {

int idg_device_count(void) {
  int n = 0;
  if (hipGetDeviceCount(&n) != hipSuccess) return 0;
  return n;
}

// The schedule build.  indptr is a HOST array either way (the row pointer is all the schedule itself depends on: virtual
// rows, tiles, launch orders, split tables).  The stored entries are either host arrays (indices / values: the entry list
// is laid out on the host and uploaded) or DEVICE arrays (d_indices / d_values, indices == values == NULL: the three
// places that look at an entry — the range check, the tiles' median columns, the layout of the (col, val) list in tile
// order — run as kernels on `st`, and nothing of size nnz crosses the bus).
static int graph_build(int device, int64_t n_rows, int64_t n_cols, int64_t nnz, const int64_t* indptr,
                       const int32_t* indices, const float* values, const int32_t* d_indices, const float* d_values,
                       hipStream_t st, uint32_t flags, int64_t split_threshold, idg_graph** out) {
  const bool on_device = indices == nullptr && nnz > 0;
  IDG_REQUIRE(out, "idg_graph_create: out is NULL");
  IDG_REQUIRE(n_rows >= 0 && n_cols >= 0 && nnz >= 0, "idg_graph_create: negative size");
  IDG_REQUIRE(indptr && (nnz == 0 || (indices && values) || (d_indices && d_values)), "idg_graph_create: NULL CSR array");
  IDG_REQUIRE(n_rows < ((int64_t)1 << 31) && n_cols < ((int64_t)1 << 31), "idg_graph_create: more than 2^31 rows/cols");
  IDG_REQUIRE(indptr[0] == 0 && indptr[n_rows] == nnz, "idg_graph_create: indptr[0]=%lld indptr[n]=%lld nnz=%lld",
              (long long)indptr[0], (long long)indptr[n_rows], (long long)nnz);
  IDG_REQUIRE(split_threshold >= 0, "idg_graph_create: negative split_threshold");
  for (int64_t r = 0; r < n_rows; ++r)
    IDG_REQUIRE(indptr[r + 1] >= indptr[r], "idg_graph_create: indptr not monotone at row %lld", (long long)r);
  if (!on_device) {
    std::atomic<int64_t> bad{-1};  // the first offending entry of the lowest range that has one is what the message names
    parallel_ranges(nnz, [&](int64_t b, int64_t e, int) {
      for (int64_t k = b; k < e; ++k)
        if (indices[k] < 0 || indices[k] >= n_cols) {
          int64_t cur = bad.load();
          while ((cur < 0 || k < cur) && !bad.compare_exchange_weak(cur, k)) {
          }
          return;
        }
    });
    const int64_t k = bad.load();
    IDG_REQUIRE(k < 0, "idg_graph_create: column %d outside [0,%lld) at entry %lld", indices[k], (long long)n_cols, (long long)k);
  }
  const int ndev = idg_device_count();
  if (ndev <= 0) return idg::fail(IDG_E_NODEVICE, "idg_graph_create: no HIP device visible (this library has no CPU path)");
  IDG_REQUIRE(device >= 0 && device < ndev, "idg_graph_create: device %d outside [0,%d)", device, ndev);
  DeviceGuard guard;
  {
    const int rc0 = guard.enter(device);
    if (rc0 != IDG_OK) return rc0;
  }
  if (on_device) {  // the range check of the column ids, on the device
    int64_t* d_bad = nullptr;
    IDG_HIP(hipMalloc(reinterpret_cast<void**>(&d_bad), sizeof(int64_t)));
    int64_t bad = -1;
    hipError_t e1 = hipMemcpyAsync(d_bad, &bad, sizeof(int64_t), hipMemcpyHostToDevice, st);
    if (e1 == hipSuccess) {
      hipLaunchKernelGGL(check_columns_kernel, dim3((unsigned)std::min<int64_t>((nnz + BLOCK - 1) / BLOCK, 1 << 16)), dim3(BLOCK), 0,
                         st, d_indices, nnz, n_cols, d_bad);
      e1 = hipMemcpyAsync(&bad, d_bad, sizeof(int64_t), hipMemcpyDeviceToHost, st);
    }
    if (e1 == hipSuccess) e1 = hipStreamSynchronize(st);
    (void)hipFree(d_bad);
    IDG_HIP(e1);
    IDG_REQUIRE(bad < 0, "idg_graph_create_from_device: a column id outside [0,%lld) at entry %lld", (long long)n_cols, (long long)bad);
  }

  idg_graph* g = new (std::nothrow) idg_graph;
  if (!g) return idg::fail(IDG_E_NOMEM, "idg_graph_create: out of memory");
  g->device = device;
  g->n_rows = n_rows;
  g->n_cols = n_cols;
  g->nnz = nnz;
  g->flags = flags;
  if (const char* v = std::getenv("IDG_SPMM_VARIANT")) g->variant = std::atoi(v);
  if (const char* v = std::getenv("IDG_FUSED_FIX")) g->no_fused_fix = std::atoi(v) == 0;
  if (const char* v = std::getenv("IDG_LIVE_UNITS")) g->no_units = std::atoi(v) == 0;
  if (const char* v = std::getenv("IDG_TILE_NNZ")) g->tile_cap = std::min<int64_t>(std::max(64, std::atoi(v)), TILE_NNZ);
  const bool exact = (flags & IDG_GRAPH_EXACT_ORDER) != 0;
  int64_t T = split_threshold > 0 ? split_threshold : DEFAULT_SPLIT;
  T = std::min<int64_t>(T, TILE_NNZ);
  g->split_threshold = exact ? 0 : T;

  // ---- virtual rows
  std::vector<int64_t> vptr;
  std::vector<int32_t> vtgt;
  std::vector<int32_t> vrow_row;  // the CSR row each vrow is (a piece of); non-decreasing
  std::vector<LongRow> longs;
  std::vector<LocalRow> locals;  // in vrow order == tile order of creation
  std::vector<int32_t> slot_row, slot_long;
  vptr.reserve((size_t)n_rows + 1);
  vtgt.reserve((size_t)n_rows);
  vptr.push_back(0);
  int64_t slots = 0;
  const int64_t C = std::min<int64_t>(CHUNK_NNZ, g->tile_cap);
  // one row, or one chunk of a row: entries [s, e) -> target tgt (row id, or ~global partial slot)
  auto emit_piece = [&](int64_t s, int64_t e, int32_t tgt, int64_t S) {
    const int64_t nseg = (e - s + S - 1) / S;
    if (nseg <= 1) {
      vtgt.push_back(tgt);
      vptr.push_back(e);
      return;
    }
    locals.push_back(LocalRow{tgt, (int32_t)vtgt.size(), (int16_t)nseg, 0});
    for (int64_t k = 0; k < nseg; ++k) {
      vtgt.push_back(LOCAL_CODE);  // LDS slot assigned when the tile is packed
      vptr.push_back(std::min(e, s + (k + 1) * S));
    }
  };
  for (int64_t r = 0; r < n_rows; ++r) {
    const int64_t s = indptr[r], e = indptr[r + 1], len = e - s;
    vrow_row.resize(vtgt.size(), r > 0 ? (int32_t)(r - 1) : 0);  // vrows of the previous row
    if (exact || len <= T) {
      vtgt.push_back((int32_t)r);
      vptr.push_back(e);
      continue;
    }
    if (vtgt.size() + (size_t)(len / 64) + 8 >= (size_t)INT32_MAX) {
      delete g;
      return idg::fail(IDG_E_UNSUPPORTED, "idg_graph_create: too many row segments");
    }
    ++g->n_split;
    g->h_long_rows.push_back(r);
    if (len <= C) {  // all segments in one tile: combined in LDS
      const int64_t S = std::max<int64_t>(seg_len_for(len), (len + LSLOTS - 1) / LSLOTS);
      g->h_seg_len.push_back(S);
      g->h_chunk_len.push_back(0);
      emit_piece(s, e, (int32_t)r, S);
      continue;
    }
    // chunks of C entries, each a local row that produces one global partial
    const int64_t S = std::max<int64_t>(seg_len_for(C), (C + LSLOTS - 1) / LSLOTS);
    const int64_t nch = (len + C - 1) / C;
    if (slots + nch >= (int64_t)INT32_MAX - 2 * LSLOTS) {
      delete g;
      return idg::fail(IDG_E_UNSUPPORTED, "idg_graph_create: too many row chunks");
    }
    g->h_seg_len.push_back(S);
    g->h_chunk_len.push_back(C);
    longs.push_back(LongRow{slots, (int32_t)r, (int32_t)nch});
    for (int64_t k = 0; k < nch; ++k) {
      slot_row.push_back((int32_t)r);
      slot_long.push_back((int32_t)longs.size() - 1);
      emit_piece(s + k * C, std::min(e, s + (k + 1) * C), (int32_t)~(int32_t)(slots + k), S);
    }
    slots += nch;
  }
  vrow_row.resize(vtgt.size(), (int32_t)std::max<int64_t>(n_rows - 1, 0));
  g->n_vrows = (int64_t)vtgt.size();
  g->n_long = (int64_t)longs.size();
  g->n_slots = slots;
  g->n_local = (int64_t)locals.size();

  // ---- tiles
  std::vector<Tile> tiles;
  std::vector<int32_t> xl;
  {
    int64_t v = 0;
    size_t li = 0;  // next local row (they are in vrow order)
    const int64_t nv = g->n_vrows;
    while (v < nv) {
      const int64_t len0 = vptr[(size_t)v + 1] - vptr[(size_t)v];
      if (len0 > TILE_NNZ) {  // only possible with EXACT_ORDER
        xl.push_back((int32_t)v);
        ++v;
        continue;
      }
      int64_t w = v;
      const int64_t nz0 = vptr[(size_t)v];
      const size_t li0 = li;
      int lsl = 0;
      while (w < nv && w - v < TILE_VROWS) {
        if (li < locals.size() && locals[li].vrow == w) {  // a local row goes into a tile whole
          LocalRow& L = locals[li];
          const bool fits = (w - v + L.n_seg <= TILE_VROWS) && (lsl + L.n_seg <= LSLOTS) &&
                            (w == v || vptr[(size_t)(w + L.n_seg)] - nz0 <= g->tile_cap);
          if (!fits) break;
          L.lslot = (int16_t)lsl;
          for (int q = 0; q < L.n_seg; ++q) vtgt[(size_t)(w + q)] = LOCAL_CODE + lsl + q;
          lsl += L.n_seg;
          w += L.n_seg;
          ++li;
        } else {
          if (!(w == v || vptr[(size_t)w + 1] - nz0 <= g->tile_cap)) break;
          ++w;
        }
      }
      Tile t{};
      t.nnz_begin = nz0;
      t.vrow_begin = (int32_t)v;
      t.n_vrows = (int16_t)(w - v);
      t.n_local = (int16_t)(li - li0);
      t.local_begin = (int32_t)li0;
      t.row_first = vrow_row[(size_t)v];
      t.row_last = vrow_row[(size_t)w - 1];
      t.nnz_count = (int32_t)(vptr[(size_t)w] - nz0);
      tiles.push_back(t);
      v = w;
    }
  }
  // Launch order.  (1) Heaviest tiles first (a tile's cost grows with its entry count; hub-row
  // segments feed the fix-up pass).  (2) XCD placement: workgroups are dealt round-robin to the
  // 8 XCDs (block b -> XCD b % 8, observed, speed only), each with a private 4 MiB L2.  Tiles are
  // ranked by the median column they gather from and cut into `bands` groups of equal entry
  // count; band k is placed on XCDs [8k/bands, 8(k+1)/bands).  For the bipartite adjacency the
  // two bands are exactly "user rows (gather item rows)" and "item rows (gather user rows)", so
  // each L2 caches one of the two panels instead of both.
  auto tile_nnz = [&](const Tile& a) { return vptr[(size_t)a.vrow_begin + a.n_vrows] - a.nnz_begin; };
  std::stable_sort(tiles.begin(), tiles.end(), [&](const Tile& a, const Tile& b) { return tile_nnz(a) > tile_nnz(b); });
  int bands = 8;
  if (const char* v = std::getenv("IDG_XCD_BANDS")) bands = std::atoi(v);
  const std::vector<Tile> tiles_plain = tiles;
  std::vector<Tile> tiles_seq;
  bool banded = false;
  if ((bands == 2 || bands == 4 || bands == 8) && tiles.size() >= 64) {
    banded = true;
    const size_t nt = tiles.size();
    std::vector<int32_t> med(nt);
    if (!on_device) {
      for (size_t t = 0; t < nt; ++t) {
        const int64_t b = tiles[t].nnz_begin, e = b + tile_nnz(tiles[t]);
        med[t] = e > b ? indices[(b + e) / 2] : 0;  // cheap proxy: the middle stored entry's column
      }
    } else {  // the same lookups as one gather on the device
      std::vector<int64_t> where(nt);
      for (size_t t = 0; t < nt; ++t) {
        const int64_t b = tiles[t].nnz_begin, e = b + tile_nnz(tiles[t]);
        where[t] = e > b ? (b + e) / 2 : -1;
      }
      int64_t* d_where = nullptr;
      int32_t* d_med = nullptr;
      hipError_t e1 = hipMalloc(reinterpret_cast<void**>(&d_where), nt * sizeof(int64_t));
      if (e1 == hipSuccess) e1 = hipMalloc(reinterpret_cast<void**>(&d_med), nt * sizeof(int32_t));
      if (e1 == hipSuccess) e1 = hipMemcpyAsync(d_where, where.data(), nt * sizeof(int64_t), hipMemcpyHostToDevice, st);
      if (e1 == hipSuccess) {
        hipLaunchKernelGGL(gather_columns_kernel, dim3((unsigned)((nt + BLOCK - 1) / BLOCK)), dim3(BLOCK), 0, st, d_indices, d_where,
                           (int64_t)nt, d_med);
        e1 = hipMemcpyAsync(med.data(), d_med, nt * sizeof(int32_t), hipMemcpyDeviceToHost, st);
      }
      if (e1 == hipSuccess) e1 = hipStreamSynchronize(st);
      (void)hipFree(d_where);
      (void)hipFree(d_med);
      if (e1 != hipSuccess) {
        delete g;
        IDG_HIP(e1);
      }
    }
    std::vector<size_t> order(nt);
    for (size_t t = 0; t < nt; ++t) order[t] = t;
    std::stable_sort(order.begin(), order.end(), [&](size_t a, size_t b) { return med[a] < med[b]; });
    std::vector<int> band(nt, 0);
    int64_t acc = 0;
    for (size_t r = 0; r < nt; ++r) {
      band[order[r]] = (int)std::min<int64_t>(bands - 1, acc * bands / std::max<int64_t>(nnz, 1));
      acc += tile_nnz(tiles[order[r]]);
    }
    std::vector<std::vector<Tile>> by_band((size_t)bands);
    for (size_t t = 0; t < nt; ++t) by_band[(size_t)band[t]].push_back(tiles[t]);  // keeps heaviest-first inside a band
    for (const auto& bt : by_band) tiles_seq.insert(tiles_seq.end(), bt.begin(), bt.end());
    std::vector<size_t> cur((size_t)bands, 0);
    std::vector<Tile> placed;
    placed.reserve(nt);
    const int per = 8 / bands;
    for (size_t p = 0; placed.size() < nt; ++p) {
      int k = (int)((p % 8) / per);
      if (cur[(size_t)k] >= by_band[(size_t)k].size()) {  // this band ran dry: take from the fullest remaining band
        size_t best = 0, left = 0;
        for (size_t q = 0; q < (size_t)bands; ++q)
          if (by_band[q].size() - cur[q] > left) left = by_band[q].size() - cur[q], best = q;
        k = (int)best;
      }
      placed.push_back(by_band[(size_t)k][cur[(size_t)k]++]);
    }
    tiles.swap(placed);
  }
  const std::vector<Tile>& tiles_banded = tiles;
  g->n_tiles = (int64_t)tiles.size();
  g->n_xl = (int64_t)xl.size();

  bool sort_tiles = true;
  if (const char* v = std::getenv("IDG_TILE_SORT")) sort_tiles = std::atoi(v) != 0;
  // (uninitialised: every entry is written exactly once below — by its tile's pass, or here when it belongs to no tile)
  std::unique_ptr<ColVal[]> cv_store(on_device ? nullptr : new (std::nothrow) ColVal[(size_t)std::max<int64_t>(nnz, 1)]);
  if (!on_device && !cv_store) {
    delete g;
    return idg::fail(IDG_E_NOMEM, "idg_graph_create: out of host memory for the entry list");
  }
  ColVal* cv = cv_store.get();
  // device mode: where (in CSR order) the entries of each vrow of the FINAL order come from; the list itself is laid out by
  // fill_entries_kernel once the vrow pointers are uploaded
  std::vector<int64_t> src_of_vrow;
  if (on_device) src_of_vrow.assign(vptr.begin(), vptr.end() - 1);
  if (!on_device && (!(sort_tiles && !tiles_plain.empty()) || !xl.empty()))
    parallel_ranges(nnz, [&](int64_t b, int64_t e, int) {
      for (int64_t k = b; k < e; ++k) cv[(size_t)k] = ColVal{indices[k], values[k]};
    });
  // Inside every tile the work units — a plain vrow, or a split row's run of segments — are laid out LONGEST FIRST.
  // The lane groups of a wave walk their vrows in lockstep (a round lasts as long as its longest walk) and draw
  // consecutive units from the tile's counter: sorted by length, the four (d = 64) walks of a round are of similar
  // length instead of random ones, and the tile ends with its short rows (longest-first list scheduling).  Only the
  // order of whole rows inside a tile changes: every row's entries, segments and summation order are untouched.
  if (sort_tiles && !tiles_plain.empty()) {
    std::vector<int64_t> vptr2(vptr);
    std::vector<int32_t> vtgt2(vtgt), vrow_row2(vrow_row);
    std::vector<size_t> local_of(vtgt.size(), SIZE_MAX);  // first segment's vrow -> its LocalRow
    for (size_t li = 0; li < locals.size(); ++li) local_of[(size_t)locals[li].vrow] = li;
    struct Unit {
      int32_t v, n;
      int64_t len;
    };
    // tiles are independent of each other here (each rewrites its own vrows, its own LocalRows and its own stretch of
    // the entry list): ranges of tiles on worker threads
    parallel_ranges((int64_t)tiles_plain.size(), [&](int64_t t_lo, int64_t t_hi, int) {
    std::vector<Unit> units;
    for (int64_t ti = t_lo; ti < t_hi; ++ti) {
      const Tile& t = tiles_plain[(size_t)ti];
      units.clear();
      for (int32_t v = t.vrow_begin; v < t.vrow_begin + t.n_vrows;) {
        const int32_t nseg = local_of[(size_t)v] != SIZE_MAX ? (int32_t)locals[local_of[(size_t)v]].n_seg : 1;
        units.push_back(Unit{v, nseg, vptr[(size_t)(v + nseg)] - vptr[(size_t)v]});
        v += nseg;
      }
      std::stable_sort(units.begin(), units.end(), [](const Unit& a, const Unit& b) { return a.len > b.len; });
      int64_t pos = t.nnz_begin;
      int32_t nv = t.vrow_begin;
      for (const Unit& u : units) {
        if (local_of[(size_t)u.v] != SIZE_MAX) locals[local_of[(size_t)u.v]].vrow = nv;
        for (int32_t q = 0; q < u.n; ++q) {
          const int64_t b = vptr[(size_t)(u.v + q)], e = vptr[(size_t)(u.v + q) + 1];
          vptr2[(size_t)nv] = pos;
          vtgt2[(size_t)nv] = vtgt[(size_t)(u.v + q)];
          vrow_row2[(size_t)nv] = vrow_row[(size_t)(u.v + q)];
          if (on_device) src_of_vrow[(size_t)nv] = b;
          else
            for (int64_t k = b; k < e; ++k) cv[(size_t)(pos + (k - b))] = ColVal{indices[k], values[k]};
          pos += e - b;
          ++nv;
        }
      }
      // (vptr2[nv] — the next tile's first entry, or an xl vrow's start — keeps the value it was copied with: tile
      //  boundaries do not move)
    }
    }, 256);
    vptr.swap(vptr2);
    vtgt.swap(vtgt2);
    vrow_row.swap(vrow_row2);
  }

  int rc = IDG_OK;
  if (nnz > 0 && !on_device) {
    rc = [&]() -> int {
      IDG_HIP(hipMalloc(reinterpret_cast<void**>(&g->d_cv), (size_t)nnz * sizeof(ColVal)));
      IDG_HIP(hipMemcpy(g->d_cv, cv, (size_t)nnz * sizeof(ColVal), hipMemcpyHostToDevice));
      return IDG_OK;
    }();
  }
  if (rc == IDG_OK) rc = upload(&g->d_vptr, vptr);
  if (rc == IDG_OK && on_device) {
    // the (col, val) list in tile order, laid out on the device: entry k of vrow v comes from CSR position
    // src_of_vrow[v] + (k - vptr[v])
    rc = [&]() -> int {
      int64_t* d_src = nullptr;
      IDG_HIP(hipMalloc(reinterpret_cast<void**>(&g->d_cv), (size_t)nnz * sizeof(ColVal)));
      IDG_HIP(hipMalloc(reinterpret_cast<void**>(&d_src), src_of_vrow.size() * sizeof(int64_t)));
      hipError_t e1 = hipMemcpyAsync(d_src, src_of_vrow.data(), src_of_vrow.size() * sizeof(int64_t), hipMemcpyHostToDevice, st);
      if (e1 == hipSuccess) {
        hipLaunchKernelGGL(fill_entries_kernel, dim3((unsigned)((g->n_vrows + BLOCK / 16 - 1) / (BLOCK / 16))), dim3(BLOCK), 0, st,
                           d_indices, d_values, g->d_vptr, d_src, g->n_vrows, g->d_cv);
        e1 = hipStreamSynchronize(st);
      }
      (void)hipFree(d_src);
      IDG_HIP(e1);
      return IDG_OK;
    }();
  }
  if (rc == IDG_OK) rc = upload(&g->d_vtgt, vtgt);
  if (rc == IDG_OK) rc = upload(&g->d_tiles, tiles_plain);
  if (rc == IDG_OK && banded) rc = upload(&g->d_tiles_banded, tiles_banded);
  if (rc == IDG_OK && banded && !std::getenv("IDG_NO_SEQ_BANDS")) rc = upload(&g->d_tiles_seq, tiles_seq);
  if (rc == IDG_OK) rc = upload(&g->d_local, locals);
  if (rc == IDG_OK) rc = upload(&g->d_long, longs);
  if (rc == IDG_OK) rc = upload(&g->d_slot_row, slot_row);
  if (rc == IDG_OK) rc = upload(&g->d_slot_long, slot_long);
  if (rc == IDG_OK) rc = upload(&g->d_long_cnt, std::vector<int>(longs.size() * MAX_PANELS, 0));  // one set per panel
  if (rc == IDG_OK) rc = upload(&g->d_xl, xl);
  if (rc == IDG_OK) rc = upload(&g->d_vrow_row, vrow_row);
  if (rc == IDG_OK) {  // row -> work unit, chunk slot -> its LocalRow (idg_graph_live_units)
    std::vector<int32_t> row_unit((size_t)n_rows, 0), slot_unit((size_t)slots, 0);
    for (size_t v = 0; v < vtgt.size(); ++v) {
      if (vtgt[v] >= 0) row_unit[(size_t)vtgt[v]] = (int32_t)v;
      else if (vtgt[v] >= LOCAL_CODE + LSLOTS) slot_unit[(size_t)(~vtgt[v])] = (int32_t)v;  // a chunk that is one plain vrow
    }
    for (size_t li = 0; li < locals.size(); ++li) {
      if (locals[li].tgt >= 0) row_unit[(size_t)locals[li].tgt] = ~(int32_t)li;
      else slot_unit[(size_t)(~locals[li].tgt)] = ~(int32_t)li;
    }
    for (size_t i = 0; i < longs.size(); ++i) row_unit[(size_t)longs[i].row] = UNIT_LONG + (int32_t)i;
    rc = upload(&g->d_row_unit, row_unit);
    if (rc == IDG_OK) rc = upload(&g->d_slot_unit, slot_unit);
  }
  if (rc != IDG_OK) {
    idg_graph_destroy(g);
    return rc;
  }
  *out = g;
  return IDG_OK;
}

int idg_graph_create_from_device(int device, int64_t n_rows, int64_t n_cols, int64_t nnz, const int64_t* d_indptr,
                                 const int32_t* d_indices, const float* d_values, uint32_t flags, int64_t split_threshold,
                                 void* stream, idg_graph** out) {
  IDG_REQUIRE(out, "idg_graph_create_from_device: out is NULL");
  IDG_REQUIRE(n_rows >= 0 && nnz >= 0 && d_indptr && (nnz == 0 || (d_indices && d_values)),
              "idg_graph_create_from_device: bad argument");
  // The row-block schedule (virtual rows, tiles, XCD bands, split tables) depends on the ROW POINTER only: that array —
  // 8 (n + 1) bytes — is what visits the host (ordered after `stream`, where the caller may just have produced the CSR).
  // The column ids and values stay where they are: range check, the tiles' median columns and the layout of the
  // (col, val) list in tile order are kernels on `stream` (graph_build).  Measured at configs[4] size (15 M rows, 4e8
  // entries): see DESIGN.md §6.
  std::unique_ptr<int64_t[]> ip(new (std::nothrow) int64_t[(size_t)n_rows + 1]);
  if (!ip) return idg::fail(IDG_E_NOMEM, "idg_graph_create_from_device: out of host memory");
  hipStream_t st = (hipStream_t)stream;
  {
    DeviceGuard guard;
    int rc = guard.enter(device);
    if (rc != IDG_OK) return rc;
    IDG_HIP(hipMemcpyAsync(ip.get(), d_indptr, ((size_t)n_rows + 1) * sizeof(int64_t), hipMemcpyDeviceToHost, st));
    IDG_HIP(hipStreamSynchronize(st));
  }
  if (nnz == 0) {
    static const int32_t no_idx = 0;
    static const float no_val = 0.f;
    return graph_build(device, n_rows, n_cols, nnz, ip.get(), &no_idx, &no_val, nullptr, nullptr, st, flags, split_threshold, out);
  }
  return graph_build(device, n_rows, n_cols, nnz, ip.get(), nullptr, nullptr, d_indices, d_values, st, flags, split_threshold, out);
}

int idg_graph_create(int device, int64_t n_rows, int64_t n_cols, int64_t nnz, const int64_t* indptr,
                     const int32_t* indices, const float* values, uint32_t flags, int64_t split_threshold,
                     idg_graph** out) {
  IDG_REQUIRE(indptr && (nnz == 0 || (indices && values)), "idg_graph_create: NULL CSR array");
  static const int32_t no_idx = 0;
  static const float no_val = 0.f;
  return graph_build(device, n_rows, n_cols, nnz, indptr, nnz > 0 ? indices : &no_idx, nnz > 0 ? values : &no_val, nullptr,
                     nullptr, nullptr, flags, split_threshold, out);
}

int idg_graph_destroy(idg_graph* g) {
  if (!g) return IDG_OK;
  if (!g->borrowed && g->d_vptr) units_forget(g->d_vptr, nullptr);  // the schedule goes away: so do its unit lists
  if (g->d_cv) units_forget(g->d_cv, nullptr);                         // ... and the entry lists compacted from these values
  if (g->device >= 0 && idg_device_count() > g->device) {
    DeviceGuard guard;
    if (guard.enter(g->device) == IDG_OK) {
      (void)hipFree(g->d_cv);
      (void)hipFree(g->d_long_cnt);
      if (!g->borrowed) {
        (void)hipFree(g->d_vptr);
        (void)hipFree(g->d_vtgt);
        (void)hipFree(g->d_tiles);
        (void)hipFree(g->d_tiles_banded);
        (void)hipFree(g->d_tiles_seq);
        (void)hipFree(g->d_local);
        (void)hipFree(g->d_long);
        (void)hipFree(g->d_slot_row);
        (void)hipFree(g->d_slot_long);
        (void)hipFree(g->d_xl);
        (void)hipFree(g->d_vrow_row);
        (void)hipFree(g->d_row_unit);
        (void)hipFree(g->d_slot_unit);
      }
    }
  }
  delete g;
  return IDG_OK;
}

// values of a copy of the handle: entry (r, c) keeps v / divisor when floor(u(r, c) + add) != 0, u ~ U[0,1) from
// Philox4x32-10(seed; stream; r, c) — or u(c, r) for the transposed mask — and becomes an explicit zero otherwise
__global__ __launch_bounds__(BLOCK) void mask_values_kernel(const ColVal* __restrict__ src, ColVal* __restrict__ dst,
                                                            const int64_t* __restrict__ vptr,
                                                            const int32_t* __restrict__ vrow_row, int64_t n_vrows, int64_t nnz,
                                                            float add, float divisor, uint64_t seed, uint64_t stream_id,
                                                            int transpose) {
  const int64_t k = (int64_t)blockIdx.x * BLOCK + threadIdx.x;
  if (k >= nnz) return;
  int64_t lo = 0, hi = n_vrows;  // the vrow v with vptr[v] <= k < vptr[v + 1] (the entry list is in vrow order)
  while (hi - lo > 1) {
    const int64_t mid = (lo + hi) >> 1;
    if (vptr[mid] <= k) lo = mid;
    else hi = mid;
  }
  const uint32_t row = (uint32_t)vrow_row[lo];
  const ColVal e = src[k];
  const uint32_t i = transpose ? (uint32_t)e.col : row, j = transpose ? row : (uint32_t)e.col;
  const uint4 x = philox4x32_10(make_uint4(i, j, (uint32_t)stream_id, (uint32_t)(stream_id >> 32)),
                                make_uint2((uint32_t)seed, (uint32_t)(seed >> 32)));
  const float u = (x.x >> 8) * (1.0f / 16777216.0f);
  const bool keep = (int)(u + add) != 0;  // torch.rand(...) + (1 - keep_prob) truncated to int (models/NGCF.py:60-61)
  dst[k] = ColVal{e.col, keep ? e.val / divisor : 0.f};
}

// values of a copy of the handle, looked up in a CSR of the same structure: entry (r, c) of the handle's list takes
// values[k] where k is c's position in row r of (indptr, indices) (ascending columns: a binary search)
__global__ __launch_bounds__(BLOCK) void revalue_kernel(const ColVal* __restrict__ src, ColVal* __restrict__ dst,
                                                        const int64_t* __restrict__ vptr, const int32_t* __restrict__ vrow_row,
                                                        int64_t n_vrows, int64_t nnz, const int64_t* __restrict__ indptr,
                                                        const int32_t* __restrict__ indices, const float* __restrict__ values) {
  const int64_t k = (int64_t)blockIdx.x * BLOCK + threadIdx.x;
  if (k >= nnz) return;
  int64_t lo = 0, hi = n_vrows;
  while (hi - lo > 1) {
    const int64_t mid = (lo + hi) >> 1;
    if (vptr[mid] <= k) lo = mid;
    else hi = mid;
  }
  const int64_t r = vrow_row[lo];
  const int32_t c = src[k].col;
  int64_t a = indptr[r], b = indptr[r + 1];  // first position with indices[pos] >= c
  while (a < b) {
    const int64_t mid = (a + b) >> 1;
    if (indices[mid] < c) a = mid + 1;
    else b = mid;
  }
  dst[k] = ColVal{c, values[a]};
}

// kept(edge_of_entry[k]) ? dinv[row_of_entry[k]] * dinv[col_of_entry[k]] : 0 for every entry of a normalised bipartite
// adjacency (tools.create_adj_mat, tools.py:67-92: D^-1/2 (A' + A'^T) D^-1/2 on the kept interactions, float32)
__global__ __launch_bounds__(BLOCK) void subgraph_values_kernel(int64_t nnz, const int32_t* __restrict__ row_of_entry,
                                                                const int32_t* __restrict__ col_of_entry,
                                                                const int32_t* __restrict__ edge_of_entry,
                                                                const uint32_t* __restrict__ kept,
                                                                const float* __restrict__ dinv, float* __restrict__ out) {
  const int64_t k = (int64_t)blockIdx.x * BLOCK + threadIdx.x;
  if (k >= nnz) return;
  const int32_t e = edge_of_entry[k];
  const bool keep = (kept[e >> 5] >> (e & 31)) & 1u;
  // (d_r * 1) * d_c: the two float32 products of degree_matrix.dot(adjacency).dot(degree_matrix)
  out[k] = keep ? (dinv[row_of_entry[k]] * 1.0f) * dinv[col_of_entry[k]] : 0.f;
}

static int graph_copy_shell(const idg_graph* g, idg_graph** out, const char* who) {
  idg_graph* c = new (std::nothrow) idg_graph(*g);  // metadata and (borrowed) device pointers
  if (!c) return idg::fail(IDG_E_NOMEM, "%s: out of memory", who);
  c->borrowed = true;
  c->d_cv = nullptr;
  c->d_long_cnt = nullptr;
  int rc = IDG_OK;
  if (g->nnz > 0 && hipMalloc(reinterpret_cast<void**>(&c->d_cv), (size_t)g->nnz * sizeof(ColVal)) != hipSuccess)
    rc = idg::fail(IDG_E_NOMEM, "%s: hipMalloc of %lld entries failed", who, (long long)g->nnz);
  if (rc == IDG_OK) rc = upload(&c->d_long_cnt, std::vector<int>((size_t)g->n_long * MAX_PANELS, 0));
  if (rc != IDG_OK) {
    idg_graph_destroy(c);
    return rc;
  }
  *out = c;
  return IDG_OK;
}

int idg_subgraph_values_f32(int64_t nnz, const int32_t* row_of_entry, const int32_t* col_of_entry, const int32_t* edge_of_entry,
                            const uint32_t* kept_bits, const float* dinv, float* values_out, void* stream) {
  IDG_REQUIRE(nnz >= 0 && (nnz == 0 || (row_of_entry && col_of_entry && edge_of_entry && kept_bits && dinv && values_out)),
              "idg_subgraph_values_f32: NULL argument");
  if (nnz == 0) return IDG_OK;
  hipLaunchKernelGGL(subgraph_values_kernel, dim3((unsigned)((nnz + BLOCK - 1) / BLOCK)), dim3(BLOCK), 0, (hipStream_t)stream, nnz,
                     row_of_entry, col_of_entry, edge_of_entry, kept_bits, dinv, values_out);
  IDG_HIP(hipGetLastError());
  return IDG_OK;
}

int idg_graph_revalued_copy(const idg_graph* g, const int64_t* d_indptr, const int32_t* d_indices, const float* d_values,
                            void* stream, idg_graph** out) {
  IDG_REQUIRE(g && out, "idg_graph_revalued_copy: NULL argument");
  IDG_REQUIRE(g->nnz == 0 || (d_indptr && d_indices && d_values), "idg_graph_revalued_copy: NULL CSR array");
  IDG_REQUIRE(g->d_vrow_row || g->nnz == 0, "idg_graph_revalued_copy: handle without a vrow -> row table");
  DeviceGuard guard;
  int rc = guard.enter(g->device);
  if (rc != IDG_OK) return rc;
  idg_graph* c = nullptr;
  rc = graph_copy_shell(g, &c, "idg_graph_revalued_copy");
  if (rc != IDG_OK) return rc;
  if (g->nnz > 0)
    hipLaunchKernelGGL(revalue_kernel, dim3((unsigned)((g->nnz + BLOCK - 1) / BLOCK)), dim3(BLOCK), 0, (hipStream_t)stream,
                       g->d_cv, c->d_cv, g->d_vptr, g->d_vrow_row, g->n_vrows, g->nnz, d_indptr, d_indices, d_values);
  IDG_HIP(hipGetLastError());
  *out = c;
  return IDG_OK;
}

int idg_graph_remask(const idg_graph* g, idg_graph* copy, float add, float divisor, uint64_t seed, uint64_t stream_id,
                     int transpose, void* stream) {
  IDG_REQUIRE(g && copy && copy->borrowed && copy->nnz == g->nnz && copy->d_vptr == g->d_vptr,
              "idg_graph_remask: `copy` is not a masked / revalued copy of `g`");
  IDG_REQUIRE(divisor != 0.f, "idg_graph_remask: divisor must be non-zero");
  if (g->nnz > 0)
    hipLaunchKernelGGL(mask_values_kernel, dim3((unsigned)((g->nnz + BLOCK - 1) / BLOCK)), dim3(BLOCK), 0, (hipStream_t)stream,
                       g->d_cv, copy->d_cv, g->d_vptr, g->d_vrow_row, g->n_vrows, g->nnz, add, divisor, seed, stream_id, transpose);
  IDG_HIP(hipGetLastError());
  return IDG_OK;
}

int idg_graph_masked_copy(const idg_graph* g, float add, float divisor, uint64_t seed, uint64_t stream_id, int transpose,
                          void* stream, idg_graph** out) {
  IDG_REQUIRE(g && out, "idg_graph_masked_copy: NULL argument");
  IDG_REQUIRE(divisor != 0.f, "idg_graph_masked_copy: divisor must be non-zero");
  IDG_REQUIRE(!transpose || g->n_rows == g->n_cols, "idg_graph_masked_copy: the transposed mask needs a square graph");
  IDG_REQUIRE(g->d_vrow_row || g->nnz == 0, "idg_graph_masked_copy: handle without a vrow -> row table");
  DeviceGuard guard;
  int rc = guard.enter(g->device);
  if (rc != IDG_OK) return rc;
  idg_graph* c = nullptr;
  rc = graph_copy_shell(g, &c, "idg_graph_masked_copy");
  if (rc != IDG_OK) return rc;
  c->flags &= ~(uint32_t)IDG_GRAPH_SYMMETRIC;  // (r, c) and (c, r) are drawn independently
  if (g->nnz > 0)
    hipLaunchKernelGGL(mask_values_kernel, dim3((unsigned)((g->nnz + BLOCK - 1) / BLOCK)), dim3(BLOCK), 0, (hipStream_t)stream,
                       g->d_cv, c->d_cv, g->d_vptr, g->d_vrow_row, g->n_vrows, g->nnz, add, divisor, seed, stream_id, transpose);
  IDG_HIP(hipGetLastError());
  *out = c;
  return IDG_OK;
}

size_t idg_graph_live_units_bytes(const idg_graph* g, int64_t max_rows) {
  if (!g || max_rows < 0) return 0;
  return sizeof(int32_t) * (size_t)(UNITS_HEADER + max_rows + g->n_slots);
}

int idg_graph_bind_live_units(const idg_graph* g, const uint32_t* bitmap, const void* units_ws, int64_t max_rows) {
  IDG_REQUIRE(g && bitmap && units_ws && max_rows >= 0, "idg_graph_bind_live_units: bad argument");
  units_register(g->d_vptr, bitmap, reinterpret_cast<const int32_t*>(units_ws), max_rows + g->n_slots);
  return IDG_OK;
}

int idg_graph_forget_units_ws(const void* ws) {
  units_forget_ws(ws);
  return IDG_OK;
}

int idg_graph_forget_live_units(const idg_graph* g, const uint32_t* bitmap) {
  IDG_REQUIRE(g, "idg_graph_forget_live_units: NULL handle");
  units_forget(g->d_vptr, bitmap);
  units_forget(g->d_cv, bitmap);  // (compacted entry lists: idg_graph_compact_inputs)
  return IDG_OK;
}

static int live_units_impl(const idg_graph* g, const uint32_t* bitmap, void* units_ws, int64_t max_rows, void* stream, bool clear) {
  IDG_REQUIRE(g && bitmap && units_ws && max_rows >= 0, "idg_graph_live_units: bad argument");
  IDG_REQUIRE(g->d_row_unit || g->n_rows == 0, "idg_graph_live_units: handle without a row -> unit table");
  hipStream_t st = (hipStream_t)stream;
  if (clear) IDG_HIP(hipMemsetAsync(units_ws, 0, UNITS_HEADER * sizeof(int32_t), st));
  const int64_t words = (g->n_rows + 31) / 32;
  if (words > 0)
    hipLaunchKernelGGL(live_units_kernel, dim3((unsigned)((words + BLOCK - 1) / BLOCK)), dim3(BLOCK), 0, st, bitmap, g->n_rows,
                       g->d_row_unit, g->d_long, g->d_slot_unit, reinterpret_cast<int32_t*>(units_ws), max_rows + g->n_slots);
  IDG_HIP(hipGetLastError());
  return idg_graph_bind_live_units(g, bitmap, units_ws, max_rows);
}

int idg_graph_live_units(const idg_graph* g, const uint32_t* bitmap, void* units_ws, int64_t max_rows, void* stream) {
  return live_units_impl(g, bitmap, units_ws, max_rows, stream, true);
}

size_t idg_graph_compact_inputs_bytes(const idg_graph* g) {
  if (!g) return 0;
  // the compacted entry list (the tiles' offsets are kept), a compacted start per vrow, a live count per tile (indexed by
  // the tile's first vrow)
  return (size_t)g->nnz * sizeof(ColVal) + 2 * (size_t)g->n_vrows * sizeof(int32_t) + 16;
}

int idg_graph_compact_inputs(const idg_graph* g, const uint32_t* bitmap, void* ws, void* stream) {
  IDG_REQUIRE(g && bitmap && ws, "idg_graph_compact_inputs: bad argument");
  IDG_REQUIRE((uintptr_t)ws % 16 == 0, "idg_graph_compact_inputs: workspace must be 16-byte aligned");
  if (g->n_tiles == 0 || g->n_xl > 0) return IDG_OK;  // (EXACT_ORDER handles with rows beyond a tile keep the in-kernel form)
  ColVal* ccv = reinterpret_cast<ColVal*>(ws);
  int32_t* cptr = reinterpret_cast<int32_t*>(ccv + g->nnz);
  int32_t* clive = cptr + g->n_vrows;
  hipLaunchKernelGGL(compact_inputs_kernel, dim3((unsigned)g->n_tiles), dim3(BLOCK), 0, (hipStream_t)stream, g->d_tiles, g->d_vptr,
                     g->d_cv, bitmap, ccv, cptr, clive);
  IDG_HIP(hipGetLastError());
  units_register(g->d_cv, bitmap, reinterpret_cast<const int32_t*>(ws), g->nnz, 1);  // (keyed by the ENTRY list: copies of a handle share its schedule, not its values)
  return IDG_OK;
}

int idg_graph_live_units_check(const void* units_ws, void* stream) {
  IDG_REQUIRE(units_ws, "idg_graph_live_units_check: NULL list");
  int32_t head[UNITS_HEADER] = {0, 0};
  IDG_HIP(hipMemcpyAsync(head, units_ws, sizeof head, hipMemcpyDeviceToHost, (hipStream_t)stream));
  IDG_HIP(hipStreamSynchronize((hipStream_t)stream));
  if (head[1] != 0)
    return idg::fail(IDG_E_INVALID, "idg_graph_live_units: the bitmap holds more rows than max_rows allows (%d work units "
                     "found): the list is incomplete and launches that use it poison their output", (int)head[0]);
  return IDG_OK;
}

int idg_graph_info(const idg_graph* g, int64_t info[8]) {
  IDG_REQUIRE(g && info, "idg_graph_info: NULL argument");
  info[0] = g->n_rows;
  info[1] = g->n_cols;
  info[2] = g->nnz;
  info[3] = g->n_tiles + g->n_xl;
  info[4] = g->n_split;
  info[5] = g->n_slots;
  info[6] = g->split_threshold;
  info[7] = g->flags;
  return IDG_OK;
}

int idg_graph_long_rows(const idg_graph* g, int64_t* long_rows, int64_t* seg_len, int64_t* chunk_len) {
  IDG_REQUIRE(g, "idg_graph_long_rows: NULL handle");
  for (size_t i = 0; i < g->h_long_rows.size(); ++i) {
    if (long_rows) long_rows[i] = g->h_long_rows[i];
    if (seg_len) seg_len[i] = g->h_seg_len[i];
    if (chunk_len) chunk_len[i] = g->h_chunk_len[i];
  }
  return IDG_OK;
}

size_t idg_spmm_workspace_bytes(const idg_graph* g, int64_t d) {
  if (!g || d <= 0) return 0;
  return (size_t)g->n_slots * (size_t)d * sizeof(float);
}

int idg_spmm_f32(const idg_graph* g, const float* X, int64_t ldx, float* Y, int64_t ldy, const float* addend,
                 int64_t d, void* ws, void* stream) {
  IDG_REQUIRE(g && X && Y, "idg_spmm_f32: NULL argument");
  IDG_REQUIRE(d > 0 && ldx >= d && ldy >= d, "idg_spmm_f32: bad d/ldx/ldy (%lld,%lld,%lld)", (long long)d,
              (long long)ldx, (long long)ldy);
  Epilogue ep{};
  ep.Y = Y;
  ep.addend = addend;
  ep.ldy = ldy;
  ep.div = 1.0f;
  return spmm_dispatch(g, X, ldx, d, ws, ep, (hipStream_t)stream);
}

int idg_spmm_ex_f32(const idg_graph* g, const float* X, int64_t ldx, float* Y, const float* addend,
                    const float* sum_in, float* sum_out, int64_t ldy, float div, int accumulate,
                    const uint32_t* out_rows, const uint32_t* x_rows, int64_t d, void* ws, void* stream) {
  IDG_REQUIRE(g && X && (Y || sum_out), "idg_spmm_ex_f32: NULL argument");
  IDG_REQUIRE(d > 0 && ldx >= d && ldy >= d, "idg_spmm_ex_f32: bad d/ldx/ldy (%lld,%lld,%lld)", (long long)d,
              (long long)ldx, (long long)ldy);
  IDG_REQUIRE(div != 0.0f, "idg_spmm_ex_f32: div must be non-zero");
  Epilogue ep{};
  ep.Y = Y;
  ep.addend = addend;
  ep.sum_in = sum_in;
  ep.sum_out = sum_out;
  ep.ldy = ldy;
  ep.div = div;
  ep.accumulate = accumulate;
  return spmm_dispatch(g, X, ldx, d, ws, ep, (hipStream_t)stream, x_rows, out_rows);
}

static void adam_constants(Epilogue& ep, float* p, float* m, float* v, double lr, double beta1, double beta2, double eps,
                           int64_t step) {
  // scalars in double on the host, exactly as idg_adam_step_f32 (and torch/optim/adam.py) forms them
  const double bc1 = 1.0 - std::pow(beta1, (double)step);
  const double bc2 = 1.0 - std::pow(beta2, (double)step);
  ep.adam_p = p, ep.adam_m = m, ep.adam_v = v;
  ep.adam_w1 = (float)(1.0 - beta1), ep.adam_beta2 = (float)beta2, ep.adam_w2 = (float)(1.0 - beta2);
  ep.adam_step_size = (float)(lr / bc1), ep.adam_bc2_sqrt = (float)std::sqrt(bc2), ep.adam_eps = (float)eps;
}

int idg_spmm_epi_f32(const idg_graph* g, const float* X, int64_t ldx, int64_t d, const idg_epilogue* e,
                     const uint32_t* out_rows, const uint32_t* x_rows, void* ws, void* stream) {
  IDG_REQUIRE(g && X && e && (e->Y || e->sum_out || e->y24), "idg_spmm_epi_f32: NULL argument");
  IDG_REQUIRE(d > 0 && ldx >= d && e->ldy >= d, "idg_spmm_epi_f32: bad d/ldx/ldy (%lld,%lld,%lld)", (long long)d,
              (long long)ldx, (long long)e->ldy);
  IDG_REQUIRE(e->div != 0.0f, "idg_spmm_epi_f32: div must be non-zero");
  IDG_REQUIRE(e->sum_in || (!e->sum_in2 && !e->sum_in3), "idg_spmm_epi_f32: sum_in2 / sum_in3 need sum_in");
  Epilogue ep{};
  ep.Y = e->Y;
  ep.addend = e->addend;
  ep.sum_in = e->sum_in, ep.sum_in2 = e->sum_in2, ep.sum_in3 = e->sum_in3;
  ep.sum_out = e->sum_out;
  ep.ldy = e->ldy;
  ep.div = e->div;
  ep.accumulate = e->accumulate;
  ep.mask = e->mask;
  if (e->act) {
    IDG_REQUIRE(e->act == IDG_ACT_TANH || e->act == IDG_ACT_TANH_BWD, "idg_spmm_epi_f32: act is 0, IDG_ACT_TANH or IDG_ACT_TANH_BWD");
    IDG_REQUIRE(e->act != IDG_ACT_TANH_BWD || (e->act_src && (uintptr_t)e->act_src % 16 == 0),
                "idg_spmm_epi_f32: IDG_ACT_TANH_BWD needs act_src (the saved tanh outputs, 16-byte aligned)");
    IDG_REQUIRE(e->act_rows >= 0 && !e->adam_param, "idg_spmm_epi_f32: act_rows >= 0; an activation and Adam do not share a launch");
    IDG_REQUIRE(!(out_rows && x_rows), "idg_spmm_epi_f32: an activation with out_rows AND x_rows is not built");
    ep.act = e->act, ep.act_src = e->act_src, ep.act_rows = e->act_rows;
  }
  if (e->y24) {
    IDG_REQUIRE(!e->adam_param && !out_rows, "idg_spmm_epi_f32: the packed output lives in dense launches without the Adam epilogue");
    IDG_REQUIRE(e->ldy % 4 == 0 && ldx % 4 == 0 && (uintptr_t)X % 16 == 0 && (d == 32 || d == 64 || d == 128 || d == 256 || d == 512),
                "idg_spmm_epi_f32: the packed output needs a tiled width (32 .. 512) and 16-byte aligned panels");
    ep.y24 = e->y24;
    if (!ep.act) ep.act = 3;  // (no activation: the EPI_ACT instantiation is selected for its packed store)
  }
  if (e->adam_param) {
    IDG_REQUIRE(e->sum_out && e->adam_exp_avg && e->adam_exp_avg_sq && e->adam_step >= 1,
                "idg_spmm_epi_f32: the Adam epilogue needs sum_out (the gradient), both moments and a 1-based step");
    IDG_REQUIRE(!out_rows && !x_rows, "idg_spmm_epi_f32: the Adam epilogue lives in the dense launch (no out_rows / x_rows)");
    IDG_REQUIRE(((uintptr_t)e->adam_param | (uintptr_t)e->adam_exp_avg | (uintptr_t)e->adam_exp_avg_sq) % 16 == 0,
                "idg_spmm_epi_f32: Adam panels must be 16-byte aligned");
    adam_constants(ep, e->adam_param, e->adam_exp_avg, e->adam_exp_avg_sq, e->adam_lr, e->adam_beta1, e->adam_beta2,
                   e->adam_eps, e->adam_step);
    ep.adam_discard = e->adam_discard_grad != 0;
  }
  return spmm_dispatch(g, X, ldx, d, ws, ep, (hipStream_t)stream, x_rows, out_rows);
}

int idg_rows_tanh_bwd_f32(const float* grad, const float* y, const uint32_t* rows, int64_t n, int64_t d, float* out,
                          void* stream) {
  IDG_REQUIRE(grad && y && out && n >= 0 && d > 0 && d % 4 == 0, "idg_rows_tanh_bwd_f32: bad argument (d must be a multiple of 4)");
  IDG_REQUIRE(((uintptr_t)grad | (uintptr_t)y | (uintptr_t)out) % 16 == 0, "idg_rows_tanh_bwd_f32: panels must be 16-byte aligned");
  if (n == 0) return IDG_OK;
  const int64_t d4 = d / 4, total = n * d4;
  hipLaunchKernelGGL(rows_tanh_bwd_kernel, dim3((unsigned)((total + BLOCK - 1) / BLOCK)), dim3(BLOCK), 0, (hipStream_t)stream, grad,
                     y, rows, n, d4, out);
  IDG_HIP(hipGetLastError());
  return IDG_OK;
}

int idg_spmm_noise_f32(const idg_graph* g, const float* X, int64_t ldx, float* Y, int64_t ldy, const uint32_t* out_rows,
                       int64_t d, float eps, uint64_t seed, uint64_t stream_id, void* ws, void* stream) {
  IDG_REQUIRE(g && X && Y, "idg_spmm_noise_f32: NULL argument");
  IDG_REQUIRE(d > 0 && ldx >= d && ldy >= d, "idg_spmm_noise_f32: bad d/ldx/ldy");
  Epilogue ep{};
  ep.Y = Y;
  ep.ldy = ldy;
  ep.div = 1.0f;
  ep.noise_eps = eps;
  ep.noise_seed = seed;
  ep.noise_stream = stream_id;
  return spmm_dispatch(g, X, ldx, d, ws, ep, (hipStream_t)stream, nullptr, out_rows);
}

int idg_perturb_f32(const float* X, float* Y, int64_t n, int64_t d, const uint32_t* rows, float eps, uint64_t seed,
                    uint64_t stream_id, void* stream) {
  IDG_REQUIRE(X && Y && n >= 0 && d > 0, "idg_perturb_f32: NULL argument");
  if (n == 0) return IDG_OK;
  Epilogue ep{};
  ep.noise_eps = eps;
  ep.noise_seed = seed;
  ep.noise_stream = stream_id;
  hipStream_t st = (hipStream_t)stream;
#define IDG_PERTURB(LPR, NB)                                                                                      \
  hipLaunchKernelGGL((perturb_rows_kernel<LPR, NB>), dim3((unsigned)((n + BLOCK / LPR - 1) / (BLOCK / LPR))),     \
                     dim3(BLOCK), 0, st, X, Y, n, d, ep, rows)
  const bool tiled = (d == 32 || d == 64 || d == 128 || d == 256 || d == 512) && ((uintptr_t)X | (uintptr_t)Y) % 16 == 0;
  if (!tiled) {  // any width / alignment: one wave per row
    hipLaunchKernelGGL(perturb_rows_generic_kernel, dim3((unsigned)((n + BLOCK / 64 - 1) / (BLOCK / 64))), dim3(BLOCK), 0, st, X,
                       Y, n, d, ep, rows);
  } else switch (d) {
    case 32: IDG_PERTURB(8, 1); break;
    case 64: IDG_PERTURB(16, 1); break;
    case 128: IDG_PERTURB(32, 1); break;
    case 256: IDG_PERTURB(64, 1); break;
    default: IDG_PERTURB(64, 2); break;
  }
#undef IDG_PERTURB
  IDG_HIP(hipGetLastError());
  return IDG_OK;
}

size_t idg_propagate_workspace_bytes(const idg_graph* g, int64_t d) {
  if (!g || d <= 0) return 0;
  // two ping-pong panels + the split-row partials
  const size_t panel = ((size_t)std::max(g->n_rows, g->n_cols) * (size_t)d * sizeof(float) + 255) / 256 * 256;
  return 2 * panel + idg_spmm_workspace_bytes(g, d);
}

static int propagate_common(const idg_graph* g, const float* in, float* out, int K, int include0, int64_t d,
                            void* ws, hipStream_t st, bool backward, int accumulate,
                            const uint32_t* in_mask = nullptr, float noise_eps = 0.f, uint64_t noise_seed = 0,
                            uint64_t noise_stream = 0, const uint32_t* out_rows = nullptr,
                            const Epilogue* adam = nullptr, const uint32_t* const* fields = nullptr) {
  IDG_REQUIRE(g && in && out && ws, "idg_propagate: NULL argument");
  IDG_REQUIRE(g->n_rows == g->n_cols, "idg_propagate: graph must be square");
  IDG_REQUIRE(K >= 1, "idg_propagate: K must be >= 1 (got %d)", K);
  IDG_REQUIRE(d > 0, "idg_propagate: d must be > 0");
  IDG_REQUIRE(in != out, "idg_propagate: in-place propagation is not supported");
  const size_t panel = ((size_t)g->n_rows * (size_t)d * sizeof(float) + 255) / 256 * 256;
  float* P[2] = {reinterpret_cast<float*>(ws), reinterpret_cast<float*>((char*)ws + panel)};
  void* partials = (char*)ws + 2 * panel;
  const float cnt = (float)(K + (include0 ? 1 : 0));
  const float* X = in;
  for (int k = 1; k <= K; ++k) {
    Epilogue ep{};
    ep.ldy = d;
    ep.div = 1.0f;
    const bool last = (k == K);
    if (!backward && noise_eps != 0.f) {
      ep.noise_eps = noise_eps;
      ep.noise_seed = noise_seed;
      ep.noise_stream = noise_stream * 64 + (uint64_t)k;  // a fresh stream per layer
    }
    if (!backward && K <= 3) {
      // forward, up to three layers: the layer outputs X_1..X_{K-1} stay in the two layer buffers, and only the LAST
      // product's epilogue forms the mean, (((E0 + X_1) + X_2) + X_K) / cnt — torch.mean(torch.stack(...))'s order — so
      // the earlier layers are plain products (no running-sum panel read and written per layer) and, with out_rows,
      // the sum is formed for the requested rows only.
      if (!last) {
        ep.Y = P[(k - 1) & 1];
      } else {
        const float* terms[3] = {nullptr, nullptr, nullptr};
        int nt = 0;
        if (include0) terms[nt++] = in;
        for (int j = 1; j < K; ++j) terms[nt++] = P[(j - 1) & 1];
        ep.sum_in = terms[0], ep.sum_in2 = terms[1], ep.sum_in3 = terms[2];
        ep.sum_out = out;
        ep.div = cnt;
      }
    } else if (!backward) {
      // forward, deeper stacks: running sum lives in `out`; the last layer divides.
      if (!last) ep.Y = P[(k - 1) & 1];
      if (k == 1) {
        if (include0 || last) {
          ep.sum_in = include0 ? in : nullptr;
          ep.sum_out = out;
        }
      } else {
        // sum so far = out (include0, or k > 2) or the layer input X_1 (k == 2, no layer 0)
        ep.sum_in = (include0 || k > 2) ? out : X;
        ep.sum_out = out;
      }
      if (last) ep.div = cnt;
    } else {
      // backward Horner step: h <- A.h + g ; the last one scales by 1/cnt.  `in_mask` flags the
      // live rows of g: the first product gathers from g itself, every step adds g.
      ep.mask = in_mask;
      if (!last) {
        ep.Y = P[(k - 1) & 1];
        ep.addend = in;
      } else {
        ep.sum_in = include0 ? in : nullptr;
        ep.sum_out = out;
        ep.div = cnt;
        ep.accumulate = accumulate;
        if (adam) {
          ep.adam_p = adam->adam_p, ep.adam_m = adam->adam_m, ep.adam_v = adam->adam_v;
          ep.adam_w1 = adam->adam_w1, ep.adam_beta2 = adam->adam_beta2, ep.adam_w2 = adam->adam_w2;
          ep.adam_step_size = adam->adam_step_size, ep.adam_bc2_sqrt = adam->adam_bc2_sqrt, ep.adam_eps = adam->adam_eps;
          ep.adam_discard = adam->adam_discard;
        }
      }
    }
    // fields (idg_propagate_mean*_fields_f32): forward, the rows layer k has to produce; backward, the live rows of
    // step k's input (step 1: those of g) — the batch's receptive field, hop by hop
    const uint32_t* x_mask = backward ? (fields ? fields[k - 1] : (k == 1 ? in_mask : nullptr)) : nullptr;
    if (backward && last && adam) x_mask = nullptr;  // (the Adam epilogue lives in the dense kernel)
    const uint32_t* o_mask = backward ? nullptr : (fields ? fields[k - 1] : (last ? out_rows : nullptr));
    int rc = spmm_dispatch(g, X, d, d, partials, ep, st, x_mask, o_mask);
    if (rc != IDG_OK) return rc;
    X = P[(k - 1) & 1];
  }
  return IDG_OK;
}

// ---- clean pass + perturbed views with a shared first product and ONE multi-panel launch for the last layer ----
size_t idg_propagate_views_workspace_bytes(const idg_graph* g, int64_t d, int n_views) {
  if (!g || d <= 0 || n_views < 0 || n_views + 1 > MAX_PANELS) return 0;
  const size_t panel = ((size_t)std::max(g->n_rows, g->n_cols) * (size_t)d * sizeof(float) + 255) / 256 * 256;
  const size_t part = (idg_spmm_workspace_bytes(g, d) + 255) / 256 * 256;
  return (size_t)(n_views + 1) * (2 * panel + part);
}

int idg_propagate_views_f32(const idg_graph* g, const float* E0, int K, int64_t d, float eps, int n_views,
                            const uint64_t* seeds, const uint64_t* stream_ids, float* out_clean, float* const* out_views,
                            const uint32_t* out_rows, void* ws, void* stream) {
  IDG_REQUIRE(g && E0 && out_clean && ws, "idg_propagate_views_f32: NULL argument");
  IDG_REQUIRE(g->n_rows == g->n_cols, "idg_propagate_views_f32: graph must be square");
  IDG_REQUIRE(K >= 2, "idg_propagate_views_f32: needs K >= 2 (the first product is what the passes share)");
  IDG_REQUIRE(n_views >= 1 && n_views + 1 <= MAX_PANELS && seeds && stream_ids && out_views,
              "idg_propagate_views_f32: 1..%d views", MAX_PANELS - 1);
  IDG_REQUIRE(d == 32 || d == 64 || d == 128 || d == 256 || d == 512, "idg_propagate_views_f32: d must be a tiled width");
  hipStream_t st = (hipStream_t)stream;
  const int np = n_views + 1;
  const int64_t n = g->n_rows;
  const size_t panel = ((size_t)n * (size_t)d * sizeof(float) + 255) / 256 * 256;
  const size_t part = (idg_spmm_workspace_bytes(g, d) + 255) / 256 * 256;
  char* base = reinterpret_cast<char*>(ws);
  float* P[MAX_PANELS][2];
  float* partials[MAX_PANELS];
  float* outs[MAX_PANELS];
  for (int p = 0; p < np; ++p) {
    P[p][0] = reinterpret_cast<float*>(base + (size_t)p * (2 * panel + part));
    P[p][1] = reinterpret_cast<float*>(base + (size_t)p * (2 * panel + part) + panel);
    partials[p] = reinterpret_cast<float*>(base + (size_t)p * (2 * panel + part) + 2 * panel);
    outs[p] = p == 0 ? out_clean : out_views[p - 1];
    IDG_REQUIRE(outs[p] && (uintptr_t)outs[p] % 16 == 0, "idg_propagate_views_f32: output %d is NULL or unaligned", p);
  }
  IDG_REQUIRE((uintptr_t)E0 % 16 == 0, "idg_propagate_views_f32: E0 must be 16-byte aligned");
  // layer 1, shared: T = A.E0
  {
    Epilogue ep{};
    ep.ldy = d;
    ep.div = 1.0f;
    ep.Y = P[0][0];
    int rc = spmm_dispatch(g, E0, d, d, partials[0], ep, st);
    if (rc != IDG_OK) return rc;
  }
  for (int p = 1; p < np; ++p) {  // each view perturbs its own copy (sub-stream 0 of its stream)
    int rc = idg_perturb_f32(P[0][0], P[p][0], n, d, nullptr, eps, seeds[p - 1], stream_ids[p - 1] * 64, stream);
    if (rc != IDG_OK) return rc;
  }
  // layers 2..K: mean(X1..XK) = propagate_mean(X1, K - 1, include_layer0 = 1), per pass
  bool listed = false;  // a live-unit list for this bitmap: the per-panel launches take the one-wave-per-unit form
  if (out_rows && !g->no_units) listed = units_find(g->d_vptr, out_rows, nullptr, nullptr);
  const bool multi_ok = out_rows && !listed && g->n_tiles > 0 && g->n_xl == 0 && !g->no_fused_fix &&
                        (g->n_slots * d * (int64_t)sizeof(float)) < ((int64_t)1 << 31);
  for (int k = 2; k <= K; ++k) {
    const bool last = (k == K);
    MultiPanel mp{};
    for (int p = 0; p < np; ++p) {
      Epilogue ep{};
      ep.ldy = d;
      ep.div = last ? (float)K : 1.0f;
      const float* Xp = P[p][(k - 2) & 1];
      if (!last) ep.Y = P[p][(k - 1) & 1];
      if (K <= 3) {  // as in propagate_common: X_1 (and X_2) stay in the pass's two buffers, the last epilogue sums them
        if (last) {
          ep.sum_in = P[p][0];
          ep.sum_in2 = K == 3 ? P[p][1] : nullptr;
          ep.sum_out = outs[p];
        }
      } else {
        ep.sum_in = (k == 2) ? P[p][0] : outs[p];
        ep.sum_out = outs[p];
      }
      if (p > 0) {
        ep.noise_eps = eps;
        ep.noise_seed = seeds[p - 1];
        ep.noise_stream = stream_ids[p - 1] * 64 + (uint64_t)(k - 1);
      }
      mp.X[p] = Xp;
      mp.partials[p] = partials[p];
      mp.cnt[p] = g->d_long_cnt + (int64_t)p * g->n_long;
      mp.ep[p] = ep;
    }
    if (last && multi_ok) {
      int rc;
      switch (d) {
        case 32: rc = launch_rows_multi<8, 1>(g, np, mp, d, d, out_rows, false, st); break;
        case 64: rc = launch_rows_multi<16, 1>(g, np, mp, d, d, out_rows, false, st); break;
        case 128: rc = launch_rows_multi<32, 1>(g, np, mp, d, d, out_rows, false, st); break;
        case 256: rc = launch_rows_multi<64, 1>(g, np, mp, d, d, out_rows, false, st); break;
        default: rc = launch_rows_multi<64, 2>(g, np, mp, d, d, out_rows, false, st); break;
      }
      if (rc != IDG_OK) return rc;
    } else {
      for (int p = 0; p < np; ++p) {
        int rc = spmm_dispatch(g, mp.X[p], d, d, partials[p], mp.ep[p], st, nullptr, last ? out_rows : nullptr);
        if (rc != IDG_OK) return rc;
      }
    }
  }
  return IDG_OK;
}

int idg_propagate_mean_f32(const idg_graph* g, const float* E0, float* out, const uint32_t* out_rows, int K,
                           int include_layer0, int64_t d, void* ws, void* stream) {
  return propagate_common(g, E0, out, K, include_layer0, d, ws, (hipStream_t)stream, false, 0, nullptr, 0.f, 0, 0,
                          out_rows);
}

int idg_propagate_mean_noise_f32(const idg_graph* g, const float* E0, float* out, const uint32_t* out_rows, int K,
                                 int include_layer0, int64_t d, float eps, uint64_t seed, uint64_t stream_id, void* ws,
                                 void* stream) {
  return propagate_common(g, E0, out, K, include_layer0, d, ws, (hipStream_t)stream, false, 0, nullptr, eps, seed, stream_id,
                          out_rows);
}

int idg_propagate_mean_bwd_f32(const idg_graph* g, const float* gout, const uint32_t* gout_mask, float* gE0, int K,
                               int include_layer0, int64_t d, int accumulate, void* ws, void* stream) {
  IDG_REQUIRE(g, "idg_propagate_mean_bwd_f32: NULL graph");
  IDG_REQUIRE(g->flags & IDG_GRAPH_SYMMETRIC,
              "idg_propagate_mean_bwd_f32: graph not flagged IDG_GRAPH_SYMMETRIC (build the transposed handle and "
              "chain idg_spmm_f32 instead)");
  return propagate_common(g, gout, gE0, K, include_layer0, d, ws, (hipStream_t)stream, true, accumulate, gout_mask);
}

static int bwd_adam_impl(const idg_graph* g, const float* gout, const uint32_t* gout_mask, float* gE0, int K,
                         int include_layer0, int64_t d, int accumulate, float* param, float* exp_avg, float* exp_avg_sq,
                         double lr, double beta1, double beta2, double eps, int64_t step, void* ws, void* stream,
                         const uint32_t* const* fields);

int idg_propagate_mean_bwd_adam_f32(const idg_graph* g, const float* gout, const uint32_t* gout_mask, float* gE0, int K,
                                    int include_layer0, int64_t d, int accumulate, float* param, float* exp_avg,
                                    float* exp_avg_sq, double lr, double beta1, double beta2, double eps, int64_t step,
                                    void* ws, void* stream) {
  return bwd_adam_impl(g, gout, gout_mask, gE0, K, include_layer0, d, accumulate, param, exp_avg, exp_avg_sq, lr, beta1, beta2,
                       eps, step, ws, stream, nullptr);
}

int idg_propagate_mean_fields_f32(const idg_graph* g, const float* E0, float* out, const uint32_t* const* layer_rows, int K,
                                  int include_layer0, int64_t d, void* ws, void* stream) {
  IDG_REQUIRE(layer_rows, "idg_propagate_mean_fields_f32: NULL layer_rows");
  IDG_REQUIRE(K <= 3, "idg_propagate_mean_fields_f32: up to three layers (the layer mean is formed by the last product)");
  return propagate_common(g, E0, out, K, include_layer0, d, ws, (hipStream_t)stream, false, 0, nullptr, 0.f, 0, 0,
                          layer_rows[K - 1], nullptr, layer_rows);
}

int idg_propagate_mean_bwd_adam_fields_f32(const idg_graph* g, const float* gout, const uint32_t* const* step_rows, float* gE0,
                                           int K, int include_layer0, int64_t d, int accumulate, float* param,
                                           float* exp_avg, float* exp_avg_sq, double lr, double beta1, double beta2,
                                           double eps, int64_t step, void* ws, void* stream) {
  IDG_REQUIRE(step_rows && step_rows[0], "idg_propagate_mean_bwd_adam_fields_f32: NULL step_rows");
  return bwd_adam_impl(g, gout, step_rows[0], gE0, K, include_layer0, d, accumulate, param, exp_avg, exp_avg_sq, lr, beta1,
                       beta2, eps, step, ws, stream, step_rows);
}

int idg_graph_expand_rows(const idg_graph* g, const uint32_t* in_rows, uint32_t* out_rows, void* stream) {
  IDG_REQUIRE(g && in_rows && out_rows && in_rows != out_rows, "idg_graph_expand_rows: bad argument");
  IDG_REQUIRE(g->n_rows == g->n_cols, "idg_graph_expand_rows: graph must be square");
  IDG_REQUIRE(g->d_vrow_row || g->nnz == 0, "idg_graph_expand_rows: handle without a vrow -> row table");
  idg::rows_changed(out_rows, (size_t)((g->n_rows + 31) / 32) * sizeof(uint32_t));
  hipStream_t st = (hipStream_t)stream;
  IDG_HIP(hipMemcpyAsync(out_rows, in_rows, (size_t)((g->n_rows + 31) / 32) * sizeof(uint32_t), hipMemcpyDeviceToDevice, st));
  if (g->n_vrows > 0)
    hipLaunchKernelGGL(expand_rows_kernel, dim3((unsigned)((g->n_vrows + BLOCK - 1) / BLOCK)), dim3(BLOCK), 0, st, g->n_vrows,
                       g->d_vptr, g->d_vrow_row, g->d_cv, in_rows, out_rows);
  IDG_HIP(hipGetLastError());
  return IDG_OK;
}

int idg_graph_mark_cols(const idg_graph* g, const uint32_t* in_rows, uint32_t* col_bits, void* stream) {
  IDG_REQUIRE(g && in_rows && col_bits, "idg_graph_mark_cols: bad argument");
  IDG_REQUIRE(g->d_vrow_row || g->nnz == 0, "idg_graph_mark_cols: handle without a vrow -> row table");
  idg::rows_changed(col_bits, (size_t)((g->n_cols + 31) / 32) * sizeof(uint32_t));
  if (g->n_vrows > 0)
    hipLaunchKernelGGL(expand_rows_kernel, dim3((unsigned)((g->n_vrows + BLOCK - 1) / BLOCK)), dim3(BLOCK), 0, (hipStream_t)stream,
                       g->n_vrows, g->d_vptr, g->d_vrow_row, g->d_cv, in_rows, col_bits);
  IDG_HIP(hipGetLastError());
  return IDG_OK;
}

int idg_graph_flag_cols(const idg_graph* g, const uint32_t* in_rows, float* col_flags, void* stream) {
  IDG_REQUIRE(g && in_rows && col_flags, "idg_graph_flag_cols: bad argument");
  IDG_REQUIRE(g->d_vrow_row || g->nnz == 0, "idg_graph_flag_cols: handle without a vrow -> row table");
  if (g->n_vrows > 0)
    hipLaunchKernelGGL(flag_cols_kernel, dim3((unsigned)((g->n_vrows + BLOCK - 1) / BLOCK)), dim3(BLOCK), 0, (hipStream_t)stream,
                       g->n_vrows, g->d_vptr, g->d_vrow_row, g->d_cv, in_rows, col_flags);
  IDG_HIP(hipGetLastError());
  return IDG_OK;
}

static int bwd_adam_impl(const idg_graph* g, const float* gout, const uint32_t* gout_mask, float* gE0, int K,
                         int include_layer0, int64_t d, int accumulate, float* param, float* exp_avg, float* exp_avg_sq,
                         double lr, double beta1, double beta2, double eps, int64_t step, void* ws, void* stream,
                         const uint32_t* const* fields) {
  IDG_REQUIRE(g, "idg_propagate_mean_bwd_adam_f32: NULL graph");
  IDG_REQUIRE(g->flags & IDG_GRAPH_SYMMETRIC, "idg_propagate_mean_bwd_adam_f32: graph not flagged IDG_GRAPH_SYMMETRIC");
  IDG_REQUIRE(param && exp_avg && exp_avg_sq && gE0, "idg_propagate_mean_bwd_adam_f32: NULL argument");
  IDG_REQUIRE(step >= 1, "idg_propagate_mean_bwd_adam_f32: step is 1-based");
  IDG_REQUIRE(((uintptr_t)param | (uintptr_t)exp_avg | (uintptr_t)exp_avg_sq | (uintptr_t)gE0) % 16 == 0,
              "idg_propagate_mean_bwd_adam_f32: pointers must be 16-byte aligned");
  // The update rides in the epilogue of the LAST product when that is a launch of the dense tiled kernel
  // (K >= 2, tiled widths); otherwise the two steps simply run one after the other.  Same bits either way.
  const bool tiled = (d == 32 || d == 64 || d == 128 || d == 256 || d == 512) && (uintptr_t)gout % 16 == 0;
  // IDG_ADAM_DISCARD_GRAD OR-ed into `accumulate`: the finished gradient feeds the update and is not written back to gE0
  // (whose rows are then only an INPUT: the rows the loss kernel stored for this step) — 4 B per element less to store
  const bool discard = (accumulate & IDG_ADAM_DISCARD_GRAD) != 0;
  accumulate &= ~IDG_ADAM_DISCARD_GRAD;
  if (K < 2 || !tiled) {
    const int rc = idg_propagate_mean_bwd_f32(g, gout, gout_mask, gE0, K, include_layer0, d, accumulate, ws, stream);
    if (rc != IDG_OK) return rc;
    return idg_adam_step_f32(param, gE0, exp_avg, exp_avg_sq, g->n_rows * d, lr, beta1, beta2, eps, step, stream);
  }
  Epilogue adam{};
  adam_constants(adam, param, exp_avg, exp_avg_sq, lr, beta1, beta2, eps, step);
  adam.adam_discard = discard ? 1 : 0;
  return propagate_common(g, gout, gE0, K, include_layer0, d, ws, (hipStream_t)stream, true, accumulate, gout_mask, 0.f, 0, 0,
                          nullptr, &adam, fields);
}

}  // extern "C"

int idg::live_units_prezeroed(const idg_graph* g, const uint32_t* bitmap, void* units_ws, int64_t max_rows, void* stream) {
  return live_units_impl(g, bitmap, units_ws, max_rows, stream, false);
}
