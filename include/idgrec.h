/*
 * idgrec.h — C-ABI of libidgrec.so, the MI355X (gfx950) implementation of ID-GRec's
 * LightGCN-style hot path: BPR sampler -> K-layer normalised-adjacency SpMM ->
 * fused gather-BPR loss/grad -> dense Adam, and full-rank scoring + mask + top-K.
 *
 * The reference (BlueGhostYi/ID-GRec) is pure Python; the "FFI" a maintainer binds is
 * ctypes.  Every entry point below names the reference call site
 * (path:line relative to the reference repo) whose arithmetic it replaces.
 *
 * Conventions
 *   - every function returns int: 0 = ok, <0 = error class (IDG_E_*); the message of the
 *     last failure on the calling thread is idg_last_error().
 *   - "host" functions take host pointers and never touch the GPU.
 *   - "device" functions take DEVICE pointers owned by the caller (torch tensors'
 *     data_ptr()), are asynchronous on the hipStream_t passed as `void* stream`
 *     (NULL = the null stream), never allocate, never synchronise.  Scratch memory is
 *     caller-allocated; sizes come from the matching *_workspace_bytes().
 *   - opaque handles (idg_rng, idg_ratings, idg_graph) are owned by the library and
 *     released by the paired *_destroy().
 *   - no C++ exception crosses this boundary.  There is NO CPU fallback for device
 *     entry points: without a gfx950 device they fail with IDG_E_NODEVICE.
 */
#ifndef IDGREC_H
#define IDGREC_H

#include <stddef.h>
#include <stdint.h>

#ifdef __cplusplus
extern "C" {
#endif

#define IDG_VERSION 140 /* 0.5.1: idg_pack24_f32 / idg_unpack24_f32 / idg_reduce24_f32 / idg_alltoall_f32 (24-bit panel exchange, rank-ordered sum), idg_score_topk_candidate_counts, idg_score_topk_option (the top-K knobs: environment read once), idg_score_topk_info fills info[8], form 3's whole-call fall-back; idg_step_run_f32 takes next_ids_token; idg_step_synchronize also drains the side stream's preparations; 0.5.0: idg_step_* (one library call per training step), idg_adam_rows_f32; 0.4.4: IDG_ADAM_DISCARD_GRAD; 0.4.3: idg_event_synchronize; 0.4.2: idg_infonce_plan / IDG_SSL_PLANNED / idg_infonce_cross_ex_f32 (InfoNCE id lists a batch ahead); 0.4.1: idg_ngcf_layer_fwd_f32 / idg_ngcf_layer_bwd_f32 (one kernel per NGCF layer and direction); 0.4.0: idg_rows_layer_mean_n_f32 (any number of layers), idg_flags_compact_f32 (the touched-item
                           agreement without a host read-back), idg_shard_prepare validates its geometry.
                           133 / 0.3.0: process-wide live-unit registry + idg_graph_live_units_check; idg_spmm_epi_f32 (every
                           epilogue option; out_rows and x_rows combined); round-3 sharded step: idg_rows_gather2 / _scatter /
                           _chain_store2 / _layer_mean, idg_grad_tail_adam_f32, idg_reduce_scatter_f32 */

/* error classes */
#define IDG_OK 0
#define IDG_E_INVALID (-1)  /* bad argument */
#define IDG_E_NODEVICE (-2) /* no HIP device / wrong device */
#define IDG_E_HIP (-3)      /* a HIP runtime call failed */
#define IDG_E_IO (-4)       /* file could not be read / parsed */
#define IDG_E_NOMEM (-5)
#define IDG_E_UNSUPPORTED (-6)

int idg_version(void);
const char* idg_last_error(void);
/* number of visible HIP devices (0 when none). Never fails. */
int idg_device_count(void);

/* ------------------------------------------------------------------------------------
 * HOST: NumPy-legacy MT19937 stream  (replaces the global np.random state the reference
 * seeds in utility/utility_function/tools.py:8-14 and draws from in
 * utility/utility_data/data_loader.py:120 and utility/utility_function/tools.py:42)
 * ---------------------------------------------------------------------------------- */
typedef struct idg_rng idg_rng;

/* np.random.seed(seed) : init_genrand(seed), pos = 624 */
int idg_rng_create(uint32_t seed, idg_rng** out);
int idg_rng_destroy(idg_rng* rng);
/* mirrors np.random.get_state()[1:3] / set_state */
int idg_rng_get_state(const idg_rng* rng, uint32_t key[624], int32_t* pos);
int idg_rng_set_state(idg_rng* rng, const uint32_t key[624], int32_t pos);
/* RandomState.bytes(n): raw little-endian uint32 stream */
int idg_rng_bytes(idg_rng* rng, int64_t nbytes, uint8_t* out);
/* `count` successive scalar np.random.randint(0, high) draws (masked rejection) */
int idg_rng_randint(idg_rng* rng, int64_t high, int64_t count, int64_t* out);

/* Data.sample_data_to_train_all (utility/utility_data/data_loader.py:108-127):
 * for every train edge i in file order emit [train_user[i], train_item[i], neg] with
 * neg = randint(0,num_items) re-drawn while neg is in the user's positive set.
 * pos_indptr[num_users+1] / pos_indices (ascending per user) are the CSR rows of the
 * train matrix (Data.all_positive).  Users with no positives are skipped (:114-115).
 * out_triples is [E,3] row-major int64; *out_count = rows written (<= E). */
int idg_sample_epoch(idg_rng* rng, const int64_t* train_user, const int64_t* train_item, int64_t E,
                     const int64_t* pos_indptr, const int32_t* pos_indices, int64_t num_users,
                     int64_t num_items, int64_t* out_triples, int64_t* out_count);

/* np.random.shuffle(np.arange(n)) (utility/utility_function/tools.py:41-42) */
int idg_shuffle_perm(idg_rng* rng, int64_t n, int64_t* out_perm);

/* random.sample(range(n), k) of Python's `random` module (CPython >= 3.10) on its own MT19937 stream — the
 * draw of SGL's kept edges (utility/utility_function/tools.py:80, never seeded by the reference).  Load the
 * module's state with idg_rng_set_state (random.getstate()[1]: 624 words + index), write it back afterwards.
 * use_pool: the branch random.py takes (n <= 21 + 4 ** ceil(log(3k, 4)) for k > 5), decided by the caller
 * with that very expression.  n < 2^32. */
int idg_py_random_sample(idg_rng* rng, int64_t n, int64_t k, int use_pool, int64_t* out);

/* ------------------------------------------------------------------------------------
 * HOST: rating-file parser and adjacency builder
 * ---------------------------------------------------------------------------------- */
typedef struct idg_ratings idg_ratings;

/* Data.read_ratings (utility/utility_data/data_loader.py:48-70): parse lines
 * "uid i1 i2 ..." (space separated ints).  Lines with no items count in n_lines but emit
 * no edge and do not move the maxima (:59-61).  max_user/max_item are -1 when no edge. */
int idg_ratings_open(const char* path, idg_ratings** out, int64_t* n_edges, int64_t* n_lines,
                     int64_t* max_user, int64_t* max_item);
/* users/items [n_edges] in file order; line_users / line_counts [n_lines] = the user id and
 * the number of items of every line.  Any output may be NULL. */
int idg_ratings_read(const idg_ratings* r, int64_t* users, int64_t* items, int64_t* line_users,
                     int64_t* line_counts);
int idg_ratings_destroy(idg_ratings* r);

/* data_graph.sparse_adjacency_matrix (utility/utility_data/data_graph.py:33-55) and, with
 * self_loops != 0, sparse_adjacency_matrix_with_self (:7-30): the symmetric bipartite
 * adjacency A=[[0,R],[R^T,0]] (+I), duplicate (u,i) pairs summed, normalised
 * D^-1/2 A D^-1/2, as CSR with ascending columns and float32 values.  Arithmetic follows
 * the reference: float32 products (d_i*a)*d_j without self loops; float64 products rounded
 * once to float32 with them (the `+ sp.eye` promotes to float64).
 * dinv (nullable, [U+I]): d^-1/2 per node as the caller computed it (the Python host passes
 * np.power(deg, -0.5), the reference's own expression, whose SIMD rounding cannot be
 * restated portably); NULL = correctly rounded 1/sqrt(deg), 0 for isolated nodes.
 * Two-call protocol: first call with indptr==NULL returns *nnz; second call fills
 * indptr[U+I+1], indices[nnz], values[nnz]. */
int idg_build_norm_adj(int64_t num_users, int64_t num_items, int64_t E, const int64_t* users,
                       const int64_t* items, int self_loops, const double* dinv, int64_t* nnz,
                       int64_t* indptr, int32_t* indices, float* values);

/* ------------------------------------------------------------------------------------
 * DEVICE: sparse graph handle
 * (replaces the coalesced torch sparse COO tensor built in models/LightGCN.py:30-32 via
 *  utility/utility_function/tools.py:95-109)
 * ---------------------------------------------------------------------------------- */
typedef struct idg_graph idg_graph;

#define IDG_GRAPH_SYMMETRIC 1u   /* A == A^T exactly: backward reuses the same CSR */
#define IDG_GRAPH_EXACT_ORDER 2u /* never split a row: every output element is the
                                    sequential CSR-order fmaf chain torch CPU computes */

/* CSR arrays are HOST pointers; the handle uploads and owns device copies plus its
 * row-block tile schedule.  split_threshold: rows with more stored entries than this are
 * cut into segments summed in a fixed published order (0 = library default 128; ignored with
 * EXACT_ORDER).  Rows of up to 512 entries are combined inside one workgroup (LDS); longer rows go
 * through one global partial per 512-entry chunk (workspace = n_long_chunks x d floats).  A handle
 * owns the arrival counters of its chunked rows: launches that use one handle must be ordered (one
 * stream, or events between streams); different handles are independent. */
int idg_graph_create(int device, int64_t n_rows, int64_t n_cols, int64_t nnz, const int64_t* indptr,
                     const int32_t* indices, const float* values, uint32_t flags,
                     int64_t split_threshold, idg_graph** out);
/* The same from DEVICE CSR arrays (SURVEY.md 8b's signature: the reference's graph is a coalesced sparse tensor that
 * already lives on the device, models/LightGCN.py:31-32).  The schedule is still built by host code: the arrays are copied
 * to host memory behind `stream` (the one call of this library that synchronises a stream: graph construction is a
 * one-off).  A device-side schedule build is future work. */
int idg_graph_create_from_device(int device, int64_t n_rows, int64_t n_cols, int64_t nnz, const int64_t* d_indptr,
                                 const int32_t* d_indices, const float* d_values, uint32_t flags,
                                 int64_t split_threshold, void* stream, idg_graph** out);
int idg_graph_destroy(idg_graph* g);

/* A copy of `g` whose stored entry (r, c) keeps values[k] / divisor when floor(u + add) != 0 and becomes an explicit
 * zero otherwise, u ~ U[0, 1) drawn by Philox4x32-10 from (seed, stream_id, r, c) — with transpose != 0 from (c, r):
 * for a symmetric `g` that copy is the TRANSPOSE of the first one, which is what the backward of a product with the
 * first one multiplies by.  This is NGCF.node_dropout (models/NGCF.py:56-65: keep where int(rand + (1 - keep_prob))
 * is non-zero, survivors divided by (1 - keep_prob)) with add = divisor = 1 - keep_prob; the zeros stay in the
 * structure (fmaf(0, x, acc) == acc), so the tile schedule, split schedule and summation order are those of `g`.
 * The copy shares g's schedule (g must outlive it) and owns its entry list; destroy it with idg_graph_destroy.
 * The mask is written on `stream`, in order with the products that follow. */
int idg_graph_masked_copy(const idg_graph* g, float add, float divisor, uint64_t seed, uint64_t stream_id,
                          int transpose, void* stream, idg_graph** out);
/* A new draw into an EXISTING copy of g (one made by idg_graph_masked_copy / idg_graph_revalued_copy): no allocation,
 * asynchronous on `stream` — the per-forward form of node dropout. */
int idg_graph_remask(const idg_graph* g, idg_graph* copy, float add, float divisor, uint64_t seed,
                     uint64_t stream_id, int transpose, void* stream);

/* Live work units of a row bitmap (optional accelerator of the row-restricted launches; round 2).  A launch given
 * `out_rows` normally visits every tile to find the few rows wanted.  idg_graph_live_units turns the bitmap into the list
 * of work units behind its rows — on `stream`: meant for the side stream that prepares a batch's index-only work — and
 * registers (schedule, bitmap pointer -> list); from then on every restricted launch on this handle, or on a masked /
 * revalued copy of it (copies share the schedule: no second registration), that names THIS bitmap pointer runs one wave
 * per listed unit and visits no tile.  Results are bit-identical to the tile form.
 * Validity.  A list describes the bitmap's contents at the time of the call.  Every library entry point that writes a row
 * bitmap (idg_bpr_touch_rows, idg_bitmap_clear, the `touched` argument of idg_bpr_backward_f32 / idg_bpr_fused_f32,
 * idg_graph_expand_rows / idg_graph_mark_cols outputs, idg_bpr_pack_rows_f32 / idg_bpr_unpack_rows_f32) drops the lists
 * registered for that pointer before it writes, and launches then fall back to the tile form (always correct) until
 * idg_graph_live_units is called again — so a buffer that was freed and handed out again is harmless once its new owner
 * fills it through the library.  A caller that rewrites a registered bitmap by other means calls
 * idg_graph_forget_live_units first (bitmap = NULL: every list of this handle's schedule); destroying the handle forgets
 * its lists.  The registry is process-wide, holds 64 lists (oldest replaced) and is thread-safe.
 * max_rows: upper bound on the set bits (3 x batch size).  More set bits than that cannot be listed: the list is marked
 * incomplete on the device, launches that use it write NaN into every row they produce, and idg_graph_live_units_check
 * (synchronises `stream`; for tests and debugging) returns IDG_E_INVALID.
 * units_ws: idg_graph_live_units_bytes(g, max_rows) bytes of device memory owned by the caller, alive while registered.
 * idg_graph_bind_live_units registers an existing list under another bitmap pointer / handle of the same schedule. */
size_t idg_graph_live_units_bytes(const idg_graph* g, int64_t max_rows);
int idg_graph_live_units(const idg_graph* g, const uint32_t* bitmap, void* units_ws, int64_t max_rows, void* stream);
int idg_graph_bind_live_units(const idg_graph* g, const uint32_t* bitmap, const void* units_ws, int64_t max_rows);
int idg_graph_forget_live_units(const idg_graph* g, const uint32_t* bitmap);
/* Drop every registered list (unit lists and compacted entry lists, of any handle) that lives in the buffer `ws`: what a
 * caller does before it frees or reuses a list buffer (the registration names the buffer; ADVICE r03: forgetting by bitmap
 * alone drops a NEW registration of the same bitmap when an OLD buffer dies). */
int idg_graph_forget_units_ws(const void* ws);
int idg_graph_live_units_check(const void* units_ws, void* stream);
/* The first backward product of a training step gathers from a panel with <= 3B live rows (d loss / d final: the batch's
 * rows).  Which stored entries point at live rows is index-only work: idg_graph_compact_inputs does it ahead of time (on
 * `stream`, typically the caller's side stream) — per tile the live entries, in their order, a compacted start per virtual
 * row and a live count — and registers the result for `bitmap` exactly as idg_graph_live_units registers a unit list (same
 * validity rules: every library call that writes the bitmap drops it; idg_graph_forget_live_units drops it).  A product
 * naming that bitmap as x_rows (idg_spmm_epi_f32, idg_propagate_mean_bwd*_f32's gout_mask) then walks the compacted
 * lists with the ordinary kernel: bit-identical to the in-kernel compaction and to the dense product.
 * ws: idg_graph_compact_inputs_bytes(g) bytes, 16-byte aligned, owned by the caller while registered. */
size_t idg_graph_compact_inputs_bytes(const idg_graph* g);
int idg_graph_compact_inputs(const idg_graph* g, const uint32_t* bitmap, void* ws, void* stream);

/* The batch's receptive field (no reference counterpart: the reference propagates the whole graph every step,
 * models/LightGCN.py:36-52, although the loss reads the final layer at the batch's rows only).  Layer K is needed at the
 * batch's rows S, layer K - 1 at S and their neighbours, and so on; the gradient flows back through the same sets.  On a
 * graph much larger than the batch's K-hop neighbourhood most of every product is work nobody reads.
 * idg_graph_expand_rows: out_rows = in_rows | {columns of the stored entries of the rows flagged in in_rows} (bitmaps of
 *   n_rows bits, distinct buffers) — one hop.
 * idg_propagate_mean_fields_f32: idg_propagate_mean_f32 with layer k producing the rows of layer_rows[k-1] only (host
 *   array of K device bitmaps; layer_rows[K-1] = the batch's rows; each set must contain the neighbours of the next).
 *   Rows outside the sets keep whatever the buffers held.  Same bits on the produced rows as the full propagation.
 * idg_propagate_mean_bwd_adam_fields_f32: idg_propagate_mean_bwd_adam_f32 where step k's input is known to be zero
 *   outside step_rows[k-1] (step_rows[0] = live rows of gout; NULL = dense; the last step is always dense: it carries
 *   the Adam update of every row).  Same bits as the unrestricted call. */
int idg_graph_expand_rows(const idg_graph* g, const uint32_t* in_rows, uint32_t* out_rows, void* stream);
/* The same hop on a graph of any shape, as float flags over the COLUMNS (caller-zeroed, n_cols floats): col_flags[c] =
 * 1.0f for every column of a stored entry of a row flagged in in_rows (n_rows bits).  The ranks of the sharded step sum
 * these to agree on the item rows a batch's users touch. */
int idg_graph_flag_cols(const idg_graph* g, const uint32_t* in_rows, float* col_flags, void* stream);
/* ... and as a bitmap over the columns (n_cols bits, NOT cleared here: col_bits |= the hop, so the row slices of one
 * matrix can mark into one bitmap).  The sharded step marks the users that reach the touched items this way. */
int idg_graph_mark_cols(const idg_graph* g, const uint32_t* in_rows, uint32_t* col_bits, void* stream);
int idg_propagate_mean_fields_f32(const idg_graph* g, const float* E0, float* out, const uint32_t* const* layer_rows, int K,
                                  int include_layer0, int64_t d, void* ws, void* stream);
int idg_propagate_mean_bwd_adam_fields_f32(const idg_graph* g, const float* gout, const uint32_t* const* step_rows, float* gE0,
                                           int K, int include_layer0, int64_t d, int accumulate, float* param,
                                           float* exp_avg, float* exp_avg_sq, double lr, double beta1, double beta2,
                                           double eps, int64_t step, void* ws, void* stream);

/* A copy of `g` with NEW VALUES on the same structure and schedule, taken from a DEVICE CSR (indptr int64 [n_rows+1],
 * indices int32 ascending per row, values fp32) that holds every entry of g (it may hold more): entry (r, c) takes the
 * value stored for (r, c) there.  This is how a per-epoch sub-graph gets onto the device without a host-side tile
 * schedule: SGL's edge-dropped views (tools.create_adj_mat, tools.py:67-92; models/SGL.py:130-143) are the full
 * adjacency's handle with dropped interactions as explicit zeros and the kept ones re-normalised.  Same ownership
 * rules as idg_graph_masked_copy; the flags (symmetry) are g's — the caller passes symmetric values. */
int idg_graph_revalued_copy(const idg_graph* g, const int64_t* d_indptr, const int32_t* d_indices,
                            const float* d_values, void* stream, idg_graph** out);

/* values_out[k] = kept(edge_of_entry[k]) ? (dinv[row_of_entry[k]] * 1.0f) * dinv[col_of_entry[k]] : 0 for the nnz
 * entries of a bipartite adjacency in CSR order (all arrays on the device): the float32 arithmetic of
 * degree_matrix.dot(adjacency_matrix).dot(degree_matrix) in tools.create_adj_mat (tools.py:84-90) on the interactions
 * whose bit is set in kept_bits (bit e = interaction e of inter_graph.nonzero()'s order); dinv = d^-1/2 of the kept
 * graph, formed by the caller with the reference's own np.power expression. */
int idg_subgraph_values_f32(int64_t nnz, const int32_t* row_of_entry, const int32_t* col_of_entry,
                            const int32_t* edge_of_entry, const uint32_t* kept_bits, const float* dinv,
                            float* values_out, void* stream);
/* info[0..7] = n_rows, n_cols, nnz, n_tiles, n_long_rows, n_long_chunks, split_threshold, flags */
int idg_graph_info(const idg_graph* g, int64_t info[8]);
/* The split schedule, so a checker can restate the exact summation order.  Define
 *   SEG(entries, S): cut the entry range into consecutive segments of S stored entries, each a
 *   sequential fmaf chain from +0; combine the segment partials p_0, p_1, ... 4-way strided:
 *   s_q = p_q + p_{q+4} + p_{q+8} + ... (left to right) for q = 0..3, result = ((s_0 + s_1) + s_2) + s_3
 *   (absent s_q skipped).
 * Row long_rows[i] with chunk_len[i] == 0 is SEG(row, seg_len[i]).  With chunk_len[i] = C > 0 the row is
 * first cut into consecutive chunks of C entries, chunk sums c_k = SEG(chunk k, seg_len[i]), and the c_k are
 * combined by the same 4-way strided rule.  Every other row is one sequential chain.
 * The arrays hold n_long_rows (info[4]) elements; any may be NULL. */
int idg_graph_long_rows(const idg_graph* g, int64_t* long_rows, int64_t* seg_len, int64_t* chunk_len);

/* Y = A.X  (torch.sparse.mm(Graph, X): models/LightGCN.py:44, SimGCL.py:48, XSimGCL.py:51,
 * NGCF.py:85, SGL.py:48,50).  X [n_cols, d], Y [n_rows, d] row-major fp32 with leading
 * dimensions ldx/ldy (elements).  If addend != NULL: Y = A.X + addend (addend [n_rows,d],
 * leading dimension ldy) — the autograd chain rule of the layer mean uses this form.
 * ws: idg_spmm_workspace_bytes(g, d) bytes of device scratch (may be NULL when that is 0). */
size_t idg_spmm_workspace_bytes(const idg_graph* g, int64_t d);
int idg_spmm_f32(const idg_graph* g, const float* X, int64_t ldx, float* Y, int64_t ldy,
                 const float* addend, int64_t d, void* ws, void* stream);

/* The same product with the whole fused epilogue exposed, for callers that assemble the layer
 * loop themselves (the user-row-sharded multi-GPU path, where an all-reduce sits between the
 * layers):  t = A.X (+ addend);  Y = t (if Y);  s = (sum_in ? sum_in + t : t) / div;
 * sum_out (+)= s (if sum_out; += when accumulate).  All panels share ldy.  out_rows (nullable):
 * bitmap of the output rows to produce; the others are left untouched.  x_rows (nullable): bitmap of the live rows of
 * X; the others are zero by agreement and are not read. */
int idg_spmm_ex_f32(const idg_graph* g, const float* X, int64_t ldx, float* Y, const float* addend,
                    const float* sum_in, float* sum_out, int64_t ldy, float div, int accumulate,
                    const uint32_t* out_rows, const uint32_t* x_rows, int64_t d, void* ws,
                    void* stream);

/* The same product with EVERY epilogue option as a struct (idg_spmm_ex_f32 is the subset without sum_in2 / sum_in3, mask
 * and Adam).  In order, for each produced row r:
 *   t = (A.X)[r]  (+ addend[r] if live(r));  Y[r] = t (if Y);
 *   s = live(r) && sum_in ? ((sum_in[r] + sum_in2[r]) + sum_in3[r]) + t : t   (absent terms skipped; this is the layer
 *       mean's left-to-right order, models/LightGCN.py:47-48);  s = s / div;
 *   sum_out[r] = accumulate && live(r) ? sum_out[r] + s : s (if sum_out);
 *   adam_param != NULL: the Adam update of row r of (param, exp_avg, exp_avg_sq) with gradient sum_out[r]
 *       (trainer.py:54-56; same arithmetic as idg_adam_step_f32; dense launches only).
 * live(r) = mask == NULL or bit r of mask: rows of addend / sum_in* / the accumulate target whose bit is clear are zero
 * by agreement and are NOT read (they may hold anything).  out_rows: produce only these rows; x_rows: rows of X outside
 * are zero and not read; the two may be combined (a backward product between two small row sets). */
typedef struct idg_epilogue {
  float* Y;
  const float* addend;
  const float* sum_in;
  const float* sum_in2;
  const float* sum_in3;
  float* sum_out;
  int64_t ldy;
  float div;
  int accumulate;
  const uint32_t* mask;
  float* adam_param;
  float* adam_exp_avg;
  float* adam_exp_avg_sq;
  double adam_lr, adam_beta1, adam_beta2, adam_eps;
  int64_t adam_step;
  int adam_discard_grad; /* != 0: the finished gradient feeds the update and is not written to sum_out (which is still
                            READ at live rows when accumulate is set): one [n, d] write less per step */
  /* EGCF's activated layers (models/EGCF.py:46-84: activation_layer(torch.sparse.mm(...)), nn.Tanh) and their backward:
   * applied to t = (A.X)[r] (+ addend[r]) of the rows r < act_rows (0: every row) BEFORE Y / sum_out see it.
   * IDG_ACT_TANH: t = tanh(t).  IDG_ACT_TANH_BWD: t = t * (1 - act_src[r]^2), act_src = the tanh outputs the forward
   * layer saved (torch's tanh_backward).  Not together with Adam; with out_rows or with x_rows, not both. */
  int act;
  const float* act_src;
  int64_t act_rows;
  /* The finished row t (after addend / activation) ALSO — or only: Y may then be NULL — as 24-bit values in idg_pack24_f32's
   * format, three words per four values at y24 + (r * ldy + f) / 4 * 3: the sharded step's packed exchange lets the product
   * write its partial straight into the send buffer (one fp32 write and one pack pass less per exchanged panel).  Dense
   * launches (x_rows allowed, out_rows not), tiled widths, not with Adam. */
  uint32_t* y24;
} idg_epilogue;
#define IDG_ACT_TANH 1
#define IDG_ACT_TANH_BWD 2
int idg_spmm_epi_f32(const idg_graph* g, const float* X, int64_t ldx, int64_t d, const idg_epilogue* epilogue,
                     const uint32_t* out_rows, const uint32_t* x_rows, void* ws, void* stream);
/* out[r, :] = grad[r, :] * (1 - y[r, :]^2) for the rows flagged in `rows` (NULL: every row) of [n, d] panels: the backward
 * of y = tanh(z) where the chain starts — EGCF's summed layer outputs at the batch's rows (models/EGCF.py:60,77).  Other
 * rows of out are left untouched. */
int idg_rows_tanh_bwd_f32(const float* grad, const float* y, const uint32_t* rows, int64_t n, int64_t d, float* out,
                          void* stream);

/* One perturbed layer (models/XSimGCL.py:51-54): Y = A.X;  Y += sign(Y) * normalize(u, dim=-1) * eps,
 * u ~ U[0,1)^d from Philox4x32-10(seed; stream_id, row, feature block).  d in {32,...,512}.
 * out_rows (nullable): bitmap of the rows to produce.  Layer k of idg_propagate_mean_noise_f32(seed, s)
 * draws from stream s * 64 + k: the same layer can be re-produced on its own. */
int idg_spmm_noise_f32(const idg_graph* g, const float* X, int64_t ldx, float* Y, int64_t ldy,
                       const uint32_t* out_rows, int64_t d, float eps, uint64_t seed,
                       uint64_t stream_id, void* ws, void* stream);
/* The perturbation alone, Y[r] = X[r] + sign(X[r]) * normalize(u_r) * eps with the same generator and arithmetic
 * as the epilogue form, for all n rows or the rows of a bitmap (nullable).  Lets passes that share a product
 * (SimGCL: A.E0 feeds the clean pass and both perturbed views) perturb it without multiplying again. */
int idg_perturb_f32(const float* X, float* Y, int64_t n, int64_t d, const uint32_t* rows, float eps,
                    uint64_t seed, uint64_t stream_id, void* stream);

/* SimGCL's encoder passes of one step (models/SimGCL.py:63-65) — the clean layer mean and n_views (1..2)
 * perturbed ones, layer 0 not in the mean — as one call: the first product A.E0 is computed once, each view
 * perturbs its own copy (sub-stream 0 of (seeds[i], stream_ids[i])) and continues with sub-streams 1..K-1 exactly as
 * idg_propagate_mean_noise_f32(X1, K - 1, include_layer0 = 1) would; with out_rows the last layer of ALL passes is ONE
 * launch of a multi-panel row-restricted kernel.  Same values as composing idg_spmm_f32, idg_perturb_f32,
 * idg_propagate_mean_f32 and idg_propagate_mean_noise_f32 by hand.  K >= 2, tiled d.
 * out_views: HOST array of n_views device pointers.  ws: idg_propagate_views_workspace_bytes. */
size_t idg_propagate_views_workspace_bytes(const idg_graph* g, int64_t d, int n_views);
int idg_propagate_views_f32(const idg_graph* g, const float* E0, int K, int64_t d, float eps, int n_views,
                            const uint64_t* seeds, const uint64_t* stream_ids, float* out_clean,
                            float* const* out_views, const uint32_t* out_rows, void* ws, void* stream);

/* LightGCN.aggregate (models/LightGCN.py:36-52) / SimGCL.aggregate(perturbed=False)
 * (models/SimGCL.py:39-60): out = mean over layers of E_k, E_{k+1} = A.E_k, k < K,
 * E_0 included iff include_layer0.  Running sum left-to-right then a true division by the
 * layer count — the order torch.mean(torch.stack(..)) produces on CPU.
 * E0, out: [n, d] contiguous.  ws: idg_propagate_workspace_bytes(g, d). */
size_t idg_propagate_workspace_bytes(const idg_graph* g, int64_t d);
/* out_rows (nullable): bitmap of the rows of `out` the caller will read.  The LAST layer's product
 * feeds nothing but the mean, so with a bitmap it is evaluated for the flagged rows only (bit-
 * identical values there; every other row of `out` is left untouched).  A training step reads the
 * mean at the <= 3B rows of its batch: idg_bpr_touch_rows builds that bitmap from the indices. */
int idg_propagate_mean_f32(const idg_graph* g, const float* E0, float* out,
                           const uint32_t* out_rows, int K, int include_layer0, int64_t d, void* ws,
                           void* stream);
/* SimGCL.aggregate(perturbed=True) (models/SimGCL.py:47-56): as idg_propagate_mean_f32, but after
 * every product X <- A.X the layer is perturbed in place, X += sign(X) * normalize(u, dim=-1) * eps
 * with u ~ U[0,1)^d, before it enters the running sum and feeds the next layer.  u comes from
 * Philox4x32-10 keyed by (seed, stream_id, layer, row, feature block): reproducible for a given
 * (seed, stream_id), independent of the tile schedule; like the reference's device generator it
 * matches a CPU run statistically, not bit for bit.  d in {32, 64, 128, 256, 512}.  out_rows as in
 * idg_propagate_mean_f32 (a row's noise depends on that row only: the produced rows are the same).
 * The gradient w.r.t. E0 is idg_propagate_mean_bwd_f32's (sign() has zero gradient, u is constant). */
int idg_propagate_mean_noise_f32(const idg_graph* g, const float* E0, float* out,
                                 const uint32_t* out_rows, int K, int include_layer0, int64_t d,
                                 float eps, uint64_t seed, uint64_t stream_id, void* ws, void* stream);

/* Backward of the above for a SYMMETRIC graph: gE0 = (1/cnt)(c0.g + A(g + A(g + ... A g))),
 * the Horner form of autograd's chain through K torch.sparse.mm nodes and the mean.
 * accumulate != 0: gE0 += (the ego-embedding regulariser's gradient is already there). */
/* gout_mask (nullable): bitmap, bit r of word r/32 set iff row r of gout is live.  Rows with a
 * clear bit are taken as zero and never read — a training batch touches <= 3B rows of gout, so
 * the first backward product skips most of its gathers (exact: fmaf(v, 0, acc) == acc) and gout
 * needs no zero-fill.  With accumulate != 0 the same bitmap governs gE0: flagged rows are added
 * to, every other row is overwritten (gE0 needs no zero-fill either).  Produced by
 * idg_bpr_backward_f32's `touched` argument. */
int idg_propagate_mean_bwd_f32(const idg_graph* g, const float* gout, const uint32_t* gout_mask,
                               float* gE0, int K, int include_layer0, int64_t d, int accumulate,
                               void* ws, void* stream);
/* The same backward followed by the dense Adam step on the [n, d] parameter panel E0 the gradient
 * belongs to (loss.backward() + optimizer.step(), utility/utility_train/trainer.py:54-56): each
 * finished gradient row is consumed by its Adam update in the epilogue of the last product instead
 * of being re-read by a second kernel.  gE0 still receives the gradient.  Bit-identical to
 * idg_propagate_mean_bwd_f32 + idg_adam_step_f32 (which is what runs when K < 2 or d is not a tiled
 * width).  Hyper-parameters as idg_adam_step_f32. */
#define IDG_ADAM_DISCARD_GRAD 2 /* OR-ed into `accumulate` of the two calls below: the finished gradient feeds the update and is
                                * NOT written back to gE0 (17.8 MB per step less to store on the yelp2018 shape) */
int idg_propagate_mean_bwd_adam_f32(const idg_graph* g, const float* gout, const uint32_t* gout_mask,
                                    float* gE0, int K, int include_layer0, int64_t d, int accumulate,
                                    float* param, float* exp_avg, float* exp_avg_sq, double lr,
                                    double beta1, double beta2, double eps, int64_t step, void* ws,
                                    void* stream);

/* ------------------------------------------------------------------------------------
 * DEVICE: fused gather + BPR + L2-reg loss and gradients
 * (LightGCN.forward models/LightGCN.py:54-72; losses.get_bpr_loss / get_reg_loss
 *  utility/utility_function/losses.py:4-21; and their autograd backward)
 * ---------------------------------------------------------------------------------- */
/* final/ego: [n, d] panels, rows [0,num_users) users then items.  users/pos/neg: int64[B]
 * (pos/neg are item ids, NOT offset by num_users).
 * loss[0] = mean_i -log(sigmoid(<f_u,f_p> - <f_u,f_n>) + 1e-7)
 * loss[1] = reg_lambda * sum over the three ego blocks of 0.5*||block||_F^2 / B
 * g_final[n,d] += d loss[0] / d final   (rows scatter-added, duplicates accumulate)
 * g_ego  [n,d] += d loss[1] / d ego
 * touched (nullable, deterministic forms only, g_final != g_ego): a zeroed bitmap of ceil(n/32)
 * words; the rows of g_final AND g_ego this batch reaches (the same rows) are then STORED (not
 * accumulated) and their bits set, every other row of both panels is left untouched and must
 * not be read — hand the bitmap to idg_propagate_mean_bwd_f32, which reads flagged rows only.
 * g_final / g_ego must be zeroed (or hold a gradient to accumulate into) by the caller;
 * either may be NULL to skip that gradient.  final == ego is the MFBPR case
 * (models/MFBPR.py:29-42).  deterministic != 0: duplicate rows are summed in batch order
 * (run-to-run reproducible); 0: float atomics.
 * ws: idg_bpr_workspace_bytes(B, d). */
size_t idg_bpr_workspace_bytes(int64_t B, int64_t d);
int idg_bpr_fused_f32(const float* final_panel, const float* ego_panel, int64_t num_users,
                      int64_t n, const int64_t* users, const int64_t* pos, const int64_t* neg,
                      int64_t B, int64_t d, float reg_lambda, float* loss, float* g_final,
                      float* g_ego, int deterministic, uint32_t* touched, void* ws, void* stream);
/* The same with final rows and ego rows of DIFFERENT widths and an optional item-only regulariser — NGCF (models/NGCF.py:
 * 108-128): the scored rows are the concatenation of the layer outputs ([n, (K+1) d]), the regulariser covers
 * item_embedding(positive) and item_embedding(negative) only (two blocks; reg_users = 0: the user block contributes
 * nothing to loss[1] and a zero row to g_ego).  g_final [n, d_final], g_ego [n, d_ego]; deterministic scatter only
 * (1, or IDG_BPR_PLANNED with the plan in ws). */
int idg_bpr_fused_ex_f32(const float* final_panel, int64_t d_final, const float* ego_panel, int64_t d_ego,
                         int64_t num_users, int64_t n, const int64_t* users, const int64_t* pos, const int64_t* neg,
                         int64_t B, float reg_lambda, int reg_users, float* loss, float* g_final, float* g_ego,
                         int deterministic, uint32_t* touched, void* ws, void* stream);
/* The same computation as two calls, for callers that sit under an autograd engine:
 * forward writes loss[2] and keeps per-triple coefficients in ws; backward (same ws, same
 * inputs) scatters the gradients scaled by upstream[0] (for loss[0]) and upstream[1] (for
 * loss[1]) — a DEVICE pointer to the two incoming gradient scalars, or NULL for 1, 1. */
/* deterministic = IDG_BPR_PLANNED: the sorted (row, slot) plan of THIS batch is already in ws,
 * put there by idg_bpr_plan_f32 (which depends on the indices only, so a caller can run it on
 * a second stream while the forward propagation is still in flight). */
/* bitmap[(r >> 5)] |= 1 << (r & 31) for r in {users[i], num_users + pos[i], num_users + neg[i]}: the
 * panel rows a batch touches.  bitmap: ceil(n/32) words, zeroed by the caller.  Index-only work. */
int idg_bpr_touch_rows(const int64_t* users, const int64_t* pos, const int64_t* neg, int64_t B,
                       int64_t num_users, uint32_t* bitmap, void* stream);
/* Clear a row bitmap of n_bits bits ((n_bits + 31) / 32 words) on `stream`. */
int idg_bitmap_clear(uint32_t* bitmap, int64_t n_bits, void* stream);

/* Gradient rows as a message, for data-parallel replicas (the reference trains on one device,
 * utility/utility_train/trainer.py:42-56 — the replicas reproduce its loss at batch size world x B).
 * After idg_bpr_fused_f32 / idg_bpr_backward_f32 with a touched-row bitmap and the sorted plan in ws, g_final holds
 * one stored row per distinct row of the batch.  idg_bpr_pack_rows_f32 copies them, the sorted row keys of the plan
 * and the two losses into `message` (idg_bpr_rows_message_floats(B, d) floats).  Replicas all-gather their messages;
 * idg_bpr_unpack_rows_f32 then clears `touched` (n bits; unless told it is clear) and merges `world` (<= 64) consecutive messages in ONE launch:
 * the lowest rank naming a row owns it and adds the ranks' rows IN RANK ORDER — g_final rows averaged over ranks,
 * g_ego = the regulariser's gradient (reg_lambda / (world B)) x multiplicity x ego row, loss[2] = mean of the ranks'
 * losses — so that every replica obtains the same bits; at world 1 the panels equal what the scatter left.  The
 * backward propagation then runs on the union bitmap `touched`. */
size_t idg_bpr_rows_message_floats(int64_t B, int64_t d);
/* clear_bitmap (nullable) / clear_bits: a bitmap the pack launch zeroes on the way — hand it the `touched` of the
 * coming merge and pass touched_is_clear = 1 there, and no memset sits between the all-gather and the merge. */
int idg_bpr_pack_rows_f32(const void* ws, int64_t B, int64_t d, const float* g_final, const float* loss,
                          float* message, uint32_t* clear_bitmap, int64_t clear_bits, void* stream);
int idg_bpr_unpack_rows_f32(const float* messages, int world, int64_t B, int64_t d, int64_t n,
                            const float* ego_panel, float reg_lambda, float* g_final, float* g_ego,
                            uint32_t* touched, int touched_is_clear, float* loss, void* stream);
#define IDG_BPR_PLANNED 2
/* OR-ed into `deterministic` together with a `touched` bitmap: the bitmap ALREADY holds exactly the batch's rows
 * (idg_bpr_touch_rows on the same ids) — the scatter stores its rows as with any touched bitmap but does not write the
 * bitmap, so unit / compacted-input lists registered for it stay valid for the backward products that follow. */
#define IDG_BPR_TOUCHED_PRESET 4
int idg_bpr_plan_f32(const int64_t* users, const int64_t* pos, const int64_t* neg, int64_t B,
                     int64_t num_users, int64_t n, void* ws, void* stream);
/* idg_bpr_plan_f32 + idg_bpr_touch_rows in one pass over the ids: the plan's first kernel also sets the batch's bits in
 * `bitmap` (ceil(n / 32) words, zeroed by the caller) — one launch less per step. */
int idg_bpr_plan_rows_f32(const int64_t* users, const int64_t* pos, const int64_t* neg, int64_t B, int64_t num_users,
                          int64_t n, void* ws, uint32_t* bitmap, void* stream);
int idg_bpr_forward_f32(const float* final_panel, const float* ego_panel, int64_t num_users,
                        int64_t n, const int64_t* users, const int64_t* pos, const int64_t* neg,
                        int64_t B, int64_t d, float reg_lambda, float* loss, void* ws,
                        void* stream);
int idg_bpr_backward_f32(const float* final_panel, const float* ego_panel, int64_t num_users,
                         int64_t n, const int64_t* users, const int64_t* pos, const int64_t* neg,
                         int64_t B, int64_t d, float reg_lambda, const float* upstream,
                         float* g_final, float* g_ego, int deterministic, uint32_t* touched, void* ws,
                         void* stream);

/* ------------------------------------------------------------------------------------
 * DEVICE: weight gradient of a thin dense layer  (autograd of torch.matmul(side, W) in NGCF's
 * per-layer transforms, models/NGCF.py:91-99):  w_grad[d1, d2] (+)= X^T . G,  X [n, d1] (leading
 * dimension ldx), G [n, d2] (ldg), row-major fp32.  All reduction (K = n), no output to speak of:
 * rows are cut into slices summed in slice order (deterministic).  ws: *_workspace_bytes.
 * ---------------------------------------------------------------------------------- */
size_t idg_linear_wgrad_workspace_bytes(int64_t n, int64_t d1, int64_t d2);
int idg_linear_wgrad_f32(const float* X, int64_t ldx, const float* G, int64_t ldg, int64_t n,
                         int64_t d1, int64_t d2, float* w_grad, int accumulate, void* ws,
                         void* stream);

/* The four parameter gradients of one NGCF layer in one pass over the rows: out = [g W_gcn (d1 x d2) | g b_gcn (d2) |
 * g W_bi (d1 x d2) | g b_bi (d2)] with g W_gcn = side^T gT, g W_bi = bi^T gT (bi = ego * side as idg_ngcf_transform_f32
 * left it), g b_* = column sums of gT (models/NGCF.py:91-99 under autograd).  d1, d2 multiples of 64; deterministic. */
size_t idg_ngcf_wgrad_workspace_bytes(int64_t d1, int64_t d2);
int idg_ngcf_wgrad_f32(const float* side, const float* bi, const float* gT, int64_t n, int64_t d1, int64_t d2,
                       float* out, void* ws, void* stream);

/* NGCF's two per-layer transforms (models/NGCF.py:88-99: torch.matmul(side, W_gcn) and torch.matmul(ego * side, W_bi))
 * on the fp32 matrix cores in one pass over the rows: S[n, d2] = side . W1 + (ego * side) . W2 (the two bias rows are
 * added by idg_ngcf_tail_f32, which takes S as its S1 with S2 = NULL); BI (nullable) receives ego * side [n, d1], the
 * left operand of W2's weight gradient.  Backward: g_side = gS . W1^T + (gS . W2^T) * ego, g_ego = (gS . W2^T) * side
 * (weight gradients: idg_linear_wgrad_f32 on (side, gS) and (BI, gS)).  W1, W2: [d1, d2] row-major.
 * Forward needs d1 % 64 == 0 and d2 % 32 == 0, backward d2 % 64 == 0 and d1 % 32 == 0, panels 16-byte aligned. */
int idg_ngcf_transform_f32(const float* side, const float* ego, const float* W1, const float* W2, int64_t n,
                           int64_t d1, int64_t d2, float* S, float* BI, void* stream);
int idg_ngcf_transform_bwd_f32(const float* gS, const float* side, const float* ego, const float* W1,
                               const float* W2, int64_t n, int64_t d1, int64_t d2, float* g_side, float* g_ego,
                               void* stream);

/* NGCF's per-layer tail after the two thin GEMMs (models/NGCF.py:95-108), one pass over the rows:
 *   t = (S1 + b1) + (S2 + b2);  a = leaky_relu(t, negative_slope);  E = dropout(a, p);  N = normalize(E, dim=1)
 * (S1 = side.W_gcn, S2 = (ego * side).W_bi, all [n, d] row-major; b1, b2 [d]; S2 = NULL: S1 already holds the sum of
 * the two, as idg_ngcf_transform_f32 produces it).  The dropout mask is a counter-based
 * function of (seed, stream_id, row, feature) — always applied, as in the reference, where nn.Dropout is built
 * inside aggregate() — and regenerated by the backward call, which needs only E:
 *   gT = d loss / d t  given  gE (gradient through E as the next layer's input, nullable) and gN (through N, nullable);
 * then gS1 = gS2 = gT, g b1 = g b2 = column sums of gT. */
int idg_ngcf_tail_f32(const float* S1, const float* S2, const float* b1, const float* b2, int64_t n,
                      int64_t d, float negative_slope, float p, uint64_t seed, uint64_t stream_id,
                      float* E, float* N, void* stream);
int idg_ngcf_tail_bwd_f32(const float* E, const float* gE, const float* gN, int64_t n, int64_t d,
                          float negative_slope, float p, uint64_t seed, uint64_t stream_id, float* gT,
                          void* stream);
/* The forms a fused, autograd-free NGCF step uses (id-grec_amd/ngcf.py): N written with leading dimension ldn — straight
 * into layer l's slot of the concatenated final rows (torch.cat(all_embeddings, dim=1), models/NGCF.py:108); gN read with
 * leading dimension ldgn and only at the rows flagged in gn_rows (NULL: every row) — d loss / d final is stored at the
 * batch's rows only; rows no gradient flows into get gT = 0. */
int idg_ngcf_tail_ex_f32(const float* S1, const float* S2, const float* b1, const float* b2, int64_t n, int64_t d,
                         float negative_slope, float p, uint64_t seed, uint64_t stream_id, float* E, float* N,
                         int64_t ldn, void* stream);
int idg_ngcf_tail_bwd_ex_f32(const float* E, const float* gE, const float* gN, int64_t ldgn, const uint32_t* gn_rows,
                             int64_t n, int64_t d, float negative_slope, float p, uint64_t seed, uint64_t stream_id,
                             float* gT, void* stream);
/* One NGCF layer as ONE kernel per direction, d = 64 (csrc/idg_ngcf.hip; models/NGCF.py:88-108 and its autograd):
 *   forward : E = dropout(leaky_relu(side . W1 + (ego * side) . W2 + b1 + b2)), N = normalize(E)  — N with leading dimension
 *             ldn (layer l's slot of the concatenated final rows); nothing else is stored;
 *   backward: from E, gE (nullable), gN (nullable; ldgn, gn_rows as idg_ngcf_tail_bwd_ex_f32), side and the layer's input
 *             ego: g_side, g_ego [n, 64] and w_grads = [g W1 (64 x 64) | g b1 (64) | g W2 (64 x 64) | g b2 (64)].
 * E, N, g_side, g_ego are bit-identical to idg_ngcf_transform_f32 + idg_ngcf_tail_ex_f32 and their backward forms; the
 * parameter gradients are sums over the rows in a fixed order of their own (slices of rows on persistent workgroups).
 * Any other d: IDG_E_INVALID (use the chain).  ws: idg_ngcf_layer_bwd_workspace_bytes(d). */
int idg_ngcf_layer_fwd_f32(const float* side, const float* ego, const float* W1, const float* W2, const float* b1,
                           const float* b2, int64_t n, int64_t d, float negative_slope, float p, uint64_t seed,
                           uint64_t stream_id, float* E, float* N, int64_t ldn, void* stream);
size_t idg_ngcf_layer_bwd_workspace_bytes(int64_t d);
int idg_ngcf_layer_bwd_f32(const float* E, const float* gE, const float* gN, int64_t ldgn, const uint32_t* gn_rows,
                           const float* side, const float* ego, const float* W1, const float* W2, int64_t n, int64_t d,
                           float negative_slope, float p, uint64_t seed, uint64_t stream_id, float* g_side, float* g_ego,
                           float* w_grads, void* ws, void* stream);
/* Glue of that step.  idg_copy_cols_f32: dst[r, 0:d] = src[r, 0:d] with leading dimensions ldd / lds.  idg_rows_add2_f32:
 * dst[r] += a[r] (+ b[r]) at the rows flagged in `rows`.  idg_colsum_f32: out[f] (+)= sum over the n rows of X[r, f] — the
 * gradient of a bias row broadcast over n rows (the differentiable layer tail's backward; the fused step gets it from
 * idg_ngcf_wgrad_f32) — slices of rows summed in slice order; ws: idg_colsum_workspace_bytes(d). */
size_t idg_colsum_workspace_bytes(int64_t d);
int idg_colsum_f32(const float* X, int64_t ldx, int64_t n, int64_t d, float* out, int accumulate, void* ws, void* stream);
int idg_copy_cols_f32(float* dst, int64_t ldd, const float* src, int64_t lds, int64_t n, int64_t d, void* stream);
int idg_rows_add2_f32(float* dst, int64_t ldd, const float* a, int64_t lda, const float* b, int64_t ldb,
                      const uint32_t* rows, int64_t n, int64_t d, void* stream);

/* ------------------------------------------------------------------------------------
 * DEVICE: in-batch InfoNCE between two views, forward + backward
 * (utility/utility_function/losses.py:24-35 get_InfoNCE_loss; call sites models/SimGCL.py:79-84,
 *  XSimGCL.py:80-86, SGL.py:96-101: once over unique(batch users), once over unique(batch positive
 *  items), rows gathered from the two [n, d] view panels, users first):
 *     a = normalize(view1[idx]), b = normalize(view2[idx])      (x / max(||x||, 1e-12))
 *     loss = mean_i -log( exp(<a_i,b_i>/t) / sum_k exp(<a_i,b_k>/t) + 1e-5 )
 * loss[0] = the user-set loss, loss[1] = the item-set loss (rows num_users + item id).
 * g1 / g2 (nullable): d(loss[0] + loss[1]) / d view1 / d view2 — the rows of the two sets are
 * STORED, every other row is left untouched (zero-fill the panels first if you need dense
 * gradients), multiplied by grad_scale; with accumulate != 0 they are ADDED to what the rows hold.
 * g1 == g2 is allowed (both views' gradients meet in one panel — a fused training step sums them
 * with the BPR gradient before ONE backward propagation).  The unique id lists never visit the
 * host: launches are shaped by B.
 * dedup = 0 keeps the id lists as they are, duplicates included (models/SGL.py:85-86 indexes its views with
 * the raw batch ids): every occurrence is a row of the in-batch matrix and the gradients of the occurrences
 * of one id are added, in list order, into its panel row.
 * ---------------------------------------------------------------------------------- */
size_t idg_infonce_workspace_bytes(int64_t n, int64_t B, int64_t d);
/* The CROSS form (models/EGCF.py:103: get_InfoNCE_loss(user_embedding, pos_embedding, t) on the RAW batch rows):
 * a_i = normalize(view[users[i]]), b_i = normalize(view[num_users + items[i]]), i < B in batch order, same loss as above
 * in loss[0]; g (nullable): d loss / d view, ADDED (times grad_scale) into the rows of the batch's users and items —
 * the occurrences of one id added in list order.  Same workspace. */
int idg_infonce_cross_f32(const float* view, int64_t n, int64_t d, const int64_t* users, const int64_t* items,
                          int64_t B, int64_t num_users, float temperature, float* loss, float* g, float grad_scale,
                          void* ws, void* stream);
int idg_infonce_pair_f32(const float* view1, const float* view2, int64_t n, int64_t d,
                         const int64_t* users, const int64_t* items, int64_t B, int64_t num_users,
                         int dedup, float temperature, float* loss, float* g1, float* g2,
                         float grad_scale, int accumulate, void* ws, void* stream);
/* The id-list stage of a call on its own — index-only work (unique / raw / cross lists of the batch, the repeat flags and
 * positions of raw lists) that a training loop runs on a side stream one batch ahead, as idg_bpr_plan_f32 does for the
 * scatter plan: idg_infonce_plan(..., mode, ws, stream) with mode 0 = unique ids (dedup = 1), 1 = raw lists (dedup = 0),
 * 2 = the cross form; then the matching call on the SAME workspace with IDG_SSL_PLANNED OR-ed into `dedup`
 * (idg_infonce_pair_f32) or planned != 0 (idg_infonce_cross_ex_f32) skips the stage.  The caller orders the two streams
 * and keeps one workspace per batch in flight. */
#define IDG_SSL_PLANNED 2
int idg_infonce_plan(const int64_t* users, const int64_t* items, int64_t B, int64_t num_users, int64_t n, int64_t d,
                     int mode, void* ws, void* stream);
int idg_infonce_cross_ex_f32(const float* view, int64_t n, int64_t d, const int64_t* users, const int64_t* items,
                             int64_t B, int64_t num_users, float temperature, float* loss, float* g, float grad_scale,
                             int planned, void* ws, void* stream);

/* ------------------------------------------------------------------------------------
 * DEVICE: dense Adam step  (torch.optim.Adam defaults, utility/utility_train/trainer.py:11,56:
 * betas (0.9, 0.999), eps 1e-8, no weight decay, no amsgrad).  step is 1-based.  The
 * hyper-parameters are doubles because torch forms lr/(1-beta1^t) and sqrt(1-beta2^t) in
 * double on the host before the fp32 kernels see them.
 * ---------------------------------------------------------------------------------- */
int idg_adam_step_f32(float* param, const float* grad, float* exp_avg, float* exp_avg_sq,
                      int64_t count, double lr, double beta1, double beta2, double eps,
                      int64_t step, void* stream);

/* out = a*x + b*y over `count` floats (y may be NULL: out = a*x; out may alias x or y).
 * Glue for the sharded path: adding an all-reduced layer into the running sum, the final
 * 1/(K+1) scaling, folding the replicated layer-0 term. */
int idg_lincomb_f32(float* out, const float* x, float a, const float* y, float b, int64_t count,
                    void* stream);

/* Row movers of the user-row-sharded step (no counterpart in the single-device reference): the ranks
 * exchange the batch's user rows through "guest" rows of their panels.
 * idg_rows_gather_f32: dst[t, :] = idx[t] >= 0 ? src[idx[t], :] : 0 for t < count (rows of d floats).
 * idg_rows_chain_add_f32: for every t with idx[t] >= 0, dst[idx[t], :] += src[t, :] + src[next[t], :] + ... following
 * `next` until -1, added in that order by one wave (the caller chains the occurrences of one destination in list
 * order): deterministic, no atomics.  idx / next: int64 device arrays. */
int idg_rows_gather_f32(float* dst, const float* src, const int64_t* idx, int64_t count, int64_t d, void* stream);
int idg_rows_chain_add_f32(float* dst, const float* src, const int64_t* idx, const int64_t* next, int64_t count,
                           int64_t d, void* stream);

/* The same movers in the forms the round-3 step uses (16-byte lanes; d % 4 == 0, panels 16-byte aligned):
 * idg_rows_gather2_f32: idg_rows_gather_f32 for two (dst, src) panel pairs sharing one index list in ONE launch (the
 *   batch users' final and ego rows go to the guest rows together); the second pair may be NULL.
 * idg_rows_scatter_f32: dst[idx[j], :] = src[j, :] (idx distinct, >= 0) — puts an exchanged compact row set back.
 * idg_rows_chain_store2_f32: idg_rows_chain_add_f32 that STORES the chain's sum (dst[idx[t]] = src[t] + src[next[t]] + ...)
 *   for two panel pairs: the destination panels are never zero-filled, their live rows are known by bitmap.
 * idg_rows_layer_mean_f32: out[ids[j], :] = (((a[ids[j]] + b[ids[j]]) + c[ids[j]]) + last[j]) / div with a, b, c nullable
 *   [*, d] panels and `last` a compact [count, d] buffer — LightGCN's layer mean (models/LightGCN.py:47-48, torch.mean of
 *   the stacked layers: left-to-right sum, true division) at the batch's item rows, whose last layer arrives as a
 *   compact exchanged row set.
 * idg_grad_tail_adam_f32: the item-row tail of a sharded step for a block of `rows` rows one rank owns, starting at
 *   global item row row0: s = (include_layer0 && live ? g + t : t) / cnt; s = live ? G + s : s; [G = s if store_grad];
 *   Adam(param, exp_avg, exp_avg_sq; s) — t = the reduced partial sums of the last backward product, g = d loss / d final
 *   and G = the regulariser's gradient, both meaningful only at live rows (bit row0 + r of live_bits; NULL = no live
 *   row).  Pointers address the block's first row.  Same operations in the same order as the last backward epilogue of
 *   the single-device step (autograd of models/LightGCN.py:43-48 + trainer.py:54-56). */
int idg_rows_gather2_f32(float* dst0, const float* src0, float* dst1, const float* src1, const int64_t* idx,
                         int64_t count, int64_t d, void* stream);
int idg_rows_scatter_f32(float* dst, const int64_t* idx, const float* src, int64_t count, int64_t d, void* stream);
int idg_rows_chain_store2_f32(float* dst0, const float* src0, float* dst1, const float* src1, const int64_t* idx,
                              const int64_t* next, int64_t count, int64_t d, void* stream);
int idg_rows_layer_mean_f32(float* out, const int64_t* ids, int64_t count, const float* a, const float* b,
                            const float* c, const float* last, float div, int64_t d, void* stream);
int idg_grad_tail_adam_f32(const float* t, const float* g, float* G, const uint32_t* live_bits, int64_t row0,
                           int64_t rows, int64_t d, int include_layer0, float cnt, int store_grad, float* param,
                           float* exp_avg, float* exp_avg_sq, double lr, double beta1, double beta2, double eps,
                           int64_t step, void* stream);
/* idg_rows_layer_mean_f32 for ANY number of earlier layers (GCN_layer is a free integer: configure/LightGCN.txt:12,
 * models/LightGCN.py:43-48): out[ids[j], :] = ((...(terms[0] + terms[1]) + ...) + terms[n_terms-1])[ids[j]] + last[j]) / div,
 * added left to right as torch.mean(torch.stack(...)) does; terms = HOST array of n_terms <= 15 device panels. */
int idg_rows_layer_mean_n_f32(float* out, const int64_t* ids, int64_t count, const float* const* terms, int n_terms,
                              const float* last, float div, int64_t d, void* stream);
/* ids[0..cap) <- the indices i (ascending) with flags[i] != 0, flags a DEVICE float vector of n entries; slots past the
 * last index repeat it (an empty vector: zeros); *count (device int64, nullable) <- the number of non-zero flags, which
 * may exceed cap (the list then holds the first cap).  Asynchronous, no host read-back: the sharded step agrees on the
 * item rows a batch touches (summed flag vectors) and moves a HOST-bounded number of rows through this
 * list — a repeated row is gathered, reduced and scattered more than once with the same value.
 * ws: idg_flags_compact_workspace_bytes(n). */
size_t idg_flags_compact_workspace_bytes(int64_t n);
int idg_flags_compact_f32(const float* flags, int64_t n, int64_t* ids, int64_t cap, int64_t* count, void* ws, void* stream);

/* The index-only preparation of ONE global batch of the user-row-sharded step as one call (id-grec_amd/sharded.py
 * HipKernels.prepare did this in twelve: the host cost of a step is its calls).  On `side_stream`, ordered after
 * `main_stream` through ev_fork:  users_bits = bitmap of own_users[0..n_own) (n_local_users bits); items_bits = bitmap
 * of pos[0..B) and neg[0..B) (n_items_padded bits); scatter_bits cleared (n_panel_rows bits: the gradient scatter flags
 * its stored rows there); the live-unit lists of users_bits on user_graph and of each item slice's share of items_bits
 * (bit offset slice_row0[j], a multiple of 32) on slice_graphs[j]; ev_rows recorded; the sorted scatter plan of
 * (guest_ids, pos, neg) over a panel of n_panel_rows rows whose items start at row n_local_users + B_cap
 * (idg_bpr_plan_f32) into plan_ws; ev_plan recorded.  Events are idg_event_create handles (NULL: skipped); unit-list
 * buffers as idg_graph_live_units_bytes(graph, B_cap) / (graph, 2 B_cap) (NULL: no list for that graph). */
typedef struct idg_shard_prep {
  const int64_t* own_users; int64_t n_own;       /* local ids of the owned triples' users */
  const int64_t* pos; const int64_t* neg; const int64_t* guest_ids; int64_t B, B_cap;
  uint32_t* users_bits; int64_t n_local_users;
  uint32_t* items_bits; int64_t n_items_padded;
  uint32_t* scatter_bits; int64_t n_panel_rows;
  const idg_graph* user_graph; void* user_units;
  int n_slices; const idg_graph* const* slice_graphs; const int64_t* slice_row0; void* const* slice_units;
  void* plan_ws;
  void* main_stream; void* side_stream;
  void* ev_fork; void* ev_rows; void* ev_plan;
} idg_shard_prep;
int idg_shard_prepare(const idg_shard_prep* prep);

/* ------------------------------------------------------------------------------------
 * DEVICE: ONE library call per training step
 * (the body of the reference's batch loop, utility/utility_train/trainer.py:42-56 — model(batch) ->
 *  [bpr, reg] (models/LightGCN.py:54-72 / models/MFBPR.py:29-42), backward(), Adam.step() — which this
 *  library's host previously drove as ~8 calls + Python slicing per step: the host was within 10 % of
 *  being the limit of a 0.26 ms step)
 * ---------------------------------------------------------------------------------- */
/* A step plan is built once per model: it names every buffer a step touches (all caller-owned — the library
 * still allocates no device memory) and owns only its events.  idg_step_run_f32 then enqueues, in order:
 *   - on `side_stream`, the index-only preparation of the NEXT batch (row bitmap, live work units, sorted scatter
 *     plan: idg_bpr_touch_rows, idg_graph_live_units, idg_bpr_plan_f32) into one of IDG_STEP_SLOTS slots;
 *   - on `stream`, this batch's step: K forward products with the layer mean at the batch's rows
 *     (idg_propagate_mean_f32 with out_rows), fused BPR + regulariser storing its gradient rows
 *     (idg_bpr_fused_f32, IDG_BPR_PLANNED | IDG_BPR_TOUCHED_PRESET), the backward chain with the Adam update in
 *     its last product's epilogue (idg_propagate_mean_bwd_adam_f32) and the end-of-step event;
 *   graph == NULL (MFBPR: no propagation): fused BPR on the raw tables with stored gradient rows, then
 *     idg_adam_rows_f32 (the dense Adam step reading the gradient at the batch's rows only).
 * Bit-identical to the same chain issued call by call.  IDG_STEP_PACED: the host blocks until the step before the
 * previous one has finished (two steps stay queued), after which a batch prepared a step ahead is complete when its
 * own step is enqueued and the step's stream does not wait for the side stream at all (a barrier packet costs ~4.5 us
 * whether or not its event has fired).
 * ids_token: any value that changes whenever the id arrays' STORAGE changes (a new epoch's triples): the side stream is
 * ordered behind `stream` once per token — the arrays may have been produced there — and 0 means "every call".
 * next_ids_token (idg_step_run_f32): the same for the storage of the NEXT batch's arrays, which need not be this
 * batch's (the first batch of a new epoch, a clone, a gather result). */
#define IDG_STEP_SLOTS 3
#define IDG_STEP_STORE_GRAD 1 /* the finished gradient is also written to `grad` (parity tests read it) */
#define IDG_STEP_PACED 2
typedef struct idg_step idg_step;
typedef struct idg_step_desc {
  const idg_graph* graph;      /* symmetric [n, n] handle, or NULL: no propagation */
  int64_t num_users, n, d;
  int n_layers, include_layer0;
  float reg_lambda;
  float* params;               /* [n, d] E0: users then items */
  float* grad;                 /* [n, d] d loss / d E0 (graph: rows outside the batch's receptive field are overwritten) */
  float* final_panel;          /* [n, d] layer mean (graph != NULL) */
  float* g_final;              /* [n, d] d loss / d final (graph != NULL) */
  float* exp_avg; float* exp_avg_sq;   /* [n, d] Adam moments */
  void* prop_ws;               /* idg_propagate_workspace_bytes(graph, d) (graph != NULL) */
  int64_t batch_capacity;      /* largest B of any call */
  uint32_t* slot_bitmap[IDG_STEP_SLOTS];  /* ceil(n / 32) words each, zeroed once */
  void* slot_units[IDG_STEP_SLOTS];       /* idg_graph_live_units_bytes(graph, 3 * batch_capacity) each (graph != NULL) */
  void* slot_bpr_ws[IDG_STEP_SLOTS];      /* idg_bpr_workspace_bytes(batch_capacity, d) each */
  void* side_stream;
  int flags;
} idg_step_desc;
int idg_step_create(const idg_step_desc* desc, idg_step** out);
int idg_step_destroy(idg_step* plan);
/* Prepare a batch a step ahead without running a step (the first batch of an epoch).  A prepared batch is identified by its
 * three pointers and B and is honoured by the NEXT idg_step_run_f32 only (any other prepared batch is dropped there): keep
 * the id arrays alive and unchanged until that call. */
int idg_step_prefetch(idg_step* plan, const int64_t* users, const int64_t* pos, const int64_t* neg, int64_t B,
                      uint64_t ids_token, void* stream);
/* next_* (nullable, next_B = 0: none): the batch the NEXT call will run.  loss: 2 floats [bpr, reg_lambda * reg].
 * adam_step: 1-based step count of this update; lr / betas / eps as idg_adam_step_f32.  flags: IDG_STEP_STORE_GRAD. */
int idg_step_run_f32(idg_step* plan, const int64_t* users, const int64_t* pos, const int64_t* neg, int64_t B,
                     const int64_t* next_users, const int64_t* next_pos, const int64_t* next_neg, int64_t next_B,
                     uint64_t ids_token, uint64_t next_ids_token, float* loss, int64_t adam_step, double lr, double beta1,
                     double beta2, double eps, int flags, void* stream);
/* The bitmap of the panel rows the LAST idg_step_run_f32 touched (its slot's; valid until that slot is reused two calls on). */
int idg_step_last_bitmap(const idg_step* plan, const uint32_t** out_bitmap);
/* Block the host until every step enqueued through the plan has finished AND every batch preparation it put on the side
 * stream has (a lookahead nobody ran still writes its slot's buffers): after this the plan's buffers may be released. */
int idg_step_synchronize(idg_step* plan);
/* Host-side accounting since the plan was made: out[0] = steps run, out[1] = nanoseconds spent inside idg_step_run_f32,
 * out[2] = of those, nanoseconds BLOCKED in the pacing wait (the device was the limit, not the host), out[3] = steps whose
 * stream did not have to wait for the side stream (the prepared batch was seen complete). */
int idg_step_stats(const idg_step* plan, int64_t out[4]);
/* Dense Adam (idg_adam_step_f32) whose gradient is read at the rows flagged in `rows` only and taken as ZERO elsewhere:
 * the update of a table whose step's gradient lives on the batch's <= 3B rows (MFBPR: no propagation spreads it) —
 * neither a zero-fill of the gradient panel nor its read.  Same arithmetic per element, so bit-identical to
 * idg_adam_step_f32 on a zero-filled panel. */
int idg_adam_rows_f32(float* param, const float* grad, const uint32_t* rows, float* exp_avg, float* exp_avg_sq, int64_t n,
                      int64_t d, double lr, double beta1, double beta2, double eps, int64_t step, void* stream);

/* ------------------------------------------------------------------------------------
 * DEVICE: full-rank scoring, train-positive masking, top-K
 * (get_rating_for_test models/LightGCN.py:74-80, MFBPR.py:44-49; batch_test.Test
 *  utility/utility_train/batch_test.py:59-68)
 * ---------------------------------------------------------------------------------- */
/* rating[b, i] = act(<user_panel[users[b]], item_panel[i]>), act = sigmoid if apply_sigmoid.
 * rating: [Bt, I] row-major.  The dense matrix the reference's evaluator mutates. */
int idg_score_dense_f32(const float* user_panel, const float* item_panel, const int64_t* users,
                        int64_t Bt, int64_t I, int64_t d, int apply_sigmoid, float* rating,
                        void* stream);

/* Fused: scores go from the MFMA accumulators through an LDS slab into the selection and are never
 * written to global memory.  Entries (b, i) with i in the train row of users[b] (excl_indptr
 * [num_users+1] int64 / excl_items int32 ascending, DEVICE pointers, indexed by user id; NULL = no
 * masking) rank as the value -1 (batch_test.py:62-65).  The k best per row are returned sorted by
 * (raw score descending, item id ascending) — sigmoid is monotone, so this is one of the orders
 * torch.topk may return on the sigmoid values; out_val holds act(score), -1 for masked entries.
 * out_idx int64 [Bt,k], out_val fp32 [Bt,k] (may be NULL).  1 <= k <= min(I, 1024): k <= 64 is one pass;
 * a larger k (torch.topk takes any k <= I, batch_test.py:68) runs one scoring pass per 64 ranks, each admitting only
 * keys strictly below the last one the previous pass emitted (same order, same values as one pass would give).
 * ws: idg_score_topk_workspace_bytes, 16-byte aligned (exact forms: one 64-key list per user and item chunk, 512 B x Bt x
 * chunks; threshold + collect form: up to 1024 candidate keys per user — 8 to 16 KB x Bt — plus the bf16 operand tables,
 * (d / 16 + 1) KiB per 32 items and (d / 16 + 2) x 32 B per user).
 * Throughput wants Bt large — hand over EVERY test user in one call (there is no [Bt, I] matrix to
 * bound; what bounds a call is this workspace): the catalogue is only cut into chunks when Bt/64 workgroups cannot fill
 * the chip. */
size_t idg_score_topk_workspace_bytes(int64_t Bt, int64_t I, int64_t d, int k);
/* Which form a call of this geometry takes, for tests and diagnostics: info[0] = 0 every wave alternates between scoring and
 * selecting, 1 producer / consumer waves on exact fp32 scores, 3 (calls of >= 8 user tiles over >= 32,768 items, d = 64 / 128 /
 * 256, k <= 42; IDG_TOPK_OPT_COLLECT = 0 turns it off) threshold + collect + exact finish on one-sided bf16 bounds of the exact
 * score: a per-user floor from a strided sample of the catalogue scored as LOWER bounds, one pass that appends every item
 * whose UPPER bound reaches the floor to the user's candidate list, exact fp32 re-scoring of the candidates near the top —
 * bit-identical to forms 0 / 1, 3x (d = 64) to 10x (d = 256) as fast.  info[1] = catalogue chunks, info[2] = 1 when
 * chunks start from a floor (form 1).  Form 3 with `ws` the workspace of a FINISHED call on `stream` (synchronises), else -1:
 * info[3] = users redone one by one, exactly, over the whole catalogue (candidate list overflowed or short; a NaN / infinity /
 * a norm beyond 2^60 in the user's row), info[4] = users the finish could not serve (counted up to just past the fall-back
 * threshold: a lower bound when info[5] = 1), info[5] = 1 when the call as a whole
 * was answered by the exact form, whose launches follow every form-3 call and return at once otherwise (more than
 * IDG_TOPK_OPT_FALLBACK_PERMILLE of the users unservable — tables that tie massively — or an irregular ITEM row: decided
 * on the device, no host read; then info[3] = 0), info[6] = 1 when an item row was irregular.  info[7] = 0 (reserved). */
int idg_score_topk_info(int64_t Bt, int64_t I, int64_t d, int k, const void* ws, int64_t info[8], void* stream);
/* Diagnostics of a form-3 call (info[0] == 3) that has been ENQUEUED on `stream` with workspace `ws` and did not fall back:
 * out_counts[b] (device, int32 [Bt]) = the number of candidates the collect pass counted for user b — how many items'
 * upper bounds reached the user's floor (a list beyond its capacity sends the user to the exact pass).  What the
 * candidate capacity and the fall-back threshold are sized by; depends on the tables' norm spread.  Asynchronous. */
int idg_score_topk_candidate_counts(int64_t Bt, int64_t I, int64_t d, int k, const void* ws, int32_t* out_counts, void* stream);
/* Knobs of the form choice, process-wide, for tests and tuning.  Their initial values are read from the environment ONCE,
 * by the first call that needs them (IDG_TOPK_FORM, IDG_TOPK_COLLECT, IDG_TOPK_FLOOR, IDG_TOPK_WGS, IDG_TOPK_CHUNKS,
 * IDG_TOPK_FALLBACK_PERMILLE).  *previous (nullable) receives the old value; value = IDG_TOPK_OPT_KEEP only reads;
 * which = IDG_TOPK_OPT_RESET restores every default.  Not thread-safe against concurrent idg_score_topk_* calls. */
#define IDG_TOPK_OPT_FORM 0               /* -1 (default): by geometry; 0 / 1 / 3: force a kernel (3 where its domain allows) */
#define IDG_TOPK_OPT_COLLECT 1            /* 1; 0: never form 3 */
#define IDG_TOPK_OPT_FLOOR 2              /* 1; 0: many-chunk calls of form 1 without their floor phase */
#define IDG_TOPK_OPT_WGS 3                /* 0; > 0: workgroups wanted per launch */
#define IDG_TOPK_OPT_CHUNKS 4             /* 0; > 0: catalogue chunks */
#define IDG_TOPK_OPT_FALLBACK_PERMILLE 5  /* 20: form 3 hands the whole call to the exact form beyond this share of unservable
                                             users (at least 4); < 0: never (every such user is redone one by one) */
#define IDG_TOPK_OPT_COUNT 6
#define IDG_TOPK_OPT_RESET (-1)
#define IDG_TOPK_OPT_KEEP INT64_MIN
int idg_score_topk_option(int which, int64_t value, int64_t* previous);
int idg_score_topk_f32(const float* user_panel, const float* item_panel, const int64_t* users,
                       int64_t Bt, int64_t I, int64_t d, const int64_t* excl_indptr,
                       const int32_t* excl_items, int k, int apply_sigmoid, int64_t* out_idx,
                       float* out_val, void* ws, void* stream);

/* ------------------------------------------------------------------------------------
 * DEVICE-LOCAL EVENTS: ordering between the caller's streams on ONE device (the step's stream and the stream that
 * prepares the next batch's index-only work).  Created without timing and WITHOUT the system-scope fence a default
 * HIP event performs when recorded (cache write-back/invalidate for the host's and other devices' benefit: ~7 us on
 * the recording stream); kernel boundaries already order memory at device scope.  Not for host-visible data.
 * ---------------------------------------------------------------------------------- */
int idg_event_create(void** out_event);
int idg_event_destroy(void* event);
int idg_event_record(void* event, void* stream);
int idg_stream_wait_event(void* stream, void* event);
int idg_event_query(void* event, int* done);
/* Block the calling host thread until the recorded work has completed (hipEventSynchronize): the engines use it to stay
 * at most two steps ahead of the device, so that a batch prepared on the side stream is KNOWN to be ready (idg_event_query)
 * when its step is enqueued and the step's stream does not have to wait for it. */
int idg_event_synchronize(void* event);

/* ------------------------------------------------------------------------------------
 * MULTI-GPU: RCCL collectives on the caller's stream (the reference is single-device,
 * utility/utility_train/trainer.py:8-74 — these serve the replicas / user-row shards around its step).
 * One communicator per process (one process per GPU).  Collectives are enqueued on `stream` in order with the
 * kernels around them.  The RCCL library is opened at run time, not linked: idg_comm_load(path) — NULL or "" for
 * the loader's default "librccl.so"; pass the copy the process already uses (PyTorch ships one).  Without it
 * every entry point below fails with IDG_E_UNSUPPORTED.
 * Bootstrap: rank 0 calls idg_comm_unique_id and hands the IDG_COMM_ID_BYTES bytes to every rank by any
 * out-of-band means; then EVERY rank calls idg_comm_create (collective: returns when all ranks have joined).
 * ---------------------------------------------------------------------------------- */
#define IDG_COMM_ID_BYTES 128
typedef struct idg_comm idg_comm;
int idg_comm_load(const char* librccl_path);
int idg_comm_rccl_version(int* version);
int idg_comm_unique_id(void* out_id /* IDG_COMM_ID_BYTES */);
int idg_comm_create(int rank, int world, const void* unique_id, int device, idg_comm** out);
int idg_comm_destroy(idg_comm* comm);
/* buf[0..count) <- sum over ranks (average != 0: mean over ranks), in place. */
int idg_allreduce_f32(idg_comm* comm, float* buf, int64_t count, int average, void* stream);
/* out[r * count .. (r + 1) * count) <- rank r's in[0..count).  In place when in == out + rank * count. */
int idg_allgather_f32(idg_comm* comm, const float* in, float* out, int64_t count, void* stream);
/* out[0..count) <- sum over ranks of their in[rank * count .. (rank + 1) * count): every rank hands in world * count
 * floats and keeps the reduced block it owns (ncclReduceScatter).  In place when out == in + rank * count.  The sharded
 * step ends its last backward exchange with this: each rank finishes the gradient and applies Adam for the item rows it
 * owns, and the updated rows go round by idg_allgather_f32. */
int idg_reduce_scatter_f32(idg_comm* comm, const float* in, float* out, int64_t count, void* stream);

/* ------------------------------------------------------------------------------------
 * MULTI-GPU, opt-in: the step's panel-sized exchanges as 24-bit rows with a rank-ordered sum (id-grec_amd/sharded.py
 * Packed24Comm; SURVEY.md 8e "fix the reduction order (rank order) so k-GPU runs are reproducible run to run").
 * An fp32 value travels as its upper 24 bits (sign, exponent, 15 mantissa bits; the dropped byte rounded to nearest even:
 * 2^-16 relative), four values in three 32-bit words — 3/4 of the bytes of the fp32 exchange.  An all-reduce becomes
 * pack -> idg_alltoall_f32 (block p to rank p) -> idg_reduce24_f32 (the N blocks added in rank order, packed again) ->
 * idg_allgather_f32 of the packed result -> unpack; a reduce-scatter stops after the sum (fp32 out).  Counts `n` are fp32
 * VALUES (multiples of 4); a packed array of n values has 3n/4 words.  Buffers 16-byte aligned.
 * ---------------------------------------------------------------------------------- */
int idg_pack24_f32(const float* src, uint32_t* dst, int64_t n, void* stream);
int idg_unpack24_f32(const uint32_t* src, float* dst, int64_t n, void* stream);
/* out <- blocks[0] + blocks[1] + ... + blocks[n_blocks - 1] in THAT order (block b: the 3n/4 words at blocks + b * 3n/4), as
 * fp32 (out_f32, nullable) and / or packed (out_packed, nullable; may be one of the input blocks' own words: each thread
 * reads its four values of every block before it writes). */
int idg_reduce24_f32(const uint32_t* blocks, int n_blocks, int64_t n, uint32_t* out_packed, float* out_f32, void* stream);
/* The same rank-ordered sum on fp32 blocks of n values each (no packing: `bench.py --reduce-order rank` with 32-bit panels —
 * RCCL's bytes on the links, one fixed sequence of adds).  out may be one of the blocks. */
int idg_reduce_blocks_f32(const float* blocks, int n_blocks, int64_t n, float* out, void* stream);
/* Block p (count floats) of `send` goes to rank p; block p of `recv` receives rank p's block for this rank (grouped
 * ncclSend / ncclRecv: on a fully connected node every peer's block travels over its own link).  Not in place.  The own
 * block is a device copy; flags = IDG_ALLTOALL_OWN_THROUGH_RCCL sends it through RCCL as well (tests on one device). */
#define IDG_ALLTOALL_OWN_THROUGH_RCCL 1
int idg_alltoall_f32(idg_comm* comm, const float* send, float* recv, int64_t count, int flags, void* stream);

#ifdef __cplusplus
}
#endif
#endif /* IDGREC_H */
