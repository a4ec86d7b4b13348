"""User-row-sharded LightGCN training step across the GPUs of one node.

The reference is single-device (models/LightGCN.py:36-72, utility/utility_train/trainer.py:36-56); this is new
functionality whose oracle is the single-GPU result.  Rank g owns a contiguous, nnz-balanced block of users: their
embedding rows (and Adam state), the rows R_g of the normalised interaction matrix (U_g x I) and the transposed block
R_g^T (I x U_g) — both with the GLOBAL d^-1/2 scaling, so the stacked blocks equal the single-GPU adjacency.  The item
TABLE is replicated (every rank gathers from all of it); its Adam state and its update are not: each rank owns 1/N of the
item rows.

One propagation layer:      X_U[g] <- R_g . X_I                           (local)
                            X_I    <- sum over ranks of R_g^T . X_U[g]    (RCCL over xGMI)
The item-side product of a layer needs local user rows only and is launched first, slice by slice, each slice's
collective right behind it on the communicator's stream; the user-side product is what waits for the previous layer's
exchange.  Only the exchanges whose consumers read every row move the [I, d] panel (K = 3: forward layer 1, the second
backward product, and the last backward product as a reduce-scatter); the others move the rows a batch touches.  After
the reduce-scatter each rank finishes the gradient and applies Adam for ITS item rows and all-gathers the updated rows
under the next step's first product, so the ranks' item tables are bit-identical by construction.

The layer loop is written once against two small interfaces — `kernels` (products with fused epilogues and row / input
bitmaps, row movers, fused BPR, the item-row tail, allocation) and `comm` (all-reduce, reduce-scatter, all-gather) — so
the world_size-2 gloo tests in tests/ drive it on CPU with a checker-backed stub that honours the same bitmaps (and
poisons every row a restricted product does not produce), while the product binds it to the HIP kernels (`HipKernels`)
and RCCL (`NativeComm` / `TorchComm`).
"""
import os

import numpy as np


# --------------------------------------------------------------------------- partitioning
def partition_users_by_nnz(user_degree, world):
    """Contiguous user blocks with ~equal stored entries.  Returns bounds[world + 1]."""
    deg = np.asarray(user_degree, dtype=np.int64)
    U = len(deg)
    csum = np.concatenate([[0], np.cumsum(deg)])
    total = csum[-1]
    bounds = [0]
    for r in range(1, world):
        target = total * r // world
        b = int(np.searchsorted(csum, target, side="left"))
        bounds.append(min(max(b, bounds[-1]), U))
    bounds.append(U)
    return np.asarray(bounds, dtype=np.int64)


def shard_adjacency(indptr, indices, values, num_users, num_items, u_lo, u_hi):
    """From the global normalised adjacency CSR ([U+I, U+I], users first) cut
    R_g   (rows = users [u_lo, u_hi), columns = items 0..I)  and
    R_g^T (rows = items 0..I, columns = users re-based to 0..u_hi-u_lo).
    Values are copied, never re-normalised."""
    U, I = int(num_users), int(num_items)
    indptr = np.asarray(indptr, dtype=np.int64)
    indices = np.asarray(indices)
    values = np.asarray(values, dtype=np.float32)
    # user rows: every column is an item (>= U)
    s, e = indptr[u_lo], indptr[u_hi]
    ui_ptr = indptr[u_lo:u_hi + 1] - s
    ui_idx = (indices[s:e].astype(np.int64) - U).astype(np.int32)
    ui_val = values[s:e].copy()
    # item rows: keep the columns that fall in this rank's user block
    s2, e2 = indptr[U], indptr[U + I]
    cols = indices[s2:e2].astype(np.int64)
    keep = (cols >= u_lo) & (cols < u_hi)
    row_of = np.repeat(np.arange(I, dtype=np.int64), np.diff(indptr[U:U + I + 1]))
    iu_cnt = np.bincount(row_of[keep], minlength=I)
    iu_ptr = np.concatenate([[0], np.cumsum(iu_cnt)]).astype(np.int64)
    iu_idx = (cols[keep] - u_lo).astype(np.int32)
    iu_val = values[s2:e2][keep].copy()
    return (ui_ptr, ui_idx, ui_val), (iu_ptr, iu_idx, iu_val)


def shard_adjacency_from_edges(users, items, num_users, num_items, u_lo, u_hi):
    """The same two pieces as shard_adjacency, straight from the interaction list — sorted by (user, item), no
    duplicate pairs (what synth.generate returns) — without building the global [U+I, U+I] CSR first: at configs[4]
    size that build is 4e8 entries and ~55 s per rank, of which a rank keeps 1/N.  Values are the reference's float32
    arithmetic (data_graph.py:46-51: np.power(rowsum, -0.5) in float32, (D.A).D left to right), i.e. bit for bit what
    host.build_norm_adj + shard_adjacency give (tests/test_sharded.py)."""
    U, I = int(num_users), int(num_items)
    users = np.asarray(users, dtype=np.int64)
    items = np.asarray(items, dtype=np.int64)
    if len(users) > 1:
        du = np.diff(users)
        if (du < 0).any() or ((du == 0) & (np.diff(items) <= 0)).any():
            raise ValueError("shard_adjacency_from_edges needs edges sorted by (user, item) without duplicates")
    deg = np.concatenate([np.bincount(users, minlength=U), np.bincount(items, minlength=I)]).astype(np.float32)
    with np.errstate(divide="ignore"):
        dinv = np.power(deg, np.float32(-0.5))
    dinv[np.isinf(dinv)] = 0.0
    s, e = np.searchsorted(users, [u_lo, u_hi], side="left")
    su, si = users[s:e], items[s:e]
    ui_ptr = np.concatenate([[0], np.cumsum(np.bincount(su - u_lo, minlength=u_hi - u_lo))]).astype(np.int64)
    ui_idx = si.astype(np.int32)
    ui_val = (dinv[su] * np.float32(1.0)) * dinv[U + si]
    order = np.argsort(si, kind="stable")  # item rows, users ascending inside each
    iu_ptr = np.concatenate([[0], np.cumsum(np.bincount(si, minlength=I))]).astype(np.int64)
    iu_idx = (su[order] - u_lo).astype(np.int32)
    iu_val = (dinv[U + si[order]] * np.float32(1.0)) * dinv[su[order]]
    return (ui_ptr, ui_idx, ui_val.astype(np.float32)), (iu_ptr, iu_idx, iu_val.astype(np.float32))


# --------------------------------------------------------------------------- the step
class GlobalBatch:
    """Index-only description of ONE global batch as one rank sees it (built on the host by ShardedEngine.make_batch
    from the batch's global ids; every array lives where the kernels want it).  B triples; triple t owns guest row t.
      pos, neg   int64 [B]  global item ids
      own_src    int64 [B]  local id of triple t's user if this rank owns it, else -1 (fills the guest rows)
      head_dst   int64 [B]  the same id for the FIRST owned occurrence of a user in the batch, else -1
      nxt        int64 [B]  batch position of the next occurrence of triple t's user, -1 at the end of its chain
      own_users  int64 [B_g]  local ids of the owned triples' users: the user rows this rank's restricted products have
                 to produce / may gather from
      items      int64 [n_items <= 2B]  the batch's distinct item ids, ascending: the only item rows of the LAST forward
                 layer anybody reads (the BPR loss gathers pos / neg rows) — what that layer's exchange carries
      touched_bound  host int or None: an upper bound of the number of item rows the batch touches (its own items + the
                 items its users interacted with) computed from the GLOBAL user degrees — the same number on every rank,
                 known before the step: the touched-item exchanges are sized by it and need no host read-back"""
    __slots__ = ("B", "pos", "neg", "own_src", "head_dst", "nxt", "own_users", "n_owned", "key", "items", "n_items",
                 "touched_bound")
    _next_key = 0


class ShardedEngine:
    """One rank's share of the LightGCN step (models/LightGCN.py:36-72 + utility/utility_train/trainer.py:42-56 cut by
    user rows; any number of propagation layers, GCN_layer in configure/LightGCN.txt:12).  Arrays are whatever `kernels`
    allocates (torch CUDA tensors in the product).

    Layout.  Every local panel has rows [0, U_g) = this rank's users, [U_g, U_g + B) = B GUEST rows (row U_g + t carries
    triple t's user, whoever owns it), then ALL items padded to I_p rows ([U_g + B, U_g + B + I_p)).  The item rows are cut
    into S slices (the same cuts on every rank; every slice a multiple of the world size long, the last one padded), and
    rank r OWNS the r-th N-th of every slice: it holds the Adam moments of those item rows only.

    What is exact and what is exchanged.  The batch is GLOBAL (every rank sees all B triples, as the single-device step
    does): after the forward propagation the owners copy the batch users' final and ego rows into the guest rows, two
    all-reduces of [B, d] (x + 0 + ... + 0: exact) hand them to every rank, and every rank evaluates the WHOLE batch's
    BPR loss — so the loss and the item-side gradient g_I are complete and bit-identical on every rank without any
    [I, d] exchange, and the user-side gradient rows flow back from the guest rows to their owners' rows (chains in batch
    order).  Nothing of size [n, d] is ever zero-filled: BPR and the chain mover STORE the rows they reach, and every
    consumer reads gradient panels through the batch's row bitmaps (kernels.prepare).
    Per training step at K = 3: forward layer 1 and the second backward product travel as sliced all-reduces of the
    [I, d] panel; forward layer K - 1 and the first backward product as the item rows the batch's users touch (agreed
    through an [I] flag vector: _agree_touched_items); the last forward layer as the batch's <= 2B item rows; and the
    LAST backward product as a reduce-scatter: each rank finishes the gradient and applies Adam for the item rows it
    owns (idg_grad_tail_adam_f32) and the UPDATED rows go round by an all-gather that runs under the next step's first
    item-side product (which needs the local user rows only).  No O(I d) work is replicated: the only replicated
    per-rank work is O(B d) (loss, guest rows) and the [I] flag vector."""

    def __init__(self, kernels, comm, ui_csr, iu_csr, n_local_users, num_items, dim, n_layers, include_layer0=True,
                 reg_lambda=1e-4, lr=1e-3, batch_sparsity=True, batch_size=1024, user_lo=0, n_slices=None, item_cuts=None,
                 live_rows_cap=None, live_rows_min_bytes=64 << 20, store_grad=True, global_user_degree=None):
        """batch_sparsity: use the batch's row bitmaps (kernels.prepare) — restricted products, stored gradient rows, no
        fills; False = every product dense over zero-filled gradient panels (the plain form, kept as an A/B check).
        batch_size: capacity of the guest rows (the global batch size).  user_lo: global id of this rank's first user.
        n_slices: row slices of R_g^T whose collectives overlap the following slices' products (default: 4 once the
        item panel reaches 256 MB, else 1); item_cuts: the slices' row bounds — the same on every rank; default: equal
        row counts (callers that know the global item degrees pass entry-balanced cuts).  live_rows_cap: rows of the
        compact buffer of the touched-item exchanges (default 64 per triple; the same on every rank);
        live_rows_min_bytes: item panels smaller than this skip the touched-item forms (tests pass 0).
        store_grad: keep the finished gradient in G — the user rows and the OWNED item rows (tests read it; the step
        does not: without it the last products feed Adam and write no gradient panel).
        global_user_degree: int array [U] of EVERY user's number of train items (the same array on every rank).  With it
        the number of item rows a batch touches is bounded on the host (make_batch), the touched-item exchanges move that
        many rows through a fixed-capacity id list built on the device, and the step has no host synchronisation; without
        it the ranks read the list's length back (one synchronisation per step)."""
        self.k, self.comm = kernels, comm
        self.world, self.rank = int(getattr(comm, "world", 1)), int(getattr(comm, "rank", 0))
        self.batch_sparsity = bool(batch_sparsity)
        self._prepared = {}
        self.Ug, self.I, self.d, self.K = int(n_local_users), int(num_items), int(dim), int(n_layers)
        if not 1 <= self.K <= 15:
            raise ValueError("ShardedEngine: 1 <= n_layers <= 15; got %d" % self.K)
        self.user_degree = None if global_user_degree is None else np.asarray(global_user_degree, dtype=np.int64)
        self.B, self.lo = int(batch_size), int(user_lo)
        self.c0 = 1 if include_layer0 else 0
        self.cnt = float(self.K + self.c0)
        self.reg_lambda, self.lr = float(reg_lambda), float(lr)
        self.store_grad = bool(store_grad)
        self.G_ui = kernels.make_graph(*ui_csr, self.Ug, self.I)
        if n_slices is None:
            n_slices = 4 if self.I * self.d * 4 >= (256 << 20) else 1
        iu_ptr, iu_idx, iu_val = iu_csr
        iu_ptr = np.asarray(iu_ptr, dtype=np.int64)
        if item_cuts is None:
            item_cuts = np.linspace(0, self.I, max(1, min(int(n_slices), self.I)) + 1).astype(np.int64)
        cuts = np.asarray(item_cuts, dtype=np.int64).copy()
        # inner cuts on multiples of 32 x world rows: a slice's share of an item-row bitmap starts on a word boundary and
        # every inner slice divides evenly among the ranks (the rounding is a function of the cuts and the world size
        # alone, so the ranks agree on it)
        q = 32 * self.world
        cuts[1:-1] = np.minimum((cuts[1:-1] + q // 2) // q * q, self.I // q * q)
        cuts = np.maximum.accumulate(cuts)
        assert cuts[0] == 0 and cuts[-1] == self.I and (np.diff(cuts) >= 0).all(), "item_cuts must tile [0, I]"
        # slices: (graph, r0, r1, r1p) — rows [r0, r1) exist, [r0, r1p) is the slice's share of the padded panel
        self.slices = []
        for r0, r1 in zip(cuts[:-1], cuts[1:]):
            r0, r1 = int(r0), int(r1)
            if r1 == r0:
                continue
            e0, e1 = int(iu_ptr[r0]), int(iu_ptr[r1])
            g = kernels.make_graph(iu_ptr[r0:r1 + 1] - e0, iu_idx[e0:e1], iu_val[e0:e1], r1 - r0, self.Ug)
            self.slices.append([g, r0, r1, r1])
        last = self.slices[-1]
        last[3] = last[1] + -(-(last[2] - last[1]) // self.world) * self.world
        self.slices = [tuple(sl) for sl in self.slices]
        self.Ip = self.slices[-1][3]
        # the blocks of item rows this rank owns: (first row, rows, offset into the compact moment arrays)
        self.own, off = [], 0
        for _, r0, _, r1p in self.slices:
            c = (r1p - r0) // self.world
            self.own.append((r0 + self.rank * c, c, off))
            off += c
        n = self.Ug + self.B + self.Ip
        z = kernels.zeros
        self.P, self.G = z((n, dim)), z((n, dim))
        self.FIN = z((n, dim))
        self.GF = z((n, dim))   # d loss / d FIN
        self.MU, self.VU = z((self.Ug, dim)), z((self.Ug, dim))        # Adam moments: the owned user rows ...
        self.MI, self.VI = z((max(off, 1), dim)), z((max(off, 1), dim))  # ... and the owned item rows (1/N of the table)
        # 24-bit panel exchange (Packed24Comm): the replicated item table then holds 24-bit values on EVERY rank (the owner's
        # copy too: replicas stay bit-identical), so the owner keeps the fp32 MASTER of its rows here — Adam updates the
        # master, the all-gather sends it packed.  Filled from the table by the first training step.
        self.packed = bool(getattr(comm, "packed", False))
        self.by_blocks = bool(getattr(comm, "by_blocks", False))  # (an explicit exchange, packed or not: slices by rank blocks)
        self.MP = z((max(off, 1), dim)) if self.packed else None
        self._master_ready = False
        self.XU = [z((self.Ug, dim)) for _ in range(max(self.K - 1, 1))]
        self.XI = [z((self.Ip, dim)) for _ in range(self.K)]
        # more than three earlier layers do not fit one epilogue (sum_in .. sum_in3): from K = 4 on the user-side layer sum
        # is carried from product to product (sum_in -> sum_out, left to right as torch.mean(torch.stack(...)) adds)
        self.SU = z((self.Ug, dim)) if self.K + self.c0 > 4 else None
        self.CI = z((2 * self.B, dim))  # the batch's item rows, compact (last forward layer)
        # flags of the item rows some rank's batch users touch, and those rows, compact (up to 64 per triple)
        self.FL = z((self.I,))
        self.live_rows_min_bytes = int(live_rows_min_bytes)
        default_cap = (128 if self.user_degree is not None else 64) * self.B  # (a bound is looser than a count)
        self.CS = z((max(1, min(self.I, default_cap if live_rows_cap is None else int(live_rows_cap))), dim))
        self._touch_misses, self._touch_skip = 0, 0   # consecutive overflows of CS; steps left before asking again
        self.timeline = None  # StepTimeline while instrumented steps run (bench.py, after the timed region)
        self.loss = z((2,))
        self.guest_ids = kernels.to_device(np.arange(self.Ug, self.Ug + self.B, dtype=np.int64))
        # views handed to the kernels every step, made once (slicing a tensor costs the host ~1.5 us)
        self.P_u, self.P_i = self._u(self.P), self._i(self.P)
        self.FIN_u, self.FIN_i = self._u(self.FIN), self._i(self.FIN)
        self.GF_u, self.GF_i = self._u(self.GF), self._i(self.GF)
        self.G_u, self.G_i = self._u(self.G), self._i(self.G)
        self._views = {}
        self._all_item_ids = None
        self._all_items_bits = None
        self._ag = []          # all-gathers of the updated item rows still in flight (waited before P_I is read again)
        self.touched_items = None
        self.step_count = 0

    def _u(self, a):
        return a[: self.Ug]

    def _guest(self, a, count=None):
        return a[self.Ug: self.Ug + (self.B if count is None else count)]

    def _i(self, a):
        return a[self.Ug + self.B:]

    def _slice_rows(self, panel_i, j, padded=False):
        """Rows of slice j of an item-row panel ([I_p, d]); padded: up to the slice's share of the padding."""
        key = (id(panel_i), j, padded)
        v = self._views.get(key)
        if v is None:
            _, r0, r1, r1p = self.slices[j]
            v = self._views[key] = (panel_i, panel_i[r0:(r1p if padded else r1)])  # (keeps the panel alive: id() stays its)
        return v[1]

    def item_rows(self, a):
        """The real item rows of a local panel (without the padding)."""
        return a[self.Ug + self.B: self.Ug + self.B + self.I]

    # ---- index-only preparation of a global batch (host)
    def make_batch(self, users, pos, neg):
        """users: GLOBAL user ids [B' <= B] (numpy), pos / neg: global item ids."""
        users = np.asarray(users, dtype=np.int64)
        Bc = len(users)
        assert 0 < Bc <= self.B, "batch of %d triples, engine built for up to %d" % (Bc, self.B)
        local = users - self.lo
        owned = (local >= 0) & (local < self.Ug)
        own_src = np.where(owned, local, -1)
        # chains over the occurrences of one user, in batch order (stable sort by user, then by position)
        order = np.argsort(users, kind="stable")
        su = users[order]
        nxt = np.full(Bc, -1, dtype=np.int64)
        same = su[1:] == su[:-1]
        nxt[order[:-1][same]] = order[1:][same]
        first = np.ones(Bc, dtype=bool)
        first[order[1:][same]] = False
        gb = GlobalBatch()
        to = self.k.to_device
        gb.B = Bc
        gb.pos, gb.neg = to(np.asarray(pos, dtype=np.int64)), to(np.asarray(neg, dtype=np.int64))
        gb.own_src = to(own_src.astype(np.int64))
        gb.head_dst = to(np.where(owned & first, local, -1).astype(np.int64))
        gb.nxt = to(nxt)
        gb.n_owned = int(owned.sum())
        gb.own_users = to(local[owned].astype(np.int64))
        items = np.unique(np.concatenate([np.asarray(pos, dtype=np.int64), np.asarray(neg, dtype=np.int64)]))
        gb.items, gb.n_items = to(items), len(items)
        gb.touched_bound = None
        if getattr(self, "user_degree", None) is not None:
            # every distinct batch user contributes at most its degree, the batch at most its own items: >= the union
            gb.touched_bound = int(min(self.I, self.user_degree[su[np.concatenate([[True], ~same])]].sum() + len(items)))
        GlobalBatch._next_key += 1
        gb.key = GlobalBatch._next_key  # (a counter: id(gb) can come back after a skipped batch is collected)
        return gb

    # ---- collectives
    def _wait(self, works):
        for w in works:
            self.comm.wait(w)

    def _wait_item_table(self):
        """The previous step's all-gathers of the updated item rows: P_I is whole again after them."""
        self._wait(self._ag)
        self._ag = []

    def _item_side(self, X_u, Y_i, out_bits=None, x_rows=None, addend=None, mask=None, reduce="all", tag="panel"):
        """Y_i[slice] = R_g^T[slice] . X_u (+ addend, on rank 0 only: the ranks' partials are summed), slice by slice;
        each slice's collective is issued as soon as the slice exists, so it runs under the following slices' products
        and under whatever the caller launches next.  reduce: "all" = all-reduce of the slice, "scatter" = reduce-scatter
        (this rank keeps its own block of the slice), None = no exchange (the caller moves rows).  Returns the works."""
        k, works = self.k, []
        add = addend if self.rank == 0 else None
        # packed exchange: a slice whose partial goes straight into an all-reduce / reduce-scatter is written PACKED by the
        # product itself, into the exchange's send buffer (nobody reads the fp32 partial: the sum overwrites it)
        target = getattr(self.comm, "packed_target", None) if (self.packed and reduce in ("all", "scatter") and out_bits is None) else None
        for j, (g, r0, r1, r1p) in enumerate(self.slices):
            y = self._slice_rows(Y_i, j)
            y24 = target(self._slice_rows(Y_i, j, padded=True), (r1 - r0) * self.d) if target is not None else None
            k.spmm(g, X_u, Y=None if y24 is not None else y, addend=None if add is None else self._slice_rows(add, j),
                   mask=None if (add is None or mask is None) else k.bits_from(mask, r0),
                   out_rows=None if out_bits is None else k.bits_from(out_bits, r0), x_rows=x_rows,
                   **({"Y24": y24} if y24 is not None else {}))
            if reduce == "all":
                self.comm.tag, self.comm.slice = tag, j
                # (the packed exchange cuts a slice into one block per rank: the padded view — its padding rows are zero)
                works.append(self.comm.all_reduce_async(self._slice_rows(Y_i, j, padded=True) if self.by_blocks else y))
            elif reduce == "scatter":
                self.comm.tag, self.comm.slice = tag, j
                works.append(self.comm.reduce_scatter_async(self._slice_rows(Y_i, j, padded=True)))
        self.comm.slice = None
        return works

    def _sum_rows(self, panel, rows, tag="rows"):
        """all-reduce of the panel's rows `rows` = (ids, n) through the compact buffer (synchronous: small).  ids may
        repeat its last entry (a list padded to a host-side bound): the repeated row is gathered, summed and written
        back more than once, with the same value."""
        ids, n = rows
        self.k.gather_rows(self.CS[:n], panel, ids)
        self.comm.tag = tag
        self.comm.wait(self.comm.all_reduce_async(self.CS[:n]))
        self.k.scatter_rows(panel, ids, self.CS[:n])

    def _agree_touched_items(self, prep, gb):
        """The item rows a training batch touches beyond its own positives / negatives: the items its USERS interacted
        with.  Forward layer K - 1 is read there only (by the last user-side product, restricted to the batch's users,
        and by FIN at the batch's items) and the first backward product's partials are zero elsewhere, so both travel as
        these rows instead of the [I, d] panel, and the products around them are restricted to them.
        One rank marks the bitmap locally.  Several ranks flag the items of the batch users they own
        (idg_graph_flag_cols over their block of R), add the batch's items and sum the flags ([I] floats: 20 MB at
        configs[4]); the ascending id list is built ON THE DEVICE into a list of gb.touched_bound slots (a host-side
        bound from the global user degrees, the same on every rank; slots past the last id repeat it:
        idg_flags_compact_f32) — no host synchronisation.  Without the degrees (global_user_degree=None) the list's
        length is read back: one synchronisation per step, and after three consecutive batches that overflow the
        compact buffer the engine stops asking for 64 steps.
        Sets prep.touched (bitmap) and self.touched_items = (ids, n) (several ranks) or leaves both None: K < 2, a panel
        too small to be worth it, or more rows than the compact buffer holds."""
        self.touched_items = None
        if prep is None or self.K < 2 or self.I * self.d * 4 < self.live_rows_min_bytes:
            return
        k = self.k
        if self.world == 1:
            k.touched_local(self, prep, gb)
            return
        cap = self.CS.shape[0]
        tl = self.timeline
        if gb.touched_bound is not None:
            if gb.touched_bound > cap:   # known before anything is launched, and the same on every rank
                return
            n = min(cap, -(-gb.touched_bound // 256) * 256)  # (whole 256-row blocks: fewer distinct collective sizes)
            k.flag_touched_items(self, prep, gb, self.FL)
            self.comm.tag = "flags"
            self.comm.wait(self.comm.all_reduce_async(self.FL))
            ids = k.compact_ids(self.FL, n)
            self.touched_items = (ids, n)
            k.touched_from_ids(self, prep, ids, n)
            return
        if self._touch_skip > 0:
            self._touch_skip -= 1
            return
        k.flag_touched_items(self, prep, gb, self.FL)
        self.comm.tag = "flags"
        self.comm.wait(self.comm.all_reduce_async(self.FL))
        t0 = tl.host_sync_begin() if tl is not None else None
        ids, n = k.nonzero_ids(self.FL)
        if tl is not None:
            tl.host_sync_end(t0)
        if n == 0 or n > cap:
            self._touch_misses += 1
            if self._touch_misses >= 3:
                self._touch_misses, self._touch_skip = 0, 64
            return
        self._touch_misses = 0
        self.touched_items = (ids, n)
        k.touched_from_ids(self, prep, ids, n)

    # ---- forward: FIN = mean_k A^k P at the rows the caller reads (a training step: the batch's; evaluation: all)
    def propagate(self, prep=None, gb=None):
        """Layer k: X_I(k) = sum over ranks of R_g^T X_U(k-1),  X_U(k) = R_g X_I(k-1).  The item-side partial of a layer
        needs only LOCAL user rows, so it is launched before the previous layer's exchange is waited for; the user-side
        product is what waits.  With a batch (gb) the step reads FIN at the batch's rows only, and each layer is produced
        on the rows its consumers read: layer K at the batch's users / items, layer K - 1 at the users near the batch's
        items / the touched items (kernels.prepare and _agree_touched_items), earlier layers everywhere.  The layer mean
        is formed once, by the last product's epilogue (users) and by one rows kernel (the batch's items), in
        torch.mean(torch.stack(...))'s left-to-right order.  Evaluation calls propagate() without a batch: every row."""
        k, K, c0, cnt = self.k, self.K, self.c0, self.cnt
        P_u, P_i, fin_u, fin_i = self.P_u, self.P_i, self.FIN_u, self.FIN_i
        train = gb is not None
        self.touched_items = None
        touched = near = None
        items_bits = prep.items if (train and prep is not None) else None
        users_bits = prep.own_users if (train and prep is not None) else None
        pending = []  # collectives of the previous layer's item side
        # the user-side layer sum: at most three earlier terms fit the last product's epilogue; beyond that (K + c0 > 4)
        # every product adds its output to the running sum SU (sum_in -> sum_out, the left-to-right order of
        # torch.mean(torch.stack(...)), as idg_propagate_mean_f32 does on one device) and the last one divides
        chain = self.SU is not None
        acc = [P_u] if c0 else []  # terms not yet folded into SU (chain form)
        for layer in range(1, K + 1):
            last, pre = layer == K, layer == K - 1
            if pre and train:
                # agreed as late as possible: the flag exchange queues behind the collectives already issued (the previous
                # step's all-gathers, layer 1's all-reduce), which the products launched so far overlap
                self._agree_touched_items(prep, gb)
                touched = getattr(prep, "touched", None) if prep is not None else None
                near = getattr(prep, "near", None) if touched is not None else None
            xu_prev = P_u if layer == 1 else self.XU[layer - 2]
            xi_prev = P_i if layer == 1 else self.XI[layer - 2]
            xi_new = self.XI[layer - 1]
            # item side of this layer (input: local user rows)
            if last and train:
                works = self._item_side(xu_prev, xi_new, out_bits=items_bits, reduce=None)
                k.gather_rows(self.CI[:gb.n_items], xi_new, gb.items)
                self.comm.tag = "F%d.items" % layer
                works = [self.comm.all_reduce_async(self.CI[:gb.n_items])]
            elif pre and touched is not None:
                works = self._item_side(xu_prev, xi_new, out_bits=touched, reduce=None)
                if self.touched_items is not None:
                    self._sum_rows(xi_new, self.touched_items, tag="F%d.touched" % layer)
            else:
                works = self._item_side(xu_prev, xi_new, tag="F%d.panel" % layer)
            # user side (input: the previous layer's item rows, exchanged; layer 1: the item table itself)
            self._wait(pending)
            if layer == 1:
                self._wait_item_table()
            if last:
                terms = acc if chain else ([P_u] if c0 else []) + self.XU[: K - 1]
                k.spmm(self.G_ui, xi_prev, sums=terms, sum_out=fin_u, div=cnt, out_rows=users_bits)
            elif chain and acc:
                k.spmm(self.G_ui, xi_prev, Y=self.XU[layer - 1], sums=acc, sum_out=self.SU, out_rows=near if pre else None)
                acc = [self.SU]
            else:
                k.spmm(self.G_ui, xi_prev, Y=self.XU[layer - 1], out_rows=near if pre else None)
                if chain:
                    acc = [self.XU[layer - 1]]  # (no layer 0 in the mean: the sum starts with layer 1)
            pending = works
        self._wait(pending)
        # the layer mean at the item rows: the batch's (their last layer arrived as the compact row set), or all
        terms = ([P_i] if c0 else []) + self.XI[: K - 1]
        if train:
            k.layer_mean(fin_i, gb.items, terms, self.CI[:gb.n_items], cnt)
        else:
            if self._all_item_ids is None:
                self._all_item_ids = k.to_device(np.arange(self.I, dtype=np.int64))
            k.layer_mean(fin_i, self._all_item_ids, terms, self.XI[K - 1][: self.I], cnt)
        return self.FIN

    # ---- backward of the above given GF = d loss / d FIN (g_I complete on every rank, g_U at the owners' rows; live
    #      rows: the batch's), accumulated onto G (which holds the regulariser gradient at the same rows), Adam included
    def propagate_backward(self, prep, gb, adam_step):
        """Horner steps h <- A.h + g, then gE0 = (A.h + c0.g)/cnt.  In block form (A.h)_U = R_g h_I (local), (A.h)_I =
        sum over ranks of R_g^T h_U.  As in the forward pass the item-side partial of a step needs only the LOCAL h_U and
        is launched before the previous exchange is waited for.  Step 1's inputs live on the batch's rows, its outputs on
        the touched items / the near users; from step 2 on everything is dense.  The LAST step's item side ends in a
        reduce-scatter: this rank finishes the gradient and applies Adam for the item rows it owns, slice by slice as each
        slice's collective lands, and sends the updated rows round (all-gathers left in flight: _wait_item_table); its
        user side applies Adam to the owned user rows in the product's epilogue."""
        k, K, c0, cnt = self.k, self.K, self.c0, self.cnt
        g_u, g_i, G_u, G_i, P_i = self.GF_u, self.GF_i, self.G_u, self.G_i, self.P_i
        sparse = prep is not None
        users_bits = prep.own_users if sparse else None      # live rows of g_U / reg_U (the owned batch users)
        items_bits = prep.items if sparse else self._ones_items()  # live rows of g_I / reg_I (the batch's items)
        touched = getattr(prep, "touched", None) if sparse else None
        near = getattr(prep, "near", None) if touched is not None else None
        h_u, h_i = g_u, g_i
        live_u, live_i = users_bits, (items_bits if sparse else None)
        pending = []
        for step in range(1, K):
            t_i, t_u = self.XI[step - 1], self.XU[step - 1]
            first = step == 1 and touched is not None
            # item side: sum over ranks of R_g^T h_U, + g_I (added once: by rank 0's partial)
            if first:
                works = self._item_side(h_u, t_i, out_bits=touched, x_rows=live_u, addend=g_i, mask=items_bits, reduce=None)
                if self.touched_items is not None:
                    self._sum_rows(t_i, self.touched_items, tag="B%d.touched" % step)
            else:
                works = self._item_side(h_u, t_i, x_rows=live_u, addend=g_i, mask=items_bits if sparse else None,
                                        tag="B%d.panel" % step)
            # user side: R_g h_I + g_U
            self._wait(pending)
            k.spmm(self.G_ui, h_i, Y=t_u, addend=g_u, mask=users_bits, out_rows=near if first else None, x_rows=live_i)
            pending = works
            h_u, h_i = t_u, t_i
            live_u, live_i = (near, touched) if first else (None, None)
        # last step, scaled by 1 / cnt
        t_i = self.XI[K - 1]
        works = self._item_side(h_u, t_i, x_rows=live_u, reduce="scatter", tag="B%d.reduce_scatter" % K)
        self._wait(pending)
        adam = (self.P_u, self.MU, self.VU, self.lr, adam_step)
        if live_i is None:   # a dense launch: the owned users' Adam update rides in its epilogue
            k.spmm(self.G_ui, h_i, sums=[g_u] if c0 else [], sum_out=G_u, div=cnt, accumulate=True, mask=users_bits, adam=adam,
                   discard_grad=not self.store_grad)
        else:                # (K <= 2 with restricted inputs: the epilogue form needs the dense kernel)
            k.spmm(self.G_ui, h_i, sums=[g_u] if c0 else [], sum_out=G_u, div=cnt, accumulate=True, mask=users_bits,
                   x_rows=live_i)
            k.adam(self.P_u, G_u, self.MU, self.VU, self.lr, adam_step)
        # the owned item rows: finish the gradient, Adam, and send the updated rows round
        for j, (w, (o0, c, off)) in enumerate(zip(works, self.own)):
            self.comm.wait(w)
            blocks = self._own_blocks(j, t_i)
            if c > 0:
                k.item_tail(blocks[0], blocks[1], blocks[2], items_bits, o0, c0, cnt, self.store_grad, blocks[3],
                            blocks[4], blocks[5], self.lr, adam_step)
            self.comm.tag, self.comm.slice = "item_table.all_gather", j
            self._ag.append(self.comm.all_gather_async(self._slice_rows(P_i, j, padded=True), blocks[3]))
        self.comm.slice = None
        return self.G

    def _own_blocks(self, j, t_i):
        """The rows of slice j this rank owns, in (t, g, G, P, M, V)."""
        key = ("own", id(t_i), j)
        v = self._views.get(key)
        if v is None:
            o0, c, off = self.own[j]
            blk = slice(o0, o0 + c)
            # (packed exchange: Adam's parameter rows and the all-gather's source are the fp32 master, not the table's own copy)
            p_rows = self.MP[off:off + c] if self.packed else self.P_i[blk]
            v = self._views[key] = (t_i[blk], self.GF_i[blk], self.G_i[blk], p_rows, self.MI[off:off + c],
                                    self.VI[off:off + c], t_i)
        return v

    def _ones_items(self):
        if self._all_items_bits is None:
            self._all_items_bits = self.k.ones_bits(self.Ip)
        return self._all_items_bits

    def train_step(self, gb):
        """gb: a GlobalBatch from make_batch() — the same global batch on every rank."""
        k = self.k
        Bc = gb.B
        if self.packed and not self._master_ready:
            self._wait_item_table()
            for o0, c, off in self.own:
                if c > 0:
                    self.MP[off:off + c][...] = self.P_i[o0:o0 + c]
            self._master_ready = True
        prep = None
        if self.batch_sparsity:
            prep = self._prepared.pop(gb.key, None)
            if prep is None:
                prep = k.prepare(self, gb)
            k.wait_rows(prep)
        self.propagate(prep, gb)
        # the batch's user rows (final and ego) travel through the guest rows: owners fill, everybody else adds zeros
        guests = self._views.get(("guest", Bc))
        if guests is None:
            guests = self._views[("guest", Bc)] = tuple(self._guest(a, Bc) for a in (self.FIN, self.P, self.GF, self.G)) \
                                                   + (self.guest_ids[:Bc],)
        fin_g, ego_g = guests[0], guests[1]
        k.gather_rows2(fin_g, self.FIN_u, ego_g, self.P_u, gb.own_src)
        self.comm.tag = "guest_rows"
        w1 = self.comm.all_reduce_async(fin_g)
        w2 = self.comm.all_reduce_async(ego_g)
        if prep is None:
            k.fill(self.G, 0.0)
            k.fill(self.GF, 0.0)
        self.comm.wait(w1)
        self.comm.wait(w2)
        k.bpr(self.FIN, self.P, self.Ug + self.B, guests[4], gb.pos, gb.neg, self.reg_lambda, self.GF, self.G,
              self.loss, prep)
        # gradients of the guest rows go home: every owned user's occurrences are added in batch order
        k.chain_rows2(self.GF_u, guests[2], self.G_u, guests[3], gb.head_dst, gb.nxt, store=prep is not None)
        self.step_count += 1
        self.propagate_backward(prep, gb, self.step_count)
        if prep is not None:
            k.release(prep)
        return self.loss

    def prefetch(self, gb):
        """One-batch lookahead of the index-only work of the NEXT step (row bitmaps + sorted scatter plan), on
        the kernels' side stream while this step's products run."""
        if self.batch_sparsity:
            while len(self._prepared) >= 2:  # lookaheads nobody came for (a skipped batch)
                self.k.release(self._prepared.pop(next(iter(self._prepared))))
            self._prepared[gb.key] = self.k.prepare(self, gb)

    def replicated_bytes_per_step(self):
        """Bytes of per-step work every rank carries whatever the world size (what caps the speed-up): the [I] flag
        vector of the touched-item agreement, the guest rows, the batch's item rows and the loss — O(I + B d), no
        O(I d) term.  (Round 2: 12 passes over the replicated [I, d] panel = 61 GB at configs[4].)"""
        flags = 2 * 4 * self.I if (self.world > 1 and self.K >= 2 and self.I * self.d * 4 >= self.live_rows_min_bytes) else 0
        return flags + 4 * self.d * (4 * self.B + 6 * 2 * self.B)

    # ---- evaluation: users by owner, items replicated, metric sums exchanged
    def evaluate(self, test_users, test_items, excl_indptr, excl_items, top_k, reduce_sums):
        """batch_test.Test (utility/utility_train/batch_test.py:37-93) for this rank's users.
        test_users: global ids of this rank's test users (ascending), test_items: their held-out item lists;
        excl_indptr / excl_items: train CSR of THIS rank's users (local ids); reduce_sums(np.float64 array) -> the
        element-wise sum over ranks.  Returns the reference's result dict, identical on every rank."""
        import utility.utility_function.metrics as metrics

        self.propagate(None)
        local = np.asarray(test_users, dtype=np.int64) - self.lo
        kmax = max(top_k)
        sums = np.zeros(3 * len(top_k) + 1, dtype=np.float64)
        if len(local):
            top = self.k.topk(self._u(self.FIN), self.item_rows(self.FIN), local, kmax, excl_indptr, excl_items)
            r = metrics.get_label(test_items, top)
            for j, kk in enumerate(top_k):
                sums[3 * j + 0] = metrics.recall_at_k(r, kk, test_items)
                sums[3 * j + 1] = metrics.precision_at_k(r, kk, test_items)
                sums[3 * j + 2] = metrics.ndcg_at_k(r, kk, test_items)
            sums[-1] = len(local)
        sums = reduce_sums(sums)
        n_users = float(sums[-1])
        return {"recall": sums[0:-1:3] / n_users, "precision": sums[1:-1:3] / n_users, "ndcg": sums[2:-1:3] / n_users,
                "hit": np.zeros(len(top_k))}


# --------------------------------------------------------------------------- per-collective timeline (instrumented steps)
XGMI_PEAK_GBS = 7 * 153.0  # what one MI355X can send to its seven peers at once (MI355X_MICROARCH.md: 7 links x ~153 GB/s)


class StepTimeline:
    """What a sub-6x multi-GPU line is diagnosed from: for a few INSTRUMENTED steps after the timed
    region, per collective tag (ShardedEngine sets comm.tag before every collective): how long the collective itself
    took on the stream it ran on, how long the step's stream stalled in wait() for it, its bytes and achieved bus
    bandwidth; the host synchronisation of the touched-item agreement (if any); the step's GPU time.  HIP events only:
    the instrumented steps are slower than the timed ones by the events' own cost (~5 us each) and are never timed as
    the headline."""

    def __init__(self, torch, world):
        self.torch, self.world = torch, int(world)
        self.records = []      # (tag, kind, bytes, ev_start, ev_end) — the collective on its stream
        self.stalls = []       # (tag, ev_before_wait, ev_after_wait) on the step's stream
        self.host_syncs = []   # seconds the host spent blocked in the touched-item read-back
        self.steps = []        # (ev_begin, ev_end) per instrumented step

    def event(self):
        return self.torch.cuda.Event(enable_timing=True)

    def host_sync_begin(self):
        import time

        return time.perf_counter()

    def host_sync_end(self, t0):
        import time

        self.host_syncs.append(time.perf_counter() - t0)

    def summary(self):
        """Per-step milliseconds (averages over the instrumented steps); call after torch.cuda.synchronize()."""
        n = max(len(self.steps), 1)
        per = {}
        for tag, kind, nbytes, e0, e1 in self.records:
            r = per.setdefault(tag, {"kind": kind, "calls": 0, "bytes": 0, "collective_ms": 0.0, "stall_ms": 0.0})
            r["calls"] += 1
            r["bytes"] += nbytes
            r["collective_ms"] += e0.elapsed_time(e1)
        for tag, e0, e1 in self.stalls:
            r = per.setdefault(tag, {"kind": "?", "calls": 0, "bytes": 0, "collective_ms": 0.0, "stall_ms": 0.0})
            r["stall_ms"] += e0.elapsed_time(e1)
        bus = 2.0 * (self.world - 1) / self.world if self.world > 1 else 0.0
        out = {}
        for tag, r in per.items():
            ms = r["collective_ms"] / n
            factor = bus if r["kind"] == "all_reduce" else bus / 2.0  # reduce-scatter / all-gather: (N-1)/N of the buffer
            gbs = (r["bytes"] / n * factor / (ms * 1e-3) / 1e9) if ms > 0 else None
            out[tag] = {"kind": r["kind"], "calls_per_step": r["calls"] / n, "bytes_per_step": r["bytes"] / n,
                        "collective_ms": ms, "main_stream_stall_ms": r["stall_ms"] / n, "bus_gbs": gbs,
                        # against what one GPU can send: 7 xGMI links x 153 GB/s (MI355X_MICROARCH.md); the 8-GPU
                        # projection of DESIGN.md §7 needs >= 0.75 of it on the three panel-sized exchanges
                        "bus_frac_of_xgmi_peak": (gbs / XGMI_PEAK_GBS) if gbs is not None else None}
        step_ms = sum(a.elapsed_time(b) for a, b in self.steps) / n
        stall = sum(v["main_stream_stall_ms"] for v in out.values())
        return {"instrumented_steps": len(self.steps), "step_gpu_ms": step_ms, "main_stream_stall_ms": stall,
                "xgmi_peak_gbs": XGMI_PEAK_GBS,
                "compute_ms": step_ms - stall,
                "host_sync_ms": (sum(self.host_syncs) / n * 1e3) if self.host_syncs else 0.0,
                "collectives": out,
                "what": "per training step, averaged over the instrumented steps (run after the timed region, HIP events "
                        "around every collective and every wait): collective_ms = the collective on the stream it ran on "
                        "(for collectives enqueued on the step's own stream that time is also a stall of the step); "
                        "main_stream_stall_ms = the step's stream blocked in wait(); compute_ms = step_gpu_ms - stalls; "
                        "bus_gbs = bytes x 2(N-1)/N (all-reduce) or (N-1)/N (reduce-scatter, all-gather) / collective_ms"}


class IssueOrder:
    """The ORDER in which one rank's host enqueues a step's products and collectives — DESIGN.md §7's overlap model as a
    checked property: the projection for 8 GPUs assumes that each slice's collective is issued before the
    next slice's product, that a panel exchange is only waited for after later products have been launched, and that the
    end-of-step all-gather of the updated item rows is still in flight under the NEXT step's first item-side products.
    None of this needs hardware: it is a property of the launch sequence, so the world-8 CPU tests and the
    8-ranks-on-one-GPU rehearsal assert it, and the first real 8-GPU run can then only disappoint on link rate.

        order = IssueOrder(); order.attach(engine)      # wraps engine.comm and engine.k
        order.begin_step(); engine.train_step(gb); ...
        assert not order.violations()

    events: ("step", i) | ("product", "user") | ("product", "item", j) | ("issue", tag, j) | ("wait", tag, j)."""

    SLICED = ("panel", "reduce_scatter")  # tags of collectives issued slice by slice from _item_side

    def __init__(self):
        self.events, self.n_steps, self.n_slices = [], 0, 1
        self.packed_violations = None

    def attach(self, eng):
        self.n_slices = len(eng.slices)
        # the 24-bit panel exchange keeps its own record of the order of its halves (Packed24Comm.order_violations):
        # slice j + 1's all-to-all queued before slice j's sum and all-gather
        self.packed_violations = getattr(eng.comm, "order_violations", None)
        names = {id(eng.G_ui): ("user",)}
        names.update({id(sl[0]): ("item", j) for j, sl in enumerate(eng.slices)})
        if isinstance(eng.comm, TimelineComm):
            eng.comm.order = self
        else:
            eng.comm = OrderComm(eng.comm, self)
        eng.k = OrderKernels(eng.k, self, names)
        return self

    def begin_step(self):
        self.events.append(("step", self.n_steps))
        self.n_steps += 1

    def end_steps(self):
        """What follows (the caller draining the last all-gather, evaluation) is not part of a step."""
        self.events.append(("end",))

    def steps(self):
        out, live = [], False
        for e in self.events:
            if e[0] == "step":
                out.append([])
                live = True
            elif e[0] == "end":
                live = False
            elif live:
                out[-1].append(e)
        return out

    def violations(self):
        """Every way the recorded sequence departs from the overlap model, as strings (empty list = as designed)."""
        bad, S = [], self.n_slices
        steps = self.steps()
        for si, ev in enumerate(steps):
            tags = []
            for i, e in enumerate(ev):
                if e[0] == "issue" and e[2] is not None and e[1].endswith(self.SLICED):
                    if e[1] not in tags:
                        tags.append(e[1])
                    # (1) a slice's collective goes out right behind that slice's product, before the next slice's
                    if i == 0 or ev[i - 1] != ("product", "item", e[2]):
                        bad.append("step %d: %s slice %d issued after %r, not right behind its own product" % (si, e[1], e[2], ev[i - 1] if i else None))
            for tag in tags:
                issued = [i for i, e in enumerate(ev) if e[0] == "issue" and e[1] == tag]
                waits = [i for i, e in enumerate(ev) if e[0] == "wait" and e[1] == tag]
                if len(issued) != S:
                    bad.append("step %d: %s issued for %d slices of %d" % (si, tag, len(issued), S))
                if not waits:
                    bad.append("step %d: %s never waited for inside the step" % (si, tag))
                    continue
                # (2) nobody waits for a sliced exchange before later products have been launched behind its last slice
                if not any(e[0] == "product" for e in ev[issued[-1] + 1: waits[0]]):
                    bad.append("step %d: %s waited for with no product launched behind it" % (si, tag))
            # (3) the updated item rows go round slice by slice: slice j's all-gather is out before slice j + 1's
            #     reduce-scatter is waited for
            ag = {e[2]: i for i, e in enumerate(ev) if e[0] == "issue" and e[1] == "item_table.all_gather"}
            rs = {e[2]: i for i, e in enumerate(ev) if e[0] == "wait" and e[1].endswith("reduce_scatter")}
            if len(ag) != S:
                bad.append("step %d: %d all-gathers of the updated item rows for %d slices" % (si, len(ag), S))
            for j in range(S - 1):
                if j in ag and j + 1 in rs and not ag[j] < rs[j + 1]:
                    bad.append("step %d: all-gather of slice %d issued after the wait for slice %d's reduce-scatter" % (si, j, j + 1))
            if any(e[0] == "wait" and e[1] == "item_table.all_gather" for e in ev[(ag[0] if 0 in ag else len(ev)):]):
                bad.append("step %d: the end-of-step all-gather is waited for inside its own step" % si)
            # (4) ... and is still in flight under the next step's first item-side products (which read local user rows only)
            if si > 0:
                w = [i for i, e in enumerate(ev) if e[0] == "wait" and e[1] == "item_table.all_gather"]
                first_user = next((i for i, e in enumerate(ev) if e == ("product", "user")), len(ev))
                n_before = sum(1 for e in ev[: (w[0] if w else 0)] if e[0] == "product" and e[1] == "item")
                if len(w) != S:
                    bad.append("step %d: %d waits for the previous step's %d all-gathers" % (si, len(w), S))
                elif n_before < S:
                    bad.append("step %d: the previous step's all-gather is waited for after %d of %d first-layer item-side "
                               "products" % (si, n_before, S))
                elif w[-1] > first_user:
                    bad.append("step %d: a user-side product (reads the item table) launched before the table was whole" % si)
        if self.packed_violations is not None:
            bad += ["24-bit exchange: " + v for v in self.packed_violations()]
        return bad

    def summary(self):
        return {"steps": self.n_steps, "slices": self.n_slices, "events": len(self.events), "violations": self.violations(),
                "what": "host issue order of the instrumented steps' products and collectives on this rank, checked against "
                        "DESIGN.md §7's overlap model (IssueOrder.violations): slice j's collective right behind slice j's "
                        "product; sliced exchanges waited for only after later products were launched; the end-of-step "
                        "all-gather waited for in the NEXT step, behind its first-layer item-side products"}


class OrderComm:
    """A comm wrapper that only records issue / wait order (IssueOrder); no events, no timing: CPU tests use it."""

    def __init__(self, inner, order):
        self.inner, self.order = inner, order
        self.world, self.rank = inner.world, inner.rank
        self.averages = getattr(inner, "averages", False)
        self.tag, self.slice = "untagged", None

    def all_reduce_async(self, t, average=False):
        self.order.events.append(("issue", self.tag, self.slice))
        return (self.inner.all_reduce_async(t, average), self.tag, self.slice)

    def all_gather_async(self, out, t):
        self.order.events.append(("issue", self.tag, self.slice))
        return (self.inner.all_gather_async(out, t), self.tag, self.slice)

    def reduce_scatter_async(self, t):
        self.order.events.append(("issue", self.tag, self.slice))
        return (self.inner.reduce_scatter_async(t), self.tag, self.slice)

    def wait(self, handle):
        if handle is None:
            return
        work, tag, sl = handle
        self.order.events.append(("wait", tag, sl))
        self.inner.wait(work)

    def __getattr__(self, name):
        return getattr(self.inner, name)


class OrderKernels:
    """A `kernels` wrapper that records every product launch (which operator: the user-side block or item slice j)."""

    def __init__(self, inner, order, names):
        self._inner, self._order, self._names = inner, order, names

    def spmm(self, graph, X, **kw):
        self._order.events.append(("product",) + self._names.get(id(graph), ("other",)))
        return self._inner.spmm(graph, X, **kw)

    def __getattr__(self, name):
        return getattr(self._inner, name)


class TimelineComm:
    """Wraps a comm (NativeComm / TorchComm / test comms) and fills a StepTimeline.  The collective's own duration is
    measured on the stream it runs on when the inner comm says which (NativeComm.timed_stream); otherwise between two
    events on the step's stream around the call (exact for collectives enqueued there, a lower bound — the issue cost —
    for a process group's internal stream, whose end is then taken at the matching wait())."""

    def __init__(self, inner, timeline):
        self.inner, self.tl = inner, timeline
        self.world, self.rank = inner.world, inner.rank
        self.averages = getattr(inner, "averages", False)
        self.tag, self.slice = "untagged", None
        self.order = None  # an IssueOrder: the same instrumented steps also record their issue order

    def _issue(self, kind, nbytes, call):
        torch, tl = self.tl.torch, self.tl
        if self.order is not None:
            self.order.events.append(("issue", self.tag, self.slice))
        e0, e1 = tl.event(), tl.event()
        own = getattr(self.inner, "timed_stream", None)
        stream = own(nbytes) if own is not None else None  # the torch stream the inner comm will run this one on
        if stream is None:
            e0.record()
            work = call()
            e1.record()
            rec = [self.tag, kind, nbytes, e0, e1]
            self.tl.records.append(rec)
            return (work, rec if work is not None else None, self.tag, self.slice)
        issued = torch.cuda.Event()
        issued.record()
        stream.wait_event(issued)   # (what the inner comm does next: the start event sits behind the same dependency)
        e0.record(stream)
        work = call()
        e1.record(stream)
        self.tl.records.append([self.tag, kind, nbytes, e0, e1])
        return (work, None, self.tag, self.slice)

    def _bytes(self, numel):
        # (a packed exchange puts 3 bytes per value on the wire: Packed24Comm.wire_bytes)
        f = getattr(self.inner, "wire_bytes", None)
        return f(numel) if f is not None else numel * 4

    def all_reduce_async(self, t, average=False):
        return self._issue("all_reduce", self._bytes(t.numel()), lambda: self.inner.all_reduce_async(t, average))

    def all_gather_async(self, out, t):
        return self._issue("all_gather", self._bytes(out.numel()), lambda: self.inner.all_gather_async(out, t))

    def reduce_scatter_async(self, t):
        return self._issue("reduce_scatter", self._bytes(t.numel()), lambda: self.inner.reduce_scatter_async(t))

    def wait(self, handle):
        if handle is None:
            return
        work, rec, tag, sl = handle
        if self.order is not None:
            self.order.events.append(("wait", tag, sl))
        a, b = self.tl.event(), self.tl.event()
        a.record()
        self.inner.wait(work)
        b.record()
        self.tl.stalls.append((tag, a, b))
        if rec is not None:
            rec[4] = b  # a work object's collective ends (at the latest) where its wait() returns on the step's stream

    def __getattr__(self, name):
        return getattr(self.inner, name)


# --------------------------------------------------------------------------- product bindings
class HipKernels:
    """`kernels` bound to libidgrec.so on the current HIP device.  The step issues ~40 library calls; what is marshalled
    here is kept minimal (the engine hands over the same preallocated buffers every step: pointers and views are cached,
    arguments are not re-validated — ops.py's wrappers do that for everybody else)."""

    def __init__(self, device=None):
        import ctypes as C

        import torch

        from . import native, ops

        self.torch, self.ops, self.C = torch, ops, C
        self.lib, self.check, self.Epilogue, self.ShardPrep = native.lib, native.check, native.Epilogue, native.ShardPrep
        self.device = torch.device("cuda", torch.cuda.current_device()) if device is None else torch.device(device)
        self._pool = []
        self._bits = {}
        self._ws = {}
        # side stream of the batch preparation, claimed at construction (ops.side_stream: hardware-queue placement)
        self._side = ops.side_stream(self.device)
        self._side_raw = self._side.cuda_stream
        self._fork = ops.LocalEvent()

    def zeros(self, shape):
        return self.torch.zeros(shape, dtype=self.torch.float32, device=self.device)

    def fill(self, a, v):
        a.fill_(v)

    def make_graph(self, indptr, indices, values, n_rows, n_cols):
        # both orientations (R_g and R_g^T) are built explicitly by shard_adjacency
        return self.ops.Graph(indptr, indices, values, n_rows, n_cols, device=self.device, symmetric=False,
                              build_transpose=False)

    def spmm(self, graph, X, Y=None, addend=None, sums=(), sum_out=None, div=1.0, accumulate=False, mask=None, adam=None,
             out_rows=None, x_rows=None, discard_grad=False, Y24=None):
        """idg_spmm_epi_f32 (include/idgrec.h): the product with the whole epilogue.  Y24: the finished rows as 24-bit values
        (a packed exchange's send buffer) instead of / beside Y."""
        d = X.shape[1]
        n_s = len(sums)
        e = self.Epilogue(None if Y is None else Y.data_ptr(), None if addend is None else addend.data_ptr(),
                          sums[0].data_ptr() if n_s > 0 else None, sums[1].data_ptr() if n_s > 1 else None,
                          sums[2].data_ptr() if n_s > 2 else None, None if sum_out is None else sum_out.data_ptr(), d, div,
                          1 if accumulate else 0, None if mask is None else mask.data_ptr())
        if adam is not None:
            p, m, v, lr, step = adam
            e.adam_param, e.adam_exp_avg, e.adam_exp_avg_sq = p.data_ptr(), m.data_ptr(), v.data_ptr()
            e.adam_lr, e.adam_beta1, e.adam_beta2, e.adam_eps, e.adam_step = lr, 0.9, 0.999, 1e-8, step
            e.adam_discard_grad = 1 if discard_grad else 0
        if Y24 is not None:
            e.y24 = Y24.data_ptr()
        key = (id(graph), d)
        ws = self._ws.get(key)
        if ws is None:
            ws = self._ws[key] = (graph, graph._workspace("spmm", d).data_ptr())
        rc = self.lib.idg_spmm_epi_f32(graph._h, X.data_ptr(), d, d, self.C.byref(e), None if out_rows is None else out_rows.data_ptr(),
                                       None if x_rows is None else x_rows.data_ptr(), ws[1], self.ops._stream())
        if rc:
            self.check(rc, "idg_spmm_epi_f32")

    # ---- 24-bit panels (Packed24Comm): values in, 3/4 as many 32-bit words out (buffers are fp32-typed: what the comms move)
    def pack24(self, src, dst, n):
        rc = self.lib.idg_pack24_f32(src.data_ptr(), dst.data_ptr(), n, self.ops._stream())
        if rc:
            self.check(rc, "idg_pack24_f32")

    def unpack24(self, src, dst, n):
        rc = self.lib.idg_unpack24_f32(src.data_ptr(), dst.data_ptr(), n, self.ops._stream())
        if rc:
            self.check(rc, "idg_unpack24_f32")

    def reduce24(self, blocks, n_blocks, n, out_packed=None, out_f32=None):
        rc = self.lib.idg_reduce24_f32(blocks.data_ptr(), n_blocks, n, None if out_packed is None else out_packed.data_ptr(),
                                       None if out_f32 is None else out_f32.data_ptr(), self.ops._stream())
        if rc:
            self.check(rc, "idg_reduce24_f32")

    def reduce_blocks(self, blocks, n_blocks, n, out):
        rc = self.lib.idg_reduce_blocks_f32(blocks.data_ptr(), n_blocks, n, out.data_ptr(), self.ops._stream())
        if rc:
            self.check(rc, "idg_reduce_blocks_f32")

    def exchange_stream(self):
        """Context manager: what is launched (and what collective is issued) inside goes to the exchange's own stream —
        NOT ordered behind the step's stream: the caller orders it by waiting for the collectives it consumes."""
        if getattr(self, "_xchg", None) is None:
            self._xchg = self.torch.cuda.Stream()
        return self.torch.cuda.stream(self._xchg)

    def record_event(self):
        """An event on the current stream (inside exchange_stream(): the exchange's), for wait_event()."""
        ev = self.torch.cuda.Event()
        ev.record()
        return ev

    def wait_event(self, ev):
        if ev is not None:
            self.torch.cuda.current_stream().wait_event(ev)

    def bits_from(self, bits, row0):
        """The bitmap of rows row0, row0 + 1, ... (row0 a multiple of 32)."""
        if row0 == 0:
            return bits
        key = (bits.data_ptr(), row0)
        v = self._bits.get(key)
        if v is None:
            v = self._bits[key] = (bits, bits[row0 // 32:])  # (holds the base: the key's address stays its)
        return v[1]

    def ones_bits(self, n_bits):
        return self.torch.full(((n_bits + 31) // 32 + 1,), -1, dtype=self.torch.int32, device=self.device)

    def bpr(self, fin, ego, n_users, users, pos, neg, reg_lambda, g_final, g_ego, loss, prep=None):
        if prep is not None:  # the sorted (row, slot) plan is in the prepared workspace; reached rows are STORED
            prep.done.wait(self.ops._stream())
            self.ops.bpr_fused_raw(fin, ego, users, pos, neg, n_users, reg_lambda, g_final, g_ego, loss=loss, deterministic=2,
                                   touched=prep.bpr_bits, ws=prep.ws)
        else:
            self.ops.bpr_fused_raw(fin, ego, users, pos, neg, n_users, reg_lambda, g_final, g_ego, loss=loss, deterministic=1)

    def to_device(self, a):
        return self.torch.from_numpy(np.ascontiguousarray(a)).to(self.device)

    def gather_rows(self, dst, src, idx):
        rc = self.lib.idg_rows_gather_f32(dst.data_ptr(), src.data_ptr(), idx.data_ptr(), idx.shape[0], dst.shape[1],
                                          self.ops._stream())
        if rc:
            self.check(rc, "idg_rows_gather_f32")

    def gather_rows2(self, dst0, src0, dst1, src1, idx):
        rc = self.lib.idg_rows_gather2_f32(dst0.data_ptr(), src0.data_ptr(), dst1.data_ptr(), src1.data_ptr(), idx.data_ptr(),
                                           idx.shape[0], dst0.shape[1], self.ops._stream())
        if rc:
            self.check(rc, "idg_rows_gather2_f32")

    def scatter_rows(self, dst, idx, src):
        """dst[idx[j]] = src[j] (idx distinct)."""
        self.ops.rows_scatter_raw(dst, idx, src)

    def chain_rows2(self, dst0, src0, dst1, src1, idx, nxt, store):
        if store:
            rc = self.lib.idg_rows_chain_store2_f32(dst0.data_ptr(), src0.data_ptr(), dst1.data_ptr(), src1.data_ptr(),
                                                    idx.data_ptr(), nxt.data_ptr(), idx.shape[0], dst0.shape[1],
                                                    self.ops._stream())
            if rc:
                self.check(rc, "idg_rows_chain_store2_f32")
        else:
            self.ops.rows_chain_add_raw(dst0, src0, idx, nxt)
            self.ops.rows_chain_add_raw(dst1, src1, idx, nxt)

    def layer_mean(self, out, ids, terms, last, div):
        self.ops.rows_layer_mean_raw(out, ids, terms, last, div)

    def item_tail(self, t, g, G, live_bits, row0, c0, cnt, store_grad, p, m, v, lr, step):
        rc = self.lib.idg_grad_tail_adam_f32(t.data_ptr(), g.data_ptr(), G.data_ptr(), live_bits.data_ptr(), row0, t.shape[0],
                                             t.shape[1], c0, cnt, 1 if store_grad else 0, p.data_ptr(), m.data_ptr(),
                                             v.data_ptr(), lr, 0.9, 0.999, 1e-8, step, self.ops._stream())
        if rc:
            self.check(rc, "idg_grad_tail_adam_f32")

    def adam(self, p, g, m, v, lr, step):
        self.ops.adam_step_raw(p, g, m, v, lr, step)

    # ---- the touched items (ShardedEngine._agree_touched_items) and the users near the batch's items
    def touched_local(self, eng, prep, gb):
        """One rank: the bitmap of the touched items from this rank's rows alone, and the near users."""
        # (a torch-side write of a bitmap: allowed here because nothing is ever registered for touched_buf at world size 1 —
        #  its unit lists are built by touched_from_ids, which only runs on several ranks — and mark_cols below is a library
        #  write of the same bitmap, which drops whatever was)
        prep.touched_buf.copy_(prep.items)
        eng.G_ui.mark_cols(prep.own_users, prep.touched_buf)
        prep.touched = prep.touched_buf
        self._near_users(eng, prep, gb)

    def flag_touched_items(self, eng, prep, gb, flags):
        """flags[i] = 1 for the items the batch's OWNED users interacted with, and for the batch's own items."""
        flags.zero_()
        eng.G_ui.flag_cols(prep.own_users, flags)
        flags.index_fill_(0, gb.items, 1.0)

    def nonzero_ids(self, flags):
        """Ascending ids of the non-zero flags and their number (a host synchronisation: the caller sizes a collective)."""
        ids = self.torch.nonzero(flags).reshape(-1)
        return ids, int(ids.numel())

    def compact_ids(self, flags, n):
        """Ascending ids of the non-zero flags in a list of n slots (the tail repeats the last id): built on the device,
        nothing read back (idg_flags_compact_f32)."""
        key = ("compact", flags.data_ptr())
        buf = self._ws.get(key)
        if buf is None or buf[0].shape[0] < n:
            ids, ws = self.ops.flags_compact_raw(flags, n)
            self._ws[key] = (ids, ws, flags)
            return ids
        ids = buf[0][:n]
        self.ops.flags_compact_raw(flags, n, ids=ids, ws=buf[1])
        return ids

    def touched_from_ids(self, eng, prep, ids, n):
        ids = ids[:n]
        self.ops.bpr_touch_rows_raw(ids, ids, ids, 0, prep.touched_buf, clear_bits=eng.Ip)
        prep.touched = prep.touched_buf
        # n is known here (and bounded by the compact buffer): the products restricted to the touched items — layer K - 1's
        # item side, the first backward product — run one wave per live work unit instead of visiting every tile
        st = self.ops._stream()
        for j, (g, r0, r1, r1p) in enumerate(eng.slices):
            ws = self._units(prep, ("t", j), g, eng.CS.shape[0])
            self.check(self.lib.idg_graph_live_units(g._h, self.bits_from(prep.touched_buf, r0).data_ptr(), ws.data_ptr(),
                                                     eng.CS.shape[0], st), "idg_graph_live_units")
        self._near_users(eng, prep, None)

    def _near_users(self, eng, prep, gb):
        """prep.near = the owned users that interacted with one of the batch's items, plus the owned batch users: where
        layer K - 1's user rows are read (by the last item-side product, restricted to the batch's items, and by FIN at
        the batch's users) and where the first backward product's user-side output lives."""
        prep.near_buf.copy_(prep.own_users)
        for g, r0, r1, r1p in eng.slices:
            g.mark_cols(self.bits_from(prep.items, r0), prep.near_buf)
        prep.near = prep.near_buf

    def _units(self, prep, key, graph, max_rows):
        ws = prep.units.get(key)
        if ws is None:
            nbytes = int(self.lib.idg_graph_live_units_bytes(graph._h, int(max_rows)))
            ws = prep.units[key] = self.torch.empty(nbytes // 4, dtype=self.torch.int32, device=self.device)
        return ws

    def topk(self, user_panel, item_panel, users, k, excl_indptr, excl_items):
        """Top-k item ids [len(users), k] (numpy) for local user ids `users`, train items excluded."""
        torch = self.torch
        idx = self.ops.score_topk(user_panel, item_panel, self.to_device(np.asarray(users, dtype=np.int64)), int(k),
                                  self.to_device(np.asarray(excl_indptr, dtype=np.int64)),
                                  self.to_device(np.asarray(excl_items, dtype=np.int32)), apply_sigmoid=True)
        torch.cuda.synchronize()
        return idx.cpu().numpy()

    class _Prepared:
        __slots__ = ("own_users", "items", "touched_buf", "near_buf", "touched", "near", "bpr_bits", "ws", "rows_done", "done",
                     "B", "busy", "units", "graphs", "args", "tables", "owner")

    def prepare(self, eng, gb):
        """Index-only work of a global batch on a side stream: the bitmap of the LOCAL user rows this rank owns in it,
        the bitmap of the batch's item rows (ALL triples' positives and negatives: every rank evaluates the whole batch),
        their lists of live work units (the last forward layer runs one wave per unit: idg_graph_live_units), a cleared
        bitmap for the gradient scatter to flag its stored rows in, and the sorted scatter plan of the whole batch over
        the guest rows.  Host cost matters here: raw stream handles and events allocated once."""
        torch, ops, lib = self.torch, self.ops, self.lib
        cap, n_users, n, d = eng.B, eng.Ug + eng.B, eng.Ug + eng.B + eng.Ip, eng.d
        # (pooled per ENGINE: the cached arguments hold that engine's graph handles, slice tables and bitmap sizes — ADVICE r03)
        prep = next((p for p in self._pool if p.owner is eng and p.B == cap and not p.busy), None)
        if prep is None:
            prep = self._Prepared()
            prep.owner = eng
            words = lambda bits: torch.zeros((bits + 31) // 32 + 1, dtype=torch.int32, device=self.device)  # noqa: E731
            prep.own_users, prep.near_buf = words(eng.Ug), words(eng.Ug)
            prep.items, prep.touched_buf = words(eng.Ip), words(eng.Ip)
            prep.bpr_bits = words(n)
            prep.ws, prep.B = ops.bpr_workspace(cap, d, self.device), cap
            prep.rows_done, prep.done = ops.LocalEvent(), ops.LocalEvent()  # device-local events
            prep.units = {}
            prep.graphs = [eng.G_ui] + [sl[0] for sl in eng.slices]  # (their unit lists name this object's bitmaps)
            C, S = self.C, len(eng.slices)
            a = prep.args = self.ShardPrep()
            a.guest_ids, a.B_cap = eng.guest_ids.data_ptr(), cap
            a.users_bits, a.n_local_users = prep.own_users.data_ptr(), eng.Ug
            a.items_bits, a.n_items_padded = prep.items.data_ptr(), eng.Ip
            a.scatter_bits, a.n_panel_rows = prep.bpr_bits.data_ptr(), n
            a.user_graph, a.user_units = eng.G_ui._h, self._units(prep, "u", eng.G_ui, cap).data_ptr()
            a.n_slices = S
            prep.tables = ((C.c_void_p * S)(*[sl[0]._h.value for sl in eng.slices]), (C.c_int64 * S)(*[sl[1] for sl in eng.slices]),
                           (C.c_void_p * S)(*[self._units(prep, ("i", j), sl[0], 2 * cap).data_ptr() for j, sl in enumerate(eng.slices)]))
            a.slice_graphs, a.slice_row0, a.slice_units = prep.tables
            a.plan_ws = prep.ws.data_ptr()
            a.side_stream = self._side_raw
            a.ev_fork, a.ev_rows, a.ev_plan = self._fork._h, prep.rows_done._h, prep.done._h
            self._pool.append(prep)
        prep.busy = True
        prep.touched = prep.near = None
        # one library call for the whole preparation (idg_shard_prepare); only the batch's own arrays change per call
        a = prep.args
        a.own_users, a.n_own = (gb.own_users.data_ptr() if gb.n_owned > 0 else None), gb.n_owned
        a.pos, a.neg, a.B = gb.pos.data_ptr(), gb.neg.data_ptr(), gb.B
        a.main_stream = ops._stream()
        rc = lib.idg_shard_prepare(self.C.byref(a))
        if rc:
            self.check(rc, "idg_shard_prepare")
        return prep

    def wait_rows(self, prep):
        """The bitmaps are first read by a product on the main stream."""
        prep.rows_done.wait(self.ops._stream())

    def release(self, prep):
        prep.busy = False  # its buffers go back to the pool; reuse is ordered by the fork event of the next prepare()

    def close(self):
        """The pooled bitmaps and unit lists die with this object: nothing may stay registered under their addresses.
        Called by whoever owns the engine when it is done (the bench does); __del__ is only the safety net."""
        for prep in self._pool:
            for ws in prep.units.values():
                self.lib.idg_graph_forget_units_ws(ws.data_ptr())
            for g in prep.graphs:
                g.forget_live_units()
        self._pool = []

    def __del__(self):
        try:
            self.close()
        except Exception:  # noqa: BLE001 - interpreter shutdown
            pass


class TorchComm:
    """`comm` on torch.distributed (backend "nccl" == RCCL over xGMI on ROCm).  With the gloo
    backend (tests: two ranks sharing one GPU, or CPU arrays) device tensors are staged through
    the host."""

    def __init__(self, dist):
        self.dist = dist
        self.backend = dist.get_backend()
        self.world, self.rank = dist.get_world_size(), dist.get_rank()
        # The sharded step issues 8 collectives; dist.all_reduce() spends ~25 us of host time per call in argument
        # checks before it reaches the process group.  Call the group object directly when this torch exposes it.
        self._pg = self._opts = self._opts_avg = None
        self.averages = self.backend == "nccl"  # RCCL divides inside the collective (ReduceOp.AVG); gloo cannot
        try:
            self._pg = dist.distributed_c10d._get_default_group()
            self._opts = dist.AllreduceOptions()
            self._opts.reduceOp = dist.ReduceOp.SUM
            self._opts_avg = dist.AllreduceOptions()
            self._opts_avg.reduceOp = dist.ReduceOp.AVG
        except Exception:  # noqa: BLE001 - private API: fall back to the public wrapper
            self._pg = None

    def all_reduce_async(self, t, average=False):
        """Sum over ranks; average=True asks for the mean and gets it only when self.averages (otherwise the sum —
        the caller scales)."""
        import torch

        if isinstance(t, np.ndarray):
            t = torch.from_numpy(t)  # shares memory
        if self.backend == "gloo" and t.is_cuda:
            host = t.cpu()
            self.dist.all_reduce(host)
            t.copy_(host)
            return None
        avg = average and self.averages
        if self._pg is not None and t.is_cuda:
            return self._pg.allreduce([t], self._opts_avg if avg else self._opts)
        return self.dist.all_reduce(t, op=self.dist.ReduceOp.AVG if avg else self.dist.ReduceOp.SUM, async_op=True)

    def all_gather_async(self, out, t):
        """out (world x len(t) elements, rank-major) <- every rank's t; t may be this rank's block of out (in place)."""
        import torch

        if isinstance(t, np.ndarray):
            t, out = torch.from_numpy(t), torch.from_numpy(out)  # share memory
        if self.backend == "gloo":
            # (the rehearsal backend: staged through a host copy, which also makes the in-place form safe)
            host = torch.empty(out.shape, dtype=out.dtype)
            self.dist.all_gather_into_tensor(host, t.cpu().contiguous())
            out.copy_(host)
            return None
        return self.dist.all_gather_into_tensor(out, t, async_op=True)

    def reduce_scatter_async(self, t):
        """t = world equal blocks; this rank's block <- the sum over ranks of that block, in place (the other blocks are
        left in an unspecified state).  gloo has no reduce-scatter: the rehearsal backend all-reduces the whole buffer."""
        import torch

        if isinstance(t, np.ndarray):
            t = torch.from_numpy(t)
        if self.backend == "gloo":
            return self.all_reduce_async(t)
        flat = t.view(-1)
        c = flat.numel() // self.world
        own = flat[self.rank * c:(self.rank + 1) * c]
        out = torch.empty_like(own)  # (c10d does not promise the in-place form: reduce into a scratch block, then copy)
        work = self.dist.reduce_scatter_tensor(out, flat, async_op=True)
        return ("copy_after", work, own, out)

    def all_to_all_async(self, recv, send):
        """Block p of `send` (world equal blocks) goes to rank p; block p of `recv` receives rank p's block for this rank."""
        import torch

        if isinstance(send, np.ndarray):
            send, recv = torch.from_numpy(send), torch.from_numpy(recv)
        if self.backend == "gloo":
            # (the rehearsal backend has no all-to-all: point-to-point pairs on host copies, completed inside the call)
            s, r = send.detach().cpu().view(self.world, -1), torch.empty(recv.shape, dtype=recv.dtype).view(self.world, -1)
            r[self.rank] = s[self.rank]
            reqs = []
            for p in range(self.world):
                if p != self.rank:
                    reqs.append(self.dist.isend(s[p].contiguous(), p))
                    reqs.append(self.dist.irecv(r[p], p))
            for q in reqs:
                q.wait()
            recv.view(self.world, -1).copy_(r)
            return None
        return self.dist.all_to_all_single(recv.view(-1), send.view(-1), async_op=True)

    def wait(self, work):
        if isinstance(work, tuple):  # reduce_scatter_async: the reduced block goes to its place once the collective is done
            _, inner, own, out = work
            inner.wait()
            own.copy_(out)
        elif work is not None:
            work.wait()


class NativeComm:
    """`comm` on libidgrec's own RCCL communicator (idg_comm_*, include/idgrec.h): collectives are enqueued on the
    CURRENT HIP stream, in order with the kernels around them — no second stream, no event pair and no work object
    per call (torch.distributed's process group costs ~20 us of host time and two cross-stream waits per
    collective, which is most of a step on the small graphs).  torch.distributed is used once, to hand rank 0's
    unique id to the other ranks."""

    averages = True
    _generation = 0

    def __init__(self, dist, device_index, overlap_bytes=64 << 20):
        """overlap_bytes: collectives of at least this many bytes run on a stream of their own, ordered after the
        current stream by an event, so that the caller's next kernels overlap them until wait(); smaller ones stay on
        the current stream (the two cross-stream waits cost ~40 us of host time per collective — measured on the
        9.7 MB item panel of the yelp2018 shape: 0.69 ms per sharded step with them, 0.53 without)."""
        import ctypes as C

        import torch

        from . import native

        self.torch, self.lib, self.check = torch, native.lib, native.check
        self.world = dist.get_world_size()
        path = os.path.join(os.path.dirname(torch.__file__), "lib", "librccl.so")  # the copy this process already runs
        self.check(self.lib.idg_comm_load(path.encode() if os.path.exists(path) else None), "idg_comm_load")
        self.rank, self.world = dist.get_rank(), dist.get_world_size()
        uid = torch.zeros(128, dtype=torch.uint8)
        NativeComm._generation += 1
        key = "idg_comm_unique_id_%d" % NativeComm._generation  # every rank constructs communicators in the same order
        failure = None
        if self.rank == 0:
            try:
                self.check(self.lib.idg_comm_unique_id(uid.data_ptr()), "idg_comm_unique_id")
            except Exception as exc:  # noqa: BLE001 - the other ranks must learn of it instead of waiting for an id
                failure = exc
        if self.world > 1:
            # through the rendezvous store, not a collective (torch's own RCCL communicator and its streams are then
            # created after this one's; see ops.side_stream for what stream order does to hardware-queue placement)
            store = dist.distributed_c10d._get_default_store()
            if self.rank == 0:
                store.set(key, b"FAILED" if failure is not None else bytes(uid.numpy().tobytes()))
            got = bytes(store.get(key))
            if got == b"FAILED" and failure is None:
                failure = RuntimeError("rank 0 could not obtain an RCCL unique id")
            if failure is None:
                uid = torch.frombuffer(bytearray(got), dtype=torch.uint8).clone()
        if self.world > 1:
            # idg_comm_create is collective (ncclCommInitRank): a rank that cannot take part — no id, or a device
            # index this process cannot open — must say so BEFORE the others enter it, or they wait there for good.
            # Every rank publishes a verdict under this communicator's generation and reads everyone else's.
            ready = failure is None and 0 <= int(device_index) < torch.cuda.device_count()
            store.set("idg_comm_ready_%d_%d" % (NativeComm._generation, self.rank), b"1" if ready else b"0")
            bad = [r for r in range(self.world)
                   if bytes(store.get("idg_comm_ready_%d_%d" % (NativeComm._generation, r))) != b"1"]
            if bad and failure is None:
                failure = RuntimeError("libidgrec communicator: rank(s) %s cannot join (no unique id or no such device)" % bad)
        if failure is not None:
            raise failure
        handle = C.c_void_p()
        self.check(self.lib.idg_comm_create(self.rank, self.world, uid.data_ptr(), int(device_index), C.byref(handle)),
                   "idg_comm_create")
        self.handle = handle
        self._force = False  # self_test(): enqueue on RCCL even at world size 1, where a collective is the identity
        self.overlap_bytes = int(overlap_bytes)
        self._own = self._own_raw = None   # the collectives' own stream, made on first use
        self._ring, self._next = [], 0     # (issued, done) event pairs, reused round-robin (a step issues up to 4 x slices collectives here; a handle is waited for within the next step)

    def _fork(self):
        """Order the communicator's stream after the current one; returns (raw stream handle, event to record when done)."""
        torch = self.torch
        if self._own is None:
            self._own = torch.cuda.Stream()
            self._own_raw = self._own.cuda_stream
            self._ring = [(torch.cuda.Event(), torch.cuda.Event()) for _ in range(128)]  # > 3 steps' worth of collectives
        issued, done = self._ring[self._next]
        self._next = (self._next + 1) % len(self._ring)
        issued.record()
        self._own.wait_event(issued)
        return self._own_raw, done

    @staticmethod
    def _stream():
        from .ops import _stream

        return _stream()

    def _f32(self, t):
        assert t.is_cuda and t.dtype == self.torch.float32 and t.is_contiguous(), "NativeComm moves contiguous fp32 device tensors"
        return t.data_ptr()

    def timed_stream(self, nbytes):
        """The torch stream a collective of nbytes will run on when it is NOT the current one (TimelineComm), else None."""
        if (self.world == 1 and not self._force) or nbytes < self.overlap_bytes:
            return None
        if self._own is None:
            self._fork()  # creates the stream (the ring slot it takes is harmless)
        return self._own

    def all_reduce_async(self, t, average=False):
        if self.world == 1 and not self._force:  # (the identity: nothing to enqueue; self_test() still runs RCCL itself)
            return None
        if t.numel() * 4 >= self.overlap_bytes:
            stream, done = self._fork()
            self.check(self.lib.idg_allreduce_f32(self.handle, self._f32(t), t.numel(), int(bool(average)), stream),
                       "idg_allreduce_f32")
            done.record(self._own)
            return done
        self.check(self.lib.idg_allreduce_f32(self.handle, self._f32(t), t.numel(), int(bool(average)), self._stream()),
                   "idg_allreduce_f32")
        return None

    def all_gather_async(self, out, t):
        assert out.numel() == t.numel() * self.world
        if self.world == 1 and not self._force:
            if out.data_ptr() != t.data_ptr():
                out.view(-1).copy_(t.view(-1))
            return None
        if out.numel() * 4 >= self.overlap_bytes:
            stream, done = self._fork()
            self.check(self.lib.idg_allgather_f32(self.handle, self._f32(t), self._f32(out), t.numel(), stream),
                       "idg_allgather_f32")
            done.record(self._own)
            return done
        self.check(self.lib.idg_allgather_f32(self.handle, self._f32(t), self._f32(out), t.numel(), self._stream()),
                   "idg_allgather_f32")
        return None

    def reduce_scatter_async(self, t):
        """t = world equal blocks; this rank's block <- the sum over ranks of that block, in place."""
        c = t.numel() // self.world
        assert c * self.world == t.numel(), "reduce-scatter of %d floats over %d ranks" % (t.numel(), self.world)
        if self.world == 1 and not self._force:
            return None
        base = self._f32(t)
        own = base + 4 * c * self.rank
        if t.numel() * 4 >= self.overlap_bytes:
            stream, done = self._fork()
            self.check(self.lib.idg_reduce_scatter_f32(self.handle, base, own, c, stream), "idg_reduce_scatter_f32")
            done.record(self._own)
            return done
        self.check(self.lib.idg_reduce_scatter_f32(self.handle, base, own, c, self._stream()), "idg_reduce_scatter_f32")
        return None

    def all_to_all_async(self, recv, send):
        """Block p of `send` goes to rank p; block p of `recv` receives rank p's block for this rank (idg_alltoall_f32:
        grouped ncclSend / ncclRecv)."""
        c = send.numel() // self.world
        assert c * self.world == send.numel() == recv.numel(), "all-to-all of %d floats over %d ranks" % (send.numel(), self.world)
        flags = 1 if self._force else 0  # (tests at world size 1: the own block through ncclSend / ncclRecv too)
        if send.numel() * 4 >= self.overlap_bytes and not (self.world == 1 and not self._force):
            stream, done = self._fork()
            self.check(self.lib.idg_alltoall_f32(self.handle, self._f32(send), self._f32(recv), c, flags, stream), "idg_alltoall_f32")
            done.record(self._own)
            return done
        self.check(self.lib.idg_alltoall_f32(self.handle, self._f32(send), self._f32(recv), c, flags, self._stream()),
                   "idg_alltoall_f32")
        return None

    def wait(self, work):
        if work is not None:
            self.torch.cuda.current_stream().wait_event(work)

    def through_rccl(self):
        """Context manager: at world size 1 a collective is the identity and is normally not enqueued at all; inside this
        context it goes through RCCL as on a larger world (tests, the self-test)."""
        import contextlib

        @contextlib.contextmanager
        def forced():
            old, self._force = self._force, True
            try:
                yield self
            finally:
                self._force = old

        return forced()

    def self_test(self):
        """All-reduce (both stream routes), all-gather and reduce-scatter + in-place all-gather with known answers; True when
        all are right on this rank.  Runs through RCCL at any world size."""
        with self.through_rccl():
            return self._self_test()

    def _self_test(self):
        torch = self.torch
        a = torch.full((1024,), float(self.rank + 1), dtype=torch.float32, device="cuda")
        g = torch.zeros(1024 * self.world, dtype=torch.float32, device="cuda")
        self.all_gather_async(g, a)
        self.all_reduce_async(a)
        big = torch.full((self.overlap_bytes // 4 + 1024,), float(self.rank + 1), dtype=torch.float32, device="cuda")
        work = self.all_reduce_async(big, average=True)   # the second-stream form
        self.wait(work)
        big += 1.0                                        # ordered after the collective by wait()
        # reduce-scatter + in-place all-gather of the reduced blocks == all-reduce (what ends a sharded step)
        rs = (torch.arange(1024 * self.world, dtype=torch.float32, device="cuda") % 7) * float(self.rank + 1)
        rs_want = (torch.arange(1024 * self.world, dtype=torch.float32, device="cuda") % 7) * (self.world * (self.world + 1) / 2)
        self.wait(self.reduce_scatter_async(rs))
        self.wait(self.all_gather_async(rs, rs[1024 * self.rank: 1024 * (self.rank + 1)]))
        torch.cuda.synchronize()
        want = torch.arange(1, self.world + 1, dtype=torch.float32, device="cuda").repeat_interleave(1024)
        return (bool((a == self.world * (self.world + 1) / 2).all().item()) and bool(torch.equal(g, want))
                and bool((big == (self.world + 1) / 2 + 1.0).all().item()) and bool(torch.equal(rs, rs_want)))

    def close(self):
        if self.handle is not None:
            self.torch.cuda.synchronize()
            self.lib.idg_comm_destroy(self.handle)
            self.handle = None


def make_comm(dist, kind="auto"):
    """kind: "torch" (torch.distributed process group), "native" (libidgrec's RCCL communicator; fails loudly if it
    cannot be set up) or "auto": native when the backend is nccl and EVERY rank both set it up and passed its
    self-test, torch.distributed otherwise (both are RCCL over xGMI; the choice is recorded in the bench line)."""
    import torch

    if kind == "torch" or (kind == "auto" and dist.get_backend() != "nccl"):
        return TorchComm(dist), "torch.distributed"
    if kind == "native":
        comm = NativeComm(dist, torch.cuda.current_device())
        assert comm.self_test(), "libidgrec RCCL communicator: self-test failed"
        return comm, "libidgrec RCCL communicator"
    # auto: agree rank by rank before the collective idg_comm_create (a rank that cannot load the library must not
    # leave the others waiting inside ncclCommInitRank)
    from . import native

    path = os.path.join(os.path.dirname(torch.__file__), "lib", "librccl.so")
    loaded = native.lib.idg_comm_load(path.encode() if os.path.exists(path) else None) == 0
    rank, world = dist.get_rank(), dist.get_world_size()
    try:
        store = dist.distributed_c10d._get_default_store()  # no collective yet: see NativeComm.__init__
    except Exception:  # noqa: BLE001 - private torch API: without it there is no collective-free way to agree
        return TorchComm(dist), "torch.distributed (rendezvous store not reachable)"
    # keys carry the generation of the communicator about to be built (every rank calls make_comm in the same order),
    # so a second make_comm on the same process group never reads the first one's verdicts
    gen = NativeComm._generation + 1
    store.set("idg_comm_loaded_%d_%d" % (gen, rank), b"1" if loaded else b"0")
    if not all(bytes(store.get("idg_comm_loaded_%d_%d" % (gen, r))) == b"1" for r in range(world)):
        return TorchComm(dist), "torch.distributed (libidgrec could not load librccl)"
    why = ""
    try:
        comm = NativeComm(dist, torch.cuda.current_device())
        good = comm.self_test()
    except Exception as exc:  # noqa: BLE001 - any failure selects the other RCCL path, and is reported
        comm, good = None, False
        why = str(exc)[:120]
    ok = torch.tensor([1 if good else 0], dtype=torch.int32, device="cuda")
    dist.all_reduce(ok, op=dist.ReduceOp.MIN)
    if int(ok.item()) == 1:
        return comm, "libidgrec RCCL communicator"
    return TorchComm(dist), "torch.distributed (libidgrec communicator unavailable%s)" % ((": " + why) if comm is None else "")


class NoComm:
    """world_size 1."""

    averages = True
    world = 1
    rank = 0

    def all_reduce_async(self, t, average=False):
        return None

    def all_gather_async(self, out, t):
        same = (out.data_ptr() == t.data_ptr()) if hasattr(out, "data_ptr") else np.shares_memory(out, t)
        if not same:  # (in place at world size 1: nothing moves)
            out[...] = t
        return None

    def reduce_scatter_async(self, t):
        return None

    def all_to_all_async(self, recv, send):
        recv[...] = send
        return None

    def wait(self, work):
        pass


# --------------------------------------------------------------------------- the panel exchanges as 24-bit rows (opt-in)
class Packed24Comm:
    """A comm wrapper (around NativeComm / TorchComm / NoComm / the tests' comms): the step's LARGE exchanges — the two
    [I, d] all-reduces, the reduce-scatter of the last backward product, the all-gather of the updated item rows, the
    touched-item row sets — travel as 24-bit values (idg_pack24_f32: sign, exponent, 15 mantissa bits, rounded to nearest;
    2^-16 = 1.5e-5 relative, inside the north star's 1e-4) and are summed by this library's own kernel IN RANK ORDER:

        all-reduce      pack -> all-to-all (block p to rank p) -> reduce24 (q0 + q1 + ... in rank order, packed again)
                        -> all-gather of the packed sums -> unpack (EVERY rank, the owner included, holds the same bits)
        reduce-scatter  pack -> all-to-all -> reduce24 into the own fp32 block
        all-gather      pack the own block -> all-gather -> unpack (the own block too: replicas stay bit-identical)

    3/4 of the fp32 exchange's bytes on the links (what lowers the link rate >= 6x at 8 GPUs needs from 88 %
    to 66 % of the xGMI peak), and a k-GPU run is bit-reproducible run to run whatever algorithm RCCL would pick
    (SURVEY.md 8e).  Collectives below `min_bytes` (the batch's guest rows, whose exchange must stay exact: x + 0 + ...)
    or whose length does not divide by 4 x world stay with the inner comm, in fp32.

    Streams (kernels.exchange_stream): pack runs on the step's stream right behind the product; the all-to-all is issued
    from there; the second half — wait for the all-to-all, reduce24, all-gather, unpack — runs on the exchange's own
    stream, and is issued only AFTER the next slice's first half (slice j + 1's all-to-all is queued before slice j's
    all-gather, so the links are not idle while slice j is being summed).  wait() joins the step's stream.
    `log` records the order of the halves (IssueOrder checks it); `stats()` the bytes put on the wire."""

    averages = False
    by_blocks = True   # (the engine hands over slices that divide into one block per rank: the padded views)

    def __init__(self, inner, kernels, min_bytes=4 << 20, bits=24):
        """bits = 32: the same explicit exchange on fp32 values — no packing, RCCL's bytes on the links, the rank-ordered sum
        (idg_reduce_blocks_f32): reproducible without the 2^-16 quantisation (`bench.py --reduce-order rank`)."""
        assert bits in (24, 32)
        self.inner, self.k = inner, kernels
        self.world, self.rank = int(inner.world), int(inner.rank)
        self.bits, self.packed = int(bits), bits == 24
        self.min_bytes = int(min_bytes)
        self._free = {}        # words -> [(snd, rcv), ...] buffers not in flight
        self._pending = []     # exchanges whose second half has not been issued yet, oldest first
        self._seq = 0
        self.log = []          # ("first" | "second" | "wait", seq, kind)
        self._pre = {}         # (address, length) of a tensor -> (snd, rcv) its producer is writing packed values into
        self.wire = {"packed_bytes_sent": 0, "fp32_bytes_it_replaces": 0, "exchanges": 0, "fp32_collectives": 0,
                     "packed_by_producer": 0}

    # ---- helpers
    @staticmethod
    def _n(t):
        return int(t.size) if isinstance(t, np.ndarray) else int(t.numel())

    @staticmethod
    def _flat(t):
        return t.reshape(-1)

    def _eligible(self, n):
        return n * 4 >= self.min_bytes and n % (4 * self.world) == 0

    def _buffers(self, words):
        """(send, receive) buffers of `words` 32-bit words: views of buffers kept per SIZE CLASS (the next power of two —
        the touched-item exchanges come in a different length for nearly every batch: one pool entry per length would
        grow without bound over a training run)."""
        cls = 1 << max(int(words) - 1, 0).bit_length()
        pool = self._free.setdefault(cls, [])
        # (fp32 blocks: the tensor itself is the send buffer — only the receive side needs one)
        snd, rcv = pool.pop() if pool else (self.k.zeros((cls if self.packed else 1,)), self.k.zeros((cls,)))
        return snd[:words], rcv[:words], (cls, snd, rcv)

    def _words(self, n):
        return n // 4 * 3 if self.packed else n

    def _count(self, n_values, phases):
        # a rank sends (N - 1) / N of the buffer in each phase (all-to-all, all-gather)
        share = (self.world - 1) / self.world
        self.wire["packed_bytes_sent"] += int(phases * share * n_values * (3 if self.packed else 4))
        self.wire["fp32_bytes_it_replaces"] += int(phases * share * n_values * 4)
        self.wire["exchanges"] += 1

    class _X:
        __slots__ = ("seq", "kind", "t", "n", "snd", "rcv", "home", "work", "done", "second", "own")

    def _first_half(self, kind, t, n):
        """pack + all-to-all of t's n values (kind "ar" / "rs"), on the current (the step's) stream."""
        words = self._words(n)
        x = self._X()
        x.seq, x.kind, x.t, x.n, x.second, x.done, x.own = self._seq, kind, t, n, False, None, None
        self._seq += 1
        pre = self._pre.pop(self._key(t), None)
        if pre is not None:   # the producer has written the packed values itself (packed_target)
            x.snd, x.rcv, x.home = pre
            self.wire["packed_by_producer"] += 1
        elif self.packed:
            x.snd, x.rcv, x.home = self._buffers(words)
            self.k.pack24(self._flat(t), x.snd, n)
        else:                 # fp32 blocks: the tensor itself is the send buffer
            _, x.rcv, x.home = self._buffers(words)
            x.snd = self._flat(t)
        x.work = self.inner.all_to_all_async(x.rcv, x.snd)
        self.log.append(("first", x.seq, kind))
        # the exchange before this one may now go on: its sum and its all-gather queue BEHIND this all-to-all
        self._flush(keep=x)
        self._pending.append(x)
        return x

    def _second_half(self, x):
        k, N = self.k, self.world
        blk_words, blk_n = self._words(x.n) // N, x.n // N
        with k.exchange_stream():
            self.inner.wait(x.work)
            if not self.packed:
                own = x.snd[self.rank * blk_n:(self.rank + 1) * blk_n]   # (x.snd is the tensor itself)
                k.reduce_blocks(x.rcv, N, blk_n, own)
                if x.kind == "ar":
                    self.inner.wait(self.inner.all_gather_async(x.snd, own))
            elif x.kind == "rs":
                own = self._flat(x.t)[self.rank * blk_n:(self.rank + 1) * blk_n]
                k.reduce24(x.rcv, N, blk_n, out_f32=own)
            else:
                own = x.snd[self.rank * blk_words:(self.rank + 1) * blk_words]
                k.reduce24(x.rcv, N, blk_n, out_packed=own)
                self.inner.wait(self.inner.all_gather_async(x.snd, own))
                k.unpack24(x.snd, self._flat(x.t), x.n)
            x.done = k.record_event()
        x.second = True
        self.log.append(("second", x.seq, x.kind))

    def _flush(self, keep=None):
        while self._pending and self._pending[0] is not keep:
            self._second_half(self._pending.pop(0))

    @staticmethod
    def _key(t):
        return (t.ctypes.data, t.size) if isinstance(t, np.ndarray) else (t.data_ptr(), t.numel())

    def packed_target(self, t, n_valid=None):
        """The send buffer of the NEXT all_reduce_async / reduce_scatter_async of `t`, for a producer that writes its result
        packed by itself (the item-side product's y24 epilogue: no fp32 write of the partial, no pack pass) — or None when
        that collective would not travel packed.  n_valid: the values the producer will write (the rest of t is zero:
        a slice's padding rows)."""
        n = self._n(t)
        if not self.packed or not self._eligible(n):
            return None
        snd, rcv, home = self._buffers(n // 4 * 3)
        if n_valid is not None and n_valid < n:
            snd[n_valid // 4 * 3:] = 0
        old = self._pre.pop(self._key(t), None)
        if old is not None:  # (a target nobody came for: its buffers go back)
            self._free[old[2][0]].append(old[2][1:])
        self._pre[self._key(t)] = (snd, rcv, home)
        return snd

    # ---- the comm interface
    def all_reduce_async(self, t, average=False):
        n = self._n(t)
        if not self._eligible(n):
            self.wire["fp32_collectives"] += 1
            return ("inner", self.inner.all_reduce_async(t, average))
        assert not average, "Packed24Comm sums"
        self._count(n, 2)
        return ("packed", self._first_half("ar", t, n))

    def reduce_scatter_async(self, t):
        n = self._n(t)
        if not self._eligible(n):
            self.wire["fp32_collectives"] += 1
            return ("inner", self.inner.reduce_scatter_async(t))
        self._count(n, 1)
        return ("packed", self._first_half("rs", t, n))

    def all_gather_async(self, out, t):
        n, nb = self._n(out), self._n(t)
        if not self.packed or not self._eligible(n) or nb * self.world != n:
            self.wire["fp32_collectives"] += 1
            if self._pending:
                self._flush()  # (collectives go out in program order on every rank)
            return ("inner", self.inner.all_gather_async(out, t))
        self._count(n, 1)
        self._flush()  # (an all-gather has no first half to hide a sum behind: whatever is pending goes first)
        words, blk_words = n // 4 * 3, nb // 4 * 3
        x = self._X()
        x.seq, x.kind, x.t, x.n, x.second = self._seq, "ag", out, n, True
        self._seq += 1
        x.snd, x.rcv, x.home = self._buffers(words)
        own = x.snd[self.rank * blk_words:(self.rank + 1) * blk_words]
        self.k.pack24(self._flat(t), own, nb)
        x.work = self.inner.all_gather_async(x.snd, own)
        self.log.append(("first", x.seq, "ag"))
        with self.k.exchange_stream():
            self.inner.wait(x.work)
            self.k.unpack24(x.snd, self._flat(out), n)
            x.done = self.k.record_event()
        self.log.append(("second", x.seq, "ag"))
        return ("packed", x)

    def all_to_all_async(self, recv, send):
        return ("inner", self.inner.all_to_all_async(recv, send))

    def wire_bytes(self, numel):
        """Bytes a collective over `numel` values moves per unit of buffer (TimelineComm's bus-rate arithmetic)."""
        return numel * (3 if (self.packed and self._eligible(numel)) else 4)

    def timed_stream(self, nbytes):
        return None  # (a compound exchange: TimelineComm brackets it on the step's stream, issue to wait)

    def wait(self, handle):
        if handle is None:
            return
        kind, x = handle
        if kind == "inner":
            self.inner.wait(x)
            return
        if not x.second:  # nobody issued a later exchange behind it: its second half goes out now (and all before it)
            while self._pending:
                y = self._pending.pop(0)
                self._second_half(y)
                if y is x:
                    break
        self.k.wait_event(x.done)
        self.log.append(("wait", x.seq, x.kind))
        self._free[x.home[0]].append(x.home[1:])
        x.snd = x.rcv = x.t = x.home = None

    def __getattr__(self, name):
        return getattr(self.inner, name)

    # ---- evidence
    def order_violations(self):
        """The halves' issue order against the pipeline this class promises: every first half is followed by its second
        half exactly once and waited for after it; the second half of exchange i is never issued before the first half
        of exchange i + 1 when that one was issued before i was waited for (its all-to-all is in the queue first)."""
        bad, pos = [], {}
        for i, (what, seq, kind) in enumerate(self.log):
            pos.setdefault(seq, {})[what] = i
        for seq, p in sorted(pos.items()):
            if "first" in p and "second" not in p and "wait" in p:
                bad.append("exchange %d waited for without its second half" % seq)
            if "second" in p and "first" in p and not p["first"] < p["second"]:
                bad.append("exchange %d: second half before the first" % seq)
            if "wait" in p and "second" in p and not p["second"] < p["wait"]:
                bad.append("exchange %d: waited for before its second half was issued" % seq)
            nxt = pos.get(seq + 1)
            # (an all-gather has no all-to-all to queue ahead: it flushes what is pending and is not part of this rule)
            if nxt and "first" in nxt and "wait" in p and nxt["first"] < p["wait"] and p.get("second", 0) < nxt["first"] \
                    and self.log[p["first"]][2] != "ag" and self.log[nxt["first"]][2] != "ag":
                bad.append("exchange %d: its sum / all-gather was issued before exchange %d's all-to-all (no overlap)" % (seq, seq + 1))
        return bad

    def release_buffers(self):
        """Drop the pooled send / receive buffers (nothing may be in flight): the bench frees them before rank 0 builds the
        single-GPU reference."""
        assert not self._pending, "an exchange is still in flight"
        self._free.clear()
        self._pre.clear()

    def stats(self):
        w = dict(self.wire)
        w["ratio"] = (w["packed_bytes_sent"] / w["fp32_bytes_it_replaces"]) if w["fp32_bytes_it_replaces"] else None
        w["order_violations"] = self.order_violations()
        return w


RankOrderComm = Packed24Comm  # (bits = 32: the explicit exchange and the rank-ordered sum on fp32 values)


# --------------------------------------------------------------------------- bench driver
PARITY_TOL = 1e-4  # BASELINE.json north_star: "within 1e-4 relative on fp32 embeddings and loss"


def parity_capture(eng, batches, tri, dist, rank, lo, hi, num_users, num_items, seed=0, n_sample=1024):
    """The N-rank half of `parity_vs_1gpu` (the only place RCCL between devices ever runs — the driver's
    `bench.py --gpus N` — must carry its own correctness evidence).  Runs the steps `batches` (GlobalBatch list; `tri` their
    global triples [n * B, 3]) on the freshly initialised sharded engine — every rank calls this — and returns on rank 0
    what the single-device engine has to reproduce from the same tables and batches (the oracle of a k-GPU
    run is the 1-GPU result):
      loss [n, 2]; per step FIN at the batch's user rows (the guest rows: batch order) and at its distinct item rows;
      after the last step, table rows of a sample of users (each from its owner, summed over the ranks: x + 0 + ...) and
      items (replicated): the batch's own and `n_sample` random ones of each side.
    Leaves the engine n steps into training (the timed region goes on from there)."""
    import torch

    B = eng.B
    n = len(batches)
    host = dist.get_backend() != "nccl"
    rng = np.random.default_rng(seed + 12345)
    user_ids = np.unique(np.concatenate([tri[:, 0], rng.integers(0, num_users, n_sample)])).astype(np.int64)
    item_ids = np.unique(np.concatenate([tri[:, 1], tri[:, 2], rng.integers(0, num_items, n_sample)])).astype(np.int64)

    def user_rows():
        mine = (user_ids >= lo) & (user_ids < hi)
        buf = torch.zeros((len(user_ids), eng.d), dtype=torch.float32, device=eng.P.device)
        if mine.any():
            buf[torch.from_numpy(np.nonzero(mine)[0]).to(buf.device)] = eng.P[torch.from_numpy(user_ids[mine] - lo).to(buf.device)]
        buf = buf.cpu() if host else buf
        dist.all_reduce(buf)
        return buf.cpu()

    eng._wait_item_table()
    item_dev = torch.from_numpy(item_ids).to(eng.P.device)
    out = {"triples": np.ascontiguousarray(tri), "user_ids": user_ids, "item_ids": item_ids, "user_rows_before": user_rows(),
           "item_rows_before": eng.item_rows(eng.P)[item_dev].cpu(), "loss": [], "fin_users": [], "fin_items": [],
           "fin_item_ids": []}
    for i, gb in enumerate(batches):
        eng.train_step(gb)
        if i == 0 and rank == 1 and os.environ.get("IDG_BENCH_TEST_BREAK_PARITY") == "1":
            eng._wait_item_table()
            eng.P_u.mul_(1.001)  # tests only: stands in for a broken exchange (tests/test_bench_contract.py)
        out["loss"].append(eng.loss.double().cpu().numpy().copy())
        out["fin_users"].append(eng._guest(eng.FIN, gb.B).cpu())
        out["fin_items"].append(eng.item_rows(eng.FIN)[gb.items].cpu())
        out["fin_item_ids"].append(gb.items.cpu().numpy())
    eng._wait_item_table()
    out["user_rows"] = user_rows()
    out["item_rows"] = eng.item_rows(eng.P)[item_dev].cpu()
    out["loss"] = np.stack(out["loss"])
    assert tri.shape[0] == n * B
    return out if rank == 0 else None


def parity_compare(single, captured, tol=PARITY_TOL):
    """The 1-GPU half: `single` = a freshly initialised PropagationEngine on the WHOLE graph with the same initial tables;
    runs the captured batches through it and returns the `parity_vs_1gpu` object of the bench line — relative errors
    (Frobenius norms over the compared rows; the loss: largest over steps and terms) of the N-rank run against this one,
    and `ok` = all three within tol and `update_rel_err` — the same table rows as UPDATES since the initial tables: an
    N-rank step that left the tables alone would show there — within 100 x tol."""
    import torch

    tri = torch.from_numpy(captured["triples"]).to(single.params.device)
    B = captured["fin_users"][0].shape[0]
    U = single.U
    dev = single.params.device
    urows = torch.from_numpy(captured["user_ids"]).to(dev)
    irows = torch.from_numpy(captured["item_ids"]).to(dev) + U
    before = torch.cat([single.params[urows], single.params[irows]]).cpu()
    init_n = torch.cat([captured["user_rows_before"], captured["item_rows_before"]])
    init_equal = bool(torch.equal(before, init_n))

    def rel(a, b):
        return float((a.double() - b.double()).norm() / b.double().norm().clamp_min(1e-300))

    loss_err, fin_err = 0.0, 0.0
    loss_1 = []
    for i in range(len(captured["fin_users"])):
        u, p, n = (tri[i * B:(i + 1) * B, c].contiguous() for c in range(3))
        loss = single.train_step(u, p, n).double().cpu().numpy().copy()
        loss_1.append(loss)
        loss_err = max(loss_err, float(np.max(np.abs(captured["loss"][i] - loss) / np.maximum(np.abs(loss), 1e-30))))
        items = torch.from_numpy(captured["fin_item_ids"][i]).to(dev)
        fin_1 = torch.cat([single.final[u], single.final[items + U]]).cpu()
        fin_n = torch.cat([captured["fin_users"][i], captured["fin_items"][i]])
        fin_err = max(fin_err, rel(fin_n, fin_1))
    after_1 = torch.cat([single.params[urows], single.params[irows]]).cpu()
    after_n = torch.cat([captured["user_rows"], captured["item_rows"]])
    table_err = rel(after_n, after_1)
    upd_err = rel(after_n - init_n, after_1 - before)
    # (the first steps of a run start at loss ~ ln 2 whatever the tables hold, and the tables move by ~lr per step against
    #  values of ~1e-2: the UPDATES are the sensitive quantity — a loose bound on them is part of the verdict)
    tol_update = 100 * tol
    ok = bool(init_equal and loss_err <= tol and fin_err <= tol and table_err <= tol and upd_err <= tol_update)
    return {"what": "%d training steps from the same initial tables and the same global batches on the N ranks and on ONE "
                    "device (the fused single-GPU engine, rank 0, same run): N-rank result against the 1-GPU result — loss: "
                    "largest relative error over steps and terms; final_rows: FIN at every batch's user and item rows; table: "
                    "%d sampled user + %d item rows after the last Adam step (each batch's own + random ones); relative "
                    "Frobenius norms.  SURVEY.md 8e: the oracle of a k-GPU run is the 1-GPU result"
                    % (len(loss_1), len(captured["user_ids"]), len(captured["item_ids"])),
            "steps": len(loss_1), "loss_rel_err": loss_err, "final_rows_rel_err": fin_err, "table_rel_err": table_err,
            "update_rel_err": upd_err, "table_max_abs_err": float((after_n - after_1).abs().max()),
            "initial_tables_equal": init_equal, "loss_n_ranks": [[float(x) for x in row] for row in captured["loss"]],
            "loss_1_gpu": [[float(x) for x in row] for row in loss_1], "tol": tol, "tol_update": tol_update, "ok": ok}


def run_sharded_bench(args, rank, world, dist, comm, comm_name, single_gpu_reference=None):
    """bench.py --gpus N (N > 1), the north-star split: ONE graph of the named shape cut across the
    ranks by nnz-balanced user-row blocks, item table replicated, ONE global batch of B triples per step (the
    reference takes one Adam step per batch_size triples, trainer.py:36-56) — strong scaling: value = B*steps /
    max-over-ranks time.  single_gpu_reference(args) -> dict: called on rank 0 AFTER the timed region and after the shards
    are freed, it measures the SAME workload unsharded on this rank's GPU (bench.py); the line then carries
    speedup_vs_1gpu.  Returns the bench line (rank 0) or None; the caller emits it and ends the process group."""
    import time

    import torch

    from . import synth as S

    phase = getattr(args, "phase", None) or (lambda name: None)  # bench.py's supervisor quotes it if the run is ended
    phase("graph")
    U, I, E = S.SHAPES[args.workload]
    d, K, B = args.dim, args.layers, args.batch
    # every rank needs the same global graph: large ones are drawn once per machine, then loaded
    if world > 1 and E >= int(os.environ.get("IDG_SYNTH_SHARED_MIN_EDGES", "50000000")):
        users, items = S.generate_shared(U, I, E, 0, rank, dist.barrier)
    else:
        users, items = S.generate(U, I, E, seed=0)
    user_degree = np.bincount(users, minlength=U)
    bounds = partition_users_by_nnz(user_degree, world)
    lo, hi = int(bounds[rank]), int(bounds[rank + 1])
    ui, iu = shard_adjacency_from_edges(users, items, U, I, lo, hi)  # this rank's rows only: no global CSR
    nnz_global, n_edges = 2 * len(users), len(users)
    # slices of the item panel: what a collective moves at a time.  From 4 ranks on the communicator's stream is the
    # critical path and a collective can start when its first slice exists: finer slices there
    n_slices = int(getattr(args, "item_slices", 0) or 0) or ((8 if world >= 4 else 4) if I * d * 4 >= (256 << 20) else 1)
    cuts = partition_users_by_nnz(np.bincount(items, minlength=I), n_slices)  # same cuts on every rank: global item degrees
    n_timeline = 3 if world > 1 or os.environ.get("IDG_BENCH_TIMELINE") == "1" else 0  # instrumented steps, after the timed ones
    # parity_vs_1gpu: the first n_parity batches run BEFORE the warm-up, on the fresh tables, and rank 0's single-GPU
    # reference repeats them after the timed region (no reference in this run -> nothing to compare with)
    n_parity = max(int(getattr(args, "parity_steps", 3)), 0) if single_gpu_reference is not None else 0
    need = (n_parity + args.steps + args.warmup + n_timeline) * B
    tri = S.draw_triples(args.seed, users, items, U, I, need)[0]  # the same global sequence on every rank
    edges = (users, items) if (rank == 0 and single_gpu_reference is not None) else None  # (rank 0 measures the 1-GPU point later)
    del users, items
    phase("engine")
    kern = HipKernels()
    panel_bits = int(getattr(args, "panel_bits", 32) or 32)
    rank_order = panel_bits == 24 or getattr(args, "reduce_order", "rccl") == "rank"
    if rank_order:
        # opt-in: the panel-sized exchanges as an explicit exchange summed in rank order by this library's own kernel —
        # 24-bit rows (idg_reduce24_f32) or fp32 blocks (idg_reduce_blocks_f32)
        comm = Packed24Comm(comm, kern, bits=panel_bits)
        comm_name += (" + 24-bit panel exchange (pack -> all-to-all -> rank-ordered sum -> all-gather)" if panel_bits == 24 else
                      " + explicit fp32 panel exchange (all-to-all -> rank-ordered sum -> all-gather)")
    # global_user_degree: the touched-item exchanges are sized by a host-side bound — no host synchronisation in the step
    eng = ShardedEngine(kern, comm, ui, iu, hi - lo, I, d, K, True, 1e-4, 1e-3, batch_size=B, user_lo=lo,
                        n_slices=n_slices, item_cuts=cuts, store_grad=False, global_user_degree=user_degree)
    del user_degree
    nnz_ui, nnz_iu = len(ui[1]), len(iu[1])
    del ui, iu
    Ug = hi - lo
    # same initialisation as a single-device run of this graph would draw; the item block on every rank
    g = torch.Generator().manual_seed(args.seed)
    bu, bi = (6.0 / (U + d)) ** 0.5, (6.0 / (I + d)) ** 0.5
    for r in range(world):  # the user table is drawn block by block so that no rank holds all of it
        blk = (torch.rand(int(bounds[r + 1] - bounds[r]), d, generator=g) * 2 - 1) * bu
        if r == rank:
            eng.P[:Ug].copy_(blk)
        del blk
    eng.item_rows(eng.P).copy_((torch.rand(I, d, generator=g) * 2 - 1) * bi)
    batches = [eng.make_batch(tri[i * B:(i + 1) * B, 0], tri[i * B:(i + 1) * B, 1], tri[i * B:(i + 1) * B, 2])
               for i in range(n_parity + args.steps + args.warmup + n_timeline)]
    first = n_parity
    last = first + args.warmup + args.steps - 1

    def step(i):
        if i < last:
            eng.prefetch(batches[i + 1])  # index-only work of the next batch, off the critical path
        return eng.train_step(batches[i])

    captured = None
    if n_parity:
        phase("parity_steps")
        captured = parity_capture(eng, batches[:n_parity], tri[:n_parity * B], dist, rank, lo, hi, U, I, seed=args.seed)
    S.ramp_clocks()
    phase("warmup")
    for i in range(first, first + args.warmup):
        step(i)
    dist.barrier()
    torch.cuda.synchronize()
    phase("timed")
    wire0 = dict(comm.wire) if rank_order else None
    t0 = time.perf_counter()
    for i in range(first + args.warmup, first + args.warmup + args.steps):
        step(i)
    t_enqueue = time.perf_counter() - t0  # host time to issue the steps (== wall time when the host is the bottleneck)
    wire1 = dict(comm.wire) if rank_order else None
    eng._wait_item_table()                # (the last step's all-gathers of the updated item rows belong to it)
    torch.cuda.synchronize()
    dist.barrier()
    dt = torch.tensor([time.perf_counter() - t0], dtype=torch.float64,
                      device="cuda" if dist.get_backend() == "nccl" else "cpu")
    dist.all_reduce(dt, op=dist.ReduceOp.MAX)
    dt = float(dt.item())
    phase("after_timed")
    # coherence of the item table (updated by its owners, all-gathered): a checksum must agree on every rank
    chk = eng.item_rows(eng.P).double().sum().reshape(1)
    chk = chk if dist.get_backend() == "nccl" else chk.cpu()
    c_lo, c_hi = chk.clone(), chk.clone()
    dist.all_reduce(c_lo, op=dist.ReduceOp.MIN)
    dist.all_reduce(c_hi, op=dist.ReduceOp.MAX)

    # per-rank roofline of the two products a layer consists of (after the timed region, this rank only)
    def timed(fn, reps=5):
        fn()
        a, b = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
        a.record()
        for _ in range(reps):
            fn()
        b.record()
        torch.cuda.synchronize()
        return a.elapsed_time(b) / reps * 1e-3

    t_ui = timed(lambda: kern.spmm(eng.G_ui, eng._i(eng.P), Y=eng.XU[0]))

    def item_side():
        for gph, r0, r1, r1p in eng.slices:
            kern.spmm(gph, eng._u(eng.P), Y=eng.XI[0][r0:r1])

    t_iu = timed(item_side)
    n_touched = eng.touched_items[1] if eng.touched_items is not None else 0
    # instrumented steps (every rank: they hold collectives): what each collective cost and how long the step's stream
    # waited for it — the diagnosis of a line below the 6x target
    timeline = None
    if n_timeline:
        phase("timeline")
        tl = StepTimeline(torch, world)
        eng._wait_item_table()
        inner, inner_k, eng.comm, eng.timeline = eng.comm, eng.k, TimelineComm(eng.comm, tl), tl
        order = IssueOrder().attach(eng)  # the same steps' launch ORDER, checked against DESIGN.md §7's overlap model
        for i in range(last + 1, last + 1 + n_timeline):
            ev = (tl.event(), tl.event())
            ev[0].record()
            order.begin_step()
            eng.train_step(batches[i])
            ev[1].record()
            tl.steps.append(ev)
        order.end_steps()
        eng._wait_item_table()
        torch.cuda.synchronize()
        eng.comm, eng.k, eng.timeline = inner, inner_k, None
        timeline = tl.summary()
        # every rank checked its own sequence: the line carries rank 0's and the number of ranks that found a violation
        bad = torch.tensor([1.0 if order.violations() else 0.0], device="cuda" if dist.get_backend() == "nccl" else "cpu")
        dist.all_reduce(bad)
        timeline["issue_order"] = dict(order.summary(), ranks_with_violations=int(bad.item()))
        timeline["touched_item_rows_exchanged"] = eng.touched_items[1] if eng.touched_items is not None else 0
        timeline["touched_item_agreement"] = ("device-built id list sized by a host-side bound: no host synchronisation"
                                              if eng.user_degree is not None else "length read back: one host synchronisation")
    rows_form = n_touched > 0
    # panel-sized exchanges per step: forward layers 1..K-2, backward steps 2..K-1 as all-reduces (+ layer K-1 and the
    # first backward step when the touched-item form does not apply), and the last backward step as reduce-scatter +
    # all-gather (the same bytes on the wire as one all-reduce)
    n_allreduce = max(K - 2, 0) + max(K - 2, 0) + (0 if rows_form or K < 2 else 2)
    bytes_ui = 4 * (Ug + 1) + 8 * nnz_ui + 4 * nnz_ui * d + 4 * Ug * d
    bytes_iu = 4 * (I + 1) + 8 * nnz_iu + 4 * nnz_iu * d + 4 * I * d
    # every array of the two products read or written once (the gathered panel once, not once per stored entry)
    bytes_min = 4 * (Ug + I + 2) + 8 * (nnz_ui + nnz_iu) + 2 * 4 * (Ug + I) * d
    out = None
    if rank == 0:
        out = {
            "metric": "BPR triples/sec, LightGCN-%d dim=%d" % (K, d),
            "value": B * args.steps / dt, "unit": "triples/s", "n_gpus": world, "steps": args.steps,
            "warmup": args.warmup, "ms_per_step": dt / args.steps * 1e3, "higher_is_better": True, "scaling": "strong",
            "vs_baseline": None, "dtype": "f32", "data": "synthetic",
            "config": {"workload": "%s graph cut across %d ranks by nnz-balanced user-row blocks: %d users x %d items, %d train "
                                   "edges, nnz(A)=%d; LightGCN K=%d d=%d, ONE global batch of B=%d triples per Adam step (as "
                                   "the reference, trainer.py:36); item table replicated, its Adam state and update owned 1/%d "
                                   "per rank; per step %d all-reduces of the [%d,%d] fp32 item panel in %d slices + 1 "
                                   "reduce-scatter and 1 all-gather of it (the last backward product: owners finish the "
                                   "gradient, apply Adam and send the updated rows round under the next step's first "
                                   "product) + 2 of [%d,%d] (the batch's user rows) + 1 of [<=%d,%d] (the last forward "
                                   "layer's item rows)%s over %s"
                                   % (args.workload, world, U, I, n_edges, nnz_global, K, d, B, world, n_allreduce, I, d,
                                      len(eng.slices), B, d, 2 * B, d,
                                      (" + the %d item rows the batch's users touch, for forward layer K-1 and the first "
                                       "backward product (an [%d] flag vector to agree on them, then those rows)"
                                       % (n_touched, I)) * rows_form,
                                      "RCCL" if dist.get_backend() == "nccl" else dist.get_backend() + " (rehearsal, host-staged)"),
                       "batch": B, "dim": d, "layers": K, "parallelism": "user-row shard x%d" % world,
                       "comm": comm_name, "item_panel_slices": len(eng.slices)},
            "loss_last": [float(x) for x in eng.loss.cpu()],
            "host_issue_ms_per_step": t_enqueue / args.steps * 1e3,
            "item_table_coherent": bool(c_lo.item() == c_hi.item()),
            "timeline": timeline,
            # per-step work every rank carries whatever N is (what caps the speed-up before any communication): O(I + B d)
            "replicated_bytes_per_step_per_rank": eng.replicated_bytes_per_step(),
            "roofline": {
                "bound": "hbm", "kernel": "spmm_tile_kernel<%d,...> on rank 0's two blocks: R_g (users x items) and R_g^T "
                                          "(items x users, %d row slices)" % (min(d // 4, 64), len(eng.slices)),
                "achieved": (bytes_ui + bytes_iu) / (t_ui + t_iu) / 1e9, "peak": 8000.0, "unit": "GB/s",
                "frac": (bytes_ui + bytes_iu) / (t_ui + t_iu) / 1e9 / 8000.0, "traffic": None,
                "us_user_side": t_ui * 1e6, "us_item_side": t_iu * 1e6, "bytes_gather_user_side": bytes_ui,
                "bytes_gather_item_side": bytes_iu, "bytes_min": bytes_min,
                "frac_bytes_min": bytes_min / (t_ui + t_iu) / 1e9 / 8000.0, "rank0_users": Ug, "rank0_nnz": nnz_ui,
                "cache_resident": bool(4 * max(I, Ug) * d < (256 << 20)),
                # bytes a rank sends (= receives) per step: an all-reduce or a reduce-scatter + all-gather pair moves
                # 2 (N-1)/N of the buffer per rank
                "exchange_bytes_per_step_per_rank": int(2 * (world - 1) / world *
                                                        ((n_allreduce + 1) * 4 * I * d + 3 * 4 * B * d
                                                         + (2 * 4 * n_touched * d + 4 * I) * rows_form)),
                "exchange_rows": {"touched_items": n_touched, "items": I},
            },
        }
        # bytes one rank puts on the wire per step (SURVEY.md 8e / DESIGN.md 7: what the 1 -> 8 speed-up hangs on)
        fp32_wire = out["roofline"]["exchange_bytes_per_step_per_rank"]
        out["panel_exchange"] = {"bits": panel_bits, "bytes_on_the_wire_per_step_per_rank": fp32_wire,
                                 "reduction_order": "RCCL's (ring / tree by its own choice)"}
        if rank_order:
            sent = (wire1["packed_bytes_sent"] - wire0["packed_bytes_sent"]) / args.steps
            repl = (wire1["fp32_bytes_it_replaces"] - wire0["fp32_bytes_it_replaces"]) / args.steps
            out["panel_exchange"].update(
                bytes_on_the_wire_per_step_per_rank=int(fp32_wire - repl + sent), packed_bytes_per_step_per_rank=int(sent),
                fp32_bytes_they_replace=int(repl), ratio=(sent / repl) if repl else None,
                exchanges_per_step=(wire1["exchanges"] - wire0["exchanges"]) / args.steps,
                reduction_order="rank order (%s: q0 + q1 + ... one fp32 add per rank and element): bit-reproducible run to run"
                                % ("idg_reduce24_f32" if panel_bits == 24 else "idg_reduce_blocks_f32"),
                order_violations=comm.order_violations(),
                what=("the [I, d] all-reduces, the last backward product's reduce-scatter, the all-gather of the updated item rows "
                      "and the touched-item row sets travel as 24-bit values (2^-16 relative), 3 bytes per value; the batch's guest "
                      "rows and other small collectives stay fp32") if panel_bits == 24 else
                     ("the [I, d] all-reduces, the last backward product's reduce-scatter and the touched-item row sets as an explicit "
                      "exchange of fp32 blocks (all-to-all, this library's rank-ordered sum, all-gather): RCCL's bytes, a fixed order"))
    kern.close()
    if rank_order:
        comm.release_buffers()
    del eng, batches, kern
    torch.cuda.empty_cache()
    phase("single_gpu_reference")
    if single_gpu_reference is not None:
        # the SAME workload unsharded on ONE GPU, measured in this run (rank 0; the other ranks wait at the barrier)
        ref = None
        if rank == 0:
            try:
                ref = single_gpu_reference(args, edges, captured)
            except Exception as exc:  # noqa: BLE001 - the headline stands without it; the line says why it is missing
                ref = {"error": "%s: %s" % (type(exc).__name__, str(exc)[:200])}
            out["single_gpu_reference"] = ref
            if n_parity:
                # correctness of the N-rank run, in the line itself: ok = False makes bench.py leave with a non-zero status
                out["parity_vs_1gpu"] = ref.pop("parity_vs_1gpu", None) or {
                    "ok": None, "error": "the single-GPU reference did not run: " + str(ref.get("error", "no result"))}
            if ref.get("ms_per_step"):
                out["speedup_vs_1gpu"] = ref["ms_per_step"] / out["ms_per_step"]
                if world >= 8 and out["speedup_vs_1gpu"] < 6.0:
                    out["north_star_6x"] = ("NOT met: %.2fx at %d GPUs with exact fp32 panels — three [I, d] exchanges per step "
                                            "(%.1f GB sent per rank) against %.0f ms of products per rank"
                                            % (out["speedup_vs_1gpu"], world,
                                               out["roofline"]["exchange_bytes_per_step_per_rank"] / 1e9,
                                               ref["ms_per_step"] / world))
        dist.barrier()
    return out
