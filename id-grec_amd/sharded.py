"""User-row-sharded LightGCN training step across the GPUs of one node.

The reference is single-device; this is new functionality whose oracle is the single-GPU
result.  Rank g owns a contiguous, nnz-balanced block of users: their embedding rows (and
Adam state), the rows R_g of the normalised interaction matrix (U_g x I) and the transposed
block R_g^T (I x U_g) — both with the GLOBAL d^-1/2 scaling, so the stacked blocks equal the
single-GPU adjacency.  The item table and its Adam state are replicated.

One propagation layer:      X_U[g] <- R_g . X_I                  (local)
                            X_I    <- all-reduce_g( R_g^T . X_U[g] )   (RCCL over xGMI)
so a step costs K all-reduces forward and K + 1 backward of one [I, d] fp32 panel (the extra
one completes the item-side BPR gradient, whose contributions are spread over the ranks by
triple ownership).  The item-side product is launched first and its all-reduce overlaps the
user-side product.  The last backward all-reduce carries the item regulariser gradient too,
so every rank ends the step with bit-identical item gradients (RCCL all-reduce returns the
same bits on every rank) and the replicated Adam updates stay coherent without further
exchange.

The layer loop is written once against two small interfaces — `kernels` (SpMM with fused
epilogue, fused BPR, Adam, linear combination, allocation) and `comm` (all-reduce) — so the
world_size-2 gloo tests in tests/ can drive it on CPU with a checker-backed stub while the
product binds it to the HIP kernels (`HipKernels`) and torch.distributed/RCCL (`TorchComm`).
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
      own_users / own_pos / own_neg  int64 [B_g]  the owned triples (local user ids): the rows this rank's restricted
                 products have to produce / may gather from
      items      int64 [n_items <= 2B]  the batch's distinct item ids, ascending: the only item rows of the LAST forward
                 layer anybody reads (the BPR loss gathers pos / neg rows) — what that layer's exchange carries"""
    __slots__ = ("B", "pos", "neg", "own_src", "head_dst", "nxt", "own_users", "own_pos", "own_neg", "n_owned", "key",
                 "items", "n_items")


class ShardedEngine:
    """One rank's share of the LightGCN step.  Arrays are whatever `kernels` allocates (torch CUDA tensors in the
    product).  Row layout of every local panel: this rank's users [0, U_g), then B GUEST rows [U_g, U_g + B) — row
    U_g + t carries triple t's user, whoever owns it — then ALL items [U_g + B, U_g + B + I).

    The batch is GLOBAL (every rank sees all B triples, as the single-device step does): after the forward
    propagation the owners copy the batch users' final and ego rows into the guest rows, two all-reduces of [B, d]
    (x + 0 + ... + 0: exact) hand them to every rank, and every rank evaluates the WHOLE batch's BPR loss — so the
    loss and the item-side gradient g_I are complete and bit-identical on every rank without any [I, d] exchange, and
    the user-side gradient rows flow back from the guest rows to their owners' rows (chained adds in batch order).
    Per training step: K - 2 forward + K - 1 backward all-reduces of the [I, d] item panel (3 at K = 3; round 1: 7), cut
    into slices that overlap the products; the two [B, d] guest-row ones; one of [<= 2B, d] — the LAST
    forward layer's item rows are read at the batch's positive / negative items only —; and the item rows the batch's
    users touch for forward layer K - 1 and the first backward product (_agree_touched_items), agreed through an [I]
    vector of flags.  Where the batch's two-hop item rows are under half the table, forward layer K - 2 and the second
    backward product travel as those rows as well (_agree_two_hop_items: one panel all-reduce left at K = 3, the last
    backward product, whose partials reach three hops).  A set that does not fit its compact buffer, and evaluation,
    use the sliced panel all-reduce."""

    def __init__(self, kernels, comm, ui_csr, iu_csr, n_local_users, num_items, dim, n_layers, include_layer0=True,
                 reg_lambda=1e-4, lr=1e-3, batch_sparsity=True, batch_size=1024, user_lo=0, n_slices=None, item_cuts=None,
                 live_rows_cap=None, live_rows_min_bytes=64 << 20, two_hop_cap=None):
        """batch_sparsity: use what a prepared batch (kernels.prepare) knows — the user side of the last forward
        layer is produced for the batch's owned users only, the first backward product gathers its live rows only,
        the gradient scatter follows a plan sorted ahead of time.  Exact; FIN's user rows outside the batch are then
        stale, which no consumer reads.  batch_size: capacity of the guest rows (the global batch size).
        user_lo: global id of this rank's first user.  n_slices: row slices of R_g^T whose all-reduces overlap the
        following slices' products (default: 4 once the item panel reaches 256 MB, else 1); item_cuts: the slices' row
        bounds — they MUST be the same on every rank (the slices are what the ranks all-reduce); default: equal row
        counts (callers that know the global item degrees pass entry-balanced cuts).  live_rows_cap: rows of the compact
        buffer of the first backward step's exchange (default 64 per triple; the same on every rank);
        live_rows_min_bytes: item panels smaller than this are all-reduced whole in that step (tests pass 0).
        two_hop_cap: rows of the compact buffer of the two-hop exchanges (default I / 2; 0 = off)."""
        self.k, self.comm = kernels, comm
        self.batch_sparsity = bool(batch_sparsity)
        self._prepared = {}
        self.Ug, self.I, self.d, self.K = int(n_local_users), int(num_items), int(dim), int(n_layers)
        self.B, self.lo = int(batch_size), int(user_lo)
        self.c0 = 1 if include_layer0 else 0
        self.cnt = float(self.K + self.c0)
        self.reg_lambda, self.lr = float(reg_lambda), float(lr)
        self.G_ui = kernels.make_graph(*ui_csr, self.Ug, self.I)
        if n_slices is None:
            n_slices = 4 if self.I * self.d * 4 >= (256 << 20) else 1
        iu_ptr, iu_idx, iu_val = iu_csr
        iu_ptr = np.asarray(iu_ptr, dtype=np.int64)
        if item_cuts is None:
            item_cuts = np.linspace(0, self.I, max(1, min(int(n_slices), self.I)) + 1).astype(np.int64)
        cuts = np.asarray(item_cuts, dtype=np.int64).copy()
        # inner cuts on multiples of 32 rows: a slice's share of an item-row bitmap then starts on a word boundary (the
        # rounding is a function of the cuts alone, so the ranks still agree on them)
        cuts[1:-1] = np.minimum((cuts[1:-1] + 16) // 32 * 32, self.I)
        assert cuts[0] == 0 and cuts[-1] == self.I and (np.diff(cuts) >= 0).all(), "item_cuts must tile [0, I]"
        self.G_iu = []
        for r0, r1 in zip(cuts[:-1], cuts[1:]):
            r0, r1 = int(r0), int(r1)
            if r1 == r0:
                continue
            e0, e1 = int(iu_ptr[r0]), int(iu_ptr[r1])
            self.G_iu.append((kernels.make_graph(iu_ptr[r0:r1 + 1] - e0, iu_idx[e0:e1], iu_val[e0:e1], r1 - r0, self.Ug), r0, r1))
        n = self.Ug + self.B + self.I
        z = kernels.zeros
        self.P, self.G, self.M, self.V = z((n, dim)), z((n, dim)), z((n, dim)), z((n, dim))
        self.FIN = z((n, dim))
        self.GF = z((n, dim))   # d loss / d FIN
        self.XU = [z((self.Ug, dim)), z((self.Ug, dim))]
        self.XI = [z((self.I, dim)), z((self.I, dim)), z((self.I, dim))]
        self.CI, self.CT = z((2 * self.B, dim)), z((2 * self.B, dim))  # the batch's item rows, compact (last forward layer)
        # first backward step: flags of the item rows some rank's partial has, and those rows, compact (up to 64 per triple)
        self.FL = z((self.I,))
        self.live_rows_min_bytes = int(live_rows_min_bytes)
        # panel-wide elementwise steps whose operand lives on the batch's items run on those rows (from the same size on:
        # a small graph's step is bound by the host issuing its calls, and these are four calls for one)
        self._rows_ops = self.I * self.d * 4 >= self.live_rows_min_bytes
        self.CS = z((max(1, min(self.I, 64 * self.B if live_rows_cap is None else int(live_rows_cap))), dim))
        # ... and of the TWO-hop item rows (forward layer K - 2, second backward product): up to half the panel
        two_cap = self.I // 2 if two_hop_cap is None else int(two_hop_cap)
        self.CS2 = z((two_cap, dim)) if (two_cap > 0 and self.K >= 3 and getattr(comm, "world", 2) > 1) else None
        self._two_hop_misses, self.two_hop_seen = 0, 0
        self.loss = z((2,))
        self.upstream = z((2,))
        kernels.fill(self.upstream, 1.0)
        self.guest_ids = kernels.to_device(np.arange(self.Ug, self.Ug + self.B, dtype=np.int64))
        self.step_count = 0

    def _u(self, a):
        return a[: self.Ug]

    def _guest(self, a, count=None):
        return a[self.Ug: self.Ug + (self.B if count is None else count)]

    def _i(self, a):
        return a[self.Ug + self.B:]

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
        gb.own_pos, gb.own_neg = to(np.asarray(pos, dtype=np.int64)[owned]), to(np.asarray(neg, dtype=np.int64)[owned])
        items = np.unique(np.concatenate([np.asarray(pos, dtype=np.int64), np.asarray(neg, dtype=np.int64)]))
        gb.items, gb.n_items = to(items), len(items)
        gb.key = id(gb)
        return gb

    # ---- item-side product, slice by slice; each slice's all-reduce starts as soon as the slice is produced
    def _item_side(self, X_u, Y_i, x_rows=None, after_first=None):
        works = []
        for j, (g, r0, r1) in enumerate(self.G_iu):
            self.k.spmm(g, X_u, Y=Y_i[r0:r1], x_rows=x_rows)
            if j == 0 and after_first is not None:
                after_first()
            works.append(self.comm.all_reduce_async(Y_i[r0:r1]))
        return works

    def _wait_all(self, works):
        for w in works:
            if isinstance(w, tuple):  # an exchange of rows (_sum_rows_async): the sums go back into their panel
                work, panel, ids, buf = w
                self.comm.wait(work)
                self.k.scatter_rows(panel, ids, buf)
            else:
                self.comm.wait(w)

    def _agree_touched_items(self, prep, gb):
        """The item rows a training batch touches beyond its own positives / negatives: the items its USERS interacted
        with.  Every rank flags those of the batch users it owns (idg_graph_flag_cols over its block of R), adds the
        batch's items, the flags are summed over the ranks ([I] floats: 20 MB at configs[4]) and every rank reads off
        the same ascending id list — the one host synchronisation of the step.  Two exchanges then carry these rows
        instead of the [I, d] panel: layer K - 1 of the forward pass (the last user-side product, restricted to the
        batch's users, gathers X_I(K-1) at exactly these rows, and FIN needs it at the batch's items) and the first
        backward product (whose partials are zero elsewhere).  Returns (ids, n) or None: world size 1, a panel too
        small to be worth it, kernels without the index work, or more rows than the compact buffer holds."""
        if gb is None or getattr(self.comm, "world", 2) == 1 or self.I * self.d * 4 < self.live_rows_min_bytes:
            return None
        k = self.k
        if not k.flag_touched_items(self, prep, gb, self.FL):
            return None
        self.comm.wait(self.comm.all_reduce_async(self.FL))
        ids, n = k.nonzero_ids(self.FL)
        if n == 0 or n > self.CS.shape[0]:
            return None
        return ids, n

    def _sum_rows(self, panel, rows):
        """all-reduce of the panel's rows `rows` = (ids, n) through the compact buffer (synchronous: small)."""
        ids, n = rows
        self.k.gather_rows(self.CS[:n], panel, ids)
        self.comm.wait(self.comm.all_reduce_async(self.CS[:n]))
        self.k.scatter_rows(panel, ids, self.CS[:n])

    def _rows_lincomb(self, dst, a, src, b, gb):
        """dst[r] = a.dst[r] + b.src[r] at the batch's item rows r — the whole-panel lincomb when src is zero elsewhere and
        a == 1, for ~2B rows of traffic instead of three passes over [I, d] (5 GB each at configs[4], on every rank)."""
        n, k = gb.n_items, self.k
        k.gather_rows(self.CI[:n], dst, gb.items)
        k.gather_rows(self.CT[:n], src, gb.items)
        k.lincomb(self.CT[:n], self.CI[:n], a, self.CT[:n], b)
        k.scatter_rows(dst, gb.items, self.CT[:n])

    def _sum_rows_async(self, panel, rows, buf):
        """The same through `buf`, the products that follow overlapping the collective; _wait_all scatters."""
        ids, n = rows
        self.k.gather_rows(buf[:n], panel, ids)
        return (self.comm.all_reduce_async(buf[:n]), panel, ids, buf[:n])

    def _agree_two_hop_items(self, prep, gb):
        """One hop further, by what each layer's consumers read (K = 3; S_U / S_I = the batch's users / items):
          near users  U' = S_U + the owned users that interacted with an item of S_I — where layer K - 1's user rows are
                           read (by the last item-side product, restricted to S_I, and by FIN at S_U) and where the
                           second backward product's input h_U lives;
          two-hop items T = S_I + the items U' interacted with — where forward layer K - 2 is read (by the user-side
                           product of layer K - 1, restricted to U') and outside which the second backward partial is 0:
                           both travel as those rows (the sets are nested: T contains the touched items);
          far users   U'' = S_U + the owned users that interacted with a TOUCHED item — where layer K - 2's user rows are
                           read (by layer K - 1's item-side product, restricted to the touched items) and where the last
                           backward product's input lives (a local fact: no exchange; that product's partial reaches
                           the three-hop items, most of the table, and is all-reduced as the panel).
        A second [I] flag exchange and host synchronisation.  Used while the two-hop items fit the compact buffer (half
        the table by default) — then the near / far users are few as well and the products around these exchanges run
        row-restricted / in sparse-input form; on graphs whose popular items reach most users within a hop (every
        power-law shape of synth.SHAPES at B = 1024: scripts/hop_sets.py) the set is the whole table, and after three
        steps in a row that did not fit the engine stops asking (every rank sees the same counts: same decision).  Sets
        self.two_hop = (ids, n), self.two_hop_bits, self.near_bits / self.far_bits (bitmaps; None with kernels that
        produce every row) and self.hops."""
        self.two_hop = self.two_hop_bits = self.near_bits = self.far_bits = None
        self.hops = False
        if self.touched_items is None or self.K < 3 or self.CS2 is None or self._two_hop_misses >= 3:
            return
        k = self.k
        if not k.flag_two_hop_items(self, prep, gb, self.FL):
            return
        self.comm.wait(self.comm.all_reduce_async(self.FL))
        ids, n = k.nonzero_ids(self.FL)
        self.two_hop_seen = n
        if n == 0 or n > self.CS2.shape[0]:
            self._two_hop_misses += 1
            if self._two_hop_misses >= 3:
                self.CS2 = None  # (its memory goes back: half an item panel)
            return
        self._two_hop_misses = 0
        self.two_hop, self.hops = (ids, n), True
        self.two_hop_bits = k.item_rows_bitmap(self, prep, self.two_hop, which=1)
        self.near_bits, self.far_bits = k.user_rows_bitmaps(self, prep, gb, self.touched_bits)

    # ---- forward: FIN = mean_k A^k P  (users: local rows, items: replicated)
    def propagate(self, prep=None, gb=None):
        """Layer k: P_I(k) = R^T X_U(k-1) (local partial) -> all-reduce -> X_I(k);  X_U(k) = R X_I(k-1).
        P_I(k+1) needs only X_U(k), not X_I(k): it is launched BEFORE waiting for all-reduce k, so the
        collectives queue back to back on the communicator while the SpMMs keep the GPU busy.
        gb (a training step's batch): X_I(K) feeds nothing but FIN's item rows, which the step reads at the batch's
        items only — the last layer's partials are exchanged as those <= 2B rows (gathered into a compact buffer, one
        small all-reduce, folded into FIN at those rows) instead of the [I, d] panel; FIN's other item rows are then
        stale, like its user rows outside the batch.  Evaluation calls propagate() without a batch: every row."""
        k, K, c0, cnt = self.k, self.K, self.c0, self.cnt
        fin_u, fin_i = self._u(self.FIN), self._i(self.FIN)
        xu_prev, xi_prev = self._u(self.P), self._i(self.P)
        pending = [None]  # (works, xi_new, layer, xi_before) of the all-reduce whose result has not been folded in yet
        self.touched_items = self._agree_touched_items(prep, gb) if K >= 2 else None
        if self.touched_items is not None:
            self.touched_bits = k.item_rows_bitmap(self, prep, self.touched_items)
        elif (gb is not None and K >= 2 and getattr(self.comm, "world", 2) == 1
              and self.I * self.d * 4 >= self.live_rows_min_bytes):
            # a single rank has nothing to agree on or exchange: the bitmap alone (no host synchronisation) restricts the
            # products exactly as on the ranks of a larger job (same size rule: on a small graph the touched items are
            # most of the table — yelp2018 shape: 70 % — and the restricted forms cost more than they save)
            self.touched_bits = k.touched_bitmap_local(self, prep, gb)
        else:
            self.touched_bits = None
        self._agree_two_hop_items(prep, gb)

        def finish():
            if pending[0] is None:
                return
            works, xi_new, layer, xi_before = pending[0]
            pending[0] = None
            self._wait_all(works)
            scale = 1.0 / cnt if layer == K else 1.0
            if layer == 1:
                base = self._i(self.P) if c0 else None
            else:
                base = fin_i if (c0 or layer > 2) else xi_before
            if xi_new is None or isinstance(xi_new, tuple) or (gb is not None and self._rows_ops):
                # a layer of a training step: FIN is updated at the batch's items, the only item rows of FIN the step
                # reads — the last layer's rows exist there only (self.CI), layer K - 1's at the touched items, and for a
                # whole panel the fold is still ~2B rows of work instead of three passes over [I, d]
                n_t = gb.n_items
                if xi_new is not None:
                    k.gather_rows(self.CI[:n_t], xi_new[0] if isinstance(xi_new, tuple) else xi_new, gb.items)
                if base is not None:
                    k.gather_rows(self.CT[:n_t], base, gb.items)
                k.lincomb(self.CT[:n_t], self.CI[:n_t], scale, self.CT[:n_t] if base is not None else None, scale)
                k.scatter_rows(fin_i, gb.items, self.CT[:n_t])
                return
            k.lincomb(fin_i, xi_new, scale, base, scale)

        for layer in range(1, K + 1):
            last = layer == K
            xi_new = self.XI[layer % 3]
            if last and gb is not None:
                # item-side partial of the last layer: produced (all rows, or the batch's with a prepared item bitmap),
                # the batch's rows gathered, ONE small all-reduce
                item_bits = getattr(prep, "item_bitmap", None) if prep is not None else None
                for j, (g, r0, r1) in enumerate(self.G_iu):
                    k.spmm(g, xu_prev, Y=xi_new[r0:r1], out_rows=None if item_bits is None else item_bits[r0 // 32:])
                    if j == 0:
                        finish()
                k.gather_rows(self.CI[:gb.n_items], xi_new, gb.items)
                works, xi_fold = [self.comm.all_reduce_async(self.CI[:gb.n_items])], None
            elif layer == K - 1 and (self.touched_items is not None or self.touched_bits is not None):
                # layer K - 1 of a training step is read at the touched items only (by the last user-side product, and
                # by FIN at the batch's items): its partials travel as those rows (one rank: nothing travels)
                bits = self.touched_bits
                for j, (g, r0, r1) in enumerate(self.G_iu):
                    k.spmm(g, xu_prev, Y=xi_new[r0:r1], out_rows=None if bits is None else bits[r0 // 32:])
                    if j == 0:
                        finish()
                if self.touched_items is not None:
                    self._sum_rows(xi_new, self.touched_items)
                works, xi_fold = [], (xi_new,)
            elif layer == K - 2 and self.two_hop is not None:
                # ... and layer K - 2 at the two-hop items (_agree_two_hop_items); the user-side product below overlaps
                # the collective
                bits = self.two_hop_bits
                for j, (g, r0, r1) in enumerate(self.G_iu):
                    k.spmm(g, xu_prev, Y=xi_new[r0:r1], out_rows=None if bits is None else bits[r0 // 32:])
                    if j == 0:
                        finish()
                works, xi_fold = [self._sum_rows_async(xi_new, self.two_hop, self.CS2)], (xi_new,)
            else:
                # item-side partial of this layer; X_I(layer-1) (the previous collective) is folded in under its first slice
                works, xi_fold = self._item_side(xu_prev, xi_new, after_first=finish), xi_new
            if layer == 1:
                sum_in = self._u(self.P) if c0 else None
            else:
                sum_in = fin_u if (c0 or layer > 2) else xu_prev
            xu_new = None if last else self.XU[layer & 1]
            if last:
                u_rows = prep.bitmap if prep is not None else None  # BPR reads the batch's users only
            elif self.hops and layer >= K - 2:  # layer K - 1 is read at the near users, layer K - 2 at the far ones
                u_rows = self.near_bits if layer == K - 1 else self.far_bits
            else:
                u_rows = None
            k.spmm(self.G_ui, xi_prev, Y=xu_new, sum_in=sum_in, sum_out=fin_u, div=cnt if last else 1.0, out_rows=u_rows)
            pending[0] = (works, xi_fold, layer, xi_prev)
            xu_prev, xi_prev = xu_new, xi_new
        finish()
        return self.FIN

    # ---- backward of the above given GF = d loss / d FIN (g_I complete on every rank, g_U at the owners' rows),
    #      accumulated onto G (which already holds the regulariser gradient: item rows complete, user rows owned)
    def propagate_backward(self, prep=None, gb=None, adam_step=None):
        """Horner steps h <- A.h + g, k = K..2, then gE0 = (A.h + c0.g)/cnt.  In block form
        (A.h)_U = R_g h_I (local), (A.h)_I = all-reduce(R_g^T h_U).  As in the forward pass the item-side
        partial of a step needs only the LOCAL h_U, so it is launched before waiting for the previous
        all-reduce; the user-side product is what waits for it.  adam_step (train_step passes its step number): the
        replicated item rows are completed AND Adam-updated slice by slice as each slice's last all-reduce lands, under
        the following slices' collectives — every rank carries this tail for all I rows, after the step's last exchange,
        so it is on the critical path of every step at any N."""
        k, K, c0, cnt = self.k, self.K, self.c0, self.cnt
        g_u, g_i = self._u(self.GF), self._i(self.GF)
        h_u = g_u
        pending = [("g", [], g_i)]                            # h_I of the coming step: g_I itself, nothing to wait for
        h_i = [None]

        def finish():                                         # -> h_I usable
            kind, works, buf = pending[0]
            self._wait_all(works)
            if kind == "t":                                   # (A h)_I + g_I
                if gb is not None and self._rows_ops:
                    self._rows_lincomb(buf, 1.0, g_i, 1.0, gb)  # (g_I is zero outside the batch's items)
                else:
                    k.lincomb(buf, buf, 1.0, g_i, 1.0)
            h_i[0] = buf

        live = prep.bitmap if prep is not None else None      # h_U = g_U has the batch's owned users as its only live rows
        for layer in range(K, 1, -1):
            t_i = self.XI[layer % 3]
            if layer == K and gb is not None:
                # the first step of a training batch: h_U = g_U is non-zero at the batch's users only — the partial is
                # zero outside the items those users interacted with and travels as those rows (_agree_touched_items)
                for j, (g, r0, r1) in enumerate(self.G_iu):
                    k.spmm(g, h_u, Y=t_i[r0:r1], x_rows=live)
                    if j == 0:
                        finish()
                if self.touched_items is not None:
                    self._sum_rows(t_i, self.touched_items)  # (its non-zero rows lie inside the touched items)
                    works = []
                else:
                    works = [self.comm.all_reduce_async(t_i[r0:r1]) for _, r0, r1 in self.G_iu]
                x_items = getattr(prep, "item_bitmap", None) if prep is not None else None  # h_I = g_I: the batch's items
            elif layer == K - 1 and gb is not None and self.hops:
                # the second step: h_U is zero outside the near users, the partial outside the two-hop items
                for j, (g, r0, r1) in enumerate(self.G_iu):
                    k.spmm(g, h_u, Y=t_i[r0:r1], x_rows=self.near_bits)
                    if j == 0:
                        finish()
                works = [self._sum_rows_async(t_i, self.two_hop, self.CS2)]
                x_items = self.touched_bits                    # h_I = (first product, zero outside the touched items) + g_I
            else:
                works = self._item_side(h_u, t_i, x_rows=live, after_first=finish)   # partial of (A h)_I: needs h_U only
                # the second step's h_I = (first product: zero outside the touched items) + g_I, however it was exchanged
                x_items = self.touched_bits if (layer == K - 1 and gb is not None) else None
            live = None
            t_u = self.XU[layer & 1]
            k.spmm(self.G_ui, h_i[0], Y=t_u, addend=g_u, x_rows=x_items)  # (A h)_U + g_U
            pending[0] = ("t", works, t_i)
            h_u = t_u
        # last Horner step, scaled by 1/cnt
        t_i = self.XI[1]  # 3-buffer rotation: never the buffer of the all-reduce still in flight (layer 2 -> XI[2])
        if K == 3 and gb is not None and self.hops:
            live = self.far_bits  # h_U(1) = R h_I(2) + g_U is zero outside the far users (h_I(2) lives on the touched items)
        works = self._item_side(h_u, t_i, x_rows=live, after_first=finish)       # (K == 1: h_U is still g_U)
        k.spmm(self.G_ui, h_i[0], sum_in=g_u if c0 else None, sum_out=self._u(self.G), div=cnt, accumulate=True)
        if c0:                                                                   # reg_I + g_I/cnt, under the collective
            if gb is not None and self._rows_ops:
                self._rows_lincomb(self._i(self.G), 1.0, g_i, 1.0 / cnt, gb)
            else:
                k.lincomb(self._i(self.G), self._i(self.G), 1.0, g_i, 1.0 / cnt)
        G_i, P_i, M_i, V_i = self._i(self.G), self._i(self.P), self._i(self.M), self._i(self.V)
        if os.environ.get("IDG_SHARD_TAIL", "1") == "0":                         # (A/B knob: the tail after ALL slices)
            self._wait_all(works)
            works = [None] * len(works)
        for w, (_, r0, r1) in zip(works, self.G_iu):                             # (one collective per slice, in order)
            if w is not None:
                self._wait_all([w])
            k.lincomb(G_i[r0:r1], t_i[r0:r1], 1.0 / cnt, G_i[r0:r1], 1.0)        # + (sum of the ranks' partials)/cnt
            if adam_step is not None:
                k.adam(P_i[r0:r1], G_i[r0:r1], M_i[r0:r1], V_i[r0:r1], self.lr, adam_step)
        return self.G

    def train_step(self, gb):
        """gb: a GlobalBatch from make_batch() — the same global batch on every rank."""
        k = self.k
        Bc = gb.B
        prep = None
        if self.batch_sparsity:
            prep = self._prepared.pop(gb.key, None)
            if prep is None:
                prep = k.prepare(self, gb)  # None for kernels without one
            if prep is not None:
                k.wait_rows(prep)
        self.propagate(prep, gb)
        # the batch's user rows (final and ego) travel through the guest rows: owners fill, everybody else adds zeros
        fin_g, ego_g = self._guest(self.FIN, Bc), self._guest(self.P, Bc)
        k.gather_rows(fin_g, self._u(self.FIN), gb.own_src)
        k.gather_rows(ego_g, self._u(self.P), gb.own_src)
        w1 = self.comm.all_reduce_async(fin_g)
        w2 = self.comm.all_reduce_async(ego_g)
        k.fill(self.G, 0.0)
        k.fill(self.GF, 0.0)
        self.comm.wait(w1)
        self.comm.wait(w2)
        k.bpr(self.FIN, self.P, self.Ug + self.B, self.guest_ids[:Bc], gb.pos, gb.neg, self.reg_lambda, self.upstream,
              self.GF, self.G, self.loss, prep)
        # gradients of the guest rows go home: every owned user's occurrences are added in batch order
        k.chain_add_rows(self._u(self.GF), self._guest(self.GF, Bc), gb.head_dst, gb.nxt)
        k.chain_add_rows(self._u(self.G), self._guest(self.G, Bc), gb.head_dst, gb.nxt)
        self.step_count += 1
        self.propagate_backward(prep, gb, adam_step=self.step_count)  # (updates the item rows, slice by slice)
        if prep is not None:
            k.release(prep)
        sl = slice(0, self.Ug)  # the owned user rows (the guest rows are not parameters)
        k.adam(self.P[sl], self.G[sl], self.M[sl], self.V[sl], self.lr, self.step_count)
        return self.loss

    def prefetch(self, gb):
        """One-batch lookahead of the index-only work of the NEXT step (row bitmap + sorted scatter plan), on
        the kernels' side stream while this step's products run."""
        if self.batch_sparsity:
            while len(self._prepared) >= 2:  # lookaheads nobody came for (a skipped batch)
                self.k.release(self._prepared.pop(next(iter(self._prepared))))
            prep = self.k.prepare(self, gb)
            if prep is not None:
                self._prepared[gb.key] = prep

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
            top = self.k.topk(self._u(self.FIN), self._i(self.FIN), local, kmax, excl_indptr, excl_items)
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


# --------------------------------------------------------------------------- product bindings
class HipKernels:
    """`kernels` bound to libidgrec.so on the current HIP device."""

    def __init__(self, device=None, deterministic=True):
        import torch

        from . import ops

        self.torch, self.ops = torch, ops
        self.device = torch.device("cuda", torch.cuda.current_device()) if device is None else torch.device(device)
        self.deterministic = deterministic
        self._pool = []
        # side stream of the batch preparation, claimed at construction (ops.side_stream: hardware-queue placement)
        self._side = ops.side_stream(self.device) if deterministic else None
        self._side_raw = self._side.cuda_stream if deterministic else None
        self._fork = ops.LocalEvent() if deterministic else None

    def zeros(self, shape):
        return self.torch.zeros(shape, dtype=self.torch.float32, device=self.device)

    def fill(self, a, v):
        a.fill_(v)

    def make_graph(self, indptr, indices, values, n_rows, n_cols):
        # both orientations (R_g and R_g^T) are built explicitly by shard_adjacency
        return self.ops.Graph(indptr, indices, values, n_rows, n_cols, device=self.device, symmetric=False,
                              build_transpose=False)

    def spmm(self, graph, X, Y=None, addend=None, sum_in=None, sum_out=None, div=1.0, accumulate=False, out_rows=None,
             x_rows=None):
        self.ops.spmm_ex_raw(graph, X, Y, addend, sum_in, sum_out, div, accumulate, out_rows=out_rows, x_rows=x_rows)

    def lincomb(self, out, x, a, y, b):
        self.ops.lincomb_raw(out, x, a, y, b)

    def bpr(self, fin, ego, n_users, users, pos, neg, reg_lambda, upstream, g_final, g_ego, loss, prep=None):
        if prep is not None:  # the sorted (row, slot) plan is in the prepared workspace already
            prep.done.wait(self.torch.cuda.current_stream().cuda_stream)
            self.ops.bpr_fwd_bwd_raw(fin, ego, users, pos, neg, n_users, reg_lambda, upstream, g_final, g_ego, loss,
                                     deterministic=2, ws=prep.ws)
        else:
            self.ops.bpr_fwd_bwd_raw(fin, ego, users, pos, neg, n_users, reg_lambda, upstream, g_final, g_ego, loss,
                                     self.deterministic)

    def to_device(self, a):
        return self.torch.from_numpy(np.ascontiguousarray(a)).to(self.device)

    def gather_rows(self, dst, src, idx):
        self.ops.rows_gather_raw(dst, src, idx)

    def scatter_rows(self, dst, idx, src):
        """dst[idx[j]] = src[j] (idx distinct)."""
        dst.index_copy_(0, idx, src)

    def flag_touched_items(self, eng, prep, gb, flags):
        """flags[i] = 1 for the items the batch's OWNED users interacted with, and for the batch's own items."""
        if prep is None:
            return False
        flags.zero_()
        eng.G_ui.flag_cols(prep.bitmap, flags)  # (prep.bitmap's first U_g bits are this rank's batch users)
        flags.index_fill_(0, gb.items, 1.0)
        return True

    def item_rows_bitmap(self, eng, prep, rows, which=0):
        """Bitmap over the item rows `rows` = (ids, n), for the row-restricted item-side products (which: 0 = the touched
        items, 1 = the two-hop items; two buffers of the prepared batch)."""
        if prep is None:
            return None
        ids, n = rows
        bits = prep.field2_bitmap if which else prep.field_bitmap
        self.ops.bpr_touch_rows_raw(ids, ids, ids, 0, bits, clear_bits=eng.I)
        return bits

    def touched_bitmap_local(self, eng, prep, gb):
        """Bitmap of the touched items from this rank's rows alone (complete when it is the only rank)."""
        if prep is None:
            return None
        prep.field_bitmap.copy_(prep.item_bitmap)
        eng.G_ui.mark_cols(prep.bitmap, prep.field_bitmap)
        return prep.field_bitmap

    def flag_two_hop_items(self, eng, prep, gb, flags):
        """prep.user_near = the owned users that interacted with one of the batch's items, plus the batch's own; flags[i]
        = 1 for the items those users interacted with, and for the batch's items."""
        if prep is None:
            return False
        self._mark_users(eng, gb, prep.user_near, prep.item_bitmap)
        flags.zero_()
        eng.G_ui.flag_cols(prep.user_near, flags)
        flags.index_fill_(0, gb.items, 1.0)
        return True

    def _mark_users(self, eng, gb, bits, item_bits):
        self.ops.bitmap_clear_raw(bits, eng.Ug)
        for g, r0, r1 in eng.G_iu:
            g.mark_cols(item_bits[r0 // 32:], bits)
        if gb.n_owned > 0:
            self.ops.bpr_touch_rows_raw(gb.own_users, gb.own_users, gb.own_users, 0, bits)

    def user_rows_bitmaps(self, eng, prep, gb, touched_bits):
        """(near, far): the far users interacted with a TOUCHED item (a local fact, marked here)."""
        self._mark_users(eng, gb, prep.user_far, touched_bits)
        return prep.user_near, prep.user_far

    def nonzero_ids(self, flags):
        """Ascending ids of the non-zero flags and their number (a host synchronisation: the caller sizes a collective)."""
        ids = self.torch.nonzero(flags).reshape(-1)
        return ids, int(ids.numel())

    def chain_add_rows(self, dst, src, idx, nxt):
        self.ops.rows_chain_add_raw(dst, src, idx, nxt)

    def topk(self, user_panel, item_panel, users, k, excl_indptr, excl_items):
        """Top-k item ids [len(users), k] (numpy) for local user ids `users`, train items excluded."""
        torch = self.torch
        idx = self.ops.score_topk(user_panel, item_panel, self.to_device(np.asarray(users, dtype=np.int64)), int(k),
                                  self.to_device(np.asarray(excl_indptr, dtype=np.int64)),
                                  self.to_device(np.asarray(excl_items, dtype=np.int32)), apply_sigmoid=True)
        torch.cuda.synchronize()
        return idx.cpu().numpy()

    class _Prepared:
        __slots__ = ("bitmap", "item_bitmap", "field_bitmap", "field2_bitmap", "user_near", "user_far", "ws", "rows_done", "done", "free",
                     "B", "busy")

    def prepare(self, eng, gb):
        """Index-only work of a global batch on a side stream: bitmap of the LOCAL user rows this rank owns in it (the
        rows its restricted products produce / gather), and the sorted scatter plan of the whole batch over the guest
        rows.  Returns None when the scatter is not the deterministic one.  Host cost matters here (the sharded step
        issues ~50 calls): raw stream handles and events allocated once, no stream context manager."""
        if not self.deterministic:
            return None
        torch, ops = self.torch, self.ops
        cap, n_users, n, d = eng.B, eng.Ug + eng.B, eng.Ug + eng.B + eng.I, eng.d
        prep = next((p for p in self._pool if p.B == cap and not p.busy), None)
        if prep is None:
            prep = self._Prepared()
            prep.bitmap = torch.zeros((n + 31) // 32, dtype=torch.int32, device=self.device)
            # ... and of the batch's item rows (ALL triples' positives and negatives: every rank evaluates the whole
            # batch), by global item id: the last forward layer's item-side product produces these rows only
            prep.item_bitmap = torch.zeros((eng.I + 31) // 32 + 1, dtype=torch.int32, device=self.device)
            prep.field_bitmap = torch.zeros((eng.I + 31) // 32 + 1, dtype=torch.int32, device=self.device)
            prep.field2_bitmap = torch.zeros((eng.I + 31) // 32 + 1, dtype=torch.int32, device=self.device)
            prep.user_near = torch.zeros((eng.Ug + 31) // 32 + 1, dtype=torch.int32, device=self.device)
            prep.user_far = torch.zeros((eng.Ug + 31) // 32 + 1, dtype=torch.int32, device=self.device)
            prep.ws, prep.B = ops.bpr_workspace(cap, d, self.device), cap
            prep.rows_done, prep.done, prep.free = ops.LocalEvent(), ops.LocalEvent(), None  # device-local events
            self._pool.append(prep)
        prep.busy = True
        main = torch.cuda.current_stream()
        self._fork.record(main.cuda_stream)  # the id tensors may have just been produced on the main stream,
        self._fork.wait(self._side_raw)      # and the step that last used these buffers is ordered before it
        if gb.n_owned > 0:
            ops.bpr_touch_rows_raw(gb.own_users, gb.own_pos, gb.own_neg, n_users, prep.bitmap, stream=self._side_raw,
                                   clear_bits=n)
        else:
            ops.bitmap_clear_raw(prep.bitmap, n, stream=self._side_raw)
        ops.bpr_touch_rows_raw(gb.pos, gb.pos, gb.neg, 0, prep.item_bitmap, stream=self._side_raw, clear_bits=eng.I)
        prep.rows_done.record(self._side_raw)
        ops.bpr_plan_raw(eng.guest_ids[:gb.B], gb.pos, gb.neg, n_users, n, d, ws=prep.ws, stream=self._side_raw)
        prep.done.record(self._side_raw)
        return prep

    def wait_rows(self, prep):
        """The bitmap is first read by a product on the main stream."""
        prep.rows_done.wait(self.torch.cuda.current_stream().cuda_stream)

    def release(self, prep):
        prep.busy = False  # its buffers go back to the pool; reuse is ordered by the fork event of the next prepare()

    def adam(self, p, g, m, v, lr, step):
        self.ops.adam_step_raw(p, g, m, v, lr, step)


class TorchComm:
    """`comm` on torch.distributed (backend "nccl" == RCCL over xGMI on ROCm).  With the gloo
    backend (tests: two ranks sharing one GPU, or CPU arrays) device tensors are staged through
    the host."""

    def __init__(self, dist):
        self.dist = dist
        self.backend = dist.get_backend()
        self.world = dist.get_world_size()
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
        """out (world x len(t) elements, rank-major) <- every rank's t."""
        import torch

        if isinstance(t, np.ndarray):
            t, out = torch.from_numpy(t), torch.from_numpy(out)  # share memory
        if self.backend == "gloo" and t.is_cuda:
            host = torch.empty(out.shape, dtype=out.dtype)
            self.dist.all_gather_into_tensor(host, t.cpu())
            out.copy_(host)
            return None
        return self.dist.all_gather_into_tensor(out, t, async_op=True)

    def wait(self, work):
        if work is not None:
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
        self.overlap_bytes = int(overlap_bytes)
        self._own = self._own_raw = None   # the collectives' own stream, made on first use
        self._ring, self._next = [], 0     # (issued, done) event pairs, reused round-robin (a layer's slices + the previous layer's: <= 8 collectives in flight)

    def _fork(self):
        """Order the communicator's stream after the current one; returns (raw stream handle, event to record when done)."""
        torch = self.torch
        if self._own is None:
            self._own = torch.cuda.Stream()
            self._own_raw = self._own.cuda_stream
            self._ring = [(torch.cuda.Event(), torch.cuda.Event()) for _ in range(32)]
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

    def all_reduce_async(self, t, average=False):
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
        if out.numel() * 4 >= self.overlap_bytes:
            stream, done = self._fork()
            self.check(self.lib.idg_allgather_f32(self.handle, self._f32(t), self._f32(out), t.numel(), stream),
                       "idg_allgather_f32")
            done.record(self._own)
            return done
        self.check(self.lib.idg_allgather_f32(self.handle, self._f32(t), self._f32(out), t.numel(), self._stream()),
                   "idg_allgather_f32")
        return None

    def wait(self, work):
        if work is not None:
            self.torch.cuda.current_stream().wait_event(work)

    def self_test(self):
        """One all-reduce and one all-gather with known answers; True when both are right on this rank."""
        torch = self.torch
        a = torch.full((1024,), float(self.rank + 1), dtype=torch.float32, device="cuda")
        g = torch.zeros(1024 * self.world, dtype=torch.float32, device="cuda")
        self.all_gather_async(g, a)
        self.all_reduce_async(a)
        big = torch.full((self.overlap_bytes // 4 + 1024,), float(self.rank + 1), dtype=torch.float32, device="cuda")
        work = self.all_reduce_async(big, average=True)   # the second-stream form
        self.wait(work)
        big += 1.0                                        # ordered after the collective by wait()
        torch.cuda.synchronize()
        want = torch.arange(1, self.world + 1, dtype=torch.float32, device="cuda").repeat_interleave(1024)
        return (bool((a == self.world * (self.world + 1) / 2).all().item()) and bool(torch.equal(g, want))
                and bool((big == (self.world + 1) / 2 + 1.0).all().item()))

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

    def all_reduce_async(self, t, average=False):
        return None

    def all_gather_async(self, out, t):
        out[...] = t
        return None

    def wait(self, work):
        pass


# --------------------------------------------------------------------------- bench driver
def run_sharded_bench(args, rank, world, dist, comm, comm_name):
    """bench.py --gpus N (N > 1), the north-star split: ONE graph of the named shape cut across the
    ranks by nnz-balanced user-row blocks, item table replicated, ONE global batch of B triples per step (the
    reference takes one Adam step per batch_size triples, trainer.py:36-56) — strong scaling: value = B*steps /
    max-over-ranks time.  Returns the bench line (rank 0) or None; the caller emits it and ends the process group."""
    import time

    import torch

    from . import host as H
    from . import synth as S

    U, I, E = S.SHAPES[args.workload]
    d, K, B = args.dim, args.layers, args.batch
    # every rank needs the same global graph: large ones are drawn once per machine, then loaded
    if world > 1 and E >= int(os.environ.get("IDG_SYNTH_SHARED_MIN_EDGES", "50000000")):
        users, items = S.generate_shared(U, I, E, 0, rank, dist.barrier)
    else:
        users, items = S.generate(U, I, E, seed=0)
    bounds = partition_users_by_nnz(np.bincount(users, minlength=U), world)
    lo, hi = int(bounds[rank]), int(bounds[rank + 1])
    ui, iu = shard_adjacency_from_edges(users, items, U, I, lo, hi)  # this rank's rows only: no global CSR
    nnz_global, n_edges = 2 * len(users), len(users)
    n_slices = 4 if I * d * 4 >= (256 << 20) else 1
    cuts = partition_users_by_nnz(np.bincount(items, minlength=I), n_slices)  # same cuts on every rank: global item degrees
    need = (args.steps + args.warmup) * B
    tri = S.draw_triples(args.seed, users, items, U, I, need)[0]  # the same global sequence on every rank
    del users, items
    kern = HipKernels(deterministic=not args.atomic)
    eng = ShardedEngine(kern, comm, ui, iu, hi - lo, I, d, K, True, 1e-4, 1e-3, batch_size=B, user_lo=lo,
                        n_slices=n_slices, item_cuts=cuts,
                        two_hop_cap=0 if os.environ.get("IDG_TWO_HOP", "1") == "0" else None)  # (0: A/B of the two-hop form)
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
    eng.P[Ug + B:].copy_((torch.rand(I, d, generator=g) * 2 - 1) * bi)
    batches = [eng.make_batch(tri[i * B:(i + 1) * B, 0], tri[i * B:(i + 1) * B, 1], tri[i * B:(i + 1) * B, 2])
               for i in range(args.steps + args.warmup)]
    last = args.warmup + args.steps - 1

    def step(i):
        if i < last:
            eng.prefetch(batches[i + 1])  # index-only work of the next batch, off the critical path
        return eng.train_step(batches[i])

    S.ramp_clocks()
    for i in range(args.warmup):
        step(i)
    dist.barrier()
    torch.cuda.synchronize()
    t0 = time.perf_counter()
    for i in range(args.warmup, args.warmup + args.steps):
        step(i)
    t_enqueue = time.perf_counter() - t0  # host time to issue the steps (== wall time when the host is the bottleneck)
    torch.cuda.synchronize()
    dist.barrier()
    dt = torch.tensor([time.perf_counter() - t0], dtype=torch.float64,
                      device="cuda" if dist.get_backend() == "nccl" else "cpu")
    dist.all_reduce(dt, op=dist.ReduceOp.MAX)
    dt = float(dt.item())
    # coherence of the replicated item table: a checksum must agree on every rank
    chk = eng.P[Ug + B:].double().sum().reshape(1)
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
        for gph, r0, r1 in eng.G_iu:
            kern.spmm(gph, eng._u(eng.P), Y=eng.XI[0][r0:r1])

    t_iu = timed(item_side)
    # what the LAST step exchanged as rows: forward layer K-1 / first backward product (the touched items), forward layer
    # K-2 / second backward product (the two-hop items)
    n_touched = eng.touched_items[1] if getattr(eng, "touched_items", None) is not None else 0
    n_two_hop = eng.two_hop[1] if getattr(eng, "two_hop", None) is not None else 0
    rows_form = n_touched > 0
    n_panel = (K - 1) + K - (2 if rows_form else 0) - (2 if n_two_hop else 0) if K >= 2 else 1
    bytes_ui = 4 * (Ug + 1) + 8 * nnz_ui + 4 * nnz_ui * d + 4 * Ug * d
    bytes_iu = 4 * (I + 1) + 8 * nnz_iu + 4 * nnz_iu * d + 4 * I * d
    # every array of the two products read or written once (the gathered panel once, not once per stored entry)
    bytes_min = 4 * (Ug + I + 2) + 8 * (nnz_ui + nnz_iu) + 2 * 4 * (Ug + I) * d
    out = None
    if rank == 0:
        n = U + I
        out = {
            "metric": "BPR triples/sec, LightGCN-%d dim=%d" % (K, d),
            "value": B * args.steps / dt, "unit": "triples/s", "n_gpus": world, "steps": args.steps,
            "warmup": args.warmup, "ms_per_step": dt / args.steps * 1e3, "higher_is_better": True, "scaling": "strong",
            "vs_baseline": None, "dtype": "f32", "data": "synthetic",
            "config": {"workload": "%s graph cut across %d ranks by nnz-balanced user-row blocks: %d users x %d items, %d train "
                                   "edges, nnz(A)=%d; LightGCN K=%d d=%d, ONE global batch of B=%d triples per Adam step (as "
                                   "the reference, trainer.py:36); item table replicated; per step %d all-reduces of the "
                                   "[%d,%d] fp32 item panel in %d slices that overlap the products + 2 of [%d,%d] (the "
                                   "batch's user rows) + 1 of [<=%d,%d] (the last forward layer's item rows, read at the "
                                   "batch's items only)%s over %s"
                                   % (args.workload, world, U, I, n_edges, nnz_global, K, d, B, n_panel, I, d, len(eng.G_iu),
                                      B, d, 2 * B, d,
                                      (" + the %d item rows the batch's users touch, for forward layer K-1 and the first "
                                       "backward product (an [%d] flag vector to agree on them, then those rows)"
                                       % (n_touched, I)) * rows_form +
                                      (" + the %d two-hop item rows, for forward layer K-2 and the second backward product "
                                       "(a second flag vector)" % n_two_hop) * bool(n_two_hop),
                                      "RCCL" if dist.get_backend() == "nccl" else dist.get_backend() + " (rehearsal, host-staged)"),
                       "batch": B, "dim": d, "layers": K, "parallelism": "user-row shard x%d" % world,
                       "comm": comm_name, "item_panel_slices": len(eng.G_iu)},
            "loss_last": [float(x) for x in eng.loss.cpu()],
            "host_issue_ms_per_step": t_enqueue / args.steps * 1e3,
            "item_table_coherent": bool(c_lo.item() == c_hi.item()),
            "roofline": {
                "bound": "hbm", "kernel": "spmm_tile_kernel<%d,...> on rank 0's two blocks: R_g (users x items) and R_g^T "
                                          "(items x users, %d row slices)" % (min(d // 4, 64), len(eng.G_iu)),
                "achieved": (bytes_ui + bytes_iu) / (t_ui + t_iu) / 1e9, "peak": 8000.0, "unit": "GB/s",
                "frac": (bytes_ui + bytes_iu) / (t_ui + t_iu) / 1e9 / 8000.0, "traffic": None,
                "us_user_side": t_ui * 1e6, "us_item_side": t_iu * 1e6, "bytes_gather_user_side": bytes_ui,
                "bytes_gather_item_side": bytes_iu, "bytes_min": bytes_min,
                "frac_bytes_min": bytes_min / (t_ui + t_iu) / 1e9 / 8000.0, "rank0_users": Ug, "rank0_nnz": nnz_ui,
                "cache_resident": bool(4 * max(I, Ug) * d < (256 << 20)),
                "exchange_bytes_per_step_per_rank": n_panel * 4 * I * d + 4 * 4 * B * d
                                                    + (2 * 4 * n_touched * d + 4 * I) * rows_form
                                                    + (2 * 4 * n_two_hop * d + 4 * I) * bool(n_two_hop),
                "exchange_rows": {"touched_items": n_touched, "two_hop_items": n_two_hop, "items": I,
                                  "two_hop_items_counted": int(getattr(eng, "two_hop_seen", 0)),
                                  "two_hop_buffer_rows": 0 if eng.CS2 is None else int(eng.CS2.shape[0])},
            },
            "single_gpu_reference": "the same workload on ONE MI355X, unsharded: profiles/r02/bench_c5_single_gpu.json "
                                    "(builder-run; not measured in this run)" if args.workload == "synth-10M" else None,
        }
    del eng, batches
    torch.cuda.empty_cache()
    return out
