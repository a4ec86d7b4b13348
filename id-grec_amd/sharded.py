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


# --------------------------------------------------------------------------- the step
class ShardedEngine:
    """One rank's share of the LightGCN step.  Arrays are whatever `kernels` allocates
    (torch CUDA tensors in the product); row layout of every local panel: this rank's users
    first ([0, U_g)), then ALL items ([U_g, U_g + I))."""

    def __init__(self, kernels, comm, ui_csr, iu_csr, n_local_users, num_items, dim, n_layers, include_layer0=True,
                 reg_lambda=1e-4, lr=1e-3, batch_sparsity=True):
        """batch_sparsity: use what a prepared batch (kernels.prepare) knows — the user side of the last forward
        layer is produced for the batch's users only, the first backward product gathers its live rows only, the
        gradient scatter follows a plan sorted ahead of time.  Exact; FIN's user rows outside the batch are then
        stale, which no consumer reads."""
        self.k, self.comm = kernels, comm
        self.batch_sparsity = bool(batch_sparsity)
        self._prepared = {}
        self.Ug, self.I, self.d, self.K = int(n_local_users), int(num_items), int(dim), int(n_layers)
        self.c0 = 1 if include_layer0 else 0
        self.cnt = float(self.K + self.c0)
        self.reg_lambda, self.lr = float(reg_lambda), float(lr)
        self.G_ui = kernels.make_graph(*ui_csr, self.Ug, self.I)
        self.G_iu = kernels.make_graph(*iu_csr, self.I, self.Ug)
        n = self.Ug + self.I
        z = kernels.zeros
        self.P, self.G, self.M, self.V = z((n, dim)), z((n, dim)), z((n, dim)), z((n, dim))
        self.FIN = z((n, dim))
        # d loss / d FIN, plus ONE extra row: its first two floats are this rank's share of the two losses, so the
        # loss rides in the first backward all-reduce (item rows of GF) instead of needing a collective of its own
        self._gf = z((n + 1, dim))
        self.GF = self._gf[:n]
        self.XU = [z((self.Ug, dim)), z((self.Ug, dim))]
        self.XI = [z((self.I, dim)), z((self.I, dim)), z((self.I, dim))]
        self.loss = self._gf[n, :2]
        self.upstream = z((2,))
        self.step_count = 0

    def _u(self, a):
        return a[: self.Ug]

    def _i(self, a):
        return a[self.Ug:]

    # ---- forward: FIN = mean_k A^k P  (users: local rows, items: replicated)
    def propagate(self, prep=None):
        """Layer k: P_I(k) = R^T X_U(k-1) (local partial) -> all-reduce -> X_I(k);  X_U(k) = R X_I(k-1).
        P_I(k+1) needs only X_U(k), not X_I(k): it is launched BEFORE waiting for all-reduce k, so the
        collectives queue back to back on the communicator while the SpMMs keep the GPU busy."""
        k, K, c0, cnt = self.k, self.K, self.c0, self.cnt
        fin_u, fin_i = self._u(self.FIN), self._i(self.FIN)
        xu_prev, xi_prev = self._u(self.P), self._i(self.P)
        pending = None  # (work, xi_new, layer) of the all-reduce whose result has not been folded in yet

        def finish(pending):
            work, xi_new, layer, xi_before = pending
            self.comm.wait(work)
            scale = 1.0 / cnt if layer == K else 1.0
            if layer == 1:
                base = self._i(self.P) if c0 else None
            else:
                base = fin_i if (c0 or layer > 2) else xi_before
            k.lincomb(fin_i, xi_new, scale, base, scale)

        for layer in range(1, K + 1):
            last = layer == K
            xi_new = self.XI[layer % 3]
            k.spmm(self.G_iu, xu_prev, Y=xi_new)                       # item-side partial of this layer
            if pending is not None:
                finish(pending)                                        # X_I(layer-1) is needed from here on
            work = self.comm.all_reduce_async(xi_new)
            if layer == 1:
                sum_in = self._u(self.P) if c0 else None
            else:
                sum_in = fin_u if (c0 or layer > 2) else xu_prev
            xu_new = None if last else self.XU[layer & 1]
            k.spmm(self.G_ui, xi_prev, Y=xu_new, sum_in=sum_in, sum_out=fin_u, div=cnt if last else 1.0,
                   out_rows=prep.bitmap if (last and prep is not None) else None)  # BPR reads the batch's users only
            pending = (work, xi_new, layer, xi_prev)
            xu_prev, xi_prev = xu_new, xi_new
        finish(pending)
        return self.FIN

    # ---- backward of the above given GF = d loss / d FIN (item rows still per-rank partials),
    #      accumulated onto G (which already holds the regulariser gradient, item rows partial)
    def propagate_backward(self, prep=None):
        """Horner steps h <- A.h + g, k = K..2, then gE0 = (A.h + c0.g)/cnt.  In block form
        (A.h)_U = R_g h_I (local), (A.h)_I = all-reduce(R_g^T h_U).  As in the forward pass the item-side
        partial of a step needs only the LOCAL h_U, so it is launched before waiting for the previous
        all-reduce; the user-side product is what waits for it."""
        k, K, c0, cnt = self.k, self.K, self.c0, self.cnt
        g_u, g_i = self._u(self.GF), self._i(self.GF)
        first = self.comm.all_reduce_async(self._gf[self.Ug:])  # completes the item-side gradient g_I (+ the loss row)
        h_u = g_u
        pending = ("g", first, g_i)                           # h_I of the coming step, not yet usable

        def finish(p):                                        # -> h_I usable
            kind, work, buf = p
            self.comm.wait(work)
            if kind == "t":
                k.lincomb(buf, buf, 1.0, g_i, 1.0)            # (A h)_I + g_I
            return buf

        live = prep.bitmap if prep is not None else None      # h_U = g_U has the batch's users as its only live rows
        for layer in range(K, 1, -1):
            t_i = self.XI[layer % 3]
            k.spmm(self.G_iu, h_u, Y=t_i, x_rows=live)        # partial of (A h)_I: needs h_U only
            live = None
            h_i = finish(pending)                             # previous all-reduce (+ g_I) -> h_I
            work = self.comm.all_reduce_async(t_i)
            t_u = self.XU[layer & 1]
            k.spmm(self.G_ui, h_i, Y=t_u, addend=g_u)         # (A h)_U + g_U
            pending = ("t", work, t_i)
            h_u = t_u
        # last Horner step, scaled by 1/cnt; regulariser gradients ride along
        t_i = self.XI[1]  # 3-buffer rotation: never the buffer of the all-reduce still in flight (layer 2 -> XI[2])
        k.spmm(self.G_iu, h_u, Y=t_i, x_rows=live)            # (live only when K == 1: h_U is still g_U)
        h_i = finish(pending)
        k.lincomb(t_i, t_i, 1.0 / cnt, self._i(self.G), 1.0)  # partial/cnt + this rank's item reg grads
        work = self.comm.all_reduce_async(t_i)
        k.spmm(self.G_ui, h_i, sum_in=g_u if c0 else None, sum_out=self._u(self.G), div=cnt, accumulate=True)
        self.comm.wait(work)
        k.lincomb(self._i(self.G), t_i, 1.0, g_i if c0 else None, 1.0 / cnt)
        return self.G

    def train_step(self, users_local, pos, neg, global_batch):
        """users_local: ids re-based to this rank's block; pos/neg: global item ids;
        global_batch: total triples over all ranks this step (the mean's divisor)."""
        k = self.k
        B = len(users_local)
        prep = None
        if self.batch_sparsity and B > 0:
            prep = self._prepared.pop(_batch_key(users_local, pos, neg), None)
            if prep is None:
                prep = k.prepare(users_local, pos, neg, self.Ug, self.Ug + self.I, self.d)  # None for kernels without one
            if prep is not None:
                k.wait_rows(prep)
        self.propagate(prep)
        k.fill(self.G, 0.0)
        k.fill(self._gf, 0.0)
        ratio = float(B) / float(global_batch)
        if B > 0:
            k.fill(self.upstream, ratio)
            k.bpr(self.FIN, self.P, self.Ug, users_local, pos, neg, self.reg_lambda, self.upstream, self.GF, self.G,
                  self.loss, prep)
            k.lincomb(self.loss, self.loss, ratio, None, 0.0)         # local mean -> share of the global mean
        self.propagate_backward(prep)                          # its first all-reduce also sums the loss shares
        if prep is not None:
            k.release(prep)
        self.step_count += 1
        k.adam(self.P, self.G, self.M, self.V, self.lr, self.step_count)
        return self.loss


    def prefetch(self, users_local, pos, neg):
        """One-batch lookahead of the index-only work of the NEXT step (row bitmap + sorted scatter plan), on
        the kernels' side stream while this step's products run."""
        if self.batch_sparsity and len(users_local) > 0:
            while len(self._prepared) >= 2:  # lookaheads nobody came for (a skipped batch)
                self.k.release(self._prepared.pop(next(iter(self._prepared))))
            prep = self.k.prepare(users_local, pos, neg, self.Ug, self.Ug + self.I, self.d)
            if prep is not None:
                self._prepared[_batch_key(users_local, pos, neg)] = prep


def _batch_key(users, pos, neg):
    ptr = (lambda t: t.data_ptr()) if hasattr(users, "data_ptr") else (lambda t: t.__array_interface__["data"][0])
    return (ptr(users), ptr(pos), ptr(neg), len(users))


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

    class _Prepared:
        __slots__ = ("bitmap", "ws", "rows_done", "done", "free", "B", "busy")

    def prepare(self, users, pos, neg, n_users, n, d):
        """Index-only work of a batch on a side stream: bitmap of the panel rows it touches, and the sorted
        scatter plan.  Returns None when the scatter is not the deterministic one.  Host cost matters here (the
        sharded step issues ~50 calls): raw stream handles and events allocated once, no stream context manager."""
        if not self.deterministic:
            return None
        torch, ops = self.torch, self.ops
        B = users.shape[0]
        prep = next((p for p in self._pool if p.B == B and not p.busy), None)
        if prep is None:
            prep = self._Prepared()
            prep.bitmap = torch.zeros((n + 31) // 32, dtype=torch.int32, device=self.device)
            prep.ws, prep.B = ops.bpr_workspace(B, d, self.device), B
            prep.rows_done, prep.done, prep.free = ops.LocalEvent(), ops.LocalEvent(), None  # device-local events
            self._pool.append(prep)
        prep.busy = True
        main = torch.cuda.current_stream()
        self._fork.record(main.cuda_stream)  # the id tensors may have just been produced on the main stream,
        self._fork.wait(self._side_raw)      # and the step that last used these buffers is ordered before it
        ops.bpr_touch_rows_raw(users, pos, neg, n_users, prep.bitmap, stream=self._side_raw, clear_bits=n)
        prep.rows_done.record(self._side_raw)
        ops.bpr_plan_raw(users, pos, neg, n_users, n, d, ws=prep.ws, stream=self._side_raw)
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
        self._ring, self._next = [], 0     # (issued, done) event pairs, reused round-robin (<= 2 collectives are in flight)

    def _fork(self):
        """Order the communicator's stream after the current one; returns (raw stream handle, event to record when done)."""
        torch = self.torch
        if self._own is None:
            self._own = torch.cuda.Stream()
            self._own_raw = self._own.cuda_stream
            self._ring = [(torch.cuda.Event(), torch.cuda.Event()) for _ in range(8)]
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

    def all_reduce_async(self, t, average=False):
        return None

    def all_gather_async(self, out, t):
        out[...] = t
        return None

    def wait(self, work):
        pass


# --------------------------------------------------------------------------- bench driver
def run_sharded_bench(args, rank, world, dist):
    """bench.py --gpus N (N > 1): weak scaling — every rank owns a block of the BASELINE-shape
    user set (U = N x the named shape's users, items fixed), B triples of its own users per
    step; value = N*B*steps / max-over-ranks time."""
    import json
    import time

    import torch

    from . import host as H
    from . import synth as S

    U1, I, E1 = S.SHAPES[args.workload]
    # weak scaling in the graph: one named-shape user block per rank — unless that would exceed any sensible host
    # budget (BASELINE config 5 is ONE 10M-user graph cut across the ranks, not eight of them)
    blocks = world if U1 * world <= 20_000_000 else 1
    U, E = U1 * blocks, E1 * blocks
    users, items = S.generate(U, I, E, seed=0)           # every rank derives the same global graph
    ip, ix, dv = H.build_norm_adj(U, I, users, items)
    deg_u = np.bincount(users, minlength=U)
    bounds = partition_users_by_nnz(deg_u, world)
    lo, hi = int(bounds[rank]), int(bounds[rank + 1])
    ui, iu = shard_adjacency(ip, ix, dv, U, I, lo, hi)
    kern = HipKernels(deterministic=not args.atomic)
    comm, comm_name = make_comm(dist, getattr(args, "comm", "auto"))
    eng = ShardedEngine(kern, comm, ui, iu, hi - lo, I, args.dim, args.layers, True, 1e-4, 1e-3)
    W0 = S.xavier_uniform_panel(U, I, args.dim, args.seed)  # same initialisation as the single-GPU run
    eng.P[: hi - lo].copy_(W0[lo:hi])
    eng.P[hi - lo:].copy_(W0[U:])
    # this rank's triples: the native sampler over its own users' edges
    sel = (users >= lo) & (users < hi)
    lu, li = users[sel] - lo, items[sel]
    pos_ptr = np.zeros(hi - lo + 1, dtype=np.int64)
    pos_ptr[1:] = np.cumsum(np.bincount(lu, minlength=hi - lo))
    rng = H.Rng(args.seed + rank)
    B = args.batch
    need = (args.steps + args.warmup) * B
    tri = np.empty((0, 3), dtype=np.int64)
    while len(tri) < need:
        t2 = rng.sample_epoch(lu, li, pos_ptr, li.astype(np.int32), I)
        tri = np.concatenate([tri, t2[rng.shuffle_perm(len(t2))]])
    tri = torch.from_numpy(tri).cuda()
    tu, tp, tn = tri[:, 0].contiguous(), tri[:, 1].contiguous(), tri[:, 2].contiguous()
    gB = B * world

    last = args.warmup + args.steps - 1

    def step(i):
        if i < last:
            n = slice((i + 1) * B, (i + 2) * B)
            eng.prefetch(tu[n], tp[n], tn[n])  # index-only work of the next batch, off the critical path
        s = slice(i * B, (i + 1) * B)
        return eng.train_step(tu[s], tp[s], tn[s], gB)

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
    if rank == 0:
        n, nnz = U + I, len(ix)
        out = {
            "metric": "BPR triples/sec, LightGCN-%d dim=%d" % (args.layers, args.dim),
            "value": gB * args.steps / dt, "unit": "triples/s", "n_gpus": world, "steps": args.steps,
            "warmup": args.warmup, "ms_per_step": dt / args.steps * 1e3, "higher_is_better": True, "scaling": "weak",
            "vs_baseline": None, "dtype": "f32", "data": "synthetic",
            "config": {"workload": "%d x %s-shape user blocks (graph cut across the ranks by user rows): %d users x %d items, %d train edges, nnz(A)=%d; "
                                   "LightGCN K=%d d=%d, B=%d per GPU (global %d); user rows sharded, item table "
                                   "replicated, %d all-reduces of [%d,%d] fp32 per step over %s"
                                   % (blocks, args.workload, U, I, len(users), nnz, args.layers, args.dim, B, gB,
                                      2 * args.layers + 1, I, args.dim,
                                      "RCCL" if dist.get_backend() == "nccl" else dist.get_backend() + " (rehearsal, host-staged)"),
                       "batch": B, "dim": args.dim, "layers": args.layers, "parallelism": "user-row shard x%d" % world,
                       "comm": comm_name},
            "loss_last": [float(x) for x in eng.loss.cpu()],
            "host_issue_ms_per_step": t_enqueue / args.steps * 1e3,
        }
        getattr(args, "emit", lambda o: print(json.dumps(o)))(out)
    dist.destroy_process_group()
