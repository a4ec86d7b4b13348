"""Worker for the world_size-2 tests of id-grec_amd/sharded.py (launched by
tests/test_sharded.py through torch.multiprocessing).  mode "cpu": the layer loop runs on
numpy arrays with a checker-backed kernel stub (arithmetic by oracle/, TEST ONLY) over gloo;
mode "gpu": the real HIP kernels, both ranks on cuda:0, gloo staging through the host."""
import os
import sys

import numpy as np

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
if ROOT not in sys.path:
    sys.path.insert(0, ROOT)


class OracleKernels:
    """`kernels` interface of ShardedEngine on numpy + oracle/ (tests only)."""

    def __init__(self):
        from oracle import oracle

        self.o = oracle

    def zeros(self, shape):
        return np.zeros(shape, dtype=np.float32)

    def fill(self, a, v):
        a[...] = v

    def make_graph(self, indptr, indices, values, n_rows, n_cols):
        return (np.asarray(indptr), np.asarray(indices), np.asarray(values), n_rows, n_cols)

    def prepare(self, eng, gb):
        return None  # no batch lookahead in the checker-backed stub: every product is the dense one

    def to_device(self, a):
        return np.ascontiguousarray(a)

    def gather_rows(self, dst, src, idx):
        dst[...] = np.where((idx >= 0)[:, None], src[np.maximum(idx, 0)], np.float32(0))

    def scatter_rows(self, dst, idx, src):
        dst[idx] = src

    def flag_touched_items(self, eng, prep, gb, flags):
        ptr, idx = eng.G_ui[0], eng.G_ui[1]
        flags[...] = 0
        for u in np.unique(gb.own_users):
            flags[idx[ptr[u]:ptr[u + 1]]] = 1
        flags[gb.items] = 1
        return True

    def item_rows_bitmap(self, eng, prep, rows, which=0):
        return None  # (the checker-backed stub produces every row)

    def touched_bitmap_local(self, eng, prep, gb):
        return None

    def flag_two_hop_items(self, eng, prep, gb, flags):
        users = set(np.unique(gb.own_users).tolist())  # near users: the batch's, and those of the batch's items
        for g, r0, r1 in eng.G_iu:
            ptr, idx = g[0], g[1]
            for i in gb.items[(gb.items >= r0) & (gb.items < r1)] - r0:
                users.update(idx[ptr[i]:ptr[i + 1]].tolist())
        ptr, idx = eng.G_ui[0], eng.G_ui[1]
        flags[...] = 0
        for u in users:
            flags[idx[ptr[u]:ptr[u + 1]]] = 1
        flags[gb.items] = 1
        return True

    def user_rows_bitmaps(self, eng, prep, gb, touched_bits):
        return None, None

    def nonzero_ids(self, flags):
        ids = np.nonzero(flags)[0].astype(np.int64)
        return ids, len(ids)

    def chain_add_rows(self, dst, src, idx, nxt):
        for t in np.nonzero(idx >= 0)[0]:
            acc, j = dst[idx[t]].copy(), t
            while j >= 0:
                acc = acc + src[j]
                j = nxt[j]
            dst[idx[t]] = acc

    def topk(self, user_panel, item_panel, users, k, excl_indptr, excl_items):
        R = self.o.score(user_panel, item_panel, np.asarray(users))
        for b, u in enumerate(users):
            R[b, excl_items[excl_indptr[u]:excl_indptr[u + 1]]] = -1
        return self.o.topk_reference(R, k)

    def spmm(self, graph, X, Y=None, addend=None, sum_in=None, sum_out=None, div=1.0, accumulate=False, out_rows=None,
             x_rows=None):
        assert out_rows is None and x_rows is None
        t = self.o.spmm(graph[0], graph[1], graph[2], X)
        if addend is not None:
            t = t + addend
        if Y is not None:
            Y[...] = t
        if sum_out is not None:
            s = t if sum_in is None else sum_in + t
            if div != 1.0:
                s = s / np.float32(div)
            sum_out[...] = sum_out + s if accumulate else s

    def lincomb(self, out, x, a, y, b):
        r = np.float32(a) * x
        if y is not None:
            r = r + np.float32(b) * y
        out[...] = r

    def bpr(self, fin, ego, n_users, users, pos, neg, reg_lambda, upstream, g_final, g_ego, loss, prep=None):
        l, gf, ge = self.o.bpr(fin, ego, n_users, users, pos, neg, reg_lambda)
        loss[...] = l
        g_final += upstream[0] * gf
        g_ego += upstream[1] * ge

    def adam(self, p, g, m, v, lr, step):
        self.o.adam(p, np.ascontiguousarray(g), m, v, lr, step)


def run(rank, world, port, mode, path, steps):
    import torch
    import torch.distributed as dist

    os.environ.update(MASTER_ADDR="127.0.0.1", MASTER_PORT=str(port), HSA_ENABLE_IPC_MODE_LEGACY="0")
    dist.init_process_group("gloo", rank=rank, world_size=world)
    import idgrec_amd.sharded as sh

    z = np.load(path)
    ip, ix, dv, W0, tri = z["indptr"], z["indices"], z["values"], z["W0"], z["triples"]
    U, I, K, B = int(z["U"]), int(z["I"]), int(z["K"]), int(z["B"])
    deg = np.diff(ip[: U + 1])
    bounds = sh.partition_users_by_nnz(deg, world)
    lo, hi = int(bounds[rank]), int(bounds[rank + 1])
    ui, iu = sh.shard_adjacency(ip, ix, dv, U, I, lo, hi)
    if mode == "cpu":
        kern, to_dev, to_np = OracleKernels(), (lambda a: np.ascontiguousarray(a)), (lambda a: a)
    else:
        torch.cuda.set_device(0)
        kern = sh.HipKernels()
        to_dev = lambda a: torch.from_numpy(np.ascontiguousarray(a)).cuda()  # noqa: E731
        to_np = lambda a: a.cpu().numpy()  # noqa: E731
    n_slices = int(z["n_slices"]) if "n_slices" in z.files else 1
    eng = sh.ShardedEngine(kern, sh.TorchComm(dist), ui, iu, hi - lo, I, W0.shape[1], K, bool(z["include0"]), 1e-4, 1e-3,
                           batch_sparsity=(mode != "gpu-dense"), batch_size=B, user_lo=lo, n_slices=n_slices,
                           live_rows_cap=int(z["live_cap"]) if "live_cap" in z.files else None, live_rows_min_bytes=0,
                           two_hop_cap=int(z["two_cap"]) if "two_cap" in z.files else None)
    Ug = hi - lo
    if mode == "cpu":
        eng.P[:Ug] = W0[lo:hi]
        eng.P[Ug + B:] = W0[U:]
    else:
        eng.P[:Ug].copy_(to_dev(W0[lo:hi]))
        eng.P[Ug + B:].copy_(to_dev(W0[U:]))
    losses = []

    def batch(s):
        b = tri[s * B:(s + 1) * B]  # the GLOBAL batch: every rank sees all of it
        return eng.make_batch(b[:, 0], b[:, 1], b[:, 2]), b

    nxt = batch(0)
    for s in range(steps):
        cur, nxt = nxt, (batch(s + 1) if s + 1 < steps else None)
        if nxt is not None and s % 2 == 0:  # every other step through the lookahead, the rest prepared in-step
            eng.prefetch(nxt[0])
        loss = eng.train_step(cur[0])
        losses.append(to_np(loss).copy())
    mine = cur[1][(cur[1][:, 0] >= lo) & (cur[1][:, 0] < hi)]
    touched = np.unique(mine[:, 0] - lo)  # local user rows of the LAST batch: the only FIN user rows guaranteed fresh

    def strip(a):  # (drop the guest rows: users, then items)
        a = to_np(a)
        return np.concatenate([a[:Ug], a[Ug + B:]])

    # item rows of FIN that the LAST step produced: the batch's positive / negative items (the last forward layer's
    # exchange carries those rows only)
    fin_items = np.unique(np.concatenate([cur[1][:, 1], cur[1][:, 2]]))
    out = dict(P=strip(eng.P), FIN=strip(eng.FIN), G=strip(eng.G), losses=np.stack(losses), lo=lo, hi=hi, fin_rows=touched,
               fin_items=fin_items,
               touched_n=-1 if getattr(eng, "touched_items", None) is None else eng.touched_items[1],
               two_hop_n=-1 if getattr(eng, "two_hop", None) is None else eng.two_hop[1],
               two_hop_misses=eng._two_hop_misses)
    if "test_users" in z.files:  # sharded evaluation: this rank's test users, its train rows as the exclusion lists
        tu, tptr, titems = z["test_users"], z["test_ptr"], z["test_items"]
        own = [(int(u), titems[tptr[j]:tptr[j + 1]].tolist()) for j, u in enumerate(tu) if lo <= u < hi]
        ex_ptr = z["train_ptr"][lo:hi + 1] - z["train_ptr"][lo]
        ex_items = z["train_items"][z["train_ptr"][lo]:z["train_ptr"][hi]]

        def reduce_sums(a):
            t = torch.from_numpy(a)
            dist.all_reduce(t)
            return t.numpy()

        res = eng.evaluate([u for u, _ in own], [t for _, t in own], ex_ptr, ex_items, [5, 10], reduce_sums)
        out.update(ev_recall=res["recall"], ev_precision=res["precision"], ev_ndcg=res["ndcg"])
    np.savez(path + ".out%d.npz" % rank, **out)
    dist.barrier()
    dist.destroy_process_group()


if __name__ == "__main__":
    run(int(sys.argv[1]), int(sys.argv[2]), int(sys.argv[3]), sys.argv[4], sys.argv[5], int(sys.argv[6]))
