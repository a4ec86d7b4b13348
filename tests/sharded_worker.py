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
    """`kernels` interface of ShardedEngine on numpy + oracle/ (tests only).  Row bitmaps are numpy bool arrays.  Every
    restricted form is honoured AND made unforgiving: a product restricted to some output rows writes NaN into the rows it
    does not produce, the gradient scatter leaves NaN in every row it does not store, and rows of an input panel outside
    its live set are never read — so a consumer that reads a row nobody produced turns the loss and the tables into NaN."""

    def __init__(self):
        from oracle import oracle

        self.o = oracle

    def zeros(self, shape):
        return np.zeros(shape, dtype=np.float32)

    def fill(self, a, v):
        a[...] = v

    def make_graph(self, indptr, indices, values, n_rows, n_cols):
        return (np.asarray(indptr), np.asarray(indices), np.asarray(values), n_rows, n_cols)

    class _Prep:
        pass

    def prepare(self, eng, gb):
        prep = self._Prep()
        prep.own_users = np.zeros(eng.Ug, dtype=bool)
        prep.own_users[gb.own_users] = True
        prep.items = np.zeros(eng.Ip, dtype=bool)
        prep.items[gb.items] = True
        prep.touched = prep.near = None
        return prep

    def wait_rows(self, prep):
        pass

    def release(self, prep):
        pass

    def bits_from(self, bits, row0):
        return bits[row0:]

    def ones_bits(self, n_bits):
        return np.ones(n_bits, dtype=bool)

    def to_device(self, a):
        return np.ascontiguousarray(a)

    def gather_rows(self, dst, src, idx):
        dst[...] = np.where((idx >= 0)[:, None], src[np.maximum(idx, 0)], np.float32(0))

    def gather_rows2(self, dst0, src0, dst1, src1, idx):
        self.gather_rows(dst0, src0, idx)
        self.gather_rows(dst1, src1, idx)

    def scatter_rows(self, dst, idx, src):
        # (a padded id list repeats its last id: every copy of that row must carry the same value)
        rep = idx == idx[-1]
        assert np.array_equal(src[rep], np.broadcast_to(src[rep][0], src[rep].shape), equal_nan=True)
        dst[idx] = src

    def _adjacent_items(self, eng, user_bits):
        ptr, idx = eng.G_ui[0], eng.G_ui[1]
        out = np.zeros(eng.Ip, dtype=bool)
        for u in np.nonzero(user_bits)[0]:
            out[idx[ptr[u]:ptr[u + 1]]] = True
        return out

    def _near(self, eng, prep):
        near = prep.own_users.copy()
        for g, r0, r1, r1p in eng.slices:
            ptr, idx = g[0], g[1]
            for i in np.nonzero(prep.items[r0:r1])[0]:
                near[idx[ptr[i]:ptr[i + 1]]] = True
        prep.near = near

    def touched_local(self, eng, prep, gb):
        prep.touched = prep.items | self._adjacent_items(eng, prep.own_users)
        self._near(eng, prep)

    def flag_touched_items(self, eng, prep, gb, flags):
        flags[...] = (prep.items | self._adjacent_items(eng, prep.own_users))[: eng.I]

    def nonzero_ids(self, flags):
        ids = np.nonzero(flags)[0].astype(np.int64)
        return ids, len(ids)

    def compact_ids(self, flags, n):
        """The device list of the product (idg_flags_compact_f32): n slots, ascending ids, the tail repeating the last."""
        ids = np.nonzero(flags)[0].astype(np.int64)
        assert 0 < len(ids) <= n, "the host-side bound must cover the touched items (%d > %d)" % (len(ids), n)
        return np.concatenate([ids, np.full(n - len(ids), ids[-1], dtype=np.int64)])

    def touched_from_ids(self, eng, prep, ids, n):
        prep.touched = np.zeros(eng.Ip, dtype=bool)
        prep.touched[ids[:n]] = True
        self._near(eng, prep)

    def chain_rows2(self, dst0, src0, dst1, src1, idx, nxt, store):
        for dst, src in ((dst0, src0), (dst1, src1)):
            for t in np.nonzero(idx >= 0)[0]:
                acc, j = (src[t].copy(), nxt[t]) if store else (dst[idx[t]].copy(), t)
                while j >= 0:
                    acc = acc + src[j]
                    j = nxt[j]
                dst[idx[t]] = acc

    def layer_mean(self, out, ids, terms, last, div):
        s = last
        if terms:
            acc = terms[0][ids]
            for x in terms[1:]:
                acc = acc + x[ids]
            s = acc + last
        out[ids] = s / np.float32(div)

    def topk(self, user_panel, item_panel, users, k, excl_indptr, excl_items):
        R = self.o.score(user_panel, np.ascontiguousarray(item_panel), np.asarray(users))
        for b, u in enumerate(users):
            R[b, excl_items[excl_indptr[u]:excl_indptr[u + 1]]] = -1
        return self.o.topk_reference(R, k)

    def spmm(self, graph, X, Y=None, addend=None, sums=(), sum_out=None, div=1.0, accumulate=False, mask=None, adam=None,
             out_rows=None, x_rows=None, discard_grad=False, Y24=None):
        ptr, idx, val, n_rows, n_cols = graph
        Xe = np.ascontiguousarray(X[:n_cols])
        if x_rows is not None:  # rows outside the live set are zero by agreement and must not be read
            Xe = np.where(x_rows[:n_cols, None], Xe, np.float32(0))
        t = self.o.spmm(ptr, idx, val, np.ascontiguousarray(Xe))
        live = np.ones(n_rows, dtype=bool) if mask is None else mask[:n_rows]
        rows = np.ones(n_rows, dtype=bool) if out_rows is None else out_rows[:n_rows]
        with np.errstate(invalid="ignore"):
            if addend is not None:
                t = np.where(live[:, None], t + addend[:n_rows], t)
            if Y is not None:
                Y[:n_rows][rows] = t[rows]
                Y[:n_rows][~rows] = np.nan
            if Y24 is not None:  # the finished rows as 24-bit values (the packed exchange's send buffer), dense launches only
                assert out_rows is None
                Y24.view(np.uint32)[: t.size // 4 * 3] = self.o.pack24(t.reshape(-1))
            if sum_out is not None:
                s = t
                if len(sums):
                    acc = sums[0][:n_rows]
                    for x in sums[1:]:
                        acc = acc + x[:n_rows]
                    s = np.where(live[:, None], acc + t, t)
                if div != 1.0:
                    s = s / np.float32(div)
                if accumulate:
                    s = np.where(live[:, None], sum_out[:n_rows] + s, s)
                sum_out[:n_rows][rows] = s[rows]
                sum_out[:n_rows][~rows] = np.nan
                if adam is not None:
                    assert out_rows is None and x_rows is None
                    p, m, v, lr, step = adam
                    self.o.adam(p, np.ascontiguousarray(s), m, v, lr, step)

    def bpr(self, fin, ego, n_users, users, pos, neg, reg_lambda, g_final, g_ego, loss, prep=None):
        reached = np.unique(np.concatenate([users, n_users + pos, n_users + neg]))
        f, e = np.zeros_like(fin), np.zeros_like(ego)  # (the rows the loss does not read may hold anything)
        f[reached], e[reached] = fin[reached], ego[reached]
        l, gf, ge = self.o.bpr(f, e, n_users, users, pos, neg, reg_lambda)
        loss[...] = l
        if prep is None:
            g_final += gf
            g_ego += ge
        else:  # the reached rows are STORED; nothing else is defined
            g_final[...] = np.nan
            g_ego[...] = np.nan
            g_final[reached] = gf[reached]
            g_ego[reached] = ge[reached]

    def item_tail(self, t, g, G, live_bits, row0, c0, cnt, store_grad, p, m, v, lr, step):
        rows = t.shape[0]
        live = live_bits[row0:row0 + rows][:, None]
        with np.errstate(invalid="ignore"):
            s = np.where(live, g + t, t) if c0 else t
            s = s / np.float32(cnt)
            s = np.where(live, G + s, s)
        if store_grad:
            G[...] = s
        self.o.adam(p, np.ascontiguousarray(s), m, v, lr, step)

    def adam(self, p, g, m, v, lr, step):
        self.o.adam(p, np.ascontiguousarray(g), m, v, lr, step)

    # ---- 24-bit panels (sharded.Packed24Comm): packed buffers are fp32-typed arrays holding the words
    def pack24(self, src, dst, n):
        dst.view(np.uint32)[: n // 4 * 3] = self.o.pack24(src.reshape(-1)[:n])

    def unpack24(self, src, dst, n):
        dst.reshape(-1)[:n] = self.o.unpack24(src.view(np.uint32)[: n // 4 * 3])

    def reduce24(self, blocks, n_blocks, n, out_packed=None, out_f32=None):
        w = n // 4 * 3
        words = blocks.view(np.uint32)
        acc = self.o.reduce24([words[b * w:(b + 1) * w] for b in range(n_blocks)])
        if out_f32 is not None:
            out_f32.reshape(-1)[:n] = acc
        if out_packed is not None:
            out_packed.view(np.uint32)[:w] = self.o.pack24(acc)

    def reduce_blocks(self, blocks, n_blocks, n, out):
        acc = blocks[:n].copy()
        for b in range(1, n_blocks):
            acc = (acc + blocks[b * n:(b + 1) * n]).astype(np.float32)
        out.reshape(-1)[:n] = acc

    def exchange_stream(self):
        import contextlib

        return contextlib.nullcontext()

    def record_event(self):
        return None

    def wait_event(self, ev):
        pass


class DeferredComm:
    """TorchComm whose asynchronous collectives do NOTHING until wait(): the engine sees un-reduced partials if it reads a
    buffer before waiting for its collective, and the wrong operands are reduced if it rewrites a buffer that is still
    in flight — either shows against the single-device oracle.  (gloo itself completes every collective inside the call,
    which hides exactly those mistakes.)  Collectives still run in the same order on every rank: waits happen in program
    order."""

    def __init__(self, inner):
        self.inner, self.world, self.rank = inner, inner.world, inner.rank
        self.averages = inner.averages

    def all_reduce_async(self, t, average=False):
        return ("ar", t, average)

    def all_gather_async(self, out, t):
        return ("ag", out, t)

    def reduce_scatter_async(self, t):
        return ("rs", t)

    def all_to_all_async(self, recv, send):
        return ("a2a", recv, send)

    def wait(self, work):
        if work is None:
            return
        kind = work[0]
        if kind == "ar":
            self.inner.wait(self.inner.all_reduce_async(work[1], work[2]))
        elif kind == "ag":
            self.inner.wait(self.inner.all_gather_async(work[1], work[2]))
        elif kind == "a2a":
            self.inner.wait(self.inner.all_to_all_async(work[1], work[2]))
        else:
            self.inner.wait(self.inner.reduce_scatter_async(work[1]))


class SideStreamComm:
    """Device tensors over gloo with the reduction done on a SIDE stream that is ordered after the step's stream when the
    collective is issued and that the step's stream only joins in wait() — the ordering contract of NativeComm's
    second-stream route and of c10d work objects, on one GPU shared by the ranks.  A missing wait() lets the products that
    follow race with the copy back."""

    def __init__(self, dist):
        import torch

        self.dist, self.torch = dist, torch
        self.world, self.rank, self.averages = dist.get_world_size(), dist.get_rank(), False
        self.side = torch.cuda.Stream()

    def _run(self, fn, *tensors):
        torch = self.torch
        issued = torch.cuda.Event()
        issued.record()
        self.side.wait_event(issued)
        with torch.cuda.stream(self.side):
            fn()
            done = torch.cuda.Event()
            done.record()
        return done

    def all_reduce_async(self, t, average=False):
        def fn():
            host = t.cpu()
            self.dist.all_reduce(host)
            t.copy_(host, non_blocking=False)
        return self._run(fn)

    def all_gather_async(self, out, t):
        def fn():
            host = self.torch.empty(out.shape, dtype=out.dtype)
            self.dist.all_gather_into_tensor(host, t.cpu().contiguous())
            out.copy_(host)
        return self._run(fn)

    def reduce_scatter_async(self, t):
        return self.all_reduce_async(t)

    def all_to_all_async(self, recv, send):
        def fn():
            W = self.world
            s, r = send.cpu().view(W, -1), self.torch.empty(recv.shape, dtype=recv.dtype).view(W, -1)
            r[self.rank] = s[self.rank]
            reqs = []
            for p in range(W):
                if p != self.rank:
                    reqs.append(self.dist.isend(s[p].contiguous(), p))
                    reqs.append(self.dist.irecv(r[p], p))
            for q in reqs:
                q.wait()
            recv.view(W, -1).copy_(r)
        return self._run(fn)

    def wait(self, work):
        if work is not None:
            self.torch.cuda.current_stream().wait_event(work)


def run(rank, world, port, mode, path, steps):
    import torch
    import torch.distributed as dist

    os.environ.update(MASTER_ADDR="127.0.0.1", MASTER_PORT=str(port), HSA_ENABLE_IPC_MODE_LEGACY="0")
    packed = mode.endswith("+p24")  # the panel exchanges as 24-bit rows with the rank-ordered sum (sharded.Packed24Comm)
    ranked = mode.endswith("+r32")  # ... or as fp32 blocks with the rank-ordered sum (the same class, bits = 32)
    mode = mode[:-4] if (packed or ranked) else mode
    rccl = mode.startswith("nccl")  # "nccl-native" / "nccl-torch": one DEVICE per rank, RCCL between them (needs >= world GPUs)
    if rccl:
        torch.cuda.set_device(rank)
    dist.init_process_group("nccl" if rccl else "gloo", rank=rank, world_size=world)
    import idgrec_amd.sharded as sh

    z = np.load(path)
    ip, ix, dv, W0, tri = z["indptr"], z["indices"], z["values"], z["W0"], z["triples"]
    U, I, K, B = int(z["U"]), int(z["I"]), int(z["K"]), int(z["B"])
    deg = np.diff(ip[: U + 1])
    bounds = sh.partition_users_by_nnz(deg, world)
    lo, hi = int(bounds[rank]), int(bounds[rank + 1])
    ui, iu = sh.shard_adjacency(ip, ix, dv, U, I, lo, hi)
    if mode.startswith("cpu"):
        kern, to_dev, to_np = OracleKernels(), (lambda a: np.ascontiguousarray(a)), (lambda a: a)
    else:
        if not rccl:
            torch.cuda.set_device(0)
        kern = sh.HipKernels()
        to_dev = lambda a: torch.from_numpy(np.ascontiguousarray(a)).cuda()  # noqa: E731
        to_np = lambda a: a.cpu().numpy()  # noqa: E731
    n_slices = int(z["n_slices"]) if "n_slices" in z.files else 1
    comm = sh.TorchComm(dist) if world > 1 or mode.startswith("cpu") or mode == "gpu-timeline" else sh.NoComm()
    if mode == "cpu-deferred":
        comm = DeferredComm(comm)
    if mode == "gpu-async":
        comm = SideStreamComm(dist)
    if rccl:
        comm, comm_name = sh.make_comm(dist, mode.split("-", 1)[1])
        assert ("libidgrec" in comm_name) == (mode == "nccl-native"), comm_name
        if mode == "nccl-native":
            comm.overlap_bytes = 1 << 16  # the second-stream route (>= 64 MB in production) on this small problem too
            comm._force = True            # at world size 1 a collective is the identity and would not be enqueued: go through RCCL
    if packed or ranked:
        comm = sh.Packed24Comm(comm, kern, min_bytes=int(z["p24_min_bytes"]) if "p24_min_bytes" in z.files else 0,
                               bits=24 if packed else 32)
    eng = sh.ShardedEngine(kern, comm, ui, iu, hi - lo, I, W0.shape[1], K, bool(z["include0"]), 1e-4, 1e-3,
                           batch_sparsity=mode not in ("gpu-dense", "cpu-dense"), batch_size=B, user_lo=lo, n_slices=n_slices,
                           live_rows_cap=int(z["live_cap"]) if "live_cap" in z.files else None,
                           live_rows_min_bytes=int(z["min_bytes"]) if "min_bytes" in z.files else 0,
                           global_user_degree=deg if ("degree_bound" in z.files and int(z["degree_bound"])) else None)
    tl = None
    if mode == "gpu-timeline":
        tl = sh.StepTimeline(torch, world)
        eng.comm = sh.TimelineComm(eng.comm, tl)
        eng.timeline = tl
    order = sh.IssueOrder().attach(eng)  # the launch sequence of every step, checked against DESIGN.md §7's overlap model
    Ug = hi - lo
    if mode.startswith("cpu"):
        eng.P[:Ug] = W0[lo:hi]
        eng.item_rows(eng.P)[...] = W0[U:]
    else:
        eng.P[:Ug].copy_(to_dev(W0[lo:hi]))
        eng.item_rows(eng.P).copy_(to_dev(W0[U:]))
    losses = []

    def batch(s):
        b = tri[s * B:(s + 1) * B]  # the GLOBAL batch: every rank sees all of it
        return eng.make_batch(b[:, 0], b[:, 1], b[:, 2]), b

    nxt = batch(0)
    for s in range(steps):
        cur, nxt = nxt, (batch(s + 1) if s + 1 < steps else None)
        if nxt is not None and s % 2 == 0:  # every other step through the lookahead, the rest prepared in-step
            eng.prefetch(nxt[0])
        if tl is not None:
            ev = (tl.event(), tl.event())
            ev[0].record()
        order.begin_step()
        loss = eng.train_step(cur[0])
        if tl is not None:
            ev[1].record()
            tl.steps.append(ev)
        losses.append(to_np(loss).copy())
    mine = cur[1][(cur[1][:, 0] >= lo) & (cur[1][:, 0] < hi)]
    touched = np.unique(mine[:, 0] - lo)  # local user rows of the LAST batch: the only FIN user rows guaranteed fresh

    order.end_steps()
    eng._wait_item_table()  # the last step's all-gathers of the updated item rows

    def strip(a):  # (drop the guest rows and the padding: users, then items)
        a = to_np(a)
        return np.concatenate([a[:Ug], a[Ug + B: Ug + B + I]])

    own_items = np.concatenate([np.arange(o0, min(o0 + c, I)) for o0, c, _ in eng.own]).astype(np.int64)

    # item rows of FIN that the LAST step produced: the batch's positive / negative items (the last forward layer's
    # exchange carries those rows only)
    fin_items = np.unique(np.concatenate([cur[1][:, 1], cur[1][:, 2]]))
    out = dict(P=strip(eng.P), FIN=strip(eng.FIN), G=strip(eng.G), losses=np.stack(losses), lo=lo, hi=hi, fin_rows=touched,
               fin_items=fin_items, own_items=own_items,
               touched_n=-1 if getattr(eng, "touched_items", None) is None else eng.touched_items[1],
               n_slices=len(eng.slices), order_violations=np.array("\n".join(order.violations())),
               order_events=np.array(repr(order.steps()[-1])))
    if packed or ranked:
        import json

        out["packed_stats"] = np.array(json.dumps(comm.stats()))
    if packed:
        out["master_rows"] = to_np(eng.MP)
    if tl is not None:
        import json

        torch.cuda.synchronize()
        out["timeline"] = np.array(json.dumps(tl.summary()))
    if "test_users" in z.files:  # sharded evaluation: this rank's test users, its train rows as the exclusion lists
        tu, tptr, titems = z["test_users"], z["test_ptr"], z["test_items"]
        own = [(int(u), titems[tptr[j]:tptr[j + 1]].tolist()) for j, u in enumerate(tu) if lo <= u < hi]
        ex_ptr = z["train_ptr"][lo:hi + 1] - z["train_ptr"][lo]
        ex_items = z["train_items"][z["train_ptr"][lo]:z["train_ptr"][hi]]

        def reduce_sums(a):
            t = torch.from_numpy(a)
            if rccl:  # (the nccl backend moves device tensors only)
                t = t.cuda()
            dist.all_reduce(t)
            return t.cpu().numpy()

        res = eng.evaluate([u for u, _ in own], [t for _, t in own], ex_ptr, ex_items, [5, 10], reduce_sums)
        out.update(ev_recall=res["recall"], ev_precision=res["precision"], ev_ndcg=res["ndcg"])
    if rccl:
        # the item table, updated by its owners and all-gathered: a checksum must agree on every rank (what the bench
        # line's item_table_coherent reports)
        chk = eng.item_rows(eng.P).double().sum().reshape(1)
        lo_, hi_ = chk.clone(), chk.clone()
        dist.all_reduce(lo_, op=dist.ReduceOp.MIN)
        dist.all_reduce(hi_, op=dist.ReduceOp.MAX)
        out["coherent"] = bool(lo_.item() == hi_.item())
    np.savez(path + ".out%d.npz" % rank, **out)
    dist.barrier()
    if hasattr(comm, "close"):
        comm.close()
    dist.destroy_process_group()


if __name__ == "__main__":
    run(int(sys.argv[1]), int(sys.argv[2]), int(sys.argv[3]), sys.argv[4], sys.argv[5], int(sys.argv[6]))
