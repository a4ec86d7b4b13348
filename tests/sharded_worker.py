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

    def prepare(self, users, pos, neg, n_users, n, d):
        return None  # no batch lookahead in the checker-backed stub: every product is the dense one

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
    eng = sh.ShardedEngine(kern, sh.TorchComm(dist), ui, iu, hi - lo, I, W0.shape[1], K, bool(z["include0"]), 1e-4, 1e-3,
                           batch_sparsity=(mode != "gpu-dense"))
    if mode == "cpu":
        eng.P[: hi - lo] = W0[lo:hi]
        eng.P[hi - lo:] = W0[U:]
    else:
        eng.P[: hi - lo].copy_(to_dev(W0[lo:hi]))
        eng.P[hi - lo:].copy_(to_dev(W0[U:]))
    losses = []
    def batch(s):
        b = tri[s * B:(s + 1) * B]
        mine = b[(b[:, 0] >= lo) & (b[:, 0] < hi)]
        return to_dev(mine[:, 0] - lo), to_dev(mine[:, 1]), to_dev(mine[:, 2]), mine

    nxt = batch(0)
    for s in range(steps):
        cur, nxt = nxt, (batch(s + 1) if s + 1 < steps else None)
        if nxt is not None and s % 2 == 0:  # every other step through the lookahead, the rest prepared in-step
            eng.prefetch(*nxt[:3])
        loss = eng.train_step(cur[0], cur[1], cur[2], B)
        losses.append(to_np(loss).copy())
    touched = np.unique(cur[3][:, 0] - lo)  # local user rows of the LAST batch: the only FIN user rows guaranteed fresh
    np.savez(path + ".out%d.npz" % rank, P=to_np(eng.P), FIN=to_np(eng.FIN), G=to_np(eng.G), losses=np.stack(losses),
             lo=lo, hi=hi, fin_rows=touched)
    dist.barrier()
    dist.destroy_process_group()


if __name__ == "__main__":
    run(int(sys.argv[1]), int(sys.argv[2]), int(sys.argv[3]), sys.argv[4], sys.argv[5], int(sys.argv[6]))
