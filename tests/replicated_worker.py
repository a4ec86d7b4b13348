"""Worker for the world_size-2 tests of id-grec_amd/replicated.py (launched by tests/test_replicated.py).
mode "cpu": a checker-backed engine (arithmetic by oracle/, TEST ONLY) over gloo; mode "gpu": the HIP
engine, both ranks on cuda:0, gloo staging through the host."""
import os
import sys

import numpy as np

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
if ROOT not in sys.path:
    sys.path.insert(0, ROOT)


class OracleReplica:
    """`engine` interface of ReplicatedStep on numpy + oracle/ (tests only)."""

    def __init__(self, adj, W0, U, K, include0):
        from oracle import oracle

        self.o, self.adj, self.U, self.K, self.inc = oracle, adj, U, K, include0
        n, d = W0.shape
        self.params = W0.copy()
        self.grad_and_loss = np.zeros((n + 1, d), dtype=np.float32)
        self.grad = self.grad_and_loss[:n]
        self.loss_row = self.grad_and_loss[n, :2]
        self.m, self.v, self.step = np.zeros_like(W0), np.zeros_like(W0), 0
        self.world = 1

    def loss_and_grad(self, users, pos, neg):
        fin = self.o.propagate_mean(*self.adj, self.params, self.K, self.inc)
        loss, gf, ge = self.o.bpr(fin, self.params, self.U, users, pos, neg, 1e-4)
        self.grad[...] = self.o.propagate_mean_bwd(*self.adj, gf, self.K, self.inc) + ge
        self.loss_row[...] = loss

    def train_step_rows(self, users, pos, neg, gather):
        """RowExchangeStep's engine protocol: this rank's gradient rows as a flat fp32 message, all-gathered, merged in
        rank order, then ONE backward propagation of the averaged rows."""
        n, d = self.params.shape
        B = len(users)
        fin = self.o.propagate_mean(*self.adj, self.params, self.K, self.inc)
        loss, gf, _ = self.o.bpr(fin, self.params, self.U, users, pos, neg, 1e-4)
        rows, cnt = np.unique(np.concatenate([users, self.U + pos, self.U + neg]), return_counts=True)
        L = 2 + 6 * B + 3 * B * d
        msg = np.zeros(L, dtype=np.float32)
        msg[:2] = loss
        ids = np.full(3 * B, -1, dtype=np.int32)
        ids[:len(rows)] = rows
        c = np.zeros(3 * B, dtype=np.int32)
        c[:len(rows)] = cnt
        msg[2:2 + 3 * B] = ids.view(np.float32)
        msg[2 + 3 * B:2 + 6 * B] = c.view(np.float32)
        msg[2 + 6 * B:].reshape(3 * B, d)[:len(rows)] = gf[rows]
        out = np.zeros(L * self.world, dtype=np.float32)
        gather(msg, out)
        scale = np.float32(1.0 / self.world)
        gfin, gego = np.zeros((n, d), dtype=np.float32), np.zeros((n, d), dtype=np.float32)
        self.loss_row[...] = 0
        for m in out.reshape(self.world, L):
            self.loss_row += m[:2] * scale
            ids = m[2:2 + 3 * B].view(np.int32)
            live = ids >= 0
            r = ids[live]
            gfin[r] += m[2 + 6 * B:].reshape(3 * B, d)[live] * scale
            gego[r] += (np.float32(1e-4 / B) * scale) * m[2 + 3 * B:2 + 6 * B].view(np.int32)[live, None].astype(np.float32) * self.params[r]
        self.grad[...] = self.o.propagate_mean_bwd(*self.adj, gfin, self.K, self.inc) + gego
        self.adam_step()
        return self.loss_row

    def scale_grad_and_loss(self, a):
        self.grad_and_loss *= np.float32(a)

    def adam_step(self):
        self.step += 1
        self.o.adam(self.params, np.ascontiguousarray(self.grad), self.m, self.v, 1e-3, self.step)


def run(rank, world, port, mode, path, steps, exchange="grad"):
    import torch
    import torch.distributed as dist

    os.environ.update(MASTER_ADDR="127.0.0.1", MASTER_PORT=str(port), HSA_ENABLE_IPC_MODE_LEGACY="0")
    dist.init_process_group("gloo", rank=rank, world_size=world)
    import idgrec_amd.replicated as rp
    import idgrec_amd.sharded as sh

    z = np.load(path)
    ip, ix, dv, W0, tri = z["indptr"], z["indices"], z["values"], z["W0"], z["triples"]
    U, I, K, B = int(z["U"]), int(z["I"]), int(z["K"]), int(z["B"])  # B: per-rank batch; global batch = world * B
    if mode == "cpu":
        eng, to_dev, to_np = OracleReplica((ip, ix, dv), W0, U, K, bool(z["include0"])), (lambda a: np.ascontiguousarray(a)), (lambda a: a)
        eng.world = world
    else:
        import idgrec_amd.ops as ops

        torch.cuda.set_device(0)
        n = U + I
        g = ops.Graph(ip, ix, dv, n, n)
        eng = rp.HipReplica(g, U, I, W0.shape[1], K, bool(z["include0"]), 1e-4, 1e-3,
                            params=torch.from_numpy(W0.copy()).cuda(), world=world)
        to_dev = lambda a: torch.from_numpy(np.ascontiguousarray(a)).cuda()  # noqa: E731
        to_np = lambda a: a.cpu().numpy()  # noqa: E731
    comm, comm_name = sh.make_comm(dist, "auto")  # gloo: the torch.distributed form (host-staged for device tensors)
    assert isinstance(comm, sh.TorchComm) and comm_name == "torch.distributed"
    step = (rp.RowExchangeStep if exchange == "rows" else rp.ReplicatedStep)(eng, comm, world)
    losses = []
    for s in range(steps):
        b = tri[s * world * B:(s + 1) * world * B][rank * B:(rank + 1) * B]
        loss = step.train_step(to_dev(b[:, 0]), to_dev(b[:, 1]), to_dev(b[:, 2]))
        losses.append(to_np(loss).copy())
    np.savez(path + ".out%d.npz" % rank, P=to_np(eng.params), G=to_np(eng.grad), losses=np.stack(losses))
    dist.barrier()
    dist.destroy_process_group()


if __name__ == "__main__":
    run(int(sys.argv[1]), int(sys.argv[2]), int(sys.argv[3]), sys.argv[4], sys.argv[5], int(sys.argv[6]),
        sys.argv[7] if len(sys.argv) > 7 else "grad")
