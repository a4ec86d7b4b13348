"""Data-parallel replicas of the fused training step: every rank holds the whole graph and both
embedding tables, a global batch of world x B triples is cut into per-rank slices, the per-rank
gradients [n, d] are averaged by ONE all-reduce per step (RCCL over xGMI), and every rank applies
the same Adam update — synchronous data-parallel SGD with the reference's loss at batch size
world x B (loss = mean over the global batch, trainer.py:42-56).

When to use which multi-GPU form (bench.py --parallel auto decides by the panel size):
  * replicas (this file): the graph is small enough that one GPU propagates it in tens of
    microseconds (BASELINE configs 1-4: panels of 18-37 MB).  Sharding such a graph by user rows
    needs 2K+1 all-reduces of the item panel per step, each longer than the product it follows;
    replicating it needs one all-reduce of the gradient.
  * user-row shards (sharded.py): the propagation itself is the cost (config 5: 70 ms per product
    on one GPU) and divides across ranks.

The engine is anything with `loss_and_grad(users, pos, neg) -> loss[2]`, `.grad` [n, d] and
`adam_step()`: the HIP engine in the product, a checker-backed stub in the CPU (gloo) tests.
"""
import numpy as np


class ReplicatedStep:
    def __init__(self, engine, comm, world):
        self.eng, self.comm, self.world = engine, comm, int(world)

    def train_step(self, users, pos, neg):
        """users/pos/neg: THIS rank's slice of the global batch.  Returns the global [bpr, reg] losses."""
        eng = self.eng
        eng.loss_and_grad(users, pos, neg)      # losses -> eng.loss_row (the row after the gradient panel)
        # mean over ranks == mean over the global batch (equal slices): RCCL averages inside the collective
        work = self.comm.all_reduce_async(eng.grad_and_loss, average=True)
        self.comm.wait(work)
        if not self.comm.averages and self.world > 1:
            eng.scale_grad_and_loss(1.0 / self.world)
        eng.adam_step()
        return eng.loss_row


class HipReplica:
    """PropagationEngine with its gradient panel extended by one row that carries the two losses, so that
    one all-reduce moves both."""

    def __init__(self, graph, num_users, num_items, dim, n_layers, include_layer0=True, reg_lambda=1e-4, lr=1e-3,
                 params=None):
        import torch

        from . import ops
        from .engine import PropagationEngine

        self.torch, self.ops = torch, ops
        self.eng = PropagationEngine(graph, num_users, num_items, dim, n_layers, include_layer0=include_layer0,
                                     reg_lambda=reg_lambda, lr=lr, params=params)
        n = num_users + num_items
        self.grad_and_loss = torch.zeros((n + 1, dim), dtype=torch.float32, device=self.eng.device)
        self.grad = self.eng.grad = self.grad_and_loss[:n]  # the engine writes its gradient straight into the buffer
        self.loss_row = self.grad_and_loss[n, :2]
        self.params = self.eng.params

    def prefetch(self, users, pos, neg):
        self.eng.prefetch(users, pos, neg)

    def loss_and_grad(self, users, pos, neg):
        return self.eng.loss_and_grad(users, pos, neg, loss_out=self.loss_row)

    def scale_grad_and_loss(self, a):
        self.ops.lincomb_raw(self.grad_and_loss, self.grad_and_loss, a)

    def adam_step(self):
        e = self.eng
        if e.exp_avg is None:
            e.exp_avg, e.exp_avg_sq = self.torch.zeros_like(e.params), self.torch.zeros_like(e.params)
        e.step_count += 1
        self.ops.adam_step_raw(e.params, e.grad, e.exp_avg, e.exp_avg_sq, e.lr, e.step_count, e.betas[0], e.betas[1], e.eps)


def run_replicated_bench(args, rank, world, dist):
    """bench.py --gpus N, small graphs: N replicas of the BASELINE-shape graph, B triples per GPU and step out of
    one globally shuffled epoch (global batch N x B); value = N*B*steps / max-over-ranks time."""
    import json
    import time

    import torch

    from . import host as H
    from . import ops
    from . import synth as S
    from .sharded import TorchComm

    U, I, E = S.SHAPES[args.workload]
    users, items = S.generate(U, I, E, seed=0)               # every rank derives the same graph and the same epoch
    ip, ix, dv = H.build_norm_adj(U, I, users, items)
    n, nnz = U + I, len(ix)
    graph = ops.Graph(ip, ix, dv, n, n, split_threshold=args.split)
    W0 = S.xavier_uniform_panel(U, I, args.dim, args.seed)
    rep = HipReplica(graph, U, I, args.dim, args.layers, True, 1e-4, 1e-3, params=W0.cuda())
    step_ = ReplicatedStep(rep, TorchComm(dist), world)
    pos_ptr = np.zeros(U + 1, dtype=np.int64)
    pos_ptr[1:] = np.cumsum(np.bincount(users, minlength=U))
    rng = H.Rng(args.seed)
    B, gB = args.batch, args.batch * world
    need = (args.steps + args.warmup) * gB
    tri = np.empty((0, 3), dtype=np.int64)
    while len(tri) < need:
        t2 = rng.sample_epoch(users, items, pos_ptr, items.astype(np.int32), I)
        tri = np.concatenate([tri, t2[rng.shuffle_perm(len(t2))]])
    # global step i = triples [i*gB, (i+1)*gB); this rank's slice of it
    mine = tri[:need].reshape(args.steps + args.warmup, world, B, 3)[:, rank]
    mine = torch.from_numpy(np.ascontiguousarray(mine)).cuda()
    tu, tp, tn = mine[..., 0].contiguous(), mine[..., 1].contiguous(), mine[..., 2].contiguous()
    last = args.warmup + args.steps - 1

    def step(i):
        if i < last:
            rep.prefetch(tu[i + 1], tp[i + 1], tn[i + 1])
        return step_.train_step(tu[i], tp[i], tn[i])

    for i in range(args.warmup):
        step(i)
    dist.barrier()
    torch.cuda.synchronize()
    t0 = time.perf_counter()
    for i in range(args.warmup, args.warmup + args.steps):
        step(i)
    t_issue = time.perf_counter() - t0
    torch.cuda.synchronize()
    dist.barrier()
    dt = torch.tensor([time.perf_counter() - t0], dtype=torch.float64,
                      device="cuda" if dist.get_backend() == "nccl" else "cpu")
    dist.all_reduce(dt, op=dist.ReduceOp.MAX)
    dt = float(dt.item())
    # replicas must stay bit-identical: compare a checksum of the tables across ranks
    chk = rep.params.double().sum().reshape(1)
    chk = chk if dist.get_backend() == "nccl" else chk.cpu()
    lo, hi = chk.clone(), chk.clone()
    dist.all_reduce(lo, op=dist.ReduceOp.MIN)
    dist.all_reduce(hi, op=dist.ReduceOp.MAX)
    if rank == 0:
        out = {
            "metric": "BPR triples/sec, LightGCN-%d dim=%d" % (args.layers, args.dim),
            "value": gB * args.steps / dt, "unit": "triples/s", "n_gpus": world, "steps": args.steps,
            "warmup": args.warmup, "ms_per_step": dt / args.steps * 1e3, "higher_is_better": True, "scaling": "weak",
            "vs_baseline": None, "dtype": "f32", "data": "synthetic",
            "config": {"workload": "%s-shape graph replicated on %d GPUs: %d users x %d items, %d train edges, nnz(A)=%d; "
                                   "LightGCN K=%d d=%d, B=%d per GPU (global batch %d, slices of one shuffled epoch); step = "
                                   "propagate + fused BPR + backward propagate on every replica, ONE all-reduce of the "
                                   "[%d,%d] fp32 gradient over %s, identical dense Adam on every replica"
                                   % (args.workload, world, U, I, len(users), nnz, args.layers, args.dim, B, gB, n, args.dim,
                                      "RCCL" if dist.get_backend() == "nccl" else dist.get_backend() + " (rehearsal, host-staged)"),
                       "batch": B, "dim": args.dim, "layers": args.layers, "parallelism": "dp%d" % world},
            "loss_last": [float(x) for x in rep.loss_row.cpu()],
            "host_issue_ms_per_step": t_issue / args.steps * 1e3,
            "replicas_bit_identical": bool(lo.item() == hi.item()),
        }
        print(json.dumps(out))
    dist.destroy_process_group()
