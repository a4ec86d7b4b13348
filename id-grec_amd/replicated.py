"""Data-parallel replicas of the fused training step: every rank holds the whole graph and both
embedding tables, a global batch of world x B triples is cut into per-rank slices, and every rank applies
the same Adam update — synchronous data-parallel SGD with the reference's loss at batch size
world x B (loss = mean over the global batch, trainer.py:42-56).

What the ranks exchange (one collective per step either way):
  * RowExchangeStep (default): backward propagation is LINEAR in the batch's gradient rows, so the ranks all-gather
    those rows (<= 3B rows of d floats per rank: 0.8 MB at B = 1024, d = 64) BEFORE it, merge them in rank order and
    every replica back-propagates the averaged rows — with the Adam update in the last product's epilogue, exactly as
    on one GPU.  Exchanged bytes do not grow with the graph.
  * ReplicatedStep: every rank back-propagates its own rows and the dense [n, d] gradients are averaged by an
    all-reduce (17.8 MB at yelp2018 size) AFTER it; Adam is a separate pass.  Kept as the plain form (and for
    engines whose backward is not exposed as rows).

When to use which multi-GPU form (bench.py --parallel auto decides by the panel size):
  * replicas (this file): the graph is small enough that one GPU propagates it in tens of
    microseconds (BASELINE configs 1-4: panels of 18-37 MB).  Sharding such a graph by user rows
    needs 2K+1 all-reduces of the item panel per step, each longer than the product it follows.
  * user-row shards (sharded.py): the propagation itself is the cost (config 5: 70 ms per product
    on one GPU) and divides across ranks.

The engine is anything with `loss_and_grad(users, pos, neg) -> loss[2]`, `.grad` [n, d] and
`adam_step()` (ReplicatedStep) or `train_step_rows(users, pos, neg, gather) -> loss[2]` (RowExchangeStep): the HIP
engine in the product, a checker-backed stub in the CPU (gloo) tests.
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


class RowExchangeStep:
    def __init__(self, engine, comm, world):
        self.eng, self.comm, self.world = engine, comm, int(world)

    def _gather(self, message, out):
        """out[r * L:(r + 1) * L] = rank r's message."""
        self.comm.wait(self.comm.all_gather_async(out, message))

    def train_step(self, users, pos, neg):
        """users/pos/neg: THIS rank's slice of the global batch (the same B on every rank).  Returns the global
        [bpr, reg] losses."""
        return self.eng.train_step_rows(users, pos, neg, self._gather)


class HipReplica:
    """PropagationEngine as one replica.  For ReplicatedStep its gradient panel is extended by one row that carries
    the two losses, so that one all-reduce moves both; for RowExchangeStep (world given) the engine's exchange hook
    packs / all-gathers / merges the batch's gradient rows between the BPR kernel and the backward propagation."""

    def __init__(self, graph, num_users, num_items, dim, n_layers, include_layer0=True, reg_lambda=1e-4, lr=1e-3,
                 params=None, world=1):
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
        self.world = int(world)
        self._msg = self._msgs = self._gather = None
        self._union = torch.zeros((n + 31) // 32, dtype=torch.int32, device=self.eng.device)

    # ---- RowExchangeStep
    def train_step_rows(self, users, pos, neg, gather):
        self._gather = gather
        self.eng.exchange = self._exchange
        return self.eng.train_step(users, pos, neg, loss_out=self.loss_row)

    def _exchange(self, slot, loss):
        e, B = self.eng, slot.ws_B
        words = self.ops.bpr_rows_message_floats(B, e.d)
        if self._msg is None or self._msg.numel() != words:
            self._msg = self.torch.zeros(words, dtype=self.torch.float32, device=e.device)
            self._msgs = self.torch.zeros(words * self.world, dtype=self.torch.float32, device=e.device)
        self.ops.bpr_pack_rows_raw(slot.ws, B, e.g_final, loss, self._msg, clear=self._union, clear_bits=e.n)
        self._gather(self._msg, self._msgs)
        self.ops.bpr_unpack_rows_raw(self._msgs, self.world, B, e.params, e.reg_lambda, e.g_final, e.grad, self._union, loss,
                                     touched_is_clear=True)
        return self._union

    # ---- ReplicatedStep
    def prefetch(self, users, pos, neg):
        self.eng.prefetch(users, pos, neg)

    def loss_and_grad(self, users, pos, neg):
        self.eng.exchange = None
        return self.eng.loss_and_grad(users, pos, neg, loss_out=self.loss_row)

    def scale_grad_and_loss(self, a):
        self.ops.lincomb_raw(self.grad_and_loss, self.grad_and_loss, a)

    def adam_step(self):
        e = self.eng
        if e.exp_avg is None:
            e.exp_avg, e.exp_avg_sq = self.torch.zeros_like(e.params), self.torch.zeros_like(e.params)
        e.step_count += 1
        self.ops.adam_step_raw(e.params, e.grad, e.exp_avg, e.exp_avg_sq, e.lr, e.step_count, e.betas[0], e.betas[1], e.eps)


def run_replicated_bench(args, rank, world, dist, comm, comm_name, workload=None, dim=None):
    """bench.py --gpus N, replica form: N replicas of the BASELINE-shape graph, B triples per GPU and step out of
    one globally shuffled epoch — a GLOBAL batch of N x B, i.e. N times the reference's batch_size per Adam step
    (trainer.py:36): a different optimisation than the reference's, reported for what it is.  value = N*B*steps /
    max-over-ranks time.  Returns the bench line (rank 0) or None."""
    import time

    import torch

    from . import host as H
    from . import ops
    from . import synth as S

    workload = workload or args.workload
    dim = dim or args.dim
    U, I, E = S.SHAPES[workload]
    users, items = S.generate(U, I, E, seed=0)               # every rank derives the same graph and the same epoch
    ip, ix, dv = H.build_norm_adj(U, I, users, items)
    n, nnz, n_edges = U + I, len(ix), len(users)
    graph = ops.Graph(ip, ix, dv, n, n, split_threshold=args.split)
    W0 = S.xavier_uniform_panel(U, I, dim, args.seed)
    rep = HipReplica(graph, U, I, dim, args.layers, True, 1e-4, 1e-3, params=W0.cuda(), world=world)
    rows = getattr(args, "dp_exchange", "rows") == "rows"
    step_ = (RowExchangeStep if rows else ReplicatedStep)(rep, comm, world)
    B, gB = args.batch, args.batch * world
    need = (args.steps + args.warmup) * gB
    tri = S.draw_triples(args.seed, users, items, U, I, need)[0]  # the same global sequence on every rank
    del users, items, ip, ix, dv
    # global step i = triples [i*gB, (i+1)*gB); this rank's slice of it
    mine = tri[:need].reshape(args.steps + args.warmup, world, B, 3)[:, rank]
    mine = torch.from_numpy(np.ascontiguousarray(mine)).cuda()
    tu, tp, tn = mine[..., 0].contiguous(), mine[..., 1].contiguous(), mine[..., 2].contiguous()
    last = args.warmup + args.steps - 1

    def step(i):
        if i < last:
            rep.prefetch(tu[i + 1], tp[i + 1], tn[i + 1])
        return step_.train_step(tu[i], tp[i], tn[i])

    S.ramp_clocks()
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
    out = None
    if rank == 0:
        out = {
            "metric": "BPR triples/sec, LightGCN-%d dim=%d" % (args.layers, dim),
            "value": gB * args.steps / dt, "unit": "triples/s", "n_gpus": world, "steps": args.steps,
            "warmup": args.warmup, "ms_per_step": dt / args.steps * 1e3, "higher_is_better": True, "scaling": "weak",
            "vs_baseline": None, "dtype": "f32", "data": "synthetic",
            "config": {"workload": "%s-shape graph replicated on %d GPUs: %d users x %d items, %d train edges, nnz(A)=%d; "
                                   "LightGCN K=%d d=%d, B=%d per GPU (global batch %d, slices of one shuffled epoch); step = "
                                   "propagate + fused BPR on every replica, %s over %s, backward propagate + identical "
                                   "dense Adam on every replica"
                                   % (workload, world, U, I, n_edges, nnz, args.layers, dim, B, gB,
                                      ("ONE all-gather of the batches' gradient rows (%d fp32 words per rank) BEFORE the "
                                       "backward propagation" % ops.bpr_rows_message_floats(B, dim)) if rows else
                                      ("ONE all-reduce of the [%d,%d] fp32 gradient AFTER the backward propagation" % (n, dim)),
                                      "RCCL" if dist.get_backend() == "nccl" else dist.get_backend() + " (rehearsal, host-staged)"),
                       "batch": B, "dim": dim, "layers": args.layers, "parallelism": "dp%d" % world,
                       "global_batch": gB, "note": "global batch N x B = %d: N x the reference's batch_size per Adam step" % gB,
                       "exchange": "gradient rows (all-gather)" if rows else "dense gradient (all-reduce)",
                       "comm": comm_name},
            "loss_last": [float(x) for x in rep.loss_row.cpu()],
            "host_issue_ms_per_step": t_issue / args.steps * 1e3,
            "replicas_bit_identical": bool(lo.item() == hi.item()),
        }
    del rep, step_, graph
    torch.cuda.empty_cache()
    return out
