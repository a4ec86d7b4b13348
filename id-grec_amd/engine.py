"""Fused training / evaluation engine for the LightGCN family: the reference's per-batch
sequence (models/LightGCN.py:54-72 + utility/utility_train/trainer.py:42-56)

    aggregate() -> gather -> bpr + reg loss -> backward -> Adam.step()

as a fixed chain of C-ABI calls on preallocated device buffers — no autograd graph, no
temporaries, no host synchronisation.  Numerically it is the same chain the autograd
operators in ops.py execute (tests compare the two), so models may use either.

Embedding layout: ONE [num_users + num_items, d] panel, users first; the model's two
nn.Embedding weights are views into it (no torch.cat per step).
"""
import torch

from . import ops


class PropagationEngine:
    def __init__(self, graph, num_users, num_items, dim, n_layers, include_layer0=True, reg_lambda=1e-4, lr=1e-3,
                 betas=(0.9, 0.999), eps=1e-8, deterministic=True, params=None):
        self.graph = graph  # None => no propagation (MFBPR)
        self.U, self.I, self.d = int(num_users), int(num_items), int(dim)
        self.n = self.U + self.I
        self.K, self.inc = int(n_layers), bool(include_layer0)
        self.reg_lambda, self.lr, self.betas, self.eps = float(reg_lambda), float(lr), betas, float(eps)
        self.deterministic = bool(deterministic)
        dev = graph.device if graph is not None else torch.device("cuda", torch.cuda.current_device())
        self.device = dev
        f32 = dict(dtype=torch.float32, device=dev)
        self.params = torch.empty((self.n, self.d), **f32) if params is None else params
        assert self.params.is_cuda and self.params.is_contiguous() and self.params.shape == (self.n, self.d)
        self.grad = torch.zeros((self.n, self.d), **f32)      # d loss / d E0
        self.exp_avg = self.exp_avg_sq = None                 # Adam moments: allocated by the first train_step()
        self.final = torch.empty((self.n, self.d), **f32) if graph is not None else None
        self.g_final = torch.zeros((self.n, self.d), **f32) if graph is not None else None
        words = (self.n + 31) // 32
        self._touched = [torch.zeros(words, dtype=torch.int32, device=dev) for _ in range(2)] if graph is not None else None
        self.touched = self._touched[0] if graph is not None else None
        self._parity = 0
        self.loss = torch.zeros(2, **f32)
        self.step_count = 0
        self._final_version = -1  # step_count the cached propagation belongs to
        self._side = None   # side stream for index-only work
        self.events = None  # bench.py: list collecting (start, end) HIP events around each propagation

    # ---- views handed to nn.Embedding
    def user_weight(self):
        return self.params[: self.U]

    def item_weight(self):
        return self.params[self.U:]

    # ---- forward only (evaluation)
    @torch.no_grad()
    def propagate(self, force=False):
        """Final user/item panels for scoring.  LightGCN re-runs aggregate() for every test
        batch (models/LightGCN.py:75) although the weights do not change between batches; the
        result is cached per optimizer step."""
        if self.graph is None:
            return self.params
        if force or self._final_version != self.step_count:
            self.graph.propagate_mean_raw(self.params, self.K, self.inc, out=self.final)
            self._final_version = self.step_count  # invalidated by loss_and_grad() / model.train()
        return self.final

    # ---- forward + backward: losses [bpr, reg_lambda*reg] and d(sum)/dE0 into self.grad
    @torch.no_grad()
    def loss_and_grad(self, users, pos, neg, loss_out=None):
        loss = self.loss if loss_out is None else loss_out
        det = int(self.deterministic)
        plan_done = None
        if self.deterministic and self.graph is not None:
            # the (row, slot) sort needs only the indices: run it on a side stream under the forward propagation
            main = torch.cuda.current_stream()
            if self._side is None:
                self._side = torch.cuda.Stream(device=self.device)
            self._side.wait_stream(main)  # the previous step's scatter has consumed the old plan
            with torch.cuda.stream(self._side):
                self.touched = self._touched[self._parity]  # this step's (already clear) bitmap
                self._parity ^= 1
                self._touched[self._parity].zero_()         # clear the next step's, off the critical path
                ops.bpr_plan_raw(users, pos, neg, self.U, self.n, self.d)
                plan_done = self._side.record_event()
            det = 2
        if not (self.graph is not None and det):
            self.grad.zero_()
        if self.graph is not None:
            ev = self._mark()
            self.graph.propagate_mean_raw(self.params, self.K, self.inc, out=self.final)
            self._mark(ev)
            if det:
                # deterministic scatter: the rows a batch reaches are stored (g_final and the regulariser
                # gradient in self.grad alike) and flagged in a bitmap; the backward propagation reads
                # flagged rows only, so neither panel is ever zero-filled
                if plan_done is None:
                    self.touched.zero_()
                touched = self.touched
            else:
                self.g_final.zero_()
                touched = None
            if plan_done is not None:
                torch.cuda.current_stream().wait_event(plan_done)
            ops.bpr_fused_raw(self.final, self.params, users, pos, neg, self.U, self.reg_lambda, self.g_final,
                              self.grad, loss=loss, deterministic=det, touched=touched)
            ev = self._mark()
            self.graph.propagate_mean_bwd_raw(self.g_final, self.K, self.inc, out=self.grad, accumulate=True,
                                              mask=touched)
            self._mark(ev)
        else:
            ops.bpr_fused_raw(self.params, self.params, users, pos, neg, self.U, self.reg_lambda, self.grad,
                              self.grad, loss=loss, deterministic=det)
        self._final_version = -1
        return loss

    # ---- one whole training step (loss_and_grad + dense Adam), for callers without a torch optimizer
    @torch.no_grad()
    def train_step(self, users, pos, neg, loss_out=None):
        loss = self.loss_and_grad(users, pos, neg, loss_out)
        if self.exp_avg is None:
            self.exp_avg, self.exp_avg_sq = torch.zeros_like(self.params), torch.zeros_like(self.params)
        self.step_count += 1
        ops.adam_step_raw(self.params, self.grad, self.exp_avg, self.exp_avg_sq, self.lr, self.step_count,
                          self.betas[0], self.betas[1], self.eps)
        return loss

    def _mark(self, start=None):
        if self.events is None:
            return None
        e = torch.cuda.Event(enable_timing=True)
        e.record()  # torch's current stream == the stream the kernels are launched on
        if start is not None:
            self.events.append((start, e))
        return e

    # ---- evaluation
    @torch.no_grad()
    def topk(self, users, k, excl_indptr=None, excl_items=None):
        fin = self.propagate()
        return ops.score_topk(fin[: self.U], fin[self.U:], users, k, excl_indptr, excl_items)

    @torch.no_grad()
    def rating(self, users):
        fin = self.propagate()
        return ops.score_dense(fin[: self.U], fin[self.U:], users)
