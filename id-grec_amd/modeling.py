"""Shared machinery of the model plugins in models/*.py.

The reference's models each own two nn.Embedding tables and call torch.cat on them in every
forward (models/LightGCN.py:38).  `PackedRecommender` keeps the same two nn.Embedding
modules (same construction order, hence the same initial weights for a given torch seed) but
backs them with ONE contiguous [num_users + num_items, d] buffer, so the propagation kernels
read the ego panel in place.  It also provides the fused, autograd-free training step and the
fused top-K evaluation that the trainer / evaluator use when a model offers them.
"""
import torch
from torch import nn

from . import ops
from .engine import PropagationEngine


class _PackedPanel(torch.autograd.Function):
    """(user_weight, item_weight) -> the [n, d] panel they both live in, without copying."""

    @staticmethod
    def forward(ctx, user_w, item_w, storage):
        ctx.U = user_w.shape[0]
        return storage.view(storage.shape)

    @staticmethod
    def backward(ctx, g):
        return g[: ctx.U], g[ctx.U:], None


class PackedRecommender(nn.Module):
    #: number of propagation layers (0 = plain matrix factorisation) and whether layer 0 is averaged in
    n_layers = 0
    include_layer0 = True

    def __init__(self, config, dataset, device):
        super().__init__()
        self.config, self.dataset, self.device = config, dataset, device
        self.reg_lambda = float(config["reg_lambda"])
        d = int(config["embedding_size"])
        # construction order fixes the torch RNG consumption: normal_(user), normal_(item),
        # then the two xavier calls (models/LightGCN.py:21-28)
        self.user_embedding = nn.Embedding(num_embeddings=dataset.num_users, embedding_dim=d)
        self.item_embedding = nn.Embedding(num_embeddings=dataset.num_items, embedding_dim=d)
        nn.init.xavier_uniform_(self.user_embedding.weight, gain=1)
        nn.init.xavier_uniform_(self.item_embedding.weight, gain=1)
        self.activation = nn.Sigmoid()
        self.Graph = None
        self._storage = None
        self._engine = None
        self._eval_cache = None
        self._pack()

    # ------------------------------------------------------------------ packed storage
    def _pack(self):
        uw, iw = self.user_embedding.weight, self.item_embedding.weight
        U = uw.shape[0]
        storage = torch.cat([uw.data, iw.data]).contiguous()
        uw.data, iw.data = storage[:U], storage[U:]
        self._storage = storage
        self._engine = None
        self._eval_cache = None

    def _apply(self, fn, *a, **kw):
        out = super()._apply(fn, *a, **kw)
        self._pack()  # .to(device) / .float() re-create the tensors: restore the aliasing
        return out

    def _is_packed(self):
        uw, iw, st = self.user_embedding.weight, self.item_embedding.weight, self._storage
        return (st is not None and uw.data_ptr() == st.data_ptr() and uw.device == st.device
                and iw.data_ptr() == st.data_ptr() + uw.numel() * 4)

    def ego_panel(self):
        """Differentiable [n, d] view of both tables."""
        if not self._is_packed():
            self._pack()
        return _PackedPanel.apply(self.user_embedding.weight, self.item_embedding.weight, self._storage)

    def train(self, mode=True):
        self._eval_cache = None  # weights may change: drop the cached evaluation panels
        return super().train(mode)

    # ------------------------------------------------------------------ graph
    def attach_graph(self, sp_mat):
        """Upload a scipy adjacency to the model's device as an ops.Graph."""
        from utility.utility_function import tools

        dev = torch.device(self.device)
        if dev.type != "cuda":
            raise RuntimeError("%s needs an MI355X: device is %s and idgrec_amd has no CPU path "
                               "(torch.cuda.is_available() = %s)." % (type(self).__name__, dev, torch.cuda.is_available()))
        self.Graph = tools.convert_sp_mat_to_graph(sp_mat, dev)
        return self.Graph

    # ------------------------------------------------------------------ fused paths
    def engine(self):
        if self._engine is None or self._engine.params.data_ptr() != self._storage.data_ptr():
            if not self._is_packed():
                self._pack()
            ops._require_device(self._storage)
            self._engine = PropagationEngine(self.Graph if self.n_layers > 0 else None, self.dataset.num_users,
                                             self.dataset.num_items, self._storage.shape[1], self.n_layers,
                                             include_layer0=self.include_layer0, reg_lambda=self.reg_lambda,
                                             params=self._storage)
        return self._engine

    #: models whose training loss is exactly [bpr, reg] over the mean-propagated panel set this
    supports_fused_step = False
    #: entries of the loss vector a fused step writes ([bpr, reg] + the model's own terms)
    n_fused_losses = 2
    #: widths the row-restricted / masked / Adam-epilogue forms of the tiled SpMM exist for (idg_graph.hip,
    #: spmm_dispatch): the fused step is built from them, any other width trains through forward() + autograd
    FUSED_WIDTHS = (32, 64, 128, 256, 512)

    def fused_step_available(self):
        """True when the trainer may use fused_train_step(): the model offers one, lives on the GPU, and — if it
        propagates — its embedding width is one the fused kernels are instantiated for.  embedding_size = 48 or 100
        (values the reference's users run) then train through the generic-width kernels under autograd instead of
        failing in the first step."""
        if not self.supports_fused_step or self._storage is None or not self._storage.is_cuda:
            return False
        return self.n_layers == 0 or int(self._storage.shape[1]) in self.FUSED_WIDTHS

    def fused_loss_and_grad(self, users, pos, neg, loss_out=None):
        """Losses [bpr, reg_lambda*reg] (device tensor) and d(sum)/d(weights) written into the
        two parameters' .grad — the work of forward() + backward() without an autograd graph."""
        eng = self.engine()
        self._eval_cache = None
        loss = eng.loss_and_grad(users, pos, neg, loss_out)
        U = self.dataset.num_users
        self.user_embedding.weight.grad = eng.grad[:U]
        self.item_embedding.weight.grad = eng.grad[U:]
        return loss

    def fused_train_step(self, users, pos, neg, loss_out, optimizer):
        """forward + backward + optimizer.step() as ONE chain of kernels (Adam is applied in the epilogue of the
        last backward product).  Returns False — nothing done — unless `optimizer` is an idgrec_amd.ops.Adam over
        exactly this model's two packed tables; its state (step, exp_avg, exp_avg_sq) stays the single source of
        truth: the moments are re-homed once into packed [n, d] panels that the state entries then view."""
        uw, iw = self.user_embedding.weight, self.item_embedding.weight
        if not isinstance(optimizer, ops.Adam) or len(optimizer.param_groups) != 1:
            return False
        group = optimizer.param_groups[0]
        if len(group["params"]) != 2 or group["params"][0] is not uw or group["params"][1] is not iw:
            return False
        eng = self.engine()
        U = self.dataset.num_users
        st_u, st_i = optimizer.state[uw], optimizer.state[iw]
        packed = getattr(self, "_packed_moments", None)
        if (packed is None or packed[0].device != self._storage.device or not st_u or not st_i
                or st_u["exp_avg"].data_ptr() != packed[0].data_ptr() or st_i["exp_avg_sq"].data_ptr() != packed[1][U:].data_ptr()):
            m, v = torch.zeros_like(self._storage), torch.zeros_like(self._storage)
            for st, sl in ((st_u, slice(0, U)), (st_i, slice(U, None))):
                if st:  # the optimizer has already stepped the other way: keep what it accumulated
                    m[sl].copy_(st["exp_avg"])
                    v[sl].copy_(st["exp_avg_sq"])
                st.setdefault("step", 0)
                st["exp_avg"], st["exp_avg_sq"] = m[sl], v[sl]
            packed = self._packed_moments = (m, v)
        if st_u["step"] != st_i["step"]:
            return False
        eng.exp_avg, eng.exp_avg_sq = packed
        eng.lr, eng.betas, eng.eps = float(group["lr"]), tuple(group["betas"]), float(group["eps"])
        eng.step_count = int(st_u["step"])
        self._eval_cache = None
        # the update consumes the gradient inside the last product's epilogue; the panel is written out only on request
        # (model.keep_fused_grad = True): otherwise .grad reads None after a fused step, as after zero_grad(set_to_none=True)
        eng.store_grad = bool(getattr(self, "keep_fused_grad", False))
        eng.train_step(users, pos, neg, loss_out)
        st_u["step"] = st_i["step"] = eng.step_count
        if eng.store_grad:
            uw.grad, iw.grad = eng.grad[:U], eng.grad[U:]
        else:
            uw.grad = iw.grad = None
        return True

    def prefetch_batch(self, users, pos, neg):
        """One-batch lookahead for the fused step (row bitmap + scatter plan on the side stream; the plan alone for
        MFBPR)."""
        if self.fused_step_available():
            self.engine().prefetch(users, pos, neg)

    def final_panels(self):
        """(users [U,d'], items [I,d']) used for scoring, cached while the weights are frozen.
        Default: the layer-mean propagation of the packed panel; encoders with their own
        aggregate() (NGCF) override `_eval_panels`."""
        if self._eval_cache is None:
            self._eval_cache = self._eval_panels()
        return self._eval_cache

    def _eval_panels(self):
        fin = self.engine().propagate(force=True)
        U = self.dataset.num_users
        return fin[:U], fin[U:]

    def get_rating_for_test(self, user):
        """sigmoid(E_u[user] . E_i^T) as a dense [B, num_items] matrix (models/LightGCN.py:74-80)."""
        with torch.no_grad():
            ue, ie = self.final_panels()
            return ops.score_dense(ue, ie, user.long(), apply_sigmoid=True)

    #: ranks one call of the fused scoring / top-K entry point returns (idg_score_topk_f32: one pass per 64 ranks)
    FUSED_TOPK_MAX = 1024

    def topk_for_test(self, user, k):
        """Top-k unseen items per user without materialising the rating matrix (k up to FUSED_TOPK_MAX; beyond that —
        torch.topk takes any k <= num_items, batch_test.py:68 — the dense rating matrix, the reference's mask and
        torch.topk, batch by batch)."""
        with torch.no_grad():
            ue, ie = self.final_panels()
            ip, ix = self.dataset.train_csr_on(ue.device)
            user = user.long()
            if k <= self.FUSED_TOPK_MAX:
                return ops.score_topk(ue, ie, user, k, ip, ix, apply_sigmoid=True)
            out = []
            for lo in range(0, user.shape[0], 256):
                u = user[lo:lo + 256]
                rating = ops.score_dense(ue, ie, u, apply_sigmoid=True)
                cnt = ip[u + 1] - ip[u]
                rows = torch.repeat_interleave(torch.arange(u.shape[0], device=u.device), cnt)
                cols = torch.cat([ix[int(a):int(b)] for a, b in zip(ip[u].tolist(), ip[u + 1].tolist())]).long() if int(cnt.sum()) else rows
                rating[rows, cols] = -1
                out.append(torch.topk(rating, k=k)[1])
            return torch.cat(out)
