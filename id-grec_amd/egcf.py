"""Fused, autograd-free training steps of EGCF's two encoders — `parallel` (EgcfEngine, described here) and `alternating`
(EgcfAltEngine, at the end of the file) — (reference: models/EGCF.py:46-111 +
utility/utility_train/trainer.py:42-56) as a fixed chain of C-ABI calls on preallocated panels — what engine.py is for the
LightGCN family (61 % of an EGCF epoch was stock ATen kernels: tanh, cat, gathers, three InfoNCE terms and
their autograd mirrors).

Forward (n = U + I rows, users first; E = the item table, the ONLY parameter):
    X0 = [tanh(R.E) ; E]                R = D_u^-1/2 R D_i^-1/2 (rectangular), tanh in the product's epilogue
    X_k = tanh(A . X_{k-1}),  k = 1..K  A = the symmetric normalised adjacency; TOT = ((X_1 + X_2) + ...) + X_K
    loss = BPR(TOT) + reg(E[pos], E[neg]) + ssl_lambda (InfoNCE(u, u) + InfoNCE(p, p) + InfoNCE(u, p))   raw batch rows
The last layer and TOT are produced at the batch's rows only (their live-unit list).  Backward, with g = d loss / d TOT
(stored at the batch's rows by the BPR scatter, the InfoNCE terms added into the same rows):
    Z_K = g . (1 - X_K^2)                                     rows kernel, batch rows
    Z_k = (A . Z_{k+1} + g) . (1 - X_k^2),  k = K-1..1        tanh' in the epilogue; the first product reads <= 3B live rows
    W   = A . Z_1 (+ reg rows);  W[:U] *= 1 - X0[:U]^2        one launch: the derivative applies to the user rows only
    dE  = R^T . W[:U] + W[U:]  -> Adam                         epilogue Adam on the item table, gradient not stored
No torch arithmetic runs in a step (the three InfoNCE losses are summed by a 3-element torch.sum: the only stock kernel)."""
import torch

from . import native, ops
from .engine import BatchPrep


class EgcfEngine:
    def __init__(self, graph, user_graph, num_users, num_items, dim, n_layers, item_weight, reg_lambda, ssl_lambda,
                 temperature, lr=1e-3, betas=(0.9, 0.999), eps=1e-8, store_grad=False):
        """graph: symmetric ops.Graph on [n, n]; user_graph: rectangular ops.Graph [U, I] with its transposed handle
        (.T); item_weight: [I, d] initial table (copied into this engine's storage: see item_table())."""
        self.G, self.R = graph, user_graph
        self.U, self.I, self.d, self.K = int(num_users), int(num_items), int(dim), int(n_layers)
        self.n = self.U + self.I
        if not 1 <= self.K <= 4:
            raise ValueError("EgcfEngine: 1 <= GCN_layer <= 4 (the layer sum is formed by the last product's epilogue)")
        self.reg_lambda, self.ssl_lambda, self.temperature = float(reg_lambda), float(ssl_lambda), float(temperature)
        self.lr, self.betas, self.eps = float(lr), betas, float(eps)
        self.store_grad = bool(store_grad)   # keep d loss / d E in grad_items() (tests); the step itself does not need it
        dev = graph.device
        self.device = dev
        f32 = dict(dtype=torch.float32, device=dev)
        n, d = self.n, self.d
        # EGO: the regulariser's view of the parameters — user rows are zero for good (EGCF has no user table: the loss's
        # reg term covers the two item blocks, models/EGCF.py:95-96; a zero block adds exactly 0), item rows ARE the table
        self.EGO = torch.zeros((n, d), **f32)
        self.EGO[self.U:].copy_(item_weight)
        self.X = [torch.empty((n, d), **f32) for _ in range(self.K + 1)]   # X0 .. XK (tanh outputs: kept for the backward)
        self.TOT = torch.empty((n, d), **f32)
        self.GT = torch.zeros((n, d), **f32)   # d loss / d TOT, rows of the batch
        self.GE = torch.zeros((n, d), **f32)   # regulariser's gradient, rows of the batch
        self.Z = [torch.empty((n, d), **f32) for _ in range(2)]            # backward ping-pong
        self.M = torch.zeros((self.I, d), **f32)
        self.V = torch.zeros((self.I, d), **f32)
        self.prep = BatchPrep(self.U, n, d, dev, units_graph=graph, extra=self._plan_ssl)  # bitmap, live units, scatter plan: side stream, one batch ahead
        self.loss = torch.zeros(3, **f32)        # [bpr, reg_lambda * reg, ssl_lambda * (three InfoNCE terms)]
        self._ssl = torch.zeros(4, **f32)        # user-user, pos-pos, user-pos (+ pad)
        self.step_count = 0
        self._final_version = -1
        self._grad_items = None

    def grad_items(self):
        """d loss / d E of the last step ([I, d]; only with store_grad)."""
        return self._grad_items

    def item_table(self):
        """The parameter storage ([I, d] view): what nn.Embedding.weight aliases."""
        return self.EGO[self.U:]

    # ---- forward over every row (evaluation): returns (users, items) views of TOT
    @torch.no_grad()
    def propagate(self, out_rows=None):
        U, K, X = self.U, self.K, self.X
        E = self.EGO[U:]
        ops.spmm_epi_raw(self.R, E, Y=X[0][:U], act=native.ACT_TANH)
        ops.lincomb_raw(X[0][U:], E, 1.0)
        for k in range(1, K):
            ops.spmm_epi_raw(self.G, X[k - 1], Y=X[k], act=native.ACT_TANH)
        terms = X[1:K] + [None, None, None]
        ops.spmm_epi_raw(self.G, X[K - 1], Y=X[K], sum_in=terms[0], sum_in2=terms[1] if terms[0] is not None else None,
                         sum_in3=terms[2] if terms[1] is not None else None, sum_out=self.TOT, act=native.ACT_TANH,
                         out_rows=out_rows)
        return self.TOT[:U], self.TOT[U:]

    def final_panels(self):
        if self._final_version != self.step_count:
            self.propagate()
            self._final_version = self.step_count
        return self.TOT[: self.U], self.TOT[self.U:]

    # ---- one training step
    @torch.no_grad()
    def train_step(self, users, pos, neg, loss_out=None):
        U, K, X, n, d = self.U, self.K, self.X, self.n, self.d
        loss = self.loss if loss_out is None else loss_out
        # index-only work (the batch's row bitmap, its live units, the sorted scatter plan): prepared on the side stream by
        # prefetch() during the previous step, or right now
        slot = self.prep.take(users, pos, neg)
        bitmap = slot.bitmap
        self.propagate(out_rows=bitmap)
        # losses: gradient rows STORED at the batch's rows (bitmap), the InfoNCE terms added into the same rows
        ops.bpr_fused_raw(self.TOT, self.EGO, users, pos, neg, U, self.reg_lambda, self.GT, self.GE, loss=loss[:2],
                          deterministic=2 | native.IDG_BPR_TOUCHED_PRESET, touched=bitmap, ws=slot.ws)
        # (their id lists, repeat flags and positions are in the slot's two workspaces: _plan_ssl, side stream)
        ops.infonce_pair_raw(self.TOT, self.TOT, users, pos, U, self.temperature, g1=self.GT, g2=self.GT, loss=self._ssl[:2],
                             dedup=False, grad_scale=self.ssl_lambda, accumulate=True, ws=slot.ssl_ws[0], planned=True)
        ops.infonce_cross_raw(self.TOT, users, pos, U, self.temperature, g=self.GT, loss=self._ssl[2:4],
                              grad_scale=self.ssl_lambda, ws=slot.ssl_ws[1], planned=True)
        torch.sum(self._ssl[:3], dim=0, keepdim=True, out=loss[2:3])
        loss[2:3].mul_(self.ssl_lambda)
        self._backward(slot, bitmap)
        self.prep.release(slot)
        return loss

    def _backward(self, slot, bitmap):
        U, K, X = self.U, self.K, self.X
        Za, Zb = self.Z
        ops.rows_tanh_bwd_raw(self.GT, X[K], bitmap, Za)                           # Z_K at the batch's rows
        x_rows = bitmap
        for k in range(K - 1, 0, -1):
            ops.spmm_epi_raw(self.G, Za, Y=Zb, addend=self.GT, mask=bitmap, act=native.ACT_TANH_BWD, act_src=X[k],
                             x_rows=x_rows)
            Za, Zb, x_rows = Zb, Za, None
        ops.spmm_epi_raw(self.G, Za, Y=Zb, addend=self.GE, mask=bitmap, act=native.ACT_TANH_BWD, act_src=X[0],
                         act_rows=U, x_rows=x_rows)
        self.step_count += 1
        ops.spmm_epi_raw(self.R.T, Zb[:U], addend=Zb[U:], sum_out=Za[U:],
                         adam=(self.EGO[U:], self.M, self.V, self.lr, self.step_count, self.betas[0], self.betas[1], self.eps),
                         adam_discard_grad=not self.store_grad)
        self._grad_items = Za[U:] if self.store_grad else None

    def _plan_ssl(self, slot, users, pos, neg, stream):
        """BatchPrep's hook: the id-list stages of the step's two InfoNCE calls (raw user / item lists; the cross form), each
        into a workspace of the slot's own."""
        B = int(users.shape[0])
        if getattr(slot, "ssl_B", -1) != B:
            slot.ssl_ws = (ops.infonce_workspace(self.n, B, self.d, self.device), ops.infonce_workspace(self.n, B, self.d, self.device))
            slot.ssl_B = B
        ops.infonce_plan_raw(users, pos, self.U, self.n, self.d, ops.SSL_RAW, slot.ssl_ws[0], stream=stream)
        ops.infonce_plan_raw(users, pos, self.U, self.n, self.d, ops.SSL_CROSS, slot.ssl_ws[1], stream=stream)

    def prefetch(self, users, pos, neg):
        """One-batch lookahead of the index-only work of the NEXT step (side stream)."""
        self.prep.prefetch(users, pos, neg)


class EgcfAltEngine(EgcfEngine):
    """The `alternating` encoder (models/EGCF.py:46-62) as the same kind of chain.  Forward, l = 1..K, i_0 = E:
        u_l = tanh(R . i_{l-1}),   i_l = tanh(R^T . u_l);      TOT = [u_1 + .. + u_K ; i_1 + .. + i_K]
    (rectangular operator and its transposed handle; layer sums in the last products' epilogues; in a training step the last
    item product and the item sum are produced at the batch's item rows only).  Backward with g = d loss / d TOT at the
    batch's rows, Zi_K = g_I . (1 - i_K^2):
        Zu_l = (R . Zi_l + g_U) . (1 - u_l^2),    Zi_{l-1} = (R^T . Zu_l + g_I) . (1 - i_{l-1}^2),   l = K..1
        dE = R^T . Zu_1 (+ the regulariser's rows)  -> Adam in that product's epilogue."""

    def __init__(self, user_graph, num_users, num_items, dim, n_layers, item_weight, reg_lambda, ssl_lambda, temperature,
                 lr=1e-3, betas=(0.9, 0.999), eps=1e-8, store_grad=False):
        self._alt_graph = user_graph
        super().__init__(user_graph, user_graph, num_users, num_items, dim, n_layers, item_weight, reg_lambda, ssl_lambda,
                         temperature, lr=lr, betas=betas, eps=eps, store_grad=store_grad)
        f32 = dict(dtype=torch.float32, device=self.device)
        # (no symmetric [n, n] operator here: the unit lists BatchPrep would build for one are not used)
        self.prep = BatchPrep(self.U, self.n, self.d, self.device, units_graph=None, extra=self._plan_alt)
        self.ZU = torch.empty((self.U, self.d), **f32)
        self.ZI = [torch.empty((self.I, self.d), **f32) for _ in range(2)]
        self.GI = torch.empty((self.I, self.d), **f32)  # d loss / d E when it is kept

    def _plan_alt(self, slot, users, pos, neg, stream):
        """BatchPrep's hook: the InfoNCE id lists, and the bitmap of the batch's ITEM rows in the item block's own numbering
        (the rectangular operators' item-side rows / columns)."""
        self._plan_ssl(slot, users, pos, neg, stream)
        if getattr(slot, "items_bitmap", None) is None:
            slot.items_bitmap = torch.zeros((self.I + 31) // 32, dtype=torch.int32, device=self.device)
        ops.bpr_touch_rows_raw(pos, pos, neg, 0, slot.items_bitmap, stream=stream, clear_bits=self.I)

    @torch.no_grad()
    def propagate(self, out_rows=None, item_rows=None):
        U, K, X = self.U, self.K, self.X
        R, Rt = self.R, self.R.T
        prev = self.EGO[U:]
        for l in range(1, K + 1):
            last = l == K
            tu = [X[j][:U] for j in range(1, K)] + [None, None, None]
            ti = [X[j][U:] for j in range(1, K)] + [None, None, None]
            ops.spmm_epi_raw(R, prev, Y=X[l][:U], act=native.ACT_TANH,
                             **(dict(sum_in=tu[0], sum_in2=tu[1] if tu[0] is not None else None,
                                     sum_in3=tu[2] if tu[1] is not None else None, sum_out=self.TOT[:U]) if last else {}))
            ops.spmm_epi_raw(Rt, X[l][:U], Y=X[l][U:], act=native.ACT_TANH,
                             **(dict(sum_in=ti[0], sum_in2=ti[1] if ti[0] is not None else None,
                                     sum_in3=ti[2] if ti[1] is not None else None, sum_out=self.TOT[U:],
                                     out_rows=item_rows) if last else {}))
            prev = X[l][U:]
        return self.TOT[:U], self.TOT[U:]

    @torch.no_grad()
    def train_step(self, users, pos, neg, loss_out=None):
        U = self.U
        loss = self.loss if loss_out is None else loss_out
        slot = self.prep.take(users, pos, neg)
        bitmap = slot.bitmap
        self.propagate(item_rows=slot.items_bitmap)
        ops.bpr_fused_raw(self.TOT, self.EGO, users, pos, neg, U, self.reg_lambda, self.GT, self.GE, loss=loss[:2],
                          deterministic=2 | native.IDG_BPR_TOUCHED_PRESET, touched=bitmap, ws=slot.ws)
        ops.infonce_pair_raw(self.TOT, self.TOT, users, pos, U, self.temperature, g1=self.GT, g2=self.GT, loss=self._ssl[:2],
                             dedup=False, grad_scale=self.ssl_lambda, accumulate=True, ws=slot.ssl_ws[0], planned=True)
        ops.infonce_cross_raw(self.TOT, users, pos, U, self.temperature, g=self.GT, loss=self._ssl[2:4],
                              grad_scale=self.ssl_lambda, ws=slot.ssl_ws[1], planned=True)
        torch.sum(self._ssl[:3], dim=0, keepdim=True, out=loss[2:3])
        loss[2:3].mul_(self.ssl_lambda)
        self._backward(slot, bitmap)
        self.prep.release(slot)
        return loss

    def _backward(self, slot, bitmap):
        U, K, X = self.U, self.K, self.X
        R, Rt = self.R, self.R.T
        BI = slot.items_bitmap
        gU, gI = self.GT[:U], self.GT[U:]
        Zi, Zn = self.ZI
        ops.rows_tanh_bwd_raw(gI, X[K][U:], BI, Zi)                                  # Zi_K at the batch's item rows
        x_rows = BI
        for l in range(K, 0, -1):
            # Zu_l = (R . Zi_l + g_U) . (1 - u_l^2); g_U lives on the batch's user rows (the first U bits of the bitmap)
            ops.spmm_epi_raw(R, Zi, Y=self.ZU, addend=gU, mask=bitmap, act=native.ACT_TANH_BWD, act_src=X[l][:U], x_rows=x_rows)
            x_rows = None
            if l > 1:
                ops.spmm_epi_raw(Rt, self.ZU, Y=Zn, addend=gI, mask=BI, act=native.ACT_TANH_BWD, act_src=X[l - 1][U:])
                Zi, Zn = Zn, Zi
        self.step_count += 1
        ops.spmm_epi_raw(Rt, self.ZU, addend=self.GE[U:], mask=BI, sum_out=self.GI,
                         adam=(self.EGO[U:], self.M, self.V, self.lr, self.step_count, self.betas[0], self.betas[1], self.eps),
                         adam_discard_grad=not self.store_grad)
        self._grad_items = self.GI if self.store_grad else None
