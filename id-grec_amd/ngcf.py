"""Fused, autograd-free training step of NGCF (reference: models/NGCF.py:67-128 + utility/utility_train/trainer.py:42-56)
as a fixed chain of C-ABI calls on preallocated panels (the loss, the gathers, the concatenation, the bias
sums and fourteen per-tensor Adam launches of an NGCF step ran on stock ATen kernels under autograd).

Forward, layer l = 1..K (ego_0 = the packed [n, d] embedding panel, G = D^-1/2 (A + I) D^-1/2):
    side = G . ego_{l-1}                                              idg_spmm_f32
    S    = side . W_gcn + (ego_{l-1} * side) . W_bi                   idg_ngcf_transform_f32 (fp32 MFMA), keeps ego * side
    ego_l, N_l = tail(S + b_gcn + b_bi): LeakyReLU, dropout, normalize  idg_ngcf_tail_ex_f32, N_l written INTO its slot of
                                                                      the concatenated final rows [n, (K+1) d] (NGCF.py:108)
    loss = BPR(final) + reg(item rows only, NGCF.py:125)              idg_bpr_fused_ex_f32 (final width (K+1) d, ego width d)
Backward, l = K..1, with gN_l = layer l's slot of d loss / d final (stored at the batch's rows only):
    gT = tail'(ego_l; gE_l, gN_l);  g b_gcn = g b_bi = column sums of gT;  g W_gcn = side^T gT;  g W_bi = (ego * side)^T gT
    g_side, g_ego = transform'(gT);  gE_{l-1} = G . g_side + g_ego    (G symmetric)
and for l = 1 the batch's rows of slot 0 and of the regulariser's gradient join g_ego, and the product's epilogue applies
Adam to the embedding panel.  The 4K small tensors live in ONE flat buffer (views handed back to the nn.Parameters): one
Adam launch for all of them.  Node dropout (off in configure/NGCF.txt): the step's products run on a masked copy of the
handle redrawn in place per step (forward) and on its transposed copy (backward).

d = 64 (round 4): the three launches of a layer's forward and the four of its backward are ONE kernel each
(idg_ngcf_layer_fwd_f32 / idg_ngcf_layer_bwd_f32, csrc/idg_ngcf.hip): 64-row tiles staged once in LDS, S, ego * side and gT
never stored — E, N, g_side, g_ego bit-identical to the chain above, parameter gradients summed in another fixed order."""
import ctypes as C
import os

import torch

from . import native, ops
from .engine import BatchPrep

lib, check = native.lib, native.check


class NgcfEngine:
    def __init__(self, graph, num_users, num_items, params, small, slope=0.2, mess_dropout=(0.1, 0.1, 0.1), reg_lambda=1e-4,
                 lr=1e-4, betas=(0.9, 0.999), eps=1e-8, store_grad=False, node_keep_prob=None):
        """params: the packed [n, d] embedding panel (users first; updated in place).  small: K tuples (W_gcn [d, d],
        b_gcn [1, d], W_bi [d, d], b_bi [1, d]) of tensors — copied into this engine's flat buffer; small_views() returns
        the views to re-point the nn.Parameters at."""
        self.G = graph
        # node dropout (models/NGCF.py:56-65, 73-79): ONE edge mask per training forward, shared by the layers — a masked
        # copy of the handle redrawn in place per step (Graph.dropout_copy); the masked operator is not symmetric, so the
        # backward products run on its transposed copy.  None: off (configure/NGCF.txt).
        self.node_keep = None if node_keep_prob is None else float(node_keep_prob)
        self._dropped = None
        self._Gf = self._Gb = graph  # the step's forward / backward operators
        self.U, self.I = int(num_users), int(num_items)
        self.n, self.d = int(params.shape[0]), int(params.shape[1])
        self.K = len(small)
        assert self.n == self.U + self.I and params.is_cuda and params.is_contiguous()
        d, n, K = self.d, self.n, self.K
        for t in small:
            if tuple(t[0].shape) != (d, d) or tuple(t[2].shape) != (d, d):
                raise ValueError("NgcfEngine: every layer must map d -> d (layer_size = [d] * K)")
        self.D = (K + 1) * d
        self.slope, self.p = float(slope), [float(x) for x in mess_dropout]
        self.reg_lambda, self.lr, self.betas, self.eps = float(reg_lambda), float(lr), betas, float(eps)
        self.store_grad = bool(store_grad)
        dev = params.device
        self.device = dev
        f32 = dict(dtype=torch.float32, device=dev)
        self.P = params
        per = 2 * d * d + 2 * d
        self.SW, self.SG = torch.empty(K * per, **f32), torch.zeros(K * per, **f32)
        self.SM, self.SV = torch.zeros(K * per, **f32), torch.zeros(K * per, **f32)
        self._views, self._gviews = [], []
        for l, (wg, bg, wb, bb) in enumerate(small):
            o = l * per
            cuts = [(o, (d, d)), (o + d * d, (1, d)), (o + d * d + d, (d, d)), (o + 2 * d * d + d, (1, d))]
            vs = [self.SW[a:a + s[0] * s[1]].view(s) for a, s in cuts]
            for v, src in zip(vs, (wg, bg, wb, bb)):
                v.copy_(src)
            self._views.append(tuple(vs))
            self._gviews.append(tuple(self.SG[a:a + s[0] * s[1]].view(s) for a, s in cuts))
        panel = lambda: torch.empty((n, d), **f32)  # noqa: E731
        # d = 64: one kernel per layer and direction (idg_ngcf_layer_fwd_f32 / _bwd_f32: S, ego * side and gT never reach
        # memory); other widths: the transform / tail / parameter-gradient chain.  IDG_NGCF_LAYER=0 forces the chain.
        self.fused_layer = d == 64 and os.environ.get("IDG_NGCF_LAYER", "1") != "0"
        self.SIDE = [panel() for _ in range(K)]
        self.E = [panel() for _ in range(K)]
        if not self.fused_layer:
            self.BI = [panel() for _ in range(K)]
            self.S = panel()
            self.gT = panel()
        self.FINAL = torch.empty((n, self.D), **f32)
        self.GFIN = torch.empty((n, self.D), **f32)
        self.GE = torch.empty((n, d), **f32)
        self.g_side = panel()
        self.g_ego = [panel(), panel()]
        self.GRAD = panel()
        self.M, self.V = torch.zeros((n, d), **f32), torch.zeros((n, d), **f32)
        self.prep = BatchPrep(self.U, n, d, dev)  # the batch's row bitmap and scatter plan: side stream, one batch ahead
        ws_bytes = lib.idg_ngcf_layer_bwd_workspace_bytes(d) if self.fused_layer else lib.idg_ngcf_wgrad_workspace_bytes(d, d)
        self.wg_ws = torch.empty(int(ws_bytes), dtype=torch.uint8, device=dev)
        self._per = per
        self.loss = torch.zeros(2, **f32)
        self.step_count = 0
        self._streams = None

    def small_views(self):
        return self._views

    def small_grads(self):
        return self._gviews

    @staticmethod
    def _p(t):
        return None if t is None else C.c_void_p(t.data_ptr())

    # ---- forward (every row); returns the [n, (K+1) d] final panel
    @torch.no_grad()
    def forward(self, streams=None, train=False):
        n, d, D, K, p_ = self.n, self.d, self.D, self.K, self._p
        st = ops._stream()
        self._Gf = self._Gb = self.G
        if train and self.node_keep is not None:
            # (drawn BEFORE the layers' dropout streams, as models.NGCF.aggregate() does: same sequence of draws)
            self._dropped = self.G.dropout_copy(self.node_keep, reuse=self._dropped)
            self._Gf, self._Gb = self._dropped, self._dropped._T
        self._streams = streams if streams is not None else [ops._next_noise_stream() for _ in range(K)]
        check(lib.idg_copy_cols_f32(p_(self.FINAL), D, p_(self.P), d, n, d, st), "idg_copy_cols_f32")
        ego = self.P
        for l in range(K):
            wg, bg, wb, bb = self._views[l]
            self._Gf.spmm_raw(ego, out=self.SIDE[l])
            seed, sid = self._streams[l]
            slot = self.FINAL.data_ptr() + 4 * (l + 1) * d
            if self.fused_layer:
                check(lib.idg_ngcf_layer_fwd_f32(p_(self.SIDE[l]), p_(ego), p_(wg), p_(wb), p_(bg), p_(bb), n, d, self.slope,
                                                 self.p[l], C.c_uint64(seed), C.c_uint64(sid), p_(self.E[l]), C.c_void_p(slot), D,
                                                 st), "idg_ngcf_layer_fwd_f32")
            else:
                check(lib.idg_ngcf_transform_f32(p_(self.SIDE[l]), p_(ego), p_(wg), p_(wb), n, d, d, p_(self.S), p_(self.BI[l]),
                                                 st), "idg_ngcf_transform_f32")
                check(lib.idg_ngcf_tail_ex_f32(p_(self.S), None, p_(bg), p_(bb), n, d, self.slope, self.p[l], C.c_uint64(seed),
                                               C.c_uint64(sid), p_(self.E[l]), C.c_void_p(slot), D, st), "idg_ngcf_tail_ex_f32")
            ego = self.E[l]
        return self.FINAL

    # ---- one training step: losses [bpr, reg_lambda * reg]
    @torch.no_grad()
    def train_step(self, users, pos, neg, loss_out=None, streams=None):
        n, d, D, K, U, p_ = self.n, self.d, self.D, self.K, self.U, self._p
        st = ops._stream()
        B = int(users.shape[0])
        loss = self.loss if loss_out is None else loss_out
        slot = self.prep.take(users, pos, neg)
        bitmap = slot.bitmap
        self.forward(streams, train=True)
        check(lib.idg_bpr_fused_ex_f32(p_(self.FINAL), D, p_(self.P), d, U, n, p_(users), p_(pos), p_(neg), B, self.reg_lambda, 0,
                                       p_(loss), p_(self.GFIN), p_(self.GE), native.IDG_BPR_PLANNED | native.IDG_BPR_TOUCHED_PRESET,
                                       p_(bitmap), p_(slot.ws), st), "idg_bpr_fused_ex_f32")
        # backward
        self.step_count += 1
        gE = None
        for l in range(K - 1, -1, -1):
            wg, bg, wb, bb = self._views[l]
            seed, sid = self._streams[l]
            slot_ptr = self.GFIN.data_ptr() + 4 * (l + 1) * d
            ego_prev = self.P if l == 0 else self.E[l - 1]
            g_ego = self.g_ego[l & 1]
            w_grads = C.c_void_p(self.SG.data_ptr() + 4 * l * self._per)  # the layer's four, straight into the flat buffer
            if self.fused_layer:
                check(lib.idg_ngcf_layer_bwd_f32(p_(self.E[l]), p_(gE), C.c_void_p(slot_ptr), D, p_(bitmap), p_(self.SIDE[l]),
                                                 p_(ego_prev), p_(wg), p_(wb), n, d, self.slope, self.p[l], C.c_uint64(seed),
                                                 C.c_uint64(sid), p_(self.g_side), p_(g_ego), w_grads, p_(self.wg_ws), st),
                      "idg_ngcf_layer_bwd_f32")
            else:
                check(lib.idg_ngcf_tail_bwd_ex_f32(p_(self.E[l]), p_(gE), C.c_void_p(slot_ptr), D, p_(bitmap), n, d, self.slope,
                                                   self.p[l], C.c_uint64(seed), C.c_uint64(sid), p_(self.gT), st),
                      "idg_ngcf_tail_bwd_ex_f32")
                # the four parameter gradients in one pass over (side, ego * side, gT)
                check(lib.idg_ngcf_wgrad_f32(p_(self.SIDE[l]), p_(self.BI[l]), p_(self.gT), n, d, d, w_grads, p_(self.wg_ws), st),
                      "idg_ngcf_wgrad_f32")
                check(lib.idg_ngcf_transform_bwd_f32(p_(self.gT), p_(self.SIDE[l]), p_(ego_prev), p_(wg), p_(wb), n, d, d,
                                                     p_(self.g_side), p_(g_ego), st), "idg_ngcf_transform_bwd_f32")
            if l > 0:
                nxt = self.g_ego[(l & 1) ^ 1]
                self._Gb.spmm_raw(self.g_side, addend=g_ego, out=nxt)   # the backward of side = G . ego: G^T . g_side (G symmetric; a dropped copy: its transposed one)
                gE = nxt
            else:
                # the batch's rows of slot 0 (ego_0 is itself part of the final rows) and of the regulariser's gradient
                check(lib.idg_rows_add2_f32(p_(g_ego), d, p_(self.GFIN), D, p_(self.GE), d, p_(bitmap), n, d, st),
                      "idg_rows_add2_f32")
                ops.spmm_epi_raw(self._Gb, self.g_side, addend=g_ego, sum_out=self.GRAD,
                                 adam=(self.P, self.M, self.V, self.lr, self.step_count, self.betas[0], self.betas[1], self.eps),
                                 adam_discard_grad=not self.store_grad)
        ops.adam_step_raw(self.SW, self.SG, self.SM, self.SV, self.lr, self.step_count, self.betas[0], self.betas[1], self.eps)
        self.prep.release(slot)
        return loss

    def prefetch(self, users, pos, neg):
        """One-batch lookahead of the index-only work of the NEXT step (side stream)."""
        self.prep.prefetch(users, pos, neg)
