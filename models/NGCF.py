"""NGCF (Wang et al., SIGIR'19) on MI355X.  The sparse product with the self-loop adjacency
D^-1/2 (A + I) D^-1/2 runs on the library's SpMM operator (symmetric graph: backward reuses the
handle; with node_dropout a re-drawn masked copy of the handle per training forward); the two
per-layer d x d transforms run on the fp32 matrix cores in one pass (idg_ngcf_transform_f32), bias adds,
LeakyReLU, message dropout and L2 normalisation in one more (idg_ngcf_tail_f32) (models/NGCF.py:67-111).  Regularisation
covers the positive and negative item rows only (models/NGCF.py:125)."""
import torch
from torch import nn

import utility.utility_data.data_graph as data_graph
import utility.utility_function.losses as losses
import utility.utility_train.trainer as trainer
from idgrec_amd import ops
from idgrec_amd.modeling import PackedRecommender
from idgrec_amd.ngcf import NgcfEngine


class NGCF(PackedRecommender):
    #: trains through a fused, autograd-free chain of library calls (idgrec_amd/ngcf.py) when every layer maps d -> d (with or
    #: without node dropout); otherwise through the differentiable operators below
    supports_fused_step = True
    n_fused_losses = 2

    def __init__(self, config, dataset, device):
        super(NGCF, self).__init__(config, dataset, device)
        self.n_layers = int(config['GCN_layer'])
        widths = [int(config['embedding_size'])] + eval(config['layer_size'])
        self.weight_dict = nn.ParameterDict()
        for layer in range(self.n_layers):  # creation order = the reference's torch RNG order
            for name, shape in (('W_gcn_%d', (widths[layer], widths[layer + 1])), ('b_gcn_%d', (1, widths[layer + 1])),
                                ('W_bi_%d', (widths[layer], widths[layer + 1])), ('b_bi_%d', (1, widths[layer + 1]))):
                self.weight_dict[name % layer] = nn.Parameter(nn.init.xavier_uniform_(torch.empty(*shape)))
        if eval(config['mess_dropout']):
            self.mess_dropout = eval(config['mess_drop_prob'])
        self.node_dropout = bool(eval(config['node_dropout']))
        self.node_keep_prob = float(config['node_keep_prob']) if self.node_dropout else 1.0
        self.attach_graph(data_graph.sparse_adjacency_matrix_with_self(dataset))
        self.activation_layer = nn.Tanh()

    # ------------------------------------------------------------------ fused path (trainer protocol)
    def fused_step_available(self):
        st = self._storage
        if st is None or not st.is_cuda or not hasattr(self, "mess_dropout"):
            return False
        d = int(st.shape[1])
        return d % 64 == 0 and d in self.FUSED_WIDTHS and all(
            tuple(self.weight_dict['W_gcn_%d' % l].shape) == (d, d) for l in range(self.n_layers))

    def ngcf_engine(self):
        """The fused engine over the packed embedding panel; the 4K small tensors move into ITS flat buffer (the
        nn.Parameters are re-pointed at views of it), so one Adam launch updates all of them."""
        if not self._is_packed():
            self._pack()
        eng = getattr(self, "_ngcf_engine", None)
        w = self.weight_dict
        names = [('W_gcn_%d', 'b_gcn_%d', 'W_bi_%d', 'b_bi_%d') for _ in range(self.n_layers)]
        if eng is None or eng.P.data_ptr() != self._storage.data_ptr() \
                or w['W_gcn_0'].data_ptr() != eng.small_views()[0][0].data_ptr():
            small = [tuple(w[nm % l].data for nm in names[l]) for l in range(self.n_layers)]
            eng = self._ngcf_engine = NgcfEngine(self.Graph, self.dataset.num_users, self.dataset.num_items, self._storage, small,
                                                 slope=0.2, mess_dropout=self.mess_dropout, reg_lambda=self.reg_lambda,
                                                 node_keep_prob=self.node_keep_prob if self.node_dropout else None)
            for l in range(self.n_layers):
                for nm, v in zip(names[l], eng.small_views()[l]):
                    w[nm % l].data = v
        return eng

    def prefetch_batch(self, users, pos, neg):
        """The trainer's one-batch lookahead: the next batch's bitmap / scatter plan on the side stream."""
        if self.fused_step_available():
            self.ngcf_engine().prefetch(users, pos, neg)

    def fused_train_step(self, users, pos, neg, loss_out, optimizer):
        """forward + backward + every Adam update as ONE chain of kernels; False (nothing done) unless `optimizer` is an
        idgrec_amd.ops.Adam over exactly this model's parameters.  Its state stays the single source of truth."""
        params = list(self.parameters())
        if not isinstance(optimizer, ops.Adam) or len(optimizer.param_groups) != 1:
            return False
        group = optimizer.param_groups[0]
        if len(group["params"]) != len(params) or any(a is not b for a, b in zip(group["params"], params)):
            return False
        eng = self.ngcf_engine()
        uw, iw = self.user_embedding.weight, self.item_embedding.weight
        U = self.dataset.num_users
        w = self.weight_dict
        # (parameter, first-moment view, second-moment view): the two tables in the packed panels, the small tensors in the
        # flat buffers
        homes = [(uw, eng.M[:U], eng.V[:U]), (iw, eng.M[U:], eng.V[U:])]
        per = 2 * eng.d * eng.d + 2 * eng.d
        for l in range(self.n_layers):
            o = l * per
            for nm, a, shape in (('W_gcn_%d', o, (eng.d, eng.d)), ('b_gcn_%d', o + eng.d * eng.d, (1, eng.d)),
                                 ('W_bi_%d', o + eng.d * eng.d + eng.d, (eng.d, eng.d)),
                                 ('b_bi_%d', o + 2 * eng.d * eng.d + eng.d, (1, eng.d))):
                cnt = shape[0] * shape[1]
                homes.append((w[nm % l], eng.SM[a:a + cnt].view(shape), eng.SV[a:a + cnt].view(shape)))
        steps = set()
        for prm, m, v in homes:
            st = optimizer.state[prm]
            if not st or st["exp_avg"].data_ptr() != m.data_ptr():
                if st:  # the optimizer has already stepped the other way: keep what it accumulated
                    m.copy_(st["exp_avg"])
                    v.copy_(st["exp_avg_sq"])
                st.setdefault("step", 0)
                st["exp_avg"], st["exp_avg_sq"] = m, v
            steps.add(int(st["step"]))
        if len(steps) != 1:
            return False
        eng.lr, eng.betas, eng.eps = float(group["lr"]), tuple(group["betas"]), float(group["eps"])
        eng.step_count = steps.pop()
        self._eval_cache = None
        eng.train_step(users, pos, neg, loss_out)
        for prm, _, _ in homes:
            optimizer.state[prm]["step"] = eng.step_count
        return True

    def fused_loss_and_grad(self, users, pos, neg, loss_out=None):
        """The trainer's fallback when the optimizer is not ours: the differentiable operators under autograd."""
        self._eval_cache = None
        ll = self.forward(users, pos, neg)
        self.zero_grad()
        sum(ll).backward()
        out = torch.stack([x.detach() for x in ll])
        if loss_out is not None:
            loss_out.copy_(out)
        return out

    def aggregate(self):
        ego = self.ego_panel()
        layers = [ego]
        # node dropout (models/NGCF.py:56-65, 73-79): ONE re-drawn edge mask per training forward, shared by the layers;
        # on the device it is a masked copy of the handle (same tile schedule; backward multiplies by the transposed mask)
        graph = self.Graph
        if self.node_dropout and self.training:  # (the pair of masked entry lists is allocated once and redrawn in place)
            graph = self._dropped = self.Graph.dropout_copy(self.node_keep_prob, reuse=getattr(self, "_dropped", None))
        for layer in range(self.n_layers):
            side = ops.spmm(graph, ego)
            w = self.weight_dict
            # side . W_gcn + (ego * side) . W_bi (models/NGCF.py:88-99): both [n, d] x [d, d] products in ONE pass over
            # the rows on the fp32 matrix cores, into one accumulator; input gradients likewise, weight gradients (all
            # reduction over the n rows) by the library's slice-summed kernel
            s = ops.ngcf_transform(side, ego, w['W_gcn_%d' % layer], w['W_bi_%d' % layer])
            # bias adds, LeakyReLU(0.2), message dropout and the L2-normalised copy in one kernel.  The reference
            # instantiates nn.Dropout inside aggregate() (models/NGCF.py:104): a fresh module is always in training
            # mode, so message dropout is applied during evaluation as well — kept as is
            ego, normed = ops.ngcf_layer_tail(s, None, w['b_gcn_%d' % layer], w['b_bi_%d' % layer], 0.2,
                                              self.mess_dropout[layer])
            layers.append(normed)
        final = torch.cat(layers, dim=1)
        return torch.split(final, [self.dataset.num_users, self.dataset.num_items])

    def forward(self, user, positive, negative):
        all_user, all_item = self.aggregate()
        bpr_loss = losses.get_bpr_loss(all_user[user.long()], all_item[positive.long()], all_item[negative.long()])
        # get_reg_loss(item_embedding(positive), item_embedding(negative)) (models/NGCF.py:125; losses.py:16-21):
        # sum_b 1/2 ||W[pos_b]||^2 + 1/2 ||W[neg_b]||^2 over the batch = sum_i count_i * 1/2 ||W_i||^2, which autograd
        # differentiates as one dense product instead of two sort-based embedding backward passes
        W = self.item_embedding.weight
        count = torch.bincount(torch.cat([positive.long(), negative.long()]), minlength=W.shape[0]).to(W.dtype)
        reg_loss = 0.5 * (count * (W * W).sum(dim=1)).sum() / float(len(user))
        return [bpr_loss, self.reg_lambda * reg_loss]

    def _eval_panels(self):
        with torch.no_grad():
            users, items = self.aggregate()
            return users.contiguous(), items.contiguous()


class Trainer():
    def __init__(self, args, config, dataset, device, logger):
        self.model = NGCF(config, dataset, device)
        self.args, self.config, self.dataset = args, config, dataset
        self.device, self.logger = device, logger

    def train(self):
        trainer.universal_trainer(self.model, self.args, self.config, self.dataset, self.device, self.logger)
