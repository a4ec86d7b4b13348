"""EGCF (Zhang et al., TOIS'24: embedding-less graph collaborative filtering) on MI355X — reference:
models/EGCF.py.  Only the item table is learned; users are tanh(R_hat . E_item) with the rectangular
normalised interaction matrix R_hat = D_u^-1/2 R D_i^-1/2 (data_graph.sparse_adjacency_matrix_R).

Two encoders (config `mode`):
  parallel     user_0 = tanh(R_hat E_i); then K layers X <- tanh(A_hat X) on the full bipartite adjacency, summed
  alternating  K rounds of  user = tanh(R_hat item),  item = tanh(R_hat^T user),  each side summed over the rounds
Loss: BPR + reg on the two item ego rows + three in-batch InfoNCE terms over the RAW batch rows (user-user,
positive-positive, user-positive; no torch.unique here either).

Every sparse product is the library's SpMM operator: R_hat is a rectangular ops.Graph whose transposed handle
serves both the R_hat^T products and the backward of the R_hat ones; A_hat is the symmetric handle the other
models use.  Evaluation goes through the fused score/mask/top-K kernel."""
import torch
from torch import nn

import utility.utility_data.data_graph as data_graph
import utility.utility_function.losses as losses
import utility.utility_function.tools as tools
import utility.utility_train.trainer as trainer
from idgrec_amd import ops
from idgrec_amd.egcf import EgcfAltEngine, EgcfEngine


class EGCF(nn.Module):
    #: both encoders train through fused, autograd-free chains of library calls (idgrec_amd/egcf.py: EgcfEngine for
    #: `parallel`, EgcfAltEngine for `alternating`); the differentiable operators below serve foreign optimizers
    supports_fused_step = True
    n_fused_losses = 3
    FUSED_WIDTHS = (32, 64, 128, 256, 512)

    def __init__(self, config, dataset, device):
        super(EGCF, self).__init__()
        self.config, self.dataset, self.device = config, dataset, device
        self.reg_lambda = float(config['reg_lambda'])
        self.ssl_lambda = float(config['ssl_lambda'])
        self.temperature = float(config['temperature'])
        self.aggregate_mode = config['mode']
        self.n_layers = int(config['GCN_layer'])
        self.user_embedding = None  # embedding-less on the user side
        self.item_embedding = nn.Embedding(num_embeddings=dataset.num_items, embedding_dim=int(config['embedding_size']))
        nn.init.xavier_uniform_(self.item_embedding.weight, gain=1)
        dev = torch.device(device)
        if dev.type != "cuda":
            raise RuntimeError("EGCF needs an MI355X: device is %s and idgrec_amd has no CPU path." % dev)
        # rectangular [U, I] operator (+ its transpose, built once on the host)
        self.user_Graph = tools.convert_sp_mat_to_graph(data_graph.sparse_adjacency_matrix_R(dataset), dev, symmetric=False)
        self.Graph = None
        if self.aggregate_mode == 'parallel':
            self.Graph = tools.convert_sp_mat_to_graph(data_graph.sparse_adjacency_matrix(dataset), dev)
        self.activation_layer = nn.Tanh()
        self.activation = nn.Sigmoid()
        self._eval_cache = None
        self._engine = None

    def train(self, mode=True):
        self._eval_cache = None
        return super().train(mode)

    # ------------------------------------------------------------------ fused path (trainer protocol)
    def fused_step_available(self):
        w = self.item_embedding.weight
        return (w.is_cuda and w.dtype == torch.float32 and int(w.shape[1]) in self.FUSED_WIDTHS and 1 <= self.n_layers <= 4)

    def engine(self):
        """The fused engine; the item table lives in ITS storage (nn.Embedding.weight is re-pointed at it), so that the
        regulariser's [n, d] view of the parameters and the table are one buffer."""
        w = self.item_embedding.weight
        if self._engine is None or w.data_ptr() != self._engine.item_table().data_ptr():
            if self.aggregate_mode == 'parallel':
                self._engine = EgcfEngine(self.Graph, self.user_Graph, self.dataset.num_users, self.dataset.num_items,
                                          int(w.shape[1]), self.n_layers, w.data, self.reg_lambda, self.ssl_lambda,
                                          self.temperature)
            else:
                self._engine = EgcfAltEngine(self.user_Graph, self.dataset.num_users, self.dataset.num_items, int(w.shape[1]),
                                             self.n_layers, w.data, self.reg_lambda, self.ssl_lambda, self.temperature)
            w.data = self._engine.item_table()
        return self._engine

    def prefetch_batch(self, users, pos, neg):
        """The trainer's one-batch lookahead: the next batch's bitmap / live units / scatter plan on the side stream."""
        if self.fused_step_available():
            self.engine().prefetch(users, pos, neg)

    def fused_train_step(self, users, pos, neg, loss_out, optimizer):
        """forward + backward + Adam as ONE chain of kernels; False (nothing done) unless `optimizer` is an
        idgrec_amd.ops.Adam over exactly the item table.  The optimizer's state stays the single source of truth."""
        w = self.item_embedding.weight
        if not isinstance(optimizer, ops.Adam) or len(optimizer.param_groups) != 1:
            return False
        group = optimizer.param_groups[0]
        if len(group["params"]) != 1 or group["params"][0] is not w:
            return False
        eng = self.engine()
        st = optimizer.state[w]
        if not st or st["exp_avg"].data_ptr() != eng.M.data_ptr():
            if st:  # the optimizer has already stepped the other way: keep what it accumulated
                eng.M.copy_(st["exp_avg"])
                eng.V.copy_(st["exp_avg_sq"])
            st.setdefault("step", 0)
            st["exp_avg"], st["exp_avg_sq"] = eng.M, eng.V
        eng.lr, eng.betas, eng.eps = float(group["lr"]), tuple(group["betas"]), float(group["eps"])
        eng.step_count = int(st["step"])
        self._eval_cache = None
        eng.train_step(users, pos, neg, loss_out)
        st["step"] = eng.step_count
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

    def alternating_aggregate(self):
        item = self.item_embedding.weight
        users, items = None, None
        for _ in range(self.n_layers):
            user = torch.tanh(ops.spmm(self.user_Graph, item))
            item = torch.tanh(ops.spmm(self.user_Graph.T, user))
            users = user if users is None else users + user
            items = item if items is None else items + item
        return users, items

    def parallel_aggregate(self):
        item = self.item_embedding.weight
        user = torch.tanh(ops.spmm(self.user_Graph, item))
        x = torch.cat([user, item])
        total = None
        for _ in range(self.n_layers):
            x = torch.tanh(ops.spmm(self.Graph, x))
            total = x if total is None else total + x
        return torch.split(total, [self.dataset.num_users, self.dataset.num_items])

    def aggregate(self):
        return self.parallel_aggregate() if self.aggregate_mode == 'parallel' else self.alternating_aggregate()

    def forward(self, user, positive, negative):
        all_user, all_item = self.aggregate()
        user_e, pos_e, neg_e = all_user[user.long()], all_item[positive.long()], all_item[negative.long()]
        bpr_loss = losses.get_bpr_loss(user_e, pos_e, neg_e)
        reg_loss = self.reg_lambda * losses.get_reg_loss(self.item_embedding(positive), self.item_embedding(negative))
        ssl = losses.get_InfoNCE_loss(user_e, user_e, self.temperature) \
            + losses.get_InfoNCE_loss(pos_e, pos_e, self.temperature) \
            + losses.get_InfoNCE_loss(user_e, pos_e, self.temperature)
        return [bpr_loss, reg_loss, self.ssl_lambda * ssl]

    def final_panels(self):
        if self._eval_cache is None:
            with torch.no_grad():
                if self._engine is not None and self.fused_step_available() \
                        and self.item_embedding.weight.data_ptr() == self._engine.item_table().data_ptr():
                    self._eval_cache = self._engine.propagate()  # the same chain the training step runs
                else:
                    u, i = self.aggregate()
                    self._eval_cache = (u.contiguous(), i.contiguous())
        return self._eval_cache

    def get_rating_for_test(self, user):
        """sigmoid(E_u[user] . E_i^T) as a dense [B, num_items] matrix (models/EGCF.py:113-121)."""
        with torch.no_grad():
            ue, ie = self.final_panels()
            return ops.score_dense(ue, ie, user.long(), apply_sigmoid=True)

    def topk_for_test(self, user, k):
        with torch.no_grad():
            ue, ie = self.final_panels()
            ip, ix = self.dataset.train_csr_on(ue.device)
            return ops.score_topk(ue, ie, user.long(), k, ip, ix, apply_sigmoid=True)


class Trainer():
    def __init__(self, args, config, dataset, device, logger):
        self.model = EGCF(config, dataset, device)
        self.args, self.config, self.dataset = args, config, dataset
        self.device, self.logger = device, logger

    def train(self):
        trainer.universal_trainer(self.model, self.args, self.config, self.dataset, self.device, self.logger)
