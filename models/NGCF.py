"""NGCF (Wang et al., SIGIR'19) on MI355X.  The sparse product with the self-loop adjacency
D^-1/2 (A + I) D^-1/2 runs on the library's SpMM operator (symmetric graph: backward reuses the
handle); the per-layer d x d transforms, LeakyReLU, message dropout, L2 normalisation and layer
concatenation are stock torch ops as in the reference (models/NGCF.py:67-111).  Regularisation
covers the positive and negative item rows only (models/NGCF.py:125)."""
import torch
from torch import nn

import utility.utility_data.data_graph as data_graph
import utility.utility_function.losses as losses
import utility.utility_train.trainer as trainer
from idgrec_amd import ops
from idgrec_amd.modeling import PackedRecommender


class NGCF(PackedRecommender):
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
        if eval(config['node_dropout']):
            raise NotImplementedError("NGCF node_dropout re-samples the sparse graph every forward; only the shipped "
                                      "configuration (node_dropout = False) is supported on the MI355X path.")
        self.attach_graph(data_graph.sparse_adjacency_matrix_with_self(dataset))
        self.activation_layer = nn.Tanh()

    def aggregate(self):
        ego = self.ego_panel()
        layers = [ego]
        for layer in range(self.n_layers):
            side = ops.spmm(self.Graph, ego)
            w = self.weight_dict
            # [n, d] x [d, d]: forward / input gradient are small GEMMs, the weight gradient (all reduction over the
            # n rows) is the library's slice-summed kernel
            s1 = ops.tall_linear(side, w['W_gcn_%d' % layer])
            s2 = ops.tall_linear(torch.mul(ego, side), w['W_bi_%d' % layer])
            # bias adds, LeakyReLU(0.2), message dropout and the L2-normalised copy in one kernel.  The reference
            # instantiates nn.Dropout inside aggregate() (models/NGCF.py:104): a fresh module is always in training
            # mode, so message dropout is applied during evaluation as well — kept as is
            ego, normed = ops.ngcf_layer_tail(s1, s2, w['b_gcn_%d' % layer], w['b_bi_%d' % layer], 0.2,
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
