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
        self.node_dropout = bool(eval(config['node_dropout']))
        self.node_keep_prob = float(config['node_keep_prob']) if self.node_dropout else 1.0
        self.attach_graph(data_graph.sparse_adjacency_matrix_with_self(dataset))
        self.activation_layer = nn.Tanh()

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
