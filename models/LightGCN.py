"""LightGCN (He et al., SIGIR'20) on MI355X.

Plugin surface identical to the reference's models/LightGCN.py: class `LightGCN(nn.Module)`
with `aggregate()`, `forward(user, positive, negative) -> [bpr_loss, reg_loss]`,
`get_rating_for_test(user)`, and a `Trainer(args, config, dataset, device, logger)` with
`.train()`.  The arithmetic runs in libidgrec.so: K-layer propagation + layer mean is one
fused operator (idg_propagate_mean_f32), gather + BPR + L2-reg is another (idg_bpr_*).
"""
import torch

import utility.utility_data.data_graph as data_graph
import utility.utility_train.trainer as trainer
from idgrec_amd import ops
from idgrec_amd.modeling import PackedRecommender


class LightGCN(PackedRecommender):
    include_layer0 = True       # E0 takes part in the layer mean (models/LightGCN.py:41-48)
    supports_fused_step = True  # loss == [bpr, reg] over the mean-propagated panel

    def __init__(self, config, dataset, device):
        super(LightGCN, self).__init__(config, dataset, device)
        self.n_layers = int(config['GCN_layer'])
        self.attach_graph(data_graph.sparse_adjacency_matrix(dataset))

    def aggregate(self):
        """(users [U,d], items [I,d]) = mean over layers 0..K of A^k E0."""
        final = ops.propagate_mean(self.Graph, self.ego_panel(), self.n_layers, include_layer0=True)
        return torch.split(final, [self.dataset.num_users, self.dataset.num_items])

    def forward(self, user, positive, negative):
        ego = self.ego_panel()
        final = ops.propagate_mean(self.Graph, ego, self.n_layers, include_layer0=True)
        bpr_loss, reg_loss = ops.bpr_loss(final, ego, user, positive, negative, self.dataset.num_users,
                                          self.reg_lambda)
        return [bpr_loss, reg_loss]


class Trainer():
    def __init__(self, args, config, dataset, device, logger):
        self.model = LightGCN(config, dataset, device)
        self.args, self.config, self.dataset = args, config, dataset
        self.device, self.logger = device, logger

    def train(self):
        trainer.universal_trainer(self.model, self.args, self.config, self.dataset, self.device, self.logger)
