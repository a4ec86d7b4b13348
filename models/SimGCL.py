"""SimGCL (Yu et al., SIGIR'22) on MI355X: LightGCN encoder without layer 0 in the mean, two
noise-perturbed views per step, InfoNCE between the views (reference: models/SimGCL.py).

The clean view is the fused propagate-mean operator; each perturbed view chains the SpMM
operator K times with the noise `X += sign(X) * normalize(U[0,1)) * eps` applied between
layers (models/SimGCL.py:49-51).  The noise comes from the device generator, as in the
reference, so perturbed views agree with a CPU run statistically, not bit for bit.
"""
import torch

import utility.utility_data.data_graph as data_graph
import utility.utility_function.losses as losses
import utility.utility_train.trainer as trainer
from idgrec_amd import ops
from idgrec_amd.modeling import PackedRecommender


class SimGCL(PackedRecommender):
    include_layer0 = False  # "Initial embedding is not included" (models/SimGCL.py:44-45)

    def __init__(self, config, dataset, device):
        super(SimGCL, self).__init__(config, dataset, device)
        self.n_layers = int(config['GCN_layer'])
        self.ssl_lambda = float(config['ssl_lambda'])
        self.epsilon = float(config['epsilon'])
        self.temperature = float(config['temperature'])
        self.attach_graph(data_graph.sparse_adjacency_matrix(dataset))

    def aggregate(self, perturbed=False):
        ego = self.ego_panel()
        if not perturbed:
            final = ops.propagate_mean(self.Graph, ego, self.n_layers, include_layer0=False)
        else:
            x, total = ego, None
            for _ in range(self.n_layers):
                x = ops.spmm(self.Graph, x)
                noise = torch.nn.functional.normalize(torch.rand_like(x), dim=-1)
                x = x + torch.sign(x) * noise * self.epsilon
                total = x if total is None else total + x
            final = total / float(self.n_layers)
        return torch.split(final, [self.dataset.num_users, self.dataset.num_items])

    def forward(self, user, positive, negative):
        ego = self.ego_panel()
        clean = ops.propagate_mean(self.Graph, ego, self.n_layers, include_layer0=False)
        user_1, item_1 = self.aggregate(perturbed=True)
        user_2, item_2 = self.aggregate(perturbed=True)

        bpr_loss, reg_loss = ops.bpr_loss(clean, ego, user, positive, negative, self.dataset.num_users,
                                          self.reg_lambda)

        user_index = torch.unique(user)
        item_index = torch.unique(positive)
        ssl = losses.get_InfoNCE_loss(user_1[user_index], user_2[user_index], self.temperature) \
            + losses.get_InfoNCE_loss(item_1[item_index], item_2[item_index], self.temperature)
        return [bpr_loss, reg_loss, self.ssl_lambda * ssl]


class Trainer():
    def __init__(self, args, config, dataset, device, logger):
        self.model = SimGCL(config, dataset, device)
        self.args, self.config, self.dataset = args, config, dataset
        self.device, self.logger = device, logger

    def train(self):
        trainer.universal_trainer(self.model, self.args, self.config, self.dataset, self.device, self.logger)
