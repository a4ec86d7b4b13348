"""SimGCL (Yu et al., SIGIR'22) on MI355X: LightGCN encoder without layer 0 in the mean, two
noise-perturbed views per step, InfoNCE between the views (reference: models/SimGCL.py).

The clean view is the fused propagate-mean operator; each perturbed view is the same chain with
the noise `X += sign(X) * normalize(U[0,1)) * eps` (models/SimGCL.py:49-51) fused into every
layer's SpMM epilogue (Philox4x32-10, seeded from torch's device seed).  Like the reference's
device generator it agrees with a CPU run statistically, not bit for bit.  The three passes
share one backward propagation (identical Jacobian).
"""
import torch

import utility.utility_data.data_graph as data_graph
import utility.utility_train.trainer as trainer
from idgrec_amd import ops
from idgrec_amd.modeling import PackedRecommender


class SimGCL(PackedRecommender):
    include_layer0 = False  # "Initial embedding is not included" (models/SimGCL.py:44-45)
    # the trainer's fused step: clean + two perturbed passes, BPR, InfoNCE and the one shared backward propagation
    # with Adam in its epilogue as a fixed chain of kernels (PropagationEngine with .ssl set); forward() below is
    # the same computation under autograd
    supports_fused_step = True
    n_fused_losses = 3

    def __init__(self, config, dataset, device):
        super(SimGCL, self).__init__(config, dataset, device)
        self.n_layers = int(config['GCN_layer'])
        self.ssl_lambda = float(config['ssl_lambda'])
        self.epsilon = float(config['epsilon'])
        self.temperature = float(config['temperature'])
        self.attach_graph(data_graph.sparse_adjacency_matrix(dataset))

    def engine(self):
        eng = super().engine()
        eng.ssl = (self.epsilon, self.temperature, self.ssl_lambda)
        return eng

    def aggregate(self, perturbed=False):
        ego = self.ego_panel()
        if not perturbed:
            final = ops.propagate_mean(self.Graph, ego, self.n_layers, include_layer0=False)
        else:
            final = ops.propagate_views(self.Graph, ego, self.n_layers, False, self.epsilon, n_views=1)[1]
        return torch.split(final, [self.dataset.num_users, self.dataset.num_items])

    def forward(self, user, positive, negative):
        U, I = self.dataset.num_users, self.dataset.num_items
        ego = self.ego_panel()
        # clean pass + two perturbed passes (models/SimGCL.py:63-65): noise fused into the SpMM epilogue,
        # one shared backward propagation
        clean, view_1, view_2 = ops.propagate_views(self.Graph, ego, self.n_layers, False, self.epsilon, n_views=2)
        bpr_loss, reg_loss = ops.bpr_loss(clean, ego, user, positive, negative, U, self.reg_lambda)
        # InfoNCE over unique(user) rows + over unique(positive) rows of the two views (models/SimGCL.py:79-84):
        # one fused forward+backward operator on the [n, d] panels instead of ~100 small launches
        ssl = ops.infonce_pair(view_1, view_2, user, positive, U, self.temperature)
        return [bpr_loss, reg_loss, self.ssl_lambda * ssl]


class Trainer():
    def __init__(self, args, config, dataset, device, logger):
        self.model = SimGCL(config, dataset, device)
        self.args, self.config, self.dataset = args, config, dataset
        self.device, self.logger = device, logger

    def train(self):
        trainer.universal_trainer(self.model, self.args, self.config, self.dataset, self.device, self.logger)
