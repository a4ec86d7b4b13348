"""XSimGCL (Yu et al.) on MI355X: ONE perturbed propagation per step serves both the
recommendation loss and the contrast between layer `cl_layer` and the final mean (reference:
models/XSimGCL.py).  Layers chain the SpMM operator with the noise
`X += sign(X) * normalize(U[0,1)) * eps` in between; evaluation uses the unperturbed encoder
(mean of layers 1..K), which is the fused propagate-mean operator."""
import torch

import utility.utility_data.data_graph as data_graph
import utility.utility_train.trainer as trainer
from idgrec_amd import ops
from idgrec_amd.modeling import PackedRecommender


class XSimGCL(PackedRecommender):
    include_layer0 = False

    def __init__(self, config, dataset, device):
        super(XSimGCL, self).__init__(config, dataset, device)
        self.n_layers = int(config['GCN_layer'])
        self.ssl_lambda = float(config['ssl_lambda'])
        self.epsilon = float(config['epsilon'])
        self.temperature = float(config['temperature'])
        self.cl_layer = int(config['cl_layer'])
        self.attach_graph(data_graph.sparse_adjacency_matrix(dataset))
        # the trainer's fused step (PropagationEngine with .xssl set: perturbed pass restricted to the batch's rows in
        # its last layer, BPR, fused InfoNCE, backward + one extra sparse product for the view's gradient) covers the
        # shipped configuration, cl_layer = 1; forward() below is the same computation under autograd for any cl_layer
        self.supports_fused_step = self.cl_layer == 1
        self.n_fused_losses = 3

    def engine(self):
        eng = super().engine()
        eng.xssl = (self.epsilon, self.temperature, self.ssl_lambda)
        return eng

    def aggregate(self, perturbed=False):
        U, I = self.dataset.num_users, self.dataset.num_items
        ego = self.ego_panel()
        if not perturbed:
            return torch.split(ops.propagate_mean(self.Graph, ego, self.n_layers, include_layer0=False), [U, I])
        x, total, view = ego, None, ego
        pass_stream = ops._next_noise_stream()  # one noise stream per pass; layer k draws from its k-th sub-stream
        for layer in range(self.n_layers):
            # SpMM + fused noise epilogue
            x = ops.spmm_perturbed(self.Graph, x, self.epsilon, stream=ops.layer_noise_stream(pass_stream, layer + 1))
            total = x if total is None else total + x
            if layer == self.cl_layer - 1:
                view = x
        final = total / float(self.n_layers)
        self._view_panels = (view, final)  # the [n, d] panels behind the four splits, for the fused InfoNCE
        return torch.split(final, [U, I]) + torch.split(view, [U, I])

    def forward(self, user, positive, negative):
        ego = self.ego_panel()
        self.aggregate(perturbed=True)
        view, final = self._view_panels  # cl-layer view and layer mean, [n, d] each (users first)
        bpr_loss, reg_loss = ops.bpr_loss(final, ego, user, positive, negative, self.dataset.num_users,
                                          self.reg_lambda)
        # InfoNCE(cl-layer view, final view) over unique(user) and unique(positive) rows (models/XSimGCL.py:80-86)
        ssl = ops.infonce_pair(view, final, user, positive, self.dataset.num_users, self.temperature)
        return [bpr_loss, reg_loss, self.ssl_lambda * ssl]


class Trainer():
    def __init__(self, args, config, dataset, device, logger):
        self.model = XSimGCL(config, dataset, device)
        self.args, self.config, self.dataset = args, config, dataset
        self.device, self.logger = device, logger

    def train(self):
        trainer.universal_trainer(self.model, self.args, self.config, self.dataset, self.device, self.logger)
