"""Matrix-factorisation BPR baseline on MI355X — the plumbing + BPR configuration of
BASELINE.json (no graph convolution).  Same plugin surface as the reference's
models/MFBPR.py; loss and gradients come from the fused gather-BPR kernels with the raw
embedding panel standing in for both the scoring and the regularised rows
(models/MFBPR.py:29-42)."""
import utility.utility_train.trainer as trainer
from idgrec_amd import ops
from idgrec_amd.modeling import PackedRecommender


class MFBPR(PackedRecommender):
    n_layers = 0
    supports_fused_step = True

    def __init__(self, config, dataset, device):
        super(MFBPR, self).__init__(config, dataset, device)

    def forward(self, user, positive, negative):
        panel = self.ego_panel()
        bpr_loss, reg_loss = ops.bpr_loss(panel, panel, user, positive, negative, self.dataset.num_users,
                                          self.reg_lambda)
        return [bpr_loss, reg_loss]


class Trainer():
    def __init__(self, args, config, dataset, device, logger):
        self.model = MFBPR(config, dataset, device)
        self.args, self.config, self.dataset = args, config, dataset
        self.device, self.logger = device, logger

    def train(self):
        trainer.universal_trainer(self.model, self.args, self.config, self.dataset, self.device, self.logger)
