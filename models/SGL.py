"""SGL (Wu et al., SIGIR'21) on MI355X: LightGCN encoder on the full graph plus two
edge-dropped views rebuilt every epoch, InfoNCE between the views (reference: models/SGL.py,
including its own epoch loop `SGL_trainer`).  Every view is a device graph handle made ON the
device from the full adjacency's handle (dropped interactions become explicit zeros, kept ones are
re-normalised: idg_graph_revalued_copy) — no per-epoch host-side graph or tile-schedule build."""
from time import time

import numpy as np
import torch

import utility.utility_data.data_graph as data_graph
import utility.utility_function.losses as losses
import utility.utility_function.tools as tools
import utility.utility_train.batch_test as batch_test
from idgrec_amd import ops
from idgrec_amd.modeling import PackedRecommender


class SGL(PackedRecommender):
    include_layer0 = True

    def __init__(self, config, dataset, device):
        super(SGL, self).__init__(config, dataset, device)
        self.n_layers = int(config['GCN_layer'])
        self.ssl_lambda = float(config['ssl_lambda'])
        self.temperature = float(config['temperature'])
        self.attach_graph(data_graph.sparse_adjacency_matrix(dataset))

    def aggregate(self, graph):
        """graph: one handle for every layer, or a list with one handle per layer ('rw')."""
        ego = self.ego_panel()
        if isinstance(graph, list):
            x, total = ego, ego
            for layer in range(self.n_layers):
                x = ops.spmm(graph[layer], x)
                total = total + x
            final = total / float(self.n_layers + 1)
        else:
            final = ops.propagate_mean(graph, ego, self.n_layers, include_layer0=True)
        return torch.split(final, [self.dataset.num_users, self.dataset.num_items])

    def forward(self, user, positive, negative, sub_graph_1, sub_graph_2):
        ego = self.ego_panel()
        final = ops.propagate_mean(self.Graph, ego, self.n_layers, include_layer0=True)
        view_1 = torch.cat(self.aggregate(sub_graph_1)) if isinstance(sub_graph_1, list) else \
            ops.propagate_mean(sub_graph_1, ego, self.n_layers, include_layer0=True)
        view_2 = torch.cat(self.aggregate(sub_graph_2)) if isinstance(sub_graph_2, list) else \
            ops.propagate_mean(sub_graph_2, ego, self.n_layers, include_layer0=True)
        bpr_loss, reg_loss = ops.bpr_loss(final, ego, user, positive, negative, self.dataset.num_users,
                                          self.reg_lambda)
        # InfoNCE between the two sub-graph views over the batch's users and its positive items, indexed with the
        # raw batch ids — no torch.unique here, duplicates count (models/SGL.py:85-86, 96-101): fused operator
        ssl = ops.infonce_pair(view_1, view_2, user, positive, self.dataset.num_users, self.temperature, dedup=False)
        return [bpr_loss, reg_loss, self.ssl_lambda * ssl]


    supports_fused_step = True  # through fused_sgl_step (the sub-graphs change every epoch)
    n_fused_losses = 3

    def fused_sgl_step(self, user, positive, negative, sub_graph_1, sub_graph_2, loss_out, optimizer):
        """forward() + backward() + optimizer.step() for single-graph views as one chain of kernels."""
        eng = self.engine()
        eng.sgl = (self.temperature, self.ssl_lambda, sub_graph_1, sub_graph_2)
        return self.fused_train_step(user, positive, negative, loss_out, optimizer)


class Trainer():
    def __init__(self, args, config, dataset, device, logger):
        self.model = SGL(config, dataset, device)
        self.args, self.config, self.dataset = args, config, dataset
        self.device, self.logger = device, logger
        self.aug_type = config['aug_type']
        self.ssl_ratio = float(config['ssl_ratio'])

    def train(self):
        self.SGL_trainer()

    def _subgraph_state(self):
        """Once per dataset: the full adjacency's CSR structure on the device and, per stored entry, its row and the
        interaction (position in inter_graph.nonzero()'s order, tools.py:70) it stands for."""
        st = getattr(self, "_sub_state", None)
        if st is None:
            import scipy.sparse as sp

            R = self.dataset.user_item_net.tocsr()
            R.sort_indices()
            U, I = R.shape
            E = R.nnz
            user_index, item_index = R.nonzero()  # CSR order: interaction e is the e-th stored entry of R
            ids = sp.csr_matrix((np.arange(1, E + 1, dtype=np.int64), R.indices, R.indptr), shape=R.shape)
            Rt = ids.T.tocsr()
            Rt.sort_indices()
            # A = [[0, R], [R^T, 0]]: user rows keep R's entry order, item rows R^T's
            indptr = np.concatenate([R.indptr.astype(np.int64), E + Rt.indptr[1:].astype(np.int64)])
            indices = np.concatenate([R.indices.astype(np.int32) + U, Rt.indices.astype(np.int32)])
            edge = np.concatenate([np.arange(E, dtype=np.int32), (Rt.data - 1).astype(np.int32)])
            row = np.repeat(np.arange(U + I, dtype=np.int32), np.diff(indptr))
            dev = torch.device(self.device)
            st = self._sub_state = dict(
                U=U, I=I, E=E, user_index=np.asarray(user_index), item_index=np.asarray(item_index),
                indptr=torch.from_numpy(indptr).to(dev), indices=torch.from_numpy(indices).to(dev),
                edge=torch.from_numpy(edge).to(dev), row=torch.from_numpy(row).to(dev))
            full = self.model.Graph
            assert full.nnz == len(indices) and full.n_rows == U + I, "the model's graph is not the plain bipartite adjacency"
        return st

    def _view(self):
        """One edge-dropped view (tools.create_adj_mat, tools.py:67-92).  The draw — random.sample on Python's own
        stream, restated natively — and the kept graph's d^-1/2 (the reference's np.power expression on its degree
        counts) are host work on index lists; the view itself never exists on the host: it is the full adjacency's
        device handle with dropped interactions as explicit zeros and the kept ones re-normalised
        (idg_subgraph_values_f32 + idg_graph_revalued_copy: same tile schedule, no per-epoch schedule build)."""
        if self.aug_type == 'nd':
            raise NotImplementedError("The method does not implemented.")
        if torch.device(self.device).type != "cuda" or self.model.Graph is None:
            mat = tools.create_adj_mat(self.dataset.user_item_net, self.aug_type, self.ssl_ratio)
            return tools.convert_sp_mat_to_graph(mat, self.device)
        import idgrec_amd.host as host

        st = self._subgraph_state()
        E, U, I = st["E"], st["U"], st["I"]
        keep = host.py_random_sample(E, int((1 - self.ssl_ratio) * E))  # == random.sample(range(E), k) (tools.py:77)
        kept = np.zeros((E + 31) // 32 * 32, dtype=bool)
        kept[keep] = True
        bits = np.packbits(kept, bitorder="little").view(np.int32)  # bit e of the little-endian bitmap = interaction e
        deg = np.concatenate([np.bincount(st["user_index"][keep], minlength=U),
                              np.bincount(st["item_index"][keep], minlength=I)]).astype(np.float32)
        with np.errstate(divide="ignore"):
            d_inv = np.power(deg, -0.5)  # tools.py:85
        d_inv[np.isinf(d_inv)] = 0.
        dev = torch.device(self.device)
        values = ops.subgraph_values_raw(st["row"], st["indices"], st["edge"], torch.from_numpy(bits).to(dev),
                                         torch.from_numpy(d_inv.astype(np.float32)).to(dev))
        return self.model.Graph.revalued_copy(st["indptr"], st["indices"], values)

    def _views(self):
        if self.aug_type in ['nd', 'ed']:
            return self._view(), self._view()
        layers = int(self.config['GCN_layer'])  # 'rw': a fresh graph per layer, interleaved draw order as the reference
        pairs = [(self._view(), self._view()) for _ in range(layers)]
        return [p[0] for p in pairs], [p[1] for p in pairs]

    def SGL_trainer(self):
        """The reference's own loop for this model (models/SGL.py:115-199): sub-graphs per epoch,
        no early-stop break, one more evaluation after the last epoch."""
        cfg, model, device = self.config, self.model, self.device
        model.to(device)
        Optim = ops.Adam(model.parameters(), lr=float(cfg['learn_rate']))
        batch_size = int(cfg['batch_size'])
        top_k = eval(cfg['top_K'])
        best_results = {'count': 0, 'epoch': 0, 'recall': [0. for _ in top_k], 'ndcg': [0. for _ in top_k]}
        best_results['stop'] = 0
        for epoch in range(int(cfg['training_epochs'])):
            print('-' * 100)
            start_time = time()
            sub_graph_1, sub_graph_2 = self._views()
            model.train()
            triples = torch.from_numpy(self.dataset.sample_data_to_train_all()).to(device)
            users, pos_items, neg_items = tools.shuffle(triples[:, 0], triples[:, 1], triples[:, 2])
            num_batch = len(users) // batch_size + 1
            step_losses = torch.zeros((num_batch, 3), dtype=torch.float32, device=device)
            # one sub-graph per view ('ed' / 'nd'): the step runs as a fixed chain of kernels (PropagationEngine.sgl);
            # per-layer graph lists ('rw') go through forward() under autograd
            fused = not isinstance(sub_graph_1, list) and torch.device(device).type == "cuda" and model.fused_step_available()
            users, pos_items, neg_items = users.contiguous(), pos_items.contiguous(), neg_items.contiguous()
            batches = list(tools.mini_batch(users, pos_items, neg_items, batch_size=batch_size))
            for batch_i, (b_u, b_p, b_n) in enumerate(batches):
                if fused:
                    if batch_i + 1 < len(batches):
                        model.prefetch_batch(*batches[batch_i + 1])
                    if model.fused_sgl_step(b_u, b_p, b_n, sub_graph_1, sub_graph_2, step_losses[batch_i], Optim):
                        continue
                loss_list = model(b_u, b_p, b_n, sub_graph_1, sub_graph_2)
                step_losses[batch_i] = torch.stack([l.detach() for l in loss_list])
                Optim.zero_grad()
                sum(loss_list).backward()
                Optim.step()
            total_loss_list = [0.] * 3
            for row in step_losses.double().cpu().numpy():
                for i, v in enumerate(row):
                    total_loss_list[i] += float(v)
            end_time = time()
            loss_strs = str(round(sum(total_loss_list) / num_batch, 6)) \
                + " = " + " + ".join([str(round(i / num_batch, 6)) for i in total_loss_list])
            print("\t Epoch: %4d| train time: %.3f | train_loss: %s" % (epoch + 1, end_time - start_time, loss_strs))
            self.logger.info("Epoch: %4d | Training time: %.3f | training loss: %s"
                             % (epoch + 1, end_time - start_time, loss_strs))
            if epoch % int(cfg['interval']) == 0:
                result, best_results = batch_test.general_test(self.dataset, model, device, cfg, epoch, best_results)
                self.logger.info("Epoch: %4d | Test recall: %s | Test NDCG: %s" % (epoch + 1, result['recall'], result['ndcg']))
        print("\t Model training process completed.")
        self.logger.info('Model training process completed.')
        result, best_results = batch_test.general_test(self.dataset, model, device, cfg, int(cfg['training_epochs']),
                                                       best_results)
        self.logger.info("Best epoch: %4d | Best recall: %s | Best NDCG: %s"
                         % (best_results['epoch'], best_results['recall'], best_results['ndcg']))
