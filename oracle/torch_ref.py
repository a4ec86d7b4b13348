"""oracle/torch_ref.py — TEST INFRASTRUCTURE / CPU BASELINE, NOT PRODUCT CODE.

The reference's LightGCN / MFBPR / SimGCL training step restated with the same stock PyTorch CPU ops
in the same order (models/LightGCN.py:36-72, models/SimGCL.py:39-90, utility/utility_function/losses.py:4-35,
utility/utility_train/trainer.py:42-56), so it can be timed on the GPU box's host cores
where the reference's own Python files are not available.  tests/test_torch_ref.py pins it
to the goldens dumped from the imported reference.
"""
import numpy as np
import torch


class RefStep:
    def __init__(self, indptr, indices, values, num_users, num_items, user_w, item_w, n_layers=3, reg_lambda=1e-4,
                 lr=1e-3, propagate=True, simgcl=None):
        """simgcl: None, or (epsilon, temperature, ssl_lambda) — then the step is SimGCL's (models/SimGCL.py:62-90):
        layer 0 left out of the mean, two noise-perturbed encoder passes, InfoNCE between them."""
        self.U, self.I = int(num_users), int(num_items)
        n = self.U + self.I
        self.K, self.reg_lambda, self.propagate = int(n_layers), float(reg_lambda), propagate
        self.simgcl = simgcl
        self.user_w = torch.nn.Parameter(torch.as_tensor(user_w, dtype=torch.float32).clone())
        self.item_w = torch.nn.Parameter(torch.as_tensor(item_w, dtype=torch.float32).clone())
        if propagate:
            rows = np.repeat(np.arange(n, dtype=np.int64), np.diff(indptr))
            idx = torch.from_numpy(np.stack([rows, np.asarray(indices, dtype=np.int64)]))
            self.Graph = torch.sparse_coo_tensor(idx, torch.as_tensor(values, dtype=torch.float32), (n, n)).coalesce()
        self.opt = torch.optim.Adam([self.user_w, self.item_w], lr=lr)
        self.sigmoid = torch.nn.Sigmoid()

    def aggregate(self, perturbed=False):
        all_embedding = torch.cat([self.user_w, self.item_w])
        if not self.propagate:
            return self.user_w, self.item_w
        embeddings = [] if self.simgcl is not None else [all_embedding]  # SimGCL.py:44-45: no layer 0 in the mean
        for _ in range(self.K):
            all_embedding = torch.sparse.mm(self.Graph, all_embedding)
            if perturbed:  # SimGCL.py:49-51 (in place: the noise feeds the next layer)
                noise = torch.rand_like(all_embedding)
                all_embedding += torch.sign(all_embedding) * torch.nn.functional.normalize(noise, dim=-1) * self.simgcl[0]
            embeddings.append(all_embedding)
        final = torch.mean(torch.stack(embeddings, dim=1), dim=1)
        return torch.split(final, [self.U, self.I])

    def losses(self, user, pos, neg):
        au, ai = self.aggregate()
        ue, pe, ne = au[user], ai[pos], ai[neg]
        eu, ep, en = self.user_w[user], self.item_w[pos], self.item_w[neg]
        pos_score = torch.sum(torch.mul(ue, pe), dim=1)
        neg_score = torch.sum(torch.mul(ue, ne), dim=1)
        bpr = torch.mean(-torch.log(torch.sigmoid(pos_score - neg_score) + 10e-8))
        reg = 0
        for e in (eu, ep, en):
            reg += 1 / 2 * e.norm(2).pow(2) / float(e.shape[0])
        if self.simgcl is None:
            return [bpr, self.reg_lambda * reg]
        _, temperature, ssl_lambda = self.simgcl
        u1, i1 = self.aggregate(perturbed=True)
        u2, i2 = self.aggregate(perturbed=True)
        ui, ii = torch.unique(user), torch.unique(pos)
        ssl = info_nce(u1[ui], u2[ui], temperature) + info_nce(i1[ii], i2[ii], temperature)
        return [bpr, self.reg_lambda * reg, ssl_lambda * ssl]

    def step(self, user, pos, neg):
        loss_list = self.losses(user, pos, neg)
        total = 0.0
        vals = []
        for l in loss_list:
            total = total + l
            vals.append(l.item())
        self.opt.zero_grad()
        total.backward()
        self.opt.step()
        return vals

    def rating(self, users):
        with torch.no_grad():
            au, ai = self.aggregate()
            return self.sigmoid(torch.matmul(au[users], ai.t()))


def info_nce(embedding_1, embedding_2, temperature):
    """utility/utility_function/losses.py:24-35 (get_InfoNCE_loss)."""
    embedding_1 = torch.nn.functional.normalize(embedding_1)
    embedding_2 = torch.nn.functional.normalize(embedding_2)
    pos_score = torch.exp((embedding_1 * embedding_2).sum(dim=-1) / temperature)
    ttl_score = torch.exp(torch.matmul(embedding_1, embedding_2.transpose(0, 1)) / temperature).sum(dim=1)
    return torch.mean(-torch.log(pos_score / ttl_score + 10e-6))
