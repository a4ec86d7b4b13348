"""Dataset object of the LightGCN-family trainers.

Mirrors the interface of the reference's `Data` (utility/utility_data/data_loader.py:8-204)
— same attribute and method names, same return types — on top of libidgrec.so's host
entry points: the rating files are parsed natively (idg_ratings_*), and the BPR negative
sampler is the native MT19937 restatement running on NumPy's global stream
(idg_sample_epoch), so `sample_data_to_train_all()` returns exactly the triples the
reference's Python loop would for the same np.random state.
"""
import os

import numpy as np
import scipy.sparse as sp

from idgrec_amd import host as _host


class Data(object):
    def __init__(self, path, config):
        self.path = path
        self.config = config
        self.num_users = self.num_items = 0
        self.num_entities = self.num_relations = 0
        self.num_nodes = self.num_train = self.num_test = 0
        self.split_test_dict = None
        self.split_state = None
        self._stream = _host.GlobalStream()
        self._device_cache = {}

        self.load_data()
        if config and int(config.get("sparsity_test", 0)) == 1:
            self.split_test_dict, self.split_state = self.create_sparsity_split()

    # ------------------------------------------------------------------ loading
    def read_ratings(self, file_name):
        """-> (users of every line, edge users, edge items, #edges, items-per-nonempty-line).
        Also raises self.num_users / self.num_items to the largest ids seen (not yet +1),
        ignoring lines without items, as data_loader.py:59-63 does."""
        users, items, line_users, max_u, max_i, counts = _host.parse_ratings(file_name, counts=True)
        if len(users):
            self.num_users = max(self.num_users, max_u)
            self.num_items = max(self.num_items, max_i)
        pos_length = counts[counts > 0].tolist()
        return line_users, users, items, int(len(users)), pos_length

    def load_data(self):
        train_file = os.path.join(self.path, "train.txt") if not self.path.endswith("/") else self.path + "train.txt"
        test_file = train_file[: -len("train.txt")] + "test.txt"
        _, self.train_user, self.train_item, self.num_train, self.pos_length = self.read_ratings(train_file)
        _, self.test_user, self.test_item, self.num_test, _ = self.read_ratings(test_file)
        self.num_users += 1
        self.num_items += 1
        self.num_nodes = self.num_users + self.num_items
        self.data_statistics()

        ones = np.ones(len(self.train_user))  # float64; duplicate (u,i) pairs sum to 2 (data_loader.py:42)
        self.user_item_net = sp.csr_matrix((ones, (self.train_user, self.train_item)),
                                           shape=(self.num_users, self.num_items))
        self.user_item_net.sort_indices()
        self._pos_indptr = self.user_item_net.indptr.astype(np.int64)
        self._pos_indices = np.ascontiguousarray(self.user_item_net.indices, dtype=np.int32)
        self.all_positive = self.get_user_pos_items(list(range(self.num_users)))
        self.test_dict = self.build_test()

    def _sparsity(self):
        return 1 - (self.num_train + self.num_test) / self.num_users / self.num_items

    def data_statistics(self):
        for label, value in (("num_users:", self.num_users), ("num_items:", self.num_items),
                             ("num_nodes:", self.num_nodes), ("num_train:", self.num_train),
                             ("num_test: ", self.num_test), ("sparisty: ", self._sparsity())):
            print("\t " + label, value)

    def get_statistics(self):
        return ("dataset:" + self.config["dataset"] + "\t"
                + "num_users:%d, num_items:%d \t" % (self.num_users, self.num_items)
                + "|num_train:%d, num_test:%d, sparsity: %.6f" % (self.num_train, self.num_test, self._sparsity()))

    # ------------------------------------------------------------------ sampling
    def sample_data_to_train_all(self):
        """One negative per train edge, in file order -> int64 [E, 3] (data_loader.py:108-127)."""
        with self._stream as rng:
            return rng.sample_epoch(self.train_user, self.train_item, self._pos_indptr, self._pos_indices,
                                    self.num_items)

    def sample_data_to_train_random(self):
        """LightGCN-official sampling (data_loader.py:89-106); not used by the shipped trainers,
        kept on NumPy because np.random.randint(low, high, size) draws from a different path
        of the legacy generator than the scalar form."""
        users = np.random.randint(0, self.num_users, len(self.train_user))
        rows = []
        for user in users:
            positives = self.all_positive[user]
            if len(positives) == 0:
                continue
            pos_item = positives[np.random.randint(0, len(positives))]
            neg_item = np.random.randint(0, self.num_items)
            while neg_item in positives:
                neg_item = np.random.randint(0, self.num_items)
            rows.append([user, pos_item, neg_item])
        return np.array(rows)

    def get_user_pos_items(self, users):
        ip, ix = self._pos_indptr, self._pos_indices
        return [ix[ip[u]:ip[u + 1]] for u in users]

    def get_user_n_neg_items(self, users, n):
        out = []
        for user in users:
            picked = []
            positives = self.all_positive[user]
            while len(picked) < n:
                cand = np.random.randint(0, self.num_items)
                if cand not in positives:
                    picked.append(cand)
            out.append(picked)
        return out

    def build_test(self):
        test = {}
        for user, item in zip(self.test_user.tolist(), self.test_item.tolist()):
            test.setdefault(user, []).append(item)
        return test

    # ------------------------------------------------------------------ device-side views
    def train_csr_on(self, device):
        """(indptr int64[U+1], items int32[nnz]) of the train matrix as device tensors — the
        exclusion lists of batch_test.py:62-65 without the per-batch Python list building."""
        import torch

        key = str(device)
        if key not in self._device_cache:
            self._device_cache[key] = (torch.from_numpy(self._pos_indptr).to(device),
                                       torch.from_numpy(self._pos_indices).to(device))
        return self._device_cache[key]

    # ------------------------------------------------------------------ sparsity buckets
    def create_sparsity_split(self):
        """Four user groups of roughly equal interaction mass (data_loader.py:161-204)."""
        by_count = {}
        for uid in self.test_dict:
            n_inter = len(self.all_positive[uid]) + len(self.test_dict[uid])
            by_count.setdefault(n_inter, []).append(uid)
        total = self.num_train + self.num_test
        groups, states = [], []
        current, mass, remaining, fold = [], 0, total, 1
        ordered = sorted(by_count)
        for idx, n_inter in enumerate(ordered):
            current += by_count[n_inter]
            step = n_inter * len(by_count[n_inter])
            mass += step
            remaining -= step
            if mass >= fold * 0.25 * total:
                groups.append(current)
                states.append("\t #inter per user<=[%d], #users=[%d], #all rates=[%d]" % (n_inter, len(current), mass))
                print(states[-1])
                current, mass = [], 0
            if idx == len(ordered) - 1 or remaining == 0:
                groups.append(current)
                states.append("\t #inter per user<=[%d], #users=[%d], #all rates=[%d]" % (n_inter, len(current), mass))
                print(states[-1])
        return groups, states
