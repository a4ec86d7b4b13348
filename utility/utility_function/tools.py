"""Helpers with the reference's names and semantics (utility/utility_function/tools.py):
seeding, `key = value` configuration files, the epoch shuffle and mini-batch slicing.

The shuffle draws its permutation from NumPy's global legacy stream through the native
Fisher-Yates restatement (idg_shuffle_perm), so sampler and shuffle keep consuming ONE
stream in the reference's order: sample(e) -> shuffle(e) -> sample(e+1) ...
"""
import os

import numpy as np
import torch

from idgrec_amd import host as _host

_stream = None


def _global_stream():
    global _stream
    if _stream is None:
        _stream = _host.GlobalStream()
    return _stream


def set_seed(seed):
    """np.random + torch (CPU and every HIP device); python's `random` is left alone, as in
    the reference (tools.py:8-14)."""
    np.random.seed(seed)
    if torch.cuda.is_available():
        torch.cuda.manual_seed(seed)
        torch.cuda.manual_seed_all(seed)
    torch.manual_seed(seed)


def read_configuration(filename, model):
    """dict[str, str] from `key = value` lines.  A line that does not split into exactly two
    parts around '=' is reported and skipped (tools.py:26-30)."""
    if not os.path.exists(filename):
        print("\tThe path does not have a configuration file for " + model + ".")
        raise IOError
    config = {}
    with open(filename, "r") as f:
        for line in f:
            parts = line.strip().split("=")
            if len(parts) != 2:
                print("\tConfiguration file format error.")
                continue
            config[parts[0].strip()] = parts[1].strip()
    return config


def _take(x, perm_np, perm_cache):
    if isinstance(x, torch.Tensor):
        key = str(x.device)
        if key not in perm_cache:
            perm_cache[key] = torch.from_numpy(perm_np).to(x.device)
        return x[perm_cache[key]]
    return x[perm_np]


def shuffle(*arrays, **kwargs):
    """Apply one random permutation to every array (numpy arrays or torch tensors on any
    device); `indices=True` also returns the permutation."""
    want_indices = kwargs.get("indices", False)
    lengths = {len(x) for x in arrays}
    if len(lengths) != 1:
        raise ValueError("Inputs to shuffle must have the same length.")
    with _global_stream() as rng:
        perm = rng.shuffle_perm(lengths.pop())
    cache = {}
    if len(arrays) == 1:
        result = _take(arrays[0], perm, cache)
    else:
        result = tuple(_take(x, perm, cache) for x in arrays)
    return (result, perm) if want_indices else result


def mini_batch(*tensors, **kwargs):
    """Consecutive slices of `batch_size` (last one short); one tensor -> slices, several ->
    tuples of slices."""
    size = kwargs.get("batch_size", 1024)
    total = len(tensors[0])
    for lo in range(0, total, size):
        if len(tensors) == 1:
            yield tensors[0][lo:lo + size]
        else:
            yield tuple(x[lo:lo + size] for x in tensors)


def create_adj_mat(inter_graph, aug_type, ssl_rate):
    """SGL's augmented graph (tools.py:67-92): keep int((1 - ssl_rate) * E) interactions chosen with
    python's `random.sample` (never seeded by the reference), rebuild the symmetric bipartite
    adjacency on the kept edges and normalise it — natively, through idg_build_norm_adj.
    Returns a scipy CSR float32 matrix like the reference."""
    import scipy.sparse as sp

    num_users, num_items = inter_graph.get_shape()
    user_index, item_index = inter_graph.nonzero()
    if aug_type == 'nd':
        raise NotImplementedError("The method does not implemented.")
    if aug_type not in ('ed', 'rw'):
        raise ValueError("unknown aug_type %r" % (aug_type,))
    edge_number = inter_graph.count_nonzero()
    # random.sample(range(edge_number), k) — the same draws from Python's `random` stream, taken natively
    keep_index = _host.py_random_sample(edge_number, int((1 - ssl_rate) * edge_number))
    keep_users = np.asarray(user_index)[keep_index]
    keep_items = np.asarray(item_index)[keep_index]
    indptr, indices, values = _host.build_norm_adj(num_users, num_items, keep_users, keep_items)
    n = num_users + num_items
    return sp.csr_matrix((values, indices, indptr), shape=(n, n))


def convert_sp_mat_to_graph(sp_mat, device, symmetric=True):
    """scipy matrix -> device graph handle for idgrec_amd.ops.spmm / propagate_mean.  Stands
    where the reference builds its coalesced torch sparse tensor (models/LightGCN.py:31-32)."""
    from idgrec_amd import ops

    return ops.Graph.from_scipy(sp_mat.astype(np.float32), device=device, symmetric=symmetric)


def convert_sp_mat_to_sp_tensor(sp_mat):
    """scipy matrix -> torch sparse COO float tensor (tools.py:95-109), for code that still
    wants a torch sparse tensor.  Indices are taken as integers (no float32 round trip)."""
    coo = sp_mat.tocoo().astype(np.float32)
    index = torch.from_numpy(np.stack([coo.row.astype(np.int64), coo.col.astype(np.int64)]))
    return torch.sparse_coo_tensor(index, torch.from_numpy(coo.data), torch.Size(coo.shape))
