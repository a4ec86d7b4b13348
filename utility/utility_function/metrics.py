"""Ranking metrics over a 0/1 hit matrix `r` [users, max_k] — same functions and values as
utility/utility_function/metrics.py:4-58 in the reference, vectorised."""
import numpy as np


def _discounts(k):
    return 1.0 / np.log2(np.arange(2, k + 2))


def ndcg_at_k(r, k, test_data):
    assert len(r) == len(test_data)
    gains = r[:, :k]
    n_rel = np.minimum(k, np.fromiter((len(t) for t in test_data), dtype=np.int64, count=len(test_data)))
    ideal = (np.arange(k)[None, :] < n_rel[:, None]).astype(float)
    disc = _discounts(k)
    idcg = np.sum(ideal * disc, axis=1)
    dcg = np.sum(gains * disc, axis=1)
    idcg[idcg == 0.0] = 1.0
    ndcg = dcg / idcg
    ndcg[np.isnan(ndcg)] = 0.0
    return np.sum(ndcg)


def recall_at_k(r, k, test_data):
    hits = r[:, :k].sum(1)
    n_rel = np.array([len(t) for t in test_data])
    return np.sum(hits / n_rel)


def precision_at_k(r, k, test_data):
    return np.sum(r[:, :k].sum(1)) / k


def F1(pre, rec):
    return [(2.0 * p * q) / (p + q) if p + q > 0 else 0.0 for p, q in zip(pre, rec)]


def get_label(true_data, pred_data):
    """r[i, j] = 1.0 iff pred_data[i][j] is one of user i's held-out items."""
    rows = []
    for truth, pred in zip(true_data, pred_data):
        rows.append(np.isin(np.asarray(pred), np.asarray(list(truth))).astype("float"))
    return np.array(rows).astype("float")
