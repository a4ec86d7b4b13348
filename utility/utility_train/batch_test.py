"""Full-rank evaluation — `general_test`, `Test`, `test_one_batch`, `sparsity_test` with the
reference's signatures, result dictionaries and early-stopping bookkeeping
(utility/utility_train/batch_test.py).

For every batch of test users the reference materialises sigmoid(E_u E_i^T) [B, I], writes -1
over the training positives through Python index lists, and calls torch.topk
(batch_test.py:59-68).  Models that provide `topk_for_test` get all of that from one fused
call (MFMA scoring + masking from the device-resident train CSR + wave-level top-K) issued
ONCE for every test user — the fused kernel has no [B, I] matrix to bound, and a user's
list does not depend on who shares its launch — and the lists are then cut into the
reference's `test_batch_size` batches so the metric sums keep their order.  Any other model
goes through its own `get_rating_for_test` and the same mask / topk steps per batch.
"""
import numpy as np
import torch

import utility.utility_function.metrics as metrics
from utility.utility_data.data_loader import Data
from utility.utility_function.tools import mini_batch


def general_test(dataset, model, device, config, epoch, best_results):
    if int(config["sparsity_test"]) != 0:
        result = sparsity_test(dataset, model, device, config)
        for level, res in enumerate(result[:4], start=1):
            print("\t level_%d: recall:" % level, res['recall'], ',ndcg:', res['ndcg'])
        return result[0], best_results

    result = Test(dataset, model, device, config)
    if result['recall'][0] > best_results['recall'][0]:  # the FIRST cut-off decides (batch_test.py:11)
        best_results.update(count=0, epoch=epoch + 1, recall=result['recall'], ndcg=result['ndcg'])
    else:
        best_results['count'] += 1
        if best_results['count'] >= int(config['early_stopping']):
            print("Early stop......")
            print("Best epoch:   ", best_results['epoch'], " Best recall:", best_results['recall'],
                  "Best NDCG:", best_results['ndcg'])
            best_results['stop'] = 99999
            return result, best_results
    print("Current epoch:", epoch + 1, " Test recall:", result['recall'], "Test NDCG:", result['ndcg'])
    print("Best epoch:   ", best_results['epoch'], " Best recall:", best_results['recall'],
          "Best NDCG:", best_results['ndcg'])
    return result, best_results


def _topk_for_users(dataset, model, device, batch_users, k):
    """Top-k recommended item ids [len(batch_users), k] (host, int64), training items excluded."""
    users_device = torch.as_tensor(np.asarray(batch_users, dtype=np.int64), device=device)
    if hasattr(model, "topk_for_test"):
        return model.topk_for_test(users_device, k).cpu()
    rating = model.get_rating_for_test(users_device)
    positives = dataset.get_user_pos_items(batch_users)
    rows = np.repeat(np.arange(len(batch_users)), [len(p) for p in positives])
    if len(rows):
        cols = np.concatenate(positives).astype(np.int64)
        rating[torch.as_tensor(rows, device=rating.device), torch.as_tensor(cols, device=rating.device)] = -1
    _, rating_k = torch.topk(rating, k=k)
    return rating_k.cpu()


def _evaluate(dataset, model, device, config, users):
    topK = eval(config['top_K'])
    totals = {name: np.zeros(len(topK)) for name in ('precision', 'recall', 'hit', 'ndcg')}
    test_batch = int(config['test_batch_size'])
    num_batch = len(users) // test_batch + 1
    batches = []
    with torch.no_grad():
        fused = _topk_for_users(dataset, model, device, users, max(topK)) if hasattr(model, "topk_for_test") and len(users) else None
        hits = _hit_matrix(dataset, users, fused.numpy()) if fused is not None else None
        lo = 0
        for batch_users in mini_batch(users, batch_size=test_batch):
            truth = [dataset.test_dict[u] for u in batch_users]
            if fused is not None:
                top = fused[lo:lo + len(batch_users)]
                label = hits[lo:lo + len(batch_users)]
                lo += len(batch_users)
            else:
                top, label = _topk_for_users(dataset, model, device, batch_users, max(topK)), None
            batches.append((top, truth, label))
    assert num_batch == len(batches)  # as the reference: breaks when test_batch_size divides #users
    for part in batches:
        res = test_one_batch(part, topK)
        for name in ('recall', 'precision', 'ndcg'):
            totals[name] += res[name]
    for name in ('recall', 'precision', 'ndcg'):
        totals[name] /= float(len(users))
    return totals


def Test(dataset: Data, model, device, config):
    model = model.eval()
    return _evaluate(dataset, model, device, config, list(dataset.test_dict.keys()))


def _hit_matrix(dataset, users, recommended):
    """metrics.get_label for every user at once: r[i, j] = 1.0 iff recommended[i, j] is one of users[i]'s held-out
    items — one sorted-key membership test over (user * num_items + item) instead of a Python loop over users."""
    keys = getattr(dataset, "_test_pair_keys", None)
    if keys is None:
        n_items = int(dataset.num_items)
        keys = np.sort(np.concatenate([np.asarray(list(items), dtype=np.int64) + np.int64(u) * n_items
                                       for u, items in dataset.test_dict.items()] or [np.empty(0, dtype=np.int64)]))
        dataset._test_pair_keys = keys
    query = np.asarray(users, dtype=np.int64)[:, None] * np.int64(dataset.num_items) + np.asarray(recommended, dtype=np.int64)
    return np.isin(query, keys).astype("float")


def test_one_batch(X, topK):
    """X = (recommended ids [B, max k], held-out item lists[, precomputed hit matrix])."""
    recommended = X[0].numpy()
    truth = X[1]
    r = X[2] if len(X) > 2 and X[2] is not None else metrics.get_label(truth, recommended)
    out = {'recall': [], 'precision': [], 'ndcg': []}
    for k in topK:
        out['recall'].append(metrics.recall_at_k(r, k, truth))
        out['precision'].append(metrics.precision_at_k(r, k, truth))
        out['ndcg'].append(metrics.ndcg_at_k(r, k, truth))
    return {name: np.array(vals) for name, vals in out.items()}


def sparsity_test(dataset: Data, model, device, config):
    """One result dict per interaction-count bucket of Data.create_sparsity_split."""
    model = model.eval()
    return [_evaluate(dataset, model, device, config, users) for users in dataset.split_test_dict]
