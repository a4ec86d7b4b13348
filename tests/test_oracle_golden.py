"""The oracle is only worth anything if it reproduces the reference: every check here
compares oracle/ against vectors dumped from the imported reference (oracle/gen_golden.py)."""
import json

import numpy as np
import pytest

from oracle import oracle


def _adj(g):
    return g["adj_indptr"], g["adj_indices"], g["adj_data"]


@pytest.mark.parametrize("gname,d", [("tiny", 64), ("tiny", 256), ("small", 64)])
def test_spmm_bit_exact_vs_torch_cpu(gname, d, golden_tiny, golden_small):
    g = golden_tiny if gname == "tiny" else golden_small
    E0 = np.concatenate([g["d%d_init_user" % d], g["d%d_init_item" % d]])
    Y = oracle.spmm(*_adj(g), E0)
    assert np.array_equal(Y, g["d%d_spmm1" % d])  # torch.sparse.mm, bit for bit


@pytest.mark.parametrize("gname,d", [("tiny", 64), ("tiny", 256), ("small", 64)])
def test_propagate_mean_bit_exact(gname, d, golden_tiny, golden_small):
    g = golden_tiny if gname == "tiny" else golden_small
    U = int(g["num_users"])
    E0 = np.concatenate([g["d%d_init_user" % d], g["d%d_init_item" % d]])
    out = oracle.propagate_mean(*_adj(g), E0, 3, include_layer0=True)
    assert np.array_equal(out[:U], g["d%d_lgcn_user" % d])
    assert np.array_equal(out[U:], g["d%d_lgcn_item" % d])
    out = oracle.propagate_mean(*_adj(g), E0, 3, include_layer0=False)
    assert np.array_equal(out[:U], g["d%d_simgcl_user" % d])
    assert np.array_equal(out[U:], g["d%d_simgcl_item" % d])


@pytest.mark.parametrize("gname", ["tiny", "small"])
def test_adjacency_bit_exact(gname, golden_tiny, golden_small):
    g = golden_tiny if gname == "tiny" else golden_small
    ip, ix, dv = oracle.norm_adj(g["num_users"], g["num_items"], g["train_user"], g["train_item"])
    assert np.array_equal(ip, g["adj_indptr"]) and np.array_equal(ix, g["adj_indices"])
    assert np.array_equal(dv, g["adj_data"])
    ip, ix, dv = oracle.norm_adj(g["num_users"], g["num_items"], g["train_user"], g["train_item"], self_loops=True)
    assert np.array_equal(ip, g["adjself_indptr"]) and np.array_equal(ix, g["adjself_indices"])
    assert np.array_equal(dv, g["adjself_data"])


def test_adjacency_symmetric_and_duplicate_edge(golden_tiny):
    import scipy.sparse as sp

    g = golden_tiny
    A = sp.csr_matrix((g["adj_data"], g["adj_indices"], g["adj_indptr"]))
    assert (A != A.T).nnz == 0  # exactly symmetric
    assert g["pos_data"].max() == 2.0  # the injected repeated pair


@pytest.mark.parametrize("gname", ["tiny", "small"])
def test_sampler_stream(gname, golden_tiny, golden_small):
    g = golden_tiny if gname == "tiny" else golden_small
    U = int(g["num_users"])
    pos = [g["pos_indices"][g["pos_indptr"][u]:g["pos_indptr"][u + 1]] for u in range(U)]
    np.random.seed(2024)
    s1 = oracle.sample_epoch(g["train_user"], g["train_item"], pos, int(g["num_items"]))
    p1 = oracle.shuffle_perm(len(s1))
    s2 = oracle.sample_epoch(g["train_user"], g["train_item"], pos, int(g["num_items"]))
    p2 = oracle.shuffle_perm(len(s2))
    for a, b in ((s1, "sample1"), (p1, "perm1"), (s2, "sample2"), (p2, "perm2")):
        assert np.array_equal(a, g[b])


@pytest.mark.parametrize("gname,d", [("tiny", 64), ("small", 64)])
def test_bpr_loss_and_grads(gname, d, golden_tiny, golden_small):
    g = golden_tiny if gname == "tiny" else golden_small
    U = int(g["num_users"])
    batch = g["d%d_batch" % d]
    E0 = np.concatenate([g["d%d_init_user" % d], g["d%d_init_item" % d]])
    fin = np.concatenate([g["d%d_lgcn_user" % d], g["d%d_lgcn_item" % d]])
    loss, gf, ge = oracle.bpr(fin, E0, U, batch[:, 0], batch[:, 1], batch[:, 2], 1e-4)
    np.testing.assert_allclose(loss, g["d%d_lgcn_loss" % d], rtol=2e-6)
    np.testing.assert_allclose(gf[:U], g["d%d_lgcn_gfinal_user" % d], rtol=1e-5, atol=1e-9)
    np.testing.assert_allclose(gf[U:], g["d%d_lgcn_gfinal_item" % d], rtol=1e-5, atol=1e-9)
    # full gradient = propagate_bwd(g_final) + g_ego
    gE0 = oracle.propagate_mean_bwd(*_adj(g), gf, 3, True)
    np.testing.assert_allclose(gE0[:U], g["d%d_lgcn_gbpr_user" % d], rtol=2e-5, atol=1e-9)
    np.testing.assert_allclose(gE0[U:], g["d%d_lgcn_gbpr_item" % d], rtol=2e-5, atol=1e-9)
    tot = gE0 + ge
    np.testing.assert_allclose(tot[:U], g["d%d_lgcn_grad_user" % d], rtol=2e-5, atol=1e-9)
    np.testing.assert_allclose(tot[U:], g["d%d_lgcn_grad_item" % d], rtol=2e-5, atol=1e-9)
    # MFBPR: final == ego
    loss, gf, ge = oracle.bpr(E0, E0, U, batch[:, 0], batch[:, 1], batch[:, 2], 1e-4)
    np.testing.assert_allclose(loss, g["d%d_mf_loss" % d], rtol=2e-6)
    np.testing.assert_allclose((gf + ge)[:U], g["d%d_mf_grad_user" % d], rtol=1e-5, atol=1e-9)
    np.testing.assert_allclose((gf + ge)[U:], g["d%d_mf_grad_item" % d], rtol=1e-5, atol=1e-9)


@pytest.mark.parametrize("model", ["lgcn", "mf"])
def test_trajectory_six_adam_steps(model, golden_small):
    """sample -> shuffle -> 6 x (forward, backward, Adam) exactly as trainer.py:26-56 orders it."""
    g = golden_small
    U, I = int(g["num_users"]), int(g["num_items"])
    pos = [g["pos_indices"][g["pos_indptr"][u]:g["pos_indptr"][u + 1]] for u in range(U)]
    W = np.concatenate([g["d64_init_user"], g["d64_init_item"]]).copy()
    m, v = np.zeros_like(W), np.zeros_like(W)
    lr = 1e-3 if model == "lgcn" else 1e-4
    np.random.seed(2024)
    # the golden run drew its first epoch right after seeding
    s = oracle.sample_epoch(g["train_user"], g["train_item"], pos, I)
    s = s[oracle.shuffle_perm(len(s))]
    for step in range(6):
        b = s[step * 128:(step + 1) * 128]
        if model == "lgcn":
            fin = oracle.propagate_mean(*_adj(g), W, 3, True)
            loss, gf, ge = oracle.bpr(fin, W, U, b[:, 0], b[:, 1], b[:, 2], 1e-4)
            grad = oracle.propagate_mean_bwd(*_adj(g), gf, 3, True) + ge
        else:
            loss, gf, ge = oracle.bpr(W, W, U, b[:, 0], b[:, 1], b[:, 2], 1e-4)
            grad = gf + ge
        np.testing.assert_allclose(loss, g["traj_%s_losses" % model][step], rtol=1e-5)
        oracle.adam(W, np.ascontiguousarray(grad), m, v, lr, step + 1)
        np.testing.assert_allclose(W[:U], g["traj_%s_user" % model][step], rtol=1e-4, atol=1e-7)
        np.testing.assert_allclose(W[U:], g["traj_%s_item" % model][step], rtol=1e-4, atol=1e-7)


@pytest.mark.parametrize("gname,d", [("tiny", 64), ("small", 64)])
def test_scores_and_topk_contract(gname, d, golden_tiny, golden_small):
    g = golden_tiny if gname == "tiny" else golden_small
    users = g["test_dict_users"][:48]
    R = oracle.score(g["d%d_lgcn_user" % d], g["d%d_lgcn_item" % d], users)
    np.testing.assert_allclose(R, g["d%d_lgcn_rating" % d], rtol=1e-6, atol=1e-7)
    idx = oracle.topk_reference(g["d%d_lgcn_rating" % d], 10)
    ok, msg = oracle.topk_is_valid(g["d%d_lgcn_rating" % d], idx, 10)
    assert ok, msg


def test_metrics(golden_misc):
    g = golden_misc
    r = g["metrics_r"]
    test = json.loads(str(g["metrics_test"]))
    for k in (1, 3, 5):
        got = [oracle.recall_at_k(r, k, test), oracle.precision_at_k(r, k, test), oracle.ndcg_at_k(r, k, test)]
        np.testing.assert_allclose(got, g["metrics_k%d" % k], rtol=1e-12)
    assert np.array_equal(oracle.get_label(test, g["label_pred"]), g["label"])


def test_goldens_regenerate_from_the_reference_at_head():
    """Every fixture under tests/golden/ is what oracle/regen_all.py produces TODAY from the imported reference and the
    frozen inputs under tests/golden/inputs/ (the pin must be reproducible at HEAD, not only self-contained):
    regenerate into a temp dir, compare array by array — dtype, shape, values.  Runs only where the reference tree
    exists (this container; never the GPU box).  convergence_medium.npz (a 40-epoch single-thread reference run, ~15 min)
    only with IDG_REGEN_SLOW=1 — `python oracle/regen_all.py --check` does all of it."""
    import importlib.util
    import os

    ref = os.environ.get("IDG_REFERENCE", "/root/reference")
    if not os.path.isdir(os.path.join(ref, "models")):
        pytest.skip("needs the reference tree (%s)" % ref)
    here = os.path.dirname(os.path.abspath(__file__))
    spec = importlib.util.spec_from_file_location("idg_regen_all", os.path.join(here, "..", "oracle", "regen_all.py"))
    regen = importlib.util.module_from_spec(spec)
    spec.loader.exec_module(regen)
    res = regen.check(fast=os.environ.get("IDG_REGEN_SLOW") != "1", quiet=True)
    assert len(res) >= 5
    for f, bad in res.items():
        assert not bad, "%s is not what the generators produce at HEAD: %s" % (f, bad[:6])


def test_oracle_reproduces_the_references_evaluation_on_a_wide_catalogue():
    """wide_small.npz (oracle/gen_golden_wide.py): the reference's trained LightGCN tables on 1,100 users x 33,500 items,
    its aggregate() rows, rating rows, best-64 lists and Test() dict.  The oracle — adjacency, the fmaf-chain products and
    layer mean, fp32 scores + sigmoid, masking, top-k by (score, id), the metric formulas — reproduces them from the
    frozen dataset and the tables: it is the checker the GPU tests of the default top-K path compare with, so it has to
    agree with the reference at THAT geometry too."""
    import os

    g = dict(np.load(os.path.join(os.path.dirname(os.path.abspath(__file__)), "golden", "wide_small.npz"), allow_pickle=False))
    U, I = int(g["num_users"]), int(g["num_items"])
    ptr, items = g["pos_indptr"], g["pos_indices"]
    users = np.repeat(np.arange(U, dtype=np.int64), np.diff(ptr))
    ip, ix, dv = oracle.norm_adj(U, I, users, items.astype(np.int64))
    fin = oracle.propagate_mean(ip, ix, dv, np.concatenate([g["user_w"], g["item_w"]]), 3, include_layer0=True)
    assert np.array_equal(fin[g["final_rows_of"]], g["final_rows"])  # torch.sparse.mm + mean, bit for bit
    test_users = g["test_users"]
    R = oracle.score(fin[:U], fin[U:], test_users, apply_sigmoid=True)
    for b, u in enumerate(test_users):
        R[b, items[ptr[u]:ptr[u + 1]]] = -1  # batch_test.py:62-65
    rows = np.searchsorted(test_users, g["rating_rows_of"])
    np.testing.assert_allclose(R[rows], g["rating_rows"], rtol=2e-6, atol=1e-7)
    top = oracle.topk_reference(R, 64)
    np.testing.assert_allclose(np.take_along_axis(R, top, 1), g["top64_val"], rtol=2e-6, atol=1e-7)
    ok, msg = oracle.topk_is_valid(R, g["top64_idx"][:, :20], 20, tol=2e-7)  # the reference's own top-20 under the oracle's scores
    assert ok, msg
    truth = [g["test_items"][g["test_indptr"][b]:g["test_indptr"][b + 1]].tolist() for b in range(len(test_users))]
    # the reference sums per batch of `test_batch_size` users and divides by the number of users (batch_test.py:84-91)
    bs = int(dict(zip(g["config_keys"].tolist(), g["config_values"].tolist()))["test_batch_size"])
    for j, k in enumerate(g["top_K"].tolist()):
        tot = np.zeros(3)
        for lo in range(0, len(test_users), bs):
            r = oracle.get_label(truth[lo:lo + bs], top[lo:lo + bs, :max(g["top_K"])])
            tot += [oracle.recall_at_k(r, k, truth[lo:lo + bs]), oracle.precision_at_k(r, k, truth[lo:lo + bs]),
                    oracle.ndcg_at_k(r, k, truth[lo:lo + bs])]
        tot /= len(test_users)
        np.testing.assert_allclose(tot, [g["test_recall"][j], g["test_precision"][j], g["test_ndcg"][j]], rtol=0, atol=1e-9)
