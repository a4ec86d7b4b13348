"""User-row sharding (id-grec_amd/sharded.py): partitioner, shard extraction, and the
distributed step against the single-device result — world_size 2 over gloo."""
import os
import sys

import numpy as np
import pytest

from oracle import oracle
from tests.ranks import run_ranks

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))


def _problem(g, K, include0, B, steps, seed=0, d=64, n_slices=1):
    U, I = int(g["num_users"]), int(g["num_items"])
    W0 = np.concatenate([g["d64_init_user"], g["d64_init_item"]])
    rng = np.random.default_rng(seed)
    if d != 64:  # BASELINE config 5 is d = 256: xavier-uniform tables of that width (the checker is the oracle, not a golden)
        W0 = np.concatenate([(rng.random((m, d)) * 2 - 1) * np.sqrt(6.0 / (m + d)) for m in (U, I)]).astype(np.float32)
    tri = g["sample1"][rng.permutation(len(g["sample1"]))][: B * steps]
    # held-out items per test user (file order of test.txt) and the train CSR: what a sharded evaluation needs
    tu = g["test_dict_users"]
    t_items = [g["test_item"][g["test_user"] == u] for u in tu]
    t_ptr = np.concatenate([[0], np.cumsum([len(t) for t in t_items])]).astype(np.int64)
    return dict(indptr=g["adj_indptr"], indices=g["adj_indices"], values=g["adj_data"], W0=W0, triples=tri, U=U, I=I,
                K=K, B=B, include0=include0, n_slices=n_slices, test_users=tu, test_ptr=t_ptr,
                test_items=np.concatenate(t_items).astype(np.int64), train_ptr=g["pos_indptr"], train_items=g["pos_indices"])


def _sparse_problem(K, include0, B, steps, d=64, n_slices=2, U=600, I=800, E=1500, seed=5):
    """A graph thin enough that the items a small batch's users touch, and the users near its items, are strict subsets
    (the golden graph is too dense for that), users and items without interactions included; triples drawn from its
    edges."""
    import idgrec_amd.host as H
    import idgrec_amd.synth as S

    users, items = S.generate(U, I, E, seed=seed)
    keep = (users % 11 != 0) & (items % 7 != 0)
    users, items = users[keep], items[keep]
    ip, ix, dv = H.build_norm_adj(U, I, users, items)
    rng = np.random.default_rng(seed)
    W0 = np.concatenate([(rng.random((m, d)) * 2 - 1) * np.sqrt(6.0 / (m + d)) for m in (U, I)]).astype(np.float32)
    pick = rng.permutation(len(users))[: B * steps]
    tri = np.stack([users[pick], items[pick], rng.integers(0, I, len(pick))], axis=1).astype(np.int64)
    return dict(indptr=ip, indices=ix, values=dv, W0=W0, triples=tri, U=U, I=I, K=K, B=B, include0=include0,
                n_slices=n_slices)


def _single_device_reference(p, steps):
    """The same steps on one device, by the oracle."""
    W = p["W0"].copy()
    m, v = np.zeros_like(W), np.zeros_like(W)
    adj = (p["indptr"], p["indices"], p["values"])
    losses = []
    for s in range(steps):
        b = p["triples"][s * p["B"]:(s + 1) * p["B"]]
        fin = oracle.propagate_mean(*adj, W, p["K"], p["include0"])
        loss, gf, ge = oracle.bpr(fin, W, p["U"], b[:, 0], b[:, 1], b[:, 2], 1e-4)
        grad = oracle.propagate_mean_bwd(*adj, gf, p["K"], p["include0"]) + ge
        oracle.adam(W, np.ascontiguousarray(grad), m, v, 1e-3, s + 1)
        losses.append(loss)
    return W, fin, grad, np.stack(losses)


def _launch(mode, path, steps, world=2):
    run_ranks("sharded_worker.py", world, [mode, path, steps])
    return [dict(np.load(path + ".out%d.npz" % r)) for r in range(world)]


def _check(p, outs, steps, rtol, atol, sparse=False):
    W, fin, grad, losses = _single_device_reference(p, steps)
    U = p["U"]
    for o in outs:
        # the launch order of every step follows the overlap model of DESIGN.md §7 (sharded.IssueOrder)
        assert str(o["order_violations"]) == "", str(o["order_violations"])
        lo, hi = int(o["lo"]), int(o["hi"])
        np.testing.assert_allclose(o["losses"], losses, rtol=rtol)                  # identical on every rank
        # a training step's forward is exact where its loss reads it: the batch's users (layer K - 1 reaches them through
        # the item rows the batch touches only; with prepared batches the last user-side product produces nothing else)
        rows = o["fin_rows"]
        np.testing.assert_allclose(o["FIN"][: hi - lo][rows], fin[lo:hi][rows], rtol=rtol, atol=atol)
        it = o["fin_items"]  # a training step's forward produces the item rows its loss reads, no others
        np.testing.assert_allclose(o["FIN"][hi - lo:][it], fin[U:][it], rtol=rtol, atol=atol)
        np.testing.assert_allclose(o["G"][: hi - lo], grad[lo:hi], rtol=rtol, atol=atol)
        own = o["own_items"]  # a rank finishes the gradient of the item rows it OWNS (1/N of each slice), no others
        np.testing.assert_allclose(o["G"][hi - lo:][own], grad[U:][own], rtol=rtol, atol=atol)
        np.testing.assert_allclose(o["P"][: hi - lo], W[lo:hi], rtol=rtol, atol=atol)
        np.testing.assert_allclose(o["P"][hi - lo:], W[U:], rtol=rtol, atol=atol)
    # the owned blocks tile the item rows, and the item table — updated by its owners, all-gathered — is bit-identical
    # on every rank
    assert sorted(np.concatenate([o["own_items"] for o in outs]).tolist()) == list(range(p["I"]))
    a = outs[0]
    for b in outs[1:]:
        assert np.array_equal(a["P"][int(a["hi"]) - int(a["lo"]):], b["P"][int(b["hi"]) - int(b["lo"]):])
        assert np.array_equal(a["losses"], b["losses"])  # every rank evaluates the whole batch's loss: same bits
    if "test_users" not in p:
        return
    # sharded evaluation (users by owner, items replicated, metric sums exchanged) == the single-device Test()
    want = _single_device_test(p, W, [5, 10])
    for o in outs:
        for name in ("recall", "precision", "ndcg"):
            np.testing.assert_allclose(o["ev_" + name], want[name], rtol=1e-9, atol=1e-12)


def _single_device_test(p, W, top_k):
    """batch_test.Test (utility/utility_train/batch_test.py:37-93) on the trained tables, by the oracle."""
    adj = (p["indptr"], p["indices"], p["values"])
    fin = oracle.propagate_mean(*adj, W, p["K"], p["include0"])
    users = p["test_users"]
    R = oracle.score(fin[: p["U"]], fin[p["U"]:], users)
    for b, u in enumerate(users):
        R[b, p["train_items"][p["train_ptr"][u]:p["train_ptr"][u + 1]]] = -1
    top = oracle.topk_reference(R, max(top_k))
    truth = [p["test_items"][p["test_ptr"][j]:p["test_ptr"][j + 1]].tolist() for j in range(len(users))]
    r = oracle.get_label(truth, top)
    n = float(len(users))
    return {"recall": np.array([oracle.recall_at_k(r, k, truth) for k in top_k]) / n,
            "precision": np.array([oracle.precision_at_k(r, k, truth) for k in top_k]) / n,
            "ndcg": np.array([oracle.ndcg_at_k(r, k, truth) for k in top_k]) / n}


def test_partition_is_contiguous_and_balanced():
    import idgrec_amd.sharded as sh

    deg = np.random.default_rng(0).zipf(1.6, 10000).clip(1, 3000)
    for world in (1, 2, 3, 8):
        b = sh.partition_users_by_nnz(deg, world)
        assert b[0] == 0 and b[-1] == len(deg) and (np.diff(b) >= 0).all() and len(b) == world + 1
        loads = [deg[b[r]:b[r + 1]].sum() for r in range(world)]
        assert max(loads) - min(loads) <= 2 * deg.max()
    assert sh.partition_users_by_nnz(np.zeros(5, dtype=int), 2).tolist()[0::2] == [0, 5]


def test_global_batch_chains_and_guest_row_movers():
    """ShardedEngine.make_batch: ownership, first occurrences and the per-user chains in batch order; the two row
    movers restated in numpy (tests/sharded_worker.OracleKernels) move exactly the owned rows and add duplicates in
    list order."""
    import idgrec_amd.sharded as sh
    from tests.sharded_worker import OracleKernels

    class Eng(sh.ShardedEngine):
        def __init__(self):  # only what make_batch needs
            self.k, self.B, self.lo, self.Ug = OracleKernels(), 8, 10, 5

    users = np.array([12, 3, 12, 14, 10, 12, 99, 14])  # owned block: global users 10..14
    gb = Eng().make_batch(users, np.arange(8), np.arange(8) + 1)
    assert gb.B == 8 and gb.n_owned == 6
    assert gb.own_src.tolist() == [2, -1, 2, 4, 0, 2, -1, 4]
    assert gb.head_dst.tolist() == [2, -1, -1, 4, 0, -1, -1, -1]       # first owned occurrence of each user
    assert gb.nxt.tolist() == [2, -1, 5, 7, -1, -1, -1, -1]            # 0 -> 2 -> 5 (user 12), 3 -> 7 (user 14)
    assert gb.own_users.tolist() == [2, 2, 4, 0, 2, 4]
    k = OracleKernels()
    src = np.arange(5 * 3, dtype=np.float32).reshape(5, 3) + 1
    guest = np.full((8, 3), np.nan, dtype=np.float32)
    k.gather_rows(guest, src, gb.own_src)
    assert np.array_equal(guest[[0, 2, 5]], np.tile(src[2], (3, 1))) and np.all(guest[[1, 6]] == 0)
    g_guest = np.random.default_rng(0).standard_normal((8, 3)).astype(np.float32)
    dst = np.zeros((5, 3), dtype=np.float32)
    dst2 = np.full((5, 3), np.nan, dtype=np.float32)
    k.chain_rows2(dst, g_guest, dst2, 2 * g_guest, gb.head_dst, gb.nxt, store=False)
    want = np.zeros_like(dst)
    for t in (0, 2, 5):
        want[2] = want[2] + g_guest[t]
    for t in (3, 7):
        want[4] = want[4] + g_guest[t]
    want[0] = g_guest[4]
    assert np.array_equal(dst, want)
    # the storing form: the same sums land in rows that held anything (here NaN); rows without a chain keep it
    k.chain_rows2(dst2, g_guest, dst2.copy(), g_guest, gb.head_dst, gb.nxt, store=True)
    assert np.array_equal(dst2[[0, 2, 4]], want[[0, 2, 4]]) and np.isnan(dst2[[1, 3]]).all()


@pytest.mark.parametrize("world", [2, 3])
def test_shards_tile_the_global_adjacency(world, golden_small):
    import scipy.sparse as sp

    import idgrec_amd.sharded as sh

    g = golden_small
    U, I = int(g["num_users"]), int(g["num_items"])
    A = sp.csr_matrix((g["adj_data"], g["adj_indices"], g["adj_indptr"]), shape=(U + I, U + I))
    b = sh.partition_users_by_nnz(np.diff(g["adj_indptr"][: U + 1]), world)
    acc_iu = sp.csr_matrix((I, U), dtype=np.float32)
    for r in range(world):
        lo, hi = int(b[r]), int(b[r + 1])
        (p1, i1, v1), (p2, i2, v2) = sh.shard_adjacency(g["adj_indptr"], g["adj_indices"], g["adj_data"], U, I, lo, hi)
        ui = sp.csr_matrix((v1, i1, p1), shape=(hi - lo, I))
        assert (ui != A[lo:hi, U:]).nnz == 0                       # R_g: same values, same order
        iu = sp.csr_matrix((v2, i2, p2), shape=(I, hi - lo))
        assert (iu != A[U:, lo:hi]).nnz == 0                       # R_g^T
        assert (iu != ui.T).nnz == 0
        pad = sp.hstack([sp.csr_matrix((I, lo)), iu, sp.csr_matrix((I, U - hi))]).tocsr()
        acc_iu = acc_iu + pad
    assert (acc_iu != A[U:, :U]).nnz == 0


@pytest.mark.parametrize("world", [1, 3, 4])
def test_shards_straight_from_the_edge_list(world, golden_small):
    """shard_adjacency_from_edges (what the multi-GPU bench uses: no global CSR per rank) gives, array for array and
    bit for bit, the pieces shard_adjacency cuts out of the reference-exact global adjacency — on the golden graph
    (whose adjacency the reference itself produced) and on a generated one with empty users and items."""
    import idgrec_amd.host as H
    import idgrec_amd.sharded as sh
    import idgrec_amd.synth as S

    g = golden_small
    cases = [(int(g["num_users"]), int(g["num_items"]), g["train_user"].astype(np.int64), g["train_item"].astype(np.int64),
              (g["adj_indptr"], g["adj_indices"], g["adj_data"]))]
    U2, I2 = 700, 900
    u2, i2 = S.generate(U2, I2, 9000, seed=3)
    keep = (u2 % 17 != 0) & (i2 % 13 != 0)  # users and items without any interaction
    u2, i2 = u2[keep], i2[keep]
    cases.append((U2, I2, u2, i2, H.build_norm_adj(U2, I2, u2, i2)))
    for U, I, users, items, (ip, ix, dv) in cases:
        order = np.lexsort((items, users))
        users, items = users[order], items[order]
        b = sh.partition_users_by_nnz(np.bincount(users, minlength=U), world)
        for r in range(world):
            lo, hi = int(b[r]), int(b[r + 1])
            want = sh.shard_adjacency(ip, ix, dv, U, I, lo, hi)
            got = sh.shard_adjacency_from_edges(users, items, U, I, lo, hi)
            for a, c in zip(want, got):
                for x, y in zip(a, c):
                    assert x.dtype == y.dtype and np.array_equal(x, y)
    with pytest.raises(ValueError):
        sh.shard_adjacency_from_edges(np.array([1, 0]), np.array([0, 0]), 2, 1, 0, 2)


@pytest.mark.parametrize("K,include0,d,n_slices", [(3, True, 64, 1), (2, False, 64, 3), (1, True, 64, 2), (3, True, 256, 2)])
def test_two_ranks_gloo_cpu_match_single_device(K, include0, d, n_slices, tmp_path, golden_small):
    """n_slices > 1: the item-side products run slice by slice, each slice's all-reduce issued as soon as it exists."""
    p = _problem(golden_small, K, include0, B=160, steps=3, d=d, n_slices=n_slices)
    path = str(tmp_path / "prob.npz")
    np.savez(path, **p)
    outs = _launch("cpu", path, 3)
    _check(p, outs, 3, rtol=1e-4, atol=2e-7)


@pytest.mark.parametrize("K,include0,world", [(4, True, 2), (4, False, 2), (5, True, 3)])
def test_more_than_three_layers(K, include0, world, tmp_path, golden_small):
    """GCN_layer is a free integer (configure/LightGCN.txt:12, models/LightGCN.py:43): from four layers on the user-side
    layer sum is carried from product to product (sum_in -> sum_out) and the item-side mean takes all its terms in one
    rows kernel — left to right, torch.mean(torch.stack(...))'s order — on the golden graph and on the thin one (where
    the restricted forms of layer K - 1 are active and unproduced rows are poisoned)."""
    p = _problem(golden_small, K, include0, B=160, steps=3, d=64, n_slices=2)
    path = str(tmp_path / "prob.npz")
    np.savez(path, **p)
    _check(p, _launch("cpu", path, 3, world=world), 3, rtol=1e-4, atol=2e-7)
    p = _sparse_problem(K, include0, B=6, steps=3)
    p["degree_bound"] = 1
    path = str(tmp_path / "thin.npz")
    np.savez(path, **p)
    _check(p, _launch("cpu-deferred", path, 3, world=world), 3, rtol=1e-4, atol=2e-7)


@pytest.mark.parametrize("K,include0,world", [(3, True, 2), (3, False, 3), (2, True, 2)])
def test_touched_items_agreed_without_a_host_read_back(K, include0, world, tmp_path):
    """With the global user degrees the number of touched item rows is BOUNDED on the host (the same bound on every
    rank), the id list is built on the device into that many slots (its tail repeats the last id) and the exchanges move
    that many rows: no host synchronisation in the step.  Same result; the exchanged row count is the
    bound rounded up to 256, not the exact count."""
    p = _sparse_problem(K, include0, B=6, steps=4)
    p["degree_bound"] = 1
    path = str(tmp_path / "prob.npz")
    np.savez(path, **p)
    outs = _launch("cpu-deferred", path, 4, world=world)
    for o in outs:
        assert int(o["touched_n"]) % 256 == 0 and 0 < int(o["touched_n"]) <= p["I"]
    _check(p, outs, 4, rtol=1e-4, atol=2e-7)
    # a bound beyond the compact buffer is known before the step starts: the panel form, on every rank alike
    p["live_cap"] = 4
    np.savez(path, **p)
    outs = _launch("cpu", path, 2, world=world)
    assert all(int(o["touched_n"]) == -1 for o in outs)
    _check(p, outs, 2, rtol=1e-4, atol=2e-7)


def test_first_backward_exchange_falls_back_to_the_panel(tmp_path, golden_small):
    """More live item rows in the first backward step than the compact buffer holds (here: a buffer of 4 rows): the
    step exchanges the sliced panel instead, same result."""
    p = _problem(golden_small, 3, True, B=160, steps=2, d=64, n_slices=2)
    p["live_cap"] = 4
    path = str(tmp_path / "prob.npz")
    np.savez(path, **p)
    _check(p, _launch("cpu", path, 2), 2, rtol=1e-4, atol=2e-7)


@pytest.mark.parametrize("K,include0,world", [(3, True, 2), (3, False, 3), (2, True, 2), (2, False, 1), (3, True, 1)])
def test_restricted_forms_on_a_thin_graph(K, include0, world, tmp_path):
    """A thin graph, where the row sets of a step are strict subsets (asserted for the touched items): layer K - 1 on
    the touched items / the near users, the first backward product between the batch's rows and those sets, gradient rows
    stored instead of accumulated.  The checker-backed stub poisons every row a restricted product does not produce and
    every gradient row the scatter does not store, so a consumer reading outside its set would turn everything into NaN.
    World size 1 marks the sets locally; 2 and 3 agree on the touched items through the flag vector."""
    p = _sparse_problem(K, include0, B=6, steps=3)
    path = str(tmp_path / "prob.npz")
    np.savez(path, **p)
    outs = _launch("cpu", path, 3, world=world)
    for o in outs:
        assert (0 < int(o["touched_n"]) < p["I"] // 2) if world > 1 else int(o["touched_n"]) == -1
    _check(p, outs, 3, rtol=1e-4, atol=2e-7)


@pytest.mark.parametrize("K,include0,world,thin", [(3, True, 2, False), (3, False, 3, True), (2, True, 2, True)])
def test_collectives_are_waited_for(K, include0, world, thin, tmp_path, golden_small):
    """The same steps with a communicator whose asynchronous collectives take effect only in wait() (sliced all-reduces,
    the reduce-scatter / owner tail / all-gather chain that ends a step and is waited for in the NEXT step, compact row
    exchanges): reading a buffer before its collective was waited for, or rewriting one still in flight, would show."""
    p = _sparse_problem(K, include0, B=6, steps=4) if thin else _problem(golden_small, K, include0, B=160, steps=4, n_slices=3)
    path = str(tmp_path / "prob.npz")
    np.savez(path, **p)
    _check(p, _launch("cpu-deferred", path, 4, world=world), 4, rtol=1e-4, atol=2e-7)


@pytest.mark.parametrize("K,include0", [(3, True), (1, False)])
def test_dense_form_matches_single_device(K, include0, tmp_path, golden_small):
    """batch_sparsity=False: every product dense over zero-filled gradient panels — the plain form of the same step."""
    p = _problem(golden_small, K, include0, B=160, steps=2, d=64, n_slices=2)
    path = str(tmp_path / "prob.npz")
    np.savez(path, **p)
    _check(p, _launch("cpu-dense", path, 2), 2, rtol=1e-4, atol=2e-7)


def test_small_panels_skip_the_touched_item_forms(tmp_path, golden_small):
    """Below live_rows_min_bytes nothing is agreed: the panel all-reduces carry layer K - 1 and the first backward product."""
    p = _problem(golden_small, 3, True, B=160, steps=2, d=64, n_slices=1)
    p["min_bytes"] = 1 << 40
    path = str(tmp_path / "prob.npz")
    np.savez(path, **p)
    outs = _launch("cpu", path, 2)
    assert all(int(o["touched_n"]) == -1 for o in outs)
    _check(p, outs, 2, rtol=1e-4, atol=2e-7)


def test_three_ranks_gloo_cpu_match_single_device(tmp_path, golden_small):
    """An odd world size: three uneven user blocks (nnz-balanced), ring all-reduces over three ranks, sliced item side."""
    p = _problem(golden_small, 3, True, B=160, steps=3, d=64, n_slices=2)
    path = str(tmp_path / "prob.npz")
    np.savez(path, **p)
    outs = _launch("cpu", path, 3, world=3)
    assert len({(int(o["lo"]), int(o["hi"])) for o in outs}) == 3
    _check(p, outs, 3, rtol=1e-4, atol=2e-7)


def _wide_problem(K, include0, B, steps, thin, d=64, n_slices=8, seed=9):
    """A graph with enough item rows that EIGHT ranks cut them into EIGHT slices (inner cuts fall on multiples of 32 x world
    = 256 rows: I >= 2048), as run_sharded_bench picks from four ranks on (n_slices = 8): 1,700 users x 2,400 items; thin
    (6 k interactions: the touched items / near users of a small batch are strict subsets) or dense (40 k)."""
    p = _sparse_problem(K, include0, B, steps, d=d, n_slices=n_slices, U=1700, I=2400, E=6000 if thin else 40000, seed=seed)
    p["thin"] = thin
    return p


@pytest.mark.parametrize("world,mode,thin,K,include0", [(8, "cpu", False, 3, True), (8, "cpu-deferred", True, 3, True),
                                                       (4, "cpu-deferred", False, 3, False), (4, "cpu", True, 2, True),
                                                       (8, "cpu-deferred", False, 4, True)])
def test_four_and_eight_ranks_eight_slices(world, mode, thin, K, include0, tmp_path):
    """The HEADLINE multi-GPU configuration's code path: world 4 and 8 with the item panel in EIGHT slices —
    the owner-slice arithmetic (every slice's share of the padded panel divided by the world size, each rank the Adam
    state of its 1/N of every slice), the reduce-scatter / owner tail / all-gather hand-over from one step to the next
    and the touched-item bound (global user degrees, device-style id list) at eight ranks — over gloo against the
    single-device oracle; plain and with the communicator whose collectives take effect only in wait().  The launch
    order of every step is checked against DESIGN.md §7's overlap model on every rank (_check)."""
    steps = 3
    p = _wide_problem(K, include0, B=24 if thin else 96, steps=steps, thin=thin)
    if thin:
        p["degree_bound"] = 1
    path = str(tmp_path / "prob.npz")
    np.savez(path, **p)
    outs = _launch(mode, path, steps, world=world)
    assert all(int(o["n_slices"]) == 8 for o in outs)
    assert len({(int(o["lo"]), int(o["hi"])) for o in outs}) == world
    # every rank owns 1/world of every slice: 8 blocks per rank, the ranks' blocks tile the items (_check asserts the tiling)
    if thin:
        assert all(0 < int(o["touched_n"]) < p["I"] and int(o["touched_n"]) % 256 == 0 for o in outs)
    _check(p, outs, steps, rtol=1e-4, atol=2e-7)


def test_issue_order_checker_flags_what_it_is_there_for():
    """sharded.IssueOrder.violations on hand-made launch sequences: the designed order passes; a collective issued after
    the NEXT slice's product, a panel exchange waited for before anything was launched behind it, and an all-gather
    drained inside its own step (or before the next step's first item-side products) are each reported."""
    import idgrec_amd.sharded as sh

    def step(S=2, drain_first=False, late_issue=False, early_wait=False, own_step_wait=False, first=False):
        ev = []
        items = [("product", "item", j) for j in range(S)]
        if not first and drain_first:
            ev += [("wait", "item_table.all_gather", j) for j in range(S)]
        if late_issue:
            ev += items + [("issue", "F1.panel", j) for j in range(S)]
        else:
            for j in range(S):
                ev += [items[j], ("issue", "F1.panel", j)]
        if not first and not drain_first:
            ev += [("wait", "item_table.all_gather", j) for j in range(S)]
        if early_wait:
            ev += [("wait", "F1.panel", j) for j in range(S)] + [("product", "user")]
        else:
            ev += [("product", "user")] + items + [("wait", "F1.panel", j) for j in range(S)]
        for j in range(S):
            ev += [items[j], ("issue", "B3.reduce_scatter", j)]
        ev += [("product", "user")]
        for j in range(S):
            ev += [("wait", "B3.reduce_scatter", j), ("issue", "item_table.all_gather", j)]
        if own_step_wait:
            ev += [("wait", "item_table.all_gather", j) for j in range(S)]
        return ev

    def run(**kw):
        o = sh.IssueOrder()
        o.n_slices = 2
        for i in range(2):
            o.begin_step()
            o.events += step(first=(i == 0), **kw)
        o.end_steps()
        o.events += [("wait", "item_table.all_gather", 0)]  # the caller draining the table after the last step: not a step
        return o.violations()

    assert run() == []
    assert any("not right behind its own product" in v for v in run(late_issue=True))
    assert any("no product launched behind it" in v for v in run(early_wait=True))
    assert any("inside its own step" in v for v in run(own_step_wait=True))
    assert any("first-layer item-side" in v for v in run(drain_first=True))


@pytest.mark.gpu
@pytest.mark.parametrize("mode", ["gpu", "gpu-dense"])
@pytest.mark.parametrize("K,include0,d", [(3, True, 64), (2, False, 64), (1, True, 64), (3, True, 256), (2, False, 256)])
def test_two_ranks_hip_kernels_match_single_device(K, include0, d, mode, tmp_path, golden_small):
    """mode "gpu": prepared batches (row-restricted last forward user product, sparse first backward product,
    planned scatter, every other step through the one-batch lookahead); "gpu-dense": every product dense.
    d = 256 is the width of BASELINE config 5 (user-row shards, 8 GPUs)."""
    p = _problem(golden_small, K, include0, B=160, steps=4, d=d, n_slices=1 + (K + d) % 3)
    path = str(tmp_path / "prob.npz")
    np.savez(path, **p)
    outs = _launch(mode, path, 4)
    _check(p, outs, 4, rtol=1e-4, atol=2e-7, sparse=(mode == "gpu"))


@pytest.mark.gpu
@pytest.mark.parametrize("K,include0,d", [(3, True, 64), (3, False, 256), (2, True, 64)])
def test_restricted_forms_hip_kernels(K, include0, d, tmp_path):
    """The thin-graph case on the HIP kernels (two ranks on cuda:0): row-restricted item- and user-side products of layer
    K - 1, the first backward products between two row sets (out_rows + x_rows in one launch), the compact exchanges of
    the touched item rows, stored gradient rows, the owner tail + all-gather."""
    p = _sparse_problem(K, include0, B=6, steps=4, d=d)
    path = str(tmp_path / "prob.npz")
    np.savez(path, **p)
    outs = _launch("gpu", path, 4)
    for o in outs:
        assert 0 < int(o["touched_n"]) < p["I"] // 2
    _check(p, outs, 4, rtol=1e-4, atol=2e-7, sparse=True)


@pytest.mark.gpu
@pytest.mark.parametrize("K,include0,d", [(4, True, 64), (4, False, 256), (6, True, 64)])
def test_more_than_three_layers_hip_kernels(K, include0, d, tmp_path, golden_small):
    """K > 3 on the HIP kernels, two ranks on cuda:0: the epilogue chain of the user-side layer sum (in place: sum_in ==
    sum_out) and idg_rows_layer_mean_n_f32; thin graph with the device-built id list as well."""
    p = _problem(golden_small, K, include0, B=160, steps=3, d=d, n_slices=2)
    path = str(tmp_path / "prob.npz")
    np.savez(path, **p)
    _check(p, _launch("gpu", path, 3), 3, rtol=1e-4, atol=2e-7, sparse=True)
    p = _sparse_problem(K, include0, B=6, steps=3, d=d)
    p["degree_bound"] = 1
    path = str(tmp_path / "thin.npz")
    np.savez(path, **p)
    _check(p, _launch("gpu-async", path, 3), 3, rtol=1e-4, atol=2e-7, sparse=True)


@pytest.mark.gpu
@pytest.mark.parametrize("K,include0,d,world", [(3, True, 64, 2), (3, False, 256, 3), (2, True, 64, 2)])
def test_touched_items_without_a_host_read_back_hip_kernels(K, include0, d, world, tmp_path):
    """idg_flags_compact_f32 inside the step (two / three ranks on cuda:0): the id list of the touched items built on the
    device into the host-side bound's slots, tail repeating the last id; the rows move through it twice per step."""
    p = _sparse_problem(K, include0, B=6, steps=4, d=d)
    p["degree_bound"] = 1
    path = str(tmp_path / "prob.npz")
    np.savez(path, **p)
    outs = _launch("gpu-async", path, 4, world=world)
    for o in outs:
        assert int(o["touched_n"]) % 256 == 0 and 0 < int(o["touched_n"]) <= p["I"]
    _check(p, outs, 4, rtol=1e-4, atol=2e-7, sparse=True)


@pytest.mark.gpu
def test_step_timeline_names_every_collective(tmp_path):
    """StepTimeline / TimelineComm (what a multi-GPU bench line's `timeline` is made of): instrumented steps give the same
    result, and every collective of the step shows up under its tag with bytes, time and the step stream's stall."""
    import json

    p = _sparse_problem(3, True, B=6, steps=3, d=64)
    p["degree_bound"] = 1
    path = str(tmp_path / "prob.npz")
    np.savez(path, **p)
    outs = _launch("gpu-timeline", path, 3)
    _check(p, outs, 3, rtol=1e-4, atol=2e-7, sparse=True)
    t = json.loads(str(outs[0]["timeline"]))
    assert t["instrumented_steps"] == 3 and t["step_gpu_ms"] > 0 and t["compute_ms"] <= t["step_gpu_ms"]
    tags = set(t["collectives"])
    assert {"F1.panel", "flags", "F2.touched", "F3.items", "guest_rows", "B1.touched", "B2.panel", "B3.reduce_scatter",
            "item_table.all_gather"} <= tags, tags
    for v in t["collectives"].values():
        assert v["bytes_per_step"] > 0 and v["calls_per_step"] > 0


@pytest.mark.gpu
@pytest.mark.parametrize("K,include0,d,thin", [(3, True, 64, False), (3, True, 256, True), (2, False, 64, True)])
def test_two_ranks_hip_kernels_with_side_stream_collectives(K, include0, d, thin, tmp_path, golden_small):
    """The HIP kernels under a communicator that reduces on a side stream and is joined only by wait() (the ordering
    contract of NativeComm's second-stream route and of c10d work objects): six steps, every other one through the
    lookahead, against the single-device oracle."""
    p = _sparse_problem(K, include0, B=6, steps=6, d=d) if thin else _problem(golden_small, K, include0, B=160, steps=6, d=d, n_slices=3)
    path = str(tmp_path / "prob.npz")
    np.savez(path, **p)
    _check(p, _launch("gpu-async", path, 6), 6, rtol=1e-4, atol=2e-7, sparse=True)


@pytest.mark.gpu
@pytest.mark.parametrize("kind", ["native", "torch"])
@pytest.mark.parametrize("K,include0,d,thin", [(3, True, 64, False), (3, False, 256, True), (4, True, 64, True)])
def test_ranks_on_distinct_devices_over_rccl(K, include0, d, thin, kind, tmp_path, golden_small):
    """RCCL BETWEEN DEVICES (everything above runs collectives over gloo, simulated comms, or RCCL at world size
    1 where a collective is the identity): one rank per GPU, backend nccl, through both communicators — libidgrec's own
    (second-stream route forced on: reduce-scatter in place, the in-place all-gather left in flight into the next step)
    and torch.distributed's (reduce_scatter_tensor into a scratch block copied at wait(), all_gather_into_tensor with the
    input aliasing its block of the output).  Six steps against the single-device oracle, item table coherent across the
    ranks.  Needs at least two GPUs: skipped on the 1-GPU boxes of this pool — UNVERIFIED until a multi-GPU box runs it."""
    import torch

    world = min(torch.cuda.device_count(), 4)
    if world < 2:
        pytest.skip("needs >= 2 GPUs (RCCL between distinct devices)")
    p = _sparse_problem(K, include0, B=6, steps=6, d=d) if thin else _problem(golden_small, K, include0, B=160, steps=6, d=d, n_slices=3)
    if thin:
        p["degree_bound"] = 1
    path = str(tmp_path / "prob.npz")
    np.savez(path, **p)
    outs = _launch("nccl-" + kind, path, 6, world=world)
    assert all(bool(o["coherent"]) for o in outs)
    _check(p, outs, 6, rtol=1e-4, atol=2e-7, sparse=True)


@pytest.mark.gpu
@pytest.mark.parametrize("kind", ["native", "torch"])
@pytest.mark.parametrize("K,include0,d,thin", [(3, True, 64, False), (3, False, 256, True)])
def test_rccl_communicators_at_world_one(K, include0, d, thin, kind, tmp_path, golden_small):
    """The same worker modes as test_ranks_on_distinct_devices_over_rccl on ONE device: backend nccl at
    world size 1, where RCCL itself runs on every box of the pool — TorchComm's nccl branches (reduce_scatter_tensor into
    a scratch block copied at wait(), all_gather_into_tensor with the input aliasing its block of the output, the process
    group called directly) and NativeComm's second-stream route (forced through RCCL: at world size 1 a collective is
    otherwise not enqueued at all), six steps against the single-device oracle.  What this cannot show is RCCL BETWEEN
    devices; that stays with the test above."""
    p = _sparse_problem(K, include0, B=6, steps=6, d=d) if thin else _problem(golden_small, K, include0, B=160, steps=6, d=d, n_slices=3)
    if thin:
        p["degree_bound"] = 1
    path = str(tmp_path / "prob.npz")
    np.savez(path, **p)
    outs = _launch("nccl-" + kind, path, 6, world=1)
    assert bool(outs[0]["coherent"])
    _check(p, outs, 6, rtol=1e-4, atol=2e-7, sparse=True)


@pytest.mark.gpu
@pytest.mark.parametrize("K,include0,d", [(3, True, 64), (2, False, 64), (3, False, 256)])
def test_one_rank_hip_kernels_match_single_device(K, include0, d, tmp_path):
    """World size 1 (what `bench.py --force-sharded` runs): nothing is agreed or exchanged as rows, but the touched-item
    and near-user bitmaps — marked locally, no host synchronisation — restrict layer K - 1's products and the first
    backward step's, as on the ranks of a larger job.  Thin graph: the sets are strict subsets, so a product reading an
    unproduced row would show."""
    p = _sparse_problem(K, include0, B=6, steps=4, d=d)
    path = str(tmp_path / "prob.npz")
    np.savez(path, **p)
    outs = _launch("gpu", path, 4, world=1)
    assert int(outs[0]["touched_n"]) == -1
    _check(p, outs, 4, rtol=1e-4, atol=2e-7, sparse=True)


def test_generate_shared_is_generate(tmp_path):
    """The multi-rank bench draws its graph once per machine (rank 0) and the other ranks load it: same arrays as
    drawing in-process — also when the cache file is unusable, stale or foreign: a garbage file, a file whose
    arrays do not match its header, a well-formed file of ANOTHER graph under this graph's name, unsorted edges."""
    import json

    import idgrec_amd.synth as S

    calls = []
    want = S.generate(300, 250, 3600, seed=0)
    first = S.generate_shared(300, 250, 3600, 0, 0, lambda: calls.append(1), cache_dir=str(tmp_path))
    other = S.generate_shared(300, 250, 3600, 0, 1, lambda: calls.append(1), cache_dir=str(tmp_path))
    files = sorted(os.listdir(tmp_path))
    assert len(files) == 2 and files[1] == files[0] + ".json" and len(calls) == 4
    npy, head = tmp_path / files[0], tmp_path / files[1]
    good_npy, good_head = npy.read_bytes(), head.read_text()

    def check(rank=1):
        got = S.generate_shared(300, 250, 3600, 0, rank, lambda: None, cache_dir=str(tmp_path))
        assert np.array_equal(got[0], want[0]) and np.array_equal(got[1], want[1])

    for got in (first, other):
        assert np.array_equal(got[0], want[0]) and np.array_equal(got[1], want[1])
    npy.write_bytes(b"not an array")
    check()
    # a well-formed file of another graph (same shape of array, other edges) under this name: the fingerprint differs
    foreign = S.generate(300, 250, 3600, seed=1)
    np.save(open(npy, "wb"), np.stack([foreign[0][: len(want[0])], foreign[1][: len(want[0])]]))
    check()
    # ... and with a header that matches the foreign arrays but not the request (another seed)
    head.write_text(json.dumps(dict(json.loads(good_head), seed=1, fingerprint=S._edges_fingerprint(*foreign))))
    check()
    # unsorted edges behind a consistent header
    perm = np.random.default_rng(0).permutation(len(want[0]))
    np.save(open(npy, "wb"), np.stack([want[0][perm], want[1][perm]]))
    head.write_text(json.dumps(dict(json.loads(good_head), fingerprint=S._edges_fingerprint(want[0][perm], want[1][perm]))))
    check()
    check(rank=0)  # rank 0 finds the bad file, draws, and replaces it
    assert npy.read_bytes() == good_npy and json.loads(head.read_text()) == json.loads(good_head)


def test_parity_compare_accepts_equal_runs_and_rejects_a_broken_one():
    """`parity_vs_1gpu` of the N-rank bench line (sharded.parity_compare): the comparison itself, on a toy engine with the
    single-device engine's surface (U, params, final, train_step) — the captured record of an identical run passes with
    zero errors, rounding-sized differences pass, a 1e-3 relative error in the final rows or the tables, a loss off by
    1e-3 or different initial tables fail, and each failure shows in the field it belongs to."""
    import copy

    import torch

    from idgrec_amd.sharded import PARITY_TOL, parity_compare

    U, I, d, B, n = 50, 40, 8, 16, 3

    class Toy:
        def __init__(self):
            g = torch.Generator().manual_seed(1)
            self.U = U
            self.params = torch.randn(U + I, d, generator=g) * 0.1
            self.final = torch.zeros(U + I, d)
            self.loss = torch.zeros(2)

        def train_step(self, u, p, q):
            self.final.copy_(self.params * 0.5 + self.params.roll(1, 0) * 0.25)
            self.loss = torch.stack([self.final[u].sum() + 3.0, self.final[U + p].abs().sum()])
            self.params[u] -= 0.01 * self.final[U + p]
            self.params[U + p] -= 0.01 * self.final[u]
            return self.loss

    rng = np.random.default_rng(0)
    tri = np.stack([rng.integers(0, U, n * B), rng.integers(0, I, n * B), rng.integers(0, I, n * B)], 1).astype(np.int64)
    run = Toy()
    user_ids, item_ids = np.unique(tri[:, 0]), np.unique(np.concatenate([tri[:, 1], tri[:, 2]]))
    cap = {"triples": tri, "user_ids": user_ids, "item_ids": item_ids, "user_rows_before": run.params[user_ids].clone(),
           "item_rows_before": run.params[U + item_ids].clone(), "loss": [], "fin_users": [], "fin_items": [], "fin_item_ids": []}
    for i in range(n):
        u, p, q = (torch.from_numpy(tri[i * B:(i + 1) * B, c]) for c in range(3))
        cap["loss"].append(run.train_step(u, p, q).double().numpy().copy())
        ids = np.unique(tri[i * B:(i + 1) * B, 1:])
        cap["fin_users"].append(run.final[u].clone())
        cap["fin_items"].append(run.final[U + ids].clone())
        cap["fin_item_ids"].append(ids)
    cap["loss"] = np.stack(cap["loss"])
    cap["user_rows"], cap["item_rows"] = run.params[user_ids].clone(), run.params[U + item_ids].clone()

    res = parity_compare(Toy(), cap)
    assert res["ok"] is True and res["tol"] == PARITY_TOL == 1e-4 and res["steps"] == n
    assert res["loss_rel_err"] == res["final_rows_rel_err"] == res["table_rel_err"] == res["update_rel_err"] == 0.0
    noisy = copy.deepcopy(cap)
    noisy["user_rows"] *= 1 + 1e-6
    noisy["fin_items"][1] *= 1 - 2e-6
    res = parity_compare(Toy(), noisy)
    assert res["ok"] is True and 0 < res["table_rel_err"] < 1e-5 and 0 < res["final_rows_rel_err"] < 1e-5
    for field, breaker in (("final_rows_rel_err", lambda c: c["fin_users"][2].mul_(1.001)),
                           ("table_rel_err", lambda c: c["item_rows"].mul_(1.001)),
                           ("loss_rel_err", lambda c: c["loss"].__setitem__((1, 0), c["loss"][1, 0] * 1.001)),
                           ("initial_tables_equal", lambda c: c["user_rows_before"][0].add_(1e-7))):
        bad = copy.deepcopy(cap)
        breaker(bad)
        res = parity_compare(Toy(), bad)
        assert res["ok"] is False, field
        assert (res[field] is False) if field == "initial_tables_equal" else (res[field] > 1e-4), (field, res)
    # a run whose tables never moved: table_rel_err stays modest (the updates are small against the tables), the error of the
    # UPDATES is total — which is what update_rel_err is there to show
    still = copy.deepcopy(cap)
    still["user_rows"], still["item_rows"] = still["user_rows_before"].clone(), still["item_rows_before"].clone()
    res = parity_compare(Toy(), still)
    assert res["ok"] is False and res["update_rel_err"] > 0.99
    # ... and it is part of the verdict on its own: updates off by 5 % move the tables by far less than tol
    half = copy.deepcopy(cap)
    half["user_rows"] = half["user_rows_before"] + (half["user_rows"] - half["user_rows_before"]) * 0.95
    half["item_rows"] = half["item_rows_before"] + (half["item_rows"] - half["item_rows_before"]) * 0.95
    res = parity_compare(Toy(), half)
    assert res["update_rel_err"] > res["tol_update"] == 1e-2 and res["ok"] is False


# ------------------------------------------------------------------------------------ 24-bit panel exchange (opt-in)
def test_pack24_format_and_rank_ordered_sum():
    """idg_pack24_f32's published format, restated in oracle.pack24 / unpack24 / reduce24 (what the CPU ranks below run; the
    HIP kernels are compared with it bit for bit in test_gpu_parity.py): upper 24 bits of the fp32 word, the dropped byte
    rounded to nearest even — at most 2^-16 relative, idempotent, sign / zeros / infinities kept — four values in three
    words; the sum over blocks is one fp32 add per block IN THE ORDER GIVEN (another order gives other bits)."""
    rng = np.random.default_rng(0)
    x = (rng.standard_normal(1 << 14) * 10.0 ** rng.uniform(-25, 25, 1 << 14)).astype(np.float32)
    w = oracle.pack24(x)
    assert w.dtype == np.uint32 and w.size == x.size // 4 * 3
    y = oracle.unpack24(w)
    assert (np.abs(y.astype(np.float64) - x) <= np.abs(x.astype(np.float64)) * 2.0 ** -16).all()
    assert (y.view(np.uint32) & 0xFF == 0).all() and np.array_equal(oracle.unpack24(oracle.pack24(y)), y)
    # ties go to even: 1 + 2^-16 lies exactly between the 24-bit neighbours 1 and 1 + 2^-15
    tie = np.array([1 + 2.0 ** -16, 1 + 3 * 2.0 ** -16, -1 - 2.0 ** -16, 0.0], dtype=np.float32)
    assert oracle.unpack24(oracle.pack24(tie)).tolist() == [1.0, 1 + 2.0 ** -14, -1.0, 0.0]
    z = np.array([0.0, -0.0, np.inf, -np.inf], dtype=np.float32)
    assert np.array_equal(oracle.unpack24(oracle.pack24(z)).view(np.uint32), z.view(np.uint32))
    blocks = [oracle.pack24((rng.standard_normal(4096) * 10.0 ** r).astype(np.float32)) for r in (0, 3, -3, 3, 0, -2, 1, 2)]
    want = oracle.unpack24(blocks[0]).copy()
    for b in blocks[1:]:
        want = (want + oracle.unpack24(b)).astype(np.float32)
    assert np.array_equal(oracle.reduce24(blocks), want)
    assert not np.array_equal(oracle.reduce24(blocks[::-1]), want)  # (fp32 addition is not associative: the ORDER is the contract)


def _check_packed(p, outs, steps):
    """A run with the 24-bit exchange against the single-device oracle: losses, FIN at the batch's rows, gradients of the
    owned rows and the tables within 1e-4 (relative Frobenius norms: single elements of a gradient that cancels to ~0 carry
    the quantisation of the terms they cancel from); the replicated item table bit-identical on every rank; the owners'
    fp32 master rows within 2^-16 of the 24-bit table rows every rank computes with."""
    import json

    W, fin, grad, losses = _single_device_reference(p, steps)
    U = p["U"]

    def rel(a, b):
        return float(np.linalg.norm(a.astype(np.float64) - b) / max(np.linalg.norm(b.astype(np.float64)), 1e-300))

    for o in outs:
        assert str(o["order_violations"]) == "", str(o["order_violations"])
        st = json.loads(str(o["packed_stats"]))
        assert st["order_violations"] == [] and st["exchanges"] > 0, st
        assert st["ratio"] is None if len(outs) == 1 else abs(st["ratio"] - 0.75) < 1e-9, st  # (one rank sends nothing)
        assert st["packed_by_producer"] > 0, st  # the item-side products wrote their partials packed (y24 epilogue)
        lo, hi = int(o["lo"]), int(o["hi"])
        np.testing.assert_allclose(o["losses"], losses, rtol=1e-4)
        rows, it, own = o["fin_rows"], o["fin_items"], o["own_items"]
        assert rel(o["FIN"][: hi - lo][rows], fin[lo:hi][rows]) <= 1e-4
        assert rel(o["FIN"][hi - lo:][it], fin[U:][it]) <= 1e-4
        assert rel(o["G"][: hi - lo], grad[lo:hi]) <= 1e-4 and rel(o["G"][hi - lo:][own], grad[U:][own]) <= 1e-4
        assert rel(o["P"][: hi - lo], W[lo:hi]) <= 1e-4 and rel(o["P"][hi - lo:], W[U:]) <= 1e-4
        # the table rows a rank owns: its fp32 master against the 24-bit copy every rank (this one too) computes with
        master = o["master_rows"][: len(own)]
        table = o["P"][hi - lo:][own]
        assert (table.view(np.uint32) & 0xFF == 0).all()
        assert (np.abs(master.astype(np.float64) - table) <= np.abs(master.astype(np.float64)) * 2.0 ** -16 + 1e-45).all()
        assert rel(master, W[U:][own]) <= 1e-4
    a = outs[0]
    for b in outs[1:]:
        assert np.array_equal(a["P"][int(a["hi"]) - int(a["lo"]):], b["P"][int(b["hi"]) - int(b["lo"]):])
        assert np.array_equal(a["losses"], b["losses"])


@pytest.mark.parametrize("world,mode,thin,K,include0", [(2, "cpu+p24", False, 3, True), (4, "cpu-deferred+p24", True, 3, True),
                                                       (8, "cpu+p24", False, 3, True), (8, "cpu-deferred+p24", True, 2, False),
                                                       (4, "cpu+p24", False, 4, True)])
def test_packed_exchange_matches_single_device_and_is_reproducible(world, mode, thin, K, include0, tmp_path):
    """the two [I, d] all-reduces, the reduce-scatter / owner tail / all-gather of the last backward product
    and the touched-item row sets as an explicit exchange of 24-bit rows summed IN RANK ORDER (sharded.Packed24Comm) — world
    2, 4 and 8 with eight item slices over gloo, plain and with the communicator whose collectives take effect only in
    wait() (a second half issued before its all-to-all has landed, or a buffer reused in flight, shows): within 1e-4 of
    the single-device oracle, the replicas bit-identical, 3/4 of the fp32 bytes on the wire — and the SAME BITS when the
    run is repeated."""
    steps = 3
    p = _wide_problem(K, include0, B=24 if thin else 96, steps=steps, thin=thin)
    if thin:
        p["degree_bound"] = 1
    path = str(tmp_path / "prob.npz")
    np.savez(path, **p)
    outs = _launch(mode, path, steps, world=world)
    assert all(int(o["n_slices"]) == 8 for o in outs)
    _check_packed(p, outs, steps)
    again = _launch(mode, path, steps, world=world)
    for a, b in zip(outs, again):
        for key in ("P", "FIN", "G", "losses", "master_rows"):
            assert np.array_equal(a[key], b[key], equal_nan=True), key


def test_packed_exchange_order_checker_flags_a_serialised_pipeline():
    """Packed24Comm.order_violations on hand-made logs: the pipelined order passes; a sum / all-gather issued before the
    next slice's all-to-all (no overlap), a wait before the second half, a second half without a first are flagged."""
    import idgrec_amd.sharded as sh

    c = sh.Packed24Comm(sh.NoComm(), None)
    c.log = [("first", 0, "ar"), ("first", 1, "ar"), ("second", 0, "ar"), ("first", 2, "rs"), ("second", 1, "ar"), ("wait", 0, "ar"),
             ("second", 2, "rs"), ("wait", 1, "ar"), ("wait", 2, "rs"), ("first", 3, "ag"), ("second", 3, "ag"), ("wait", 3, "ag")]
    assert c.order_violations() == []
    c.log = [("first", 0, "ar"), ("second", 0, "ar"), ("first", 1, "ar"), ("wait", 0, "ar"), ("second", 1, "ar"), ("wait", 1, "ar")]
    assert any("no overlap" in v for v in c.order_violations())
    c.log = [("first", 0, "ar"), ("wait", 0, "ar"), ("second", 0, "ar")]
    assert any("before its second half" in v for v in c.order_violations())
    c.log = [("first", 0, "rs"), ("wait", 0, "rs")]
    assert any("without its second half" in v for v in c.order_violations())


@pytest.mark.gpu
def test_pack24_kernels_equal_the_published_format():
    """idg_pack24_f32 / idg_unpack24_f32 / idg_reduce24_f32 against the oracle's restatement, bit for bit: ragged lengths
    (the < 16 values at the end go four at a time), special values, eight blocks in rank order, the packed result written
    over one of the input blocks."""
    import torch

    import idgrec_amd.sharded as sh

    k = sh.HipKernels()
    rng = np.random.default_rng(1)
    for n in (4, 16, 20, 64 * 1000 + 12, 1 << 20):
        x = (rng.standard_normal(n) * 10.0 ** rng.uniform(-30, 30, n)).astype(np.float32)
        x[: min(n, 4)] = np.array([0.0, -0.0, np.inf, -np.inf], dtype=np.float32)[: min(n, 4)]
        src = torch.from_numpy(x).cuda()
        dst = torch.zeros(n // 4 * 3, dtype=torch.float32, device="cuda")
        k.pack24(src, dst, n)
        w = dst.cpu().numpy().view(np.uint32)
        assert np.array_equal(w, oracle.pack24(x)), n
        back = torch.zeros(n, dtype=torch.float32, device="cuda")
        k.unpack24(dst, back, n)
        assert np.array_equal(back.cpu().numpy().view(np.uint32), oracle.unpack24(w).view(np.uint32)), n
    n, N = 64 * 513 + 8, 8
    parts = [(rng.standard_normal(n) * 10.0 ** r).astype(np.float32) for r in (0, 3, -3, 3, 0, -2, 1, 2)]
    blocks = np.concatenate([oracle.pack24(p) for p in parts])
    want = oracle.reduce24([blocks[b * (n // 4 * 3):(b + 1) * (n // 4 * 3)] for b in range(N)])
    buf = torch.from_numpy(blocks.view(np.float32).copy()).cuda()
    out_f = torch.zeros(n, dtype=torch.float32, device="cuda")
    k.reduce24(buf, N, n, out_packed=buf[3 * (n // 4 * 3): 4 * (n // 4 * 3)], out_f32=out_f)  # (packed result over block 3)
    assert np.array_equal(out_f.cpu().numpy().view(np.uint32), want.view(np.uint32))
    assert np.array_equal(buf.cpu().numpy().view(np.uint32)[3 * (n // 4 * 3): 4 * (n // 4 * 3)], oracle.pack24(want))


@pytest.mark.gpu
@pytest.mark.parametrize("mode,K,include0,d,thin", [("gpu+p24", 3, True, 64, False), ("gpu-async+p24", 3, True, 256, True),
                                                    ("gpu+p24", 2, False, 64, True), ("gpu-async+p24", 4, True, 64, False)])
def test_packed_exchange_hip_kernels(mode, K, include0, d, thin, tmp_path, golden_small):
    """The 24-bit exchange on the HIP kernels, two ranks on cuda:0 over gloo — plain, and under the communicator that works on
    a side stream joined only by wait() (the exchange's own stream then waits for the all-to-all before it sums, and the
    step's stream for the unpacked panel: a missing dependency races): against the single-device oracle within 1e-4,
    replicas bit-identical, the same bits when repeated."""
    steps = 4
    p = _sparse_problem(K, include0, B=6, steps=steps, d=d) if thin else _problem(golden_small, K, include0, B=160, steps=steps, d=d, n_slices=3)
    if thin:
        p["degree_bound"] = 1
    p.pop("test_users", None)
    path = str(tmp_path / "prob.npz")
    np.savez(path, **p)
    outs = _launch(mode, path, steps)
    _check_packed(p, outs, steps)
    again = _launch(mode, path, steps)
    for a, b in zip(outs, again):
        for key in ("P", "losses", "master_rows"):
            assert np.array_equal(a[key], b[key], equal_nan=True), key


@pytest.mark.gpu
@pytest.mark.parametrize("kind", ["native", "torch"])
def test_packed_exchange_over_rccl_at_world_one(kind, tmp_path, golden_small):
    """The exchange's RCCL calls on the one device there is: backend nccl at world size 1 through both communicators —
    libidgrec's (idg_alltoall_f32 with the own block sent through the grouped ncclSend / ncclRecv, the second-stream route,
    the in-place all-gather of the packed sums) and torch.distributed's (all_to_all_single, all_gather_into_tensor)."""
    steps = 4
    p = _problem(golden_small, 3, True, B=160, steps=steps, d=64, n_slices=3)
    p.pop("test_users", None)
    path = str(tmp_path / "prob.npz")
    np.savez(path, **p)
    outs = _launch("nccl-%s+p24" % kind, path, steps, world=1)
    assert bool(outs[0]["coherent"])
    _check_packed(p, outs, steps)


@pytest.mark.gpu
@pytest.mark.parametrize("kind", ["native", "torch"])
def test_packed_exchange_between_devices_over_rccl(kind, tmp_path, golden_small):
    """RCCL BETWEEN DEVICES for the 24-bit exchange (grouped ncclSend / ncclRecv all-to-all, rank-ordered sum, all-gather of
    the packed sums): needs at least two GPUs — skipped on this pool's 1-GPU boxes, the first thing to run on a real node."""
    import torch

    world = min(torch.cuda.device_count(), 4)
    if world < 2:
        pytest.skip("needs >= 2 GPUs (RCCL between distinct devices)")
    steps = 4
    p = _problem(golden_small, 3, True, B=160, steps=steps, d=64, n_slices=3)
    p.pop("test_users", None)
    path = str(tmp_path / "prob.npz")
    np.savez(path, **p)
    outs = _launch("nccl-%s+p24" % kind, path, steps, world=world)
    assert all(bool(o["coherent"]) for o in outs)
    _check_packed(p, outs, steps)


@pytest.mark.parametrize("world,mode,thin,K", [(2, "cpu+r32", False, 3), (8, "cpu-deferred+r32", True, 3), (4, "cpu+r32", False, 4)])
def test_rank_ordered_fp32_exchange(world, mode, thin, K, tmp_path):
    """`--reduce-order rank` without the packing (RankOrderComm, bits = 32): the panel reductions as all-to-all + this
    library's sum in rank order + all-gather on fp32 blocks — RCCL's bytes on the links, no quantisation: the ordinary
    fp32 tolerances against the single-device oracle hold (_check), the exchange's halves are pipelined, and a repeated run
    gives the same bits (a fixed reduction order makes k-GPU runs reproducible)."""
    import json

    steps = 3
    p = _wide_problem(K, True, B=24 if thin else 96, steps=steps, thin=thin)
    if thin:
        p["degree_bound"] = 1
    path = str(tmp_path / "prob.npz")
    np.savez(path, **p)
    outs = _launch(mode, path, steps, world=world)
    _check(p, outs, steps, rtol=1e-4, atol=2e-7)
    for o in outs:
        st = json.loads(str(o["packed_stats"]))
        assert st["order_violations"] == [] and st["exchanges"] > 0 and abs(st["ratio"] - 1.0) < 1e-9 and st["packed_by_producer"] == 0, st
    again = _launch(mode, path, steps, world=world)
    for a, b in zip(outs, again):
        for key in ("P", "FIN", "G", "losses"):
            assert np.array_equal(a[key], b[key], equal_nan=True), key


@pytest.mark.gpu
@pytest.mark.parametrize("mode", ["gpu+r32", "gpu-async+r32"])
def test_rank_ordered_fp32_exchange_hip_kernels(mode, tmp_path, golden_small):
    """The same on the HIP kernels (idg_reduce_blocks_f32), two ranks on cuda:0, plain and under the side-stream communicator."""
    steps = 4
    p = _problem(golden_small, 3, True, B=160, steps=steps, d=64, n_slices=3)
    path = str(tmp_path / "prob.npz")
    np.savez(path, **p)
    outs = _launch(mode, path, steps)
    _check(p, outs, steps, rtol=1e-4, atol=2e-7, sparse=True)


@pytest.mark.gpu
@pytest.mark.parametrize("mode", ["gpu-async+p24", "gpu-async+r32"])
def test_explicit_exchange_is_race_free_over_many_steps(mode, tmp_path):
    """Thirty steps of the explicit exchange (24-bit and fp32 blocks) on the HIP kernels under the side-stream communicator,
    eight item slices, thin graph (the touched-item exchanges change length from batch to batch: the buffer pool's size
    classes are reused across lengths), every other step through the lookahead — twice: a buffer handed out while still in
    flight, or a second half issued before its all-to-all has landed, shows as different bits between the two runs; the
    losses stay within 1e-3 of the single-device oracle's trajectory (the 24-bit noise feeds Adam for thirty steps)."""
    steps = 30
    p = _wide_problem(3, True, B=48, steps=steps, thin=True)
    p["degree_bound"] = 1
    path = str(tmp_path / "prob.npz")
    np.savez(path, **p)
    a = _launch(mode, path, steps)
    b = _launch(mode, path, steps)
    for x, y in zip(a, b):
        for key in ("P", "FIN", "G", "losses"):
            assert np.array_equal(x[key], y[key], equal_nan=True), key
        assert str(x["order_violations"]) == ""
    _, _, _, losses = _single_device_reference(p, steps)
    np.testing.assert_allclose(a[0]["losses"], losses, rtol=1e-3)
    assert np.isfinite(a[0]["P"]).all()
