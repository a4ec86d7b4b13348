"""GPU parity at the sizes BASELINE.json's configs name, where the oracle cannot restate a whole panel in seconds:

* config 5 (LightGCN-3 d=256 on a 10M-user graph): a panel of MORE than 2^32 fp32 elements (n.d = 4.3e9, 17.2 GB) —
  every 32-bit element or byte offset in a kernel would wrap here.  The oracle checks SAMPLED rows layer by layer
  (a row of layer k needs only its neighbours' rows of layer k-1, which are read back from the device: by induction
  over the layers every sampled row of the propagation is pinned to the sequential fmaf chain), the backward
  propagation is pinned by its adjoint identity against the (verified) forward, BPR and Adam by the oracle on the
  batch's rows.
* config 4 (SimGCL-3 d=64, amazon-book shape): the shared-first-product / multi-panel row-restricted encoder passes
  against the single-purpose kernels, and the fused step against the autograd composition.
"""
import os

import numpy as np
import pytest

torch = pytest.importorskip("torch")
pytestmark = pytest.mark.gpu

from oracle import oracle  # noqa: E402


def dev(a):
    return torch.from_numpy(np.ascontiguousarray(a)).cuda()


@pytest.fixture(autouse=True)
def _free_device_memory():
    """Every test here wants most of the device: return what earlier tests left in torch's caching allocator first (the
    free-memory checks below look at the device, not at the cache — a cached 150 GB would turn a test into a skip)."""
    import gc

    gc.collect()
    torch.cuda.empty_cache()
    yield
    gc.collect()
    torch.cuda.empty_cache()


def _sampled_rows_problem(ip, ix, dv, rows, X_dev):
    """The CSR rows `rows` of (ip, ix, dv) as a small CSR over a compact column space + the gathered panel rows of
    X_dev they need (read back from the device).  Column order inside a row is preserved (the remap is monotone)."""
    rows = np.asarray(rows, dtype=np.int64)
    lens = (ip[rows + 1] - ip[rows]).astype(np.int64)
    sel = np.concatenate([np.arange(ip[r], ip[r + 1]) for r in rows]) if len(rows) else np.empty(0, np.int64)
    cols = ix[sel].astype(np.int64)
    uniq = np.unique(cols)
    sub_ptr = np.zeros(len(rows) + 1, dtype=np.int64)
    sub_ptr[1:] = np.cumsum(lens)
    sub_idx = np.searchsorted(uniq, cols).astype(np.int32)
    Xs = X_dev.index_select(0, dev(uniq)).cpu().numpy()
    return sub_ptr, sub_idx, dv[sel], Xs


def _oracle_rows(ip, ix, dv, rows, X_dev, sched):
    """(A.X)[rows] by the oracle's sequential fmaf chain (split rows in the handle's published schedule)."""
    sub_ptr, sub_idx, sub_val, Xs = _sampled_rows_problem(ip, ix, dv, rows, X_dev)
    long_rows, seg, chunk = sched
    pos = {int(r): i for i, r in enumerate(long_rows)}
    mine = [(i, pos[int(r)]) for i, r in enumerate(rows) if int(r) in pos]
    if mine:
        lr = np.array([i for i, _ in mine], dtype=np.int64)
        sl = np.array([seg[j] for _, j in mine], dtype=np.int64)
        cl = np.array([chunk[j] for _, j in mine], dtype=np.int64)
        return oracle.spmm(sub_ptr, sub_idx, sub_val, Xs, lr, sl, cl)
    return oracle.spmm(sub_ptr, sub_idx, sub_val, Xs)


def test_config5_panel_beyond_2_pow_32_elements():
    """LightGCN-3 d=256 on a graph whose [n, d] panel has more than 2^32 elements (n = 18 M, 18.4 GB per panel,
    ~150 GB resident during the step; 1.2 M rows lie beyond element 2^32): every layer of the forward and of the
    backward propagation, BPR and Adam of ONE training step, on sampled rows, bit for bit."""
    import idgrec_amd.host as H
    import idgrec_amd.ops as ops
    import idgrec_amd.synth as S
    from idgrec_amd.engine import PropagationEngine

    free, total = torch.cuda.mem_get_info()
    if free < 190 * (1 << 30):
        pytest.skip("needs ~170 GB of free HBM (MI355X: 288 GB)")
    U, I, E, d, K, B = 13_000_000, 5_000_000, 40_000_000, 256, 3, 1024
    n = U + I
    assert n * d > 2 ** 32
    users, items = S.generate(U, I, E, seed=3, match_edges=False)
    ip, ix, dv = H.build_norm_adj(U, I, users, items)
    G = ops.Graph(ip, ix, dv, n, n)
    sched = G.long_rows()
    assert len(sched[0]) > 0 and (np.asarray(sched[2]) > 0).any(), "the graph must exercise split and chunked rows"
    gen = torch.Generator(device="cuda").manual_seed(5)
    X = (torch.rand((n, d), device="cuda", generator=gen) - 0.5) * 0.2

    # the batch: reaches the far end of the panel (rows beyond 2^32 elements), with a duplicate user and item
    rng = np.random.default_rng(0)
    pos_ptr = np.zeros(U + 1, dtype=np.int64)
    pos_ptr[1:] = np.cumsum(np.bincount(users, minlength=U))
    pick = np.sort(rng.choice(len(users), 4 * B, replace=False))
    tri = H.Rng(2024).sample_epoch(users[pick], items[pick], pos_ptr, items.astype(np.int32), I)
    b = tri[rng.permutation(len(tri))[:B]].copy()
    b[0] = [U - 1, I - 1, I - 2]
    b[1] = [U - 1, I - 1, 0]
    touched = np.unique(np.concatenate([b[:, 0], U + b[:, 1], U + b[:, 2]]))
    touched_d = dev(touched)

    # rows to check: around every 2^31 / 2^32 element and byte boundary of the panel, the panel's ends, the user/item
    # seam, split rows (segments combined in LDS, chunks combined by the last arriver), part of the batch's rows (the
    # live rows of the masked epilogues), neighbours of batch rows (rows the sparse-input backward product reaches),
    # and a random sample
    marks = [0, U - 1, U, n - 1, 2 ** 31 // d, 2 ** 32 // d, 2 ** 31 // (4 * d), 2 ** 32 // (4 * d), 2 ** 33 // (4 * d),
             2 ** 34 // (4 * d)]
    near = np.concatenate([np.arange(max(m - 3, 0), min(m + 4, n)) for m in marks])
    longs = np.asarray(sched[0])
    deg = np.diff(ip)
    nbrs = np.concatenate([ix[ip[r]:ip[r + 1]][:4] for r in touched[::16]]).astype(np.int64)
    rows = np.unique(np.concatenate([near, longs[:: max(1, len(longs) // 150)], longs[np.argsort(deg[longs])[-4:]],
                                     touched[::8], nbrs, rng.integers(0, n, 1200), rng.integers(2 ** 32 // d, n, 300)]))
    rows_d = dev(rows)
    live = np.isin(rows, touched)

    def at(t):
        return t.index_select(0, rows_d).cpu().numpy()

    eng = PropagationEngine(G, U, I, d, K, include_layer0=True, params=X)
    ws = G._workspace("prop", d)  # the handle's two layer buffers: layer outputs stay there after a propagation
    panel_bytes = (n * d * 4 + 255) // 256 * 256
    L1 = ws[:n * d * 4].view(torch.float32).view(n, d)
    L2 = ws[panel_bytes:panel_bytes + n * d * 4].view(torch.float32).view(n, d)

    # ---- forward, layer by layer on the sampled rows
    fin = eng.propagate(force=True)
    y1 = _oracle_rows(ip, ix, dv, rows, X, sched)
    assert np.array_equal(at(L1), y1), "forward layer 1 differs from the fmaf chain"
    y2 = _oracle_rows(ip, ix, dv, rows, L1, sched)
    assert np.array_equal(at(L2), y2), "forward layer 2 differs"
    y3 = _oracle_rows(ip, ix, dv, rows, L2, sched)
    x0 = at(X)
    mean = (((x0 + y1) + y2) + y3) / np.float32(4.0)  # torch.mean(torch.stack([...])) order (SURVEY.md §8a note)
    assert np.array_equal(at(fin), mean), "layer mean differs"
    Y = G.spmm_raw(X)  # the plain product through the other entry point
    assert np.array_equal(at(Y), y1)
    del Y
    fin_rows = fin.index_select(0, touched_d).cpu().numpy()
    ego_rows = X.index_select(0, touched_d).cpu().numpy()
    W_old = x0.copy()

    # ---- one training step (row-restricted last forward layer, BPR, sparse-input first backward product, masked
    #      epilogues, Adam in the last epilogue)
    loss = eng.train_step(dev(b[:, 0]), dev(b[:, 1]), dev(b[:, 2])).cpu().numpy()
    torch.cuda.synchronize()
    assert np.array_equal(eng.final.index_select(0, touched_d).cpu().numpy(), fin_rows), "row-restricted forward differs"
    nu = int((touched < U).sum())  # compact problem: the touched rows renumbered 0.., users first (ids < U sort first)
    cu = np.searchsorted(touched, b[:, 0])
    cp = np.searchsorted(touched, U + b[:, 1]) - nu
    cn = np.searchsorted(touched, U + b[:, 2]) - nu
    loss_o, gf, ge = oracle.bpr(fin_rows, ego_rows, nu, cu, cp, cn, 1e-4)
    np.testing.assert_allclose(loss, loss_o, rtol=1e-5)
    gF = eng.g_final
    np.testing.assert_allclose(gF.index_select(0, touched_d).cpu().numpy(), gf, rtol=1e-4, atol=1e-9)
    # backward Horner chain, on the device's own g_final: h1 = A.g + g, h2 = A.h1 + g, grad = (g + A.h2)/4 (+ reg rows)
    g_rows = at(gF)
    g_rows[~live] = 0.0  # rows outside the batch are not part of g (never read: masked)
    h1 = _oracle_rows(ip, ix, dv, rows, gF, sched)
    h1[live] = h1[live] + g_rows[live]
    assert np.array_equal(at(L1), h1), "backward step 1 differs"
    h2 = _oracle_rows(ip, ix, dv, rows, L1, sched)
    h2[live] = h2[live] + g_rows[live]
    assert np.array_equal(at(L2), h2), "backward step 2 differs"
    t3 = _oracle_rows(ip, ix, dv, rows, L2, sched)
    want = t3.copy()
    want[live] = g_rows[live] + t3[live]
    want = want / np.float32(4.0)
    ge_rows = np.zeros_like(want)
    ge_rows[live] = ge[np.searchsorted(touched, rows[live])]
    got = at(eng.grad)
    assert np.array_equal(got[~live], want[~live]), "gradient differs outside the batch's rows"
    np.testing.assert_allclose(got[live], ge_rows[live] + want[live], rtol=1e-5, atol=1e-10)
    # Adam (first step) on the sampled rows, fed with the device's gradient rows
    W, m, v = W_old.copy(), np.zeros_like(W_old), np.zeros_like(W_old)
    oracle.adam(W, np.ascontiguousarray(got), m, v, 1e-3, 1)
    np.testing.assert_allclose(at(eng.params), W, rtol=1e-6, atol=1e-9)
    np.testing.assert_allclose(at(eng.exp_avg), m, rtol=1e-6, atol=1e-12)
    np.testing.assert_allclose(at(eng.exp_avg_sq), v, rtol=1e-6, atol=1e-20)


def test_simgcl_encoder_passes_and_fused_step_at_amazon_book_size():
    """BASELINE configs[3] (SimGCL-3 d=64, amazon-book shape, B=2048): (i) idg_propagate_views_f32 — shared first
    product, per-view perturbation, multi-panel row-restricted last layer — against the single-purpose kernels: the
    clean pass bit-equal to propagate_mean over the whole panel, restricted outputs bit-equal to the unrestricted
    ones on the requested rows; (ii) the fused training step against the same
    computation composed from the differentiable operators under autograd, same noise streams."""
    import idgrec_amd.host as H
    import idgrec_amd.ops as ops
    import idgrec_amd.synth as S
    from idgrec_amd.engine import PropagationEngine

    U, I, E = S.SHAPES["amazon-book"]
    d, K, B, eps, temperature, ssl_lambda = 64, 3, 2048, 0.05, 0.2, 0.5
    users, items = S.generate(U, I, E, seed=0)
    ip, ix, dv = H.build_norm_adj(U, I, users, items)
    n = U + I
    G = ops.Graph(ip, ix, dv, n, n)
    W0 = S.xavier_uniform_panel(U, I, d, 2024).cuda()
    pos_ptr = np.zeros(U + 1, dtype=np.int64)
    pos_ptr[1:] = np.cumsum(np.bincount(users, minlength=U))
    rng = H.Rng(2024)
    tri = rng.sample_epoch(users, items, pos_ptr, items.astype(np.int32), I)
    tri = tri[rng.shuffle_perm(len(tri))][:3 * B]
    streams = [(1234, 1), (1234, 2)]

    # (i) whole-panel passes
    full = [torch.empty_like(W0) for _ in range(3)]
    ops.propagate_views_raw(G, W0, K, False, eps, streams, full)
    assert torch.equal(full[0], G.propagate_mean_raw(W0, K, False)), "clean pass differs from propagate_mean"
    for v, (seed, sid) in zip(full[1:], streams):
        # the composition from single-purpose entry points: A.E0, perturb, then K-1 perturbed layers
        T = G.spmm_raw(W0)
        X1 = ops.perturb_raw(T, eps, seed, sid * 64)
        want = G.propagate_mean_noise_raw(X1, K - 1, True, eps, seed, sid)
        assert torch.equal(v, want), "a perturbed view differs from its composition"
        delta = (v - full[0]).norm(dim=1)
        assert float(delta.max()) < K * eps * 1.01
    b0 = tri[:B]
    bitmap = torch.zeros((n + 31) // 32, dtype=torch.int32, device="cuda")
    ops.bpr_touch_rows_raw(dev(b0[:, 0]), dev(b0[:, 1]), dev(b0[:, 2]), U, bitmap)
    rows = dev(np.unique(np.concatenate([b0[:, 0], U + b0[:, 1], U + b0[:, 2]])))
    part = [torch.full_like(W0, float("nan")) for _ in range(3)]
    ops.propagate_views_raw(G, W0, K, False, eps, streams, part, out_rows=bitmap)
    for p, f in zip(part, full):
        # (rows outside the request hold the running layer sum of the dense layers before the last one: unspecified)
        assert torch.equal(p.index_select(0, rows), f.index_select(0, rows)), "restricted rows differ from the full pass"
    del full, part

    # (ii) fused step vs autograd composition, three steps
    bt = [tuple(dev(tri[i * B:(i + 1) * B, c]) for c in range(3)) for i in range(3)]
    res = []
    for fused in (True, False):
        torch.cuda.manual_seed(77)
        ops.reset_noise_stream()
        P = W0.clone()
        loss = torch.zeros((3, 3), device="cuda")
        if fused:
            eng = PropagationEngine(G, U, I, d, K, include_layer0=False, reg_lambda=1e-4, lr=1e-3, params=P)
            eng.ssl = (eps, temperature, ssl_lambda)
            for i in range(3):
                eng.train_step(*bt[i], loss_out=loss[i])
            res.append((loss.cpu().numpy(), eng.grad.clone(), P))
        else:
            P.requires_grad_(True)
            opt = ops.Adam([P], lr=1e-3)
            for i in range(3):
                u, p_, n_ = bt[i]
                clean, v1, v2 = ops.propagate_views(G, P, K, False, eps, n_views=2)
                bpr, reg = ops.bpr_loss(clean, P, u, p_, n_, U, 1e-4)
                ssl = ssl_lambda * ops.infonce_pair(v1, v2, u, p_, U, temperature)
                loss[i] = torch.stack([bpr.detach(), reg.detach(), ssl.detach()])
                opt.zero_grad()
                (bpr + reg + ssl).backward()
                opt.step()
            res.append((loss.cpu().numpy(), P.grad.clone(), P.detach()))
    (l_f, g_f, w_f), (l_a, g_a, w_a) = res
    np.testing.assert_allclose(l_f, l_a, rtol=2e-5)
    gmax = float(g_a.abs().max())
    assert float((g_f - g_a).abs().max()) <= 1e-3 * gmax
    assert torch.allclose(w_f, w_a, rtol=1e-4, atol=1e-6)


def test_simgcl_fused_step_at_amazon_book_size_vs_the_cpu_port():
    """BASELINE configs[3] at full size against the ORACLE, not against this library: the fused SimGCL step
    (clean + two view passes, fused BPR, MFMA InfoNCE forward / backward at B = 2048 with the batch's real duplicate
    structure, ONE shared backward propagation, Adam in its epilogue) with epsilon = 0 — where the reference's step is
    deterministic — against oracle/torch_ref.RefStep(simgcl=(0, 0.2, 0.5)), the CPU port of models/SimGCL.py:62-90 that
    tests/test_torch_ref.py pins to the imported reference's own epsilon-0 trajectory.  Three steps: every loss term and
    both tables within 1e-4."""
    import idgrec_amd.host as H
    import idgrec_amd.ops as ops
    import idgrec_amd.synth as S
    from idgrec_amd.engine import PropagationEngine
    from oracle.torch_ref import RefStep

    U, I, E = S.SHAPES["amazon-book"]
    d, K, B, temperature, ssl_lambda = 64, 3, 2048, 0.2, 0.5
    users, items = S.generate(U, I, E, seed=0)
    ip, ix, dv = H.build_norm_adj(U, I, users, items)
    n = U + I
    W0 = S.xavier_uniform_panel(U, I, d, 2024)
    tri = S.draw_triples(2024, users, items, U, I, 3 * B)[0][: 3 * B]
    ref = RefStep(ip, ix, dv, U, I, W0[:U], W0[U:], n_layers=K, lr=1e-3, simgcl=(0.0, temperature, ssl_lambda))
    torch.set_num_threads(min(32, os.cpu_count() or 1))
    want = [ref.step(*(torch.from_numpy(tri[i * B:(i + 1) * B, c].copy()) for c in range(3))) for i in range(3)]
    G = ops.Graph(ip, ix, dv, n, n)
    eng = PropagationEngine(G, U, I, d, K, include_layer0=False, reg_lambda=1e-4, lr=1e-3, params=W0.cuda())
    eng.ssl = (0.0, temperature, ssl_lambda)
    eng.store_grad = False  # (as the trainer runs it)
    loss = torch.zeros((3, 3), device="cuda")
    bt = [tuple(dev(tri[i * B:(i + 1) * B, c]) for c in range(3)) for i in range(3)]
    for i in range(3):
        if i + 1 < 3:
            eng.prefetch(*bt[i + 1])
        eng.train_step(*bt[i], loss_out=loss[i])
    np.testing.assert_allclose(loss.cpu().numpy(), np.array(want), rtol=1e-4)
    assert len(np.unique(tri[:B, 0])) < B and len(np.unique(tri[:B, 1])) < B  # the batch does repeat users and items
    W = eng.params.cpu().numpy()
    for mine, ref_w in ((W[:U], ref.user_w.detach().numpy()), (W[U:], ref.item_w.detach().numpy())):
        # (Adam's quotient m / (sqrt(v) + 1e-8) amplifies last-place differences of gradient elements that are themselves of
        #  the order of rounding noise: all but a handful of elements within 1e-4, none further than a tenth of lr x steps)
        off = ~np.isclose(mine, ref_w, rtol=1e-4, atol=1e-6)
        assert off.mean() < 1e-3, off.mean()
        assert np.abs(mine - ref_w).max() < 3e-4, np.abs(mine - ref_w).max()


def test_topk_at_full_catalogue_geometry(monkeypatch):
    """Full-rank evaluation at amazon-book size in ONE call (52,643 users x 91,599 items: the launch geometry of a real
    evaluation — catalogue chunks, per-chunk lists, merge; train-item masking from the full train CSR) against the
    oracle's dense scores for a sample of users: tie-aware top-20 / top-100 (two passes) equality."""
    import idgrec_amd.ops as ops
    import idgrec_amd.synth as S

    U, I, E = S.SHAPES["amazon-book"]
    users, items = S.generate(U, I, E, seed=0)
    pos_ptr = np.zeros(U + 1, dtype=np.int64)
    pos_ptr[1:] = np.cumsum(np.bincount(users, minlength=U))
    rng = np.random.default_rng(3)
    Ue = (rng.standard_normal((U, 64)) * 0.3).astype(np.float32)
    Ie = (rng.standard_normal((I, 64)) * 0.3).astype(np.float32)
    ue, ie = dev(Ue), dev(Ie)
    ip, ix = dev(pos_ptr), dev(items.astype(np.int32))
    all_users = torch.arange(U, device="cuda")
    sample = np.unique(np.concatenate([[0, 1, 63, 64, U - 1], np.argsort(np.diff(pos_ptr))[-5:], rng.integers(0, U, 150)]))
    R = oracle.score(Ue, Ie, sample)
    for b, u in enumerate(sample):
        R[b, items[pos_ptr[u]:pos_ptr[u + 1]]] = -1
    for k in (20, 100):
        info = {}
        idx = ops.score_topk(ue, ie, all_users, k, ip, ix, info=info).cpu().numpy()
        # 823 user tiles over 91,599 items: k = 20 takes the threshold + collect form on bf16 bounds (form 3), k = 100 (two
        # passes) the producer / consumer kernel on exact scores
        assert info["form"] == (3 if k == 20 else 1), info
        if k == 20:
            assert info["calls_fallen_back"] == 0 and info["users_redone"] <= U // 1000, info
        ok, msg = oracle.topk_is_valid(R, idx[sample], k, tol=2e-6)
        assert ok, msg
        # a user's list does not depend on who shares its launch (the 160-user call runs the alternating kernel)
        part = ops.score_topk(ue, ie, dev(sample), k, ip, ix, info=info).cpu().numpy()
        assert info["form"] == 0, info
        assert np.array_equal(part, idx[sample])
        # ... and all users through the alternating kernel and the producer / consumer kernel too: same ids, same values,
        # bit for bit
        vals = ops.score_topk(ue, ie, all_users, k, ip, ix, return_values=True)[1]
        for form in (0, 1):
            with ops.topk_options(form=form):
                idx0, val0 = ops.score_topk(ue, ie, all_users, k, ip, ix, return_values=True, info=info)
            assert info["form"] == form, info
            assert np.array_equal(idx0.cpu().numpy(), idx) and torch.equal(val0, vals)


def test_training_trajectory_at_yelp_size_vs_cpu_port():
    """BASELINE configs[1] at size: 24 training steps (B = 1024) of the fused engine against the reference's op sequence
    on torch CPU (oracle/torch_ref.py, pinned to the imported reference by tests/test_torch_ref.py), same graph, same
    initial tables, same native-sampler batches: every step's two losses and the tables after the last step."""
    import idgrec_amd.host as H
    import idgrec_amd.ops as ops
    import idgrec_amd.synth as S
    from idgrec_amd.engine import PropagationEngine
    from oracle.torch_ref import RefStep

    U, I, E = S.SHAPES["yelp2018"]
    users, items = S.generate(U, I, E, seed=0)
    ip, ix, dv = H.build_norm_adj(U, I, users, items)
    n, d, K, B, steps = U + I, 64, 3, 1024, 24
    W0 = S.xavier_uniform_panel(U, I, d, 2024)
    tri = S.draw_triples(2024, users, items, U, I, steps * B)[0][: steps * B]
    G = ops.Graph(ip, ix, dv, n, n)
    eng = PropagationEngine(G, U, I, d, K, include_layer0=True, reg_lambda=1e-4, lr=1e-3, params=W0.cuda())
    t = dev(tri)
    tu, tp, tn = t[:, 0].contiguous(), t[:, 1].contiguous(), t[:, 2].contiguous()  # epoch-long id tensors, as the trainer's
    losses = torch.zeros((steps, 2), device="cuda")
    for s in range(steps):
        cur, nxt = slice(s * B, (s + 1) * B), slice((s + 1) * B, (s + 2) * B)
        if s + 1 < steps:
            eng.prefetch(tu[nxt], tp[nxt], tn[nxt])
        eng.train_step(tu[cur], tp[cur], tn[cur], loss_out=losses[s])
    torch.set_num_threads(min(32, torch.get_num_threads()))
    ref = RefStep(ip, ix, dv, U, I, W0[:U].numpy(), W0[U:].numpy(), n_layers=K, lr=1e-3)
    tc = torch.from_numpy(tri)
    ref_losses = [ref.step(tc[s * B:(s + 1) * B, 0], tc[s * B:(s + 1) * B, 1], tc[s * B:(s + 1) * B, 2]) for s in range(steps)]
    np.testing.assert_allclose(losses.cpu().numpy(), np.array(ref_losses), rtol=1e-4)
    Wr = np.concatenate([ref.user_w.detach().numpy(), ref.item_w.detach().numpy()])
    Wg = eng.params.cpu().numpy()
    # Adam's first steps move every touched weight by ~lr: compare the MOVEMENT, where it is above rounding level
    moved = np.abs(Wr - W0.numpy()) > 1e-4
    np.testing.assert_allclose((Wg - W0.numpy())[moved], (Wr - W0.numpy())[moved], rtol=2e-2, atol=2e-5)
    np.testing.assert_allclose(Wg, Wr, rtol=1e-4, atol=2e-5)


def test_sharded_step_at_config5_shape_equals_the_fused_engine():
    """BASELINE configs[4] at its REAL shape — synth-10M: 10 M users x 5 M items, 199 M edges, d = 256, K = 3, B = 1024 —
    through the user-row-sharded engine at world size 1 (4 item-panel slices, the touched-item and near-user forms
    active, gradient rows stored, owner tail, what `bench.py --gpus N` runs on every rank) and through the fused
    single-GPU engine, from the same tables and the same batches: two training steps each.  Same split schedule, same
    fmaf chains, same epilogue operations in the same order => the same BITS: both losses, FIN at the batch's rows, the
    gradient and the post-Adam tables on sampled user and item rows (batch rows, their neighbours, hub rows, slice cuts,
    random rows).  The fused engine itself is pinned to the oracle's chains on sampled rows of a wider panel above."""
    import gc

    import idgrec_amd.host as H
    import idgrec_amd.ops as ops
    import idgrec_amd.sharded as sh
    import idgrec_amd.synth as S
    from idgrec_amd.engine import PropagationEngine

    free, total = torch.cuda.mem_get_info()
    if free < 200 * (1 << 30):
        pytest.skip("needs ~150 GB of free HBM (MI355X: 288 GB)")
    U, I, E = S.SHAPES["synth-10M"]
    d, K, B, steps = 256, 3, 1024, 2
    users, items = S.generate(U, I, E, seed=0)
    tri = S.draw_triples(2024, users, items, U, I, steps * B)[0]
    g = torch.Generator().manual_seed(7)
    W_u = (torch.rand(U, d, generator=g) * 2 - 1) * (6.0 / (U + d)) ** 0.5
    W_i = (torch.rand(I, d, generator=g) * 2 - 1) * (6.0 / (I + d)) ** 0.5
    rng = np.random.default_rng(1)
    deg_u, deg_i = np.bincount(users, minlength=U), np.bincount(items, minlength=I)
    b_last = tri[(steps - 1) * B: steps * B]
    # sampled rows: the last batch's, neighbours of its users / items, the hubs, rows at the slice cuts, random ones
    s_users = np.unique(np.concatenate([b_last[:, 0], users[np.isin(items, b_last[:64, 1])][:500], np.argsort(deg_u)[-8:],
                                        [0, U - 1], rng.integers(0, U, 1500)]))
    cuts = sh.partition_users_by_nnz(deg_i, 4)
    s_items = np.unique(np.concatenate([b_last[:, 1], b_last[:, 2], items[np.isin(users, b_last[:64, 0])][:500],
                                        np.argsort(deg_i)[-8:], [0, I - 1], rng.integers(0, I, 1500),
                                        np.clip(np.concatenate([cuts[1:-1] + k for k in (-33, -1, 0, 1, 32)]), 0, I - 1)]))
    su_d, si_d = dev(s_users), dev(s_items)

    # ---- the sharded engine at world size 1
    ui, iu = sh.shard_adjacency_from_edges(users, items, U, I, 0, U)
    eng = sh.ShardedEngine(sh.HipKernels(), sh.NoComm(), ui, iu, U, I, d, K, True, 1e-4, 1e-3, batch_size=B, user_lo=0,
                           n_slices=4, item_cuts=cuts)
    del ui, iu
    assert len(eng.slices) == 4
    eng.P[:U].copy_(W_u)
    eng.item_rows(eng.P).copy_(W_i)
    batches = [eng.make_batch(tri[s * B:(s + 1) * B, 0], tri[s * B:(s + 1) * B, 1], tri[s * B:(s + 1) * B, 2]) for s in range(steps)]
    sh_loss = []
    for s in range(steps):
        if s + 1 < steps:
            eng.prefetch(batches[s + 1])
        sh_loss.append(eng.train_step(batches[s]).cpu().numpy().copy())
    eng._wait_item_table()
    torch.cuda.synchronize()
    bu, bi = dev(np.unique(b_last[:, 0])), dev(np.unique(np.concatenate([b_last[:, 1], b_last[:, 2]])))
    got = dict(P_u=eng.P[:U].index_select(0, su_d).cpu(), P_i=eng.item_rows(eng.P).index_select(0, si_d).cpu(),
               G_u=eng.G[:U].index_select(0, su_d).cpu(), G_i=eng.item_rows(eng.G).index_select(0, si_d).cpu(),
               M_u=eng.MU.index_select(0, su_d).cpu(), V_i=eng.VI[:I].index_select(0, si_d).cpu(),
               F_u=eng.FIN[:U].index_select(0, bu).cpu(), F_i=eng.item_rows(eng.FIN).index_select(0, bi).cpu())
    # (world size 1 owns every item row: the compact moment arrays are the table's, block by block in slice order)
    assert [o for o, _, _ in eng.own] == [r0 for _, r0, _, _ in eng.slices]
    del eng, batches
    gc.collect()
    torch.cuda.empty_cache()

    # ---- the fused single-GPU engine on the global adjacency
    ip, ix, dv = H.build_norm_adj(U, I, users, items)
    del users, items
    n = U + I
    G = ops.Graph(ip, ix, dv, n, n)
    ip_keep, ix_keep, dv_keep = ip, ix, dv  # (3.3 GB of host memory) for the oracle's rows below
    del ip, ix, dv
    params = torch.empty((n, d), dtype=torch.float32, device="cuda")
    params[:U].copy_(W_u)
    params[U:].copy_(W_i)
    del W_u, W_i
    fe = PropagationEngine(G, U, I, d, K, include_layer0=True, reg_lambda=1e-4, lr=1e-3, params=params)
    t = dev(tri)
    fu_loss = []
    for s in range(steps):
        sl = slice(s * B, (s + 1) * B)
        fu_loss.append(fe.train_step(t[sl, 0].contiguous(), t[sl, 1].contiguous(), t[sl, 2].contiguous()).cpu().numpy().copy())
    torch.cuda.synchronize()
    assert np.array_equal(np.stack(sh_loss), np.stack(fu_loss)), (sh_loss, fu_loss)
    want = dict(P_u=fe.params[:U].index_select(0, su_d).cpu(), P_i=fe.params[U:].index_select(0, si_d).cpu(),
                G_u=fe.grad[:U].index_select(0, su_d).cpu(), G_i=fe.grad[U:].index_select(0, si_d).cpu(),
                M_u=fe.exp_avg[:U].index_select(0, su_d).cpu(), V_i=fe.exp_avg_sq[U:].index_select(0, si_d).cpu(),
                F_u=fe.final[:U].index_select(0, bu).cpu(), F_i=fe.final[U:].index_select(0, bi).cpu())
    for key in want:
        assert torch.equal(got[key], want[key]), "%s differs between the sharded and the fused step" % key
    del got, want

    # ---- and the fused engine pinned to the ORACLE on THIS graph (not only to itself, and not only on the
    # 40 M-edge graph above — the hub rows here are ~5x longer): the dense forward layer by layer on sampled rows
    # against the sequential fmaf chain (models/LightGCN.py:43-48), the backward by its adjoint identity
    # <mean_k A^k X, g> = <X, mean_k A^k g> (A symmetric: the autograd of the same lines) in float64.
    sched = G.long_rows()
    longs, deg = np.asarray(sched[0]), np.diff(ip_keep)
    b_rows = np.unique(np.concatenate([b_last[:, 0], U + b_last[:, 1], U + b_last[:, 2]]))
    cut_rows = U + np.clip(np.concatenate([cuts[1:-1] + k for k in (-1, 0, 1)]), 0, I - 1)
    rows = np.unique(np.concatenate([[0, U - 1, U, n - 1], b_rows[::3], cut_rows, longs[:: max(1, len(longs) // 150)],
                                     longs[np.argsort(deg[longs])[-2:]], np.argsort(deg[:U])[-2:],
                                     rng.integers(0, n, 1200)]))
    assert 1500 <= len(rows) <= 3000
    rows_d = dev(rows)

    def at(t_):
        return t_.index_select(0, rows_d).cpu().numpy()

    X = fe.params
    fin = fe.propagate(force=True)  # the dense forward: layer outputs stay in the handle's two layer buffers
    ws = G._workspace("prop", d)
    panel_bytes = (n * d * 4 + 255) // 256 * 256
    L1 = ws[:n * d * 4].view(torch.float32).view(n, d)
    L2 = ws[panel_bytes:panel_bytes + n * d * 4].view(torch.float32).view(n, d)
    y1 = _oracle_rows(ip_keep, ix_keep, dv_keep, rows, X, sched)
    assert np.array_equal(at(L1), y1), "configs[4] shape: forward layer 1 differs from the fmaf chain"
    y2 = _oracle_rows(ip_keep, ix_keep, dv_keep, rows, L1, sched)
    assert np.array_equal(at(L2), y2), "configs[4] shape: forward layer 2 differs"
    y3 = _oracle_rows(ip_keep, ix_keep, dv_keep, rows, L2, sched)
    mean = (((at(X) + y1) + y2) + y3) / np.float32(4.0)
    assert np.array_equal(at(fin), mean), "configs[4] shape: layer mean differs"
    del y1, y2, y3, mean

    def dot64(a, b_):
        tot = 0.0
        for r0 in range(0, n, 1 << 20):
            tot += float((a[r0:r0 + (1 << 20)].double() * b_[r0:r0 + (1 << 20)].double()).sum())
        return tot

    gen = torch.Generator(device="cuda").manual_seed(11)
    g_out = fe.g_final  # a panel the engine owns already (15.4 GB): filled with a dense random cotangent
    g_out.copy_(torch.rand((n, d), device="cuda", generator=gen).sub_(0.5))
    back = G.propagate_mean_bwd_raw(g_out, K, True, out=fe.grad)
    lhs, rhs = dot64(fin, g_out), dot64(X, back)
    assert abs(lhs - rhs) <= 1e-6 * max(abs(lhs), abs(rhs), 1e-30), (lhs, rhs)
    # the identity is blind to a symmetric error: the backward's own Horner chain on the sampled rows as well —
    # h1 = A.g + g, h2 = A.h1 + g, grad = (g + A.h2) / 4 — each step against the fmaf chain on the device's previous step
    g_rows = at(g_out)
    h1 = _oracle_rows(ip_keep, ix_keep, dv_keep, rows, g_out, sched) + g_rows
    assert np.array_equal(at(L1), h1), "configs[4] shape: backward step 1 differs"
    h2 = _oracle_rows(ip_keep, ix_keep, dv_keep, rows, L1, sched) + g_rows
    assert np.array_equal(at(L2), h2), "configs[4] shape: backward step 2 differs"
    t3 = _oracle_rows(ip_keep, ix_keep, dv_keep, rows, L2, sched)
    assert np.array_equal(at(back), (g_rows + t3) / np.float32(4.0)), "configs[4] shape: backward result differs"
