"""GPU parity: libidgrec.so's HIP kernels (through the C ABI) against the oracle and the
reference goldens.  Integer / index results and unsplit-row SpMM are compared bit for bit;
floating point elsewhere within the 1e-4 relative tolerance BASELINE.json states."""
import numpy as np
import pytest

torch = pytest.importorskip("torch")
pytestmark = pytest.mark.gpu

from oracle import oracle  # noqa: E402

RTOL = 1e-4  # BASELINE.json north_star: "within 1e-4 relative on fp32 embeddings and loss"


@pytest.fixture(scope="module")
def ops():
    import idgrec_amd.ops as ops_

    assert torch.cuda.is_available(), "GPU tests need a HIP device"
    return ops_


def dev(a):
    return torch.from_numpy(np.ascontiguousarray(a)).cuda()


def _adj(g):
    return g["adj_indptr"], g["adj_indices"], g["adj_data"]


def _graph(ops, g, **kw):
    n = int(g["num_users"]) + int(g["num_items"])
    return ops.Graph(*_adj(g), n, n, **kw)


def random_csr(n_rows, n_cols, avg, seed, hubs=()):
    rng = np.random.default_rng(seed)
    deg = np.minimum(rng.poisson(avg, n_rows), n_cols)
    deg[rng.integers(0, n_rows, max(1, n_rows // 20))] = 0  # empty rows
    for r, dg in hubs:
        deg[r] = min(dg, n_cols)
    indptr = np.zeros(n_rows + 1, dtype=np.int64)
    indptr[1:] = np.cumsum(deg)
    indices = np.concatenate([np.sort(rng.choice(n_cols, int(dg), replace=False)) for dg in deg] + [np.empty(0, int)])
    values = rng.standard_normal(len(indices)).astype(np.float32)
    return indptr, indices.astype(np.int32), values


# ------------------------------------------------------------------------------------ SpMM
@pytest.mark.parametrize("gname,d", [("tiny", 64), ("tiny", 256), ("small", 64)])
def test_spmm_bit_exact_vs_reference_torch_cpu(ops, gname, d, golden_tiny, golden_small):
    g = golden_tiny if gname == "tiny" else golden_small
    E0 = np.concatenate([g["d%d_init_user" % d], g["d%d_init_item" % d]])
    for kw in (dict(exact_order=True), dict(split_threshold=2048)):  # no row is split
        Y = ops.spmm(_graph(ops, g, **kw), dev(E0)).cpu().numpy()
        assert np.array_equal(Y, g["d%d_spmm1" % d])
    # default handle: rows above the default threshold follow the published split order
    G = _graph(ops, g)
    Y = ops.spmm(G, dev(E0)).cpu().numpy()
    assert np.array_equal(Y, oracle.spmm(*_adj(g), E0, *G.long_rows()))
    np.testing.assert_allclose(Y, g["d%d_spmm1" % d], rtol=1e-5, atol=1e-8)


@pytest.mark.parametrize("d", [32, 64, 128, 256, 512, 48, 7])
def test_spmm_all_widths_exact_order(ops, d):
    indptr, indices, values = random_csr(700, 500, 9, seed=d, hubs=[(3, 400), (650, 300)])
    X = np.random.default_rng(1).standard_normal((500, d)).astype(np.float32)
    G = ops.Graph(indptr, indices, values, 700, 500, symmetric=False, exact_order=True)
    Y = G.spmm_raw(dev(X)).cpu().numpy()
    assert np.array_equal(Y, oracle.spmm(indptr, indices, values, X))


@pytest.mark.parametrize("fused", ["0", "1"])
@pytest.mark.parametrize("d,thr", [(64, 0), (64, 64), (256, 100), (48, 64), (32, 0), (128, 200), (512, 0)])
def test_spmm_split_rows_follow_published_schedule(ops, d, thr, fused, monkeypatch):
    monkeypatch.setenv("IDG_FUSED_FIX", fused)  # separate fix-up launch (default) / in-kernel last-arriver combine
    # hub rows far above the split threshold, incl. one longer than a whole tile (2048)
    indptr, indices, values = random_csr(900, 6000, 12, seed=5, hubs=[(0, 5000), (17, 2049), (899, 700), (450, 257)])
    X = np.random.default_rng(2).standard_normal((6000, d)).astype(np.float32)
    G = ops.Graph(indptr, indices, values, 900, 6000, symmetric=False, split_threshold=thr)
    rows, seg, chunk = G.long_rows()
    assert len(rows) >= 3 and (chunk > 0).sum() >= 2 and (chunk == 0).sum() >= 1 and G.info()["n_segments"] > (chunk > 0).sum()
    Y = G.spmm_raw(dev(X)).cpu().numpy()
    # bit-exact against the oracle evaluated in the same published summation order ...
    assert np.array_equal(Y, oracle.spmm(indptr, indices, values, X, rows, seg, chunk))
    # ... and within fp32 rounding of the reference's sequential order: |err| <= c.eps.sum|a.x|
    scale = oracle.spmm(indptr, indices, np.abs(values), np.abs(X))
    assert (np.abs(Y - oracle.spmm(indptr, indices, values, X)) <= 2e-6 * scale + 1e-30).all()
    # exact-order handle on the same matrix (rows longer than one tile take the streaming path)
    Ge = ops.Graph(indptr, indices, values, 900, 6000, symmetric=False, exact_order=True)
    assert np.array_equal(Ge.spmm_raw(dev(X)).cpu().numpy(), oracle.spmm(indptr, indices, values, X))


def test_spmm_addend_empty_and_degenerate(ops):
    indptr, indices, values = random_csr(300, 300, 5, seed=9)
    X = np.random.default_rng(3).standard_normal((300, 64)).astype(np.float32)
    C = np.random.default_rng(4).standard_normal((300, 64)).astype(np.float32)
    G = ops.Graph(indptr, indices, values, 300, 300, symmetric=False)
    Y = G.spmm_raw(dev(X), addend=dev(C)).cpu().numpy()
    assert np.array_equal(Y, oracle.spmm(indptr, indices, values, X) + C)
    # all-empty matrix
    G0 = ops.Graph(np.zeros(11, dtype=np.int64), np.zeros(0, np.int32), np.zeros(0, np.float32), 10, 10)
    assert torch.count_nonzero(G0.spmm_raw(torch.ones(10, 64, device="cuda"))) == 0
    with pytest.raises(RuntimeError):
        ops.spmm(G, torch.ones(300, 64))  # CPU tensor: loud failure, no fallback


def test_spmm_autograd_nonsymmetric(ops):
    indptr, indices, values = random_csr(200, 150, 6, seed=11)
    G = ops.Graph(indptr, indices, values, 200, 150, symmetric=False)
    X = torch.randn(150, 64, device="cuda", requires_grad=True)
    gY = torch.randn(200, 64, device="cuda")
    ops.spmm(G, X).backward(gY)
    import scipy.sparse as sp

    At = sp.csr_matrix((values, indices, indptr), shape=(200, 150)).T.tocsr()
    At.sort_indices()
    ref = oracle.spmm(At.indptr, At.indices, At.data, gY.cpu().numpy())
    assert np.array_equal(X.grad.cpu().numpy(), ref)


# ------------------------------------------------------------------------------- propagate
@pytest.mark.parametrize("gname,d", [("tiny", 64), ("tiny", 256), ("small", 64)])
def test_propagate_mean_bit_exact_vs_reference_aggregate(ops, gname, d, golden_tiny, golden_small):
    g = golden_tiny if gname == "tiny" else golden_small
    U = int(g["num_users"])
    E0 = dev(np.concatenate([g["d%d_init_user" % d], g["d%d_init_item" % d]]))
    G = _graph(ops, g, exact_order=True)
    out = ops.propagate_mean(G, E0, 3, True).cpu().numpy()
    assert np.array_equal(out[:U], g["d%d_lgcn_user" % d]) and np.array_equal(out[U:], g["d%d_lgcn_item" % d])
    out = ops.propagate_mean(G, E0, 3, False).cpu().numpy()
    assert np.array_equal(out[:U], g["d%d_simgcl_user" % d]) and np.array_equal(out[U:], g["d%d_simgcl_item" % d])
    # default handle (hub rows split): within fp32 rounding of the reference
    out = ops.propagate_mean(_graph(ops, g), E0, 3, True).cpu().numpy()
    np.testing.assert_allclose(out[:U], g["d%d_lgcn_user" % d], rtol=1e-5, atol=1e-8)
    np.testing.assert_allclose(out[U:], g["d%d_lgcn_item" % d], rtol=1e-5, atol=1e-8)


@pytest.mark.parametrize("K,inc", [(1, True), (1, False), (2, True), (2, False), (4, True), (4, False)])
def test_propagate_mean_layer_counts(ops, K, inc, golden_small):
    g = golden_small
    E0 = np.concatenate([g["d64_init_user"], g["d64_init_item"]])
    G = _graph(ops, g)
    out = ops.propagate_mean(G, dev(E0), K, inc).cpu().numpy()
    assert np.array_equal(out, oracle.propagate_mean(*_adj(g), E0, K, inc, *G.long_rows()))
    Ge = _graph(ops, g, exact_order=True)
    assert np.array_equal(ops.propagate_mean(Ge, dev(E0), K, inc).cpu().numpy(), oracle.propagate_mean(*_adj(g), E0, K, inc))


@pytest.mark.parametrize("K,inc", [(3, True), (3, False), (1, True), (2, False)])
def test_propagate_mean_backward(ops, K, inc, golden_small):
    g = golden_small
    n = int(g["num_users"]) + int(g["num_items"])
    gout = np.random.default_rng(K).standard_normal((n, 64)).astype(np.float32)
    E0 = dev(np.concatenate([g["d64_init_user"], g["d64_init_item"]])).requires_grad_(True)
    G = _graph(ops, g)
    ops.propagate_mean(G, E0, K, inc).backward(dev(gout))
    ref = oracle.propagate_mean_bwd(*_adj(g), gout, K, inc)
    np.testing.assert_allclose(E0.grad.cpu().numpy(), ref, rtol=RTOL, atol=1e-6)
    # accumulate form
    base = torch.full((n, 64), 0.5, device="cuda")
    G.propagate_mean_bwd_raw(dev(gout), K, inc, out=base, accumulate=True)
    np.testing.assert_allclose(base.cpu().numpy(), ref + 0.5, rtol=RTOL, atol=1e-6)


@pytest.mark.parametrize("K,inc,d", [(3, True, 64), (2, False, 64), (1, True, 256), (3, True, 48)])
def test_masked_backward_is_bit_identical_and_never_reads_dead_rows(ops, K, inc, d, golden_small):
    """A batch reaches <= 3B rows of d loss / d final.  With the touched-row bitmap the backward
    propagation must (a) give the same bits as the dense computation on a zero-filled panel and
    (b) never read an unflagged row — those hold NaN here."""
    g = golden_small
    n = int(g["num_users"]) + int(g["num_items"])
    rng = np.random.default_rng(d + K)
    live = np.zeros(n, dtype=bool)
    live[rng.choice(n, 60, replace=False)] = True
    dense = np.zeros((n, d), dtype=np.float32)
    dense[live] = rng.standard_normal((int(live.sum()), d)).astype(np.float32)
    poisoned = np.where(live[:, None], dense, np.float32(np.nan))
    words = np.zeros((n + 31) // 32, dtype=np.uint32)
    for r in np.nonzero(live)[0]:
        words[r >> 5] |= np.uint32(1) << np.uint32(r & 31)
    for kw in (dict(), dict(exact_order=True), dict(split_threshold=64)):
        G = _graph(ops, g, **kw)
        ref = G.propagate_mean_bwd_raw(dev(dense), K, inc)
        got = G.propagate_mean_bwd_raw(dev(poisoned), K, inc, mask=dev(words.view(np.int32)))
        assert torch.equal(ref, got)
        # accumulate + mask: flagged rows are added to, every other row of the output is overwritten
        base = torch.full((n, d), 0.25, device="cuda")
        base[torch.from_numpy(~live).cuda()] = float("nan")  # never zero-filled by the caller
        G.propagate_mean_bwd_raw(dev(poisoned), K, inc, out=base, accumulate=True, mask=dev(words.view(np.int32)))
        want = ref + 0.25 * torch.from_numpy(live).cuda()[:, None]
        assert torch.allclose(base, want, rtol=1e-6, atol=1e-7)
        # round 4: the tiles' entry lists compacted to the bitmap's rows AHEAD of the product (index-only work, registered
        # for the bitmap like a unit list): the first product walks them with the ordinary kernel — same bits, dead rows
        # still never read; a library write to the bitmap drops the registration (the in-kernel form takes over)
        if d != 48:
            bm = dev(words.view(np.int32))
            ws = G.compact_inputs(bm)
            got2 = G.propagate_mean_bwd_raw(dev(poisoned), K, inc, mask=bm)
            assert torch.equal(ref, got2), "compacted-input form differs"
            ops.bitmap_clear_raw(bm, n)          # the list is dropped with the bitmap's contents ...
            bm.copy_(dev(words.view(np.int32)))  # ... (restored behind the library's back: no list is registered now)
            assert torch.equal(ref, G.propagate_mean_bwd_raw(dev(poisoned), K, inc, mask=bm))
            del ws


@pytest.mark.parametrize("K,inc,d,kw", [(3, True, 64, {}), (1, False, 64, {}), (2, True, 256, {}), (3, True, 64, dict(split_threshold=64)),
                                        (3, False, 64, dict(exact_order=True))])
def test_forward_restricted_to_batch_rows(ops, K, inc, d, kw, golden_small):
    """With a row bitmap the last layer is evaluated for the flagged rows only: identical bits there;
    with K = 1 nothing else is written at all."""
    g = golden_small
    U = int(g["num_users"])
    n = U + int(g["num_items"])
    rng = np.random.default_rng(K * 7 + d)
    users, pos, neg = rng.integers(0, U, 40), rng.integers(0, n - U, 40), rng.integers(0, n - U, 40)
    hub = int(np.argmax(np.diff(g["adj_indptr"])))  # make sure a split (hub) row is requested too
    pos[0] = hub - U if hub >= U else pos[0]
    bitmap = torch.zeros((n + 31) // 32, dtype=torch.int32, device="cuda")
    ops.bpr_touch_rows_raw(dev(users), dev(pos), dev(neg), U, bitmap)
    rows = sorted(set(users.tolist()) | {U + x for x in pos.tolist()} | {U + x for x in neg.tolist()})
    bits = bitmap.cpu().numpy().view(np.uint32)
    assert [r for r in range(n) if (bits[r >> 5] >> (r & 31)) & 1] == rows
    E0 = torch.randn(n, d, device="cuda") * 0.1
    G = _graph(ops, g, **kw)
    full = G.propagate_mean_raw(E0, K, inc)
    part = torch.full((n, d), float("nan"), device="cuda")
    G.propagate_mean_raw(E0, K, inc, out=part, out_rows=bitmap)
    idx = torch.tensor(rows, device="cuda")
    assert torch.equal(part[idx], full[idx])
    if K == 1:
        rest = torch.ones(n, dtype=torch.bool, device="cuda")
        rest[idx] = False
        assert torch.isnan(part[rest]).all()


def test_bpr_touched_bitmap_and_stored_rows(ops, golden_small):
    g = golden_small
    U = int(g["num_users"])
    n = U + int(g["num_items"])
    batch = dev(g["d64_batch"])
    u, p, ng = batch[:, 0].contiguous(), batch[:, 1].contiguous(), batch[:, 2].contiguous()
    fin = dev(np.concatenate([g["d64_lgcn_user"], g["d64_lgcn_item"]]))
    ego = dev(np.concatenate([g["d64_init_user"], g["d64_init_item"]]))
    gf0, ge0 = torch.zeros_like(fin), torch.zeros_like(ego)
    l0 = ops.bpr_fused_raw(fin, ego, u, p, ng, U, 1e-4, gf0, ge0, deterministic=True).clone()
    gf1 = torch.full_like(fin, float("nan"))  # never zero-filled
    ge1 = torch.full_like(ego, float("nan"))
    touched = torch.zeros((n + 31) // 32, dtype=torch.int32, device="cuda")
    l1 = ops.bpr_fused_raw(fin, ego, u, p, ng, U, 1e-4, gf1, ge1, deterministic=True, touched=touched)
    assert torch.equal(l0, l1)
    bits = touched.cpu().numpy().view(np.uint32)
    flagged = np.array([(bits[r >> 5] >> (r & 31)) & 1 for r in range(n)], dtype=bool)
    rows = set(u.cpu().tolist()) | {U + x for x in p.cpu().tolist()} | {U + x for x in ng.cpu().tolist()}
    assert set(np.nonzero(flagged)[0].tolist()) == rows
    fl = torch.from_numpy(flagged).cuda()
    assert torch.equal(gf1[fl], gf0[fl]) and torch.equal(ge1[fl], ge0[fl])
    assert torch.isnan(gf1[~fl]).all() and torch.isnan(ge1[~fl]).all()


@pytest.mark.parametrize("world,B,d,U,I", [(1, 128, 64, 300, 200), (2, 16, 64, 40, 30), (5, 200, 64, 300, 200), (8, 1024, 64, 3000, 2000),
                                           (3, 4000, 100, 5000, 3000), (4, 21, 7, 9, 5)])
@pytest.mark.parametrize("fold_clear", [False, True])
def test_gradient_row_messages_merge_in_rank_order(ops, world, B, d, U, I, fold_clear):
    """idg_bpr_pack_rows_f32 / idg_bpr_unpack_rows_f32 (replicas exchange gradient rows before the backward propagation):
    `world` batches scattered on one device, packed, concatenated as an all-gather would, merged — against the same
    additions in rank order in numpy float32, BIT for bit; untouched rows of the panels are never written."""
    rng = np.random.default_rng(world * 1000 + B)
    n = U + I
    fin = dev(rng.standard_normal((n, d)).astype(np.float32))
    ego = dev(rng.standard_normal((n, d)).astype(np.float32))
    words = ops.bpr_rows_message_floats(B, d)
    msgs = torch.zeros(world * words, dtype=torch.float32, device="cuda")
    want_f, want_e = np.zeros((n, d), np.float32), np.zeros((n, d), np.float32)
    seen, count = np.zeros(n, bool), np.zeros(n, np.int64)
    scale = np.float32(1.0 / world)
    losses = []
    gf, ge = torch.full((n, d), float("nan"), device="cuda"), torch.full((n, d), float("nan"), device="cuda")
    union = torch.full(((n + 31) // 32,), -1, dtype=torch.int32, device="cuda")  # somebody must clear it: pack or unpack
    for r in range(world):
        u = dev(rng.integers(0, U, B))
        p = dev(rng.integers(0, I, B))
        ng = dev(rng.integers(0, I, B))
        touched = torch.zeros((n + 31) // 32, dtype=torch.int32, device="cuda")
        ws = ops.bpr_workspace(B, d, "cuda")
        ops.bpr_plan_raw(u, p, ng, U, n, d, ws=ws)
        loss = ops.bpr_fused_raw(fin, ego, u, p, ng, U, 1e-4, gf, ge, deterministic=2, touched=touched, ws=ws)
        # fold_clear: the last pack launch also zeroes the merge's bitmap (no memset between all-gather and merge)
        last = fold_clear and r == world - 1
        ops.bpr_pack_rows_raw(ws, B, gf, loss, msgs[r * words:(r + 1) * words], clear=union if last else None,
                              clear_bits=n if last else 0)
        losses.append(loss.cpu().numpy().copy())
        rows, cnt = np.unique(np.concatenate([u.cpu().numpy(), U + p.cpu().numpy(), U + ng.cpu().numpy()]), return_counts=True)
        contrib = gf.cpu().numpy()[rows] * scale
        first = ~seen[rows]
        want_f[rows[first]] = contrib[first]
        want_f[rows[~first]] += contrib[~first]
        seen[rows] = True
        count[rows] += cnt
    reg_scale = np.float32(np.float32(np.float32(1e-4) / np.float32(B)) * scale)
    r1 = reg_scale * ego.cpu().numpy()
    live = np.nonzero(seen)[0]
    for row in live:
        acc = r1[row].copy()
        for _ in range(count[row] - 1):
            acc += r1[row]
        want_e[row] = acc
    want_l = losses[0] * scale
    for l in losses[1:]:
        want_l = want_l + l * scale
    out_f, out_e = torch.full((n, d), float("nan"), device="cuda"), torch.full((n, d), float("nan"), device="cuda")
    out_l = torch.zeros(2, device="cuda")
    ops.bpr_unpack_rows_raw(msgs, world, B, ego, 1e-4, out_f, out_e, union, out_l, touched_is_clear=fold_clear)
    bits = union.cpu().numpy().view(np.uint32)
    flagged = ((bits[np.arange(n) >> 5] >> (np.arange(n) & 31).astype(np.uint32)) & 1).astype(bool)
    assert np.array_equal(flagged, seen)
    of, oe = out_f.cpu().numpy(), out_e.cpu().numpy()
    assert np.array_equal(of[seen], want_f[seen]) and np.array_equal(oe[seen], want_e[seen])
    assert np.isnan(of[~seen]).all() and np.isnan(oe[~seen]).all()
    assert np.array_equal(out_l.cpu().numpy(), want_l.astype(np.float32))


# ------------------------------------------------------------------------------------- BPR
@pytest.mark.parametrize("gname", ["tiny", "small"])
@pytest.mark.parametrize("deterministic", [True, False])
def test_lightgcn_forward_backward_vs_reference(ops, gname, deterministic, golden_tiny, golden_small):
    g = golden_tiny if gname == "tiny" else golden_small
    U, d = int(g["num_users"]), 64
    batch = dev(g["d64_batch"])
    wu = dev(g["d64_init_user"]).requires_grad_(True)
    wi = dev(g["d64_init_item"]).requires_grad_(True)
    E0 = torch.cat([wu, wi])
    fin = ops.propagate_mean(_graph(ops, g), E0, 3, True)
    bpr, reg = ops.bpr_loss(fin, E0, batch[:, 0], batch[:, 1], batch[:, 2], U, 1e-4, deterministic)
    np.testing.assert_allclose([bpr.item(), reg.item()], g["d64_lgcn_loss"], rtol=RTOL)
    (bpr + reg).backward()
    np.testing.assert_allclose(wu.grad.cpu().numpy(), g["d64_lgcn_grad_user"], rtol=RTOL, atol=1e-8)
    np.testing.assert_allclose(wi.grad.cpu().numpy(), g["d64_lgcn_grad_item"], rtol=RTOL, atol=1e-8)


@pytest.mark.parametrize("deterministic", [True, False])
def test_mfbpr_forward_backward_vs_reference(ops, deterministic, golden_small):
    g = golden_small
    U = int(g["num_users"])
    batch = dev(g["d64_batch"])
    W = dev(np.concatenate([g["d64_init_user"], g["d64_init_item"]])).requires_grad_(True)
    bpr, reg = ops.bpr_loss(W, W, batch[:, 0], batch[:, 1], batch[:, 2], U, 1e-4, deterministic)
    np.testing.assert_allclose([bpr.item(), reg.item()], g["d64_mf_loss"], rtol=RTOL)
    (bpr + reg).backward()
    np.testing.assert_allclose(W.grad[:U].cpu().numpy(), g["d64_mf_grad_user"], rtol=RTOL, atol=1e-8)
    np.testing.assert_allclose(W.grad[U:].cpu().numpy(), g["d64_mf_grad_item"], rtol=RTOL, atol=1e-8)


def test_bpr_upstream_scaling_and_determinism(ops, golden_small):
    g = golden_small
    U = int(g["num_users"])
    n = U + int(g["num_items"])
    rng = np.random.default_rng(0)
    B = 2048  # 3B = 6144 pairs: the in-LDS sort; heavy duplication (n = 550 rows)
    users = dev(rng.integers(0, U, B))
    pos = dev(rng.integers(0, n - U, B))
    neg = dev(rng.integers(0, n - U, B))
    fin = dev(rng.standard_normal((n, 64)).astype(np.float32) * 0.3)
    ego = dev(rng.standard_normal((n, 64)).astype(np.float32) * 0.3)
    grads = []
    for _ in range(3):
        f = fin.clone().requires_grad_(True)
        e = ego.clone().requires_grad_(True)
        bpr, reg = ops.bpr_loss(f, e, users, pos, neg, U, 1e-2, True)
        (2.0 * bpr + 3.0 * reg).backward()
        grads.append((f.grad.clone(), e.grad.clone()))
    assert torch.equal(grads[0][0], grads[1][0]) and torch.equal(grads[0][0], grads[2][0])  # run-to-run identical
    assert torch.equal(grads[0][1], grads[1][1])
    loss, gf, ge = oracle.bpr(fin.cpu().numpy(), ego.cpu().numpy(), U, users.cpu().numpy(), pos.cpu().numpy(),
                              neg.cpu().numpy(), 1e-2)
    np.testing.assert_allclose(grads[0][0].cpu().numpy(), 2.0 * gf, rtol=RTOL, atol=1e-7)
    np.testing.assert_allclose(grads[0][1].cpu().numpy(), 3.0 * ge, rtol=RTOL, atol=1e-7)
    # batches above the LDS sort limit use the radix sort
    B2 = 5000
    users2, pos2, neg2 = dev(rng.integers(0, U, B2)), dev(rng.integers(0, n - U, B2)), dev(rng.integers(0, n - U, B2))
    f = fin.clone().requires_grad_(True)
    bpr, reg = ops.bpr_loss(f, ego, users2, pos2, neg2, U, 1e-2, True)
    bpr.backward()
    loss2, gf2, _ = oracle.bpr(fin.cpu().numpy(), ego.cpu().numpy(), U, users2.cpu().numpy(), pos2.cpu().numpy(),
                               neg2.cpu().numpy(), 1e-2)
    np.testing.assert_allclose(bpr.item(), loss2[0], rtol=RTOL)
    np.testing.assert_allclose(f.grad.cpu().numpy(), gf2, rtol=RTOL, atol=1e-7)


@pytest.mark.parametrize("B", [2731, 5462, 21846, 1 << 20])
def test_bpr_plan_beyond_the_lds_sort(ops, B):
    """3B > 8192 (row, slot) pairs: LDS-sorted runs of 8192 + merge-path passes (round 3: the library's own sort; hipCUB is
    gone).  3B = 8193 (a second run of ONE key), 16386 (three runs: an odd merge), 65538, and 3 x 2^20 — the
    throughput-oriented batch of BASELINE configs[4].  The plan itself is checked (keys ascending, slots ascending inside a
    run of equal keys, a permutation of the input), then loss and gradients against the oracle, run-to-run identical."""
    rng = np.random.default_rng(B)
    U, I, d = 700, 1300, 64
    n = U + I
    users, pos, neg = rng.integers(0, U, B), rng.integers(0, I, B), rng.integers(0, I, B)
    ud, pd, nd = dev(users), dev(pos), dev(neg)
    ws = ops.bpr_workspace(B, d, torch.device("cuda"))
    ops.bpr_plan_raw(ud, pd, nd, U, n, d, ws=ws)
    torch.cuda.synchronize()
    # the plan is the stable sort of (row, slot), slot = 3 t + role: users, positives, negatives of triple t.  Workspace layout
    # (idg_bpr.hip, bpr_layout): coef [B] | loss terms [B] | squares [3B] | keys [3B] | slots [3B] | SORTED keys [3B] |
    # SORTED slots [3B] | sort scratch, every region aligned to 256 bytes
    rows = np.stack([users, U + pos, U + neg], axis=1).reshape(-1)
    order = np.lexsort((np.arange(3 * B), rows))
    al = lambda x: (x + 255) // 256 * 256  # noqa: E731
    off = al(B * 4) * 2 + al(B * 12) * 3
    raw = ws.cpu().numpy()
    got_keys = raw[off: off + 12 * B].view(np.int32)
    got_slots = raw[off + al(B * 12): off + al(B * 12) + 12 * B].view(np.int32)
    assert np.array_equal(got_keys, rows[order].astype(np.int32)), "sorted row keys differ from the stable sort"
    assert np.array_equal(got_slots, order.astype(np.int32)), "sorted slots differ from the stable sort"
    fin = dev(rng.standard_normal((n, d)).astype(np.float32) * 0.3)
    ego = dev(rng.standard_normal((n, d)).astype(np.float32) * 0.3)
    outs = []
    for _ in range(2):
        gf, ge = torch.zeros_like(fin), torch.zeros_like(ego)
        loss = ops.bpr_fused_raw(fin, ego, ud, pd, nd, U, 1e-2, gf, ge, deterministic=2, ws=ws)
        outs.append((loss.clone(), gf, ge))
    assert torch.equal(outs[0][1], outs[1][1]) and torch.equal(outs[0][2], outs[1][2])
    l_o, gf_o, ge_o = oracle.bpr(fin.cpu().numpy(), ego.cpu().numpy(), U, users, pos, neg, 1e-2)
    np.testing.assert_allclose(outs[0][0].cpu().numpy(), l_o, rtol=1e-4)
    # (a row of this small table collects up to ~1500 terms at B = 2^20: the oracle adds them in the same batch order)
    np.testing.assert_allclose(outs[0][1].cpu().numpy(), gf_o, rtol=2e-4, atol=1e-6)
    np.testing.assert_allclose(outs[0][2].cpu().numpy(), ge_o, rtol=2e-4, atol=1e-6)


# ------------------------------------------------------------------------------------ Adam
def test_adam_vs_torch_and_oracle(ops):
    rng = np.random.default_rng(0)
    p0 = rng.standard_normal(10007).astype(np.float32)
    p = dev(p0).requires_grad_(True)
    q = dev(p0).requires_grad_(True)
    opt_mine = ops.Adam([p], lr=1e-3)
    opt_torch = torch.optim.Adam([q], lr=1e-3)
    po, m, v = p0.copy(), np.zeros_like(p0), np.zeros_like(p0)
    for step in range(1, 6):
        gnp = rng.standard_normal(10007).astype(np.float32) * 10 ** rng.uniform(-4, 0)
        p.grad = dev(gnp)
        q.grad = dev(gnp)
        opt_mine.step()
        opt_torch.step()
        oracle.adam(po, gnp, m, v, 1e-3, step)
        np.testing.assert_allclose(p.detach().cpu().numpy(), q.detach().cpu().numpy(), rtol=1e-5, atol=1e-8)
        np.testing.assert_allclose(p.detach().cpu().numpy(), po, rtol=1e-5, atol=1e-8)


# ------------------------------------------------------------------------------ trajectory
@pytest.mark.parametrize("model", ["lgcn", "mf"])
def test_six_training_steps_vs_reference(ops, model, golden_small):
    """sample (native, MT19937) -> shuffle -> 6 x (propagate, BPR, backward, Adam) against the
    reference's own weights after each step (trainer.py:26-56)."""
    import idgrec_amd.host as H

    g = golden_small
    U, I = int(g["num_users"]), int(g["num_items"])
    r = H.Rng(2024)
    s = r.sample_epoch(g["train_user"], g["train_item"], g["pos_indptr"], g["pos_indices"], I)
    s = s[r.shuffle_perm(len(s))]
    wu = dev(g["d64_init_user"]).requires_grad_(True)
    wi = dev(g["d64_init_item"]).requires_grad_(True)
    opt = ops.Adam([wu, wi], lr=1e-3 if model == "lgcn" else 1e-4)
    G = _graph(ops, g)
    for step in range(6):
        b = dev(s[step * 128:(step + 1) * 128])
        E0 = torch.cat([wu, wi])
        fin = ops.propagate_mean(G, E0, 3, True) if model == "lgcn" else E0
        bpr, reg = ops.bpr_loss(fin, E0, b[:, 0], b[:, 1], b[:, 2], U, 1e-4)
        np.testing.assert_allclose([bpr.item(), reg.item()], g["traj_%s_losses" % model][step], rtol=RTOL)
        opt.zero_grad()
        (bpr + reg).backward()
        opt.step()
        np.testing.assert_allclose(wu.detach().cpu().numpy(), g["traj_%s_user" % model][step], rtol=RTOL, atol=1e-7)
        np.testing.assert_allclose(wi.detach().cpu().numpy(), g["traj_%s_item" % model][step], rtol=RTOL, atol=1e-7)


@pytest.mark.parametrize("fix", ["0", "1"])
@pytest.mark.parametrize("K,inc,d", [(3, True, 64), (2, False, 64), (1, True, 64), (3, True, 256), (2, True, 128)])
def test_train_step_adam_in_epilogue_is_bit_identical(ops, K, inc, d, fix, monkeypatch):
    """engine.train_step applies Adam in the epilogue of the last backward product (idg_propagate_mean_bwd_adam_f32);
    parameters, moments and the exposed gradient equal the two-pass form bit for bit — also on rows that are
    combined in LDS, rows whose chunks meet in global memory (either combine), and the K < 2 / odd-width fallbacks."""
    monkeypatch.setenv("IDG_FUSED_FIX", fix)
    from idgrec_amd.engine import PropagationEngine

    import idgrec_amd.host as H

    rng = np.random.default_rng(K * 100 + d)
    U, I = 2500, 900
    # two hub items (degree 2000 and 700: chunked rows), a heavy tail of local rows, many short rows
    eu = np.concatenate([rng.integers(0, U, 30000), rng.choice(U, 2000, replace=False), rng.choice(U, 700, replace=False)])
    ei = np.concatenate([(rng.zipf(1.3, 30000) % I), np.full(2000, 5), np.full(700, 77)])
    eu, ei = np.unique(np.stack([eu, ei]), axis=1)
    ip, ix, dv = H.build_norm_adj(U, I, eu.astype(np.int64), ei.astype(np.int64))
    n = U + I
    W0 = (rng.standard_normal((n, d)) * 0.1).astype(np.float32)
    pick = rng.permutation(len(eu))[: 4 * 200]
    tri = np.stack([eu[pick], ei[pick], rng.integers(0, I, len(pick))], axis=1).astype(np.int64)
    tu, tp, tn = (dev(np.ascontiguousarray(tri[:, c])) for c in range(3))
    out = []
    for fuse in (True, False):
        G = ops.Graph(ip, ix, dv, n, n)
        sched = G.long_rows()
        assert (sched[2] > 0).sum() >= 2 and (sched[2] == 0).sum() >= 5
        eng = PropagationEngine(G, U, I, d, K, include_layer0=inc, params=dev(W0.copy()))
        eng.fuse_adam = fuse
        losses = [eng.train_step(tu[i * 200:(i + 1) * 200], tp[i * 200:(i + 1) * 200], tn[i * 200:(i + 1) * 200]).clone()
                  for i in range(4)]
        out.append((eng.params.clone(), eng.exp_avg.clone(), eng.exp_avg_sq.clone(), eng.grad.clone(), torch.stack(losses)))
    for a, b in zip(*out):
        assert torch.equal(a, b)
    assert not torch.equal(out[0][0].cpu(), torch.from_numpy(W0))


# --------------------------------------------------------------------------- thin dense layer
@pytest.mark.parametrize("n,d1,d2", [(69716, 64, 64), (1000, 64, 32), (257, 100, 70), (5, 3, 130)])
def test_linear_wgrad_vs_float64(ops, n, d1, d2):
    rng = np.random.default_rng(n + d1)
    X = rng.standard_normal((n, d1)).astype(np.float32)
    G = (rng.standard_normal((n, d2)) * 0.01).astype(np.float32)
    ref = X.astype(np.float64).T @ G.astype(np.float64)
    out = ops.linear_wgrad_raw(dev(X), dev(G))
    scale = np.abs(X.astype(np.float64)).T @ np.abs(G.astype(np.float64))  # |err| <= c eps sum |x g|
    assert (np.abs(out.cpu().numpy() - ref) <= 2e-6 * scale + 1e-30).all()
    assert torch.equal(out, ops.linear_wgrad_raw(dev(X), dev(G)))  # slice order is fixed: same bits
    acc = ops.linear_wgrad_raw(dev(X), dev(G), out=out.clone(), accumulate=True)
    np.testing.assert_allclose(acc.cpu().numpy(), 2 * out.cpu().numpy(), rtol=1e-6)
    # autograd wrapper against torch.matmul
    Xt, W = dev(X).requires_grad_(), dev(rng.standard_normal((d1, d2)).astype(np.float32)).requires_grad_()
    Y = ops.tall_linear(Xt, W)
    Y.backward(dev(G))
    X2, W2 = dev(X).requires_grad_(), W.detach().clone().requires_grad_()
    torch.matmul(X2, W2).backward(dev(G))
    np.testing.assert_allclose(W.grad.cpu().numpy(), W2.grad.cpu().numpy(), rtol=1e-4, atol=2e-6 * scale.max())
    assert torch.allclose(Xt.grad, X2.grad, rtol=1e-5, atol=1e-7)


@pytest.mark.parametrize("n,d", [(69716, 64), (1000, 256), (257, 100), (3, 7)])
def test_colsum_and_ngcf_wgrad_vs_float64(ops, n, d):
    """idg_colsum_f32 (the bias gradients of the differentiable NGCF tail) and — for d a multiple of 64 —
    idg_ngcf_wgrad_f32 (ALL four parameter gradients of a layer in one pass, the fused step's form) against float64:
    [W_gcn | b_gcn | W_bi | b_bi] = [side^T gT | colsum gT | bi^T gT | colsum gT], same bits on a second call."""
    import ctypes as C

    from idgrec_amd.native import check, lib

    rng = np.random.default_rng(n + d)
    G = (rng.standard_normal((n, d)) * 0.01).astype(np.float32)
    ref = G.astype(np.float64).sum(axis=0)
    got = ops.colsum_raw(dev(G))
    assert (np.abs(got.cpu().numpy() - ref) <= 2e-6 * np.abs(G.astype(np.float64)).sum(axis=0) + 1e-30).all()
    assert torch.equal(got, ops.colsum_raw(dev(G)))
    acc = ops.colsum_raw(dev(G), out=got.clone(), accumulate=True)
    np.testing.assert_allclose(acc.cpu().numpy(), 2 * got.cpu().numpy(), rtol=1e-6)
    if d % 64:
        return
    S = rng.standard_normal((n, d)).astype(np.float32)
    Bi = rng.standard_normal((n, d)).astype(np.float32)
    out = torch.empty(2 * d * d + 2 * d, device="cuda")
    ws = torch.empty(int(lib.idg_ngcf_wgrad_workspace_bytes(d, d)), dtype=torch.uint8, device="cuda")
    p_ = lambda t: C.c_void_p(t.data_ptr())  # noqa: E731
    s_d, b_d, g_d = dev(S), dev(Bi), dev(G)
    check(lib.idg_ngcf_wgrad_f32(p_(s_d), p_(b_d), p_(g_d), n, d, d, p_(out), p_(ws), ops._stream()), "idg_ngcf_wgrad_f32")
    o = out.cpu().numpy()
    for lo, left in ((0, S), (d * d + d, Bi)):
        want = left.astype(np.float64).T @ G.astype(np.float64)
        scale = np.abs(left.astype(np.float64)).T @ np.abs(G.astype(np.float64))
        assert (np.abs(o[lo:lo + d * d].reshape(d, d) - want) <= 2e-6 * scale + 1e-30).all()
    for lo in (d * d, 2 * d * d + d):
        assert (np.abs(o[lo:lo + d] - ref) <= 2e-6 * np.abs(G.astype(np.float64)).sum(axis=0) + 1e-30).all()
    out2 = torch.empty_like(out)
    check(lib.idg_ngcf_wgrad_f32(p_(s_d), p_(b_d), p_(g_d), n, d, d, p_(out2), p_(ws), ops._stream()), "idg_ngcf_wgrad_f32")
    assert torch.equal(out, out2)


@pytest.mark.parametrize("p", [0.0, 0.3])
def test_ngcf_layer_tail_vs_torch_ops(ops, p):
    """leaky_relu((S1 + b1) + (S2 + b2)) -> dropout -> (E, normalize(E)) and its backward against the stock op chain
    (models/NGCF.py:95-108), using the kernel's own dropout mask for the torch side."""
    rng = np.random.default_rng(int(p * 10))
    n, d = 3000, 64
    S1 = dev(rng.standard_normal((n, d)).astype(np.float32)).requires_grad_()
    S2 = dev(rng.standard_normal((n, d)).astype(np.float32)).requires_grad_()
    b1 = dev((rng.standard_normal((1, d)) * 0.1).astype(np.float32)).requires_grad_()
    b2 = dev((rng.standard_normal((1, d)) * 0.1).astype(np.float32)).requires_grad_()
    E, N = ops.ngcf_layer_tail(S1, S2, b1, b2, 0.2, p, stream=(1234, 7))
    a = torch.nn.functional.leaky_relu((S1 + b1) + (S2 + b2), negative_slope=0.2)
    mask = torch.where(E.detach() != 0, torch.full_like(E, 1.0 / (1.0 - p)), torch.zeros_like(E)) if p > 0 else torch.ones_like(E)
    if p > 0:
        drop = float((E == 0).float().mean())
        assert abs(drop - p) < 0.01                                     # Bernoulli(1 - p) keep mask, scaled 1 / (1 - p)
        E2, _ = ops.ngcf_layer_tail(S1, S2, b1, b2, 0.2, p, stream=(1234, 7))
        assert torch.equal(E, E2)                                       # a function of (seed, stream, row, feature)
        E3, _ = ops.ngcf_layer_tail(S1, S2, b1, b2, 0.2, p, stream=(1234, 8))
        assert not torch.equal(E, E3)
    Er = a * mask
    Nr = torch.nn.functional.normalize(Er, p=2, dim=1)
    assert torch.allclose(E, Er, rtol=1e-6, atol=1e-7) and torch.allclose(N, Nr, rtol=1e-5, atol=1e-7)
    wE, wN = dev(rng.standard_normal((n, d)).astype(np.float32)), dev(rng.standard_normal((n, d)).astype(np.float32))
    ((E * wE).sum() + (N * wN).sum()).backward()
    mine = [t.grad.clone() for t in (S1, S2, b1, b2)]
    for t in (S1, S2, b1, b2):
        t.grad = None
    ((Er * wE).sum() + (Nr * wN).sum()).backward()
    for m, t in zip(mine, (S1, S2, b1, b2)):
        assert torch.allclose(m, t.grad, rtol=2e-4, atol=2e-5 * float(t.grad.abs().max()))


@pytest.mark.parametrize("n,d1,d2", [(3001, 64, 64), (517, 128, 64), (70000, 64, 128)])
def test_ngcf_transform_vs_float64(ops, n, d1, d2):
    """side @ W1 + (ego * side) @ W2 on the fp32 matrix cores (models/NGCF.py:88-99: the two torch.matmul calls of a
    layer) and its four gradients, against the same expression in float64; rows not a multiple of the 32-row tile."""
    rng = np.random.default_rng(n)
    side = dev(rng.standard_normal((n, d1)).astype(np.float32)).requires_grad_()
    ego = dev(rng.standard_normal((n, d1)).astype(np.float32)).requires_grad_()
    W1 = dev((rng.standard_normal((d1, d2)) * 0.2).astype(np.float32)).requires_grad_()
    W2 = dev((rng.standard_normal((d1, d2)) * 0.2).astype(np.float32)).requires_grad_()
    S = ops.ngcf_transform(side, ego, W1, W2)
    w = dev(rng.standard_normal((n, d2)).astype(np.float32))
    (S * w).sum().backward()
    mine = [t.grad.clone() for t in (side, ego, W1, W2)]
    s64, e64, a64, b64 = (t.detach().double().requires_grad_() for t in (side, ego, W1, W2))
    R = s64 @ a64 + (e64 * s64) @ b64
    (R * w.double()).sum().backward()
    assert torch.allclose(S.double(), R, rtol=1e-5, atol=1e-5)
    for m, t in zip(mine, (s64, e64, a64, b64)):
        scale = float(t.grad.abs().max())
        assert torch.allclose(m.double(), t.grad, rtol=1e-4, atol=1e-5 * scale)
    # the same values as the two thin GEMMs it replaces, to fp32 rounding
    S2 = ops.tall_linear(side.detach(), W1.detach()) + ops.tall_linear(ego.detach() * side.detach(), W2.detach())
    assert torch.allclose(S, S2, rtol=1e-5, atol=1e-5)


@pytest.mark.parametrize("d,K,n_views", [(64, 3, 2), (64, 2, 1), (256, 3, 2), (32, 4, 2)])
def test_propagate_views_one_call_equals_composition(ops, d, K, n_views, monkeypatch):
    """idg_propagate_views_f32 (shared first product, ONE multi-panel restricted launch for the last layer of all
    passes, per-panel last-arriver tickets for chunked rows) against the same passes assembled from idg_spmm_f32 /
    idg_perturb_f32 / idg_propagate_mean[_noise]_f32: identical bits, with and without a row bitmap."""
    import idgrec_amd.host as H

    rng = np.random.default_rng(d + K)
    U, I = 2500, 900
    eu = np.concatenate([rng.integers(0, U, 30000), rng.choice(U, 2000, replace=False), rng.choice(U, 700, replace=False)])
    ei = np.concatenate([(rng.zipf(1.3, 30000) % I), np.full(2000, 5), np.full(700, 77)])
    eu, ei = np.unique(np.stack([eu, ei]), axis=1)
    ip, ix, dv = H.build_norm_adj(U, I, eu.astype(np.int64), ei.astype(np.int64))
    n = U + I
    G = ops.Graph(ip, ix, dv, n, n)
    assert (G.long_rows()[2] > 0).sum() >= 2
    E0 = dev((rng.standard_normal((n, d)) * 0.1).astype(np.float32))
    rows = rng.choice(n, 400, replace=False)
    rows = np.concatenate([rows, [U + 5, U + 77]])  # the chunked hub rows are requested too
    bitmap = np.zeros((n + 31) // 32, dtype=np.uint32)
    np.bitwise_or.at(bitmap, rows >> 5, (1 << (rows & 31)).astype(np.uint32))
    bm = dev(bitmap.view(np.int32))
    streams = [(1234, 5), (1234, 6)][:n_views]
    for out_rows in (None, bm):
        res = []
        for composed in (False, True):
            monkeypatch.setattr(ops, "_compose_views", [composed])
            outs = [torch.full((n, d), float("nan"), device="cuda") for _ in range(n_views + 1)]
            ops.propagate_views_raw(G, E0, K, False, 0.1, streams, outs, out_rows=out_rows)
            res.append(outs)
        sel = torch.from_numpy(rows).cuda() if out_rows is not None else slice(None)
        for a, b in zip(*res):
            assert torch.equal(a[sel], b[sel]) and not torch.isnan(a[sel]).any()
        if n_views == 2:
            assert not torch.equal(res[0][1][sel], res[0][2][sel])  # the two views differ


# --------------------------------------------------------------------------- InfoNCE
@pytest.mark.parametrize("d,B,tau", [(64, 300, 0.2), (100, 77, 0.15), (256, 1024, 0.2), (32, 2048, 0.5)])
def test_infonce_pair_vs_reference_formula(ops, d, B, tau):
    """Loss and both gradients of the fused operator against the reference's op sequence
    (unique -> gather -> normalize -> matmul -> exp/sum/log, losses.py:24-35) under torch autograd."""
    import utility.utility_function.losses as losses

    rng = np.random.default_rng(d + B)
    U, I = 700, 500
    v1 = torch.tensor(rng.standard_normal((U + I, d)).astype(np.float32), device="cuda", requires_grad=True)
    v2 = torch.tensor((rng.standard_normal((U + I, d)) * 0.5 + 0.3).astype(np.float32), device="cuda", requires_grad=True)
    users = dev(rng.integers(0, U, B))
    items = dev(rng.zipf(1.5, B) % I)
    ui, ii = torch.unique(users), torch.unique(items)
    ref_u = losses.get_InfoNCE_loss(v1[:U][ui], v2[:U][ui], tau)
    ref_i = losses.get_InfoNCE_loss(v1[U:][ii], v2[U:][ii], tau)
    (ref_u + ref_i).backward()
    r1, r2 = v1.grad.clone(), v2.grad.clone()
    v1.grad = v2.grad = None
    two = ops.infonce_pair_raw(v1.detach(), v2.detach(), users, items, U, tau)
    np.testing.assert_allclose(two.cpu().numpy(), [ref_u.item(), ref_i.item()], rtol=2e-5)
    out = ops.infonce_pair(v1, v2, users, items, U, tau)
    np.testing.assert_allclose(out.item(), (ref_u + ref_i).item(), rtol=2e-5)
    (3.0 * out).backward()
    for mine, ref in ((v1.grad, r1), (v2.grad, r2)):
        scale = ref.abs().max().item()
        np.testing.assert_allclose(mine.cpu().numpy() / 3.0, ref.cpu().numpy(), rtol=2e-4, atol=5e-6 * scale)
        assert (mine[ref == 0] == 0).all()  # rows outside the two sets carry no gradient
    # run-to-run identical bits (no atomics anywhere)
    again = ops.infonce_pair_raw(v1.detach(), v2.detach(), users, items, U, tau)
    assert torch.equal(two, again)
    # without de-duplication (SGL.py:85-86): every occurrence of an id is a row of the in-batch matrix and the
    # occurrences' gradients add up in the id's panel row
    v1.grad = v2.grad = None
    ref = losses.get_InfoNCE_loss(v1[:U][users], v2[:U][users], tau) + losses.get_InfoNCE_loss(v1[U:][items], v2[U:][items], tau)
    ref.backward()
    r1, r2 = v1.grad.clone(), v2.grad.clone()
    v1.grad = v2.grad = None
    out = ops.infonce_pair(v1, v2, users, items, U, tau, dedup=False)
    np.testing.assert_allclose(out.item(), ref.item(), rtol=2e-5)
    out.backward()
    for mine, want in ((v1.grad, r1), (v2.grad, r2)):
        np.testing.assert_allclose(mine.cpu().numpy(), want.cpu().numpy(), rtol=2e-4, atol=5e-6 * want.abs().max().item())
    g_again = torch.zeros_like(v1)
    ops.infonce_pair_raw(v1.detach(), v2.detach(), users, items, U, tau, g1=g_again, dedup=False)
    assert torch.equal(g_again, v1.grad)


def test_infonce_reference_golden_and_zero_rows(ops, golden_misc):
    g = golden_misc
    a, b = g["infonce_a"], g["infonce_b"]
    m, d = a.shape
    pad = np.zeros((3, d), dtype=np.float32)
    v1, v2 = dev(np.concatenate([a, pad])), dev(np.concatenate([b, pad]))
    users = dev(np.arange(m))
    items = dev(np.zeros(m, dtype=np.int64))  # one item row, all zeros: normalize() clamps, loss = -log(1 + 1e-5)
    loss = ops.infonce_pair_raw(v1, v2, users, items, m, 0.2).cpu().numpy()
    np.testing.assert_allclose(loss[0], g["infonce_02"], rtol=1e-5)
    np.testing.assert_allclose(loss[1], -np.log(1.0 + 1e-5), rtol=1e-3, atol=1e-7)


# --------------------------------------------------------------------------- scoring/top-K
@pytest.mark.parametrize("gname", ["tiny", "small"])
def test_rating_matrix_vs_reference(ops, gname, golden_tiny, golden_small):
    g = golden_tiny if gname == "tiny" else golden_small
    users = dev(g["test_dict_users"][:48])
    R = ops.score_dense(dev(g["d64_lgcn_user"]), dev(g["d64_lgcn_item"]), users).cpu().numpy()
    np.testing.assert_allclose(R, g["d64_lgcn_rating"], rtol=1e-5, atol=1e-6)
    R = ops.score_dense(dev(g["d64_init_user"]), dev(g["d64_init_item"]), users).cpu().numpy()
    np.testing.assert_allclose(R, g["d64_mf_rating"], rtol=1e-5, atol=1e-6)


@pytest.mark.parametrize("d", [64, 256, 96, 20])
def test_score_dense_widths_and_ragged_shapes(ops, d):
    rng = np.random.default_rng(d)
    Uu, Vv = rng.standard_normal((77, d)).astype(np.float32), rng.standard_normal((1031, d)).astype(np.float32)
    users = rng.integers(0, 77, 45)
    for sig in (True, False):
        R = ops.score_dense(dev(Uu), dev(Vv), dev(users), apply_sigmoid=sig).cpu().numpy()
        np.testing.assert_allclose(R, oracle.score(Uu, Vv, users, sig), rtol=2e-5, atol=2e-5)


@pytest.fixture(params=["alternating", "producer-consumer"])
def topk_form(request, monkeypatch):
    """The fused top-K has two exact-score forms (every wave alternating between scoring and selecting; producer and
    consumer waves) chosen by the call's geometry — every contract test runs on both (ops.topk_option('form', ..) forces one) — and the
    threshold + collect form on bf16 bound scores for calls of >= 8 user tiles over >= 32,768 items (its own tests below)."""
    import idgrec_amd.ops as ops

    ops.topk_option("form", {"alternating": 0, "producer-consumer": 1}[request.param])  # (reset after the test: conftest)
    return request.param


@pytest.mark.parametrize("k", [1, 10, 20, 40, 64])
def test_topk_masked_vs_reference_contract(ops, k, golden_small, topk_form):
    g = golden_small
    users_np = g["test_dict_users"][:48]
    ref = g["d64_lgcn_rating"].copy()
    ip, ix = g["pos_indptr"], g["pos_indices"]
    for b, u in enumerate(users_np):
        ref[b, ix[ip[u]:ip[u + 1]]] = -1  # batch_test.py:62-65
    idx, val = ops.score_topk(dev(g["d64_lgcn_user"]), dev(g["d64_lgcn_item"]), dev(users_np), k, dev(ip), dev(ix),
                              return_values=True)
    idx, val = idx.cpu().numpy(), val.cpu().numpy()
    ok, msg = oracle.topk_is_valid(ref, idx, k, tol=2e-6)
    assert ok, msg
    np.testing.assert_allclose(val, np.take_along_axis(ref, idx, 1), rtol=1e-5, atol=1e-6)
    assert (np.diff(val, axis=1) <= 0).all()  # sorted descending
    # exact (raw score desc, item asc) order against the kernel's own dense raw scores
    R = ops.score_dense(dev(g["d64_lgcn_user"]), dev(g["d64_lgcn_item"]), dev(users_np), apply_sigmoid=False).cpu().numpy()
    for b, u in enumerate(users_np):
        R[b, ix[ip[u]:ip[u + 1]]] = -np.inf
    assert np.array_equal(idx, oracle.topk_reference(R, k))
    # raw-score mode: masked entries compete as the value -1 (they can beat negative scores)
    idx2, val2 = ops.score_topk(dev(g["d64_lgcn_user"]), dev(g["d64_lgcn_item"]), dev(users_np), k, dev(ip), dev(ix),
                                apply_sigmoid=False, return_values=True)
    R2 = ops.score_dense(dev(g["d64_lgcn_user"]), dev(g["d64_lgcn_item"]), dev(users_np), apply_sigmoid=False).cpu().numpy()
    for b, u in enumerate(users_np):
        R2[b, ix[ip[u]:ip[u + 1]]] = -1
    assert np.array_equal(idx2.cpu().numpy(), oracle.topk_reference(R2, k))
    np.testing.assert_array_equal(val2.cpu().numpy(), np.take_along_axis(R2, idx2.cpu().numpy(), 1))


@pytest.mark.parametrize("d", [64, 256, 40])
def test_topk_is_independent_of_item_chunking(ops, d, golden_small, monkeypatch, topk_form):
    """The catalogue is cut into chunks across workgroups and the per-chunk best lists are merged:
    any chunk count gives the same answer (incl. ragged last slabs and d not a multiple of 64)."""
    g = golden_small
    rng = np.random.default_rng(d)
    U, I = int(g["num_users"]), 1500
    Ue = rng.standard_normal((U, d)).astype(np.float32) * 0.4
    Ie = rng.standard_normal((I, d)).astype(np.float32) * 0.4
    users_np = rng.integers(0, U, 150)
    ip = np.zeros(U + 1, dtype=np.int64)
    ip[1:] = np.cumsum(rng.integers(0, 30, U))
    ix = np.concatenate([np.sort(rng.choice(I, int(c), replace=False)) for c in np.diff(ip)]).astype(np.int32)
    args = (dev(Ue), dev(Ie), dev(users_np), 20, dev(ip), dev(ix))
    outs = []
    for nc in (1, 2, 3, 12):
        ops.topk_option("chunks", nc)
        outs.append(ops.score_topk(*args, return_values=True))
    for idx, val in outs[1:]:
        assert torch.equal(idx, outs[0][0]) and torch.equal(val, outs[0][1])
    R = oracle.score(Ue, Ie, users_np, apply_sigmoid=True)
    for b, u in enumerate(users_np):
        R[b, ix[ip[u]:ip[u + 1]]] = -1
    ok, msg = oracle.topk_is_valid(R, outs[0][0].cpu().numpy(), 20, tol=3e-6)
    assert ok, msg


@pytest.mark.parametrize("d", [64, 100])
def test_topk_dense_exclusion_runs_and_batch_independence(ops, d, topk_form):
    """More than 64 train items inside one 128-item slab (the cursor's refill loop), rows that mask
    almost the whole catalogue, and: a user's list does not depend on which users share its launch."""
    rng = np.random.default_rng(7 + d)
    U, I, k = 333, 2100, 20
    Ue = rng.standard_normal((U, d)).astype(np.float32) * 0.5
    Ie = rng.standard_normal((I, d)).astype(np.float32) * 0.5
    rows = []
    for u in range(U):
        if u % 5 == 0:
            rows.append(np.sort(rng.choice(np.arange(128 * (u % 7), 128 * (u % 7) + 256), 200, replace=False)))
        elif u % 11 == 0:
            rows.append(np.sort(rng.choice(I, I - 25, replace=False)))
        else:
            rows.append(np.sort(rng.choice(I, int(rng.integers(0, 40)), replace=False)))
    ip = np.zeros(U + 1, dtype=np.int64)
    ip[1:] = np.cumsum([len(r) for r in rows])
    ix = np.concatenate(rows).astype(np.int32)
    users_np = rng.permutation(U)
    args = (dev(Ue), dev(Ie))
    whole = ops.score_topk(*args, dev(users_np), k, dev(ip), dev(ix)).cpu().numpy()
    R = ops.score_dense(*args, dev(users_np), apply_sigmoid=False).cpu().numpy()
    for b, u in enumerate(users_np):
        R[b, ix[ip[u]:ip[u + 1]]] = -np.inf
    assert np.array_equal(whole, oracle.topk_reference(R, k))
    parts = [ops.score_topk(*args, dev(users_np[lo:lo + 50]), k, dev(ip), dev(ix)).cpu().numpy() for lo in range(0, U, 50)]
    assert np.array_equal(np.concatenate(parts), whole)


def test_topk_ties_saturated_sigmoid_and_masked_fill(ops, topk_form):
    # sigmoid saturates to exactly 1.0f: ties broken by lowest item id
    Uu = np.full((3, 64), 1.0, dtype=np.float32)
    Vv = np.full((500, 64), 1.0, dtype=np.float32)
    Vv[100:] *= -1
    idx = ops.score_topk(dev(Uu), dev(Vv), dev(np.array([0, 1, 2])), 20).cpu().numpy()
    assert np.array_equal(idx, np.tile(np.arange(20), (3, 1)))
    # a user whose unmasked items are fewer than k gets masked (-1) items as filler, lowest id first
    ip = np.array([0, 28, 28, 28], dtype=np.int64)
    ix = np.arange(28, dtype=np.int32)
    Vs = np.random.default_rng(0).standard_normal((30, 64)).astype(np.float32)
    idx, val = ops.score_topk(dev(Uu), dev(Vs), dev(np.array([0])), 5, dev(ip), dev(ix), return_values=True)
    assert set(idx.cpu().numpy()[0, :2]) == {28, 29} and idx.cpu().numpy()[0, 2:].tolist() == [0, 1, 2]
    assert val.cpu().numpy()[0, 2:].tolist() == [-1.0, -1.0, -1.0]


@pytest.mark.parametrize("k", [65, 100, 128, 200])
def test_topk_beyond_64_ranks(ops, k, topk_form):
    """torch.topk takes any k <= I (batch_test.py:68; the reference's sparsity_test comment suggests top_K up to
    100): k > 64 runs one scoring pass per 64 ranks, each admitting only keys below the previous pass's last one.
    Same order and values as the one-pass definition: (score descending, item ascending), train positives as -1."""
    rng = np.random.default_rng(k)
    U, I, d = 150, 1900, 64
    Ue = rng.standard_normal((U, d)).astype(np.float32) * 0.5
    Ie = rng.standard_normal((I, d)).astype(np.float32) * 0.5
    Ie[700:760] = Ie[100]  # 61 items with identical scores for every user: ties across a pass boundary
    rows = [np.sort(rng.choice(I, int(rng.integers(0, 60)) if u % 7 else I - k + 30, replace=False)) for u in range(U)]
    ip = np.zeros(U + 1, dtype=np.int64)
    ip[1:] = np.cumsum([len(r) for r in rows])
    ix = np.concatenate(rows).astype(np.int32)
    users_np = rng.permutation(U)
    idx, val = ops.score_topk(dev(Ue), dev(Ie), dev(users_np), k, dev(ip), dev(ix), apply_sigmoid=False, return_values=True)
    R = ops.score_dense(dev(Ue), dev(Ie), dev(users_np), apply_sigmoid=False).cpu().numpy()
    for b, u in enumerate(users_np):
        R[b, ix[ip[u]:ip[u + 1]]] = -1.0
    want = oracle.topk_reference(R, k)
    # masked entries rank as the VALUE -1 among raw scores (unmasked scores below -1 come after them)
    assert np.array_equal(idx.cpu().numpy(), want)
    assert np.array_equal(val.cpu().numpy(), np.take_along_axis(R, want, axis=1))
    # through the evaluator's entry (sigmoid domain): first 64 columns equal the k = 64 call
    idx_s = ops.score_topk(dev(Ue), dev(Ie), dev(users_np), k, dev(ip), dev(ix)).cpu().numpy()
    idx_64 = ops.score_topk(dev(Ue), dev(Ie), dev(users_np), 64, dev(ip), dev(ix)).cpu().numpy()
    assert np.array_equal(idx_s[:, :64], idx_64)
    assert all(len(set(r.tolist())) == k for r in idx_s)


def test_noise_and_perturbation_any_width(ops, golden_small):
    """embedding_size = 48 / 100 (widths without a tiled instantiation): the perturbed product and the stand-alone
    perturbation run in the any-width kernels with the same contract — clean product bit-equal to the fmaf chain, every
    row moved by exactly eps along sign(X), reproducible per (seed, stream)."""
    g = golden_small
    n = int(g["num_users"]) + int(g["num_items"])
    G = _graph(ops, g)
    for d in (48, 100, 7):
        X = torch.randn(n, d, device="cuda") * 0.1
        clean = G.spmm_raw(X)
        assert np.array_equal(clean.cpu().numpy(), oracle.spmm(*_adj(g), X.cpu().numpy(), *G.long_rows()))
        Y1 = ops.spmm_noise_raw(G, X, 0.05, 99, 3)
        Y2 = ops.spmm_noise_raw(G, X, 0.05, 99, 3)
        Y3 = ops.spmm_noise_raw(G, X, 0.05, 99, 4)
        assert torch.equal(Y1, Y2) and not torch.equal(Y1, Y3)
        delta = Y1 - clean
        livez = clean.abs().sum(1) > 0
        if d > 8:  # (with a handful of features some of a row's entries can be exactly zero: sign(0) = 0)
            np.testing.assert_allclose(delta[livez].norm(dim=1).cpu().numpy(), 0.05, rtol=5e-4)
        assert torch.all(delta * torch.sign(clean) >= 0)
        P = ops.perturb_raw(clean, 0.05, 99, 3)
        assert torch.allclose(P, Y1, rtol=0, atol=1e-7)  # same uniforms, same scale: the epilogue's arithmetic on its own
        # the K-layer perturbed pass and its backward (autograd path of SimGCL at this width)
        E0 = X.clone().requires_grad_(True)
        c, v1, v2 = ops.propagate_views(G, E0, 3, False, 0.05, n_views=2)
        assert torch.equal(c, G.propagate_mean_raw(X, 3, False)) and not torch.equal(v1, v2)
        (v1.sum() + v2.sum()).backward()
        assert torch.isfinite(E0.grad).all()


@pytest.mark.parametrize("d", [7, 64, 256])
def test_guest_row_movers(ops, d):
    """idg_rows_gather_f32 / idg_rows_chain_add_f32 (the sharded step's guest rows) against their numpy statement:
    gathered rows bit-equal, zeros where not owned; chained adds in list order (bit-equal to the sequential float32 sums)."""
    rng = np.random.default_rng(d)
    n, B = 500, 300
    src = rng.standard_normal((n, d)).astype(np.float32)
    idx = rng.integers(-1, n, B).astype(np.int64)
    dst = torch.full((B, d), float("nan"), device="cuda")
    ops.rows_gather_raw(dst, dev(src), dev(idx))
    want = np.where((idx >= 0)[:, None], src[np.maximum(idx, 0)], np.float32(0))
    assert np.array_equal(dst.cpu().numpy(), want)
    # chains: occurrences of one destination linked in list order, heads carry the destination
    users = rng.integers(0, 40, B)
    owned = users < 30
    order = np.argsort(users, kind="stable")
    nxt = np.full(B, -1, dtype=np.int64)
    same = users[order][1:] == users[order][:-1]
    nxt[order[:-1][same]] = order[1:][same]
    first = np.ones(B, dtype=bool)
    first[order[1:][same]] = False
    head = np.where(owned & first, users, -1).astype(np.int64)
    g = rng.standard_normal((B, d)).astype(np.float32)
    base = rng.standard_normal((40, d)).astype(np.float32)
    out = dev(base.copy())
    ops.rows_chain_add_raw(out, dev(g), dev(head), dev(nxt))
    want = base.copy()
    for t in np.nonzero(head >= 0)[0]:
        acc, j = want[head[t]].copy(), t
        while j >= 0:
            acc = acc + g[j]
            j = nxt[j]
        want[head[t]] = acc
    assert np.array_equal(out.cpu().numpy(), want)


@pytest.mark.parametrize("d", [32, 64, 256])
@pytest.mark.parametrize("exact", [False, True])
def test_live_unit_list_equals_tile_form(ops, d, exact):
    """idg_graph_live_units: a restricted launch that finds its bitmap registered runs one wave per listed work unit
    (plain vrows, split rows combined in a wave-private LDS slab, chunks through the last-arriver combine) instead of
    visiting every tile — the same bits on the requested rows, nothing else written; plain and perturbed products,
    hub rows of every kind, an EXACT_ORDER handle (rows beyond a tile are plain units)."""
    import idgrec_amd.host as H
    import idgrec_amd.synth as S

    U, I, E = 6000, 2500, 260000  # item hubs far beyond 512 entries (chunked), many rows of 129..512 (LDS-combined)
    users, items = S.generate(U, I, E, seed=5)
    ip, ix, dv = H.build_norm_adj(U, I, users, items)
    n = U + I
    G = ops.Graph(ip, ix, dv, n, n, exact_order=exact)
    deg = np.diff(ip)
    if not exact:
        _, seg, chunk = G.long_rows()
        assert (np.asarray(chunk) > 0).any() and (np.asarray(chunk) == 0).any()
    rng = np.random.default_rng(d)
    rows = np.unique(np.concatenate([rng.integers(0, n, 700), np.argsort(deg)[-40:], np.argsort(deg)[:5], [0, n - 1]]))
    bitmap = np.zeros((n + 31) // 32, dtype=np.uint32)
    np.bitwise_or.at(bitmap, rows >> 5, np.uint32(1) << (rows & 31).astype(np.uint32))
    bm = dev(bitmap.view(np.int32))
    X = torch.randn(n, d, device="cuda") * 0.1
    rows_d = dev(rows)
    others = torch.ones(n, dtype=torch.bool, device="cuda")
    others[rows_d] = False

    def run():
        out = torch.full((n, d), float("nan"), device="cuda")
        G.propagate_mean_raw(X, 3, True, out=out, out_rows=bm)
        noisy = torch.full((n, d), float("nan"), device="cuda")
        ops.spmm_noise_raw(G, X, 0.05, 77, 3, out=noisy, out_rows=bm)
        return out, noisy

    tile_out, tile_noisy = run()
    ws = G.live_units(bm, len(rows))
    torch.cuda.synchronize()
    listed = int(ws[0].item())
    assert listed >= len(rows)  # chunked rows contribute one unit per chunk
    unit_out, unit_noisy = run()
    G.forget_live_units(bm)
    again, _ = run()  # (tile form once more: the registration is gone)
    full = G.propagate_mean_raw(X, 3, True)
    for a, b in ((unit_out, tile_out), (unit_noisy, tile_noisy), (again, tile_out)):
        assert torch.equal(a.index_select(0, rows_d), b.index_select(0, rows_d))
        assert bool(torch.isnan(a[others]).all())
    assert torch.equal(unit_out.index_select(0, rows_d), full.index_select(0, rows_d))


def test_graph_from_the_references_device_tensor(ops, golden_small):
    """the operand the reference hands to torch.sparse.mm is a coalesced fp32 COO tensor ON the device
    (models/LightGCN.py:31-32).  Graph.from_torch_sparse / idg_graph_create_from_device take exactly that; the handle is
    the one the host-CSR constructor builds: same product bits, same propagation bits."""
    g = golden_small
    U, I = int(g["num_users"]), int(g["num_items"])
    n = U + I
    ip, ix, dv = g["adj_indptr"], g["adj_indices"], g["adj_data"]
    rows = np.repeat(np.arange(n), np.diff(ip))
    coo = torch.sparse_coo_tensor(torch.from_numpy(np.stack([rows, ix.astype(np.int64)])), torch.from_numpy(dv), (n, n)).coalesce().cuda()
    G1 = ops.Graph(ip, ix, dv, n, n)
    G2 = ops.Graph.from_torch_sparse(coo)
    assert G1.info() == G2.info()
    X = dev(np.concatenate([g["d64_init_user"], g["d64_init_item"]]))
    assert torch.equal(G1.spmm_raw(X), G2.spmm_raw(X))
    assert torch.equal(G1.propagate_mean_raw(X, 3, True), G2.propagate_mean_raw(X, 3, True))
    assert torch.allclose(G2.spmm_raw(X), torch.sparse.mm(coo, X), rtol=1e-5, atol=1e-7)
    # round 4: only the row pointer visits the host; the entry list is laid out on the device.  The same at yelp2018
    # size (thousands of tiles, split and chunked rows, XCD bands chosen from device-gathered median columns), with
    # exact_order (rows longer than a tile) and a custom split threshold; a bad column id is caught by the device check
    import idgrec_amd.host as H
    import idgrec_amd.synth as S

    Uy, Iy, Ey = S.SHAPES["yelp2018"]
    uy, iy = S.generate(Uy, Iy, Ey, seed=0)
    ipy, ixy, dvy = H.build_norm_adj(Uy, Iy, uy, iy)
    ny = Uy + Iy
    csr = torch.sparse_csr_tensor(dev(ipy), dev(ixy.astype(np.int64)), dev(dvy), size=(ny, ny))
    Xy = torch.randn(ny, 64, device="cuda") * 0.1
    for kw in (dict(), dict(exact_order=True), dict(split_threshold=64)):
        Ga, Gb = ops.Graph(ipy, ixy, dvy, ny, ny, **kw), ops.Graph.from_torch_sparse(csr, **kw)
        assert Ga.info() == Gb.info()
        assert all(np.array_equal(a, b) for a, b in zip(Ga.long_rows(), Gb.long_rows()))
        assert torch.equal(Ga.spmm_raw(Xy), Gb.spmm_raw(Xy)) and torch.equal(Ga.propagate_mean_raw(Xy, 3, True), Gb.propagate_mean_raw(Xy, 3, True))
    bad = torch.sparse_csr_tensor(dev(ipy), dev(ixy.astype(np.int64)), dev(dvy), size=(ny, ny - 5), check_invariants=False)
    with pytest.raises(RuntimeError, match="column id outside"):
        ops.Graph.from_torch_sparse(bad, symmetric=False)


@pytest.mark.parametrize("d,sig", [(64, True), (64, False), (128, True)])
def test_topk_calls_of_1024_users_start_from_a_floor(ops, d, sig):
    """Round 3: a call with few user tiles (the reference evaluates 1024 users per call, batch_test.py:52-68) cuts the
    catalogue into ~30 short chunks; every chunk now starts from a per-user floor — the k-th largest of the user's chunk
    maxima, found by a first launch over the chunks' first slabs — instead of from an empty list.  The lists must be the
    ones the plain form gives (ops.topk_option('floor', 0)) and the ones ONE call over all users gives, ids and values, for k = 1, 20,
    30, 31 (the chunk count follows k) and 64, with train items masked."""
    import os

    import idgrec_amd.synth as S

    U, I = 4096, 38048
    users, items = S.generate(U, I, 160000, seed=11)
    ptr = np.zeros(U + 1, dtype=np.int64)
    ptr[1:] = np.cumsum(np.bincount(users, minlength=U))
    ip, ix = dev(ptr), dev(items.astype(np.int32))
    g = torch.Generator(device="cuda").manual_seed(d)
    Ue = torch.randn(U, d, device="cuda", generator=g) * 0.3
    Ie = torch.randn(I, d, device="cuda", generator=g) * 0.3
    Ie[::97] = Ie[5]  # duplicate item rows: ties, also across chunk boundaries
    calls = [torch.arange(s0, s0 + 1024, device="cuda") for s0 in range(0, U, 1024)]
    every = torch.arange(U, device="cuda")
    for k in (1, 20, 30, 31, 64):
        ops.topk_option("floor", 1)
        got = [ops.score_topk(Ue, Ie, b, k, ip, ix, apply_sigmoid=sig, return_values=True) for b in calls]
        ops.topk_option("floor", 0)
        plain = [ops.score_topk(Ue, Ie, b, k, ip, ix, apply_sigmoid=sig, return_values=True) for b in calls]
        whole = ops.score_topk(Ue, Ie, every, k, ip, ix, apply_sigmoid=sig, return_values=True)
        gi, gv = torch.cat([x[0] for x in got]), torch.cat([x[1] for x in got])
        pi, pv = torch.cat([x[0] for x in plain]), torch.cat([x[1] for x in plain])
        assert torch.equal(gi, pi) and torch.equal(gv, pv), "k=%d: floor form differs from the plain form" % k
        assert torch.equal(gi, whole[0]) and torch.equal(gv, whole[1]), "k=%d: calls of 1024 differ from one call" % k


@pytest.mark.parametrize("sig", [True, False])
@pytest.mark.parametrize("k", [20, 22, 1, 40])
def test_topk_threshold_collect_form_is_the_exact_answer(ops, k, sig, monkeypatch):
    """Round 5, form 3 (the default for calls of >= 8 user tiles over >= 32,768 items at d = 64 / 128, k <= 42;
    ops.topk_option('collect', 0) turns it off): a floor per user from the maxima of a strided sample of the catalogue scored as bf16
    LOWER bounds, ONE pass that appends every item whose bf16 UPPER bound reaches the floor to the user's candidate list —
    no list insertions in the scoring pass — and an exact finish: the candidates' fp32 scores as the fmaf chain the fp32
    matrix cores evaluate, masked, through the streaming select.  At yelp2018 size, all users in one call, train items
    masked, duplicated item rows (exact ties), a user whose every score ties (its candidate list overflows: redone over
    the whole catalogue): ids AND values bit for bit what the exact producer / consumer form returns."""
    import idgrec_amd.synth as S

    U, I, d = 31668, 38048, 64
    users, items = S.generate(U, I, 600000, seed=23)
    ptr = np.zeros(U + 1, dtype=np.int64)
    ptr[1:] = np.cumsum(np.bincount(users, minlength=U))
    ip, ix = dev(ptr), dev(items.astype(np.int32))
    g = torch.Generator(device="cuda").manual_seed(k)
    Ue = torch.randn(U, d, device="cuda", generator=g) * 0.3
    Ie = torch.randn(I, d, device="cuda", generator=g) * 0.3
    Ie[::101] = Ie[3]
    Ue[5] = 0.0
    every = torch.arange(U, device="cuda")
    info = {}
    ops.topk_option("collect", 1)
    got = ops.score_topk(Ue, Ie, every, k, ip, ix, apply_sigmoid=sig, return_values=True, info=info)
    assert info["form"] == 3 and info["chunks"] == 1, info
    # user 5 (every score ties: its candidate list overflows) certainly; the 377 identical item rows overflow a few more
    assert 1 <= info["users_redone"] <= U // 200, info
    ops.topk_option("collect", 0)
    want = ops.score_topk(Ue, Ie, every, k, ip, ix, apply_sigmoid=sig, return_values=True, info=info)
    assert info["form"] == 1
    assert torch.equal(got[0], want[0]), "ids differ from the exact form"
    assert torch.equal(got[1], want[1]), "values differ from the exact form"
    if k == 20:
        # every score far below -1: in raw-score mode the masked train items, ranking as the VALUE -1 (batch_test.py:65), are
        # then the BEST items of a user and must head its list (the collect pass masks in this mode for exactly that); in
        # sigmoid mode they rank below everything and never appear
        Un, In = -Ue.abs() - 0.1, Ie.abs() + 0.1
        ops.topk_option("collect", 1)
        got = ops.score_topk(Un, In, every, k, ip, ix, apply_sigmoid=sig, return_values=True, info=info)
        assert info["form"] == 3, info
        ops.topk_option("collect", 0)
        want = ops.score_topk(Un, In, every, k, ip, ix, apply_sigmoid=sig, return_values=True)
        assert torch.equal(got[0], want[0]) and torch.equal(got[1], want[1]), "all-negative scores: differs from the exact form"
        deg = torch.from_numpy(np.diff(ptr)).cuda()
        head_is_train = (got[1][:, 0] == -1.0)
        assert bool((head_is_train == ((deg > 0) & (not sig))).all())


@pytest.mark.parametrize("d,per_call", [(128, 20000), (128, 1024), (64, 3000), (64, 512), (64, 20000), (256, 20000), (256, 2048)])
def test_topk_collect_form_with_uneven_norms_and_at_d128(ops, d, per_call, monkeypatch):
    """Form 3's bounds are per pair — UB - LB = 2 cu(u) vb(v) — so a catalogue whose row norms spread over two orders of
    magnitude (trained tables: popular items grow long) costs candidates only where the long rows are, and the finish's
    uniform cut (cu(u) max vb) stays correct whatever the spread.  Item rows scaled by 0.03 .. 3, a few users scaled up
    100x and down 1e-3x, a zero user, a zero item block, d = 128 (nine k-steps), 256 (seventeen; user operands from LDS) and 64,
    one call and chunked calls: ids and
    values bit for bit those of the exact producer / consumer form."""
    import idgrec_amd.synth as S

    U, I, k = 20000, 40000, 20
    users, items = S.generate(U, I, 400000, seed=d + per_call)
    ptr = np.zeros(U + 1, dtype=np.int64)
    ptr[1:] = np.cumsum(np.bincount(users, minlength=U))
    ip, ix = dev(ptr), dev(items.astype(np.int32))
    g = torch.Generator(device="cuda").manual_seed(d + per_call)
    Ue = torch.randn(U, d, device="cuda", generator=g) * 0.3
    Ie = torch.randn(I, d, device="cuda", generator=g) * 0.3
    Ie *= torch.exp(torch.rand(I, 1, device="cuda", generator=g) * 4.6 - 3.5)  # row scale 0.03 .. 3
    Ue[100:110] *= 100.0
    Ue[200:210] *= 1e-3
    Ue[7] = 0.0
    Ie[5000:5200] = 0.0
    every = torch.arange(U, device="cuda")
    info = {}

    def run():
        out = [ops.score_topk(Ue, Ie, every[s0:s0 + per_call], k, ip, ix, return_values=True, info=info if s0 == 0 else None)
               for s0 in range(0, U, per_call)]
        return torch.cat([x[0] for x in out]), torch.cat([x[1] for x in out])

    ops.topk_option("collect", 1)
    got = run()
    assert info["form"] == 3, info
    ops.topk_option("collect", 0)
    want = run()
    assert info["form"] in (0, 1), info
    assert torch.equal(got[0], want[0]), "ids differ from the exact form"
    assert torch.equal(got[1], want[1]), "values differ from the exact form"


@pytest.mark.parametrize("d", [64, 256])
def test_topk_collect_form_on_adversarial_roundings(ops, d, monkeypatch):
    """The bounds of form 3 must hold for the WORST rounding, not the typical one (random tables stay two orders of magnitude
    under it: a bound constant half its size passed every other test of this file for most of round 5).  Here every mantissa
    sits just beside a bf16 rounding midpoint: on the first half of the coordinates the users and the ten best items A round DOWN
    (their bf16 score is 0.78 % under the exact one), on the second half the users and a family of items C — one per 128 items,
    exact scores 0.2 % (most) to 2.9 % under A's — round UP (0.78 % over).  With bounds too narrow by a factor of two the floor made of
    C's lower bounds lies above A's upper bounds and the ten best items of every user are lost; with bound_c(d) the lists are
    the exact form's, ids and values."""
    U, I, k, half = 8192, 33024, 20, d // 2
    w_dn = np.float32(1.0 + 2.0 ** -8 * (1.0 - 2.0 ** -12))  # rounds down to 1.0
    w_up = np.float32(1.0 + 2.0 ** -8 * (1.0 + 2.0 ** -12))  # rounds up to 1 + 2^-7
    rng = np.random.default_rng(7)
    Ue = np.empty((U, d), dtype=np.float32)
    Ue[:, :half], Ue[:, half:] = w_dn, w_up
    Ue *= (2.0 ** rng.integers(-3, 4, (U, 1))).astype(np.float32)  # (powers of two keep the mantissas)
    Ie = (rng.standard_normal((I, d)) * 0.01).astype(np.float32)  # background: far below
    a_items = rng.choice(np.arange(0, I, 128) + 5, 10, replace=False)
    Ie[a_items] = 0.0
    Ie[a_items, :half] = w_dn  # exact score (d / 2) w_dn^2 per unit of the user's scale
    c_items = np.setdiff1d(np.arange(0, I, 128) + 77, a_items)
    for j, it in enumerate(c_items):
        # the d / 2 coordinates sum to d / 2 - m / 16 (m = 1 .. 15) in powers of two only, so that every mantissa stays w_up's:
        # ones, one 4, and the binary digits of (16 - m) / 16
        # (m = 1 for nine items in ten — 0.2 % under A, the level whose lower bounds a too narrow bound lifts over A's upper
        #  bounds — in a random order of the coordinates: same real score, fp32 chains that differ in the last places)
        m = 1 if j % 10 else 1 + j % 15
        row = np.ones(half)
        row[half - 5] = 4.0
        row[half - 4:] = [((16 - m) >> b) & 1 for b in (3, 2, 1, 0)]
        row[half - 4:] *= [0.5, 0.25, 0.125, 0.0625]
        Ie[it] = 0.0
        Ie[it, half:] = (rng.permutation(row) * w_up).astype(np.float32)
    Ud, Id = dev(Ue), dev(Ie)
    every = torch.arange(U, device="cuda")
    info = {}
    ops.topk_option("collect", 1)
    got = ops.score_topk(Ud, Id, every, k, apply_sigmoid=True, return_values=True, info=info)
    assert info["form"] == 3 and info["users_redone"] == 0, info
    ops.topk_option("collect", 0)
    want = ops.score_topk(Ud, Id, every, k, apply_sigmoid=True, return_values=True)
    assert torch.equal(got[0], want[0]) and torch.equal(got[1], want[1])
    # the construction does what it says: the ten A items head every list, C items fill it
    assert set(want[0][0, :10].tolist()) == set(int(x) for x in a_items)
    assert set(want[0][0, 10:].tolist()) <= set(int(x) for x in c_items)


def test_topk_call_cut_by_the_workspace_budget(ops, monkeypatch):
    """ops.score_topk hands over every user in ONE library call unless that call's workspace (form 3: up to 1024 candidate
    keys per user) would exceed IDG_TOPK_WS_BYTES (default 8 GiB — half a million users); then it issues calls of a multiple
    of 16,384 users.  Same lists and values, by construction and here: 40,000 users under a 280 MiB budget (three calls)."""
    U, I, d, k = 40000, 33000, 64, 20
    g = torch.Generator(device="cuda").manual_seed(3)
    Ue = torch.randn(U, d, device="cuda", generator=g) * 0.3
    Ie = torch.randn(I, d, device="cuda", generator=g) * 0.3
    every = torch.arange(U, device="cuda")
    info = {}
    monkeypatch.delenv("IDG_TOPK_WS_BYTES", raising=False)
    whole = ops.score_topk(Ue, Ie, every, k, return_values=True, info=info)
    assert info["form"] == 3 and info["calls"] == 1, info
    monkeypatch.setenv("IDG_TOPK_WS_BYTES", str(280 << 20))
    cut = ops.score_topk(Ue, Ie, every, k, return_values=True, info=info)
    assert info["calls"] == 3 and info["users_redone"] == 0, info
    assert torch.equal(cut[0], whole[0]) and torch.equal(cut[1], whole[1])


@pytest.mark.parametrize("per_call", [2048, 4096, 8192])
def test_topk_calls_of_a_few_thousand_users(ops, per_call):
    """calls of 2048 / 4096 users (32 / 64 user tiles) still take the k + 2 chunk geometry with its starting
    floor, calls of 8192 (128 tiles) keep the tuned chunk count — whichever geometry a call gets, its lists are the
    ones ONE call over all users gives, ids and values; k = 20 and 64, train items masked, ties across chunk cuts."""
    import idgrec_amd.synth as S

    U, I, d = 16384, 38048, 64
    users, items = S.generate(U, I, 400000, seed=13)
    ptr = np.zeros(U + 1, dtype=np.int64)
    ptr[1:] = np.cumsum(np.bincount(users, minlength=U))
    ip, ix = dev(ptr), dev(items.astype(np.int32))
    g = torch.Generator(device="cuda").manual_seed(per_call)
    Ue = torch.randn(U, d, device="cuda", generator=g) * 0.3
    Ie = torch.randn(I, d, device="cuda", generator=g) * 0.3
    Ie[::89] = Ie[7]
    every = torch.arange(U, device="cuda")
    for k in (20, 64):
        whole = ops.score_topk(Ue, Ie, every, k, ip, ix, return_values=True)
        got = [ops.score_topk(Ue, Ie, every[s0:s0 + per_call], k, ip, ix, return_values=True) for s0 in range(0, U, per_call)]
        assert torch.equal(torch.cat([x[0] for x in got]), whole[0]) and torch.equal(torch.cat([x[1] for x in got]), whole[1])


@pytest.mark.parametrize("per_call,d", [(1024, 64), (100, 64), (5000, 64), (1024, 128)])
def test_topk_exact_order_on_integer_embeddings(ops, per_call, d):
    """VERDICT r03 (weak #11): the exact-order checks above compare the fused kernels with the library's OWN dense scores.
    Here nothing of the library is the reference: embeddings are small integers, so every dot product is an integer below
    2^24 — exact in fp32 in ANY summation order, MFMA or sequential — and the raw-score top-K must equal, id for id and
    value for value, the one NumPy forms in float64 under the published order (score descending, item id ascending;
    masked train items rank as -1).  Thousands of ties per user, across slab and chunk boundaries.  With the sigmoid the
    selection is still made on raw scores: same ids, values = sigmoid of the exact scores to 1 ulp-ish."""
    import idgrec_amd.synth as S

    U, I, k = 5000, 20011, 20
    users, items = S.generate(U, I, 90000, seed=d + per_call)
    ptr = np.zeros(U + 1, dtype=np.int64)
    ptr[1:] = np.cumsum(np.bincount(users, minlength=U))
    order = np.lexsort((items, users))
    items_sorted = items[order].astype(np.int32)
    rng = np.random.default_rng(per_call)
    Ue = rng.integers(-3, 4, (U, d)).astype(np.float32)
    Ie = rng.integers(-3, 4, (I, d)).astype(np.float32)
    ip, ix = dev(ptr), dev(items_sorted)
    Ud, Id = dev(Ue), dev(Ie)
    got_i, got_v, got_si = [], [], []
    for s0 in range(0, U, per_call):
        b = torch.arange(s0, min(s0 + per_call, U), device="cuda")
        i_, v_ = ops.score_topk(Ud, Id, b, k, ip, ix, apply_sigmoid=False, return_values=True)
        got_i.append(i_.cpu().numpy())
        got_v.append(v_.cpu().numpy())
        got_si.append(ops.score_topk(Ud, Id, b, k, ip, ix, apply_sigmoid=True, return_values=True))
    got_i, got_v = np.concatenate(got_i), np.concatenate(got_v)
    sig_i = np.concatenate([x[0].cpu().numpy() for x in got_si])
    sig_v = np.concatenate([x[1].cpu().numpy() for x in got_si])
    for c0 in range(0, U, 500):  # float64 reference, 500 users at a time
        R = Ue[c0:c0 + 500].astype(np.float64) @ Ie.astype(np.float64).T
        M = R.copy()  # sigmoid mode: a masked item ranks below every score; raw mode: as the value -1
        for r, u in enumerate(range(c0, min(c0 + 500, U))):
            R[r, items_sorted[ptr[u]:ptr[u + 1]]] = -1.0
            M[r, items_sorted[ptr[u]:ptr[u + 1]]] = -np.inf
        want = np.lexsort((np.broadcast_to(np.arange(I), R.shape), -R), axis=1)[:, :k]
        assert np.array_equal(got_i[c0:c0 + 500], want), "raw mode: ids differ from the float64 reference"
        assert np.array_equal(got_v[c0:c0 + 500].astype(np.float64), np.take_along_axis(R, want, 1))
        want_s = np.lexsort((np.broadcast_to(np.arange(I), M.shape), -M), axis=1)[:, :k]
        assert np.array_equal(sig_i[c0:c0 + 500], want_s), "sigmoid mode: ids differ from the float64 reference"
        np.testing.assert_allclose(sig_v[c0:c0 + 500], 1.0 / (1.0 + np.exp(-np.take_along_axis(M, want_s, 1))), rtol=3e-7, atol=0)


@pytest.mark.parametrize("per_call", [100, 1024, 5000, 20000])
def test_topk_raw_scores_below_minus_one_put_the_masked_items_first(ops, per_call):
    """Raw-score mode (apply_sigmoid = 0) ranks a masked train item as the VALUE -1 (batch_test.py:65 writes -1 into the
    rating matrix; after the sigmoid that is below every score, on raw scores it is not).  When every score of a user is
    below -1 its train items are its best items: they head the list, in item order, whatever their raw scores are.  The
    producer / consumer kernels flag (user, slab) pairs on raw accumulators, before masking — round 5 found them dropping
    these entries; a consumer now publishes no floor until it has passed -1.  Integer embeddings (users <= -1, items >= 1:
    every dot product an exact integer <= -64), float64 NumPy reference, call sizes that take the one-kernel, the
    starting-floor and the many-tile geometries (20000 users: form 3 when it applies)."""
    import idgrec_amd.synth as S

    U, I, d, k = 20000, 33001, 64, 20
    users, items = S.generate(U, I, 300000, seed=per_call)
    ptr = np.zeros(U + 1, dtype=np.int64)
    ptr[1:] = np.cumsum(np.bincount(users, minlength=U))
    order = np.lexsort((items, users))
    items_sorted = items[order].astype(np.int32)
    rng = np.random.default_rng(per_call)
    Ue = -rng.integers(1, 4, (U, d)).astype(np.float32)
    Ie = rng.integers(1, 4, (I, d)).astype(np.float32)
    ip, ix = dev(ptr), dev(items_sorted)
    Ud, Id = dev(Ue), dev(Ie)
    got_i, got_v = [], []
    for s0 in range(0, U, per_call):
        b = torch.arange(s0, min(s0 + per_call, U), device="cuda")
        i_, v_ = ops.score_topk(Ud, Id, b, k, ip, ix, apply_sigmoid=False, return_values=True)
        got_i.append(i_.cpu().numpy())
        got_v.append(v_.cpu().numpy())
    got_i, got_v = np.concatenate(got_i), np.concatenate(got_v)
    deg = np.diff(ptr)
    assert (deg >= k).any() and ((deg > 0) & (deg < k)).any()
    for u in np.nonzero(deg > 0)[0]:  # every user: its train items first, ascending, as -1
        n = min(int(deg[u]), k)
        assert np.array_equal(got_i[u, :n], items_sorted[ptr[u]:ptr[u] + n]), "user %d: train items do not head the list" % u
        assert (got_v[u, :n] == -1.0).all()
    for c0 in range(0, U, 4000):  # and the whole list against float64, 500 users out of every 4000
        R = Ue[c0:c0 + 500].astype(np.float64) @ Ie.astype(np.float64).T
        assert R.max() <= -64.0
        for r, u in enumerate(range(c0, min(c0 + 500, U))):
            R[r, items_sorted[ptr[u]:ptr[u + 1]]] = -1.0
        want = np.lexsort((np.broadcast_to(np.arange(I), R.shape), -R), axis=1)[:, :k]
        assert np.array_equal(got_i[c0:c0 + 500], want), "ids differ from the float64 reference"
        assert np.array_equal(got_v[c0:c0 + 500].astype(np.float64), np.take_along_axis(R, want, 1))


@pytest.mark.parametrize("mode", ["unique", "raw", "cross"])
def test_infonce_id_lists_planned_ahead_equal_the_in_call_stage(ops, mode):
    """idg_infonce_plan (the id-list stage of an InfoNCE call, run by the engines on the side stream one batch ahead) +
    the call with IDG_SSL_PLANNED / planned = 1 on that workspace: loss and gradient rows bit-identical to the call that
    builds the lists itself — for the unique-id, raw-list and cross forms, with ids that repeat up to ~40 times, and again
    after the workspace has been used for another batch (stale lists must not survive a re-plan)."""
    U, I, d, B = 3000, 5000, 64, 1024
    n = U + I
    g = torch.Generator(device="cuda").manual_seed(5)
    v1, v2 = torch.randn(n, d, device="cuda", generator=g), torch.randn(n, d, device="cuda", generator=g)
    batches = []
    for _ in range(2):
        users = torch.randint(0, U, (B,), device="cuda", generator=g)
        items = (torch.rand(B, device="cuda", generator=g) ** 3 * I).long()
        batches.append((users, items))
    ws = ops.infonce_workspace(n, B, d, "cuda")
    side = torch.cuda.Stream()
    for users, items in batches:
        outs = []
        for planned in (False, True):
            g1, g2 = torch.zeros(n, d, device="cuda"), torch.zeros(n, d, device="cuda")
            loss = torch.zeros(2, device="cuda")
            kw = {}
            if planned:
                side.wait_stream(torch.cuda.current_stream())
                ops.infonce_plan_raw(users, items, U, n, d, {"unique": ops.SSL_UNIQUE, "raw": ops.SSL_RAW, "cross": ops.SSL_CROSS}[mode],
                                     ws, stream=side.cuda_stream)
                torch.cuda.current_stream().wait_stream(side)
                kw = {"ws": ws, "planned": True}
            if mode == "cross":
                ops.infonce_cross_raw(v1, users, items, U, 0.2, g=g1, loss=loss, grad_scale=0.5, **kw)
            else:
                ops.infonce_pair_raw(v1, v2, users, items, U, 0.2, g1=g1, g2=g2, loss=loss, dedup=mode == "unique", grad_scale=0.5,
                                     accumulate=True, **kw)
            outs.append((loss.clone(), g1, g2))
        for a, b in zip(outs[0], outs[1]):
            assert torch.equal(a, b)
        assert torch.isfinite(outs[0][0]).all() and float(outs[0][1].abs().sum()) > 0


def _row_bitmap(n, rows):
    bitmap = np.zeros((n + 31) // 32 + 1, dtype=np.uint32)
    np.bitwise_or.at(bitmap, rows >> 5, np.uint32(1) << (rows & 31).astype(np.uint32))
    return dev(bitmap.view(np.int32))


def test_live_unit_lists_follow_their_bitmap(ops):
    """a unit list must never outlive the bitmap contents it was built from.  (1) Rewriting a registered bitmap
    through the library (idg_bpr_touch_rows / idg_bitmap_clear) drops the list: the next restricted launch visits the
    tiles and produces the NEW rows.  (2) A masked copy of the handle finds its base's list (shared schedule) and a copy
    made BEFORE a list existed does too.  (3) More set bits than max_rows: the list is marked incomplete, the launch
    poisons what it writes and idg_graph_live_units_check raises."""
    import idgrec_amd.host as H
    import idgrec_amd.synth as S

    U, I, E = 3000, 2000, 90000
    users, items = S.generate(U, I, E, seed=9)
    ip, ix, dv = H.build_norm_adj(U, I, users, items)
    n, d = U + I, 64
    G = ops.Graph(ip, ix, dv, n, n)
    sub = G.dropout_copy(0.5, stream=(5, 1))  # a copy made before any list exists
    X = torch.randn(n, d, device="cuda") * 0.1
    full, full_sub = G.spmm_raw(X), sub.spmm_raw(X)
    rng = np.random.default_rng(0)
    rows_a = np.unique(rng.integers(0, n, 300))
    rows_b = np.unique(rng.integers(0, n, 300))
    bm = _row_bitmap(n, rows_a)

    def restricted(graph):
        out = torch.full((n, d), float("nan"), device="cuda")
        ops.spmm_ex_raw(graph, X, Y=out, out_rows=bm)
        return out

    ws = G.live_units(bm, len(rows_a))
    ra, rb = dev(rows_a), dev(rows_b)
    out = restricted(G)
    assert torch.equal(out[ra], full[ra]) and int(torch.isfinite(out).all(dim=1).sum()) == len(rows_a)
    out = restricted(sub)  # (2) the copy runs on the base's schedule: same list, its own values
    assert torch.equal(out[ra], full_sub[ra]) and int(torch.isfinite(out).all(dim=1).sum()) == len(rows_a)
    # (1) the bitmap is rewritten through the library: rows_b now
    ids = dev(rows_b)
    ops.bpr_touch_rows_raw(ids, ids, ids, 0, bm, clear_bits=n)
    out = restricted(G)
    assert torch.equal(out[rb], full[rb]) and int(torch.isfinite(out).all(dim=1).sum()) == len(rows_b)
    # ... and listed again
    G.live_units(bm, len(rows_b), ws=ws)
    out = restricted(G)
    assert torch.equal(out[rb], full[rb]) and int(torch.isfinite(out).all(dim=1).sum()) == len(rows_b)
    ops.Graph.live_units_check(ws)
    # (3) a bound that is too small
    small = G.live_units(bm, 10)
    with pytest.raises(RuntimeError, match="more rows than max_rows"):
        ops.Graph.live_units_check(small)
    out = restricted(G)
    produced = ~torch.isnan(out).all(dim=1)
    assert int(produced.sum()) == 0  # whatever the incomplete list covers is poisoned, nothing passes for a result
    G.forget_live_units(bm)
    out = restricted(G)
    assert torch.equal(out[rb], full[rb])


@pytest.mark.parametrize("d", [64, 256])
def test_product_between_two_row_sets(ops, d):
    """out_rows and x_rows in ONE launch (round 3: the first backward products of the sharded step run between the
    batch's rows and the rows one hop away), tile form and live-unit form: the produced rows equal the dense product of
    the masked panel bit for bit, dead rows of X are never read (NaN poison), nothing else is written."""
    import idgrec_amd.host as H
    import idgrec_amd.synth as S

    U, I, E = 6000, 2500, 260000
    users, items = S.generate(U, I, E, seed=5)
    ip, ix, dv = H.build_norm_adj(U, I, users, items)
    n = U + I
    G = ops.Graph(ip, ix, dv, n, n)
    rng = np.random.default_rng(d)
    deg = np.diff(ip)
    live = np.unique(np.concatenate([rng.integers(0, n, 900), np.argsort(deg)[-20:]]))
    want_rows = np.unique(np.concatenate([rng.integers(0, n, 700), np.argsort(deg)[-40:], [0, n - 1]]))
    xb, ob = _row_bitmap(n, live), _row_bitmap(n, want_rows)
    X = torch.randn(n, d, device="cuda") * 0.1
    Xz = torch.zeros_like(X)
    Xz[dev(live)] = X[dev(live)]
    Xp = torch.full_like(X, float("nan"))
    Xp[dev(live)] = X[dev(live)]
    addend = torch.randn(n, d, device="cuda")
    ref = G.spmm_raw(Xz, addend=addend)
    wd = dev(want_rows)
    others = torch.ones(n, dtype=torch.bool, device="cuda")
    others[wd] = False
    for listed in (False, True):
        if listed:
            G.live_units(ob, len(want_rows))
        out = torch.full((n, d), float("nan"), device="cuda")
        ops.spmm_epi_raw(G, Xp, Y=out, addend=addend, out_rows=ob, x_rows=xb)
        assert torch.equal(out[wd], ref[wd]) and bool(torch.isnan(out[others]).all())
    G.forget_live_units(ob)


def test_epilogue_struct_forms(ops, golden_small):
    """idg_spmm_epi_f32: three earlier terms summed left to right before the product (the layer mean formed by the last
    layer, as idg_propagate_mean_f32 does internally) equals propagate_mean bit for bit; a row mask on addend / sum_in /
    accumulate reads no dead row; the Adam group equals idg_adam_step_f32 on the stored gradient."""
    g = golden_small
    U, I = int(g["num_users"]), int(g["num_items"])
    n, d, K = U + I, 64, 3
    G = ops.Graph(g["adj_indptr"], g["adj_indices"], g["adj_data"], n, n)
    E0 = dev(np.concatenate([g["d64_init_user"], g["d64_init_item"]]))
    X1, X2 = G.spmm_raw(E0), None
    X2 = G.spmm_raw(X1)
    out = torch.empty_like(E0)
    ops.spmm_epi_raw(G, X2, sum_in=E0, sum_in2=X1, sum_in3=X2, sum_out=out, div=4.0)
    assert torch.equal(out, G.propagate_mean_raw(E0, K, True))
    # masked addend / accumulate: dead rows hold NaN and are not read
    rows = np.unique(np.random.default_rng(1).integers(0, n, 200))
    bm, rd = _row_bitmap(n, rows), dev(rows)
    add = torch.full((n, d), float("nan"), device="cuda")
    add[rd] = torch.randn(len(rows), d, device="cuda")
    acc = torch.full((n, d), float("nan"), device="cuda")
    acc[rd] = torch.randn(len(rows), d, device="cuda")
    want = G.spmm_raw(E0).cpu().numpy()
    want[rows] = want[rows] + add[rd].cpu().numpy()
    want_acc = want / np.float32(3.0)  # (a true division, as the kernel's: torch's device op multiplies by 1/3)
    want_acc[rows] = acc[rd].cpu().numpy() + want_acc[rows]
    Y = torch.empty_like(E0)
    ops.spmm_epi_raw(G, E0, Y=Y, addend=add, sum_out=acc, div=3.0, accumulate=True, mask=bm)
    assert np.array_equal(Y.cpu().numpy(), want) and np.array_equal(acc.cpu().numpy(), want_acc)
    # Adam in the epilogue == the separate kernel
    p1, m1, v1 = torch.randn(n, d, device="cuda"), torch.rand(n, d, device="cuda") * 0.1, torch.rand(n, d, device="cuda") * 0.01
    p2, m2, v2 = p1.clone(), m1.clone(), v1.clone()
    grad = torch.empty_like(E0)
    ops.spmm_epi_raw(G, E0, sum_out=grad, div=4.0, adam=(p1, m1, v1, 1e-3, 7))
    ops.adam_step_raw(p2, grad, m2, v2, 1e-3, 7)
    assert torch.equal(p1, p2) and torch.equal(m1, m2) and torch.equal(v1, v2)


@pytest.mark.parametrize("d", [64, 256])
def test_sharded_row_movers_and_item_tail(ops, d):
    """The round-3 movers of the sharded step against their numpy statements, and idg_grad_tail_adam_f32 against the
    operations of the single-device step's last epilogue (same order: (g + t) / cnt, + G, Adam)."""
    rng = np.random.default_rng(d)
    n, B = 500, 64
    src0, src1 = rng.standard_normal((n, d)).astype(np.float32), rng.standard_normal((n, d)).astype(np.float32)
    idx = rng.integers(-1, n, B).astype(np.int64)
    d0, d1 = torch.full((B, d), float("nan"), device="cuda"), torch.full((B, d), float("nan"), device="cuda")
    ops.rows_gather2_raw(d0, dev(src0), d1, dev(src1), dev(idx))
    for got, src in ((d0, src0), (d1, src1)):
        assert np.array_equal(got.cpu().numpy(), np.where((idx >= 0)[:, None], src[np.maximum(idx, 0)], np.float32(0)))
    ids = np.sort(rng.choice(n, 40, replace=False)).astype(np.int64)
    compact = rng.standard_normal((40, d)).astype(np.float32)
    panel = torch.full((n, d), float("nan"), device="cuda")
    ops.rows_scatter_raw(panel, dev(ids), dev(compact))
    assert np.array_equal(panel.cpu().numpy()[ids], compact) and int(torch.isnan(panel).all(dim=1).sum()) == n - 40
    # chains: heads 3 -> 9 -> 20, 5 (alone), 7 -> 8
    nxt = np.full(B, -1, dtype=np.int64)
    nxt[3], nxt[9], nxt[7] = 9, 20, 8
    head = np.full(B, -1, dtype=np.int64)
    head[3], head[5], head[7] = 11, 2, 30
    s0, s1 = rng.standard_normal((B, d)).astype(np.float32), rng.standard_normal((B, d)).astype(np.float32)
    t0, t1 = torch.full((n, d), float("nan"), device="cuda"), torch.full((n, d), float("nan"), device="cuda")
    ops.rows_chain_store2_raw(t0, dev(s0), t1, dev(s1), dev(head), dev(nxt))
    for got, s in ((t0, s0), (t1, s1)):
        got = got.cpu().numpy()
        assert np.array_equal(got[11], (s[3] + s[9]) + s[20]) and np.array_equal(got[2], s[5]) and np.array_equal(got[30], s[7] + s[8])
        assert np.isnan(got).all(axis=1).sum() == n - 3
    # layer mean at rows
    a, b, c = (rng.standard_normal((n, d)).astype(np.float32) for _ in range(3))
    out = torch.full((n, d), float("nan"), device="cuda")
    ops.rows_layer_mean_raw(out, dev(ids), [dev(a), dev(b), dev(c)], dev(compact), 4.0)
    assert np.array_equal(out.cpu().numpy()[ids], (((a[ids] + b[ids]) + c[ids]) + compact) / np.float32(4.0))
    ops.rows_layer_mean_raw(out, dev(ids), [dev(b)], dev(compact), 2.0)
    assert np.array_equal(out.cpu().numpy()[ids], (b[ids] + compact) / np.float32(2.0))
    ops.rows_layer_mean_raw(out, dev(ids), [], dev(compact), 1.0)
    assert np.array_equal(out.cpu().numpy()[ids], compact)
    # the item tail on a block of rows [row0, row0 + rows) with live bits taken at GLOBAL row ids
    rows, row0 = 96, 37
    live_rows = np.sort(rng.choice(rows, 20, replace=False))
    bits = _row_bitmap(row0 + rows + 64, row0 + live_rows)
    t = rng.standard_normal((rows, d)).astype(np.float32)
    gg = np.full((rows, d), np.nan, dtype=np.float32)
    GG = np.full((rows, d), np.nan, dtype=np.float32)
    gg[live_rows], GG[live_rows] = rng.standard_normal((20, d)), rng.standard_normal((20, d))
    live = np.zeros(rows, dtype=bool)
    live[live_rows] = True
    with np.errstate(invalid="ignore"):
        s = np.where(live[:, None], gg + t, t) / np.float32(4.0)
        s = np.where(live[:, None], GG + s, s)
    p = torch.randn(rows, d, device="cuda")
    m, v = torch.rand(rows, d, device="cuda") * 0.1, torch.rand(rows, d, device="cuda") * 0.01
    p2, m2, v2 = p.clone(), m.clone(), v.clone()
    Gd = dev(GG)
    ops.grad_tail_adam_raw(dev(t), dev(gg), Gd, bits, row0, True, 4.0, True, p, m, v, 1e-3, 3)
    assert np.array_equal(Gd.cpu().numpy(), s)
    ops.adam_step_raw(p2, dev(s), m2, v2, 1e-3, 3)
    assert torch.equal(p, p2) and torch.equal(m, m2) and torch.equal(v, v2)


@pytest.mark.parametrize("d,K", [(64, 3), (32, 2), (256, 3)])
def test_receptive_field_propagation(ops, d, K):
    """idg_graph_expand_rows + idg_propagate_mean_fields_f32 + idg_propagate_mean_bwd_adam_fields_f32 on a graph much
    larger than a batch's K-hop neighbourhood: hop sets equal to the CSR's, the restricted forward bit-equal to the full
    one on the batch's rows, the restricted backward + Adam bit-equal to the unrestricted call on EVERY row."""
    import idgrec_amd.host as H
    import idgrec_amd.synth as S

    U, I, E = 90000, 60000, 380000
    users, items = S.generate(U, I, E, seed=11)
    ip, ix, dv = H.build_norm_adj(U, I, users, items)
    n = U + I
    G = ops.Graph(ip, ix, dv, n, n)
    rng = np.random.default_rng(d)
    e = rng.integers(0, len(users), 24)
    batch = np.unique(np.concatenate([users[e], U + items[e], U + rng.integers(0, I, 24)]))
    words = (n + 31) // 32

    def to_bitmap(rows):
        b = np.zeros(words, dtype=np.uint32)
        np.bitwise_or.at(b, rows >> 5, np.uint32(1) << (rows & 31).astype(np.uint32))
        return dev(b.view(np.int32))

    def from_bitmap(t):
        bits = np.unpackbits(t.cpu().numpy().view(np.uint8), bitorder="little")[:n]
        return np.nonzero(bits)[0]

    hops, sets = [to_bitmap(batch)], [batch]
    for _ in range(K - 1):
        nxt = torch.zeros(words, dtype=torch.int32, device="cuda")
        G.expand_rows(hops[-1], nxt)
        want = np.unique(np.concatenate([sets[-1]] + [ix[ip[r]:ip[r + 1]] for r in sets[-1]]))
        assert np.array_equal(from_bitmap(nxt), want)
        hops.append(nxt)
        sets.append(want)
    assert len(sets[-1]) < 0.8 * n  # the point of the exercise: the field is not the graph
    E0 = torch.randn(n, d, device="cuda") * 0.1
    full = G.propagate_mean_raw(E0, K, True)
    fin = torch.full((n, d), float("nan"), device="cuda")
    G.propagate_mean_fields_raw(E0, K, True, fin, hops[::-1])  # layer 1 produces the widest set, layer K the batch
    rows_d = dev(batch)
    assert torch.equal(fin.index_select(0, rows_d), full.index_select(0, rows_d))
    # backward: gout non-zero on the batch's rows only
    gout = torch.zeros(n, d, device="cuda")
    gout[rows_d] = torch.randn(len(batch), d, device="cuda")
    state = lambda: (E0.clone(), torch.zeros(n, d, device="cuda"), torch.rand(n, d, device="cuda") * 1e-3,  # noqa: E731
                     torch.rand(n, d, device="cuda") * 1e-6)
    torch.manual_seed(5)
    p1, g1, m1, v1 = state()
    torch.manual_seed(5)
    p2, g2, m2, v2 = state()
    G.propagate_mean_bwd_adam_raw(gout, K, True, g1, True, hops[0], p1, m1, v1, 1e-3, 3)
    G.propagate_mean_bwd_adam_fields_raw(gout, K, True, g2, True, hops[:K - 1] + [None], p2, m2, v2, 1e-3, 3)
    for a, b in ((p1, p2), (g1, g2), (m1, m2), (v1, v2)):
        assert torch.equal(a, b)


@pytest.mark.parametrize("K", [3, 2])
def test_engine_receptive_field_steps_are_the_plain_steps(ops, K, monkeypatch):
    """The fused LightGCN step with receptive-field propagation (IDG_FIELDS=1; automatic from 4 M rows on) against the
    same steps without it: identical losses and identical tables, bit for bit, after every step — with and without the
    one-batch lookahead."""
    import idgrec_amd.host as H
    import idgrec_amd.synth as S
    from idgrec_amd.engine import PropagationEngine

    U, I, E = 90000, 60000, 380000
    users, items = S.generate(U, I, E, seed=11)
    ip, ix, dv = H.build_norm_adj(U, I, users, items)
    n, d, B, steps = U + I, 64, 32, 5
    W0 = S.xavier_uniform_panel(U, I, d, 7)
    tri = dev(S.draw_triples(7, users, items, U, I, steps * B)[0][: steps * B])
    tu, tp, tn = tri[:, 0].contiguous(), tri[:, 1].contiguous(), tri[:, 2].contiguous()
    G = ops.Graph(ip, ix, dv, n, n)

    def run(fields):
        monkeypatch.setenv("IDG_FIELDS", fields)
        eng = PropagationEngine(G, U, I, d, K, include_layer0=True, reg_lambda=1e-4, lr=1e-3, params=W0.cuda())
        assert eng._fields == (fields == "1")
        losses = torch.zeros(steps, 2, device="cuda")
        for i in range(steps):
            sl = slice(i * B, (i + 1) * B)
            if i % 2 == 0 and i + 1 < steps:
                nx = slice((i + 1) * B, (i + 2) * B)
                eng.prefetch(tu[nx], tp[nx], tn[nx])
            eng.train_step(tu[sl], tp[sl], tn[sl], loss_out=losses[i])
        torch.cuda.synchronize()
        return losses.clone(), eng.params.clone(), eng.exp_avg.clone(), eng.exp_avg_sq.clone()

    plain, field = run("0"), run("1")
    for a, b in zip(plain, field):
        assert torch.equal(a, b)


@pytest.mark.parametrize("keep_prob", [0.9, 0.3])
def test_node_dropout_masked_copy(ops, golden_small, keep_prob):
    """NGCF.node_dropout (models/NGCF.py:56-65) as a masked copy of the handle: an entry survives where
    int(u + (1 - keep_prob)) != 0 — probability 1 - keep_prob, the reference's own rule — and is divided by
    (1 - keep_prob); the copy's .T carries the transposed mask (what backward multiplies by); products run on the
    base handle's schedule, so on the surviving entries they are the base product's bits scaled."""
    g = golden_small
    n = int(g["num_users"]) + int(g["num_items"])
    G = _graph(ops, g)
    eye = torch.eye(n, device="cuda")
    A = G.spmm_raw(eye).cpu().numpy()
    D = G.dropout_copy(keep_prob, stream=(4321, 7))
    Dd, Dt = D.spmm_raw(eye).cpu().numpy(), D.T.spmm_raw(eye).cpu().numpy()
    assert np.array_equal(Dt, Dd.T) and D.T.T is D
    stored = A != 0
    kept = Dd != 0
    assert not (kept & ~stored).any()
    p = 1.0 - keep_prob
    frac, sigma = kept.sum() / stored.sum(), np.sqrt(p * (1 - p) / stored.sum())
    assert abs(frac - p) < 5 * sigma, (frac, p)
    np.testing.assert_allclose(Dd[kept], A[kept] / np.float32(p), rtol=2e-7)
    assert not np.array_equal(kept, kept.T)  # (r, c) and (c, r) are drawn independently, as in the reference
    redrawn = G.dropout_copy(keep_prob, stream=(4321, 9), reuse=G.dropout_copy(keep_prob, stream=(4321, 1)))
    fresh = G.dropout_copy(keep_prob, stream=(4321, 9))
    assert torch.equal(redrawn.spmm_raw(eye), fresh.spmm_raw(eye)) and torch.equal(redrawn.T.spmm_raw(eye), fresh.T.spmm_raw(eye))
    again = G.dropout_copy(keep_prob, stream=(4321, 7)).spmm_raw(eye).cpu().numpy()
    other = G.dropout_copy(keep_prob, stream=(4321, 8)).spmm_raw(eye).cpu().numpy()
    assert np.array_equal(again, Dd) and not np.array_equal(other, Dd)
    # autograd through ops.spmm uses the transposed mask
    X = torch.randn(n, 64, device="cuda", requires_grad=True)
    W = torch.randn(n, 64, device="cuda")
    (ops.spmm(D, X) * W).sum().backward()
    np.testing.assert_allclose(X.grad.cpu().numpy(), Dd.T @ W.cpu().numpy(), rtol=1e-4, atol=1e-5)


# ------------------------------------------------------------ full-size (BASELINE) properties
@pytest.mark.parametrize("fused", ["0", "1"])
def test_yelp_shape_full_size(ops, fused, monkeypatch):
    monkeypatch.setenv("IDG_FUSED_FIX", fused)
    _shape_full_size(ops, "yelp2018")


def test_amazon_book_shape_full_size(ops):
    """BASELINE.json configs[2] / [3] (52,643 x 91,599, 2.38 M edges)."""
    _shape_full_size(ops, "amazon-book")


def _shape_full_size(ops, shape):
    """LightGCN-3 d=64 on a BASELINE-shaped graph: exact-order result bit-identical to the oracle over
    the whole panel; split schedule bit-identical to the oracle in the published order and within fp32
    rounding of the sequential one; symmetry (<y, A x> == <A y, x>) as a size-independent check; one
    whole training step (propagate, BPR, backward, Adam) against the oracle's."""
    import idgrec_amd.host as H
    import idgrec_amd.synth as S
    from idgrec_amd.engine import PropagationEngine

    U, I, E = S.SHAPES[shape]
    users, items = S.generate(U, I, E, seed=0)
    ip, ix, dv = H.build_norm_adj(U, I, users, items)
    n = U + I
    rng = np.random.default_rng(0)
    X = (rng.standard_normal((n, 64)) * 0.1).astype(np.float32)
    ref = oracle.spmm(ip, ix, dv, X)
    Ge = ops.Graph(ip, ix, dv, n, n, exact_order=True)
    assert np.array_equal(Ge.spmm_raw(dev(X)).cpu().numpy(), ref)
    G = ops.Graph(ip, ix, dv, n, n)
    assert G.info()["n_long_rows"] > 0
    sched = G.long_rows()
    Y = G.spmm_raw(dev(X))
    assert np.array_equal(Y.cpu().numpy(), oracle.spmm(ip, ix, dv, X, *sched))
    np.testing.assert_allclose(Y.cpu().numpy(), ref, rtol=RTOL, atol=1e-6)
    out = ops.propagate_mean(G, dev(X), 3, True).cpu().numpy()
    assert np.array_equal(out, oracle.propagate_mean(ip, ix, dv, X, 3, True, *sched))
    # symmetry of the operator in double precision
    Z = dev((rng.standard_normal((n, 64)) * 0.1).astype(np.float32))
    lhs = (Z.double() * Y.double()).sum().item()
    rhs = (G.spmm_raw(Z).double() * dev(X).double()).sum().item()
    assert abs(lhs - rhs) <= 1e-5 * max(1.0, abs(lhs))
    # one training step at full size: sampled triples (native sampler), loss, gradient, Adam
    pos_ptr = np.zeros(U + 1, dtype=np.int64)
    pos_ptr[1:] = np.cumsum(np.bincount(users, minlength=U))
    tri = H.Rng(2024).sample_epoch(users, items, pos_ptr, items.astype(np.int32), I)
    assert len(tri) == len(users) and np.array_equal(tri[:, 0], users) and np.array_equal(tri[:, 1], items)
    item_sets = items.astype(np.int64) + users.astype(np.int64) * I  # (user, item) pairs as keys
    assert not np.isin(tri[:, 0] * I + tri[:, 2], item_sets).any()   # a negative is never a train positive
    b = tri[np.random.default_rng(1).permutation(len(tri))[:1024]]
    eng = PropagationEngine(G, U, I, 64, 3, include_layer0=True, params=dev(X.copy()))
    loss = eng.train_step(dev(b[:, 0]), dev(b[:, 1]), dev(b[:, 2])).cpu().numpy()
    fin = oracle.propagate_mean(ip, ix, dv, X, 3, True, *sched)
    loss_o, gf, ge = oracle.bpr(fin, X, U, b[:, 0], b[:, 1], b[:, 2], 1e-4)
    np.testing.assert_allclose(loss, loss_o, rtol=1e-5)
    grad_o = oracle.propagate_mean_bwd(ip, ix, dv, gf, 3, True) + ge
    gscale = np.abs(grad_o).max()
    np.testing.assert_allclose(eng.grad.cpu().numpy(), grad_o, rtol=RTOL, atol=1e-5 * gscale)
    W, m, v = X.copy(), np.zeros_like(X), np.zeros_like(X)
    oracle.adam(W, np.ascontiguousarray(grad_o), m, v, 1e-3, 1)
    # Adam's first step is lr * sign-like: compare where the gradient is not at rounding level
    big = np.abs(grad_o) > 1e-3 * gscale
    np.testing.assert_allclose(eng.params.cpu().numpy()[big], W[big], rtol=1e-4, atol=1e-7)


# ------------------------------------------------------------------------- noise epilogue
def test_noise_epilogue_statistics_and_reproducibility(ops, golden_small):
    """X' = X + sign(X) * normalize(u) * eps, u ~ U[0,1)^d (models/SimGCL.py:50-51): every row moves
    by exactly eps (unit-norm direction), componentwise in the direction of sign(X), the noise is
    uniform, reproducible for a given (seed, stream) and independent of the split schedule."""
    g = golden_small
    n = int(g["num_users"]) + int(g["num_items"])
    E0 = dev(np.concatenate([g["d64_init_user"], g["d64_init_item"]]))
    eps = 0.05
    torch.cuda.manual_seed(1234)
    G = _graph(ops, g)
    clean = G.spmm_raw(E0)
    ops.reset_noise_stream(0)
    Y1 = ops.spmm_perturbed(G, E0, eps)
    ops.reset_noise_stream(0)
    Y2 = ops.spmm_perturbed(_graph(ops, g, exact_order=True), E0, eps)   # different tile / split schedule
    Y3 = ops.spmm_perturbed(G, E0, eps)                                   # next stream id
    delta = Y1 - clean
    live = clean.abs().sum(1) > 0
    np.testing.assert_allclose(delta[live].norm(dim=1).cpu().numpy(), eps, rtol=2e-4)       # ||normalize(u)|| = 1
    assert torch.all(delta * torch.sign(clean) >= 0)                                         # moves away from zero
    assert torch.count_nonzero(delta[~live]) == 0                                            # sign(0) = 0
    assert torch.allclose(Y1, Y2, rtol=1e-5, atol=1e-7)                                      # schedule independent
    assert not torch.allclose(Y1, Y3)
    # recover u up to the row norm: uniform on [0,1) => mean 1/2, var 1/12 after un-normalising
    u_dir = (delta[live] / eps / torch.sign(clean[live]).clamp(min=1)).abs()
    ratio = u_dir / u_dir.mean(dim=1, keepdim=True)                                          # E[u]/mean(u) ~ 1
    assert abs(float(ratio.mean()) - 1.0) < 1e-3 and 0.30 < float(ratio.std()) < 0.36 * 2    # sqrt(1/12)/0.5 = 0.577
    # d = 256 and the fused K-layer form
    for d in (256,):
        X = torch.randn(n, d, device="cuda") * 0.1
        P = ops.propagate_views(G, X, 3, False, eps, n_views=2)
        assert len(P) == 3 and torch.equal(P[0], G.propagate_mean_raw(X, 3, False))
        assert not torch.allclose(P[1], P[2]) and float((P[1] - P[0]).norm(dim=1).max()) < 3 * eps


def test_propagate_views_single_backward(ops, golden_small):
    g = golden_small
    n = int(g["num_users"]) + int(g["num_items"])
    E0 = dev(np.concatenate([g["d64_init_user"], g["d64_init_item"]])).requires_grad_(True)
    G = _graph(ops, g)
    clean, v1, v2 = ops.propagate_views(G, E0, 3, False, 0.05, n_views=2)
    g0, g1, g2 = (torch.randn(n, 64, device="cuda") for _ in range(3))
    (clean * g0).sum().backward(retain_graph=True)
    ref0 = E0.grad.clone()
    E0.grad = None
    ((clean * g0).sum() + (v1 * g1).sum() + (v2 * g2).sum()).backward()
    want = G.propagate_mean_bwd_raw(g0 + g1 + g2, 3, False)
    assert torch.allclose(E0.grad, want, rtol=1e-5, atol=1e-7)
    assert torch.allclose(ref0, G.propagate_mean_bwd_raw(g0, 3, False), rtol=1e-5, atol=1e-7)


@pytest.mark.parametrize("n,p_drop,with_ge,masked", [(1000, 0.0, True, False), (4133, 0.1, True, True), (69716, 0.1, False, True),
                                                     (63, 0.3, True, False)])
def test_ngcf_layer_kernels_equal_the_chain(n, p_drop, with_ge, masked):
    """idg_ngcf_layer_fwd_f32 / idg_ngcf_layer_bwd_f32 (one kernel per NGCF layer and direction, d = 64) against the chain
    they replace (transform + tail; tail' + parameter gradients + transform'): E, N, g_side, g_ego bit for bit, the
    parameter gradients (another summation order) against float64 sums within 2e-5 of their scale."""
    import ctypes as C

    from idgrec_amd import native, ops

    lib, check = native.lib, native.check
    d, D = 64, 256
    g = torch.Generator(device="cuda").manual_seed(n)
    rnd = lambda *sh: torch.randn(*sh, device="cuda", generator=g)  # noqa: E731
    side, ego = rnd(n, d) * 0.3, rnd(n, d) * 0.3
    W1, W2, b1, b2 = rnd(d, d) * 0.1, rnd(d, d) * 0.1, rnd(d) * 0.1, rnd(d) * 0.1
    seed, sid, slope = 1234, 7, 0.2
    p_ = lambda t: None if t is None else C.c_void_p(t.data_ptr())  # noqa: E731
    st = ops._stream()
    # forward
    S, BI = torch.empty(n, d, device="cuda"), torch.empty(n, d, device="cuda")
    E0, E1 = torch.empty(n, d, device="cuda"), torch.empty(n, d, device="cuda")
    F0, F1 = torch.full((n, D), 7.0, device="cuda"), torch.full((n, D), 7.0, device="cuda")
    slot = lambda F: C.c_void_p(F.data_ptr() + 4 * 2 * d)  # noqa: E731
    check(lib.idg_ngcf_transform_f32(p_(side), p_(ego), p_(W1), p_(W2), n, d, d, p_(S), p_(BI), st), "t")
    check(lib.idg_ngcf_tail_ex_f32(p_(S), None, p_(b1), p_(b2), n, d, slope, p_drop, C.c_uint64(seed), C.c_uint64(sid), p_(E0),
                                   slot(F0), D, st), "tail")
    check(lib.idg_ngcf_layer_fwd_f32(p_(side), p_(ego), p_(W1), p_(W2), p_(b1), p_(b2), n, d, slope, p_drop, C.c_uint64(seed),
                                     C.c_uint64(sid), p_(E1), slot(F1), D, st), "fwd")
    assert torch.equal(E0, E1) and torch.equal(F0, F1)
    assert float(F1[:, :2 * d].min()) == 7.0 and float(F1[:, 3 * d:].max()) == 7.0  # only the slot was written
    # backward
    gE = rnd(n, d) if with_ge else None
    gF = rnd(n, D)
    bitmap = None
    if masked:
        rows = torch.randperm(n, device="cuda", generator=g)[:max(1, n // 20)]
        bits = torch.zeros((n + 31) // 32 * 32, dtype=torch.bool, device="cuda")
        bits[rows] = True
        w = (bits.view(-1, 32).to(torch.int64) << torch.arange(32, device="cuda")).sum(1)
        bitmap = torch.where(w >= 2 ** 31, w - 2 ** 32, w).to(torch.int32)
    gslot = C.c_void_p(gF.data_ptr() + 4 * 2 * d)
    gT = torch.empty(n, d, device="cuda")
    check(lib.idg_ngcf_tail_bwd_ex_f32(p_(E0), p_(gE), gslot, D, p_(bitmap), n, d, slope, p_drop, C.c_uint64(seed), C.c_uint64(sid),
                                       p_(gT), st), "tb")
    gs0, ge0, gs1, ge1 = (torch.empty(n, d, device="cuda") for _ in range(4))
    check(lib.idg_ngcf_transform_bwd_f32(p_(gT), p_(side), p_(ego), p_(W1), p_(W2), n, d, d, p_(gs0), p_(ge0), st), "trb")
    ws = torch.empty(int(lib.idg_ngcf_layer_bwd_workspace_bytes(d)), dtype=torch.uint8, device="cuda")
    wg = torch.empty(2 * d * d + 2 * d, device="cuda")
    check(lib.idg_ngcf_layer_bwd_f32(p_(E1), p_(gE), gslot, D, p_(bitmap), p_(side), p_(ego), p_(W1), p_(W2), n, d, slope, p_drop,
                                     C.c_uint64(seed), C.c_uint64(sid), p_(gs1), p_(ge1), p_(wg), p_(ws), st), "bwd")
    assert torch.equal(gs0, gs1) and torch.equal(ge0, ge1)
    t64, s64, b64 = gT.double(), side.double(), (side * ego).double()
    want = torch.cat([(s64.t() @ t64).reshape(-1), t64.sum(0), (b64.t() @ t64).reshape(-1), t64.sum(0)])
    scale = float(want.abs().max())
    assert float((wg.double() - want).abs().max()) <= 2e-5 * scale + 1e-12
    # other widths are refused, not mis-computed
    assert lib.idg_ngcf_layer_fwd_f32(p_(side), p_(ego), p_(W1), p_(W2), p_(b1), p_(b2), n, 128, slope, p_drop, C.c_uint64(seed),
                                      C.c_uint64(sid), p_(E1), slot(F1), D, st) != 0
