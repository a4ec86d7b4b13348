"""GPU end-to-end: the model plugins + trainer + evaluator against what the reference logged
and learned on the same data and seed (goldens `loop_*` dumped through the reference's own
universal_trainer), and fused-vs-autograd equivalence of the two step implementations."""
import io
import logging
import re

import numpy as np
import pytest

torch = pytest.importorskip("torch")
pytestmark = pytest.mark.gpu

RTOL = 1e-4


def _dataset(tmp_path, g, name, **cfg):
    import utility.utility_data.data_loader as data_loader

    d = tmp_path / name
    d.mkdir(exist_ok=True)
    (d / "train.txt").write_bytes(g["train_txt"].tobytes())
    (d / "test.txt").write_bytes(g["test_txt"].tobytes())
    config = dict(dataset=name, dataset_path=str(tmp_path) + "/", sparsity_test="0")
    config.update({k: str(v) for k, v in cfg.items()})
    return data_loader.Data(str(d), config), config


def _numbers(line):
    return [float(x) for x in re.findall(r"[-+]?\d+\.?\d*(?:e[-+]?\d+)?", line.replace("Training time: T", ""))]


BASE = dict(embedding_size=64, reg_lambda=0.0001, GCN_layer=3, batch_size=256, test_batch_size=64, training_epochs=3,
            interval=2, top_K="[5, 10]", early_stopping=10)


@pytest.mark.parametrize("mname,lr", [("lgcn", 0.001), ("mf", 0.0001)])
def test_full_loop_vs_reference_run(mname, lr, tmp_path, golden_small):
    import utility.utility_function.tools as tools
    import utility.utility_train.trainer as trainer
    from models.LightGCN import LightGCN
    from models.MFBPR import MFBPR

    g = golden_small
    data, cfg = _dataset(tmp_path, g, "small", learn_rate=lr, **BASE)
    stream = io.StringIO()
    logger = logging.getLogger("gpu_loop_" + mname)
    logger.setLevel(logging.INFO)
    logger.handlers = [logging.StreamHandler(stream)]
    tools.set_seed(2024)
    model = (LightGCN if mname == "lgcn" else MFBPR)(cfg, data, torch.device("cuda"))
    # identical initial weights: same torch RNG call order as the reference's constructor
    assert np.array_equal(model.user_embedding.weight.detach().cpu().numpy(), g["d64_init_user"])
    assert np.array_equal(model.item_embedding.weight.detach().cpu().numpy(), g["d64_init_item"])
    trainer.universal_trainer(model, None, cfg, data, torch.device("cuda"), logger)
    lines = stream.getvalue().splitlines()
    ref_lines = g["loop_%s_log" % mname].tolist()
    assert len(lines) == len(ref_lines)
    for mine, ref in zip(lines, ref_lines):
        mine = re.sub(r"Training time: [0-9.]+", "Training time: T", mine)
        a, b = _numbers(mine), _numbers(ref)
        assert len(a) == len(b), (mine, ref)
        if "training loss" in ref:
            # 6-decimal strings; a last-digit flip at a rounding boundary is fp32 noise
            np.testing.assert_allclose(a, b, rtol=RTOL, atol=1.5e-6)
        else:
            np.testing.assert_allclose(a, b, rtol=1e-6, atol=1e-8)  # recall / ndcg / best epoch
    np.testing.assert_allclose(model.user_embedding.weight.detach().cpu().numpy(), g["loop_%s_user" % mname],
                               rtol=RTOL, atol=1e-7)
    np.testing.assert_allclose(model.item_embedding.weight.detach().cpu().numpy(), g["loop_%s_item" % mname],
                               rtol=RTOL, atol=1e-7)


def test_recall_curve_matches_reference_over_40_epochs(tmp_path):
    """BASELINE.json "Recall@20 parity": the reference trained LightGCN-3 d=64 for 40 epochs on the medium
    synthetic dataset (oracle/gen_golden_convergence.py, CPU, 4000 x 3000, 120 k edges); the MI355X path trains on
    the same files, seed and configuration.  Same sampled triples, same initial weights, the same summation order
    in the sparse products: the curves coincide — Recall@K / NDCG@K at every logged test within 1e-5 absolute
    (measured 3e-7; the north-star bar is 1e-3), logged epoch losses within 1e-4 relative (measured: identical
    6-decimal strings), the learned tables after 3,960 Adam steps within 1e-4 of the reference's (measured 5e-7)."""
    import os

    import utility.utility_data.data_loader as data_loader
    import utility.utility_function.tools as tools
    import utility.utility_train.trainer as trainer
    from models.LightGCN import LightGCN

    path = os.path.join(os.path.dirname(__file__), "golden", "convergence_medium.npz")
    g = np.load(path)
    d = tmp_path / "medium"
    d.mkdir()
    (d / "train.txt").write_bytes(g["train_txt"].tobytes())
    (d / "test.txt").write_bytes(g["test_txt"].tobytes())
    cfg = dict(zip(g["config_keys"].tolist(), g["config_values"].tolist()))
    cfg.update(dataset="medium", dataset_path=str(tmp_path) + "/")
    stream = io.StringIO()
    logger = logging.getLogger("gpu_convergence")
    logger.setLevel(logging.INFO)
    logger.handlers = [logging.StreamHandler(stream)]
    tools.set_seed(2024)
    data = data_loader.Data(cfg["dataset_path"] + cfg["dataset"], cfg)
    model = LightGCN(cfg, data, torch.device("cuda"))
    trainer.universal_trainer(model, None, cfg, data, torch.device("cuda"), logger)
    mine = [re.sub(r"Training time: [0-9.]+", "Training time: T", ln) for ln in stream.getvalue().splitlines()]
    ref = g["log"].tolist()
    assert len(mine) == len(ref)
    tests_seen = 0
    for a, b in zip(mine, ref):
        x, y = np.array(_numbers(a)), np.array(_numbers(b))
        assert x.shape == y.shape, (a, b)
        if "training loss" in b:
            np.testing.assert_allclose(x[1:], y[1:], rtol=1e-4, atol=2e-6)
        elif "Test recall" in b:
            assert x[0] == y[0]  # epoch number
            np.testing.assert_allclose(x[1:], y[1:], rtol=0, atol=1e-5)
            tests_seen += 1
        elif "Best epoch" in b:
            assert x[0] == y[0]
            np.testing.assert_allclose(x[1:], y[1:], rtol=0, atol=1e-5)
    assert tests_seen == 8  # epochs 1, 6, ..., 36
    # the learned tables themselves after 40 epochs x 99 Adam steps
    wu, wi = model.user_embedding.weight.detach().cpu().numpy(), model.item_embedding.weight.detach().cpu().numpy()
    scale = np.abs(g["final_user"]).max()
    assert np.abs(wu - g["final_user"]).max() <= 1e-4 * scale and np.abs(wi - g["final_item"]).max() <= 1e-4 * scale


@pytest.mark.parametrize("mname", ["lgcn", "mf"])
def test_fused_step_equals_autograd_step(mname, tmp_path, golden_small):
    import utility.utility_function.tools as tools
    from idgrec_amd import ops
    from models.LightGCN import LightGCN
    from models.MFBPR import MFBPR

    g = golden_small
    data, cfg = _dataset(tmp_path, g, "small", learn_rate=0.001, **BASE)
    batch = torch.from_numpy(g["d64_batch"]).cuda()
    grads, losses_ = [], []
    for fused in (False, True):
        tools.set_seed(2024)
        model = (LightGCN if mname == "lgcn" else MFBPR)(cfg, data, torch.device("cuda")).to("cuda")
        if fused:
            loss = model.fused_loss_and_grad(batch[:, 0].contiguous(), batch[:, 1].contiguous(), batch[:, 2].contiguous())
            losses_.append(loss.cpu().numpy().copy())
        else:
            ll = model(batch[:, 0], batch[:, 1], batch[:, 2])
            sum(ll).backward()
            losses_.append(np.array([x.item() for x in ll], dtype=np.float32))
        grads.append((model.user_embedding.weight.grad.cpu().numpy().copy(),
                      model.item_embedding.weight.grad.cpu().numpy().copy()))
    assert np.array_equal(losses_[0], losses_[1])
    np.testing.assert_allclose(grads[0][0], grads[1][0], rtol=1e-6, atol=1e-10)
    np.testing.assert_allclose(grads[0][1], grads[1][1], rtol=1e-6, atol=1e-10)
    key = "d64_lgcn" if mname == "lgcn" else "d64_mf"
    np.testing.assert_allclose(grads[1][0], g[key + "_grad_user"], rtol=RTOL, atol=1e-8)
    np.testing.assert_allclose(grads[1][1], g[key + "_grad_item"], rtol=RTOL, atol=1e-8)


@pytest.mark.parametrize("mname", ["lgcn", "mf"])
def test_fused_train_step_equals_grad_then_optimizer_step(mname, tmp_path, golden_small):
    """One chain (Adam in the last backward epilogue, moments in packed panels viewed by the optimizer state)
    == fused gradients + optimizer.step(), bit for bit, including switching between the two mid-run."""
    import utility.utility_function.tools as tools
    from idgrec_amd import ops
    from models.LightGCN import LightGCN
    from models.MFBPR import MFBPR

    g = golden_small
    data, cfg = _dataset(tmp_path, g, "small", learn_rate=0.001, **BASE)
    tri = torch.from_numpy(g["sample1"][:5 * 256]).cuda()
    bt = [tuple(tri[i * 256:(i + 1) * 256, c].contiguous() for c in range(3)) for i in range(5)]
    res = []
    for plan in ("TTTTT", "FFFFF", "FTTFT", "ttttt"):
        tools.set_seed(2024)
        model = (LightGCN if mname == "lgcn" else MFBPR)(cfg, data, torch.device("cuda")).to("cuda")
        # upper case: the fused step also writes the gradient panel out (.grad readable afterwards); lower case: the
        # trainer's default — the update consumes the gradient in the epilogue, .grad is None after the step
        model.keep_fused_grad = plan.isupper()
        plan = plan.upper()
        opt = ops.Adam(model.parameters(), lr=0.001)
        loss = torch.zeros((5, 2), device="cuda")
        for i, one_chain in enumerate(plan):
            if one_chain == "T":
                assert model.fused_train_step(*bt[i], loss[i], opt)
            else:
                model.fused_loss_and_grad(*bt[i], loss_out=loss[i])
                opt.step()
        st = opt.state[model.item_embedding.weight]
        assert st["step"] == 5 and opt.state[model.user_embedding.weight]["step"] == 5
        gr = model.user_embedding.weight.grad
        assert (gr is not None) == model.keep_fused_grad, "gradient panel: readable exactly when asked for"
        res.append((model._storage.clone(), st["exp_avg"].clone(), st["exp_avg_sq"].clone(), loss.clone(),
                    None if gr is None else gr.clone()))
    for other in res[1:]:
        for a, b in zip(res[0], other):
            assert b is None or torch.equal(a, b)
    # a foreign optimizer is left to the two-call form
    assert not model.fused_train_step(*bt[0], loss[0], torch.optim.Adam(model.parameters(), lr=0.001))


def test_packed_storage_survives_to_and_optimizer(tmp_path, golden_small):
    from idgrec_amd import ops
    from models.MFBPR import MFBPR

    data, cfg = _dataset(tmp_path, golden_small, "small", learn_rate=0.001, **BASE)
    model = MFBPR(cfg, data, torch.device("cuda"))
    model.to("cuda")
    assert model._is_packed() and model.user_embedding.weight.is_cuda
    opt = ops.Adam(model.parameters(), lr=0.01)
    batch = torch.from_numpy(golden_small["d64_batch"]).cuda()
    before = model._storage.clone()
    sum(model(batch[:, 0], batch[:, 1], batch[:, 2])).backward()
    opt.step()
    assert model._is_packed() and not torch.equal(before, model._storage)
    assert torch.equal(model.user_embedding.weight.data, model._storage[: data.num_users])


def test_evaluator_fused_vs_dense_path(tmp_path, golden_small):
    import utility.utility_train.batch_test as batch_test
    from models.LightGCN import LightGCN

    g = golden_small
    data, cfg = _dataset(tmp_path, g, "small", learn_rate=0.001, **dict(BASE, top_K="[10, 20]"))
    import utility.utility_function.tools as tools

    tools.set_seed(2024)
    model = LightGCN(cfg, data, torch.device("cuda")).to("cuda")
    res = batch_test.Test(data, model, torch.device("cuda"), cfg)  # fused top-K path
    np.testing.assert_allclose(np.stack([res["recall"], res["precision"], res["ndcg"]]), g["d64_lgcn_test_10_20"],
                               rtol=1e-6, atol=1e-9)
    # the dense protocol (get_rating_for_test + mask + torch.topk) gives the same metrics
    class Dense:
        def __init__(self, m):
            self.m = m

        def eval(self):
            self.m.eval()
            return self

        def get_rating_for_test(self, u):
            return self.m.get_rating_for_test(u)

    res2 = batch_test.Test(data, Dense(model), torch.device("cuda"), cfg)
    np.testing.assert_allclose(res2["recall"], res["recall"], rtol=1e-6, atol=1e-9)
    R = model.get_rating_for_test(torch.from_numpy(g["test_dict_users"][:48]).cuda()).cpu().numpy()
    np.testing.assert_allclose(R, g["d64_lgcn_rating"], rtol=1e-5, atol=1e-6)


def test_topk_for_test_any_k(tmp_path):
    """model.topk_for_test for k below, at and beyond what the fused entry point returns in one call (64 ranks per pass,
    1024 per call; beyond: dense rating + the reference's mask + torch.topk): the same lists as ranking the dense
    rating matrix by (score descending, item ascending) wherever scores are distinct."""
    import idgrec_amd.synth as S
    import utility.utility_data.data_loader as data_loader
    import utility.utility_function.tools as tools
    from models.LightGCN import LightGCN

    S.make_dataset(str(tmp_path), "medium", n_test=2)
    cfg = _cfg("LightGCN", dataset="medium", dataset_path=str(tmp_path) + "/")
    data = data_loader.Data(str(tmp_path / "medium"), cfg)
    tools.set_seed(2024)
    model = LightGCN(cfg, data, torch.device("cuda")).to("cuda").eval()
    users = torch.arange(0, 97, device="cuda")
    rating = model.get_rating_for_test(users).cpu().numpy()
    for b, u in enumerate(users.tolist()):
        rating[b, data.get_user_pos_items([u])[0]] = -1
    order = np.lexsort((np.arange(rating.shape[1])[None, :].repeat(len(users), 0), -rating), axis=1)
    for k in (20, 64, 65, 1024, 1100):
        got = model.topk_for_test(users, k).cpu().numpy()
        assert got.shape == (len(users), k)
        vals = np.take_along_axis(rating, got, axis=1)
        np.testing.assert_allclose(vals, np.take_along_axis(rating, order[:, :k], axis=1), rtol=0, atol=2e-6)
        assert all(len(set(r.tolist())) == k for r in got)


def test_simgcl_runs_and_clean_view_matches_reference(tmp_path, golden_small):
    import utility.utility_function.tools as tools
    from idgrec_amd import ops
    from models.SimGCL import SimGCL

    g = golden_small
    data, cfg = _dataset(tmp_path, g, "small", learn_rate=0.001, ssl_lambda=0.5, temperature=0.2, epsilon=0.05, **BASE)
    tools.set_seed(2024)
    model = SimGCL(cfg, data, torch.device("cuda")).to("cuda")
    with torch.no_grad():
        u, i = model.aggregate(perturbed=False)
    np.testing.assert_allclose(u.cpu().numpy(), g["d64_simgcl_user"], rtol=1e-5, atol=1e-8)
    np.testing.assert_allclose(i.cpu().numpy(), g["d64_simgcl_item"], rtol=1e-5, atol=1e-8)
    with torch.no_grad():
        pu, pi = model.aggregate(perturbed=True)
    # the perturbation moves every row by exactly eps per layer on the unit sphere: bounded, non-zero
    delta = (torch.cat([pu, pi]) - torch.cat([u, i])).norm(dim=1)
    assert float(delta.max()) < 3 * 0.05 and float(delta.median()) > 0.01  # isolated nodes (zero rows) stay put: sign(0) = 0
    batch = torch.from_numpy(g["d64_batch"]).cuda()
    opt = ops.Adam(model.parameters(), lr=1e-3)
    first = None
    for _ in range(5):
        ll = model(batch[:, 0], batch[:, 1], batch[:, 2])
        assert len(ll) == 3 and all(torch.isfinite(x) for x in ll)
        opt.zero_grad()
        sum(ll).backward()
        opt.step()
        first = first if first is not None else float(sum(ll))
    assert float(sum(ll)) < first  # five steps on one batch reduce its loss


@pytest.mark.parametrize("keep_grad", [True, False])
@pytest.mark.parametrize("layers", [2, 3, 4])
def test_xsimgcl_fused_step_equals_autograd_step(layers, keep_grad, tmp_path, golden_small):
    """XSimGCL's fused trainer step (cl_layer = 1) against forward() under autograd + optimizer.step(), same noise
    streams: three loss terms, gradients and weights over 3 steps.  keep_grad False is what the trainer and the bench run
    (the Horner chain whose last product applies Adam and writes no gradient panel: weights and moments are compared);
    GCN_layer = 2 has an empty inner loop, 4 two inner products."""
    import utility.utility_function.tools as tools
    from idgrec_amd import ops
    from models.XSimGCL import XSimGCL

    g = golden_small
    data, cfg = _dataset(tmp_path, g, "small", learn_rate=0.001, ssl_lambda=0.2, temperature=0.15, epsilon=0.2, cl_layer=1,
                         **dict(BASE, GCN_layer=layers))
    tri = torch.from_numpy(g["sample1"][:3 * 256]).cuda()
    bt = [tuple(tri[i * 256:(i + 1) * 256, c].contiguous() for c in range(3)) for i in range(3)]
    res = []
    for fused in (True, False):
        tools.set_seed(2024)
        ops.reset_noise_stream()
        model = XSimGCL(cfg, data, torch.device("cuda")).to("cuda")
        assert model.supports_fused_step
        model.keep_fused_grad = keep_grad  # (by default a fused step does not write the gradient panel out)
        opt = ops.Adam(model.parameters(), lr=0.001)
        loss = torch.zeros((3, 3), device="cuda")
        for i in range(3):
            if fused:
                assert model.fused_train_step(*bt[i], loss[i], opt)
            else:
                ll = model(*bt[i])
                loss[i] = torch.stack([x.detach() for x in ll])
                opt.zero_grad()
                sum(ll).backward()
                opt.step()
        grad = model.user_embedding.weight.grad
        assert fused and not keep_grad or grad is not None
        res.append((loss.cpu().numpy(), None if grad is None else grad.cpu().numpy(), model._storage.cpu().numpy(),
                    opt.state[model.item_embedding.weight]["exp_avg"].cpu().numpy(),
                    opt.state[model.user_embedding.weight]["exp_avg_sq"].cpu().numpy()))
    (l_f, g_f, w_f, m_f, v_f), (l_a, g_a, w_a, m_a, v_a) = res
    np.testing.assert_allclose(l_f, l_a, rtol=2e-5)
    if keep_grad:
        np.testing.assert_allclose(g_f, g_a, rtol=1e-3, atol=1e-5 * np.abs(g_a).max())
    else:
        assert g_f is None
    np.testing.assert_allclose(w_f, w_a, rtol=1e-4, atol=1e-6)
    np.testing.assert_allclose(m_f, m_a, rtol=1e-3, atol=1e-5 * np.abs(m_a).max())
    np.testing.assert_allclose(v_f, v_a, rtol=2e-3, atol=1e-6 * np.abs(v_a).max())


def test_simgcl_fused_step_equals_autograd_step(tmp_path, golden_small):
    """The trainer's fused SimGCL step (row-restricted clean + perturbed passes, BPR, fused InfoNCE whose gradients
    join the BPR gradient before ONE shared backward propagation with Adam in its epilogue) against forward()
    under autograd + optimizer.step(), same noise streams: losses, gradients and weights over 3 steps."""
    import utility.utility_function.tools as tools
    from idgrec_amd import ops
    from models.SimGCL import SimGCL

    g = golden_small
    data, cfg = _dataset(tmp_path, g, "small", learn_rate=0.001, ssl_lambda=0.5, temperature=0.2, epsilon=0.05, **BASE)
    tri = torch.from_numpy(g["sample1"][:3 * 256]).cuda()
    bt = [tuple(tri[i * 256:(i + 1) * 256, c].contiguous() for c in range(3)) for i in range(3)]
    res = []
    for fused in (True, False):
        tools.set_seed(2024)
        ops.reset_noise_stream()
        model = SimGCL(cfg, data, torch.device("cuda")).to("cuda")
        model.keep_fused_grad = True  # (.grad is compared below)
        opt = ops.Adam(model.parameters(), lr=0.001)
        loss = torch.zeros((3, 3), device="cuda")
        for i in range(3):
            if fused:
                assert model.fused_train_step(*bt[i], loss[i], opt)
            else:
                ll = model(*bt[i])
                loss[i] = torch.stack([x.detach() for x in ll])
                opt.zero_grad()
                sum(ll).backward()
                opt.step()
        res.append((loss.cpu().numpy(), model.user_embedding.weight.grad.cpu().numpy(), model._storage.cpu().numpy(),
                    opt.state[model.item_embedding.weight]["exp_avg_sq"].cpu().numpy()))
    (l_f, g_f, w_f, v_f), (l_a, g_a, w_a, v_a) = res
    np.testing.assert_allclose(l_f, l_a, rtol=2e-5)
    np.testing.assert_allclose(g_f, g_a, rtol=1e-3, atol=1e-5 * np.abs(g_a).max())
    np.testing.assert_allclose(w_f, w_a, rtol=1e-4, atol=1e-6)
    np.testing.assert_allclose(v_f, v_a, rtol=2e-3, atol=1e-6 * np.abs(v_a).max())


@pytest.mark.parametrize("model_name", ["LightGCN", "SimGCL", "XSimGCL", "SGL", "MFBPR"])
def test_embedding_size_without_tiled_kernels_trains(model_name, tmp_path, golden_small):
    """embedding_size = 48 (the reference runs any width): the fused step is built from kernels that exist for
    32/64/128/256/512 only, so the trainer must take the autograd path (any-width kernels) instead of failing in the
    first step — two epochs train, the loss falls, evaluation runs.  top_K = [20, 100] also takes the evaluator past
    the 64 ranks one top-K pass holds."""
    import importlib

    import utility.utility_function.tools as tools

    g = golden_small
    cfg = _cfg(model_name, embedding_size=48, batch_size=256, test_batch_size=64, training_epochs=2, interval=1,
               top_K="[20, 100]", early_stopping=10, sparsity_test=0)
    data = _data_with(tmp_path, g, cfg)
    stream = io.StringIO()
    logger = logging.getLogger("gpu_d48_" + model_name)
    logger.setLevel(logging.INFO)
    logger.handlers = [logging.StreamHandler(stream)]
    tools.set_seed(2024)
    tr = importlib.import_module("models." + model_name).Trainer(None, cfg, data, torch.device("cuda"), logger)
    tr.train()
    assert tr.model.fused_step_available() == (model_name == "MFBPR")  # (plain matrix factorisation has no tiled kernel)
    lines = stream.getvalue().splitlines()
    losses = [_numbers(l.split("training loss:")[1])[0] for l in lines if "training loss" in l]
    assert len(losses) == 2 and np.isfinite(losses).all() and losses[1] < losses[0]
    recalls = [l for l in lines if "Test recall" in l]
    assert len(recalls) >= 2


# ----------------------------------------------------------------- next models
@pytest.fixture(scope="module")
def golden_next():
    import os

    return dict(np.load(os.path.join(os.path.dirname(__file__), "golden", "next_small.npz")))


def _cfg(name, **kw):
    import os

    import utility.utility_function.tools as tools

    root = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
    cfg = tools.read_configuration(os.path.join(root, "configure", name + ".txt"), name)
    cfg.update({k: str(v) for k, v in kw.items()})
    return cfg


def _data_with(tmp_path, g, cfg):
    import utility.utility_data.data_loader as data_loader

    d = tmp_path / "small"
    d.mkdir(exist_ok=True)
    (d / "train.txt").write_bytes(g["train_txt"].tobytes())
    (d / "test.txt").write_bytes(g["test_txt"].tobytes())
    cfg.update(dataset="small", dataset_path=str(tmp_path) + "/")
    return data_loader.Data(str(d), cfg)


def test_ngcf_vs_reference(tmp_path, golden_small, golden_next):
    import utility.utility_function.tools as tools
    from models.NGCF import NGCF

    g, nx = golden_small, golden_next
    cfg = _cfg("NGCF", mess_drop_prob="[0.0, 0.0, 0.0]")  # the goldens were taken with message dropout off
    data = _data_with(tmp_path, g, cfg)
    tools.set_seed(2024)
    m = NGCF(cfg, data, torch.device("cuda")).to("cuda")
    assert np.array_equal(m.user_embedding.weight.detach().cpu().numpy(), nx["ngcf_init_user"])
    for k, v in m.weight_dict.items():  # same torch RNG order as the reference's constructor
        assert np.array_equal(v.detach().cpu().numpy(), nx["ngcf_" + k]), k
    m.eval()
    au, ai = m.aggregate()
    np.testing.assert_allclose(au.detach().cpu().numpy(), nx["ngcf_user"], rtol=RTOL, atol=1e-6)
    np.testing.assert_allclose(ai.detach().cpu().numpy(), nx["ngcf_item"], rtol=RTOL, atol=1e-6)
    b = torch.from_numpy(nx["batch"]).cuda()
    ll = m(b[:, 0], b[:, 1], b[:, 2])
    np.testing.assert_allclose([x.item() for x in ll], nx["ngcf_loss"], rtol=RTOL)
    sum(ll).backward()
    np.testing.assert_allclose(m.user_embedding.weight.grad.cpu().numpy(), nx["ngcf_grad_user"], rtol=1e-3, atol=1e-8)
    np.testing.assert_allclose(m.item_embedding.weight.grad.cpu().numpy(), nx["ngcf_grad_item"], rtol=1e-3, atol=1e-8)
    np.testing.assert_allclose(m.weight_dict["W_gcn_0"].grad.cpu().numpy(), nx["ngcf_grad_W_gcn_0"], rtol=1e-3, atol=1e-7)
    np.testing.assert_allclose(m.weight_dict["b_bi_2"].grad.cpu().numpy(), nx["ngcf_grad_b_bi_2"], rtol=1e-3, atol=1e-7)
    R = m.get_rating_for_test(torch.from_numpy(g["test_dict_users"][:32]).cuda()).cpu().numpy()
    np.testing.assert_allclose(R, nx["ngcf_rating"], rtol=1e-4, atol=1e-5)
    idx = m.topk_for_test(torch.from_numpy(g["test_dict_users"][:32]).cuda(), 10)  # d' = 256 scoring path
    assert idx.shape == (32, 10)


@pytest.mark.parametrize("drop", ["[0.0, 0.0, 0.0]", "[0.1, 0.1, 0.1]", "node"])
def test_ngcf_fused_step_vs_reference_and_autograd(drop, tmp_path, golden_small, golden_next):
    """The fused, autograd-free NGCF step (idgrec_amd/ngcf.py: products, fp32-MFMA transforms, layer tails writing into the
    concatenated final rows, BPR over (K+1)d-wide rows with the item-only regulariser, the whole backward chain, Adam in
    the last product's epilogue and ONE Adam launch for the 12 small tensors): with message dropout off against the
    REFERENCE's losses and gradients (next_small.npz); with the configured dropout against the differentiable operators
    on the same dropout streams — losses, every gradient, every parameter after the update."""
    import utility.utility_function.tools as tools
    from idgrec_amd import ops
    from models.NGCF import NGCF

    g, nx = golden_small, golden_next
    if drop == "node":
        # node dropout too (a masked copy of the graph redrawn per step, its transposed copy for the backward products):
        # the fused chain against the differentiable operators on the same sequence of draws
        cfg = _cfg("NGCF", mess_drop_prob="[0.1, 0.1, 0.1]", node_dropout=True, node_keep_prob=0.2)
    else:
        cfg = _cfg("NGCF", mess_drop_prob=drop)
    data = _data_with(tmp_path, g, cfg)
    b = torch.from_numpy(nx["batch"]).cuda()
    bu, bp, bn = b[:, 0].contiguous(), b[:, 1].contiguous(), b[:, 2].contiguous()
    lr = float(cfg["learn_rate"])

    tools.set_seed(2024)
    m1 = NGCF(cfg, data, torch.device("cuda")).to("cuda")
    opt1 = ops.Adam(list(m1.parameters()), lr=lr)
    ops.reset_noise_stream()
    ll = m1(bu, bp, bn)
    opt1.zero_grad()
    sum(ll).backward()
    grads1 = {k: v.grad.clone() for k, v in m1.named_parameters()}
    opt1.step()

    tools.set_seed(2024)
    m2 = NGCF(cfg, data, torch.device("cuda")).to("cuda")
    assert m2.fused_step_available()
    opt2 = ops.Adam(list(m2.parameters()), lr=lr)
    eng = m2.ngcf_engine()
    eng.store_grad = True
    ops.reset_noise_stream()
    out = torch.zeros(2, device="cuda")
    assert m2.fused_train_step(bu, bp, bn, out, opt2)
    np.testing.assert_allclose(out.cpu().numpy(), [x.item() for x in ll], rtol=2e-5)
    U = data.num_users
    got = {"user_embedding.weight": eng.GRAD[:U], "item_embedding.weight": eng.GRAD[U:]}
    for l in range(m2.n_layers):
        for nm, gv in zip(("W_gcn_%d", "b_gcn_%d", "W_bi_%d", "b_bi_%d"), eng.small_grads()[l]):
            got["weight_dict." + nm % l] = gv
    assert set(got) == set(grads1)
    for k, want in grads1.items():
        scale = float(want.abs().max())
        assert float((got[k] - want).abs().max()) <= 2e-4 * scale + 1e-12, k
    if drop == "[0.0, 0.0, 0.0]":  # the reference's own numbers
        np.testing.assert_allclose(out.cpu().numpy(), nx["ngcf_loss"], rtol=RTOL)
        np.testing.assert_allclose(got["user_embedding.weight"].cpu().numpy(), nx["ngcf_grad_user"], rtol=1e-3, atol=1e-8)
        np.testing.assert_allclose(got["item_embedding.weight"].cpu().numpy(), nx["ngcf_grad_item"], rtol=1e-3, atol=1e-8)
        np.testing.assert_allclose(got["weight_dict.W_gcn_0"].cpu().numpy(), nx["ngcf_grad_W_gcn_0"], rtol=1e-3, atol=1e-7)
        np.testing.assert_allclose(got["weight_dict.b_bi_2"].cpu().numpy(), nx["ngcf_grad_b_bi_2"], rtol=1e-3, atol=1e-7)
    # after the update: the same step size everywhere the gradient is above rounding level
    p1, p2 = dict(m1.named_parameters()), dict(m2.named_parameters())
    for k in p1:
        big = grads1[k].abs() > 1e-3 * grads1[k].abs().max()
        assert torch.allclose(p2[k].detach()[big], p1[k].detach()[big], rtol=0, atol=0.05 * lr), k
        assert float((p2[k].detach() - p1[k].detach()).abs().max()) <= 2.1 * lr, k
    assert all(int(opt2.state[p]["step"]) == 1 for p in m2.parameters())
    # a second step through the trainer's protocol keeps the optimizer's state consistent
    assert m2.fused_train_step(bu, bp, bn, out, opt2) and all(int(opt2.state[p]["step"]) == 2 for p in m2.parameters())
    assert torch.isfinite(out).all()


def test_ngcf_node_dropout_trains(tmp_path, golden_small):
    """node_dropout = True (models/NGCF.py:56-79; the reference reads config['node_keep_prob'], a key its shipped
    config file does not have): a re-drawn edge mask per training forward, the plain graph in evaluation."""
    import utility.utility_function.tools as tools
    from idgrec_amd import ops
    from models.NGCF import NGCF

    g = golden_small
    cfg = _cfg("NGCF", node_dropout=True)
    data = _data_with(tmp_path, g, cfg)
    with pytest.raises(KeyError):
        NGCF(cfg, data, torch.device("cuda"))
    cfg["node_keep_prob"] = "0.2"  # survivors: probability 0.8 (the reference's rule), scaled by 1 / 0.8
    tools.set_seed(2024)
    model = NGCF(cfg, data, torch.device("cuda")).to("cuda")
    batch = torch.from_numpy(g["d64_batch"]).cuda()
    opt = ops.Adam(model.parameters(), lr=1e-3)
    model.train()
    with torch.no_grad():
        a1 = torch.cat(model.aggregate())
        a2 = torch.cat(model.aggregate())
    assert not torch.equal(a1[:, 64:], a2[:, 64:])  # a fresh mask (and fresh message dropout) per forward
    losses_seen = []
    for _ in range(12):
        ll = model(batch[:, 0], batch[:, 1], batch[:, 2])
        assert all(torch.isfinite(x) for x in ll)
        opt.zero_grad()
        sum(ll).backward()
        assert all(torch.isfinite(p.grad).all() for p in model.parameters())
        opt.step()
        losses_seen.append(float(sum(ll).detach()))
    assert np.mean(losses_seen[-3:]) < np.mean(losses_seen[:3])
    model.eval()
    r = model.get_rating_for_test(torch.arange(8, device="cuda"))
    assert torch.isfinite(r).all()


def test_sgl_three_views_vs_reference(tmp_path, golden_small, golden_next):
    import scipy.sparse as sp

    import utility.utility_function.tools as tools
    from models.SGL import SGL

    g, nx = golden_small, golden_next
    cfg = _cfg("SGL")
    data = _data_with(tmp_path, g, cfg)
    tools.set_seed(2024)
    m = SGL(cfg, data, torch.device("cuda")).to("cuda")
    n = data.num_users + data.num_items
    subs = [tools.convert_sp_mat_to_graph(sp.csr_matrix((nx[k + "_data"], nx[k + "_indices"], nx[k + "_indptr"]), shape=(n, n)),
                                          torch.device("cuda")) for k in ("sgl_sub1", "sgl_sub2")]
    b = torch.from_numpy(nx["batch"]).cuda()
    ll = m(b[:, 0], b[:, 1], b[:, 2], subs[0], subs[1])
    np.testing.assert_allclose([x.item() for x in ll], nx["sgl_loss"], rtol=RTOL)
    sum(ll).backward()
    # gradients through three propagations + the in-batch softmax: elements near zero are differences of O(1e-3)
    # terms, so the bound is relative to the table's largest gradient
    for mine, ref in ((m.user_embedding.weight.grad, nx["sgl_grad_user"]), (m.item_embedding.weight.grad, nx["sgl_grad_item"])):
        np.testing.assert_allclose(mine.cpu().numpy(), ref, rtol=1e-3, atol=3e-4 * np.abs(ref).max())
    # per-layer graph lists ('rw') give the same encoder when every layer uses the same graph
    u1, i1 = m.aggregate(subs[0])
    u2, i2 = m.aggregate([subs[0]] * 3)
    assert torch.allclose(u1, u2, rtol=1e-6, atol=1e-8) and torch.allclose(i1, i2, rtol=1e-6, atol=1e-8)


def test_sgl_fused_step_equals_autograd_step(tmp_path, golden_small, golden_next):
    """SGL's fused trainer step (three row-restricted encoder passes, BPR, InfoNCE over the raw batch ids, three
    backward propagations into one gradient, Adam) against forward() under autograd + optimizer.step()."""
    import scipy.sparse as sp

    import utility.utility_function.tools as tools
    from idgrec_amd import ops
    from models.SGL import SGL

    g, nx = golden_small, golden_next
    cfg = _cfg("SGL")
    data = _data_with(tmp_path, g, cfg)
    n = data.num_users + data.num_items
    tri = torch.from_numpy(g["sample1"][:3 * 256]).cuda()
    bt = [tuple(tri[i * 256:(i + 1) * 256, c].contiguous() for c in range(3)) for i in range(3)]
    res = []
    for fused in (True, False):
        tools.set_seed(2024)
        m = SGL(cfg, data, torch.device("cuda")).to("cuda")
        m.keep_fused_grad = True  # (.grad is compared below)
        subs = [tools.convert_sp_mat_to_graph(sp.csr_matrix((nx[k + "_data"], nx[k + "_indices"], nx[k + "_indptr"]), shape=(n, n)),
                                              torch.device("cuda")) for k in ("sgl_sub1", "sgl_sub2")]
        opt = ops.Adam(m.parameters(), lr=0.001)
        loss = torch.zeros((3, 3), device="cuda")
        for i in range(3):
            if fused:
                assert m.fused_sgl_step(*bt[i], subs[0], subs[1], loss[i], opt)
            else:
                ll = m(*bt[i], subs[0], subs[1])
                loss[i] = torch.stack([x.detach() for x in ll])
                opt.zero_grad()
                sum(ll).backward()
                opt.step()
        res.append((loss.cpu().numpy(), m.user_embedding.weight.grad.cpu().numpy(), m._storage.cpu().numpy()))
    (l_f, g_f, w_f), (l_a, g_a, w_a) = res
    np.testing.assert_allclose(l_f, l_a, rtol=2e-5)
    np.testing.assert_allclose(g_f, g_a, rtol=1e-3, atol=3e-4 * np.abs(g_a).max())
    np.testing.assert_allclose(w_f, w_a, rtol=1e-4, atol=1e-6)


def test_sgl_device_views_equal_create_adj_mat(tmp_path, golden_small):
    """The trainer's per-epoch views — built on the device from the full graph's handle (idg_subgraph_values_f32 +
    idg_graph_revalued_copy) — against tools.create_adj_mat (tools.py:67-92) on the same draws of Python's `random`
    stream: the same matrix bit for bit (dropped interactions are explicit zeros), the same final stream state."""
    import random

    import utility.utility_function.tools as tools
    from models.SGL import Trainer

    g = golden_small
    cfg = _cfg("SGL", aug_type="ed", ssl_ratio=0.1)
    data = _data_with(tmp_path, g, cfg)
    tools.set_seed(2024)
    tr = Trainer(None, cfg, data, torch.device("cuda"), logging.getLogger("sgl_views"))
    tr.model.to("cuda")
    n = data.num_users + data.num_items
    eye = torch.eye(n, device="cuda")
    random.seed(99)
    state = random.getstate()
    want = [tools.create_adj_mat(data.user_item_net, "ed", 0.1).toarray() for _ in range(2)]
    end_state = random.getstate()
    random.setstate(state)
    views = tr._views()
    assert random.getstate() == end_state
    for v, w in zip(views, want):
        got = v.spmm_raw(eye).cpu().numpy()
        assert np.array_equal(got, w.astype(np.float32))
        assert v.symmetric and abs(float((w != 0).sum()) / float((tr.model.Graph.spmm_raw(eye) != 0).sum().item()) - 0.9) < 0.01
    # the views propagate (forward and backward) on the full graph's schedule
    X = torch.randn(n, 64, device="cuda")
    out = views[0].propagate_mean_raw(X, 3, True).cpu().numpy()
    import scipy.sparse as sp

    A = sp.csr_matrix(want[0].astype(np.float32))
    x = X.cpu().numpy()
    layers = [x]
    for _ in range(3):
        layers.append(A @ layers[-1])
    np.testing.assert_allclose(out, np.mean(np.stack(layers, 1), 1), rtol=1e-4, atol=1e-6)


def test_sgl_trainer_loop_runs(tmp_path, golden_small):
    import utility.utility_function.tools as tools
    from models.SGL import Trainer

    cfg = _cfg("SGL", training_epochs=2, batch_size=512, test_batch_size=64, top_K="[5, 10]")
    data = _data_with(tmp_path, golden_small, cfg)
    stream = io.StringIO()
    logger = logging.getLogger("sgl_loop")
    logger.setLevel(logging.INFO)
    logger.handlers = [logging.StreamHandler(stream)]
    tools.set_seed(2024)
    Trainer(None, cfg, data, torch.device("cuda"), logger).train()
    lines = stream.getvalue().splitlines()
    assert [ln.split("|")[0].strip() for ln in lines[:4]] == ["Epoch:    1", "Epoch:    1", "Epoch:    2", "Epoch:    2"]
    assert lines[0].count(" + ") == 2 and lines[-2] == "Model training process completed." and lines[-1].startswith("Best epoch:")


def test_xsimgcl_encoder_and_training(tmp_path, golden_small, golden_next):
    import utility.utility_function.tools as tools
    from idgrec_amd import ops
    from models.XSimGCL import XSimGCL

    g, nx = golden_small, golden_next
    cfg = _cfg("XSimGCL")
    data = _data_with(tmp_path, g, cfg)
    tools.set_seed(2024)
    m = XSimGCL(cfg, data, torch.device("cuda")).to("cuda")
    with torch.no_grad():
        u, i = m.aggregate(perturbed=False)
    np.testing.assert_allclose(u.cpu().numpy(), nx["xsimgcl_user"], rtol=1e-5, atol=1e-8)
    np.testing.assert_allclose(i.cpu().numpy(), nx["xsimgcl_item"], rtol=1e-5, atol=1e-8)
    out = m.aggregate(perturbed=True)
    assert len(out) == 4 and out[2].shape == u.shape
    b = torch.from_numpy(nx["batch"]).cuda()
    opt = ops.Adam(m.parameters(), lr=1e-3)
    vals = []
    for _ in range(5):
        ll = m(b[:, 0], b[:, 1], b[:, 2])
        assert len(ll) == 3 and all(torch.isfinite(x) for x in ll)
        opt.zero_grad()
        sum(ll).backward()
        opt.step()
        vals.append(float(sum(ll).detach()))
    assert vals[-1] < vals[0]


@pytest.mark.parametrize("keep_grad", [True, False])
@pytest.mark.parametrize("name", ["SimGCL", "XSimGCL"])
def test_ssl_step_with_epsilon_zero_vs_reference(name, keep_grad, tmp_path, golden_small, golden_next):
    """SimGCL / XSimGCL pinned to the REFERENCE itself beyond their clean encoders: with epsilon = 0 the
    reference's step is deterministic (the perturbation is noise * 0), so its forward() losses, the .grad of both tables
    through its InfoNCE (models/SimGCL.py:62-90, XSimGCL.py:69-95, losses.py:24-35) and the tables after three
    torch.optim.Adam steps are goldens (oracle/gen_golden_next.py: simgcl0_* / xsimgcl0_*).  Against them: forward() under
    autograd, and the fused trainer step — with the gradient panel written out and, as the trainer runs it, consumed by
    the Adam epilogue (keep_grad False: the XSimGCL Horner chain that folds the view's gradient into the penultimate
    product)."""
    import importlib

    import utility.utility_function.tools as tools
    from idgrec_amd import ops

    g, nx = golden_small, golden_next
    tag = name.lower() + "0"
    cfg = _cfg(name, epsilon=0.0)
    data = _data_with(tmp_path, g, cfg)
    Model = getattr(importlib.import_module("models." + name), name)
    tools.set_seed(2024)
    m = Model(cfg, data, torch.device("cuda")).to("cuda")
    b = torch.from_numpy(nx["batch"]).cuda()
    ll = m(b[:, 0], b[:, 1], b[:, 2])
    np.testing.assert_allclose([x.item() for x in ll], nx[tag + "_loss"], rtol=RTOL)
    sum(ll).backward()
    for mine, ref in ((m.user_embedding.weight.grad, nx[tag + "_grad_user"]), (m.item_embedding.weight.grad, nx[tag + "_grad_item"])):
        np.testing.assert_allclose(mine.cpu().numpy(), ref, rtol=1e-3, atol=3e-4 * np.abs(ref).max())
    # the fused step, three batches, against the reference's own Adam trajectory
    tri = torch.from_numpy(nx["eps0_batches"]).cuda()
    tools.set_seed(2024)
    ops.reset_noise_stream()
    m = Model(cfg, data, torch.device("cuda")).to("cuda")
    m.keep_fused_grad = keep_grad
    opt = ops.Adam(m.parameters(), lr=float(cfg["learn_rate"]))
    loss = torch.zeros((3, 3), device="cuda")
    for i in range(3):
        bt = tuple(tri[i * 256:(i + 1) * 256, c].contiguous() for c in range(3))
        assert m.fused_train_step(*bt, loss[i], opt)
        assert (m.user_embedding.weight.grad is not None) == keep_grad
    np.testing.assert_allclose(loss.cpu().numpy(), nx[tag + "_traj_loss"], rtol=RTOL)
    # Adam moves an element by lr.m / (sqrt(v) + 1e-8): where the gradient is of the order of its own rounding error (rows
    # far from the batch receive |g| ~ 1e-9 through the propagation) the quotient turns a last-place difference of g into a
    # visible fraction of lr.  So: 1e-4 relative on all but a handful of elements, and nowhere more than a tenth of one
    # step's reach (lr = 1e-3, three steps)
    for mine, ref in ((m.user_embedding.weight, nx[tag + "_traj_user"]), (m.item_embedding.weight, nx[tag + "_traj_item"])):
        mine = mine.detach().cpu().numpy()
        off = ~np.isclose(mine, ref, rtol=RTOL, atol=1e-6)
        assert off.mean() < 1e-3, off.mean()
        assert np.abs(mine - ref).max() < 1e-4, np.abs(mine - ref).max()


def test_training_is_bit_reproducible_run_to_run(tmp_path, golden_small):
    """Side stream, one-batch lookahead, sorted scatter: two identical trainings must end with identical
    bits (weights and every logged loss)."""
    import utility.utility_function.tools as tools
    import utility.utility_train.trainer as trainer
    from models.LightGCN import LightGCN

    outs = []
    for run in range(2):
        data, cfg = _dataset(tmp_path, golden_small, "small", learn_rate=0.001, **dict(BASE, training_epochs=4, interval=2))
        stream = io.StringIO()
        logger = logging.getLogger("repro_%d" % run)
        logger.setLevel(logging.INFO)
        logger.handlers = [logging.StreamHandler(stream)]
        tools.set_seed(2024)
        model = LightGCN(cfg, data, torch.device("cuda"))
        trainer.universal_trainer(model, None, cfg, data, torch.device("cuda"), logger)
        lines = [re.sub(r"Training time: [0-9.]+", "T", ln) for ln in stream.getvalue().splitlines()]
        outs.append((lines, model.user_embedding.weight.detach().clone(), model.item_embedding.weight.detach().clone()))
    assert outs[0][0] == outs[1][0]
    assert torch.equal(outs[0][1], outs[1][1]) and torch.equal(outs[0][2], outs[1][2])


@pytest.mark.parametrize("model_name", ["LightGCN", "MFBPR", "SimGCL"])
def test_main_py_entry_point_end_to_end(model_name, tmp_path, golden_small):
    """`python main.py --model=<M>` exactly as a user of the reference would run it: working directory with
    ./configure/<M>.txt, ./dataset/<name>/{train,test}.txt and ./log/; two epochs; the log file is written in the
    reference's layout and the losses are finite and decreasing."""
    import os
    import shutil
    import subprocess
    import sys

    root = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
    work = tmp_path / "run"
    work.mkdir()
    for name in ("main.py", "Parser.py", "idgrec_amd.py", "models", "utility", "id-grec_amd"):
        os.symlink(os.path.join(root, name), work / name)
    (work / "configure").mkdir()
    cfg = open(os.path.join(root, "configure", model_name + ".txt")).read().splitlines()
    over = {"training_epochs": "2", "interval": "1", "dataset": "small", "batch_size": "256", "test_batch_size": "64",
            "top_K": "[5, 10]"}
    lines = []
    for ln in cfg:
        key = ln.split("=")[0].strip()
        lines.append("%s = %s" % (key, over[key]) if key in over else ln)
    (work / "configure" / (model_name + ".txt")).write_text("\n".join(lines) + "\n")
    d = work / "dataset" / "small"
    d.mkdir(parents=True)
    (d / "train.txt").write_bytes(golden_small["train_txt"].tobytes())
    (d / "test.txt").write_bytes(golden_small["test_txt"].tobytes())
    (work / "log").mkdir()
    out = subprocess.run([sys.executable, "main.py", "--model=" + model_name], cwd=str(work), capture_output=True, text=True,
                         timeout=600, env=dict(os.environ, PYTHONPATH=""))
    assert out.returncode == 0, out.stdout[-2000:] + out.stderr[-2000:]
    assert "Model training process completed." in out.stdout
    log = (work / "log" / model_name / "small.log").read_text().splitlines()
    assert any("Run with %s on small" % model_name in ln for ln in log)
    losses = [float(ln.split("training loss: ")[1].split(" = ")[0]) for ln in log if "training loss: " in ln]
    assert len(losses) == 2 and all(np.isfinite(losses)) and losses[1] < losses[0]
    assert sum("Test recall" in ln for ln in log) == 2
    shutil.rmtree(work / "log")


@pytest.mark.parametrize("mode", ["parallel", "alternating"])
def test_egcf_vs_reference(mode, tmp_path):
    """EGCF (rectangular R / R^T operator pair) against the reference's encoder outputs, three loss
    terms, item gradient and rating rows (tests/golden/egcf_small.npz, oracle/gen_golden_egcf.py), both modes."""
    import os

    import utility.utility_function.tools as tools
    from models.EGCF import EGCF

    eg = dict(np.load(os.path.join(os.path.dirname(__file__), "golden", "egcf_small.npz")))
    cfg = dict(zip(eg["config_keys"].tolist(), eg["config_values"].tolist()))
    cfg["mode"] = mode
    data = _data_with(tmp_path, eg, cfg)  # the fixture carries its own dataset files
    tools.set_seed(2024)
    m = EGCF(cfg, data, torch.device("cuda")).to("cuda")
    assert np.array_equal(m.item_embedding.weight.detach().cpu().numpy(), eg[mode + "_init_item"])
    with torch.no_grad():
        u, i = m.aggregate()
    np.testing.assert_allclose(u.cpu().numpy(), eg[mode + "_user"], rtol=1e-5, atol=1e-7)
    np.testing.assert_allclose(i.cpu().numpy(), eg[mode + "_item"], rtol=1e-5, atol=1e-7)
    b = torch.from_numpy(eg["batch"]).cuda()
    ll = m(b[:, 0], b[:, 1], b[:, 2])
    np.testing.assert_allclose([x.item() for x in ll], eg[mode + "_loss"], rtol=RTOL)
    sum(ll).backward()
    ref = eg[mode + "_grad_item"]
    np.testing.assert_allclose(m.item_embedding.weight.grad.cpu().numpy(), ref, rtol=1e-3, atol=3e-4 * np.abs(ref).max())
    m.eval()
    users = torch.from_numpy(eg["rating_users"]).cuda()
    np.testing.assert_allclose(m.get_rating_for_test(users).cpu().numpy(), eg[mode + "_rating"], rtol=1e-5, atol=1e-6)
    top = m.topk_for_test(users, 10).cpu().numpy()
    dense = m.get_rating_for_test(users).cpu().numpy()
    for r, uid in enumerate(eg["rating_users"]):
        dense[r, data.all_positive[int(uid)]] = -1
    ok, msg = __import__("oracle.oracle", fromlist=["x"]).topk_is_valid(dense, top, 10, tol=2e-6)
    assert ok, msg


def test_egcf_fused_step_vs_reference(tmp_path):
    """The fused, autograd-free EGCF step (idgrec_amd/egcf.py: tanh and its derivative in the products' epilogues, BPR +
    the three raw-row InfoNCE terms through the library, Adam in the last epilogue) against the reference's own numbers
    for the `parallel` encoder (egcf_small.npz): encoder outputs, the three losses, d loss / d item table, and the table
    after the Adam step against torch.optim.Adam fed with the REFERENCE's gradient.  Then the model's fused_train_step
    under the trainer's optimizer: same losses, same table as the engine on its own."""
    import os

    import utility.utility_function.tools as tools
    from idgrec_amd import ops
    from idgrec_amd.egcf import EgcfEngine
    from models.EGCF import EGCF

    eg = dict(np.load(os.path.join(os.path.dirname(__file__), "golden", "egcf_small.npz")))
    cfg = dict(zip(eg["config_keys"].tolist(), eg["config_values"].tolist()))
    cfg["mode"] = "parallel"
    data = _data_with(tmp_path, eg, cfg)
    tools.set_seed(2024)
    m = EGCF(cfg, data, torch.device("cuda")).to("cuda")
    W0 = m.item_embedding.weight.detach().clone()
    U, I, d = data.num_users, data.num_items, W0.shape[1]
    eng = EgcfEngine(m.Graph, m.user_Graph, U, I, d, m.n_layers, W0, m.reg_lambda, m.ssl_lambda, m.temperature, lr=1e-3,
                     store_grad=True)
    u, i = eng.propagate()
    np.testing.assert_allclose(u.cpu().numpy(), eg["parallel_user"], rtol=1e-5, atol=1e-7)
    np.testing.assert_allclose(i.cpu().numpy(), eg["parallel_item"], rtol=1e-5, atol=1e-7)
    full = eng.TOT.clone()
    b = torch.from_numpy(eg["batch"]).cuda()
    bu, bp, bn = b[:, 0].contiguous(), b[:, 1].contiguous(), b[:, 2].contiguous()
    eng.TOT.fill_(float("nan"))  # the step produces the batch's rows only, and reads no other
    loss = eng.train_step(bu, bp, bn).cpu().numpy()
    rows = torch.unique(torch.cat([bu, U + bp, U + bn]))
    assert torch.equal(eng.TOT[rows], full[rows]), "row-restricted last layer differs from the full one"
    np.testing.assert_allclose(loss, eg["parallel_loss"], rtol=RTOL)
    ref = eg["parallel_grad_item"]
    np.testing.assert_allclose(eng.grad_items().cpu().numpy(), ref, rtol=1e-3, atol=3e-4 * np.abs(ref).max())
    w = torch.nn.Parameter(W0.cpu().clone())
    opt = torch.optim.Adam([w], lr=1e-3)
    w.grad = torch.from_numpy(ref.copy())
    opt.step()
    moved = (w.detach() - W0.cpu()).abs() > 5e-4   # Adam's first step moves a weight by ~lr * sign(g): compare where g is not ~0
    got = eng.item_table().cpu()
    assert torch.allclose(got[moved], w.detach()[moved], rtol=0, atol=2e-5)
    assert float((got - w.detach()).abs().max()) <= 2.1e-3  # (a gradient of rounding size may flip sign: one lr step each way)

    # the model under the trainer's optimizer: the same chain
    tools.set_seed(2024)
    m2 = EGCF(cfg, data, torch.device("cuda")).to("cuda")
    assert m2.fused_step_available()
    opt2 = ops.Adam(list(m2.parameters()), lr=1e-3)
    out = torch.zeros(3, device="cuda")
    assert m2.fused_train_step(bu, bp, bn, out, opt2)
    np.testing.assert_allclose(out.cpu().numpy(), loss, rtol=1e-6)
    assert torch.equal(m2.item_embedding.weight.detach().cpu(), got) and int(opt2.state[m2.item_embedding.weight]["step"]) == 1
    m2.eval()
    users = torch.from_numpy(eg["rating_users"]).cuda()
    r_fused = m2.get_rating_for_test(users)
    m2._engine = None  # the same table through the differentiable operators
    m2._eval_cache = None
    np.testing.assert_allclose(r_fused.cpu().numpy(), m2.get_rating_for_test(users).cpu().numpy(), rtol=1e-5, atol=1e-6)


def test_egcf_alternating_fused_step_vs_reference(tmp_path):
    """The fused step of the `alternating` encoder (EgcfAltEngine: 2K rectangular products with tanh / tanh' epilogues, the
    last item product at the batch's item rows only, Adam in the last epilogue) against the reference's own numbers
    (egcf_small.npz): encoder outputs, the three losses, d loss / d item table, the table after one Adam step; then the
    model's fused_train_step under the trainer's optimizer gives the same losses and table."""
    import os

    import utility.utility_function.tools as tools
    from idgrec_amd import ops
    from idgrec_amd.egcf import EgcfAltEngine
    from models.EGCF import EGCF

    eg = dict(np.load(os.path.join(os.path.dirname(__file__), "golden", "egcf_small.npz")))
    cfg = dict(zip(eg["config_keys"].tolist(), eg["config_values"].tolist()))
    cfg["mode"] = "alternating"
    data = _data_with(tmp_path, eg, cfg)
    tools.set_seed(2024)
    m = EGCF(cfg, data, torch.device("cuda")).to("cuda")
    W0 = m.item_embedding.weight.detach().clone()
    np.testing.assert_array_equal(W0.cpu().numpy(), eg["alternating_init_item"])
    U, I, d = data.num_users, data.num_items, W0.shape[1]
    eng = EgcfAltEngine(m.user_Graph, U, I, d, m.n_layers, W0, m.reg_lambda, m.ssl_lambda, m.temperature, lr=1e-3, store_grad=True)
    u, i = eng.propagate()
    np.testing.assert_allclose(u.cpu().numpy(), eg["alternating_user"], rtol=1e-5, atol=1e-7)
    np.testing.assert_allclose(i.cpu().numpy(), eg["alternating_item"], rtol=1e-5, atol=1e-7)
    full = eng.TOT.clone()
    b = torch.from_numpy(eg["batch"]).cuda()
    bu, bp, bn = b[:, 0].contiguous(), b[:, 1].contiguous(), b[:, 2].contiguous()
    eng.TOT[U:].fill_(float("nan"))  # the step produces the item rows of the batch only, and reads no other
    loss = eng.train_step(bu, bp, bn).cpu().numpy()
    rows = torch.unique(torch.cat([bu, U + bp, U + bn]))
    assert torch.equal(eng.TOT[rows], full[rows]), "row-restricted last item layer differs from the full one"
    np.testing.assert_allclose(loss, eg["alternating_loss"], rtol=RTOL)
    ref = eg["alternating_grad_item"]
    np.testing.assert_allclose(eng.grad_items().cpu().numpy(), ref, rtol=1e-3, atol=3e-4 * np.abs(ref).max())
    w = torch.nn.Parameter(W0.cpu().clone())
    opt = torch.optim.Adam([w], lr=1e-3)
    w.grad = torch.from_numpy(ref.copy())
    opt.step()
    moved = (w.detach() - W0.cpu()).abs() > 5e-4
    got = eng.item_table().cpu()
    assert torch.allclose(got[moved], w.detach()[moved], rtol=0, atol=2e-5)
    assert float((got - w.detach()).abs().max()) <= 2.1e-3
    # the model under the trainer's optimizer: the same chain
    tools.set_seed(2024)
    m2 = EGCF(cfg, data, torch.device("cuda")).to("cuda")
    assert m2.fused_step_available()
    opt2 = ops.Adam(list(m2.parameters()), lr=1e-3)
    out = torch.zeros(3, device="cuda")
    assert m2.fused_train_step(bu, bp, bn, out, opt2)
    np.testing.assert_allclose(out.cpu().numpy(), loss, rtol=1e-6)
    assert torch.equal(m2.item_embedding.weight.detach().cpu(), got) and int(opt2.state[m2.item_embedding.weight]["step"]) == 1
    m2.eval()
    users = torch.from_numpy(eg["rating_users"]).cuda()
    r_fused = m2.get_rating_for_test(users)
    m2._engine = None  # the same table through the differentiable operators
    m2._eval_cache = None
    np.testing.assert_allclose(r_fused.cpu().numpy(), m2.get_rating_for_test(users).cpu().numpy(), rtol=1e-5, atol=1e-6)


@pytest.mark.parametrize("mode", ["parallel", "alternating"])
def test_egcf_trainer_loop_runs(mode, tmp_path, golden_small):
    import utility.utility_function.tools as tools
    from models.EGCF import Trainer

    cfg = _cfg("EGCF", training_epochs=2, batch_size=256, test_batch_size=64, top_K="[5, 10]", mode=mode)
    data = _data_with(tmp_path, golden_small, cfg)
    stream = io.StringIO()
    logger = logging.getLogger("egcf_loop")
    logger.setLevel(logging.INFO)
    logger.handlers = [logging.StreamHandler(stream)]
    tools.set_seed(2024)
    Trainer(None, cfg, data, torch.device("cuda"), logger).train()
    lines = stream.getvalue().splitlines()
    losses_ = [float(ln.split("training loss: ")[1].split(" = ")[0]) for ln in lines if "training loss: " in ln]
    assert len(losses_) == 2 and losses_[1] < losses_[0]


@pytest.mark.parametrize("mname", ["EGCF", "EGCF:alternating", "NGCF", "LightGCN", "SimGCL", "XSimGCL", "MFBPR"])
def test_batch_lookahead_changes_nothing(mname, tmp_path, golden_small):
    """The side-stream preparation of the NEXT batch (engine.BatchPrep / PropagationEngine._prepare: row bitmap, live units,
    scatter plan, the id lists of the step's InfoNCE calls — called by the trainer through prefetch_batch) is index-only
    work: four steps over four different batches with the lookahead equal, bit for bit, the same steps without it — losses
    and every parameter.  (With the lookahead the step's stream takes no event wait at all once the host can see the
    preparation complete; without it the batch is prepared inside the step and waited for: both orders of the two streams.)"""
    import importlib

    import utility.utility_function.tools as tools
    from idgrec_amd import ops

    mname, _, mode = mname.partition(":")
    cls = getattr(importlib.import_module("models." + mname), mname)
    cfg = _cfg(mname, **({"mode": mode} if mode else {}))
    data = _data_with(tmp_path, golden_small, cfg)
    gen = torch.Generator().manual_seed(7)
    B, steps = 192, 4
    epoch = torch.stack([torch.randint(0, data.num_users, (B * steps,), generator=gen),
                         torch.randint(0, data.num_items, (B * steps,), generator=gen),
                         torch.randint(0, data.num_items, (B * steps,), generator=gen)]).cuda()  # int64 ids, as the trainer's
    batches = [tuple(epoch[c, s * B:(s + 1) * B] for c in range(3)) for s in range(steps)]
    runs = []
    for ahead in (False, True):
        tools.set_seed(2024)
        m = cls(cfg, data, torch.device("cuda")).to("cuda")
        assert m.fused_step_available()
        opt = ops.Adam(list(m.parameters()), lr=float(cfg["learn_rate"]))
        ops.reset_noise_stream()
        out = torch.zeros(steps, m.n_fused_losses, device="cuda")
        if ahead:
            m.prefetch_batch(*batches[0])
        for s in range(steps):
            if ahead and s + 1 < steps:
                m.prefetch_batch(*batches[s + 1])
            assert m.fused_train_step(*batches[s], out[s], opt)
        torch.cuda.synchronize()
        runs.append((out.cpu(), {k: v.detach().cpu().clone() for k, v in m.named_parameters()}))
    assert torch.isfinite(runs[0][0]).all() and torch.equal(runs[0][0], runs[1][0])
    for k, v in runs[0][1].items():
        assert torch.equal(v, runs[1][1][k]), k


def test_sparsity_test_evaluation_path(tmp_path, golden_small, capsys):
    """sparsity_test = 1 (batch_test.py:110-170): the four interaction-count buckets are evaluated through the fused
    top-K path and printed in the reference's format; the first bucket is what general_test returns."""
    import utility.utility_function.tools as tools
    import utility.utility_train.batch_test as batch_test
    from models.LightGCN import LightGCN

    data, cfg = _dataset(tmp_path, golden_small, "small", learn_rate=0.001, **BASE)
    cfg["sparsity_test"] = "1"
    import utility.utility_data.data_loader as data_loader

    data = data_loader.Data(str(tmp_path / "small"), cfg)  # the split is built when the flag is set at load time
    tools.set_seed(2024)
    model = LightGCN(cfg, data, torch.device("cuda")).to("cuda")
    best = {'count': 0, 'epoch': 0, 'recall': [0., 0.], 'ndcg': [0., 0.], 'stop': 0}
    result, best = batch_test.general_test(data, model, torch.device("cuda"), cfg, 0, best)
    out = capsys.readouterr().out
    assert all(("level_%d: recall:" % lv) in out for lv in (1, 2, 3, 4))
    whole = batch_test.sparsity_test(data, model, torch.device("cuda"), cfg)
    assert len(whole) >= 4 and np.allclose(result["recall"], whole[0]["recall"])
    # every test user falls in exactly one bucket
    assert sorted(u for bucket in data.split_test_dict for u in bucket) == sorted(data.test_dict.keys())
