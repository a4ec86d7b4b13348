"""The one-call training step (idg_step_run_f32, id-grec_amd/csrc/idg_step.cpp) against the same chain issued call by call
from Python (PropagationEngine with the plan switched off): bit-identical weights, moments, losses and gradient over 50
steps — LightGCN and MFBPR, with and without the one-batch lookahead, a short last batch included."""
import numpy as np
import pytest

torch = pytest.importorskip("torch")
pytestmark = pytest.mark.gpu


def _setup(g, model, d=64, seed=0):
    import idgrec_amd.ops as ops

    U, I = int(g["num_users"]), int(g["num_items"])
    n = U + I
    rng = np.random.default_rng(seed)
    W0 = torch.from_numpy((rng.standard_normal((n, d)) * 0.1).astype(np.float32)).cuda()
    graph = ops.Graph(g["adj_indptr"], g["adj_indices"], g["adj_data"], n, n) if model == "lgcn" else None
    return U, I, W0, graph


def _run(g, model, plan, lookahead, store_grad, steps=50, B=96, K=3):
    from idgrec_amd.engine import PropagationEngine

    U, I, W0, graph = _setup(g, model)
    eng = PropagationEngine(graph, U, I, W0.shape[1], K, include_layer0=True, reg_lambda=1e-4, lr=1e-3, params=W0.clone())
    eng._plan_on = plan
    eng.store_grad = store_grad
    tri = torch.from_numpy(np.concatenate([g["sample1"], g["sample2"]])[: steps * B - 37]).cuda()  # the last batch is short
    cols = [tri[:, c].contiguous() for c in range(3)]
    losses = torch.zeros((steps, 2), device="cuda")
    batch = lambda i: tuple(c[i * B:(i + 1) * B] for c in cols)  # noqa: E731
    for i in range(steps):
        if lookahead and i + 1 < steps and i % 5 != 3:  # (every fifth batch is NOT announced: prepared inside its own step)
            eng.prefetch(*batch(i + 1))
        eng.train_step(*batch(i), loss_out=losses[i])
    torch.cuda.synchronize()
    assert (eng._plan is not None) == plan
    return (eng.params.clone(), eng.exp_avg.clone(), eng.exp_avg_sq.clone(), losses.clone(),
            eng.grad.clone() if store_grad else None, eng.touched.clone() if graph is not None else None)


@pytest.mark.parametrize("model", ["lgcn", "mf"])
@pytest.mark.parametrize("lookahead", [True, False])
@pytest.mark.parametrize("store_grad", [True, False])
def test_one_call_step_is_the_call_by_call_chain(model, lookahead, store_grad, golden_small):
    a = _run(golden_small, model, True, lookahead, store_grad)
    b = _run(golden_small, model, False, lookahead, store_grad)
    for x, y, name in zip(a[:4], b[:4], ("params", "exp_avg", "exp_avg_sq", "losses")):
        assert torch.equal(x, y), name
    if store_grad:
        assert torch.equal(a[4], b[4]), "grad"
    if model == "lgcn":
        assert torch.equal(a[5], b[5]), "touched bitmap of the last step"
    assert float(a[3][-1].sum()) < float(a[3][0].sum())


def test_one_call_step_checks_its_arguments(golden_small):
    from idgrec_amd.engine import PropagationEngine

    U, I, W0, graph = _setup(golden_small, "lgcn")
    eng = PropagationEngine(graph, U, I, 64, 3, params=W0.clone())
    tri = torch.from_numpy(golden_small["sample1"][:64]).cuda()
    u, p, n = (tri[:, c].contiguous() for c in range(3))
    eng.train_step(u, p, n)
    with pytest.raises(TypeError):
        eng.train_step(tri[:, 0], p, n)  # a strided column
    with pytest.raises(TypeError):
        eng.train_step(u.int(), p, n)
    # a larger batch than the plan was built for: the plan is rebuilt, not overrun
    tri2 = torch.from_numpy(golden_small["sample1"][:200]).cuda()
    eng.train_step(*(tri2[:, c].contiguous() for c in range(3)))
    assert eng._plan.B_cap >= 200
    torch.cuda.synchronize()
    assert torch.isfinite(eng.params).all()


def test_a_lookahead_nobody_came_for_is_dropped(golden_small):
    """A batch announced with prefetch() but never run must not be mistaken for a later batch at the same addresses: the
    plan honours a prepared slot in the NEXT step only.  Here the announced batch's id tensors are rewritten in place after
    another batch ran in between (what a recycled allocation looks like to the library); the step on them must use the
    new ids — the same bits as the call-by-call chain run on the three batches without any lookahead."""
    from idgrec_amd.engine import PropagationEngine

    tri = torch.from_numpy(golden_small["sample1"][:64 * 4]).cuda()
    bt = [tuple(tri[i * 64:(i + 1) * 64, c].contiguous() for c in range(3)) for i in range(4)]
    res = []
    for plan in (True, False):
        U, I, W0, graph = _setup(golden_small, "lgcn")
        eng = PropagationEngine(graph, U, I, 64, 3, params=W0.clone())
        eng._plan_on = plan
        if plan:
            scratch = tuple(t.clone() for t in bt[1])
            eng.prefetch(*scratch)
            eng.train_step(*bt[0])      # prepares `scratch` ahead ...
            eng.train_step(*bt[2])      # ... but another batch runs
            for t, src in zip(scratch, bt[3]):
                t.copy_(src)            # the announced batch's storage now holds other ids
            eng.train_step(*scratch)
        else:
            for i in (0, 2, 3):
                eng.train_step(*bt[i])
        torch.cuda.synchronize()
        res.append((eng.params.clone(), eng.exp_avg_sq.clone(), eng.loss.clone()))
    for x, y in zip(*res):
        assert torch.equal(x, y)


def test_a_lookahead_in_another_storage_waits_for_its_ids(golden_small):
    """the announced batch may live in a storage the caller's stream has only just written (the first batch of
    a new epoch, a clone, a gather result) while the batch being run lives elsewhere.  The side stream must be ordered
    behind the caller's stream for THAT storage (idg_step_run_f32's next_ids_token) — with the slots warm, so that no
    first-use fork hides a missing one.  The announced tensors first hold another batch's (valid) ids and receive the
    real ones behind a long kernel on the main stream: a side stream that does not wait plans the wrong batch and the
    weights differ from the call-by-call chain's."""
    from idgrec_amd.engine import PropagationEngine

    B, steps = 64, 8
    tri = torch.from_numpy(golden_small["sample1"][:B * (steps + 1)]).cuda()
    bt = [tuple(tri[i * B:(i + 1) * B, c].contiguous() for c in range(3)) for i in range(steps + 1)]
    ballast = torch.randn(4096, 4096, device="cuda")
    res = []
    for plan in (True, False):
        U, I, W0, graph = _setup(golden_small, "lgcn")
        eng = PropagationEngine(graph, U, I, 64, 3, params=W0.clone())
        eng._plan_on = plan
        for i in range(steps):
            if plan and i >= 3:
                nxt = tuple(t.clone() for t in bt[0])   # a fresh storage holding some OTHER batch's ids ...
                torch.cuda.synchronize()
                for _ in range(6):
                    ballast @ ballast                   # ... the main stream is busy for a few milliseconds ...
                for t, src in zip(nxt, bt[i + 1]):
                    t.copy_(src)                        # ... and only then produces the announced ids
                eng.prefetch(*nxt)
                eng.train_step(*(bt[i] if i == 3 else cur))
                cur = nxt
            else:
                eng.train_step(*bt[i])
        torch.cuda.synchronize()
        assert (eng._plan is not None) == plan
        res.append((eng.params.clone(), eng.exp_avg_sq.clone(), eng.loss.clone()))
    for x, y in zip(*res):
        assert torch.equal(x, y)


def test_a_plan_is_closed_behind_its_side_stream(golden_small):
    """replacing a plan (a larger batch) releases its slot buffers; a lookahead the old plan put on the side
    stream must have finished by then (idg_step_synchronize drains the preparations too).  The replaced plan's announced
    batch is never run; the steps after the switch equal the call-by-call chain's."""
    from idgrec_amd.engine import PropagationEngine

    tri = torch.from_numpy(golden_small["sample1"][:1024]).cuda()
    small = [tuple(tri[i * 64:(i + 1) * 64, c].contiguous() for c in range(3)) for i in range(4)]
    big = tuple(tri[256:256 + 512, c].contiguous() for c in range(3))
    res = []
    for plan in (True, False):
        U, I, W0, graph = _setup(golden_small, "lgcn")
        eng = PropagationEngine(graph, U, I, 64, 3, params=W0.clone())
        eng._plan_on = plan
        eng.train_step(*small[0])
        eng.prefetch(*small[2])
        eng.train_step(*small[1])
        eng.train_step(*big)        # B > B_cap: the plan is closed with small[2]'s preparation possibly in flight
        eng.train_step(*small[3])
        torch.cuda.synchronize()
        res.append((eng.params.clone(), eng.exp_avg_sq.clone(), eng.loss.clone()))
    for x, y in zip(*res):
        assert torch.equal(x, y)


def test_mfbpr_width_beyond_the_plan_keeps_the_call_by_call_chain():
    """idg_step_create takes widths up to 1024 without a graph; a wider table must not be sent there."""
    from idgrec_amd.engine import PropagationEngine

    eng = PropagationEngine(None, 50, 40, 1028, 0, params=torch.randn(90, 1028, device="cuda") * 0.1)
    assert not eng._plan_eligible(16)
    eng2 = PropagationEngine(None, 50, 40, 1024, 0, params=torch.randn(90, 1024, device="cuda") * 0.1)
    assert eng2._plan_eligible(16)
    u = torch.randint(0, 50, (16,), device="cuda")
    p = torch.randint(0, 40, (16,), device="cuda")
    n = torch.randint(0, 40, (16,), device="cuda")
    for e in (eng, eng2):
        e.train_step(u, p, n)
        torch.cuda.synchronize()
        assert torch.isfinite(e.params).all()
    assert eng._plan is None and eng2._plan is not None
