"""Data-parallel replicas (id-grec_amd/replicated.py): two ranks, each with its half of a global batch,
against ONE device processing the whole global batch — world_size 2 over gloo."""
import os
import sys

import numpy as np
import pytest

from oracle import oracle
from tests.ranks import run_ranks

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))


def _problem(g, K, include0, B, steps, world=2, seed=0):
    U, I = int(g["num_users"]), int(g["num_items"])
    W0 = np.concatenate([g["d64_init_user"], g["d64_init_item"]])
    rng = np.random.default_rng(seed)
    tri = g["sample1"][rng.permutation(len(g["sample1"]))][: world * B * steps]
    return dict(indptr=g["adj_indptr"], indices=g["adj_indices"], values=g["adj_data"], W0=W0, triples=tri, U=U, I=I,
                K=K, B=B, include0=include0)


def _single_device_reference(p, steps, world=2):
    """The same steps on one device with the global batch (world x B), by the oracle."""
    W = p["W0"].copy()
    m, v = np.zeros_like(W), np.zeros_like(W)
    adj = (p["indptr"], p["indices"], p["values"])
    gB = world * p["B"]
    losses = []
    for s in range(steps):
        b = p["triples"][s * gB:(s + 1) * gB]
        fin = oracle.propagate_mean(*adj, W, p["K"], p["include0"])
        loss, gf, ge = oracle.bpr(fin, W, p["U"], b[:, 0], b[:, 1], b[:, 2], 1e-4)
        grad = oracle.propagate_mean_bwd(*adj, gf, p["K"], p["include0"]) + ge
        oracle.adam(W, np.ascontiguousarray(grad), m, v, 1e-3, s + 1)
        losses.append(loss)
    return W, grad, np.stack(losses)


def _launch(mode, path, steps, world=2, exchange="grad"):
    run_ranks("replicated_worker.py", world, [mode, path, steps, exchange])
    return [dict(np.load(path + ".out%d.npz" % r)) for r in range(world)]


def _check(p, outs, steps):
    W, grad, losses = _single_device_reference(p, steps)
    for o in outs:
        # the regulariser term divides by the batch size: mean over ranks of per-rank means == the global-batch value
        np.testing.assert_allclose(o["losses"], losses, rtol=1e-4)
        np.testing.assert_allclose(o["G"], grad, rtol=1e-4, atol=2e-7)
        np.testing.assert_allclose(o["P"], W, rtol=1e-4, atol=2e-7)
    assert np.array_equal(outs[0]["P"], outs[1]["P"])  # replicas stay bit-identical


@pytest.mark.parametrize("exchange", ["grad", "rows"])
@pytest.mark.parametrize("K,include0", [(3, True), (2, False)])
def test_two_replicas_gloo_cpu_match_single_device(K, include0, exchange, tmp_path, golden_small):
    p = _problem(golden_small, K, include0, B=96, steps=3)
    path = str(tmp_path / "prob.npz")
    np.savez(path, **p)
    _check(p, _launch("cpu", path, 3, exchange=exchange), 3)


@pytest.mark.gpu
@pytest.mark.parametrize("exchange", ["grad", "rows"])
@pytest.mark.parametrize("K,include0", [(3, True), (2, False)])
def test_two_replicas_hip_engine_match_single_device(K, include0, exchange, tmp_path, golden_small):
    p = _problem(golden_small, K, include0, B=96, steps=4)
    path = str(tmp_path / "prob.npz")
    np.savez(path, **p)
    _check(p, _launch("gpu", path, 4, exchange=exchange), 4)


@pytest.mark.gpu
def test_row_exchange_world_1_is_the_plain_fused_step(golden_small):
    """One rank: packing the gradient rows into a message and merging that one message back must not change a bit
    of the fused step (same stored rows, same regulariser arithmetic, Adam in the same epilogue)."""
    import torch

    import idgrec_amd.ops as ops
    import idgrec_amd.replicated as rp
    from idgrec_amd.engine import PropagationEngine
    from idgrec_amd.sharded import NoComm

    p = _problem(golden_small, 3, True, B=128, steps=3, world=1)
    n = p["U"] + p["I"]
    g = ops.Graph(p["indptr"], p["indices"], p["values"], n, n)
    plain = PropagationEngine(g, p["U"], p["I"], 64, 3, params=torch.from_numpy(p["W0"].copy()).cuda())
    rep = rp.HipReplica(g, p["U"], p["I"], 64, 3, params=torch.from_numpy(p["W0"].copy()).cuda(), world=1)
    step = rp.RowExchangeStep(rep, NoComm(), 1)
    for s in range(3):
        b = torch.from_numpy(p["triples"][s * 128:(s + 1) * 128]).cuda()
        u, i, j = b[:, 0].contiguous(), b[:, 1].contiguous(), b[:, 2].contiguous()
        l0 = plain.train_step(u, i, j).clone()
        l1 = step.train_step(u, i, j).clone()
        assert torch.equal(l0, l1)
        assert torch.equal(plain.grad, rep.grad)
        assert torch.equal(plain.params, rep.params)



class _World1:
    """torch.distributed's face for a single rank (NativeComm asks it for rank / world only)."""

    @staticmethod
    def get_rank():
        return 0

    @staticmethod
    def get_world_size():
        return 1

    @staticmethod
    def get_backend():
        return "nccl"


@pytest.mark.gpu
@pytest.mark.parametrize("form", ["rows", "grad"])
def test_native_rccl_communicator_world_1(form, golden_small):
    """libidgrec's own RCCL communicator (idg_comm_*): set-up, self-test, and a replica step through it — collectives
    on the step's stream — against the plain fused step (bit-equal at world size 1)."""
    import torch

    import idgrec_amd.ops as ops
    import idgrec_amd.replicated as rp
    from idgrec_amd.engine import PropagationEngine
    from idgrec_amd.sharded import NativeComm

    torch.cuda.set_device(0)
    comm = NativeComm(_World1, 0)
    try:
        assert comm.self_test()
        x = torch.arange(4096, dtype=torch.float32, device="cuda")
        y = x.clone()
        assert comm.all_reduce_async(y, average=True) is None  # world size 1: the identity, nothing is enqueued
        out = torch.zeros_like(x)
        comm.all_gather_async(out, x)
        torch.cuda.synchronize()
        assert torch.equal(x, y) and torch.equal(x, out)
        with comm.through_rccl():  # ... unless asked to: the same calls as RCCL launches, as on a larger world
            comm.all_reduce_async(y, average=True)
            out.zero_()
            comm.all_gather_async(out, x)
            big = torch.arange(17 << 20, dtype=torch.float32, device="cuda")  # 68 MB: the second-stream form
            ref = big.clone()
            work = comm.all_reduce_async(big)
            assert work is not None
            comm.wait(work)
            rs = torch.arange(4096, dtype=torch.float32, device="cuda")
            comm.wait(comm.reduce_scatter_async(rs))
            comm.wait(comm.all_gather_async(rs, rs))
            torch.cuda.synchronize()
            assert torch.equal(big, ref) and torch.equal(x, y) and torch.equal(x, out) and torch.equal(rs, x)
        p = _problem(golden_small, 3, True, B=128, steps=3, world=1)
        n = p["U"] + p["I"]
        g = ops.Graph(p["indptr"], p["indices"], p["values"], n, n)
        plain = PropagationEngine(g, p["U"], p["I"], 64, 3, params=torch.from_numpy(p["W0"].copy()).cuda())
        plain.fuse_adam = form == "rows"
        rep = rp.HipReplica(g, p["U"], p["I"], 64, 3, params=torch.from_numpy(p["W0"].copy()).cuda(), world=1)
        step = (rp.RowExchangeStep if form == "rows" else rp.ReplicatedStep)(rep, comm, 1)
        for s in range(3):
            b = torch.from_numpy(p["triples"][s * 128:(s + 1) * 128]).cuda()
            u, i, j = b[:, 0].contiguous(), b[:, 1].contiguous(), b[:, 2].contiguous()
            l0 = plain.train_step(u, i, j).clone()
            l1 = step.train_step(u, i, j).clone()
            assert torch.equal(l0, l1)
            assert torch.equal(plain.params, rep.params)
    finally:
        comm.close()
