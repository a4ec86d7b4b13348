"""Data-parallel replicas (id-grec_amd/replicated.py): two ranks, each with its half of a global batch,
against ONE device processing the whole global batch — world_size 2 over gloo."""
import os
import socket
import subprocess
import sys

import numpy as np
import pytest

from oracle import oracle

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))


def _free_port():
    with socket.socket() as s:
        s.bind(("127.0.0.1", 0))
        return s.getsockname()[1]


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


def _launch(mode, path, steps, world=2):
    port = _free_port()
    env = dict(os.environ, PYTHONPATH=ROOT, HSA_ENABLE_IPC_MODE_LEGACY="0")
    procs = [subprocess.Popen([sys.executable, os.path.join(ROOT, "tests", "replicated_worker.py"), str(r), str(world),
                               str(port), mode, path, str(steps)], env=env, stdout=subprocess.PIPE,
                              stderr=subprocess.STDOUT, text=True) for r in range(world)]
    outs = [p.communicate(timeout=600)[0] for p in procs]
    for p, o in zip(procs, outs):
        assert p.returncode == 0, o[-3000:]
    return [dict(np.load(path + ".out%d.npz" % r)) for r in range(world)]


def _check(p, outs, steps):
    W, grad, losses = _single_device_reference(p, steps)
    for o in outs:
        # the regulariser term divides by the batch size: mean over ranks of per-rank means == the global-batch value
        np.testing.assert_allclose(o["losses"], losses, rtol=1e-4)
        np.testing.assert_allclose(o["G"], grad, rtol=1e-4, atol=2e-7)
        np.testing.assert_allclose(o["P"], W, rtol=1e-4, atol=2e-7)
    assert np.array_equal(outs[0]["P"], outs[1]["P"])  # replicas stay bit-identical


@pytest.mark.parametrize("K,include0", [(3, True), (2, False)])
def test_two_replicas_gloo_cpu_match_single_device(K, include0, tmp_path, golden_small):
    p = _problem(golden_small, K, include0, B=96, steps=3)
    path = str(tmp_path / "prob.npz")
    np.savez(path, **p)
    _check(p, _launch("cpu", path, 3), 3)


@pytest.mark.gpu
@pytest.mark.parametrize("K,include0", [(3, True), (2, False)])
def test_two_replicas_hip_engine_match_single_device(K, include0, tmp_path, golden_small):
    p = _problem(golden_small, K, include0, B=96, steps=4)
    path = str(tmp_path / "prob.npz")
    np.savez(path, **p)
    _check(p, _launch("gpu", path, 4), 4)
