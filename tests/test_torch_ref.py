"""The CPU-baseline port (oracle/torch_ref.py) must BE the reference's step: the same op sequence on the same torch
build, so the same weights after six Adam steps on the golden trajectory — to fp32 reassociation (the port keeps ONE
[n, d] panel where the reference concatenates two tables, which moves a few additions: 3e-6 relative observed on
freshly drawn data), far inside the 1e-4 bar of SURVEY.md §8c."""
import numpy as np
import pytest
import torch

from oracle import oracle
from oracle.torch_ref import RefStep


@pytest.mark.parametrize("model", ["lgcn", "mf"])
def test_port_reproduces_reference_trajectory(model, golden_small):
    g = golden_small
    torch.set_num_threads(1)
    U, I = int(g["num_users"]), int(g["num_items"])
    pos = [g["pos_indices"][g["pos_indptr"][u]:g["pos_indptr"][u + 1]] for u in range(U)]
    np.random.seed(2024)
    s = oracle.sample_epoch(g["train_user"], g["train_item"], pos, I)
    s = s[oracle.shuffle_perm(len(s))]
    ref = RefStep(g["adj_indptr"], g["adj_indices"], g["adj_data"], U, I, g["d64_init_user"], g["d64_init_item"],
                  lr=1e-3 if model == "lgcn" else 1e-4, propagate=(model == "lgcn"))
    for step in range(6):
        b = torch.from_numpy(s[step * 128:(step + 1) * 128])
        vals = ref.step(b[:, 0], b[:, 1], b[:, 2])
        np.testing.assert_allclose(vals, g["traj_%s_losses" % model][step], rtol=1e-7)
        np.testing.assert_allclose(ref.user_w.detach().numpy(), g["traj_%s_user" % model][step], rtol=1e-5, atol=1e-9)
        np.testing.assert_allclose(ref.item_w.detach().numpy(), g["traj_%s_item" % model][step], rtol=1e-5, atol=1e-9)


def test_port_rating_matches_reference(golden_small):
    g = golden_small
    U, I = int(g["num_users"]), int(g["num_items"])
    ref = RefStep(g["adj_indptr"], g["adj_indices"], g["adj_data"], U, I, g["d64_init_user"], g["d64_init_item"])
    R = ref.rating(torch.from_numpy(g["test_dict_users"][:48])).numpy()
    np.testing.assert_allclose(R, g["d64_lgcn_rating"], rtol=1e-6, atol=1e-7)


def test_port_simgcl_pieces_match_reference(golden_small, golden_misc):
    """The SimGCL leg of the CPU baseline: the clean encoder pass (no layer 0 in the mean) and InfoNCE reproduce the
    imported reference's outputs bit for bit / to 1e-6; the perturbed passes draw from torch's CPU generator (nothing to
    pin them to), so the whole step is only run once and asked for three finite losses."""
    g = golden_small
    torch.set_num_threads(1)
    U, I = int(g["num_users"]), int(g["num_items"])
    ref = RefStep(g["adj_indptr"], g["adj_indices"], g["adj_data"], U, I, g["d64_init_user"], g["d64_init_item"],
                  simgcl=(0.1, 0.2, 0.5))
    with torch.no_grad():
        u, i = ref.aggregate(perturbed=False)
        np.testing.assert_array_equal(u.numpy(), g["d64_simgcl_user"])
        np.testing.assert_array_equal(i.numpy(), g["d64_simgcl_item"])
        from oracle.torch_ref import info_nce

        val = info_nce(torch.from_numpy(golden_misc["infonce_a"]), torch.from_numpy(golden_misc["infonce_b"]), 0.2)
        np.testing.assert_allclose(val.item(), golden_misc["infonce_02"], rtol=1e-6)
    b = torch.arange(64)
    losses = ref.step(b % U, b % I, (b * 7) % I)
    assert len(losses) == 3 and all(np.isfinite(losses))


def test_port_simgcl_step_with_epsilon_zero_is_the_references(golden_small):
    """With epsilon = 0 SimGCL's step is deterministic (the perturbation is noise * 0): the port's three losses and its
    tables after three Adam steps equal the imported reference's (oracle/gen_golden_next.py: simgcl0_traj_*) — which makes
    RefStep(simgcl=(0, tau, lambda)) a pinned checker for the fused SimGCL step at full size (tests/test_gpu_scale.py)."""
    import os

    g = golden_small
    nx = dict(np.load(os.path.join(os.path.dirname(__file__), "golden", "next_small.npz")))
    torch.set_num_threads(1)
    U, I = int(g["num_users"]), int(g["num_items"])
    # (the golden model was built under set_seed(2024): its initial tables are the d64_init_* of graph_small.npz)
    ref = RefStep(g["adj_indptr"], g["adj_indices"], g["adj_data"], U, I, g["d64_init_user"], g["d64_init_item"], lr=1e-3,
                  simgcl=(0.0, 0.2, 0.5))
    tri = torch.from_numpy(nx["eps0_batches"])
    for i in range(3):
        b = tri[i * 256:(i + 1) * 256]
        vals = ref.step(b[:, 0], b[:, 1], b[:, 2])
        np.testing.assert_allclose(vals, nx["simgcl0_traj_loss"][i], rtol=1e-6)
    np.testing.assert_allclose(ref.user_w.detach().numpy(), nx["simgcl0_traj_user"], rtol=1e-5, atol=1e-8)
    np.testing.assert_allclose(ref.item_w.detach().numpy(), nx["simgcl0_traj_item"], rtol=1e-5, atol=1e-8)
