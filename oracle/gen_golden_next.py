"""oracle/gen_golden_next.py — TEST INFRASTRUCTURE.  Goldens for the "next" models of SURVEY §8(f)
(NGCF, SGL, XSimGCL) from the imported reference; same conventions as gen_golden.py.

    PYTHONDONTWRITEBYTECODE=1 python -B oracle/gen_golden_next.py
"""
import os
import random
import shutil
import sys
import tempfile

sys.dont_write_bytecode = True
HERE = os.path.dirname(os.path.abspath(__file__))
sys.path.insert(0, HERE)
import gen_golden as G  # noqa: E402  (puts the reference first on sys.path and imports it)

import numpy as np  # noqa: E402
import torch  # noqa: E402
from models.NGCF import NGCF as RefNGCF  # noqa: E402
from models.SGL import SGL as RefSGL  # noqa: E402
from models.SimGCL import SimGCL as RefSimGCL  # noqa: E402
from models.XSimGCL import XSimGCL as RefXSimGCL  # noqa: E402

ref_tools, ref_loader = G.ref_tools, G.ref_loader


def main():
    tmp = tempfile.mkdtemp(prefix="idg_golden_next_")
    try:
        gname = "small"
        path = G.make_data(tmp, gname)
        out = {}
        # message dropout off: the reference builds nn.Dropout(p) afresh inside aggregate(), so it stays active
        # under model.eval() too and would put RNG noise into the vectors
        cfg = G.base_config("NGCF", dataset=gname, dataset_path=tmp + "/", mess_drop_prob="[0.0, 0.0, 0.0]")
        ref_tools.set_seed(G.SEED)
        data = ref_loader.Data(path, cfg)
        np.random.seed(G.SEED)
        s1 = data.sample_data_to_train_all()
        B = 96
        bu, bp, bn = (torch.from_numpy(s1[:B, c].copy()) for c in range(3))
        bu[1], bp[1], bn[2] = bu[0], bp[0], bn[0]
        out["batch"] = torch.stack([bu, bp, bn], 1).numpy()

        # ---- NGCF (eval mode: message dropout inactive, so the numbers are deterministic)
        ref_tools.set_seed(G.SEED)
        m = RefNGCF(cfg, data, G.CPU)
        for k, v in m.weight_dict.items():
            out["ngcf_" + k] = v.detach().numpy().copy()
        out["ngcf_init_user"] = m.user_embedding.weight.detach().numpy().copy()
        m.eval()
        au, ai = m.aggregate()
        out["ngcf_user"], out["ngcf_item"] = au.detach().numpy().copy(), ai.detach().numpy().copy()
        m.zero_grad()
        ll = m(bu, bp, bn)
        sum(ll).backward()
        out["ngcf_loss"] = np.array([x.item() for x in ll])
        out["ngcf_grad_user"] = m.user_embedding.weight.grad.numpy().copy()
        out["ngcf_grad_item"] = m.item_embedding.weight.grad.numpy().copy()
        out["ngcf_grad_W_gcn_0"] = m.weight_dict["W_gcn_0"].grad.numpy().copy()
        out["ngcf_grad_b_bi_2"] = m.weight_dict["b_bi_2"].grad.numpy().copy()
        with torch.no_grad():
            out["ngcf_rating"] = m.get_rating_for_test(torch.from_numpy(np.array(list(data.test_dict.keys()))[:32])).numpy()

        # ---- SGL: edge-dropped adjacency under a fixed python `random` state, and the 3-view loss
        cfg_s = G.base_config("SGL", dataset=gname, dataset_path=tmp + "/")
        random.seed(7)
        A1 = ref_tools.create_adj_mat(data.user_item_net, "ed", 0.1)
        A2 = ref_tools.create_adj_mat(data.user_item_net, "ed", 0.1)
        for name, A in (("sgl_sub1", A1), ("sgl_sub2", A2)):
            A = A.tocsr()
            A.sort_indices()
            out[name + "_indptr"], out[name + "_indices"], out[name + "_data"] = A.indptr.astype(np.int64), A.indices.astype(np.int32), A.data
        ref_tools.set_seed(G.SEED)
        sgl = RefSGL(cfg_s, data, G.CPU)
        g1 = ref_tools.convert_sp_mat_to_sp_tensor(A1)
        g2 = ref_tools.convert_sp_mat_to_sp_tensor(A2)
        ll = sgl(bu, bp, bn, g1, g2)
        sum(ll).backward()
        out["sgl_loss"] = np.array([x.item() for x in ll])
        out["sgl_grad_user"] = sgl.user_embedding.weight.grad.numpy().copy()
        out["sgl_grad_item"] = sgl.item_embedding.weight.grad.numpy().copy()

        # ---- XSimGCL: the unperturbed encoder and the cl_layer view
        cfg_x = G.base_config("XSimGCL", dataset=gname, dataset_path=tmp + "/")
        ref_tools.set_seed(G.SEED)
        x = RefXSimGCL(cfg_x, data, G.CPU)
        with torch.no_grad():
            xu, xi = x.aggregate(perturbed=False)
        out["xsimgcl_user"], out["xsimgcl_item"] = xu.numpy().copy(), xi.numpy().copy()

        # ---- SimGCL / XSimGCL with epsilon = 0: the perturbation is sign(X) * normalize(noise) * 0, so the
        # step is deterministic although the reference draws its noise from the device RNG — forward() losses, .grad of
        # both tables (the InfoNCE gradient path of models/SimGCL.py:62-90 / XSimGCL.py:69-95 through autograd) and the
        # tables after three torch.optim.Adam steps on three batches (utility/utility_train/trainer.py:42-56)
        tri3 = torch.from_numpy(s1[:3 * 256].copy())
        out["eps0_batches"] = tri3.numpy()
        for tag, Ref, name in (("simgcl0", RefSimGCL, "SimGCL"), ("xsimgcl0", RefXSimGCL, "XSimGCL")):
            cfg_e = G.base_config(name, dataset=gname, dataset_path=tmp + "/", epsilon="0.0")
            ref_tools.set_seed(G.SEED)
            m = Ref(cfg_e, data, G.CPU)
            m.zero_grad()
            ll = m(bu, bp, bn)
            sum(ll).backward()
            out[tag + "_loss"] = np.array([x.item() for x in ll])
            out[tag + "_grad_user"] = m.user_embedding.weight.grad.numpy().copy()
            out[tag + "_grad_item"] = m.item_embedding.weight.grad.numpy().copy()
            opt = torch.optim.Adam(m.parameters(), lr=float(cfg_e["learn_rate"]))
            traj = []
            for i in range(3):
                b = tri3[i * 256:(i + 1) * 256]
                ll = m(b[:, 0], b[:, 1], b[:, 2])
                opt.zero_grad()
                sum(ll).backward()
                opt.step()
                traj.append([x.item() for x in ll])
            out[tag + "_traj_loss"] = np.array(traj)
            out[tag + "_traj_user"] = m.user_embedding.weight.detach().numpy().copy()
            out[tag + "_traj_item"] = m.item_embedding.weight.detach().numpy().copy()
        G.golden_io.save_npz(os.path.join(G.OUT, "next_small.npz"), **out)
        print("wrote next_small.npz (%d arrays)" % len(out))
    finally:
        shutil.rmtree(tmp, ignore_errors=True)


if __name__ == "__main__":
    main()
