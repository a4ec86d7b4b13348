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
        G.golden_io.save_npz(os.path.join(G.OUT, "next_small.npz"), **out)
        print("wrote next_small.npz (%d arrays)" % len(out))
    finally:
        shutil.rmtree(tmp, ignore_errors=True)


if __name__ == "__main__":
    main()
