"""oracle/gen_golden_egcf.py — TEST INFRASTRUCTURE.  Goldens for EGCF (the rectangular
R / R^T operator pair) from the imported reference, both aggregation modes; same conventions as gen_golden.py.

    PYTHONDONTWRITEBYTECODE=1 python -B oracle/gen_golden_egcf.py
"""
import os
import shutil
import sys
import tempfile

sys.dont_write_bytecode = True
HERE = os.path.dirname(os.path.abspath(__file__))
sys.path.insert(0, HERE)
import gen_golden as G  # noqa: E402  (puts the reference first on sys.path and imports it)

import numpy as np  # noqa: E402
import torch  # noqa: E402
from models.EGCF import EGCF as RefEGCF  # noqa: E402

ref_tools, ref_loader = G.ref_tools, G.ref_loader


def main():
    tmp = tempfile.mkdtemp(prefix="idg_golden_egcf_")
    try:
        gname = "small"
        path = G.make_data(tmp, gname, frozen="small_egcf")
        out = {}
        for mode in ("parallel", "alternating"):
            cfg = G.base_config("EGCF", dataset=gname, dataset_path=tmp + "/", mode=mode)
            ref_tools.set_seed(G.SEED)
            data = ref_loader.Data(path, cfg)
            np.random.seed(G.SEED)
            s1 = data.sample_data_to_train_all()
            B = 96
            bu, bp, bn = (torch.from_numpy(s1[:B, c].copy()) for c in range(3))
            bu[1], bp[1], bn[2] = bu[0], bp[0], bn[0]  # duplicates inside the batch
            out["batch"] = torch.stack([bu, bp, bn], 1).numpy()
            ref_tools.set_seed(G.SEED)
            m = RefEGCF(cfg, data, G.CPU)
            out[mode + "_init_item"] = m.item_embedding.weight.detach().numpy().copy()
            with torch.no_grad():
                au, ai = m.parallel_aggregate() if mode == "parallel" else m.alternating_aggregate()
            out[mode + "_user"], out[mode + "_item"] = au.numpy().copy(), ai.numpy().copy()
            m.zero_grad()
            ll = m(bu, bp, bn)
            sum(ll).backward()
            out[mode + "_loss"] = np.array([x.item() for x in ll])
            out[mode + "_grad_item"] = m.item_embedding.weight.grad.numpy().copy()
            with torch.no_grad():
                users = torch.from_numpy(np.array(list(data.test_dict.keys()))[:32])
                out[mode + "_rating"] = m.get_rating_for_test(users).numpy()
                out["rating_users"] = users.numpy()
        # the dataset itself (the synthetic generator may change; the fixture must not depend on it)
        out["train_txt"] = np.frombuffer(open(os.path.join(path, "train.txt"), "rb").read(), dtype=np.uint8)
        out["test_txt"] = np.frombuffer(open(os.path.join(path, "test.txt"), "rb").read(), dtype=np.uint8)
        cfg_keys = sorted(cfg)
        stored = dict(cfg, dataset_path="<tmp>/")  # (the scratch directory's name is not part of the fixture)
        out["config_keys"], out["config_values"] = np.array(cfg_keys), np.array([stored[k] for k in cfg_keys])
        G.golden_io.save_npz(os.path.join(G.OUT, "egcf_small.npz"), **out)
        print("wrote egcf_small.npz (%d arrays)" % len(out), {k: out[k] for k in out if k.endswith("_loss")})
    finally:
        shutil.rmtree(tmp, ignore_errors=True)


if __name__ == "__main__":
    main()
