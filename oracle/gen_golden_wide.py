"""oracle/gen_golden_wide.py — TEST INFRASTRUCTURE.  Runs ONLY where /root/reference exists.

The evaluator's golden on a WIDE catalogue: every earlier reference-generated Test() golden has 40-3,000
items, while this library's default top-K path for real data — form 3, threshold + collect on bf16 bounds — needs at
least 32,768 items and 512 users per call.  Here the REFERENCE (LightGCN, d = 64, its own configure/LightGCN.txt) trains
two epochs with its own universal_trainer on a frozen dataset of 1,100 users x 33,500 items and is then asked for

  * Test()'s result dict                       (utility/utility_train/batch_test.py:37-93)
  * the trained tables and aggregate()'s output (models/LightGCN.py:36-52)
  * per test user the 64 best (id, value) of the masked rating row — get_rating_for_test, rows of train items set to
    -1, torch.topk: the statements of batch_test.py:52-68 on the same model — whose first 20 are Test()'s own lists
  * eight whole rating rows

wide_small.npz holds those arrays (inputs + outputs: data); tests/test_gpu_topk_form3.py runs this repo's evaluator on
the same weights.  Same conventions as gen_golden.py (frozen inputs, deterministic .npz, IDG_GOLDEN_OUT).

    PYTHONDONTWRITEBYTECODE=1 python -B oracle/gen_golden_wide.py
"""
import io
import logging
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

ref_tools, ref_loader, ref_test, ref_trainer = G.ref_tools, G.ref_loader, G.ref_test, G.ref_trainer
NAME, SHAPE = "wide", (1100, 33500, 42000)
KEEP = 64        # best entries of every masked rating row that are stored
FULL_ROWS = 8    # whole rating rows stored


def make_data(tmp):
    def draw(d):
        U, I, E = SHAPE
        users, items = G.synth.generate(U, I, E, seed=21)
        (tu, ti), (su, si) = G.synth.split_test(users, items, U, n_test=2, seed=22)
        if ti.max() < I - 1 and si.max() < I - 1:  # the loader takes num_items = max id + 1 (data_loader.py:62-63)
            tu, ti = np.append(tu, 0), np.append(ti, I - 1)
        G.synth.write_ratings(os.path.join(d, "train.txt"), tu, ti)
        G.synth.write_ratings(os.path.join(d, "test.txt"), su, si)

    return G.golden_io.frozen_dataset(NAME, os.path.join(tmp, NAME), draw)


def main():
    tmp = tempfile.mkdtemp(prefix="idg_golden_wide_")
    try:
        path = make_data(tmp)
        cfg = dict(G.base_config("LightGCN", dataset=NAME, dataset_path=tmp + "/"), training_epochs="2")
        keys = sorted(k for k in cfg if k != "dataset_path")  # (a temporary directory: not part of the fixture)
        out = {"config_keys": np.array(keys), "config_values": np.array([cfg[k] for k in keys])}
        ref_tools.set_seed(G.SEED)
        data = ref_loader.Data(path, cfg)
        out["num_users"], out["num_items"] = data.num_users, data.num_items
        assert data.num_items >= 33000 and len(data.test_dict) >= 600, (data.num_items, len(data.test_dict))
        out["pos_indptr"] = data.user_item_net.indptr.astype(np.int64)
        out["pos_indices"] = data.user_item_net.indices.astype(np.int32)
        test_users = np.array(list(data.test_dict.keys()), dtype=np.int64)
        out["test_users"] = test_users
        out["test_indptr"] = np.concatenate([[0], np.cumsum([len(data.test_dict[u]) for u in test_users])]).astype(np.int64)
        out["test_items"] = np.concatenate([np.asarray(data.test_dict[u], dtype=np.int64) for u in test_users])

        stream = io.StringIO()
        logger = logging.getLogger("golden_wide")
        logger.setLevel(logging.INFO)
        logger.handlers = [logging.StreamHandler(stream)]
        ref_tools.set_seed(G.SEED)
        model = G.RefLightGCN(cfg, data, G.CPU)
        out["init_user"] = model.user_embedding.weight.detach().numpy().copy()[:64]   # (a corner: the init is seed-pinned elsewhere)
        ref_trainer.universal_trainer(model, None, cfg, data, G.CPU, logger)
        out["user_w"] = model.user_embedding.weight.detach().numpy().copy()
        out["item_w"] = model.item_embedding.weight.detach().numpy().copy()

        res = ref_test.Test(data, model, G.CPU, cfg)
        for key in ("recall", "precision", "ndcg"):
            out["test_" + key] = np.asarray(res[key], dtype=np.float64)
        out["top_K"] = np.array(eval(cfg["top_K"]), dtype=np.int64)

        model.eval()
        with torch.no_grad():
            fu, fi = model.aggregate()
            rows = np.concatenate([np.arange(0, data.num_users, 9), data.num_users + np.arange(0, data.num_items, 131)])
            out["final_rows_of"] = rows.astype(np.int64)
            out["final_rows"] = torch.cat([fu, fi])[torch.from_numpy(rows)].numpy().copy()
            # the statements of batch_test.py:52-68, batch by batch, on the same model
            top_i, top_v, full = [], [], {}
            keep_full = set(test_users[:: max(1, len(test_users) // FULL_ROWS)][:FULL_ROWS].tolist())
            bs = int(cfg["test_batch_size"])
            users = list(data.test_dict.keys())
            for lo in range(0, len(users), bs):
                batch_users = users[lo:lo + bs]
                all_positive = data.get_user_pos_items(batch_users)
                rating = model.get_rating_for_test(torch.Tensor(batch_users).long())
                exclude_users, exclude_items = [], []
                for i, items in enumerate(all_positive):
                    exclude_users.extend([i] * len(items))
                    exclude_items.extend(items)
                rating[exclude_users, exclude_items] = -1
                v, i = torch.topk(rating, k=KEEP)
                top_i.append(i.numpy().copy())
                top_v.append(v.numpy().copy())
                for j, u in enumerate(batch_users):
                    if u in keep_full:
                        full[u] = rating[j].numpy().copy()
        out["top64_idx"] = np.concatenate(top_i).astype(np.int64)
        out["top64_val"] = np.concatenate(top_v)
        out["rating_rows_of"] = np.array(sorted(full), dtype=np.int64)
        out["rating_rows"] = np.stack([full[u] for u in sorted(full)])
        G.golden_io.save_npz(os.path.join(G.OUT, "wide_small.npz"), **out)
        print("wrote wide_small.npz: %d users x %d items, %d test users, Test() = %s" % (data.num_users, data.num_items,
                                                                                        len(test_users), {k: res[k] for k in ("recall", "ndcg")}))
    finally:
        shutil.rmtree(tmp, ignore_errors=True)


if __name__ == "__main__":
    main()
