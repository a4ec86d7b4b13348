"""oracle/gen_golden_convergence.py — TEST INFRASTRUCTURE.  Runs ONLY where /root/reference exists.

Recall@K / NDCG@K parity over a whole training run (BASELINE.json: "Recall@20 parity"): the imported
reference trains LightGCN-3 d=64 with its own universal_trainer on a "medium" synthetic dataset
(4000 x 3000, 120 k edges, 5 held-out items per user) for 40 epochs on CPU; the dataset files and
every logged test line (epochs 1, 6, 11, ...) become the fixture tests/golden/convergence_medium.npz.
tests/test_gpu_models.py trains the MI355X path on the same files with the same seed and compares
the curves.  The fixture is data (dataset text, numbers), never reference source.

    PYTHONDONTWRITEBYTECODE=1 python -B oracle/gen_golden_convergence.py
"""
import io
import logging
import os
import re
import sys
import tempfile
import time

sys.dont_write_bytecode = True
REF = os.environ.get("IDG_REFERENCE", "/root/reference")
HERE = os.path.dirname(os.path.abspath(__file__))
ROOT = os.path.dirname(HERE)
sys.path.insert(0, HERE)
import golden_io  # noqa: E402

sys.path.insert(0, REF)

import numpy as np  # noqa: E402
import torch  # noqa: E402
import importlib.util  # noqa: E402

_spec = importlib.util.spec_from_file_location("idg_synth", os.path.join(ROOT, "id-grec_amd", "synth.py"))
synth = importlib.util.module_from_spec(_spec)
_spec.loader.exec_module(synth)

import utility.utility_data.data_loader as ref_loader  # noqa: E402
import utility.utility_function.tools as ref_tools  # noqa: E402
import utility.utility_train.trainer as ref_trainer  # noqa: E402
from models.LightGCN import LightGCN as RefLightGCN  # noqa: E402

EPOCHS, INTERVAL = 40, 5


def main():
    torch.set_num_threads(1)  # ONE thread: the 8-thread run is reproducible only to ~5e-7 (its log lines are, its weights are not)
    tmp = tempfile.mkdtemp(prefix="idg_conv_")
    def draw(d):
        U, I, E = synth.SHAPES["medium"]
        users, items = synth.generate(U, I, E, seed=11)
        (tu, ti), (su, si) = synth.split_test(users, items, U, n_test=5, seed=12)
        synth.write_ratings(os.path.join(d, "train.txt"), tu, ti)
        synth.write_ratings(os.path.join(d, "test.txt"), su, si)

    # frozen input (tests/golden/inputs/medium_conv/): the generator runs only if it does not exist yet
    d = golden_io.frozen_dataset("medium_conv", os.path.join(tmp, "medium"), draw)
    cfg = ref_tools.read_configuration(os.path.join(REF, "configure", "LightGCN.txt"), "LightGCN")
    cfg.update(dataset="medium", dataset_path=tmp + "/", training_epochs=str(EPOCHS), interval=str(INTERVAL),
               early_stopping="1000", top_K="[10, 20]")
    stream = io.StringIO()
    logger = logging.getLogger("golden_convergence")
    logger.setLevel(logging.INFO)
    logger.handlers = [logging.StreamHandler(stream)]
    ref_tools.set_seed(2024)
    data = ref_loader.Data(cfg["dataset_path"] + cfg["dataset"], cfg)
    t0 = time.time()
    model = RefLightGCN(cfg, data, torch.device("cpu"))
    ref_trainer.universal_trainer(model, None, cfg, data, torch.device("cpu"), logger)
    print("reference run: %.0f s" % (time.time() - t0))
    lines = [re.sub(r"Training time: [0-9.]+", "Training time: T", ln) for ln in stream.getvalue().splitlines()]
    out = {
        "train_txt": np.frombuffer(open(os.path.join(d, "train.txt"), "rb").read(), dtype=np.uint8),
        "test_txt": np.frombuffer(open(os.path.join(d, "test.txt"), "rb").read(), dtype=np.uint8),
        # (the scratch directory's name is not part of the fixture)
        "config_keys": np.array(sorted(cfg)), "config_values": np.array([dict(cfg, dataset_path="<tmp>/")[k] for k in sorted(cfg)]),
        "log": np.array(lines),
        "final_user": model.user_embedding.weight.detach().numpy().copy(),
        "final_item": model.item_embedding.weight.detach().numpy().copy(),
    }
    path = os.path.join(golden_io.out_dir(), "convergence_medium.npz")
    golden_io.save_npz(path, **out)
    print("wrote", path, os.path.getsize(path), "bytes")
    for ln in lines:
        if "Test recall" in ln or "Best epoch" in ln:
            print(ln)


if __name__ == "__main__":
    main()
