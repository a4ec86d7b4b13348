import io, logging, os, re, sys, tempfile
sys.path.insert(0, os.getcwd())
import numpy as np, torch
import utility.utility_data.data_loader as data_loader, utility.utility_function.tools as tools, utility.utility_train.trainer as trainer
from models.LightGCN import LightGCN
g = np.load("tests/golden/convergence_medium.npz")
tmp = tempfile.mkdtemp(); d = os.path.join(tmp, "medium"); os.mkdir(d)
open(os.path.join(d, "train.txt"), "wb").write(g["train_txt"].tobytes()); open(os.path.join(d, "test.txt"), "wb").write(g["test_txt"].tobytes())
cfg = dict(zip(g["config_keys"].tolist(), g["config_values"].tolist())); cfg.update(dataset="medium", dataset_path=tmp + "/")
stream = io.StringIO(); logger = logging.getLogger("x"); logger.setLevel(logging.INFO); logger.handlers = [logging.StreamHandler(stream)]
tools.set_seed(2024)
data = data_loader.Data(cfg["dataset_path"] + cfg["dataset"], cfg)
model = LightGCN(cfg, data, torch.device("cuda"))
sys.stdout = io.StringIO(); trainer.universal_trainer(model, None, cfg, data, torch.device("cuda"), logger); sys.stdout = sys.__stdout__
num = lambda l: np.array([float(x) for x in re.findall(r"[-+]?\d+\.?\d*(?:e[-+]?\d+)?", re.sub(r"Training time: [0-9.T]+", "", l))])
mx = 0; ml = 0
for a, b in zip(stream.getvalue().splitlines(), g["log"].tolist()):
    x, y = num(a), num(b)
    if "Test recall" in b:
        mx = max(mx, np.abs(x[1:] - y[1:]).max()); print(b.split("|")[0], "max |d metric| = %.2e" % np.abs(x[1:] - y[1:]).max())
    elif "training loss" in b:
        ml = max(ml, (np.abs(x[1:] - y[1:]) / np.abs(y[1:])).max())
print("max metric dev %.2e, max rel loss dev %.2e" % (mx, ml))
wu = model.user_embedding.weight.detach().cpu().numpy(); print("max |dW|/max|W| = %.2e" % (np.abs(wu - g["final_user"]).max() / np.abs(g["final_user"]).max()))
