"""Where does one full-rank evaluation (batch_test.Test) spend its time?  yelp2018-shape, LightGCN."""
import cProfile, importlib, logging, os, pstats, sys, tempfile, time
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT); os.chdir(ROOT)
import torch
import idgrec_amd.synth as S
import utility.utility_data.data_loader as data_loader
import utility.utility_function.tools as tools
import utility.utility_train.batch_test as batch_test
root = tempfile.mkdtemp(prefix="idg_tb_")
S.make_dataset(root, "yelp2018", n_test=8)
cfg = tools.read_configuration("./configure/LightGCN.txt", "LightGCN")
cfg.update(dataset="yelp2018", dataset_path=root + "/")
tools.set_seed(2024)
data = data_loader.Data(cfg["dataset_path"] + cfg["dataset"], cfg)
model = importlib.import_module("models.LightGCN").LightGCN(cfg, data, torch.device("cuda")).to("cuda")
for _ in range(2):
    t0 = time.time(); r = batch_test.Test(data, model, torch.device("cuda"), cfg); print("Test() %.1f ms" % ((time.time() - t0) * 1e3), r["recall"])
pr = cProfile.Profile(); pr.enable(); batch_test.Test(data, model, torch.device("cuda"), cfg); pr.disable()
pstats.Stats(pr).sort_stats("cumtime").print_stats(18)
