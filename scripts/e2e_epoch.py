"""End-to-end epochs through the plugin surface on a BASELINE-shape synthetic dataset:
dataset files -> Data -> models.<Model>.Trainer(...).train(), exactly what main.py does.
    python scripts/e2e_epoch.py [LightGCN|MFBPR|SimGCL|XSimGCL|SGL|NGCF] [epochs] [shape]"""
import importlib, logging, os, sys, tempfile, time
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
os.chdir(ROOT)
import torch
import idgrec_amd.synth as S
import utility.utility_data.data_loader as data_loader
import utility.utility_function.tools as tools

model = sys.argv[1] if len(sys.argv) > 1 else "LightGCN"
epochs = int(sys.argv[2]) if len(sys.argv) > 2 else 3
shape = sys.argv[3] if len(sys.argv) > 3 else "yelp2018"
root = tempfile.mkdtemp(prefix="idg_e2e_")
t0 = time.time(); S.make_dataset(root, shape, n_test=8); t_gen = time.time() - t0
cfg = tools.read_configuration("./configure/%s.txt" % model, model)
for kv in os.environ.get("E2E_CONFIG", "").split(","):  # e.g. E2E_CONFIG=mode=alternating
    if "=" in kv:
        cfg[kv.split("=", 1)[0]] = kv.split("=", 1)[1]
cfg.update(dataset=shape, dataset_path=root + "/", training_epochs=str(epochs), interval="1")
logger = logging.getLogger("e2e"); logger.setLevel(logging.INFO); logger.addHandler(logging.StreamHandler(sys.stdout))
tools.set_seed(2024)
t0 = time.time(); data = data_loader.Data(cfg["dataset_path"] + cfg["dataset"], cfg); t_load = time.time() - t0
t0 = time.time(); trainer = importlib.import_module("models." + model).Trainer(None, cfg, data, torch.device("cuda"), logger); t_init = time.time() - t0
t0 = time.time(); trainer.train(); t_train = time.time() - t0
print("E2E %s on %s: dataset files %.1fs | Data() %.2fs | model+graph init %.2fs | %d epochs (train+test each) %.2fs"
      % (model, shape, t_gen, t_load, t_init, epochs, t_train))
