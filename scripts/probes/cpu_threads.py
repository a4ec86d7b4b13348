"""How does the CPU baseline (oracle/torch_ref.py) scale with torch threads on this host?"""
import os, sys, time
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import numpy as np, torch
import idgrec_amd.host as H, idgrec_amd.synth as S
from oracle.torch_ref import RefStep
U, I, E = S.SHAPES["yelp2018"]
users, items = S.generate(U, I, E, seed=0)
ip, ix, dv = H.build_norm_adj(U, I, users, items)
W0 = S.xavier_uniform_panel(U, I, 64, 2024).numpy()
rng = np.random.default_rng(0)
b = torch.from_numpy(np.stack([rng.integers(0, U, 1024), rng.integers(0, I, 1024), rng.integers(0, I, 1024)], 1))
print("cpu_count", os.cpu_count())
for t in (8, 16, 32, 64, 128, 256):
    torch.set_num_threads(t)
    ref = RefStep(ip, ix, dv, U, I, W0[:U], W0[U:])
    ref.step(b[:, 0], b[:, 1], b[:, 2])
    t0 = time.perf_counter(); ref.step(b[:, 0], b[:, 1], b[:, 2]); ref.step(b[:, 0], b[:, 1], b[:, 2]); dt = (time.perf_counter() - t0) / 2
    print("threads %3d: %.3f s/step -> %.0f triples/s" % (t, dt, 1024 / dt), flush=True)
