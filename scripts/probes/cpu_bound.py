"""Is the fused step host-bound?  Compare host time to ENQUEUE N steps with the GPU time they take."""
import os, sys, time
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import numpy as np, torch
import idgrec_amd.host as H, idgrec_amd.ops as ops, idgrec_amd.synth as S
from idgrec_amd.engine import PropagationEngine
U, I, E = S.SHAPES["yelp2018"]
users, items = S.generate(U, I, E, seed=0)
ip, ix, dv = H.build_norm_adj(U, I, users, items)
n = U + I
G = ops.Graph(ip, ix, dv, n, n)
eng = PropagationEngine(G, U, I, 64, 3, params=S.xavier_uniform_panel(U, I, 64, 0).cuda())
B, N = 1024, 300
tu = torch.randint(0, U, (B * (N + 20),), device="cuda"); tp = torch.randint(0, I, (B * (N + 20),), device="cuda"); tn = torch.randint(0, I, (B * (N + 20),), device="cuda")
for i in range(20): eng.train_step(tu[i*B:(i+1)*B], tp[i*B:(i+1)*B], tn[i*B:(i+1)*B])
torch.cuda.synchronize()
t0 = time.perf_counter()
for i in range(20, 20 + N): eng.train_step(tu[i*B:(i+1)*B], tp[i*B:(i+1)*B], tn[i*B:(i+1)*B])
t_enq = time.perf_counter() - t0
torch.cuda.synchronize()
t_all = time.perf_counter() - t0
print("host enqueue %.1f us/step, wall %.1f us/step -> %s" % (t_enq / N * 1e6, t_all / N * 1e6, "HOST-BOUND" if t_enq > 0.9 * t_all else "gpu-bound"))
