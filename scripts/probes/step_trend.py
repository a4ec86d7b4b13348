"""Per-step GPU time of the first steps after bench.py's clock ramp (diagnostic; yelp2018 shape, LightGCN-3 d=64 B=1024):
python scripts/step_trend.py [ramp_seconds] [gemm|spmm|none]"""
import os, sys, time
import numpy as np, torch
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import idgrec_amd.host as H, idgrec_amd.ops as ops, idgrec_amd.synth as S  # noqa: E402
from idgrec_amd.engine import PropagationEngine  # noqa: E402

ramp = float(sys.argv[1]) if len(sys.argv) > 1 else 0.5
mode = sys.argv[2] if len(sys.argv) > 2 else "gemm"
U, I, E = S.SHAPES["yelp2018"]
users, items = S.generate(U, I, E, seed=0)
ip, ix, dv = H.build_norm_adj(U, I, users, items)
n, d, K, B, steps = U + I, 64, 3, 1024, 160
W0 = S.xavier_uniform_panel(U, I, d, 2024)
tri = torch.from_numpy(S.draw_triples(2024, users, items, U, I, steps * B)[0][: steps * B]).cuda()
tu, tp, tn = tri[:, 0].contiguous(), tri[:, 1].contiguous(), tri[:, 2].contiguous()
G = ops.Graph(ip, ix, dv, n, n)
eng = PropagationEngine(G, U, I, d, K, include_layer0=True, reg_lambda=1e-4, lr=1e-3, params=W0.cuda())
losses = torch.zeros((steps, 2), device="cuda")
time.sleep(2.0)  # the GPU idles, as it does during a bench's host-side graph construction
if mode == "gemm":
    S.ramp_clocks(ramp)
elif mode == "spmm":
    X = torch.randn(n, d, device="cuda"); Y = torch.empty_like(X)
    t0 = time.perf_counter()
    while time.perf_counter() - t0 < ramp:
        for _ in range(50):
            ops.spmm_ex_raw(G, X, Y=Y)
        torch.cuda.synchronize()
ev = [torch.cuda.Event(enable_timing=True) for _ in range(steps + 1)]
torch.cuda.synchronize()
ev[0].record()
for i in range(steps):
    s = slice(i * B, (i + 1) * B)
    if i + 1 < steps:
        s2 = slice((i + 1) * B, (i + 2) * B)
        eng.prefetch(tu[s2], tp[s2], tn[s2])
    eng.train_step(tu[s], tp[s], tn[s], loss_out=losses[i])
    ev[i + 1].record()
torch.cuda.synchronize()
t = np.array([ev[i].elapsed_time(ev[i + 1]) for i in range(steps)]) * 1e3
print("ramp %.2f s (%s): us per step, means of steps 0-4 / 5-24 / 25-49 / 50-99 / 100-159: %.1f %.1f %.1f %.1f %.1f" % (
    ramp, mode, t[:5].mean(), t[5:25].mean(), t[25:50].mean(), t[50:100].mean(), t[100:].mean()))
print("first 12:", " ".join("%.0f" % x for x in t[:12]))
