"""IDG_NT_COLD experiment (GPU box): dense SpMM time with the H most gathered panel rows loaded with the default
cache policy and the rest non-temporally, for a sweep of H; bits must not change.
    python scripts/nt_probe.py [workload] [d] [H,H,...]"""
import os
import sys

import numpy as np
import torch

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
import idgrec_amd.host as H  # noqa: E402
import idgrec_amd.ops as ops  # noqa: E402
import idgrec_amd.synth as S  # noqa: E402

wl = sys.argv[1] if len(sys.argv) > 1 else "yelp2018"
d = int(sys.argv[2]) if len(sys.argv) > 2 else 64
hs = [int(x) for x in (sys.argv[3] if len(sys.argv) > 3 else "0,2000,4000,8000,12000,16000,24000,32000,48000").split(",")]
U, I, E = S.SHAPES[wl]
users, items = S.generate(U, I, E, seed=0)
ip, ix, dv = H.build_norm_adj(U, I, users, items)
n, nnz = U + I, len(ix)
X = torch.randn(n, d, device="cuda") * 0.1
gather = 4 * (n + 1) + 8 * nnz + 4 * nnz * d + 4 * n * d
graphs = []
for h in hs:
    if h:
        os.environ["IDG_NT_COLD"] = str(h)
    else:
        os.environ.pop("IDG_NT_COLD", None)
    graphs.append(ops.Graph(ip, ix, dv, n, n))
os.environ.pop("IDG_NT_COLD", None)
Y, Y2 = torch.empty_like(X), torch.empty_like(X)
ref = graphs[0].spmm_raw(X).clone()
times = {h: [] for h in hs}
reps = 50 if nnz < 1e7 else 8
for rnd in range(5):
    for h, G in zip(hs, graphs):
        G.spmm_raw(X, out=Y)
        assert torch.equal(Y, ref), "bits changed with H=%d" % h
        torch.cuda.synchronize()
        a, b = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
        a.record()
        for _ in range(reps):
            G.spmm_raw(X, out=Y)
            G.spmm_raw(Y, out=Y2)
        b.record()
        torch.cuda.synchronize()
        times[h].append(a.elapsed_time(b) / (2 * reps) * 1e3)
print("%s d=%d n=%d nnz=%d gather=%.1f MB" % (wl, d, n, nnz, gather / 1e6))
for h in hs:
    t = np.median(times[h])
    print("hot=%6d : %8.1f us (min %7.1f)  %6.2f TB/s gather" % (h, t, min(times[h]), gather / t / 1e6))
