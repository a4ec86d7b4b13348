"""Micro-benchmark of idg_spmm_f32 on a BASELINE-shape graph (GPU box).  Variants are selected
through the library's tuning environment variables, read at graph creation:
    python scripts/spmm_bench.py [workload] [d]"""
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
U, I, E = S.SHAPES[wl]
users, items = S.generate(U, I, E, seed=0)
ip, ix, dv = H.build_norm_adj(U, I, users, items)
n, nnz = U + I, len(ix)
X = torch.randn(n, d, device="cuda") * 0.1
gather = 4 * (n + 1) + 8 * nnz + 4 * nnz * d + 4 * n * d
configs = []
VARS = [int(x) for x in os.environ.get("VARS", "5").split(",")]
SPLITS = [int(x) for x in os.environ.get("SPLITS", "128").split(",")]
CAPS = [int(x) for x in os.environ.get("CAPS", "512,1024").split(",")]
BANDS = [int(x) for x in os.environ.get("BANDS", "1,2,4,8").split(",")]
for var in VARS:
    for split in SPLITS:
        for cap in CAPS:
            for bands in BANDS:
                configs.append((var, split, cap, bands))
graphs = []
for var, split, cap, bands in configs:
    os.environ["IDG_SPMM_VARIANT"], os.environ["IDG_TILE_NNZ"], os.environ["IDG_XCD_BANDS"] = str(var), str(cap), str(bands)
    graphs.append(ops.Graph(ip, ix, dv, n, n, split_threshold=split))
ref = None
Y = torch.empty_like(X)
times = {c: [] for c in configs}
for rnd in range(5):  # interleaved rounds in ONE process
    for c, G in zip(configs, graphs):
        G.spmm_raw(X, out=Y)
        torch.cuda.synchronize()
        a, b = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
        a.record()
        for _ in range(50):
            G.spmm_raw(X, out=Y)
        b.record()
        torch.cuda.synchronize()
        times[c].append(a.elapsed_time(b) / 50 * 1e3)
print("%s d=%d n=%d nnz=%d gather=%.1f MB" % (wl, d, n, nnz, gather / 1e6))
for c, G in zip(configs, graphs):
    t = np.median(times[c])
    info = G.info()
    print("variant=%d split=%4d cap=%4d bands=%d tiles=%5d long=%4d segs=%5d : %7.1f us (min %6.1f)  %6.2f TB/s gather"
          % (c[0], c[1], c[2], c[3], info["n_tiles"], info["n_long_rows"], info["n_segments"], t, min(times[c]),
             gather / t / 1e6))
