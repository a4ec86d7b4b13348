"""What bounds the dense SpMM on a cache-resident graph?  The same CSR structure (rows, row lengths, tiles) with the
COLUMNS folded into a small range, so that every gathered panel row is an L2 (or L1) hit: if the launch gets much
faster, the kernel is bound by where its rows come from (L2 misses served by the Infinity Cache); if not, by its own
issue / latency structure.    python scripts/l2_probe.py [workload] [d]"""
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
print("%s d=%d n=%d nnz=%d gather=%.1f MB" % (wl, d, n, nnz, gather / 1e6))
for fold in (0, 65536, 16384, 8192, 2048, 128):
    cols = ix if fold == 0 else (ix % fold).astype(np.int32)
    G = ops.Graph(ip, cols, dv, n, n)
    Y, Y2 = torch.empty_like(X), torch.empty_like(X)
    ts = []
    for rnd in range(5):
        G.spmm_raw(X, out=Y)
        torch.cuda.synchronize()
        a, b = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
        a.record()
        for _ in range(25):
            G.spmm_raw(X, out=Y)
            G.spmm_raw(Y, out=Y2)
        b.record()
        torch.cuda.synchronize()
        ts.append(a.elapsed_time(b) / 50 * 1e3)
    t = np.median(ts)
    print("columns folded into %6d rows (%7.2f MB gathered panel): %7.1f us  %6.2f TB/s gather-equivalent"
          % (fold or n, (fold or n) * d * 4 / 1e6, t, gather / t / 1e6))
