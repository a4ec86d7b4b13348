"""Stress the in-kernel combine of chunked rows (last-arriver protocol): the same product 3000 times, on two graphs
with hubs, every result compared bit for bit with the first and with the separate-fix-up form."""
import os, sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import numpy as np, torch
import idgrec_amd.host as H, idgrec_amd.ops as ops, idgrec_amd.synth as S
for wl, d in (("yelp2018", 64), ("amazon-book", 64), ("yelp2018", 256)):
    U, I, E = S.SHAPES[wl]
    users, items = S.generate(U, I, E, seed=0)
    ip, ix, dv = H.build_norm_adj(U, I, users, items)
    n = U + I
    X = torch.randn(n, d, device="cuda") * 0.1
    os.environ["IDG_FUSED_FIX"] = "0"
    ref = ops.Graph(ip, ix, dv, n, n).spmm_raw(X)
    os.environ["IDG_FUSED_FIX"] = "1"
    G = ops.Graph(ip, ix, dv, n, n)
    Y = torch.empty_like(X)
    bad = 0
    for it in range(3000):
        G.spmm_raw(X, out=Y)
        if it % 50 == 0 or it > 2950:
            bad += int(not torch.equal(Y, ref))
    torch.cuda.synchronize()
    print(wl, d, "chunked rows:", int((G.long_rows()[2] > 0).sum()), "mismatching checks:", bad)
