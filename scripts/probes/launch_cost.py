"""Host cost of one C-ABI launch through the Python binding (GPU box): python scripts/launch_cost.py"""
import os
import sys
import time

import numpy as np
import torch

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import idgrec_amd.host as H  # noqa: E402
import idgrec_amd.ops as ops  # noqa: E402
import idgrec_amd.synth as S  # noqa: E402
from idgrec_amd.native import lib  # noqa: E402

U, I, E = 2000, 1500, 40000
users, items = S.generate(U, I, E, seed=0)
ip, ix, dv = H.build_norm_adj(U, I, users, items)
n = U + I
g = ops.Graph(ip, ix, dv, n, n)
X = torch.randn(n, 64, device="cuda")
Y = torch.empty_like(X)
Z = torch.empty_like(X)


def bench(name, fn, reps=3000):
    for _ in range(50):
        fn()
    torch.cuda.synchronize()
    t0 = time.perf_counter()
    for _ in range(reps):
        fn()
    t1 = time.perf_counter()
    torch.cuda.synchronize()
    t2 = time.perf_counter()
    print("%-42s issue %6.2f us/call   (drain %6.2f us/call)" % (name, (t1 - t0) / reps * 1e6, (t2 - t0) / reps * 1e6))


bench("lincomb_raw", lambda: ops.lincomb_raw(Z, X, 0.5))
bench("spmm_ex_raw (Y only)", lambda: ops.spmm_ex_raw(g, X, Y=Y))
bench("spmm_ex_raw (Y, sum_in, sum_out)", lambda: ops.spmm_ex_raw(g, X, Y=Y, sum_in=X, sum_out=Z))
ws = g._workspace("spmm", 64)
st = ops._stream()
args = (g._h, X.data_ptr(), 64, Y.data_ptr(), None, None, None, 64, 1.0, 0, None, None, 64, ws.data_ptr(), st)
bench("lib.idg_spmm_ex_f32 (prebuilt args)", lambda: lib.idg_spmm_ex_f32(*args))
bench("ops._stream()", lambda: ops._stream(), 20000)
bench("g._workspace", lambda: g._workspace("spmm", 64), 20000)
bench("torch fill_", lambda: Z.fill_(0.0))
fin = torch.empty_like(X)
bench("propagate_mean_raw K=3", lambda: g.propagate_mean_raw(X, 3, True, out=fin), 1000)
