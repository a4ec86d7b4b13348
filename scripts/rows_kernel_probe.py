"""Where does the row-restricted product spend its time?  yelp2018 shape, d=64: an all-clear bitmap (every workgroup
leaves after the mask test), a 3072-row batch bitmap, every row flagged, and the dense kernel."""
import os
import sys

import numpy as np
import torch

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import idgrec_amd.host as H  # noqa: E402
import idgrec_amd.ops as ops  # noqa: E402
import idgrec_amd.synth as S  # noqa: E402

U, I, E = S.SHAPES["yelp2018"]
users, items = S.generate(U, I, E, seed=0)
ip, ix, dv = H.build_norm_adj(U, I, users, items)
n = U + I
g = ops.Graph(ip, ix, dv, n, n)
X = torch.randn(n, 64, device="cuda")
Y = torch.empty_like(X)
words = (n + 31) // 32
rng = np.random.default_rng(0)
e = rng.integers(0, len(users), 1024)
rows = np.unique(np.concatenate([users[e], U + items[e], U + rng.integers(0, I, 1024)]))


def bitmap(rs):
    b = np.zeros(words, dtype=np.uint32)
    np.bitwise_or.at(b, rs >> 5, (1 << (rs & 31)).astype(np.uint32))
    return torch.from_numpy(b.view(np.int32)).cuda()


def timeit(name, fn, reps=300):
    for _ in range(20):
        fn()
    a, b = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    torch.cuda.synchronize()
    a.record()
    for _ in range(reps):
        fn()
    b.record()
    torch.cuda.synchronize()
    print("%-44s %7.2f us" % (name, a.elapsed_time(b) / reps * 1e3))


clear, batch, full = bitmap(np.zeros(0, dtype=np.int64)), bitmap(rows), bitmap(np.arange(n))
timeit("dense", lambda: ops.spmm_ex_raw(g, X, Y=Y))
timeit("out_rows: no row flagged", lambda: ops.spmm_ex_raw(g, X, Y=Y, out_rows=clear))
timeit("out_rows: %d batch rows" % len(rows), lambda: ops.spmm_ex_raw(g, X, Y=Y, out_rows=batch))
ws = g.live_units(batch, len(rows))
torch.cuda.synchronize()
timeit("out_rows: %d batch rows, live-unit list" % len(rows), lambda: ops.spmm_ex_raw(g, X, Y=Y, out_rows=batch))
print("units in the list:", int(ws[0].item()))
g.forget_live_units(batch)
timeit("out_rows: every row", lambda: ops.spmm_ex_raw(g, X, Y=Y, out_rows=full))
timeit("x_rows: no live row", lambda: ops.spmm_ex_raw(g, X, Y=Y, x_rows=clear))
timeit("x_rows: %d batch rows" % len(rows), lambda: ops.spmm_ex_raw(g, X, Y=Y, x_rows=batch))
timeit("x_rows: every row", lambda: ops.spmm_ex_raw(g, X, Y=Y, x_rows=full))
