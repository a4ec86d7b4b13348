"""The fused score + mask + top-K call on its own (for profiling): python scripts/topk_only.py [workload] [reps] [nomask]"""
import os
import sys
import time

import numpy as np
import torch

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import idgrec_amd.ops as ops  # noqa: E402
import idgrec_amd.synth as S  # noqa: E402

wl = sys.argv[1] if len(sys.argv) > 1 else "yelp2018"
reps = int(sys.argv[2]) if len(sys.argv) > 2 else 5
nomask = len(sys.argv) > 3 and sys.argv[3] == "nomask"
U, I, E = S.SHAPES[wl]
users, items = S.generate(U, I, E, seed=0)
pos_ptr = np.zeros(U + 1, dtype=np.int64)
pos_ptr[1:] = np.cumsum(np.bincount(users, minlength=U))
d, k = 64, 20
g = torch.Generator(device="cuda").manual_seed(0)
Ue = torch.randn(U, d, device="cuda", generator=g) * 0.3
Ie = torch.randn(I, d, device="cuda", generator=g) * 0.3
ip, ix = torch.from_numpy(pos_ptr).cuda(), torch.from_numpy(items.astype(np.int32)).cuda()
all_users = torch.arange(U, device="cuda")
args = (None, None) if nomask else (ip, ix)
info = {}
ops.score_topk(Ue, Ie, all_users, k, *args, info=info)
print("form / chunks / floor / users redone:", info)
torch.cuda.synchronize()
t0 = time.perf_counter()
for _ in range(reps):
    ops.score_topk(Ue, Ie, all_users, k, *args)
torch.cuda.synchronize()
dt = (time.perf_counter() - t0) / reps
print("%s %s: %.3f ms per full evaluation, %.1f TFLOP/s" % (wl, "no mask" if nomask else "masked", dt * 1e3, 2.0 * U * I * d / dt / 1e12))
