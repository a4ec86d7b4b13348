"""One fused score + mask + top-K call at a given geometry (random embeddings, a thin synthetic train list per user):
python scripts/topk_geometry.py USERS_PER_CALL ITEMS DIM [k] [reps]   e.g. 1024 5000000 256  (configs[4]'s evaluation call)"""
import os
import sys
import time

import numpy as np
import torch

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import idgrec_amd.ops as ops  # noqa: E402

Bt, I, d = int(sys.argv[1]), int(sys.argv[2]), int(sys.argv[3])
k = int(sys.argv[4]) if len(sys.argv) > 4 else 20
reps = int(sys.argv[5]) if len(sys.argv) > 5 else 5
g = torch.Generator(device="cuda").manual_seed(0)
Ue = torch.randn(Bt, d, device="cuda", generator=g) * 0.3
Ie = torch.randn(I, d, device="cuda", generator=g) * 0.3
rng = np.random.default_rng(0)
deg = 50  # train items per user (configs[4]: 5e8 edges / 1e7 users)
ix = np.sort(rng.integers(0, I, (Bt, deg)), axis=1)
ix += np.arange(deg)[None, :]  # strictly ascending within a user
ix = np.minimum(ix, I - 1 - (deg - 1 - np.arange(deg))[None, :])
ip = torch.arange(0, (Bt + 1) * deg, deg, dtype=torch.int64, device="cuda")
ixd = torch.from_numpy(ix.reshape(-1).astype(np.int32)).cuda()
users = torch.arange(Bt, device="cuda")
info = {}
ops.score_topk(Ue, Ie, users, k, ip, ixd, info=info)
torch.cuda.synchronize()
t0 = time.perf_counter()
for _ in range(reps):
    ops.score_topk(Ue, Ie, users, k, ip, ixd)
torch.cuda.synchronize()
dt = (time.perf_counter() - t0) / reps
print("%d users x %d items x d=%d, k=%d: %.3f ms per call, %.1f TFLOP/s, %s" % (Bt, I, d, k, dt * 1e3, 2.0 * Bt * I * d / dt / 1e12, info))
