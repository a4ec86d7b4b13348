"""Top-K in calls of 1024 users with and without the train-item exclusion lists: how much of a call is the consumers'
start-up (parking the exclusion cursors by binary search)?"""
import os, sys, time
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__)))))
import numpy as np, torch
import idgrec_amd.ops as ops, idgrec_amd.synth as S

U, I, E = S.SHAPES["yelp2018"]
users, items = S.generate(U, I, E, seed=0)
ptr = np.zeros(U + 1, dtype=np.int64); ptr[1:] = np.cumsum(np.bincount(users, minlength=U))
g = torch.Generator(device="cuda").manual_seed(0)
Ue = torch.randn(U, 64, device="cuda", generator=g) * 0.3
Ie = torch.randn(I, 64, device="cuda", generator=g) * 0.3
ip, ix = torch.from_numpy(ptr).cuda(), torch.from_numpy(items.astype(np.int32)).cuda()
batches = [torch.arange(s, min(s + 1024, U), device="cuda") for s in range(0, U, 1024)]
for name, a, b in (("with exclusion lists", ip, ix), ("without", None, None)):
    f = lambda: [ops.score_topk(Ue, Ie, bb, 20, a, b) for bb in batches]
    f(); torch.cuda.synchronize()
    t0 = time.perf_counter()
    for _ in range(5): f()
    torch.cuda.synchronize()
    print("%-22s %.2f ms per evaluation, %.1f us per call" % (name, (time.perf_counter() - t0) / 5 * 1e3, (time.perf_counter() - t0) / 5 / len(batches) * 1e6))
