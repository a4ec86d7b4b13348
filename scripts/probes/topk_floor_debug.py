"""Timing of one 1024-user top-K call in the three floor modes (events around single calls)."""
import os, sys
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
b = torch.arange(0, 1024, device="cuda")
for mode in ("1", "2", "0"):
    os.environ["IDG_TOPK_FLOOR"] = mode
    for _ in range(3): ops.score_topk(Ue, Ie, b, 20, ip, ix)
    a, c = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    a.record()
    for _ in range(20): ops.score_topk(Ue, Ie, b, 20, ip, ix)
    c.record(); torch.cuda.synchronize()
    print("IDG_TOPK_FLOOR=%s: %.1f us per call of 1024 users" % (mode, a.elapsed_time(c) / 20 * 1e3))
