"""Worst case of top-K form 3: EVERY user's candidate list overflows (all-zero user table: every score ties) — what the
cooperative fallback (topk_redo_kernel) costs, and that it still returns the exact form's lists."""
import os, sys, time
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__)))))
import torch
import idgrec_amd.ops as ops

U, I, d, k = int(sys.argv[1]), int(sys.argv[2]), int(sys.argv[3]), 20
g = torch.Generator(device="cuda").manual_seed(0)
Ue = torch.zeros(U, d, device="cuda")
Ie = torch.randn(I, d, device="cuda", generator=g) * 0.3
every = torch.arange(U, device="cuda")
info = {}
got = ops.score_topk(Ue, Ie, every, k, return_values=True, info=info)
torch.cuda.synchronize()
t0 = time.perf_counter()
got = ops.score_topk(Ue, Ie, every, k, return_values=True)
torch.cuda.synchronize()
dt = time.perf_counter() - t0
os.environ["IDG_TOPK_COLLECT"] = "0"
want = ops.score_topk(Ue, Ie, every, k, return_values=True)
print("%d users x %d items x d=%d, all redone: %s, %.2f ms per call; equal to the exact form: %s" % (U, I, d, info, dt * 1e3, bool(torch.equal(got[0], want[0]) and torch.equal(got[1], want[1]))))
