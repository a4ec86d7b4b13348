"""Full-rank evaluation throughput: idg_score_topk_f32 vs the reference's op sequence on stock
PyTorch-ROCm (matmul + sigmoid + index_put(-1) + topk), yelp2018-shape, k=20, 1024 users/batch."""
import os, sys, time
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import numpy as np, torch
import idgrec_amd.host as H, idgrec_amd.ops as ops, idgrec_amd.synth as S

wl = sys.argv[1] if len(sys.argv) > 1 else "yelp2018"
U, I, E = S.SHAPES[wl]
users, items = S.generate(U, I, E, seed=0)
pos_ptr = np.zeros(U + 1, dtype=np.int64); pos_ptr[1:] = np.cumsum(np.bincount(users, minlength=U))
d, k, Bt = int(os.environ.get("EVAL_D", "64")), 20, int(os.environ.get("EVAL_BT", "1024"))
g = torch.Generator(device="cuda").manual_seed(0)
Ue = torch.randn(U, d, device="cuda", generator=g) * 0.3
Ie = torch.randn(I, d, device="cuda", generator=g) * 0.3
ip, ix = torch.from_numpy(pos_ptr).cuda(), torch.from_numpy(items.astype(np.int32)).cuda()
batches = [torch.arange(s, min(s + Bt, U), device="cuda") for s in range(0, U, Bt)]
rows = [torch.repeat_interleave(torch.arange(len(b), device="cuda"), ip[b + 1] - ip[b]) for b in batches]
cols = [torch.cat([ix[ip[u]:ip[u + 1]] for u in b.tolist()]).long() if False else None for b in batches]

def mine():
    return [ops.score_topk(Ue, Ie, b, k, ip, ix) for b in batches]

all_users = torch.arange(U, device="cuda")
def mine_one_call():
    return [ops.score_topk(Ue, Ie, all_users, k, ip, ix)]

def stock():
    out = []
    for b, r in zip(batches, rows):
        rating = torch.sigmoid(Ue[b] @ Ie.t())
        lo, hi = int(ip[b[0]]), int(ip[b[-1] + 1])
        rating[r, ix[lo:hi].long()] = -1
        out.append(torch.topk(rating, k)[1])
    return out

for name, fn in (("idg_score_topk_f32", mine), ("idg_score_topk 1 call", mine_one_call), ("stock torch ops", stock)):
    fn(); torch.cuda.synchronize()
    t0 = time.perf_counter()
    for _ in range(3): fn()
    torch.cuda.synchronize()
    dt = (time.perf_counter() - t0) / 3
    print("%-20s %s: %7.2f ms per full evaluation of %d users (%.1f Musers*items/s, %.1f TFLOP/s)" % (name, wl, dt * 1e3, U, U * I / dt / 1e6, 2 * U * I * d / dt / 1e12))
a, b = mine(), stock()
same = sum(int((x == y).all(dim=1).sum()) for x, y in zip(a, b))
c = mine_one_call()[0]
print("one call == batched:", bool((torch.cat(a) == c).all()))
print("rows with identical top-%d lists: %d / %d" % (k, same, U))
