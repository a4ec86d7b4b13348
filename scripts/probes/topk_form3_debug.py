"""Debug probe: top-K form 3 vs the exact form on all-negative scores in raw mode."""
import os, sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__)))))
import numpy as np, torch
import idgrec_amd.ops as ops, idgrec_amd.synth as S

U, I, d, k = 31668, 38048, 64, 20
users, items = S.generate(U, I, 600000, seed=23)
ptr = np.zeros(U + 1, dtype=np.int64); ptr[1:] = np.cumsum(np.bincount(users, minlength=U))
ip, ix = torch.from_numpy(ptr).cuda(), torch.from_numpy(items.astype(np.int32)).cuda()
g = torch.Generator(device="cuda").manual_seed(20)
Ue = torch.randn(U, d, device="cuda", generator=g) * 0.3
Ie = torch.randn(I, d, device="cuda", generator=g) * 0.3
Un, In = -Ue.abs() - 0.1, Ie.abs() + 0.1
every = torch.arange(U, device="cuda")
info = {}
got = ops.score_topk(Un, In, every, k, ip, ix, apply_sigmoid=False, return_values=True, info=info)
print(info)
ops.topk_option("collect", 0)
want = ops.score_topk(Un, In, every, k, ip, ix, apply_sigmoid=False, return_values=True)
bad = (got[0] != want[0]).any(dim=1).nonzero().flatten()
print("users differing:", len(bad), bad[:10].tolist())
for u in bad[:2].tolist():
    tr = items[ptr[u]:ptr[u + 1]]
    print("user", u, "train items", len(tr), tr[:25].tolist())
    print(" got ", got[0][u].tolist(), [round(x, 3) for x in got[1][u].tolist()])
    print(" want", want[0][u].tolist(), [round(x, 3) for x in want[1][u].tolist()])
    R = (Un[u:u + 1] @ In.t()).flatten()
    R[torch.from_numpy(tr).cuda()] = -1
    top = torch.topk(R, k)
    print(" torch", top.indices.tolist(), [round(x, 3) for x in top.values.tolist()])
