"""Debug probe: NaN / infinity user rows through form 3 (redo kernel) and through the exact form — where do they differ?"""
import os, sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__)))))
import numpy as np, torch
import idgrec_amd.ops as ops

U, I, d, k = 2048, 38048, 64, 20
rng = np.random.default_rng(3)
Ue = (rng.standard_normal((U, d)) * 0.3).astype(np.float32)
Ie = (rng.standard_normal((I, d)) * 0.3).astype(np.float32)
Ue[3, 5] = np.nan
Ue[70, :] = np.nan
Ue[200, 9] = np.inf
Ue[201, 0] = -np.inf
Ue[640] *= np.float32(1e30)
ue, ie = torch.from_numpy(Ue).cuda(), torch.from_numpy(Ie).cuda()
every = torch.arange(U, device="cuda")
for sig in (True, False):
    info = {}
    got = ops.score_topk(ue, ie, every, k, apply_sigmoid=sig, return_values=True, info=info)
    with ops.topk_options(collect=0):
        want = ops.score_topk(ue, ie, every, k, apply_sigmoid=sig, return_values=True)
    bad = (got[0] != want[0]).any(dim=1).nonzero().flatten().tolist()
    print("sigmoid", sig, info, "users differing:", bad)
    for u in bad[:6]:
        print(" user", u, "\n  got ", got[0][u].tolist(), got[1][u].tolist()[:4], "\n  want", want[0][u].tolist(), want[1][u].tolist()[:4])
    dense = ops.score_dense(ue, ie, torch.tensor([3, 70, 200], device="cuda"), apply_sigmoid=False)
    print(" dense raw bits user 3:", [hex(x) for x in dense[0, :6].view(torch.int32).tolist()], "user 200:", dense[2, :6].tolist())
