"""Debug probe: chunked form 3 vs the exact form for one call of 1024 users; where do the lists differ?"""
import os, sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__)))))
import numpy as np, torch
import idgrec_amd.ops as ops

Bt, I, d, k = int(sys.argv[1]), int(sys.argv[2]), 64, 20
seed = int(sys.argv[3]) if len(sys.argv) > 3 else 0
g = torch.Generator(device="cuda").manual_seed(seed)
Ue = torch.randn(Bt, d, device="cuda", generator=g) * 0.3
Ie = torch.randn(I, d, device="cuda", generator=g) * 0.3
users = torch.arange(Bt, device="cuda")
info = {}
got = ops.score_topk(Ue, Ie, users, k, None, None, return_values=True, info=info)
print(info)
ops.topk_option("collect", 0)
want = ops.score_topk(Ue, Ie, users, k, None, None, return_values=True)
bad = (got[0] != want[0]).any(dim=1).nonzero().flatten()
print("users differing:", len(bad), "of", Bt)
nc = info["chunks"]
ci = ((I + nc - 1) // nc + 255) // 256 * 256
missing = []
for u in bad.tolist():
    w = set(want[0][u].tolist()); gg = set(got[0][u].tolist())
    missing += list(w - gg)
    for it in (w - gg):
        print("  user %d (%%64=%d) item %d: chunk %d slab %d col %d rank %d" % (u, u % 64, it, it // ci, (it % ci) // 128, it % 128, want[0][u].tolist().index(it)))
missing = np.array(missing)
if len(missing):
    print("missing items:", len(missing))
    print(" chunk histogram:", np.bincount(missing // ci, minlength=nc).tolist())
    print(" slab-in-chunk histogram:", np.bincount((missing % ci) // 128).tolist())
    print(" position-in-slab / 32 histogram:", np.bincount((missing % 128) // 32, minlength=4).tolist())
    print(" user % 64 // 16 histogram:", np.bincount((bad.cpu().numpy() % 64) // 16, minlength=4).tolist())
