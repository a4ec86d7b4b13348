"""Differential fuzz of the top-K forms: random geometries (users, items, width, k, score mode, train lists, duplicated rows,
zero rows, skewed norms) through form 3 and through the exact form — ids and values must agree bit for bit.
usage: python scripts/probes/topk_fuzz.py [cases] [seed]"""
import os, sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__)))))
import numpy as np, torch
import idgrec_amd.ops as ops

cases = int(sys.argv[1]) if len(sys.argv) > 1 else 40
rng = np.random.default_rng(int(sys.argv[2]) if len(sys.argv) > 2 else 0)
bad = 0
for c in range(cases):
    d = int(rng.choice([64, 128, 256]))
    Bt = int(rng.integers(512, 7000))
    I = int(rng.integers(32768, 140000))
    k = int(rng.integers(1, 43))
    sig = bool(rng.integers(0, 2))
    g = torch.Generator(device="cuda").manual_seed(int(rng.integers(1 << 30)))
    spread = float(rng.choice([0.0, 0.5, 1.0]))
    Ue = torch.randn(Bt, d, device="cuda", generator=g) * 0.3 * torch.exp(spread * torch.randn(Bt, 1, device="cuda", generator=g))
    Ie = torch.randn(I, d, device="cuda", generator=g) * 0.3 * torch.exp(spread * torch.randn(I, 1, device="cuda", generator=g))
    if rng.random() < 0.5:
        Ie[:: int(rng.integers(50, 500))] = Ie[int(rng.integers(0, I))]          # exact ties
    if rng.random() < 0.3:
        Ue[torch.randint(0, Bt, (max(1, Bt // 200),), device="cuda", generator=g)] = 0.0  # users whose lists overflow
    deg = rng.integers(0, 60, Bt)
    ptr = np.zeros(Bt + 1, dtype=np.int64); ptr[1:] = np.cumsum(deg)
    items = np.concatenate([np.sort(rng.choice(I, int(x), replace=False)) for x in deg] + [np.empty(0, int)]).astype(np.int32)
    ip, ix = torch.from_numpy(ptr).cuda(), torch.from_numpy(items).cuda()
    users = torch.arange(Bt, device="cuda")
    info = {}
    got = ops.score_topk(Ue, Ie, users, k, ip, ix, apply_sigmoid=sig, return_values=True, info=info)
    with ops.topk_options(collect=0):
        want = ops.score_topk(Ue, Ie, users, k, ip, ix, apply_sigmoid=sig, return_values=True)
    ok = bool(torch.equal(got[0], want[0]) and torch.equal(got[1], want[1]))
    bad += not ok
    print("case %2d: %5d users x %6d items d=%3d k=%2d sigmoid=%d spread=%.1f form=%s chunks=%s redone=%s fell_back=%s -> %s"
          % (c, Bt, I, d, k, sig, spread, info.get("form"), info.get("chunks"), info.get("users_redone"), info.get("calls_fallen_back"),
             "equal" if ok else "DIFFERENT"), flush=True)
print("fuzz: %d cases, %d different" % (cases, bad))
sys.exit(1 if bad else 0)
