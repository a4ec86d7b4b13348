"""InfoNCE (pair dedup / pair raw / cross) on fixed inputs: prints a checksum of every output and the time per call.
Run once with the working tree's library and once with IDG_LIB_PATH=<other build>: equal checksums = bit-identical."""
import hashlib
import os
import sys

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__)))))
import torch

from idgrec_amd import ops

U, I, d = 31668, 38048, 64
n = U + I
B = int(os.environ.get("SSL_B", 2048))
g = torch.Generator(device="cuda").manual_seed(1)
v1, v2 = torch.randn(n, d, device="cuda", generator=g), torch.randn(n, d, device="cuda", generator=g)
users = torch.randint(0, U, (B,), device="cuda", generator=g)
pop = (torch.rand(B, device="cuda", generator=g) ** 3 * I).long()  # popular items repeat
h = lambda t: hashlib.sha1(t.detach().cpu().numpy().tobytes()).hexdigest()[:12]  # noqa: E731


def run(name, fn, reps=100):
    out = fn()
    torch.cuda.synchronize()
    a, b = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    a.record()
    for _ in range(reps):
        fn()
    b.record()
    torch.cuda.synchronize()
    print("%-22s %8.1f us  %s" % (name, a.elapsed_time(b) * 1e3 / reps, " ".join(h(t) for t in out)), flush=True)


def pair(dedup):
    g1, g2 = torch.zeros_like(v1), torch.zeros_like(v2)
    loss = ops.infonce_pair_raw(v1, v2, users, pop, U, 0.2, g1=g1, g2=g2, dedup=dedup, grad_scale=0.1, accumulate=True)
    return loss, g1, g2


def cross():
    gg = torch.zeros_like(v1)
    loss = torch.zeros(2, device="cuda")
    ops.infonce_cross_raw(v1, users, pop, U, 0.2, gg, loss=loss, grad_scale=0.1)
    return loss, gg


run("pair, unique ids", lambda: pair(True))
run("pair, raw ids", lambda: pair(False))
run("cross, raw ids", cross)
