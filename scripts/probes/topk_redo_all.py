"""Worst case of top-K form 3: EVERY user's candidate list overflows (all-zero user table: every score ties).  Round 6: such
a call falls back to the exact form as a whole, decided on the device (collect_verdict_kernel); with the fall-back off
(fallback_permille = -1) every user is redone one by one by topk_redo_kernel — the cliff the fall-back removes.  Both are
timed against the exact form and compared with it."""
import os, sys, time
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__)))))
import torch
import idgrec_amd.ops as ops

U, I, d, k = int(sys.argv[1]), int(sys.argv[2]), int(sys.argv[3]), 20
g = torch.Generator(device="cuda").manual_seed(0)
Ue = torch.zeros(U, d, device="cuda")
Ie = torch.randn(I, d, device="cuda", generator=g) * 0.3
every = torch.arange(U, device="cuda")


def timed(reps=3):
    info = {}
    out = ops.score_topk(Ue, Ie, every, k, return_values=True, info=info)
    torch.cuda.synchronize()
    t0 = time.perf_counter()
    for _ in range(reps):
        out = ops.score_topk(Ue, Ie, every, k, return_values=True)
    torch.cuda.synchronize()
    return out, (time.perf_counter() - t0) / reps * 1e3, info


with ops.topk_options(collect=0):
    want, t_exact, _ = timed()
got, t_fb, info_fb = timed()
with ops.topk_options(fallback_permille=-1):
    redo, t_redo, info_redo = timed(1)
eq = lambda a, b: bool(torch.equal(a[0], b[0]) and torch.equal(a[1], b[1]))  # noqa: E731
print("%d users x %d items x d=%d, every list overflows" % (U, I, d))
print("  exact form                    %9.2f ms" % t_exact)
print("  form 3, whole-call fall-back  %9.2f ms (%.2fx)  equal: %s  %s" % (t_fb, t_fb / t_exact, eq(got, want), info_fb))
print("  form 3, every user redone     %9.2f ms (%.2fx)  equal: %s  %s" % (t_redo, t_redo / t_exact, eq(redo, want), info_redo))
