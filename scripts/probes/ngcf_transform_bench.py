"""Times the NGCF layer kernels on their own at the yelp2018 shape (n = 69,716 rows, d = 64): transform forward / backward,
parameter gradients, tails — event-timed, 200 launches each.  IDG_NGCF_PROBE / IDG_NGCF_WGS are read by the library."""
import ctypes as C
import os
import sys

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__)))))
import torch

from idgrec_amd import native, ops

lib, check = native.lib, native.check
n, d = int(os.environ.get("N_ROWS", 69716)), 64
g = torch.Generator(device="cuda").manual_seed(0)
side, ego, gS = (torch.randn(n, d, device="cuda", generator=g) for _ in range(3))
W1, W2 = (torch.randn(d, d, device="cuda", generator=g) * 0.1 for _ in range(2))
S, BI, g1, g2 = (torch.empty(n, d, device="cuda") for _ in range(4))
p = lambda t: C.c_void_p(t.data_ptr())
st = ops._stream()


def timed(name, fn, reps=200):
    for _ in range(10):
        fn()
    torch.cuda.synchronize()
    a, b = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    a.record()
    for _ in range(reps):
        fn()
    b.record()
    torch.cuda.synchronize()
    print("%-28s %7.2f us" % (name, a.elapsed_time(b) * 1e3 / reps), flush=True)


timed("transform fwd (+BI)", lambda: check(lib.idg_ngcf_transform_f32(p(side), p(ego), p(W1), p(W2), n, d, d, p(S), p(BI), st), "f"))
timed("transform fwd (no BI)", lambda: check(lib.idg_ngcf_transform_f32(p(side), p(ego), p(W1), p(W2), n, d, d, p(S), None, st), "f"))
timed("transform bwd", lambda: check(lib.idg_ngcf_transform_bwd_f32(p(gS), p(side), p(ego), p(W1), p(W2), n, d, d, p(g1), p(g2), st), "b"))
timed("copy 3 panels (torch)", lambda: (S.copy_(side), BI.copy_(ego)))

b1, b2 = (torch.randn(d, device="cuda", generator=g) * 0.1 for _ in range(2))
E1 = torch.empty(n, d, device="cuda")
F = torch.empty(n, 4 * d, device="cuda")
gF = torch.randn(n, 4 * d, device="cuda", generator=g)
bm = torch.randint(-2 ** 31, 2 ** 31 - 1, ((n + 31) // 32,), device="cuda", dtype=torch.int64).to(torch.int32)
ws = torch.empty(int(lib.idg_ngcf_layer_bwd_workspace_bytes(d)), dtype=torch.uint8, device="cuda")
wg = torch.empty(2 * d * d + 2 * d, device="cuda")
ws2 = torch.empty(int(lib.idg_ngcf_wgrad_workspace_bytes(d, d)), dtype=torch.uint8, device="cuda")
slot = C.c_void_p(F.data_ptr() + 4 * d)
gslot = C.c_void_p(gF.data_ptr() + 4 * d)
u64 = C.c_uint64
timed("tail fwd", lambda: check(lib.idg_ngcf_tail_ex_f32(p(S), None, p(b1), p(b2), n, d, 0.2, 0.1, u64(1), u64(2), p(E1), slot, 4 * d, st), "t"))
timed("tail bwd", lambda: check(lib.idg_ngcf_tail_bwd_ex_f32(p(E1), p(gS), gslot, 4 * d, p(bm), n, d, 0.2, 0.1, u64(1), u64(2), p(g1), st), "t"))
timed("wgrad", lambda: check(lib.idg_ngcf_wgrad_f32(p(side), p(BI), p(gS), n, d, d, p(wg), p(ws2), st), "w"))
PD = float(os.environ.get("P_DROP", "0.1"))
timed("LAYER fwd (one kernel)", lambda: check(lib.idg_ngcf_layer_fwd_f32(p(side), p(ego), p(W1), p(W2), p(b1), p(b2), n, d, 0.2, PD, u64(1), u64(2), p(E1), slot, 4 * d, st), "f"))
timed("LAYER bwd (one kernel+reduce)", lambda: check(lib.idg_ngcf_layer_bwd_f32(p(E1), p(gS), gslot, 4 * d, p(bm), p(side), p(ego), p(W1), p(W2), n, d, 0.2, PD, u64(1), u64(2), p(g1), p(g2), p(wg), p(ws), st), "b"))
