"""Do the item-side and user-side half-products of one layer overlap when launched on two streams?  (yelp2018 shape, d=64,
world size 1: R_g^T . X_U and R_g . X_I are 27.9 us each back to back.)  python scripts/probes/two_stream_pair.py"""
import os, sys, time
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__)))))
import numpy as np, torch
import idgrec_amd.sharded as sh, idgrec_amd.synth as S, idgrec_amd.ops as ops

U, I, E = S.SHAPES["yelp2018"]
users, items = S.generate(U, I, E, seed=0)
ui, iu = sh.shard_adjacency_from_edges(users, items, U, I, 0, U)
k = sh.HipKernels()
G_ui = k.make_graph(*ui, U, I)
G_iu = k.make_graph(*iu, I, U)
d = 64
XU, XI = torch.randn(U, d, device="cuda"), torch.randn(I, d, device="cuda")
YU, YI = torch.empty_like(XU), torch.empty_like(XI)
s2 = torch.cuda.Stream()
main = torch.cuda.current_stream()

def serial(n):
    for _ in range(n):
        k.spmm(G_iu, XU, Y=YI)
        k.spmm(G_ui, XI, Y=YU)

def parallel(n):
    for _ in range(n):
        ev = torch.cuda.Event(); ev.record(main); s2.wait_event(ev)
        with torch.cuda.stream(s2):
            k.spmm(G_iu, XU, Y=YI)
        k.spmm(G_ui, XI, Y=YU)
        ev2 = torch.cuda.Event(); ev2.record(s2); main.wait_event(ev2)

for name, fn in (("serial", serial), ("two streams", parallel), ("serial", serial), ("two streams", parallel)):
    fn(20); torch.cuda.synchronize()
    a, b = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    t0 = time.perf_counter(); a.record(); fn(200); b.record(); t_host = time.perf_counter() - t0
    torch.cuda.synchronize()
    print("%-12s %.1f us per pair on the device, %.1f us of host time per pair" % (name, a.elapsed_time(b) * 1e3 / 200, t_host * 1e6 / 200))
