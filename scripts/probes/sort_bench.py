import os, sys, torch
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import idgrec_amd.ops as ops
n, U = 69716, 31668
for B in (256, 1024, 2048, 2730):
    u = torch.randint(0, U, (B,), device="cuda"); p = torch.randint(0, n - U, (B,), device="cuda"); q = torch.randint(0, n - U, (B,), device="cuda")
    ops.bpr_plan_raw(u, p, q, U, n, 64); torch.cuda.synchronize()
    a, b = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    a.record()
    for _ in range(100): ops.bpr_plan_raw(u, p, q, U, n, 64)
    b.record(); torch.cuda.synchronize()
    print("B=%d (3B=%d): plan (keys + sort) %.1f us" % (B, 3 * B, a.elapsed_time(b) * 10))
