"""Where does the host time of one sharded step go?  (world size 1, cProfile over 200 steps)"""
import cProfile, os, pstats, sys, types
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
os.environ.update(RANK="0", WORLD_SIZE="1", LOCAL_RANK="0", MASTER_ADDR="127.0.0.1", MASTER_PORT="29513")
import numpy as np, torch, torch.distributed as dist
torch.cuda.set_device(0)
dist.init_process_group("nccl")
import idgrec_amd.sharded as sh, idgrec_amd.synth as S, idgrec_amd.host as H
U, I, E = S.SHAPES["yelp2018"]
users, items = S.generate(U, I, E, seed=0)
ip, ix, dv = H.build_norm_adj(U, I, users, items)
ui, iu = sh.shard_adjacency(ip, ix, dv, U, I, 0, U)
eng = sh.ShardedEngine(sh.HipKernels(), sh.TorchComm(dist), ui, iu, U, I, 64, 3, True, 1e-4, 1e-3)
eng.P.copy_(S.xavier_uniform_panel(U, I, 64, 2024))
rng = np.random.default_rng(0)
B, steps = 1024, 230
pick = rng.integers(0, len(users), B * steps)
tu, tp = torch.from_numpy(users[pick]).cuda(), torch.from_numpy(items[pick]).cuda()
tn = torch.from_numpy(rng.integers(0, I, B * steps)).cuda()
def run(a, b):
    for i in range(a, b):
        if i + 1 < steps:
            n = slice((i + 1) * B, (i + 2) * B); eng.prefetch(tu[n], tp[n], tn[n])
        s = slice(i * B, (i + 1) * B); eng.train_step(tu[s], tp[s], tn[s], B)
run(0, 30); torch.cuda.synchronize()
import time
acc = {}
def timed(obj, name):
    f = getattr(obj, name)
    def g(*a, **k):
        t0 = time.perf_counter(); r = f(*a, **k); acc[name] = acc.get(name, 0.0) + time.perf_counter() - t0; acc[name + "#"] = acc.get(name + "#", 0) + 1
        return r
    setattr(obj, name, g)
for nm in ("spmm", "lincomb", "fill", "bpr", "adam", "prepare", "release", "wait_rows"):
    timed(eng.k, nm)
for nm in ("all_reduce_async", "wait"):
    timed(eng.comm, nm)
import idgrec_amd.ops as _ops
for nm in ("bpr_touch_rows_raw", "bpr_plan_raw", "bpr_fwd_bwd_raw", "spmm_ex_raw", "lincomb_raw"):
    timed(_ops, nm)
t0 = time.perf_counter(); run(30, 230); t_issue = time.perf_counter() - t0; torch.cuda.synchronize(); t_all = time.perf_counter() - t0
print("issue %.1f us/step, wall %.1f us/step" % (t_issue / 200 * 1e6, t_all / 200 * 1e6))
for k in sorted(k for k in acc if not k.endswith("#")):
    print("  %-18s %6.1f us/step  (%.1f calls/step, %.1f us/call)" % (k, acc[k] / 200 * 1e6, acc[k + "#"] / 200, acc[k] / acc[k + "#"] * 1e6))
print("  accounted %.1f us/step" % (sum(v for k, v in acc.items() if not k.endswith("#")) / 200 * 1e6))
dist.destroy_process_group()
