#!/usr/bin/env python3
"""bench.py — BPR triples/sec of the LightGCN-3 training step on MI355X.

A step = one pass of the hot path over one batch of B sampled triples: K-layer propagation
of the full graph, fused gather-BPR-loss-grad, K-layer backward propagation, dense Adam —
exactly the work of one iteration of the reference's loop (trainer.py:42-56), on a synthetic
graph of the shape BASELINE.json names, inputs resident in HBM before the timed region.

    python bench.py [--gpus N --steps K --warmup W] [--workload yelp2018|amazon-book|synth-1M|synth-10M]

Prints ONE JSON line (see DESIGN.md §Measurement for every field).
"""
import argparse
import json
import os
import sys
import time

ROOT = os.path.dirname(os.path.abspath(__file__))
if ROOT not in sys.path:
    sys.path.insert(0, ROOT)

import numpy as np  # noqa: E402
import torch  # noqa: E402

HBM_PEAK_GBS = 8000.0  # MI355X_MICROARCH.md: HBM3E 8.0 TB/s spec
INFINITY_CACHE_BYTES = 256 << 20
# MI355X_MICROARCH.md "Indexed rows": chip-wide rates of whole-row gathers by where the rows are served from (GB/s;
# midpoints of the measured ranges): rows resident in the XCD's L2 16.8-18.8 TB/s; a 38 MB table out of the Infinity
# Cache 8.6 TB/s; random whole rows of a buffer far beyond the Infinity Cache 5.5-5.8 TB/s.
GATHER_L2_GBS, GATHER_MALL_GBS, GATHER_HBM_GBS = 17800.0, 8600.0, 5650.0


def parse():
    p = argparse.ArgumentParser()
    p.add_argument("--gpus", type=int, default=1)
    p.add_argument("--steps", type=int, default=300)
    p.add_argument("--warmup", type=int, default=30)
    p.add_argument("--workload", default=None,
                   help="graph shape (idgrec_amd.synth.SHAPES).  Default: yelp2018 (BASELINE configs[1]) on one GPU; synth-10M "
                        "(BASELINE configs[4]: 10M users x 5M items, 200M edges) for the user-row-sharded multi-GPU form")
    p.add_argument("--dim", type=int, default=None, help="embedding width (default 64; 256 on synth-10M, as configs[4])")
    p.add_argument("--layers", type=int, default=3)
    p.add_argument("--batch", type=int, default=1024)
    p.add_argument("--model", default="LightGCN", choices=["LightGCN", "MFBPR", "SimGCL"],
                   help="SimGCL (BASELINE configs[3]): hyper-parameters from configure/SimGCL.txt")
    p.add_argument("--atomic", action="store_true", help="float-atomic scatter instead of the deterministic one")
    p.add_argument("--split", type=int, default=0, help="row split threshold (0 = library default)")
    p.add_argument("--cpu-seconds", type=float, default=12.0, help="budget of the CPU baseline leg")
    p.add_argument("--no-cpu-baseline", action="store_true")
    p.add_argument("--separate-adam", action="store_true",
                   help="Adam as its own kernel instead of the last backward epilogue (PMC calibration runs)")
    p.add_argument("--seed", type=int, default=2024)
    p.add_argument("--hbm-leg", default="auto", choices=["auto", "on", "off"],
                   help="after the headline, time the dominant dense launch where it is HBM-bound — synth-10M at d=64, "
                        "gathered panel 3.84 GB = 15 x the Infinity Cache, on the scale point's graph — as roofline.hbm_bound, "
                        "and on synth-1M (384 MB panel, mostly Infinity-Cache hits) as the labelled roofline.cache_boundary; "
                        "auto = only for the default headline workload (yelp2018 LightGCN d=64)")
    p.add_argument("--epoch-leg", default="auto", choices=["auto", "on", "off"],
                   help="after the headline, run epochs through the plugin surface main.py uses (dataset files -> Data -> "
                        "models.<Model>.Trainer.train(), native sampler included) and report E / epoch_time as `epoch`; "
                        "auto = for graphs up to amazon-book size")
    p.add_argument("--backend", default="nccl", choices=["nccl", "gloo"],
                   help="torch.distributed backend of the sharded path: nccl (= RCCL over xGMI) for real runs; gloo only "
                        "to rehearse the multi-rank code path with several ranks sharing one GPU (host-staged collectives)")
    p.add_argument("--parallel", default="auto", choices=["auto", "dp", "shard"],
                   help="multi-GPU form: shard = user-row shards + per-layer all-reduces of the replicated item panel, ONE "
                        "global batch of B triples per Adam step (the north-star split, SURVEY.md §8e); dp = replicas + one "
                        "exchange of gradient rows per step, global batch N x B; auto (default) = shard as the headline and, "
                        "on more than one GPU, the replica form on the yelp2018 shape as a second, labelled field")
    p.add_argument("--comm", default="auto", choices=["auto", "native", "torch"],
                   help="collectives through libidgrec's RCCL communicator (native: on the step's own stream, or — from "
                        "64 MB — on a second stream the next product overlaps) or through torch.distributed (torch); auto = "
                        "native when every rank can set it up and passes its self-test, torch otherwise")
    p.add_argument("--dp-exchange", default="rows", choices=["rows", "grad"],
                   help="what dp replicas exchange: rows = all-gather of the batches' gradient rows before the backward "
                        "propagation (default), grad = all-reduce of the dense [n, d] gradient after it")
    p.add_argument("--ramp", default="graph", choices=["graph", "gemm"],
                   help="untimed clock ramp before the warm-up steps: the graph's own product on a scratch panel (default) or a "
                        "dense GEMM loop (profile runs: keeps the ramp's launches out of the per-kernel statistics)")
    p.add_argument("--scale-point", default="auto", choices=["auto", "on", "off"],
                   help="the multi-GPU lines' workload (synth-10M, d=256: BASELINE configs[4]) measured unsharded on ONE GPU in "
                        "this run: as `scale_point` of the default 1-GPU line (auto: only there) and as `single_gpu_reference` "
                        "/ `speedup_vs_1gpu` of a sharded line (auto: when the shape fits one GPU)")
    p.add_argument("--item-slices", type=int, default=0,
                   help="sharded form: row slices of the item panel whose collectives overlap the following slices' products "
                        "(0 = auto: 8 from four ranks on, 4 below, 1 while the panel is under 256 MB)")
    p.add_argument("--worker", action="store_true",
                   help="(internal) this process is a rank's WORKER: started by the rank's supervisor — the process the "
                        "launcher started, which never touches the GPU — so that a hung collective can be ended from outside "
                        "and the run retried once with --comm torch in fresh processes")
    p.add_argument("--attempt", type=int, default=1, help="(internal) 1 = first try, 2 = the retry over torch.distributed")
    p.add_argument("--force-sharded", action="store_true",
                   help="run a multi-GPU path even at world size 1 (exercises the RCCL code path): the user-row-sharded one, "
                        "or the replicas with --parallel dp")
    return p.parse_args()


def spmm_bytes(n, nnz, d):
    """algorithmic bytes of one Y = A.X."""
    gather = 4 * (n + 1) + 8 * nnz + 4 * nnz * d + 4 * n * d
    minimum = 4 * (n + 1) + 8 * nnz + 8 * n * d
    return gather, minimum


def build_workload(args, rank, world):
    import idgrec_amd.host as H
    import idgrec_amd.synth as S

    U, I, E = S.SHAPES[args.workload]
    t0 = time.time()
    users, items = S.generate(U, I, E, seed=0)
    ip, ix, dv = H.build_norm_adj(U, I, users, items)
    need = (args.steps + args.warmup) * args.batch
    tri, sampler_rate, pos_ptr, items32 = S.draw_triples(args.seed, users, items, U, I, need)  # native sampler, one host core
    return dict(U=U, I=I, E=len(users), indptr=ip, indices=ix, values=dv, triples=tri, prep_s=time.time() - t0,
                sampler_rate=sampler_rate, pos_ptr=pos_ptr, items32=items32)


def cpu_baseline(args, wl, W0):
    """The reference's step restated on stock PyTorch CPU ops (oracle/torch_ref.py), timed on
    this box's host cores on a bounded number of steps of the same workload."""
    from oracle.torch_ref import RefStep

    U, I = wl["U"], wl["I"]
    simgcl = None
    if args.model == "SimGCL":
        import utility.utility_function.tools as tools

        c = tools.read_configuration(os.path.join(ROOT, "configure", "SimGCL.txt"), "SimGCL")
        simgcl = (float(c["epsilon"]), float(c["temperature"]), float(c["ssl_lambda"]))
    ref = RefStep(wl["indptr"], wl["indices"], wl["values"], U, I, W0[:U], W0[U:], n_layers=args.layers,
                  lr=1e-3, propagate=(args.model != "MFBPR"), simgcl=simgcl)
    tri = torch.from_numpy(wl["triples"])
    B = args.batch

    def one(i):
        b = tri[i * B:(i + 1) * B]
        t0 = time.perf_counter()
        ref.step(b[:, 0], b[:, 1], b[:, 2])
        return time.perf_counter() - t0

    # torch's CPU sparse kernels do not scale to every hardware thread (256 threads: 30x slower than
    # 32 on the 2 x 64-core host): probe a few thread counts for one step each, keep the fastest
    ncpu = os.cpu_count() or 1
    best_t, best_dt, i = 1, None, 0
    for t in sorted({min(ncpu, c) for c in (16, 32, 64)}):
        torch.set_num_threads(t)
        one(i)          # warm-up at this thread count
        dt = one(i + 1)
        i += 2
        if best_dt is None or dt < best_dt:
            best_t, best_dt = t, dt
    torch.set_num_threads(best_t)
    done, t_used = 0, 0.0
    while t_used < args.cpu_seconds and (i + 1) * B <= len(tri):
        t_used += one(i)
        i += 1
        done += 1
    return {"value": done * B / t_used, "unit": "triples/s", "cores": torch.get_num_threads(), "kind": "port",
            "sample": "%d steps of the same %s B=%d workload, torch %s CPU ops (oracle/torch_ref.py), %.1f s, "
                      "fastest of 16/32/64 threads on %d hardware threads"
                      % (done, args.model, B, torch.__version__, t_used, ncpu)}


def cpu_baseline_scaled(args):
    """CPU leg of the multi-GPU lines (rank 0 only, after the timed region): the reference's step on torch CPU ops on
    the named shape when that is small, else on a 1/64-scale graph of the same degree law ("for C5 the
    CPU baseline times ... a 1/16-scale graph, stated explicitly" — 1/64 here to stay within ~30 s)."""
    import idgrec_amd.host as H
    import idgrec_amd.synth as S

    U, I, E = S.SHAPES[args.workload]
    scale = 1
    while E // scale > 4_000_000:
        scale *= 2
    U, I, E = U // scale, I // scale, E // scale
    users, items = S.generate(U, I, E, seed=0)
    ip, ix, dv = H.build_norm_adj(U, I, users, items)
    need = 64 * args.batch
    tri = S.draw_triples(args.seed, users, items, U, I, need)[0]
    wl = dict(U=U, I=I, indptr=ip, indices=ix, values=dv, triples=tri)
    W0 = S.xavier_uniform_panel(U, I, args.dim, args.seed).numpy()
    out = cpu_baseline(args, wl, W0)
    out["sample"] += "; graph: %s at 1/%d scale (%d users x %d items, %d edges) — a step's cost grows with nnz, so the " \
                     "full-size rate is about 1/%d of this" % (args.workload, scale, U, I, len(users), scale)
    return out


BENCH_TIMEOUT_S = float(os.environ.get("IDG_BENCH_TIMEOUT", "1500"))  # both attempts of a multi-GPU run together


def _error_line(args, world, message, **extra):
    """The ONE JSON line of a multi-GPU run that did not produce a measurement."""
    out = {"metric": "BPR triples/sec, LightGCN-%d dim=%s" % (args.layers, args.dim or 256), "value": None, "unit": "triples/s",
           "n_gpus": world, "steps": args.steps, "warmup": args.warmup, "ms_per_step": None, "higher_is_better": True,
           "scaling": "strong", "vs_baseline": None, "dtype": "f32", "data": "synthetic",
           "config": {"workload": args.workload or "synth-10M", "parallelism": "user-row shard x%d" % world}, "error": message}
    out.update(extra)
    return out


def self_launch(args):
    """`python bench.py --gpus N` without a launcher: start the N ranks as FRESH child processes (torch.distributed.run,
    one per GPU) before this process has touched the GPU, relay rank 0's JSON line, leave with the launcher's exit code.
    (Never an exec: a process that has initialised HIP must not be replaced — and this one has not, but a child is the
    form that is always safe.)  The launcher runs in its own process group and under a deadline (the ranks' supervisors
    keep their own, shorter ones): if it is still there after IDG_BENCH_TIMEOUT + 120 s the whole group is ended and a
    JSON line with "error" is printed instead of nothing."""
    import shutil
    import signal
    import subprocess
    import tempfile

    port = _free_port()
    cmd = [sys.executable, "-m", "torch.distributed.run", "--nnodes=1", "--nproc-per-node", str(args.gpus),
           "--master-addr", "127.0.0.1", "--master-port", str(port), os.path.abspath(__file__)] + sys.argv[1:]
    # the ranks' status files live in a directory of THIS run's own, made here and removed on the way out (a
    # name built from pid and port can meet a stale directory of an earlier run)
    sdir = tempfile.mkdtemp(prefix="idg_bench_")
    env = dict(os.environ, HSA_ENABLE_IPC_MODE_LEGACY=os.environ.get("HSA_ENABLE_IPC_MODE_LEGACY", "0"),
               IDG_BENCH_STATUS_DIR=sdir)
    rc = 1
    try:
        proc = subprocess.Popen(cmd, stdout=subprocess.PIPE, env=env, start_new_session=True)
        try:
            stdout, _ = proc.communicate(timeout=BENCH_TIMEOUT_S + 120)
        except subprocess.TimeoutExpired:
            try:
                os.killpg(proc.pid, signal.SIGKILL)  # the group this call created: launcher, supervisors, workers
            except ProcessLookupError:
                pass
            stdout, _ = proc.communicate()
            lines = [ln for ln in stdout.decode("utf-8", "replace").splitlines() if ln.startswith("{")]
            print(lines[-1] if lines else json.dumps(_error_line(args, args.gpus, "launcher still running after %.0f s: ended"
                                                                 % (BENCH_TIMEOUT_S + 120))), flush=True)
            sys.exit(1)
        lines = [ln for ln in stdout.decode("utf-8", "replace").splitlines() if ln.startswith("{")]
        if lines:
            print(lines[-1], flush=True)
        rc = proc.returncode if proc.returncode or lines else 1
    finally:
        shutil.rmtree(sdir, ignore_errors=True)
    sys.exit(rc)


def _free_port(preferred=None):
    """A TCP port nobody listens on right now: `preferred` if it can be bound, else one the kernel picks."""
    import socket

    for want in ([preferred] if preferred else []) + [0]:
        try:
            with socket.socket() as sk:
                sk.setsockopt(socket.SOL_SOCKET, socket.SO_REUSEADDR, 1)
                sk.bind(("127.0.0.1", int(want)))
                return sk.getsockname()[1]
        except OSError:
            continue
    raise RuntimeError("no free TCP port on 127.0.0.1")


# ---- multi-GPU ranks: a supervisor per rank (the process the launcher started) and its worker (the process that runs)
def _status_dir():
    """A directory of THIS run's own: named after the launcher's pid (the parent of every rank's supervisor; the same
    MASTER_PORT may serve the next run), handed to the workers through the environment."""
    d = os.environ.get("IDG_BENCH_STATUS_DIR")
    if not d:
        d = os.path.join(os.environ.get("TMPDIR", "/tmp"), "idg_bench_%d_%s_%d" % (os.getuid(), os.environ.get("MASTER_PORT", "0"),
                                                                                  os.getppid()))
    os.makedirs(d, mode=0o700, exist_ok=True)
    return d


def _write(path, text):
    tmp = "%s.%d.tmp" % (path, os.getpid())
    with open(tmp, "w") as f:
        f.write(text)
    os.replace(tmp, path)  # readers never see half a file


def _read(path):
    try:
        with open(path) as f:
            return f.read()
    except OSError:
        return None


def worker_phase(args, name):
    """Worker side: say where this rank is (the supervisor quotes it when the run is ended), and — tests only — stop
    there for good: IDG_BENCH_TEST_HANG="rank:phase[:attempt]" stands in for a rank that never returns from a collective."""
    rank = int(os.environ.get("RANK", "0"))
    if args.worker:
        _write(os.path.join(_status_dir(), "phase_a%d_r%d" % (args.attempt, rank)), name)
    hang = os.environ.get("IDG_BENCH_TEST_HANG", "")
    if hang:
        parts = hang.split(":")
        if parts[0] in ("*", str(rank)) and parts[1] == name and (len(parts) < 3 or int(parts[2]) == args.attempt):
            print("[bench] rank %d: IDG_BENCH_TEST_HANG at phase %s" % (rank, name), file=sys.stderr, flush=True)
            while True:
                time.sleep(3600)


def supervise_rank(args):
    """What the launcher's rank process does for N > 1: it never touches the GPU.  It starts the rank's WORKER (this file
    with --worker) as a child, watches it, its deadline and the other ranks' verdicts (files in a directory of this run's
    own: one node, as the contract says), and
      * on success relays rank 0's JSON line;
      * when ANY rank fails or the deadline passes — a rank hung inside ncclCommInitRank or a collective keeps every other
        rank waiting — ends its worker and starts the run ONCE more, in fresh processes, with --comm torch (torch's own
        process group instead of the library's communicator) on a rendezvous of its own;
      * when that fails too prints ONE JSON line with "error" (which attempt, which comm, the phase every rank had reached)
        and leaves with a non-zero status — within IDG_BENCH_TIMEOUT seconds (default 1500: inside the driver's 1800)."""
    import signal
    import subprocess

    rank, world = int(os.environ["RANK"]), int(os.environ["WORLD_SIZE"])
    sdir = _status_dir()
    if "IDG_BENCH_STATUS_DIR" not in os.environ:
        # a foreign launcher (the driver's torch.distributed.run): the directory's name is built from its pid and port and
        # may have served an earlier run — this rank's own files of that run must not be read as this run's verdicts
        for name in os.listdir(sdir):
            if name.endswith("_r%d" % rank) and name.split("_")[0] in ("status", "phase", "stdout", "port"):
                try:
                    os.remove(os.path.join(sdir, name))
                except OSError:
                    pass
    t_start = time.time()
    child = [None]

    def on_term(signum, frame):  # the launcher is ending the run: do not leave the worker behind
        if child[0] is not None and child[0].poll() is None:
            child[0].kill()
        sys.exit(128 + signum)

    signal.signal(signal.SIGTERM, on_term)
    signal.signal(signal.SIGINT, on_term)
    argv = [a for a in sys.argv[1:]]
    summary = []
    for attempt in (1, 2):
        budget = min(0.6 * BENCH_TIMEOUT_S, 900.0) if attempt == 1 else max(BENCH_TIMEOUT_S - (time.time() - t_start) - 20.0, min(30.0, BENCH_TIMEOUT_S / 2))
        deadline = time.time() + budget
        env = dict(os.environ, IDG_BENCH_STATUS_DIR=sdir)
        extra = ["--worker", "--attempt", str(attempt)]
        if attempt == 2:
            # a rendezvous of its own: the first attempt's keys stay in the launcher's store.  Rank 0 picks a port that is
            # free NOW (the computed one if nobody listens there) and publishes it; the others read it
            port_file = os.path.join(sdir, "port_a2_r0")
            if rank == 0:
                _write(port_file, str(_free_port(int(os.environ["MASTER_PORT"]) + 1 + (os.getuid() + int(os.environ["MASTER_PORT"])) % 89)))
            t_port = time.time() + 30
            while _read(port_file) is None and time.time() < t_port:
                time.sleep(0.1)
            env["MASTER_PORT"] = _read(port_file) or str(int(os.environ["MASTER_PORT"]) + 1)
            env["TORCHELASTIC_USE_AGENT_STORE"] = "False"  # rank 0's worker hosts the store
            extra += ["--comm", "torch"]
        mine = os.path.join(sdir, "status_a%d_r%d" % (attempt, rank))
        out_path = os.path.join(sdir, "stdout_a%d_r%d" % (attempt, rank))
        with open(out_path, "wb") as out_f:
            child[0] = subprocess.Popen([sys.executable, os.path.abspath(__file__)] + argv + extra, env=env, stdout=out_f)
        verdict = None
        while verdict is None:
            rc = child[0].poll()
            if rc is not None:
                verdict = "done" if rc == 0 else "failed: worker exit status %d" % rc
                break
            others = [_read(os.path.join(sdir, "status_a%d_r%d" % (attempt, r))) for r in range(world) if r != rank]
            if any(o is not None and o.startswith("failed") for o in others):
                verdict = "failed: another rank failed"
            elif time.time() > deadline:
                verdict = "failed: no result after %.0f s (deadline of attempt %d)" % (budget, attempt)
            else:
                time.sleep(0.25)
        if child[0].poll() is None:
            child[0].kill()     # the exact process this supervisor started
            child[0].wait()
        _write(mine, verdict)
        # every rank's verdict for this attempt (a rank that succeeded while another failed still goes round again: the
        # measurement needs all of them)
        t_wait = time.time() + 90
        while time.time() < t_wait:
            all_v = [_read(os.path.join(sdir, "status_a%d_r%d" % (attempt, r))) for r in range(world)]
            if all(v is not None for v in all_v):
                break
            time.sleep(0.25)
        all_v = [v or "unknown (no verdict)" for v in all_v]
        phases = {str(r): (_read(os.path.join(sdir, "phase_a%d_r%d" % (attempt, r))) or "not started") for r in range(world)}
        ok = all(v == "done" for v in all_v)
        summary.append({"attempt": attempt, "comm": "torch" if attempt == 2 else args.comm, "verdicts": all_v,
                        "phase_reached": phases, "seconds": round(time.time() - t_start, 1)})
        if ok:
            if rank == 0:
                lines = [ln for ln in (_read(out_path) or "").splitlines() if ln.startswith("{")]
                if not lines:
                    print(json.dumps(_error_line(args, world, "rank 0's worker ended without a result line", attempts=summary)),
                          flush=True)
                    sys.exit(1)
                line = json.loads(lines[-1])
                if attempt == 2:
                    line["retried"] = {"why": summary[0], "note": "first attempt ended by the watchdog; this line is the retry "
                                                                "over torch.distributed's process group (--comm torch)"}
                print(json.dumps(line), flush=True)
            sys.exit(0)
        print("[bench] rank %d: attempt %d failed: %s (phases %s)" % (rank, attempt, all_v, phases), file=sys.stderr, flush=True)
    if rank == 0:
        hung = [r for r, v in enumerate(summary[-1]["verdicts"]) if "deadline" in v or "another rank" in v]
        print(json.dumps(_error_line(args, world, "both attempts failed (library communicator, then torch.distributed); ranks "
                                                  "ended by the watchdog in the last attempt: %s" % hung, attempts=summary)),
              flush=True)
    sys.exit(1)


def single_gpu_point(args, workload="synth-10M", dim=256, steps=4, warmup=2, edges=None, before_engine=None):
    """The multi-GPU lines' workload on ONE GPU, unsharded, through the fused single-GPU engine (PropagationEngine): the
    1-GPU point the N-GPU speed-ups are quoted against, measured in the same run.  ~140 GB resident at synth-10M d=256.
    edges: the (users, items) arrays when the caller has drawn the graph already.  before_engine(graph, U, I, n, nnz,
    n_edges) -> dict: run on the built graph handle before the d=`dim` engine exists (the default line's HBM-bound
    roofline leg at d=64: same graph, drawn and scheduled once); its result is returned under "before_engine"."""
    import idgrec_amd.host as H
    import idgrec_amd.ops as ops
    import idgrec_amd.synth as S
    from idgrec_amd.engine import PropagationEngine

    t_all = time.perf_counter()
    U, I, E = S.SHAPES[workload]
    K, B = args.layers, args.batch
    users, items = edges if edges is not None else S.generate(U, I, E, seed=0)
    ip, ix, dv = H.build_norm_adj(U, I, users, items)
    n, nnz, n_edges = U + I, len(ix), len(users)
    tri = torch.from_numpy(S.draw_triples(args.seed, users, items, U, I, (steps + warmup) * B)[0]).cuda()
    del users, items
    graph = ops.Graph(ip, ix, dv, n, n, split_threshold=args.split)
    del ip, ix, dv
    extra = None
    if before_engine is not None:
        t_x = time.perf_counter()
        try:
            extra = before_engine(graph, U, I, n, nnz, n_edges)
        except Exception as exc:  # noqa: BLE001 - the scale point stands without it
            extra = {"error": "%s: %s" % (type(exc).__name__, str(exc)[:200])}
        graph._ws.clear()  # the d=64 layer buffers
        torch.cuda.empty_cache()
        t_all += time.perf_counter() - t_x  # not part of the scale point's own time
    eng = PropagationEngine(graph, U, I, dim, K, include_layer0=True, reg_lambda=1e-4, lr=1e-3,
                            params=S.xavier_uniform_panel(U, I, dim, args.seed).cuda())
    eng.store_grad = False  # (as the trainer path: see main())
    tu, tp, tn = tri[:, 0].contiguous(), tri[:, 1].contiguous(), tri[:, 2].contiguous()

    def step(i):
        s = slice(i * B, (i + 1) * B)
        if i + 1 < steps + warmup:
            s2 = slice((i + 1) * B, (i + 2) * B)
            eng.prefetch(tu[s2], tp[s2], tn[s2])
        eng.train_step(tu[s], tp[s], tn[s])

    for i in range(warmup):
        step(i)
    torch.cuda.synchronize()
    t0 = time.perf_counter()
    for i in range(warmup, warmup + steps):
        step(i)
    torch.cuda.synchronize()
    dt = time.perf_counter() - t0
    out = {"workload": "%s: %d users x %d items, %d train edges, nnz(A)=%d; LightGCN K=%d d=%d B=%d, ONE GPU, unsharded "
                       "(fused single-GPU engine)" % (workload, U, I, n_edges, nnz, K, dim, B),
           "ms_per_step": dt / steps * 1e3, "value": B * steps / dt, "unit": "triples/s", "steps": steps, "warmup": warmup,
           "loss_last": float(eng.loss.sum().item())}
    del eng, graph, tri, tu, tp, tn
    torch.cuda.empty_cache()
    out["seconds_including_graph_build"] = time.perf_counter() - t_all
    if extra is not None:
        out["before_engine"] = extra
    return out


def main():
    args = parse()
    if args.gpus > 1 and "WORLD_SIZE" not in os.environ and "RANK" not in os.environ:
        self_launch(args)
    if int(os.environ.get("WORLD_SIZE", "1")) > 1 and not args.worker and os.environ.get("IDG_BENCH_SUPERVISE", "1") != "0":
        supervise_rank(args)  # never returns: the measurement runs in a worker process this one watches
    worker_phase(args, "start")
    # stdout carries ONE line, the JSON.  Libraries print there too (RCCL announces its version on fd 1 when a
    # communicator comes up): hand fd 1 to stderr for the run and keep the real stdout for the result alone.
    sys.stdout.flush()
    result_fd = os.dup(1)
    os.dup2(2, 1)

    def emit(obj):
        sys.stdout.flush()
        os.write(result_fd, (json.dumps(obj) + "\n").encode())

    args.emit = emit
    rank = int(os.environ.get("RANK", "0"))
    world = int(os.environ.get("WORLD_SIZE", "1"))
    local = int(os.environ.get("LOCAL_RANK", "0"))
    if world != args.gpus:
        # a mislabelled scaling line is worse than none: the launcher's world size and --gpus must agree
        if world == 1:
            sys.exit("bench.py --gpus %d must be launched with torch.distributed.run --nproc-per-node %d"
                     % (args.gpus, args.gpus))
        sys.exit("bench.py: launched with WORLD_SIZE=%d but --gpus %d; pass --gpus %d" % (world, args.gpus, world))
    if (world > 1 or args.force_sharded) and args.parallel != "dp" and "GPU_MAX_HW_QUEUES" not in os.environ:
        # The sharded form runs several streams per process (step, batch preparation, communicators) and waits between
        # them ~2K times per step (the replicas wait once, and measure the same with 3 or 4 queues).  HIP spreads them
        # over GPU_MAX_HW_QUEUES (default 4) hardware queues in order of first use, and waits between streams on
        # DIFFERENT queues cost the host tens of microseconds each: measured on the sharded step at world size 1,
        # 0.53 ms with 2-3 queues, 0.59-0.70 ms (depending on which stream met which queue) with 4, 0.90 ms with 8.
        # Must be set before the HIP runtime initialises (the first torch.cuda call below).
        os.environ["GPU_MAX_HW_QUEUES"] = "3"
    if not torch.cuda.is_available():
        sys.exit("bench.py needs an MI355X (torch.cuda.is_available() is False); there is no CPU path to time.")
    torch.cuda.set_device(local if args.backend == "nccl" else local % torch.cuda.device_count())
    dist = None
    if world > 1 or args.force_sharded:
        import torch.distributed as dist_

        dist = dist_
        worker_phase(args, "init_process_group")
        dist.init_process_group(args.backend)

    import idgrec_amd.ops as ops
    import idgrec_amd.synth as S
    from idgrec_amd.engine import PropagationEngine

    multi = world > 1 or args.force_sharded
    form = "single" if not multi else ("dp" if args.parallel == "dp" else "shard")
    if args.workload is None:
        args.workload = "synth-10M" if (form == "shard" and world > 1) else "yelp2018"
    if args.dim is None:
        args.dim = 256 if args.workload == "synth-10M" else 64
    if multi:
        from idgrec_amd.replicated import run_replicated_bench
        from idgrec_amd.sharded import make_comm, run_sharded_bench

        worker_phase(args, "make_comm")
        comm, comm_name = make_comm(dist, args.comm)
        args.phase = lambda name: worker_phase(args, name)
        if form == "dp":
            out = run_replicated_bench(args, rank, world, dist, comm, comm_name)
        else:
            # the same workload on ONE GPU, measured in this run (rank 0, after the timed region): what speedup_vs_1gpu
            # divides by.  auto: when the unsharded step fits one GPU (9 panels + the graph)
            U_, I_, E_ = S.SHAPES[args.workload]
            fits = 9 * 4 * (U_ + I_) * args.dim + 24 * E_ < 250e9
            ref = None
            if args.scale_point == "on" or (args.scale_point == "auto" and world > 1 and fits):
                ref = lambda a, edges=None: single_gpu_point(a, a.workload, a.dim, edges=edges)  # noqa: E731
            out = run_sharded_bench(args, rank, world, dist, comm, comm_name, single_gpu_reference=ref)
            if args.parallel == "auto" and world > 1:
                rep = run_replicated_bench(args, rank, world, dist, comm, comm_name, workload="yelp2018", dim=64)
                if rank == 0:
                    out["replicas"] = {k: rep[k] for k in ("value", "unit", "ms_per_step", "config", "replicas_bit_identical")}
                    out["replicas"]["what"] = ("NOT the headline: %d replicas of the yelp2018-shape graph, global batch %d x B — "
                                               "%d x the reference's batch_size per Adam step (trainer.py:36); one all-gather of "
                                               "gradient rows per step" % (world, world, world))
        worker_phase(args, "cpu_baseline")
        if rank == 0 and not args.no_cpu_baseline:
            out["cpu_baseline"] = cpu_baseline_scaled(args)
        dist.barrier()
        if rank == 0:
            args.emit(out)
        worker_phase(args, "teardown")
        if hasattr(comm, "close"):
            comm.close()
        dist.destroy_process_group()
        worker_phase(args, "done")
        return

    wl = build_workload(args, rank, world)
    U, I, d, K, B = wl["U"], wl["I"], args.dim, args.layers, args.batch
    n, nnz = U + I, len(wl["indices"])
    graph = None
    if args.model in ("LightGCN", "SimGCL"):
        graph = ops.Graph(wl["indptr"], wl["indices"], wl["values"], n, n, split_threshold=args.split)
    W0 = S.xavier_uniform_panel(U, I, d, args.seed)
    eng = PropagationEngine(graph, U, I, d, K, include_layer0=(args.model != "SimGCL"), reg_lambda=1e-4, lr=1e-3,
                            deterministic=not args.atomic, params=W0.cuda())
    if args.model == "SimGCL":
        import utility.utility_function.tools as tools

        torch.cuda.manual_seed(args.seed)  # the noise streams follow the device seed: same seed, same run

        c = tools.read_configuration(os.path.join(ROOT, "configure", "SimGCL.txt"), "SimGCL")
        eng.ssl = (float(c["epsilon"]), float(c["temperature"]), float(c["ssl_lambda"]))
    eng.fuse_adam = not args.separate_adam
    # the step of the trainer path (PackedRecommender.fused_train_step): the Adam update consumes the finished gradient in
    # the last backward product's epilogue and the [n, d] gradient panel is not written out (nobody reads it after a step)
    eng.store_grad = os.environ.get("IDG_BENCH_STORE_GRAD") == "1"  # (A/B knob: "1" = the panel is written out as well)
    tri = torch.from_numpy(wl["triples"]).cuda()
    tu, tp, tn = tri[:, 0].contiguous(), tri[:, 1].contiguous(), tri[:, 2].contiguous()
    losses = torch.zeros((args.steps + args.warmup, 3 if args.model == "SimGCL" else 2), dtype=torch.float32, device="cuda")

    # the mini-batches as views of the epoch's id tensors, made once (as the trainer does: list(tools.mini_batch(...)),
    # utility/utility_function/tools.py:55-64), and one row of the loss record per step
    batches = [(tu[i * B:(i + 1) * B], tp[i * B:(i + 1) * B], tn[i * B:(i + 1) * B]) for i in range(args.steps + args.warmup)]
    loss_rows = list(losses.unbind(0))

    def batch(i):
        return batches[i]

    last = args.warmup + args.steps - 1

    def step(i):
        if i < last:
            eng.prefetch(*batches[i + 1])  # one-batch lookahead of the index-only work (as the trainer does)
        eng.train_step(*batches[i], loss_out=loss_rows[i])

    # (the ramp's panel has a different width than the bench's, so its launches are another instantiation of the kernel
    #  and stay out of the measured kernel's row in a per-kernel trace of this command)
    S.ramp_clocks(graph=graph if graph is not None and nnz * d <= 4e9 and args.ramp == "graph" else None, d=32 if d != 32 else 64)
    if graph is not None and args.model == "LightGCN":
        for k in range(1, K):
            eng.forward_layer(k)  # allocates the two layer buffers of the launch-timing leg BEFORE the timed region (at
            #                       config-5 size: 2 x 15 GB of hipMalloc, ~1 s)
    for i in range(args.warmup):
        step(i)
    torch.cuda.synchronize()

    plan0 = eng._plan.stats() if getattr(eng, "_plan", None) is not None else None
    # timed region: exactly --steps steps, nothing else (no events, no instrumented launches)
    t0 = time.perf_counter()
    for i in range(args.warmup, args.warmup + args.steps):
        step(i)
    t_issue = time.perf_counter() - t0  # host time to issue the steps; < dt means the GPU, not the host, is the limit
    torch.cuda.synchronize()
    dt = time.perf_counter() - t0
    plan1 = eng._plan.stats() if getattr(eng, "_plan", None) is not None else None

    ms_per_step = dt / args.steps * 1e3
    value = B * args.steps / dt
    out = {
        "metric": "BPR triples/sec, %s-%d dim=%d" % (args.model, K, d),
        "value": value, "unit": "triples/s", "n_gpus": 1, "steps": args.steps, "warmup": args.warmup,
        "ms_per_step": ms_per_step, "higher_is_better": True, "scaling": "weak", "vs_baseline": None,
        "dtype": "f32", "data": "synthetic",
        "config": {"workload": "%s-shape graph: %d users x %d items, %d train edges, nnz(A)=%d; %s K=%d d=%d B=%d; "
                               "step = %s"
                               % (args.workload, U, I, wl["E"], nnz, args.model, K, d, B,
                                  "clean + 2 perturbed propagations + fused BPR + InfoNCE + one shared backward propagation + dense Adam"
                                  if args.model == "SimGCL" else "propagate + fused BPR + backward propagate + dense Adam"),
                   "batch": B, "dim": d, "layers": K, "scatter": "atomic" if args.atomic else "deterministic",
                   "gradient_panel": "consumed by the Adam update in the last backward product's epilogue, not written out "
                                     "(the trainer path's setting; weights, moments and losses are those of the stored form)"},
        "loss_first_last": [float(x) for x in (losses[args.warmup].sum().item(), losses[-1].sum().item())],
        "host_issue_ms_per_step": t_issue / args.steps * 1e3,
    }
    if plan0 is not None and plan1 is not None and plan1["steps"] - plan0["steps"] == args.steps:
        # the step is ONE library call (idg_step_run_f32): the host keeps at most two steps queued and BLOCKS in the call
        # until the step before the previous one has finished, so host_issue_ms_per_step follows the device; what the host
        # actually spends per step is the issue time minus that blocked time
        blocked = plan1["ms_blocked"] - plan0["ms_blocked"]
        out["host_busy_ms_per_step"] = max(t_issue * 1e3 - blocked, 0.0) / args.steps
        out["step_call"] = {"library_calls_per_step": 1, "entry_point": "idg_step_run_f32",
                            "ms_in_call_per_step": (plan1["ms_in_calls"] - plan0["ms_in_calls"]) / args.steps,
                            "ms_blocked_in_pacing_per_step": blocked / args.steps,
                            "steps_without_a_stream_wait": plan1["waits_skipped"] - plan0["waits_skipped"],
                            "what": "host_busy = (issue time of the timed loop - time blocked in the pacing wait) / steps: "
                                    "Python loop + the one call's own work (the next batch's preparation on the side stream, "
                                    "K forward products, fused BPR, backward chain with the Adam epilogue, end-of-step event)"}
    hbm_leg = False
    if graph is not None:
        out["roofline"] = roofline(args, eng, graph, n, nnz, d, K, args.workload, in_step_form=(args.model == "LightGCN"),
                                   bitmap=eng.touched)
        hbm_leg = args.hbm_leg == "on" or (args.hbm_leg == "auto" and args.workload == "yelp2018" and d == 64
                                           and args.model == "LightGCN")
    # beside the headline: the host sampler's own rate, and full-rank evaluation (propagate + fused
    # score/mask/top-20 of every user against the train CSR) on the same tables — both outside the timed region
    out["sampler"] = {"value": wl["sampler_rate"], "unit": "triples/s", "what": "native MT19937 sampler + shuffle permutation, "
                      "bit-identical to the reference's NumPy stream, one host core"}
    if U * I <= 4e10:
        with torch.no_grad():
            fin = eng.propagate(force=True)
            ip_d, ix_d = torch.from_numpy(wl["pos_ptr"]).cuda(), torch.from_numpy(wl["items32"]).cuda()
            all_users = torch.arange(U, device="cuda")
            ops.score_topk(fin[:U], fin[U:], all_users, 20, ip_d, ix_d)
            torch.cuda.synchronize()
            t_e = time.perf_counter()
            for _ in range(3):
                fin = eng.propagate(force=True)
                ops.score_topk(fin[:U], fin[U:], all_users, 20, ip_d, ix_d)
            torch.cuda.synchronize()
            t_e = (time.perf_counter() - t_e) / 3
        out["eval"] = {"value": U / t_e, "unit": "users/s", "ms_per_full_evaluation": t_e * 1e3,
                       "what": "propagate + score + train-positive mask + top-20 for all %d users x %d items" % (U, I)}
    if args.epoch_leg == "on" or (args.epoch_leg == "auto" and wl["E"] <= 3_000_000):
        out["epoch"] = epoch_leg(args)
    if not args.no_cpu_baseline:
        out["cpu_baseline"] = cpu_baseline(args, wl, W0.numpy())
    scale_point = args.scale_point == "on" or (args.scale_point == "auto" and args.workload == "yelp2018" and d == 64
                                               and args.model == "LightGCN")
    del eng, graph, tri, tu, tp, tn, losses
    torch.cuda.empty_cache()
    if hbm_leg:
        try:
            out["roofline"]["cache_boundary"] = cache_boundary_leg(args)
        except Exception as exc:  # noqa: BLE001 - the headline stands without it
            out["roofline"]["cache_boundary"] = {"error": "%s: %s" % (type(exc).__name__, str(exc)[:200])}
    if scale_point:
        # the 1-GPU point of the multi-GPU curve: `bench.py --gpus N` (N > 1) measures BASELINE configs[4], not this
        # line's workload — the same step on ONE GPU, so that a 1 -> N ratio compares like with like.  The HBM-bound
        # roofline leg (d=64) runs on the same graph handle first: synth-10M is drawn and scheduled once.
        def hook(g, U_, I_, n_, nnz_, ne_):
            out["roofline"]["hbm_bound"] = hbm_bound_leg(args, graph=g, shape=(U_, I_, n_, nnz_, ne_))

        try:
            out["scale_point"] = single_gpu_point(args, before_engine=hook if hbm_leg else None)
            err = out["scale_point"].pop("before_engine", None)
            if err and "hbm_bound" not in out["roofline"]:
                out["roofline"]["hbm_bound"] = err
        except Exception as exc:  # noqa: BLE001 - the headline stands without it
            out["scale_point"] = {"error": "%s: %s" % (type(exc).__name__, str(exc)[:200])}
    elif hbm_leg:
        try:
            out["roofline"]["hbm_bound"] = hbm_bound_leg(args)
        except Exception as exc:  # noqa: BLE001
            out["roofline"]["hbm_bound"] = {"error": "%s: %s" % (type(exc).__name__, str(exc)[:200])}
    args.emit(out)


def _time_launches(fn, launches_per_call, reps, warm=2):
    """Average device time of one launch: `reps` calls of fn() (each `launches_per_call` back-to-back launches on
    torch's current stream — the stream the library launches on) between ONE HIP event pair.  The figure includes the
    kernel boundary between consecutive launches (~1.5 us each), not the ~5 us an event pair around every single
    launch would add; rocprofv3's per-kernel average of the same command is the check (profiles/)."""
    for _ in range(warm):
        fn()
    a, b = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    a.record()
    for _ in range(reps):
        fn()
    b.record()
    torch.cuda.synchronize()
    return a.elapsed_time(b) / (reps * launches_per_call) * 1e-3  # seconds


def roofline(args, eng, graph, n, nnz, d, K, workload, in_step_form, bitmap=None, reps=None):
    """SURVEY.md §8(d) for the dominant kernel, measured live AFTER the timed region: achieved = algorithmic gather
    bytes of one Y = A.X over the average launch time of the dense tiled SpMM in the form a training step launches it
    (layers 1..K-1 of the forward: product + running layer sum).  `traffic*` fields are NOT measured in this run: they
    come from a builder-side rocprofv3 --pmc pass committed under profiles/ (traffic_source says which)."""
    dense_layers = list(range(1, K)) if in_step_form and K >= 2 else []
    reps = reps or max(3, min(40, int(2e-2 / max(1e-9, nnz * d * 4 / 6e12))))  # ~20 ms of launches, 3..40 calls
    if dense_layers:
        def fwd():
            for k in dense_layers:
                eng.forward_layer(k)
        spmm_s = _time_launches(fwd, len(dense_layers), reps)
        form = "layers 1..%d of the step's forward (plain products into the layer buffers; the layer sum is formed by the " \
               "last product's epilogue), %d launches between one HIP event pair" % (K - 1, reps * len(dense_layers))
    else:
        # this step form does not run its dense products one call per layer (SimGCL shares its first product between
        # the passes): time the same dense launch on its own
        Y = torch.empty_like(eng.params)
        spmm_s = _time_launches(lambda: graph.spmm_raw(eng.params, out=Y), 1, reps)
        form = "plain Y = A.X on the parameter panel, %d launches between one HIP event pair" % reps
    rows_s = None
    if in_step_form and bitmap is not None and dense_layers:
        # the row-restricted last layer (+ the layer sum in its epilogue): the whole restricted forward minus its dense layers
        fwd_s = _time_launches(lambda: graph.propagate_mean_raw(eng.params, K, eng.inc, out=eng.final, out_rows=bitmap), 1, reps)
        rows_s = max(fwd_s - len(dense_layers) * spmm_s, 0.0)
    gather, minimum = spmm_bytes(n, nnz, d)
    achieved = gather / spmm_s / 1e9
    traffic, source, hit = None, None, None
    tname = "traffic_%s_d%d.json" % (workload, d)
    rounds = sorted((x for x in os.listdir(os.path.join(ROOT, "profiles")) if x.startswith("r")), reverse=True)
    for sub in rounds + [""]:  # the newest round's file
        tfile = os.path.join(ROOT, "profiles", sub, tname)
        if os.path.exists(tfile):
            tj = json.load(open(tfile))
            traffic, hit = tj.get("hbm_bytes_per_launch"), tj.get("l2_hit_rate")
            source = "profiles/%s (builder-side rocprofv3 --pmc passes on an earlier run of this command; NOT measured in " \
                     "this run; FETCH_SIZE x2 per the gfx950 guide + WRITE_SIZE — bytes at the L2s' memory side, Infinity-Cache " \
                     "hits included, i.e. not DRAM traffic when the panel is cache resident)" % os.path.join(sub, tname)
            break
    info = graph.info()
    resident = bool(4 * n * d < INFINITY_CACHE_BYTES)
    # what the hardware guide's gather rates allow for THIS hit mix: L2 hits at the L2-resident gather rate, the rest out of
    # the Infinity Cache (cache-resident panel) or out of HBM (random whole rows) — harmonic mix by gathered bytes
    miss_gbs = GATHER_MALL_GBS if resident else GATHER_HBM_GBS
    ceiling = (1.0 / (hit / GATHER_L2_GBS + (1.0 - hit) / miss_gbs)) if hit is not None else miss_gbs
    return {
        "bound": "l2+infinity-cache gather" if resident else "hbm",
        "kernel": "spmm_tile_kernel<%d,1,8,dyn> (split rows combined in-kernel)" % (d // 4),
        "achieved": achieved, "peak": HBM_PEAK_GBS, "unit": "GB/s", "frac": achieved / HBM_PEAK_GBS,
        # `frac` is SURVEY.md §8(d)'s figure as specified (gather-equivalent bytes over the HBM peak): with a cache-resident
        # panel it is not an HBM fraction and may exceed 1.  `model_estimate` is a two-level MODEL, not an upper bound
        # (ADVICE r03): the guide's gather rates mixed by an L2 hit rate taken from an EARLIER profiled run; gathers served
        # by the CU's own L1 / LDS never reach the L2 the hit rate is counted at, so achieved / model_estimate can exceed 1.
        "model_estimate": ceiling, "frac_of_model_estimate": achieved / ceiling,
        "model_estimate_basis": "NOT an upper bound. MI355X_MICROARCH.md 'Indexed rows': L2-resident rows %.1f TB/s, %s %.2f TB/s, "
                                "harmonic mix by the L2 hit rate %s; L1-served gathers are outside the model"
                         % (GATHER_L2_GBS / 1e3, "Infinity Cache" if resident else "random rows out of HBM", miss_gbs / 1e3,
                            ("%.3f (PMC, same file as `traffic`)" % hit) if hit is not None else "not measured: miss rate assumed for every gather"),
        "traffic": traffic, "traffic_source": source,
        "us_per_launch": spmm_s * 1e6, "timing": form,
        # PMC-measured fabric-side bytes per launch over the launch time measured here: what the memory side of
        # the L2s moved (Infinity-Cache hits included), as a rate and as a fraction of the HBM peak
        "traffic_gbs": (traffic / spmm_s / 1e9) if traffic else None,
        "frac_traffic": (traffic / spmm_s / 1e9 / HBM_PEAK_GBS) if traffic else None,
        "row_restricted_last_layer_us": rows_s * 1e6 if rows_s is not None else None,
        "bytes_gather": gather, "bytes_min": minimum,
        "frac_bytes_min": minimum / spmm_s / 1e9 / HBM_PEAK_GBS,
        "cache_resident": resident,
        "tiles": info["n_tiles"], "split_rows": info["n_long_rows"],
    }


def dense_launch_leg(args, workload, d=64, K=3, graph=None, shape=None, reps=5):
    """The dominant dense launch on another graph, after the headline's timed region: same kernel, same in-step launch
    form, timed by roofline().  graph / shape=(U, I, n, nnz, n_edges): a handle the caller has built already (the scale
    point's synth-10M graph); otherwise the graph is drawn, scheduled and freed here."""
    import idgrec_amd.host as H
    import idgrec_amd.ops as ops
    import idgrec_amd.synth as S
    from idgrec_amd.engine import PropagationEngine

    own = graph is None
    if own:
        U, I, E = S.SHAPES[workload]
        users, items = S.generate(U, I, E, seed=0)
        ip, ix, dv = H.build_norm_adj(U, I, users, items)
        n, nnz, n_edges = U + I, len(ix), len(users)
        del users, items
        graph = ops.Graph(ip, ix, dv, n, n, split_threshold=args.split)
        del ip, ix, dv
    else:
        U, I, n, nnz, n_edges = shape
    g = torch.Generator(device="cuda").manual_seed(args.seed)
    params = (torch.rand((n, d), device="cuda", generator=g) * 2 - 1) * (6.0 / (U + d)) ** 0.5
    eng = PropagationEngine(graph, U, I, d, K, include_layer0=True, deterministic=False, params=params)
    r = roofline(args, eng, graph, n, nnz, d, K, workload, in_step_form=True, reps=reps)
    r["workload"] = "%s: %d users x %d items, %d train edges, nnz(A)=%d, d=%d (gathered panel %.0f MB = %.1f x the %d MiB " \
                    "Infinity Cache)" % (workload, U, I, n_edges, nnz, d, 4 * n * d / 1e6, 4 * n * d / INFINITY_CACHE_BYTES,
                                         INFINITY_CACHE_BYTES >> 20)
    for k in ("peak", "unit", "row_restricted_last_layer_us"):
        r.pop(k, None)
    del eng, params
    if own:
        del graph
    torch.cuda.empty_cache()
    return r


def hbm_bound_leg(args, graph=None, shape=None):
    """"the d=64 >= 60 % of HBM peak target should be demonstrated on a graph with 4nd > 256 MiB".
    synth-10M at d=64: 15 M rows, nnz 398 M, gathered panel 3.84 GB = 15 x the Infinity Cache — the dense launch is
    HBM-bound there (L2 hit rate 11 %, profiles/r0N/traffic_synth-10M_d64.json).  In the default run the graph is the
    scale point's (drawn and scheduled once)."""
    return dense_launch_leg(args, "synth-10M", d=64, K=3, graph=graph, shape=shape, reps=5)


def cache_boundary_leg(args):
    """synth-1M at d=64 (gathered panel 384 MB = 1.5 x the Infinity Cache): between the cache-resident headline and the
    HBM-bound leg — most gathers still hit the Infinity Cache, so this is NOT an HBM figure and is labelled
    so."""
    r = dense_launch_leg(args, "synth-1M", d=64, K=3, reps=5)
    r["bound"] = "infinity-cache boundary (panel 1.5 x the Infinity Cache: mostly cache hits, not an HBM figure)"
    return r


def epoch_leg(args, epochs=3):
    """the end-to-end epoch rate E / epoch_time through the plugin surface main.py drives — dataset
    files -> data_loader.Data -> models.<Model>.Trainer(...).train() — native sampler, shuffle, host->device copy of
    the triples and the per-epoch loss read-back included; evaluation excluded (the trainer logs its training time
    per epoch before it tests).  The fastest of epochs 2..N is reported (epoch 1 allocates)."""
    import importlib
    import logging
    import re
    import shutil
    import tempfile

    import idgrec_amd.synth as S
    import utility.utility_data.data_loader as data_loader
    import utility.utility_function.tools as tools

    root = tempfile.mkdtemp(prefix="idg_bench_epoch_")
    cwd = os.getcwd()
    try:
        os.chdir(ROOT)
        S.make_dataset(root, args.workload, n_test=1)
        cfg = tools.read_configuration("./configure/%s.txt" % args.model, args.model)
        cfg.update(dataset=args.workload, dataset_path=root + "/", training_epochs=str(epochs), interval=str(10 ** 6),
                   embedding_size=str(args.dim), batch_size=str(args.batch))
        if "GCN_layer" in cfg:
            cfg["GCN_layer"] = str(args.layers)
        times = []

        class Grab(logging.Handler):
            def emit(self, record):
                m = re.search(r"Training time: ([0-9.]+)", record.getMessage())
                if m:
                    times.append(float(m.group(1)))

        logger = logging.getLogger("idg_bench_epoch")
        logger.setLevel(logging.INFO)
        logger.addHandler(Grab())
        tools.set_seed(args.seed)
        t0 = time.perf_counter()
        data = data_loader.Data(cfg["dataset_path"] + cfg["dataset"], cfg)
        t_data = time.perf_counter() - t0
        t0 = time.perf_counter()
        trainer = importlib.import_module("models." + args.model).Trainer(None, cfg, data, torch.device("cuda"), logger)
        t_init = time.perf_counter() - t0
        import contextlib
        import io

        with contextlib.redirect_stderr(io.StringIO()):  # tqdm bars
            trainer.train()
        E = int(len(data.train_user))
        best = min(times[1:]) if len(times) > 1 else times[0]
        return {"value": E / best, "unit": "triples/s", "epoch_s": best, "epochs_timed": times, "train_edges": E,
                "data_load_s": t_data, "model_and_graph_init_s": t_init,
                "what": "E / epoch_time through models.%s.Trainer.train() (the path main.py drives): native sampler + shuffle "
                        "+ upload + %d fused steps + one loss read-back per epoch; evaluation not included"
                        % (args.model, E // args.batch + 1)}
    finally:
        os.chdir(cwd)
        shutil.rmtree(root, ignore_errors=True)


if __name__ == "__main__":
    main()
