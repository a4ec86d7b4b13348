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
BF16_MFMA_PEAK_FLOPS = 2.5e15  # the dense bf16 MFMA peak (the headline 5 PFLOP/s figure includes 2:1 sparsity)
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
    p.add_argument("--parity-steps", type=int, default=3,
                   help="sharded form with a single-GPU reference in the run: this many training steps are taken on the N ranks "
                        "BEFORE the warm-up, from the fresh tables, and repeated on ONE device (rank 0, the fused engine) from "
                        "the same tables and batches; the line carries `parity_vs_1gpu` {loss_rel_err, final_rows_rel_err, "
                        "table_rel_err, tol 1e-4, ok} and the run leaves with status 3 when ok is false (0 = no check)")
    p.add_argument("--panel-bits", type=int, default=32, choices=[32, 24],
                   help="sharded form: 24 = the panel-sized exchanges travel as 24-bit values (2^-16 relative: inside the 1e-4 "
                        "bar) through an explicit exchange — pack, all-to-all, this library's rank-ordered sum, all-gather — "
                        "3/4 of the bytes on the links and bit-reproducible run to run; 32 (default) = RCCL's fp32 collectives")
    p.add_argument("--reduce-order", default="rccl", choices=["rccl", "rank"],
                   help="sharded form: rank = the panel-sized reductions as an explicit exchange (all-to-all, this library's sum "
                        "over the ranks IN RANK ORDER, all-gather) instead of RCCL's all-reduce / reduce-scatter: the same fp32 "
                        "bytes on the links, bit-reproducible run to run (SURVEY 8e); implied by --panel-bits 24")
    p.add_argument("--item-slices", type=int, default=0,
                   help="sharded form: row slices of the item panel whose collectives overlap the following slices' products "
                        "(0 = auto: 8 from four ranks on, 4 below, 1 while the panel is under 256 MB)")
    p.add_argument("--configs-leg", default="auto", choices=["auto", "on", "off"],
                   help="after the headline, one short measured leg for each of the other single-GPU BASELINE configs — [0] MFBPR "
                        "B=2048 yelp2018, [2] LightGCN amazon-book, [3] SimGCL amazon-book B=2048 — as the `configs` array of the "
                        "line (auto: only beside the default headline)")
    p.add_argument("--pmc", default="auto", choices=["auto", "on", "off"],
                   help="measure roofline.traffic IN THIS RUN: after every other leg, child processes `rocprofv3 --kernel-trace "
                        "--pmc <counters> -- python3 bench.py --roofline-only ...` (one counter set per pass) re-run the dense "
                        "launches of the roofline legs' graphs; auto: beside the default headline, when rocprofv3 is on PATH")
    p.add_argument("--roofline-only", default=None, metavar="GRAPHS",
                   help="(internal, the child of --pmc) comma-separated graphs: run only the dominant dense launch on each, a few "
                        "times, and print one JSON line of launch times and tile counts")
    p.add_argument("--steady-steps", type=int, default=-1,
                   help="after the timed region, this many further steps in 50-step windows between HIP events -> "
                        "`steady_state` {min, median, max} (default: 500 beside the default headline, else 0)")
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
    need = (args.steps + args.warmup + max(getattr(args, "steady_steps", 0), 0)) * args.batch
    tri, sampler_rate, pos_ptr, items32 = S.draw_triples(args.seed, users, items, U, I, need)  # native sampler, one host core
    return dict(U=U, I=I, E=len(users), indptr=ip, indices=ix, values=dv, triples=tri, prep_s=time.time() - t0,
                sampler_rate=sampler_rate, pos_ptr=pos_ptr, items32=items32, users=users, items=items)


def cpu_baseline(args, wl, W0, threads=None, warm=True):
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
    if threads:  # a short leg beside the headline: the thread count the headline's probe chose, no warm-up beyond one step
        best_t = int(threads)
        torch.set_num_threads(best_t)
        if warm:
            one(i)
            i += 1
    else:
        for t in sorted({min(ncpu, c) for c in (16, 32, 64)}):
            torch.set_num_threads(t)
            one(i)          # warm-up at this thread count
            dt = one(i + 1)
            i += 2
            if best_dt is None or dt < best_dt:
                best_t, best_dt = t, dt
        torch.set_num_threads(best_t)
    done, t_used = 0, 0.0
    while (t_used < args.cpu_seconds or done == 0) and (i + 1) * B <= len(tri):
        t_used += one(i)
        i += 1
        done += 1
    return {"value": done * B / t_used, "unit": "triples/s", "cores": torch.get_num_threads(), "kind": "port",
            "sample": "%d steps of the same %s B=%d workload, torch %s CPU ops (oracle/torch_ref.py), %.1f s, %s on %d hardware "
                      "threads" % (done, args.model, B, torch.__version__, t_used,
                                   ("%d threads (the headline leg's choice)%s" % (best_t, "" if warm else ", no warm-up step"))
                                   if threads else "fastest of 16/32/64 threads", ncpu)}


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


PARITY_EXIT = 3  # exit status of a multi-GPU run whose result differs from the single-GPU result (the line says where)


def _flag_parity_failure(line):
    """A measured N-rank line whose `parity_vs_1gpu.ok` is False is not a measurement: it gets an "error" field and the
    process that prints it leaves with PARITY_EXIT — no retry (a wrong result is not a hang).  ok = None (the reference
    could not run) is reported, not failed."""
    par = line.get("parity_vs_1gpu") or {}
    if par.get("ok") is False:
        line["error"] = ("parity_vs_1gpu FAILED: the %d-rank run differs from the single-GPU run of the same steps beyond %g "
                         "(loss %.3g, final rows %.3g, tables %.3g)" % (line.get("n_gpus", 0), par.get("tol", 0),
                                                                       par.get("loss_rel_err", float("nan")),
                                                                       par.get("final_rows_rel_err", float("nan")),
                                                                       par.get("table_rel_err", float("nan"))))
        return True
    return False


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
        try:  # (the launcher answers 1 for any failed rank: a result that failed its parity check keeps its own status)
            if lines and (json.loads(lines[-1]).get("parity_vs_1gpu") or {}).get("ok") is False:
                rc = PARITY_EXIT
        except ValueError:
            pass
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
                if (line.get("parity_vs_1gpu") or {}).get("ok") is False:
                    sys.exit(PARITY_EXIT)  # the line is printed (with "error"); a wrong result is not retried
            sys.exit(0)
        print("[bench] rank %d: attempt %d failed: %s (phases %s)" % (rank, attempt, all_v, phases), file=sys.stderr, flush=True)
    if rank == 0:
        hung = [r for r, v in enumerate(summary[-1]["verdicts"]) if "deadline" in v or "another rank" in v]
        print(json.dumps(_error_line(args, world, "both attempts failed (library communicator, then torch.distributed); ranks "
                                                  "ended by the watchdog in the last attempt: %s" % hung, attempts=summary)),
              flush=True)
    sys.exit(1)


def single_gpu_point(args, workload="synth-10M", dim=256, steps=4, warmup=2, edges=None, before_engine=None, parity=None,
                     big_batch=0):
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
    tri_big = None
    if big_batch:
        tri_big = torch.from_numpy(S.draw_triples(args.seed + 1, users, items, U, I, 4 * big_batch)[0][: 4 * big_batch]).cuda()
    del users, items
    if before_engine is not None:
        _save_csr(args, workload, ip, ix, dv, U, I, 64)  # (the d=64 HBM-bound leg's graph: for the traffic children)
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
    parity_out = None
    if parity is not None:
        # the N ranks ran these batches from these tables before their warm-up (sharded.parity_capture): the same steps on
        # this one device are the oracle of the N-rank run — before anything else touches the engine
        from idgrec_amd.sharded import parity_compare

        try:
            parity_out = parity_compare(eng, parity)
        except Exception as exc:  # noqa: BLE001 - reported in the line; the reference timing below still runs
            parity_out = {"ok": None, "error": "%s: %s" % (type(exc).__name__, str(exc)[:300])}
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
    if big_batch and tri_big is not None:
        # configs[4] reports B = 1024 AND a throughput-oriented B — a step's time is nearly independent of B << nnz
        # (the products run over the whole graph either way), so triples/s grows with the batch: B = 2^20, same engine
        try:
            bu, bp, bn = (tri_big[:, c].contiguous() for c in range(3))
            n_big = tri_big.shape[0] // big_batch
            for i in range(n_big):
                if i == 1:
                    torch.cuda.synchronize()
                    t0 = time.perf_counter()
                sl = slice(i * big_batch, (i + 1) * big_batch)
                eng.train_step(bu[sl], bp[sl], bn[sl])
            torch.cuda.synchronize()
            dt_big = (time.perf_counter() - t0) / (n_big - 1)
            out["batch_2p20"] = {"batch": big_batch, "ms_per_step": dt_big * 1e3, "value": big_batch / dt_big, "unit": "triples/s",
                                 "steps": n_big - 1, "warmup": 1, "loss_last": float(eng.loss.sum().item())}
            del bu, bp, bn
        except Exception as exc:  # noqa: BLE001 - the B = 1024 point stands without it
            out["batch_2p20"] = {"error": "%s: %s" % (type(exc).__name__, str(exc)[:200])}
    del eng, graph, tri, tu, tp, tn, tri_big
    torch.cuda.empty_cache()
    out["seconds_including_graph_build"] = time.perf_counter() - t_all
    if extra is not None:
        out["before_engine"] = extra
    if parity_out is not None:
        out["parity_vs_1gpu"] = parity_out
    return out


def main():
    args = parse()
    if args.roofline_only:
        return roofline_only(args)
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
                ref = lambda a, edges=None, parity=None: single_gpu_point(a, a.workload, a.dim, edges=edges, parity=parity)  # noqa: E731
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
        parity_failed = False
        if rank == 0:
            parity_failed = _flag_parity_failure(out)
            args.emit(out)
        worker_phase(args, "teardown")
        if hasattr(comm, "close"):
            comm.close()
        dist.destroy_process_group()
        worker_phase(args, "done")
        if parity_failed and not args.worker:
            sys.exit(PARITY_EXIT)  # (under a supervisor the worker ends cleanly and the SUPERVISOR leaves with this status)
        return

    default_headline = args.workload == "yelp2018" and args.dim == 64 and args.model == "LightGCN" and args.layers == 3
    if args.steady_steps < 0:
        args.steady_steps = 500 if default_headline else 0
    wl = build_workload(args, rank, world)
    U, I, d, K, B = wl["U"], wl["I"], args.dim, args.layers, args.batch
    n, nnz = U + I, len(wl["indices"])
    graph = None
    if args.model in ("LightGCN", "SimGCL"):
        graph = ops.Graph(wl["indptr"], wl["indices"], wl["values"], n, n, split_threshold=args.split)
        _save_csr(args, args.workload, wl["indptr"], wl["indices"], wl["values"], U, I, d)  # (kept only when --pmc will run)
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
    n_all = args.steps + args.warmup + args.steady_steps
    losses = torch.zeros((n_all, 3 if args.model == "SimGCL" else 2), dtype=torch.float32, device="cuda")

    # the mini-batches as views of the epoch's id tensors, made once (as the trainer does: list(tools.mini_batch(...)),
    # utility/utility_function/tools.py:55-64), and one row of the loss record per step
    batches = [(tu[i * B:(i + 1) * B], tp[i * B:(i + 1) * B], tn[i * B:(i + 1) * B]) for i in range(n_all)]
    loss_rows = list(losses.unbind(0))

    def batch(i):
        return batches[i]

    last = n_all - 1

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

    steady = None
    if args.steady_steps >= 100:
        # the driver times 20 steps (5 ms): a 2 % round-to-round delta cannot be told from box noise there.  500 further
        # steps of the same loop, in 50-step windows between HIP events on the step's stream (no host synchronisation in
        # between: the pipeline stays full), give the spread
        win = 50
        n_win = args.steady_steps // win
        evs = [torch.cuda.Event(enable_timing=True) for _ in range(n_win + 1)]
        i0 = args.warmup + args.steps
        evs[0].record()
        for w in range(n_win):
            for i in range(i0 + w * win, i0 + (w + 1) * win):
                step(i)
            evs[w + 1].record()
        torch.cuda.synchronize()
        per = sorted(evs[w].elapsed_time(evs[w + 1]) / win for w in range(n_win))
        steady = {"ms_per_step_min": per[0], "ms_per_step_median": per[len(per) // 2], "ms_per_step_max": per[-1],
                  "windows": n_win, "steps_per_window": win,
                  "what": "%d further steps of the same loop right after the timed region, %d-step windows between HIP events "
                          "on the step's stream: the spread a %d-step timed region cannot show" % (n_win * win, win, args.steps)}
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
        "loss_first_last": [float(x) for x in (losses[args.warmup].sum().item(), losses[args.warmup + args.steps - 1].sum().item())],
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
    if steady is not None:
        out["steady_state"] = steady
    if graph is not None and args.model == "LightGCN" and K >= 2:
        out["step_bytes"] = step_bytes(wl, n, nnz, d, K, B, ms_per_step)
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
    del eng, graph, tri, tu, tp, tn, losses, batches, loss_rows
    torch.cuda.empty_cache()
    if args.configs_leg == "on" or (args.configs_leg == "auto" and default_headline):
        threads = (out.get("cpu_baseline") or {}).get("cores")
        out["configs"] = other_configs(args, wl, threads)
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
            out["scale_point"] = single_gpu_point(args, before_engine=hook if hbm_leg else None, big_batch=1 << 20)
            err = out["scale_point"].pop("before_engine", None)
            if err and "hbm_bound" not in out["roofline"]:
                out["roofline"]["hbm_bound"] = err
        except Exception as exc:  # noqa: BLE001 - the headline stands without it
            out["scale_point"] = {"error": "%s: %s" % (type(exc).__name__, str(exc)[:200])}
        # and the evaluation of that configuration: one fused score + mask + top-20 call at its geometry
        try:
            out["eval_scale_point"] = eval_scale_point()
        except Exception as exc:  # noqa: BLE001 - the headline stands without it
            out["eval_scale_point"] = {"error": "%s: %s" % (type(exc).__name__, str(exc)[:200])}
    elif hbm_leg:
        try:
            out["roofline"]["hbm_bound"] = hbm_bound_leg(args)
        except Exception as exc:  # noqa: BLE001
            out["roofline"]["hbm_bound"] = {"error": "%s: %s" % (type(exc).__name__, str(exc)[:200])}
    if hbm_leg:
        try:
            out["roofline"]["hbm_reuse_free"] = reuse_free_leg(args)
        except Exception as exc:  # noqa: BLE001
            out["roofline"]["hbm_reuse_free"] = {"error": "%s: %s" % (type(exc).__name__, str(exc)[:200])}
    if args.pmc == "on" or (args.pmc == "auto" and default_headline and hbm_leg):
        torch.cuda.empty_cache()
        try:
            attach_measured_traffic(args, out)
        except Exception as exc:  # noqa: BLE001 - the file-based traffic figures stay
            out["roofline"]["traffic_in_run_error"] = "%s: %s" % (type(exc).__name__, str(exc)[:300])
    flatten_north_star(out)
    args.emit(out)


def flatten_north_star(out):
    """The figures BASELINE.json's north star is stated on, as TOP-LEVEL scalars of `roofline` (the driver's
    record keeps top-level scalars only; the nested legs they come from stay where they were):
      hbm_*             the dense launch where it is HBM-bound (synth-10M, d=64: panel 15 x the Infinity Cache) — fraction
                        of the 8 TB/s peak by SURVEY 8d's gather bytes and by the bytes measured at the L2s' memory side
      hbm_reuse_free_*  the same on the regular graph with uniformly scattered neighbours (nothing to reuse: DRAM bytes),
                        and on its strided twin
      eval_bf16_mfma_frac  the evaluation's collect pass against the dense bf16 MFMA peak (2.5 PFLOP/s), beside the
                        fp32-equivalent figure that is a speed-up label, not a roofline fraction"""
    rf = out.get("roofline")
    if not isinstance(rf, dict):
        return
    hb, free = rf.get("hbm_bound"), rf.get("hbm_reuse_free")
    if isinstance(hb, dict) and "frac" in hb:
        rf.update(hbm_frac_gather=hb["frac"], hbm_frac_traffic=hb.get("frac_traffic"), hbm_us_per_launch=hb.get("us_per_launch"),
                  hbm_traffic_over_bytes_gather=hb.get("traffic_over_bytes_gather"), hbm_l2_hit_rate=hb.get("traffic_l2_hit_rate"),
                  hbm_workload="synth-10M d=64: 15 M rows, 398 M entries, gathered panel 3.84 GB = 15 x the Infinity Cache")
    if isinstance(free, dict) and "frac" in free:
        rf.update(hbm_reuse_free_frac=free["frac"], hbm_reuse_free_frac_traffic=free.get("frac_traffic"),
                  hbm_reuse_free_us_per_launch=free.get("us_per_launch"), hbm_reuse_free_l2_hit_rate=free.get("traffic_l2_hit_rate"),
                  hbm_reuse_free_pattern=free.get("access_pattern"))
        st = free.get("strided")
        if isinstance(st, dict) and "frac" in st:
            rf.update(hbm_reuse_free_strided_frac=st["frac"], hbm_reuse_free_strided_us_per_launch=st.get("us_per_launch"))
    ev = out.get("eval_scale_point")
    if isinstance(ev, dict) and "bf16_mfma_frac" in ev:
        rf["eval_bf16_mfma_frac"] = ev["bf16_mfma_frac"]
        rf["eval_fp32_equivalent_tflops"] = ev.get("fp32_equivalent_tflops")
    sp = out.get("scale_point")
    if isinstance(sp, dict) and sp.get("ms_per_step"):
        # (also inside `roofline`: the one nested object whose scalars the driver's record is known to keep)
        out["scale_point_ms_per_step"] = rf["scale_point_ms_per_step"] = sp["ms_per_step"]
        if isinstance(sp.get("batch_2p20"), dict) and sp["batch_2p20"].get("value"):
            out["scale_point_batch_2p20_triples_per_s"] = rf["scale_point_batch_2p20_triples_per_s"] = sp["batch_2p20"]["value"]

def trained_like_tables(users, items, d, seed=0):
    """Tables with the norm spread of a TRAINED model instead of i.i.d. Gaussians (form 3's candidate count
    depends on it): item popularity p_i ~ (rank + 1)^-0.8 under a random id permutation (the bench graphs' law), item norms
    ~ p_i^0.3 (a factor ~40 between the most and the least popular of 5 M items), user norms log-normal (sigma 0.3), and
    a direction every row shares to a degree (0.3 of its length: popular items score high for everybody)."""
    g = torch.Generator(device="cuda").manual_seed(seed)
    common = torch.nn.functional.normalize(torch.randn(1, d, device="cuda", generator=g), dim=1)

    def rows(n, norms):
        x = torch.nn.functional.normalize(torch.randn(n, d, device="cuda", generator=g), dim=1)
        x = torch.nn.functional.normalize(x + 0.3 * common, dim=1)
        return (x * norms[:, None]).contiguous()

    rank = torch.randperm(items, device="cuda", generator=g).float()
    pop = (rank + 1.0) ** -0.8
    item_norm = 0.3 * d ** 0.5 * (pop / pop.mean()) ** 0.3
    user_norm = 0.3 * d ** 0.5 * torch.exp(0.3 * torch.randn(users, device="cuda", generator=g))
    return rows(users, user_norm), rows(items, item_norm)


def eval_scale_point(users=16384, items=5_000_000, d=256, k=20, degree=50):
    """BASELINE configs[4]'s evaluation (1e7 users x 5e6 items x d = 256: 25.6 PFLOP in fp32 per evaluation), one call of it:
    `users` test users against the whole catalogue, `degree` train items each masked.  Tables with a trained model's norm
    spread (trained_like_tables; round 5 ran on Gaussian tables, on which `users_redone: 0` says little) — and, beside
    them, the Gaussian tables once.  Reports the candidates per user (what the collect pass lets through: the capacity is
    4096 at this catalogue size), the users redone, whether the call fell back, the exact form once, and the call's rate
    against the dense bf16 MFMA peak."""
    import idgrec_amd.ops as ops

    rng = np.random.default_rng(0)
    ix = np.sort(rng.integers(0, items - degree, (users, degree)), axis=1) + np.arange(degree)[None, :]  # ascending, distinct
    ip = torch.arange(0, (users + 1) * degree, degree, dtype=torch.int64, device="cuda")
    ixd = torch.from_numpy(ix.reshape(-1).astype(np.int32)).cuda()
    every = torch.arange(users, device="cuda")

    def timed(Ue, Ie, reps):
        ops.score_topk(Ue, Ie, every, k, ip, ixd)
        torch.cuda.synchronize()
        t0 = time.perf_counter()
        for _ in range(reps):
            ops.score_topk(Ue, Ie, every, k, ip, ixd)
        torch.cuda.synchronize()
        return (time.perf_counter() - t0) / reps

    def leg(Ue, Ie):
        info = {"candidates": True}
        ops.score_topk(Ue, Ie, every, k, ip, ixd, info=info)
        return timed(Ue, Ie, 3), info

    Ue, Ie = trained_like_tables(users, items, d)
    t, info = leg(Ue, Ie)
    with ops.topk_options(collect=0):
        t_exact = timed(Ue, Ie, 1)
    del Ue, Ie
    g = torch.Generator(device="cuda").manual_seed(0)
    Ug, Ig = torch.randn(users, d, device="cuda", generator=g) * 0.3, torch.randn(items, d, device="cuda", generator=g) * 0.3
    t_gauss, info_gauss = leg(Ug, Ig)
    del Ug, Ig
    torch.cuda.empty_cache()
    flop = 2.0 * users * items * d
    # what the bf16 matrix cores execute in a form-3 call: d data features + the 16-feature bound k-step, the collect pass over
    # the whole catalogue + the floor pass over about a tenth of it
    bf16_flop = 2.0 * users * items * (d + 16) * 1.1
    return {"what": "one fused score + mask + top-%d call at configs[4]'s geometry: %d users x %d items x d=%d, %d train items "
                    "per user masked; tables with a trained model's norm spread (item norms ~ popularity^0.3, a shared "
                    "direction; bench.py trained_like_tables)" % (k, users, items, d, degree),
            "ms_per_call": t * 1e3, "fp32_equivalent_tflops": flop / t / 1e12,
            "bf16_mfma_frac": bf16_flop / t / BF16_MFMA_PEAK_FLOPS,
            "bf16_mfma_frac_what": "bf16 MFMA work of the call (2 x users x items x (d + 16) x 1.1: collect pass + floor pass) / "
                                   "the whole call's time / the %.1f PFLOP/s dense bf16 peak — the roofline fraction; "
                                   "fp32_equivalent_tflops is a speed-up label against the exact form's arithmetic" % (BF16_MFMA_PEAK_FLOPS / 1e15),
            "form": info.get("form"), "users_redone": info.get("users_redone"), "calls_fallen_back": info.get("calls_fallen_back"),
            "candidates_per_user": info.get("candidates"), "candidate_capacity": 4096 if items >= 1_000_000 else 1024,
            "exact_form_ms_per_call": t_exact * 1e3, "exact_form_tflops": flop / t_exact / 1e12,
            "gaussian_tables": {"ms_per_call": t_gauss * 1e3, "users_redone": info_gauss.get("users_redone"),
                                "candidates_per_user": info_gauss.get("candidates")},
            "s_per_evaluation_of_1e7_users": 1e7 / users * t, "exact_form_s_per_evaluation_of_1e7_users": 1e7 / users * t_exact}


# ---- graphs handed to the --pmc children: the CSR adjacency a leg built, left in a directory of this run's own
def _pmc_dir(args, create=False):
    d = getattr(args, "_pmc_dir_path", None)
    if d is None and create:
        import tempfile

        d = args._pmc_dir_path = tempfile.mkdtemp(prefix="idg_bench_pmc_")
    return d


def _pmc_wanted(args):
    import shutil

    if args.pmc == "off" or getattr(args, "roofline_only", None):
        return False
    default_headline = args.workload == "yelp2018" and args.dim == 64 and args.model == "LightGCN" and args.layers == 3
    return (args.pmc == "on" or default_headline) and shutil.which("rocprofv3") is not None


def _save_csr(args, name, ip, ix, dv, U, I, d):
    """Leave a leg's adjacency for the traffic children (uncompressed .npy files: written and read at memory speed)."""
    if not _pmc_wanted(args):
        return
    root = os.path.join(_pmc_dir(args, create=True), name)
    os.makedirs(root, exist_ok=True)
    np.save(os.path.join(root, "indptr.npy"), np.asarray(ip))
    np.save(os.path.join(root, "indices.npy"), np.asarray(ix))
    np.save(os.path.join(root, "values.npy"), np.asarray(dv))
    json.dump({"U": int(U), "I": int(I), "d": int(d)}, open(os.path.join(root, "shape.json"), "w"))


def roofline_only(args):
    """The child of --pmc, run under `rocprofv3 --kernel-trace --pmc ...`: for every graph directory named, the dominant
    dense launch in the step's form (layers 1..K-1 of the forward), a few times, nothing else.  Prints one JSON line:
    per graph the tile count (= grid size / 256: how the parent tells the graphs' dispatches apart) and the launch time
    seen here (under counter collection: not a performance figure)."""
    import idgrec_amd.ops as ops
    from idgrec_amd.engine import PropagationEngine

    torch.cuda.set_device(0)
    res = {}
    for root in args.roofline_only.split(","):
        name = os.path.basename(root.rstrip("/"))
        try:
            shape = json.load(open(os.path.join(root, "shape.json")))
            ip, ix, dv = (np.load(os.path.join(root, f + ".npy")) for f in ("indptr", "indices", "values"))
            U, I, d = shape["U"], shape["I"], shape["d"]
            n, nnz = U + I, len(ix)
            graph = ops.Graph(ip, ix, dv, n, n, split_threshold=args.split)
            del ip, ix, dv
            g = torch.Generator(device="cuda").manual_seed(args.seed)
            params = (torch.rand((n, d), device="cuda", generator=g) * 2 - 1) * (6.0 / (U + d)) ** 0.5
            eng = PropagationEngine(graph, U, I, d, 3, include_layer0=True, deterministic=False, params=params)
            reps = 3 if nnz * d > 4e9 else 20

            def fwd():
                for k in (1, 2):
                    eng.forward_layer(k)

            t = _time_launches(fwd, 2, reps, warm=1)
            res[name] = {"tiles": graph.info()["n_tiles"], "launches": 2 * (reps + 1), "us_per_launch_under_counters": t * 1e6,
                         "n": n, "nnz": nnz, "d": d}
            del eng, params, graph
            torch.cuda.empty_cache()
        except Exception as exc:  # noqa: BLE001 - the other graphs still count
            res[name] = {"error": "%s: %s" % (type(exc).__name__, str(exc)[:200])}
    print(json.dumps({"roofline_only": res}), flush=True)


def collect_dense_counters(pdir, grid_of, counters):
    """rocprofv3's *counter_collection.csv files under pdir -> counters[graph][counter name] = values per dispatch, for
    the dispatches of the PLAIN dense product (spmm_tile_kernel<LPR, NB, 8, true, MINW, EPI = 0, fused>: what the
    --roofline-only child launches) whose grid size names a graph (grid_of: tiles x 256 threads -> graph)."""
    import csv
    import glob
    import re

    for f in glob.glob(os.path.join(pdir, "**", "*counter_collection.csv"), recursive=True):
        with open(f) as fh:
            for row in csv.DictReader(fh):
                if not re.search(r"spmm_tile_kernel<\d+, \d+, 8, true, \d+, 0, (true|false)>", row["Kernel_Name"]):
                    continue
                g = grid_of.get(int(row["Grid_Size"]))
                if g is not None:
                    counters.setdefault(g, {}).setdefault(row["Counter_Name"], []).append(float(row["Counter_Value"]))
    return counters


def attach_measured_traffic(args, out):
    """roofline.traffic MEASURED IN THIS RUN: after every other leg, two child processes

        rocprofv3 --kernel-trace --pmc <counters> --output-format csv -d <tmp> -o p -- python3 bench.py --roofline-only <graphs>

    (the interpreter directly after `--`; one counter set per pass as MI355X_MICROARCH.md prescribes: FETCH_SIZE alone —
    it takes 3 of the 4 TCC slots — then WRITE_SIZE with the L2 hit / miss counters) re-run the dense launch on the CSR
    files the legs left behind.  Per graph: traffic = 2 x FETCH_SIZE + WRITE_SIZE per dispatch (KB units; gfx950 tallies
    its 128-byte read requests at 64 B), averaged over the dispatches whose grid is that graph's tile count.  Children,
    never an exec: this process has initialised the GPU."""
    import shutil
    import subprocess

    exe = shutil.which("rocprofv3")
    root = _pmc_dir(args)
    if exe is None or root is None:
        out["roofline"]["traffic_in_run_error"] = "rocprofv3 not on PATH" if exe is None else "no graph was saved for the children"
        return
    graphs = sorted(os.path.join(root, g) for g in os.listdir(root) if os.path.isdir(os.path.join(root, g)) and g != "_prof")
    t_all = time.perf_counter()
    counters, child_info, errors = {}, None, []
    try:
        for cset in (["FETCH_SIZE"], ["WRITE_SIZE", "TCC_HIT_sum", "TCC_MISS_sum"], ["WRITE_SIZE"]):
            if cset == ["WRITE_SIZE"] and "WRITE_SIZE" in {c for v in counters.values() for c in v}:
                break  # (the combined pass worked)
            pdir = os.path.join(root, "_prof", "_".join(cset))
            cmd = [exe, "--kernel-trace", "--pmc"] + cset + ["--output-format", "csv", "-d", pdir, "-o", "p", "--",
                                                           sys.executable, os.path.abspath(__file__), "--roofline-only",
                                                           ",".join(graphs), "--seed", str(args.seed), "--split", str(args.split)]
            env = dict(os.environ, TMPDIR="/tmp")
            try:
                r = subprocess.run(cmd, cwd="/tmp", env=env, stdin=subprocess.DEVNULL, stdout=subprocess.PIPE,
                                   stderr=subprocess.PIPE, timeout=float(os.environ.get("IDG_BENCH_PMC_TIMEOUT", "240")))
            except subprocess.TimeoutExpired:
                errors.append("pass %s: no result after the timeout" % "+".join(cset))
                continue
            lines = [ln for ln in r.stdout.decode("utf-8", "replace").splitlines() if ln.startswith('{"roofline_only"')]
            if r.returncode != 0 or not lines:
                errors.append("pass %s: exit status %d: %s" % ("+".join(cset), r.returncode, r.stderr.decode("utf-8", "replace")[-200:]))
                continue
            child_info = json.loads(lines[-1])["roofline_only"]
            grid_of = {v["tiles"] * 256: k for k, v in child_info.items() if "tiles" in v}
            collect_dense_counters(pdir, grid_of, counters)
    finally:
        shutil.rmtree(root, ignore_errors=True)
    took = time.perf_counter() - t_all
    free = out["roofline"].get("hbm_reuse_free")
    legs = {"synth-10M": out["roofline"].get("hbm_bound"), "regular-15M-perm": free,
            "regular-15M": free.get("strided") if isinstance(free, dict) else None,
            "synth-1M": out["roofline"].get("cache_boundary")}
    legs[args.workload] = out["roofline"]  # (the headline's own graph)
    legs = {k: [v] for k, v in legs.items() if isinstance(v, dict)}
    for c in out.get("configs") or []:
        if isinstance(c.get("roofline"), dict) and c.get("graph") and "bytes_gather" in c["roofline"]:
            legs.setdefault(c["graph"], []).append(c["roofline"])
    attached = []
    for name, ctr in counters.items():
        for leg in legs.get(name, []):
            if not isinstance(leg, dict) or "FETCH_SIZE" not in ctr or "WRITE_SIZE" not in ctr or "us_per_launch" not in leg:
                continue
            fetch = sum(ctr["FETCH_SIZE"]) / len(ctr["FETCH_SIZE"]) * 1024.0
            write = sum(ctr["WRITE_SIZE"]) / len(ctr["WRITE_SIZE"]) * 1024.0
            traffic = 2.0 * fetch + write
            hit = None
            if "TCC_HIT_sum" in ctr and "TCC_MISS_sum" in ctr:
                h, m = sum(ctr["TCC_HIT_sum"]), sum(ctr["TCC_MISS_sum"])
                hit = h / (h + m) if h + m > 0 else None
            sec = leg["us_per_launch"] * 1e-6
            leg["traffic_from_file"] = {"traffic": leg.get("traffic"), "source": leg.get("traffic_source")}
            leg.update(traffic=traffic, traffic_gbs=traffic / sec / 1e9, frac_traffic=traffic / sec / 1e9 / HBM_PEAK_GBS,
                       traffic_over_bytes_gather=traffic / leg["bytes_gather"] if leg.get("bytes_gather") else None,
                       traffic_l2_hit_rate=hit,
                       traffic_source="measured in this run: rocprofv3 --kernel-trace --pmc FETCH_SIZE | WRITE_SIZE (separate passes, "
                                      "child processes after every other leg), 2 x FETCH_SIZE + WRITE_SIZE (KB units) averaged over %d / "
                                      "%d dispatches of this graph's dense launch; bytes at the L2s' memory side (Infinity-Cache hits "
                                      "included); divided by the launch time measured WITHOUT counters in this run"
                                      % (len(ctr["FETCH_SIZE"]), len(ctr["WRITE_SIZE"])))
            if name not in attached:
                attached.append(name)
    out["roofline"]["traffic_in_run"] = {"graphs": attached, "seconds": took, "errors": errors,
                                         "child": child_info}


def reuse_free_leg(args, d=64):
    """The dense launch where NOTHING can be reused (synth-10M's Zipf(0.8) item popularity keeps hub rows in
    the 256 MiB Infinity Cache, so its bytes at the L2s' memory side are an upper bound of DRAM bytes): regular bipartite
    graphs — 10 M users x 5 M items, every user exactly 20 items, every item exactly 40 users — 15 M rows, 4e8 entries,
    gathered panel 3.84 GB = 14 x the Infinity Cache.  Every gathered row comes from DRAM, so fabric bytes ~ DRAM bytes ~
    SURVEY 8d's gather bytes and `frac` is a fraction of the HBM peak in the plain sense.  TWO access patterns, side by
    side: `strided` — idgrec_amd.synth.regular_adjacency: interaction t joins item (t P) mod I, consecutive
    gathered rows a constant 564,239 rows apart, perfectly even over channels and banks: a strided sweep — and, the
    figure this leg reports at its top level, `random` — the same graph with both id ranges relabelled by random
    permutations (regular_adjacency_permuted): same degrees, a user's 20 items uniformly scattered ids."""
    import idgrec_amd.ops as ops
    import idgrec_amd.synth as S

    U, I, D = 10_000_000, 5_000_000, 20
    n = U + I
    legs = {}
    for name, tag, make in (("random", "regular-15M-perm", lambda: S.regular_adjacency_permuted(U, I, D, seed=args.seed)),
                            ("strided", "regular-15M", lambda: S.regular_adjacency(U, I, D))):
        t0 = time.perf_counter()
        ip, ix, dv = make()
        t_gen = time.perf_counter() - t0
        nnz = len(ix)
        if name == "random":
            _save_csr(args, tag, ip, ix, dv, U, I, d)  # (the traffic children measure the random pattern; the strided one
            #                                             keeps profiles/r05/traffic_regular-15M_d64.json)
        graph = ops.Graph(ip, ix, dv, n, n, split_threshold=args.split)
        del ip, ix, dv
        r = dense_launch_leg(args, tag, d=d, K=3, graph=graph, shape=(U, I, n, nnz, nnz // 2), reps=5)
        r["workload"] = "regular bipartite graph, no reuse, %s neighbours: %d users x %d items, every user %d items, every item " \
                        "%d users; nnz(A)=%d, d=%d (gathered panel %.0f MB = %.1f x the Infinity Cache); drawn in %.1f s" \
                        % ("uniformly scattered (both id ranges relabelled by random permutations)" if name == "random" else
                           "in arithmetic progression (item (t P) mod I: a constant stride between consecutive gathered rows)",
                           U, I, D, U * D // I, nnz, d, 4 * n * d / 1e6, 4 * n * d / INFINITY_CACHE_BYTES, t_gen)
        r["bound"] = "hbm (no row is re-read while it could still be cached: fabric bytes ~ DRAM bytes)"
        r["access_pattern"] = name
        legs[name] = r
        del graph
        torch.cuda.empty_cache()
    out = legs["random"]
    out["strided"] = legs["strided"]
    return out


def other_configs(args, wl_yelp, cpu_threads):
    """One short measured leg for each of the other single-GPU BASELINE configs (the driver's line covered one
    config): [0] MFBPR d=64 yelp2018 B=2048 (configure/MFBPR.txt), [2] LightGCN-3 d=64 amazon-book B=1024
    (configure/LightGCN.txt), [3] SimGCL-3 d=64 amazon-book B=2048 (configure/SimGCL.txt; models/SimGCL.py:62-90).  Each:
    100 timed steps after 20 warm-up steps through the same engine path as the headline, the dominant kernel's launch
    time and roofline fractions, and a CPU leg of a few seconds (the headline's thread count, no probing)."""
    import argparse

    import idgrec_amd.ops as ops
    import idgrec_amd.synth as S
    from idgrec_amd.engine import PropagationEngine

    res = []
    steps, warm = 100, 20
    amazon = {}

    def leg(index, model, workload, batch):
        a = argparse.Namespace(**vars(args))
        a.model, a.workload, a.batch, a.steps, a.warmup, a.steady_steps, a.cpu_seconds = model, workload, batch, steps, warm, 0, 3.0
        t0 = time.perf_counter()
        if workload == "yelp2018":
            wl = dict(wl_yelp)
            wl["triples"] = S.draw_triples(a.seed, wl["users"], wl["items"], wl["U"], wl["I"], (steps + warm) * batch)[0]
        else:
            if "wl" not in amazon:
                a.batch = 2048  # (triples for the larger of the two amazon-book legs)
                amazon["wl"] = build_workload(a, 0, 1)
                a.batch = batch
            wl = amazon["wl"]
        U, I, d, K, B = wl["U"], wl["I"], a.dim, a.layers, batch
        n, nnz = U + I, len(wl["indices"])
        graph = None
        if model != "MFBPR":
            if "graph" not in amazon:
                amazon["graph"] = ops.Graph(wl["indptr"], wl["indices"], wl["values"], n, n, split_threshold=a.split)
                _save_csr(args, workload, wl["indptr"], wl["indices"], wl["values"], U, I, d)
            graph = amazon["graph"]
        W0 = S.xavier_uniform_panel(U, I, d, a.seed)
        eng = PropagationEngine(graph, U, I, d, K, include_layer0=(model != "SimGCL"), reg_lambda=1e-4, lr=1e-3, params=W0.cuda())
        if model == "SimGCL":
            import utility.utility_function.tools as tools

            torch.cuda.manual_seed(a.seed)
            c = tools.read_configuration(os.path.join(ROOT, "configure", "SimGCL.txt"), "SimGCL")
            eng.ssl = (float(c["epsilon"]), float(c["temperature"]), float(c["ssl_lambda"]))
        eng.store_grad = False
        tri = torch.from_numpy(wl["triples"][: (steps + warm) * B]).cuda()
        cols = [tri[:, c].contiguous() for c in range(3)]
        batches = [tuple(c[i * B:(i + 1) * B] for c in cols) for i in range(steps + warm)]
        losses = torch.zeros((steps + warm, 3 if model == "SimGCL" else 2), dtype=torch.float32, device="cuda")
        rows = list(losses.unbind(0))

        def step(i):
            if i + 1 < steps + warm:
                eng.prefetch(*batches[i + 1])
            eng.train_step(*batches[i], loss_out=rows[i])

        for i in range(warm):
            step(i)
        torch.cuda.synchronize()
        t1 = time.perf_counter()
        for i in range(warm, warm + steps):
            step(i)
        torch.cuda.synchronize()
        dt = time.perf_counter() - t1
        one = {"baseline_config": index, "metric": "BPR triples/sec, %s%s dim=%d" % (model, "" if model == "MFBPR" else "-%d" % K, d),
               "workload": "%s-shape graph: %d users x %d items, %d train edges; %s d=%d B=%d" % (workload, U, I, wl["E"], model, d, B),
               "graph": workload, "ms_per_step": dt / steps * 1e3, "value": B * steps / dt, "unit": "triples/s", "steps": steps,
               "warmup": warm, "loss_first_last": [float(losses[warm].sum().item()), float(losses[-1].sum().item())]}
        if graph is not None:
            r = roofline(a, eng, graph, n, nnz, d, K, workload, in_step_form=(model == "LightGCN"),
                         bitmap=eng.touched if model == "LightGCN" else None, reps=10)
            one["dominant_kernel_us"] = r["us_per_launch"]
            one["roofline"] = {k: r[k] for k in ("bound", "kernel", "achieved", "peak", "unit", "frac", "traffic", "traffic_source",
                                                   "frac_traffic", "us_per_launch", "bytes_gather", "bytes_min", "frac_bytes_min",
                                                   "cache_resident", "tiles")}
        else:
            # no propagation: the step is the fused BPR (latency-bound gathers of 3B rows) and the dense Adam update, which
            # streams the three [n, d] panels in and out: 24 B per element
            import ctypes as C

            from idgrec_amd import native

            bm = eng.touched
            call = lambda: native.check(native.lib.idg_adam_rows_f32(  # noqa: E731
                eng.params.data_ptr(), eng.grad.data_ptr(), bm.data_ptr(), eng.exp_avg.data_ptr(), eng.exp_avg_sq.data_ptr(), n, d,
                C.c_double(0.0), C.c_double(0.9), C.c_double(0.999), C.c_double(1e-8), 1, ops._stream()), "idg_adam_rows_f32")
            sec = _time_launches(call, 1, 20)
            nbytes = 24 * n * d
            one["dominant_kernel_us"] = sec * 1e6
            one["roofline"] = {"bound": "l2+infinity-cache streaming (three %.0f MB panels: cache resident)" % (4 * n * d / 1e6),
                               "kernel": "adam_rows_kernel (dense Adam, gradient read at the batch's rows only)", "achieved": nbytes / sec / 1e9,
                               "peak": HBM_PEAK_GBS, "unit": "GB/s", "frac": nbytes / sec / 1e9 / HBM_PEAK_GBS, "traffic": None,
                               "us_per_launch": sec * 1e6, "bytes": nbytes, "cache_resident": True}
        if not args.no_cpu_baseline:
            one["cpu_baseline"] = cpu_baseline(a, wl, W0.numpy(), threads=cpu_threads or 32, warm=(model != "SimGCL"))
        one["seconds"] = time.perf_counter() - t0
        del eng, tri, cols, batches, losses, rows
        torch.cuda.empty_cache()
        return one

    for index, model, workload, batch in ((0, "MFBPR", "yelp2018", 2048), (2, "LightGCN", "amazon-book", 1024),
                                          (3, "SimGCL", "amazon-book", 2048)):
        try:
            res.append(leg(index, model, workload, batch))
        except Exception as exc:  # noqa: BLE001 - the headline stands without it
            res.append({"baseline_config": index, "error": "%s: %s" % (type(exc).__name__, str(exc)[:200])})
    amazon.clear()
    torch.cuda.empty_cache()
    return res


def step_bytes(wl, n, nnz, d, K, B, ms_per_step):
    """Bytes of one training step: SURVEY.md §8(d)'s per-step formula (2K dense products + Adam + the mean's write and the
    gradient's read) beside what the step EXECUTES: K - 1 dense forward products, the last forward layer at
    the batch's <= 3B rows only, a first backward product whose input lives on those rows (the whole entry list is read,
    panel rows only where the column is live), K - 2 dense backward products and a last one with the Adam update in its
    epilogue (p, m, v in and out; the finished gradient is not stored).  The restricted products' entries are counted on
    the host for the first timed batch.  Gather-equivalent bytes in both cases: the panel is cache resident, so neither
    figure over the step time is an HBM rate — the line's `roofline` says what the memory side moved."""
    gather, _ = spmm_bytes(n, nnz, d)
    formula = 2 * K * gather + 28 * n * d + 8 * n * d
    U = wl["U"]
    b = wl["triples"][:B]
    rows = np.unique(np.concatenate([b[:, 0], U + b[:, 1], U + b[:, 2]]))
    deg = np.diff(wl["indptr"])
    e_rows = int(deg[rows].sum())                     # entries of the requested output rows (last forward layer)
    live = np.zeros(n, dtype=bool)
    live[rows] = True
    e_live = int(live[wl["indices"]].sum())           # entries whose INPUT row is live (first backward product)
    restricted_fwd = 8 * e_rows + 4 * d * e_rows + 4 * d * len(rows)
    sparse_bwd = 4 * (n + 1) + 8 * nnz + 4 * d * e_live + 4 * n * d
    adam_epilogue = gather - 4 * n * d + 24 * n * d   # the product's gathers; p, m, v in and out instead of the gradient's store
    dense = (K - 1) + max(K - 2, 0)
    executed = dense * gather + restricted_fwd + sparse_bwd + adam_epilogue
    return {"formula_8d": formula, "executed": executed,
            "executed_breakdown": {"dense_products": dense, "bytes_per_dense_product": gather, "row_restricted_last_forward_layer": restricted_fwd,
                                   "sparse_input_first_backward_product": sparse_bwd, "last_backward_product_with_adam_epilogue": adam_epilogue,
                                   "batch_rows": int(len(rows)), "entries_of_batch_rows": e_rows, "entries_with_live_input": e_live},
            "formula_8d_over_step_time_gbs": formula / (ms_per_step * 1e-3) / 1e9,
            "executed_over_step_time_gbs": executed / (ms_per_step * 1e-3) / 1e9,
            "what": "gather-equivalent bytes (a panel row counted once per stored entry that reads it): SURVEY 8(d)'s per-step formula "
                    "assumes 2K dense products; the step runs %d dense ones, two restricted to the batch's rows (exact) and one with "
                    "the Adam update in its epilogue" % dense}


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
        _save_csr(args, workload, ip, ix, dv, U, I, d)
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
