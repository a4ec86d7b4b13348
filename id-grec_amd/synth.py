"""Deterministic synthetic interaction graphs of the shapes BASELINE.json names
(SURVEY.md §8d).  The real yelp2018 / amazon-book `train.txt` are not distributable, so
benchmarks and parity tests run on graphs with the same node/edge counts and a comparable
skew: user degree ~ lognormal(0, 1) rescaled to the target mean (min 1, cap I/2), items drawn
from a Zipf-like popularity p_i ~ (rank_i + 1)^-0.8 under a fixed random id permutation,
de-duplicated per user.  The generator is NOT parity-relevant; only its output is.
"""
import os

import numpy as np

SHAPES = {
    # name: (num_users, num_items, train_edges)
    "tiny": (50, 40, 300),
    "small": (300, 250, 3600),
    "medium": (4000, 3000, 120000),
    "yelp2018": (31668, 38048, 1237259),
    "amazon-book": (52643, 91599, 2380730),
    "synth-1M": (1000000, 500000, 20000000),
    "synth-10M": (10000000, 5000000, 200000000),
}


def _draw(num_users, num_items, target, seed, zipf_a):
    rng = np.random.default_rng(seed)
    U, I = int(num_users), int(num_items)
    deg = rng.lognormal(0.0, 1.0, U)
    deg = deg * (target / deg.sum())
    deg = np.clip(np.rint(deg), 1, max(1, I // 2)).astype(np.int64)
    # popularity CDF over ranks, ranks mapped to ids by a fixed permutation
    p = (np.arange(I, dtype=np.float64) + 1.0) ** (-zipf_a)
    cdf = np.cumsum(p)
    cdf /= cdf[-1]
    perm = rng.permutation(I)
    total = int(deg.sum())
    r = rng.random(total)
    if total < (1 << 22):
        users = np.repeat(np.arange(U, dtype=np.int64), deg)
        items = perm[np.searchsorted(cdf, r, side="right").clip(0, I - 1)].astype(np.int64)
        key = np.unique(users * I + items)
        return key // I, key % I
    # The same arithmetic in blocks of whole users on a few threads (NumPy's search, gather and sort release the GIL):
    # draws belong to users in order, so the sorted unique keys of a block of users are a contiguous piece of the
    # sorted unique keys of all of them — identical output, ~8x sooner on the 2e8-edge shape.
    from concurrent.futures import ThreadPoolExecutor

    ends = np.cumsum(deg)
    n_blocks = int(min(U, max(1, total >> 22)))
    cut_users = np.searchsorted(ends, np.linspace(0, total, n_blocks + 1)[1:-1], side="left") + 1
    cut_users = np.unique(np.concatenate([[0], cut_users, [U]]))

    def block(b):
        u0, u1 = int(cut_users[b]), int(cut_users[b + 1])
        e0, e1 = (int(ends[u0 - 1]) if u0 else 0), int(ends[u1 - 1])
        it = perm[np.searchsorted(cdf, r[e0:e1], side="right").clip(0, I - 1)].astype(np.int64)
        us = np.repeat(np.arange(u0, u1, dtype=np.int64), deg[u0:u1])
        return np.unique(us * I + it)

    with ThreadPoolExecutor(max_workers=min(16, os.cpu_count() or 1)) as ex:
        key = np.concatenate(list(ex.map(block, range(len(cut_users) - 1))))
    return key // I, key % I


def generate(num_users, num_items, num_edges, seed=0, zipf_a=0.8, match_edges=True):
    """Returns (users int64[E'], items int64[E']) sorted by (user, item), duplicates removed, every user
    keeps >= 1 item.  Popular items collide, so a plain draw of num_edges pairs loses ~5 % to
    de-duplication; with match_edges the draw is repeated (same seed) with a corrected target until
    E' is within 0.5 % of num_edges (graphs up to 5e7 edges; larger ones take the first draw)."""
    E = int(num_edges)
    target = float(E)
    users, items = _draw(num_users, num_items, target, seed, zipf_a)
    if not match_edges or E > 50_000_000:
        return users, items
    for _ in range(4):
        if abs(len(users) - E) <= 0.005 * E:
            break
        target *= E / max(len(users), 1)
        users, items = _draw(num_users, num_items, target, seed, zipf_a)
    return users, items


def regular_adjacency(num_users, num_items, user_degree, multiplier=None):
    """The normalised adjacency (CSR: indptr int64, indices int32, values float32) of a REGULAR bipartite graph with no
    reuse to exploit: every user has exactly `user_degree` items, every item exactly U * user_degree / I users (which must
    be whole), and neighbours are scattered over the whole id range — interaction t = u * D + j (j < D) joins user u with
    item (t * P) mod I for a multiplier P coprime to I.  Uniform item popularity, constant degree: what bench.py's
    `hbm_reuse_free` leg gathers from, so that the bytes at the L2s' memory side are DRAM bytes and not Infinity-Cache
    hits on hub rows.  Built directly, without an edge list or a sort of 2e8 keys: a user's items are D
    values sorted per row; an item's users are ((t0 + m I) // D, m = 0 .. U D / I - 1) with t0 = i P^-1 mod I, ascending
    as they come.  Every value is 1 / sqrt(D * D_item) (the reference's D^-1/2 A D^-1/2, data_graph.py:46-51, on a graph
    whose degrees are constant)."""
    U, I, D = int(num_users), int(num_items), int(user_degree)
    if U * D % I or D >= I:
        raise ValueError("regular_adjacency: U * D must be a multiple of I and D < I")
    Di = U * D // I
    P = int(multiplier) if multiplier else 2654435761 % I
    import math

    while math.gcd(P, I) != 1:
        P += 1
    Pinv = pow(P, -1, I)
    n = U + I
    indptr = np.empty(n + 1, dtype=np.int64)
    indptr[: U + 1] = np.arange(U + 1, dtype=np.int64) * D
    indptr[U:] = U * D + np.arange(I + 1, dtype=np.int64) * Di
    indices = np.empty(U * D + I * Di, dtype=np.int32)
    # user rows, in blocks of whole users on a few threads (the modular products and the per-row sort release the GIL)
    from concurrent.futures import ThreadPoolExecutor

    def user_block(b):
        u0, u1 = b
        t = np.arange(u0 * D, u1 * D, dtype=np.int64)
        it = ((t % I) * P % I).astype(np.int32).reshape(u1 - u0, D)
        it.sort(axis=1)
        indices[u0 * D: u1 * D] = (it + U).reshape(-1)  # columns of the [n, n] adjacency: items start at U

    def item_block(b):
        i0, i1 = b
        t0 = (np.arange(i0, i1, dtype=np.int64) * Pinv) % I
        us = (t0[:, None] + np.arange(Di, dtype=np.int64)[None, :] * I) // D
        indices[U * D + i0 * Di: U * D + i1 * Di] = us.astype(np.int32).reshape(-1)

    step_u, step_i = max(1, (1 << 22) // D), max(1, (1 << 22) // Di)
    with ThreadPoolExecutor(max_workers=min(16, os.cpu_count() or 1)) as ex:
        list(ex.map(user_block, [(a, min(a + step_u, U)) for a in range(0, U, step_u)]))
        list(ex.map(item_block, [(a, min(a + step_i, I)) for a in range(0, I, step_i)]))
    val = np.float32(np.power(np.float32(D), np.float32(-0.5)) * np.float32(1.0) * np.power(np.float32(Di), np.float32(-0.5)))
    values = np.full(len(indices), val, dtype=np.float32)
    return indptr, indices, values


def regular_adjacency_permuted(num_users, num_items, user_degree, seed=0):
    """regular_adjacency's graph with BOTH id ranges relabelled by random permutations (user u -> sigma[u], item i -> pi[i]).
    Same degrees — every user `user_degree` items, every item U * user_degree / I users, nothing to reuse — but the rows a
    wave gathers are no longer an arithmetic progression: in regular_adjacency interaction t joins item (t P) mod I, so
    consecutive gathered rows lie a constant ~564,239 rows apart, perfectly even over channels and banks (a
    strided sweep, not a random gather).  Here a user's items are `user_degree` uniformly scattered ids and an item's users
    likewise.  The relabelling (two gathers and two row sorts over 2e8 ids) runs on the GPU when there is one."""
    import torch

    U, I, D = int(num_users), int(num_items), int(user_degree)
    indptr, indices, values = regular_adjacency(U, I, D)
    Di = U * D // I
    dev = torch.device("cuda") if torch.cuda.is_available() else torch.device("cpu")
    g = torch.Generator(device=dev).manual_seed(int(seed))
    pi, sigma = torch.randperm(I, device=dev, generator=g), torch.randperm(U, device=dev, generator=g)
    ix = torch.from_numpy(indices)
    out = torch.empty_like(ix)
    # user rows: row sigma[u] = sort(U + pi[items of u]); item rows: row pi[i] = sort(sigma[users of i])
    rows = (pi[ix[: U * D].to(dev).long() - U] + U).to(torch.int32).view(U, D).sort(dim=1).values
    moved = torch.empty_like(rows)
    moved[sigma] = rows
    out[: U * D] = moved.view(-1).cpu()
    del rows, moved
    rows = sigma[ix[U * D:].to(dev).long()].to(torch.int32).view(I, Di).sort(dim=1).values
    moved = torch.empty_like(rows)
    moved[pi] = rows
    out[U * D:] = moved.view(-1).cpu()
    del rows, moved, pi, sigma
    return indptr, out.numpy(), values


GENERATOR_VERSION = 2  # part of generate_shared's cache file name: bump when generate()'s output changes


def _edges_fingerprint(users, items):
    """(E, wrapping int64 sums of the two id arrays and of users*31 + items): cheap (one pass), and any truncated,
    permuted-between-columns or foreign file changes it."""
    with np.errstate(over="ignore"):
        return [int(len(users)), int(users.sum(dtype=np.int64)), int(items.sum(dtype=np.int64)),
                int((users * np.int64(31) + items).sum(dtype=np.int64))]


def _valid_edges(users, items, num_users, num_items):
    """What shard_adjacency_from_edges needs: ids in range, sorted by (user, item), no duplicate pair."""
    if users.ndim != 1 or users.shape != items.shape or users.dtype != np.int64 or items.dtype != np.int64 or len(users) == 0:
        return False
    if users[0] < 0 or users[-1] >= num_users or items.min() < 0 or items.max() >= num_items:
        return False
    du = np.diff(users)
    return not ((du < 0).any() or ((du == 0) & (np.diff(items) <= 0)).any())


def generate_shared(num_users, num_items, num_edges, seed, rank, barrier, cache_dir=None):
    """generate() for the ranks of one multi-process run on one host: rank 0 draws the graph (or finds it already drawn
    by an earlier run of THIS user on this machine) and leaves it in a cache file, the others load it — the 2e8-edge
    shape costs ~45 s and 10 GB of host memory to draw, per process.  `barrier` is the process group's barrier (called
    twice by every rank).  The cache lives in a directory of the calling user's own (mode 0700) and carries a header
    (generator version, shape, edge count, fingerprint of the id arrays); a file is used only if the header matches what
    was asked for AND the arrays match the header AND they are what shard_adjacency_from_edges needs (ids in range,
    sorted by (user, item), no duplicates).  Anything else — no space, unreadable, stale, foreign, truncated — falls
    back to drawing in this process: generate() is deterministic, so every rank ends with the same arrays whichever way
    it got them, and no rank is left behind at the next collective."""
    import json
    import tempfile

    d = cache_dir or os.environ.get("IDG_SYNTH_CACHE")
    if d is None:
        d = os.path.join(tempfile.gettempdir(), "idgrec_synth_%d" % os.getuid())
    try:
        os.makedirs(d, mode=0o700, exist_ok=True)
    except OSError:
        pass
    path = os.path.join(d, "idgrec_synth_v%d_%d_%d_%d_%d.npy" % (GENERATOR_VERSION, num_users, num_items, num_edges, seed))
    want = {"version": GENERATOR_VERSION, "num_users": int(num_users), "num_items": int(num_items),
            "num_edges": int(num_edges), "seed": int(seed)}

    def load():
        try:
            with open(path + ".json") as f:
                head = json.load(f)
            if any(head.get(k) != v for k, v in want.items()):
                return None
            both = np.load(path)
            if both.ndim != 2 or both.shape[0] != 2 or both.dtype != np.int64:
                return None
            users, items = both[0], both[1]
            if _edges_fingerprint(users, items) != head.get("fingerprint") or not _valid_edges(users, items, num_users, num_items):
                return None
            return users, items
        except (OSError, ValueError, KeyError, TypeError):
            return None

    got = None
    barrier()
    if rank == 0:
        got = load()
        if got is None:
            got = generate(num_users, num_items, num_edges, seed=seed)
            try:
                tmp = "%s.%d.tmp" % (path, os.getpid())
                with open(tmp, "wb") as f:
                    np.save(f, np.stack(got))
                with open(tmp + ".json", "w") as f:
                    json.dump(dict(want, fingerprint=_edges_fingerprint(*got)), f)
                os.replace(tmp, path)
                os.replace(tmp + ".json", path + ".json")
            except OSError:
                pass
    barrier()
    if got is None:
        got = load()
    if got is None:
        got = generate(num_users, num_items, num_edges, seed=seed)
    return got


def split_test(users, items, num_users, n_test=1, seed=1):
    """Hold out up to n_test items per user (never a user's last train item).
    Returns train (users, items) and test (users, items)."""
    rng = np.random.default_rng(seed)
    order = np.lexsort((rng.random(len(users)), users))
    u_s, i_s = users[order], items[order]
    start = np.searchsorted(u_s, np.arange(num_users), side="left")
    cnt = np.diff(np.append(start, len(u_s)))
    rank = np.arange(len(u_s)) - np.repeat(start, cnt)
    take = np.minimum(n_test, np.maximum(cnt - 1, 0))
    is_test = rank < np.repeat(take, cnt)
    tr = np.lexsort((i_s[~is_test], u_s[~is_test]))
    te = np.lexsort((i_s[is_test], u_s[is_test]))
    return (u_s[~is_test][tr], i_s[~is_test][tr]), (u_s[is_test][te], i_s[is_test][te])


def write_ratings(path, users, items):
    """The reference's text format: one line per user, `uid i1 i2 ...` (data_loader.py:48-70)."""
    os.makedirs(os.path.dirname(path), exist_ok=True)
    order = np.lexsort((items, users))
    u, i = users[order], items[order]
    bounds = np.nonzero(np.diff(u))[0] + 1
    with open(path, "w") as f:
        for uu, chunk in zip(u[np.r_[0, bounds]] if len(u) else [], np.split(i, bounds)):
            f.write(str(int(uu)) + " " + " ".join(map(str, chunk.tolist())) + "\n")


def make_dataset(root, name, shape=None, seed=0, n_test=1):
    """Create <root>/<name>/{train,test}.txt for a named or explicit shape; returns the dir."""
    U, I, E = SHAPES[name] if shape is None else shape
    users, items = generate(U, I, E, seed=seed)
    (tu, ti), (su, si) = split_test(users, items, U, n_test=n_test, seed=seed + 1)
    d = os.path.join(root, name)
    write_ratings(os.path.join(d, "train.txt"), tu, ti)
    write_ratings(os.path.join(d, "test.txt"), su, si)
    return d


def xavier_uniform_panel(num_users, num_items, d, seed):
    """[U+I, d] torch CPU tensor: nn.init.xavier_uniform_ applied to the user and the item table
    separately (models/LightGCN.py:27-28), from torch's CPU generator seeded with `seed`."""
    import torch

    g = torch.Generator().manual_seed(seed)
    out = torch.empty(num_users + num_items, d)
    for lo, hi in ((0, num_users), (num_users, num_users + num_items)):
        bound = (6.0 / ((hi - lo) + d)) ** 0.5
        out[lo:hi] = (torch.rand(hi - lo, d, generator=g) * 2 - 1) * bound
    return out


def draw_triples(seed, users, items, num_users, num_items, need):
    """`need` (or more) BPR triples for a bench run, drawn with the native sampler exactly as the reference's epoch
    loop draws them (sample every edge, shuffle; further epochs when one does not suffice).  On a graph with far more
    edges than the run consumes, negatives are drawn for a uniform subset of the edges only — the same triple
    distribution as slicing a shuffled epoch.  Deterministic in `seed`: every rank of a multi-GPU run calls this and
    slices the same global sequence.  Returns (triples [>= need, 3] int64, sampler rate in triples/s)."""
    import time

    from . import host as H

    pos_ptr = np.zeros(num_users + 1, dtype=np.int64)
    pos_ptr[1:] = np.cumsum(np.bincount(users, minlength=num_users))
    items32 = items.astype(np.int32)
    rng = H.Rng(seed)
    if len(users) > 8 * need:
        pick = np.sort(np.random.default_rng(seed).choice(len(users), size=2 * need, replace=False))
        su, si = users[pick], items[pick]
    else:
        su, si = users, items
    t_s = time.perf_counter()
    tri = rng.sample_epoch(su, si, pos_ptr, items32, num_items)
    perm = rng.shuffle_perm(len(tri))
    rate = len(tri) / (time.perf_counter() - t_s)
    tri = tri[perm]
    while len(tri) < need:
        t2 = rng.sample_epoch(su, si, pos_ptr, items32, num_items)
        tri = np.concatenate([tri, t2[rng.shuffle_perm(len(t2))]])
    return tri, rate, pos_ptr, items32


def ramp_clocks(seconds=0.5, graph=None, d=64):
    """Half a second of untimed GPU work before a bench's warm-up steps.  A GPU that has idled (a fresh box, or the
    seconds of host-side graph construction) starts in a low power state, and the W warm-up steps of a small workload
    (5 x 0.3 ms at the driver's flags) are over before the clocks are up.  With a graph handle the work is that
    graph's own product on a scratch panel — what the steps will run, so the clocks settle where the steps hold them
    (measured, steps 5-24 after the ramp: 276 us; after a GEMM ramp or none at all 282 us; steady state 274 us;
    scripts/probes/step_trend.py) — otherwise a dense GEMM loop."""
    import time

    import torch

    t0 = time.perf_counter()
    if graph is not None:
        from . import ops

        x = torch.randn(graph.n_cols, d, device="cuda")
        y = torch.empty(graph.n_rows, d, device="cuda")
        while time.perf_counter() - t0 < seconds:
            for _ in range(20):
                ops.spmm_ex_raw(graph, x, Y=y)
            torch.cuda.synchronize()
        return
    a = torch.randn(4096, 4096, device="cuda")
    while time.perf_counter() - t0 < seconds:
        for _ in range(8):
            a = torch.tanh(a @ a * 1e-3)
        torch.cuda.synchronize()
