"""oracle/oracle.py — TEST INFRASTRUCTURE, NOT PRODUCT CODE.

CPU restatement of the ID-GRec reference's hot path used to check libidgrec.so:
  * plain C (oracle/idg_oracle.c, built by `build()` with gcc) for the fp32 arithmetic,
  * NumPy's own legacy generator for the sampler / shuffle (NumPy *is* the reference's
    arithmetic there: utility/utility_data/data_loader.py:120, utility/utility_function/tools.py:42),
  * SciPy for the adjacency (the reference's own expression, data_graph.py:46-51, on a
    matrix assembled without the 336-second DOK/LIL detour),
  * NumPy for the metrics (utility/utility_function/metrics.py).

Only tests/, __graft_entry__.smoke() and bench.py's cpu_baseline leg may import this module.
Pinned against the imported reference by tests/golden/*.npz (see oracle/gen_golden.py).
"""
import ctypes as C
import os
import subprocess

import numpy as np

_HERE = os.path.dirname(os.path.abspath(__file__))
_SRC = os.path.join(_HERE, "idg_oracle.c")
_OUT = os.path.join(_HERE, "_build")
_LIB = os.path.join(_OUT, "liboracle.so")
_lib = None

_i64p = C.POINTER(C.c_int64)
_i32p = C.POINTER(C.c_int32)
_f32p = C.POINTER(C.c_float)


def build(force=False):
    """gcc -O2, contraction off so the only fused operations are the explicit fmaf() calls."""
    os.makedirs(_OUT, exist_ok=True)
    if force or not os.path.exists(_LIB) or os.path.getmtime(_LIB) < os.path.getmtime(_SRC):
        cmd = ["gcc", "-O2", "-std=c11", "-ffp-contract=off", "-fno-fast-math", "-shared", "-fPIC", "-o", _LIB,
               _SRC, "-lm"]
        subprocess.run(cmd, check=True)
    return _LIB


def _load():
    global _lib
    if _lib is None:
        build()
        _lib = C.CDLL(_LIB)
        for name in ("orc_spmm_f32", "orc_spmm_sched_f32", "orc_propagate_mean_f32", "orc_propagate_mean_bwd_f32",
                     "orc_bpr_f32", "orc_adam_f32", "orc_score_f32"):
            getattr(_lib, name).restype = None
    return _lib


def _p(a, t):
    return a.ctypes.data_as(C.POINTER(t)) if a is not None else None


def _csr(indptr, indices, values):
    return (np.ascontiguousarray(indptr, dtype=np.int64), np.ascontiguousarray(indices, dtype=np.int32),
            np.ascontiguousarray(values, dtype=np.float32))


# ---------------------------------------------------------------------------- fp32 kernels
def spmm(indptr, indices, values, X, long_rows=None, seg_len=None, chunk_len=None):
    indptr, indices, values = _csr(indptr, indices, values)
    X = np.ascontiguousarray(X, dtype=np.float32)
    n, d = indptr.shape[0] - 1, X.shape[1]
    Y = np.empty((n, d), dtype=np.float32)
    L = _load()
    if long_rows is not None and len(long_rows):
        lr = np.ascontiguousarray(long_rows, dtype=np.int64)
        sl = np.ascontiguousarray(seg_len, dtype=np.int64)
        cl = np.ascontiguousarray(np.zeros_like(lr) if chunk_len is None else chunk_len, dtype=np.int64)
        L.orc_spmm_sched_f32(C.c_int64(n), _p(indptr, C.c_int64), _p(indices, C.c_int32), _p(values, C.c_float),
                             _p(X, C.c_float), C.c_int64(d), _p(lr, C.c_int64), _p(sl, C.c_int64), _p(cl, C.c_int64),
                             C.c_int64(len(lr)), _p(Y, C.c_float))
    else:
        L.orc_spmm_f32(C.c_int64(n), _p(indptr, C.c_int64), _p(indices, C.c_int32), _p(values, C.c_float),
                       _p(X, C.c_float), C.c_int64(d), _p(Y, C.c_float))
    return Y


def propagate_mean(indptr, indices, values, E0, K, include_layer0=True, long_rows=None, seg_len=None, chunk_len=None):
    indptr, indices, values = _csr(indptr, indices, values)
    E0 = np.ascontiguousarray(E0, dtype=np.float32)
    n, d = E0.shape
    out = np.empty((n, d), dtype=np.float32)
    tmp = np.empty(2 * n * d, dtype=np.float32)
    nl = 0 if long_rows is None else len(long_rows)
    lr = np.ascontiguousarray(long_rows if nl else [0], dtype=np.int64)
    sl = np.ascontiguousarray(seg_len if nl else [0], dtype=np.int64)
    cl = np.ascontiguousarray(chunk_len if (nl and chunk_len is not None) else np.zeros_like(lr), dtype=np.int64)
    _load().orc_propagate_mean_f32(C.c_int64(n), _p(indptr, C.c_int64), _p(indices, C.c_int32),
                                   _p(values, C.c_float), _p(E0, C.c_float), C.c_int64(d), C.c_int(K),
                                   C.c_int(int(include_layer0)), _p(lr, C.c_int64), _p(sl, C.c_int64),
                                   _p(cl, C.c_int64), C.c_int64(nl), _p(out, C.c_float), _p(tmp, C.c_float))
    return out


def propagate_mean_bwd(indptr, indices, values, g, K, include_layer0=True):
    indptr, indices, values = _csr(indptr, indices, values)
    g = np.ascontiguousarray(g, dtype=np.float32)
    n, d = g.shape
    out = np.empty((n, d), dtype=np.float32)
    tmp = np.empty(3 * n * d, dtype=np.float32)
    _load().orc_propagate_mean_bwd_f32(C.c_int64(n), _p(indptr, C.c_int64), _p(indices, C.c_int32),
                                       _p(values, C.c_float), _p(g, C.c_float), C.c_int64(d), C.c_int(K),
                                       C.c_int(int(include_layer0)), _p(out, C.c_float), _p(tmp, C.c_float))
    return out


def bpr(fin, ego, num_users, users, pos, neg, reg_lambda, want_grad=True):
    fin = np.ascontiguousarray(fin, dtype=np.float32)
    ego = np.ascontiguousarray(ego, dtype=np.float32)
    users = np.ascontiguousarray(users, dtype=np.int64)
    pos = np.ascontiguousarray(pos, dtype=np.int64)
    neg = np.ascontiguousarray(neg, dtype=np.int64)
    n, d = fin.shape
    loss = np.zeros(2, dtype=np.float32)
    gf = np.zeros((n, d), dtype=np.float32) if want_grad else None
    ge = np.zeros((n, d), dtype=np.float32) if want_grad else None
    _load().orc_bpr_f32(_p(fin, C.c_float), _p(ego, C.c_float), C.c_int64(num_users), _p(users, C.c_int64),
                        _p(pos, C.c_int64), _p(neg, C.c_int64), C.c_int64(len(users)), C.c_int64(d),
                        C.c_float(reg_lambda), _p(loss, C.c_float), _p(gf, C.c_float), _p(ge, C.c_float))
    return loss, gf, ge


def adam(p, g, m, v, lr, step, beta1=0.9, beta2=0.999, eps=1e-8):
    """In place on p, m, v (contiguous float32 arrays)."""
    _load().orc_adam_f32(_p(p, C.c_float), _p(g, C.c_float), _p(m, C.c_float), _p(v, C.c_float),
                         C.c_int64(p.size), C.c_double(lr), C.c_double(beta1), C.c_double(beta2), C.c_double(eps),
                         C.c_int64(step))


def score(U, V, users, apply_sigmoid=True):
    U = np.ascontiguousarray(U, dtype=np.float32)
    V = np.ascontiguousarray(V, dtype=np.float32)
    users = np.ascontiguousarray(users, dtype=np.int64)
    Bt, I, d = len(users), V.shape[0], V.shape[1]
    R = np.empty((Bt, I), dtype=np.float32)
    _load().orc_score_f32(_p(U, C.c_float), _p(V, C.c_float), _p(users, C.c_int64), C.c_int64(Bt), C.c_int64(I),
                          C.c_int64(d), C.c_int(int(apply_sigmoid)), _p(R, C.c_float))
    return R


# ------------------------------------------------------------------- sampler / shuffle
def sample_epoch(train_user, train_item, all_positive, num_items):
    """Data.sample_data_to_train_all (data_loader.py:108-127) on np.random's global stream."""
    rows = []
    for i in range(len(train_user)):
        user = train_user[i]
        positive_items = all_positive[user]
        if len(positive_items) == 0:
            continue
        while True:
            negative_item = np.random.randint(0, num_items)
            if negative_item in positive_items:
                continue
            break
        rows.append([user, train_item[i], negative_item])
    return np.array(rows, dtype=np.int64).reshape(-1, 3)


def shuffle_perm(n):
    """tools.shuffle's permutation (tools.py:41-42) on np.random's global stream."""
    idx = np.arange(n)
    np.random.shuffle(idx)
    return idx


# ------------------------------------------------------------------------------ adjacency
def norm_adj(num_users, num_items, users, items, self_loops=False):
    """sparse_adjacency_matrix / _with_self (data_graph.py:7-55) with the reference's own
    normalisation expression; only the assembly of A skips DOK/LIL."""
    import scipy.sparse as sp

    U, I = int(num_users), int(num_items)
    R = sp.csr_matrix((np.ones(len(users)), (users, items)), shape=(U, I))  # data_loader.py:42
    R32 = R.astype(np.float32)
    A = sp.bmat([[None, R32], [R32.T, None]], format="csr", dtype=np.float32)
    if self_loops:
        A = (A + sp.eye(A.shape[0])).tocsr()  # float64, as in data_graph.py:20
    row_sum = np.array(A.sum(axis=1))
    with np.errstate(divide="ignore"):
        d_inv = np.power(row_sum, -0.5).flatten()
    d_inv[np.isinf(d_inv)] = 0.0
    D = sp.diags(d_inv)
    N = D.dot(A).dot(D).tocsr()
    N.sort_indices()
    return N.indptr.astype(np.int64), N.indices.astype(np.int32), N.data.astype(np.float32)


# -------------------------------------------------------------------------------- metrics
def get_label(true_data, pred_data):
    """metrics.get_label (metrics.py:49-58)."""
    return np.array([[float(x in set(t)) for x in p] for t, p in zip(true_data, pred_data)], dtype=float)


def recall_at_k(r, k, test_data):
    hits = r[:, :k].sum(1)
    return float(np.sum(hits / np.array([len(t) for t in test_data])))


def precision_at_k(r, k, test_data):
    return float(np.sum(r[:, :k].sum(1)) / k)


def ndcg_at_k(r, k, test_data):
    pred = r[:, :k]
    ideal = np.zeros((len(pred), k))
    for i, items in enumerate(test_data):
        ideal[i, : min(k, len(items))] = 1
    disc = 1.0 / np.log2(np.arange(2, k + 2))
    idcg = np.sum(ideal * disc, axis=1)
    dcg = np.sum(pred * disc, axis=1)
    idcg[idcg == 0.0] = 1.0
    nd = dcg / idcg
    nd[np.isnan(nd)] = 0.0
    return float(np.sum(nd))


def topk_reference(rating, k):
    """Deterministic statement of torch.topk's contract used by the tests: sort by
    (score descending, item ascending)."""
    order = np.lexsort((np.arange(rating.shape[1])[None, :].repeat(rating.shape[0], 0), -rating), axis=1)
    return order[:, :k]


def topk_is_valid(ref_rating, idx, k, tol=0.0):
    """Tie-aware check: every item whose reference score exceeds the reference's
    k-th best score by more than tol must be present, and nothing returned may score more
    than tol below that k-th score.  Returns (ok, message)."""
    for b in range(ref_rating.shape[0]):
        row = ref_rating[b]
        kth = np.sort(row)[::-1][k - 1]
        got = set(int(x) for x in idx[b])
        if len(got) != k:
            return False, "row %d: duplicate or missing indices" % b
        must = set(np.nonzero(row > kth + tol)[0].tolist())
        if not must <= got:
            return False, "row %d: missing %s" % (b, sorted(must - got)[:5])
        worst = min(row[list(got)])
        if worst < kth - tol:
            return False, "row %d: returned score %g below k-th %g" % (b, worst, kth)
    return True, ""


# ---- 24-bit panels of the sharded step's opt-in exchange (id-grec_amd/csrc/idg_shard.hip: pack24 / unpack24 / reduce24).
# No reference counterpart (the reference trains on one device, utility/utility_train/trainer.py:8-74): these restate the
# library's own published format so that the CPU tests of sharded.Packed24Comm run the same arithmetic.
def top24(x):
    """Upper 24 bits of each fp32 word, the dropped byte rounded to nearest even (a uint32 array of 24-bit values)."""
    b = np.ascontiguousarray(x, dtype=np.float32).view(np.uint32).astype(np.uint64)
    return ((b + 0x7F + ((b >> 8) & 1)) >> 8).astype(np.uint32)


def pack24(x):
    """fp32 values (a multiple of 4 of them) -> 3 words per 4 values: A | B << 24, B >> 8 | C << 16, C >> 16 | D << 8."""
    t = top24(np.asarray(x).reshape(-1)).reshape(-1, 4)
    a, b, c, d = (t[:, j] for j in range(4))
    w = np.stack([a | (b << np.uint32(24)), (b >> np.uint32(8)) | (c << np.uint32(16)), (c >> np.uint32(16)) | (d << np.uint32(8))], 1)
    return w.astype(np.uint32).reshape(-1)


def unpack24(w):
    w = np.ascontiguousarray(w, dtype=np.uint32).reshape(-1, 3)
    w0, w1, w2 = w[:, 0], w[:, 1], w[:, 2]
    a = w0 & np.uint32(0xFFFFFF)
    b = (w0 >> np.uint32(24)) | ((w1 & np.uint32(0xFFFF)) << np.uint32(8))
    c = (w1 >> np.uint32(16)) | ((w2 & np.uint32(0xFF)) << np.uint32(16))
    d = w2 >> np.uint32(8)
    return (np.stack([a, b, c, d], 1).astype(np.uint32) << np.uint32(8)).reshape(-1).view(np.float32)


def reduce24(blocks):
    """Sum of packed blocks IN THE ORDER GIVEN, one fp32 add per block and element (the rank-ordered sum)."""
    acc = unpack24(blocks[0]).copy()
    with np.errstate(invalid="ignore", over="ignore"):
        for b in blocks[1:]:
            acc = (acc + unpack24(b)).astype(np.float32)
    return acc
