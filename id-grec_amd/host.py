"""Host-side entry points of libidgrec.so as numpy-in / numpy-out functions: the MT19937
stream, BPR negative sampler, epoch permutation, rating-file parser and the normalised
adjacency builder.  No GPU needed.

The reference draws from NumPy's *global* legacy generator (np.random.seed in
utility/utility_function/tools.py:10, np.random.randint in
utility/utility_data/data_loader.py:120, np.random.shuffle in tools.py:42).  `GlobalStream`
keeps that contract: it lifts the MT19937 state out of np.random, lets the native code
advance it, and writes it back — so native and NumPy consumers share one stream.
"""
import ctypes as C

import numpy as np

from . import native
from .native import check, lib, np_ptr


class Rng:
    """Owned MT19937 stream, bit-compatible with np.random.RandomState(seed)."""

    def __init__(self, seed=0):
        h = C.c_void_p()
        check(lib.idg_rng_create(C.c_uint32(int(seed) & 0xFFFFFFFF), C.byref(h)), "idg_rng_create")
        self._h = h

    def __del__(self):
        h, self._h = getattr(self, "_h", None), None
        if h and lib is not None:
            lib.idg_rng_destroy(h)

    def get_state(self):
        key = np.empty(624, dtype=np.uint32)
        pos = C.c_int32()
        check(lib.idg_rng_get_state(self._h, np_ptr(key, C.c_uint32), C.byref(pos)), "idg_rng_get_state")
        return key, int(pos.value)

    def set_state(self, key, pos):
        key = np.ascontiguousarray(key, dtype=np.uint32)
        if key.shape != (624,):
            raise ValueError("MT19937 key must have 624 words")
        check(lib.idg_rng_set_state(self._h, np_ptr(key, C.c_uint32), int(pos)), "idg_rng_set_state")

    def bytes(self, n):
        out = np.empty(int(n), dtype=np.uint8)
        check(lib.idg_rng_bytes(self._h, int(n), out.ctypes.data_as(C.c_void_p)), "idg_rng_bytes")
        return out.tobytes()

    def randint(self, high, count):
        out = np.empty(int(count), dtype=np.int64)
        check(lib.idg_rng_randint(self._h, int(high), int(count), np_ptr(out, C.c_int64)), "idg_rng_randint")
        return out

    def sample_epoch(self, train_user, train_item, pos_indptr, pos_indices, num_items):
        train_user = np.ascontiguousarray(train_user, dtype=np.int64)
        train_item = np.ascontiguousarray(train_item, dtype=np.int64)
        pos_indptr = np.ascontiguousarray(pos_indptr, dtype=np.int64)
        pos_indices = np.ascontiguousarray(pos_indices, dtype=np.int32)
        E = train_user.shape[0]
        if train_item.shape[0] != E:
            raise ValueError("train_user and train_item differ in length")
        out = np.empty((E, 3), dtype=np.int64)
        cnt = C.c_int64()
        check(lib.idg_sample_epoch(self._h, np_ptr(train_user, C.c_int64), np_ptr(train_item, C.c_int64), E,
                                   np_ptr(pos_indptr, C.c_int64), np_ptr(pos_indices, C.c_int32),
                                   pos_indptr.shape[0] - 1, int(num_items), np_ptr(out, C.c_int64),
                                   C.byref(cnt)), "idg_sample_epoch")
        return out[: cnt.value]

    def shuffle_perm(self, n):
        out = np.empty(int(n), dtype=np.int64)
        check(lib.idg_shuffle_perm(self._h, int(n), np_ptr(out, C.c_int64)), "idg_shuffle_perm")
        return out


class GlobalStream:
    """Context manager: run native draws on NumPy's global legacy stream."""

    def __init__(self):
        self._rng = Rng(0)

    def __enter__(self):
        st = np.random.get_state()
        if st[0] != "MT19937":
            raise RuntimeError("np.random global generator is not MT19937")
        self._tail = st[3:]
        self._rng.set_state(st[1], st[2])
        return self._rng

    def __exit__(self, *exc):
        key, pos = self._rng.get_state()
        # has_gauss / cached_gaussian are untouched by integer draws
        np.random.set_state(("MT19937", key, pos) + tuple(self._tail))
        return False


def py_random_sample(n, k):
    """random.sample(range(n), k) — same values, same final state of Python's `random` module — computed natively
    (the module's MT19937 state is loaded into an idg_rng, advanced there and written back).  int64 array [k]."""
    import random
    from math import ceil, log

    n, k = int(n), int(k)
    if not 0 <= k <= n:
        raise ValueError("Sample larger than population or is negative")
    setsize = 21
    if k > 5:
        setsize += 4 ** ceil(log(k * 3, 4))  # random.py's own expression decides the branch
    version, internal, gauss = random.getstate()
    if version != 3 or len(internal) != 625:
        raise RuntimeError("unexpected random.getstate() layout")
    rng = Rng(0)
    rng.set_state(np.asarray(internal[:624], dtype=np.uint32), int(internal[624]))
    out = np.empty(k, dtype=np.int64)
    check(lib.idg_py_random_sample(rng._h, n, k, int(n <= setsize), np_ptr(out, C.c_int64)), "idg_py_random_sample")
    key, pos = rng.get_state()
    random.setstate((version, tuple(int(x) for x in key) + (pos,), gauss))
    return out


def parse_ratings(path, counts=False):
    """Native Data.read_ratings (data_loader.py:48-70).
    Returns (users[E], items[E], line_users[L], max_user, max_item); with counts=True also
    line_counts[L] (items per line) as a sixth element."""
    h = C.c_void_p()
    ne, nl, mu, mi = C.c_int64(), C.c_int64(), C.c_int64(), C.c_int64()
    check(lib.idg_ratings_open(str(path).encode(), C.byref(h), C.byref(ne), C.byref(nl), C.byref(mu), C.byref(mi)),
          "idg_ratings_open")
    try:
        users = np.empty(ne.value, dtype=np.int64)
        items = np.empty(ne.value, dtype=np.int64)
        lines = np.empty(nl.value, dtype=np.int64)
        cnts = np.empty(nl.value, dtype=np.int64)
        check(lib.idg_ratings_read(h, np_ptr(users, C.c_int64), np_ptr(items, C.c_int64), np_ptr(lines, C.c_int64),
                                   np_ptr(cnts, C.c_int64)), "idg_ratings_read")
    finally:
        lib.idg_ratings_destroy(h)
    if counts:
        return users, items, lines, int(mu.value), int(mi.value), cnts
    return users, items, lines, int(mu.value), int(mi.value)


def build_norm_adj(num_users, num_items, users, items, self_loops=False, numpy_power=True):
    """Native sparse_adjacency_matrix[_with_self] (data_graph.py:7-55).
    Returns CSR (indptr int64[n+1], indices int32[nnz], values float32[nnz]).

    numpy_power=True forms d^-1/2 with the reference's own expression
    (np.power(rowsum, -0.5), float32 without self loops / float64 with) so the values carry
    the same bits as scipy's build on this machine; False lets the library use the correctly
    rounded 1/sqrt(d)."""
    users = np.ascontiguousarray(users, dtype=np.int64)
    items = np.ascontiguousarray(items, dtype=np.int64)
    E = users.shape[0]
    if items.shape[0] != E:
        raise ValueError("users and items differ in length")
    U, I = int(num_users), int(num_items)
    dinv_p = None
    if numpy_power:
        if E and (users.min() < 0 or users.max() >= U or items.min() < 0 or items.max() >= I):
            raise ValueError("edge endpoint outside [0,num_users) x [0,num_items)")
        deg = np.concatenate([np.bincount(users, minlength=U), np.bincount(items, minlength=I)])
        deg = (deg + 1).astype(np.float64) if self_loops else deg.astype(np.float32)
        with np.errstate(divide="ignore"):
            dinv = np.power(deg, -0.5)  # data_graph.py:47 / :22
        dinv[np.isinf(dinv)] = 0.0
        dinv = np.ascontiguousarray(dinv, dtype=np.float64)
        dinv_p = np_ptr(dinv, C.c_double)
    nnz = C.c_int64()
    up, ip = np_ptr(users, C.c_int64), np_ptr(items, C.c_int64)
    # one call into arrays sized for the case without duplicate pairs (2E entries + the diagonal): the size query
    # (NULL outputs) would run the whole sort + de-duplication pass a second time
    cap = 2 * E + (U + I if self_loops else 0)
    indptr = np.empty(U + I + 1, dtype=np.int64)
    indices = np.empty(cap, dtype=np.int32)
    values = np.empty(cap, dtype=np.float32)
    check(lib.idg_build_norm_adj(U, I, E, up, ip, int(bool(self_loops)), dinv_p, C.byref(nnz),
                                 np_ptr(indptr, C.c_int64), np_ptr(indices, C.c_int32), np_ptr(values, C.c_float)),
          "idg_build_norm_adj")
    if nnz.value != cap:  # duplicate pairs were merged
        indices, values = indices[:nnz.value].copy(), values[:nnz.value].copy()
    return indptr, indices, values


__all__ = ["Rng", "GlobalStream", "parse_ratings", "build_norm_adj", "native"]
