"""Host half of libidgrec.so (C++ through the C ABI) against the reference goldens and
against NumPy's own legacy generator."""
import os
import re

import numpy as np
import pytest

import idgrec_amd.host as H
from idgrec_amd import native


def test_abi_exports_every_declared_symbol():
    hdr = open(os.path.join(os.path.dirname(native.LIB_PATH), "..", "..", "include", "idgrec.h")).read()
    declared = set(re.findall(r"\b(idg_[a-z0-9_]+)\s*\(", hdr))
    assert declared, "no declarations parsed"
    for name in sorted(declared):
        assert hasattr(native.lib, name), "libidgrec.so does not export %s" % name
    assert declared == set(native.PROTOTYPES), declared ^ set(native.PROTOTYPES)
    version = int(re.search(r"#define IDG_VERSION (\d+)", hdr).group(1))
    assert native.lib.idg_version() == native.ABI_VERSION == version  # header, binding and built library agree


def test_errors_are_reported_not_thrown():
    with pytest.raises(native.IdgError) as e:
        H.Rng(1).randint(0, 3)
    assert "high must be > 0" in str(e.value)
    with pytest.raises(native.IdgError):
        H.parse_ratings("/nonexistent/train.txt")


def test_mt19937_stream_matches_numpy(golden_tiny):
    assert H.Rng(2024).bytes(256) == golden_tiny["rng_bytes"].tobytes()
    for seed in (0, 1, 2024, 2**32 - 1):
        assert H.Rng(seed).bytes(4099) == np.random.RandomState(seed).bytes(4099)
    key, pos = H.Rng(5).get_state()
    st = np.random.RandomState(5).get_state()
    assert np.array_equal(key, st[1]) and pos == st[2]


@pytest.mark.parametrize("n", [1, 2, 3, 40, 38048, 65536, 65537, 91599, 5_000_000, 2**32 - 1, 2**32, 2**32 + 5, 2**40])
def test_randint_masked_rejection(n):
    rs = np.random.RandomState(7)
    assert np.array_equal(H.Rng(7).randint(n, 500), np.array([rs.randint(0, n) for _ in range(500)]))


@pytest.mark.parametrize("n", [0, 1, 2, 5, 1000, 65537])
def test_shuffle_matches_numpy(n):
    rs = np.random.RandomState(11)
    p = np.arange(n)
    rs.shuffle(p)
    r = H.Rng(11)
    assert np.array_equal(r.shuffle_perm(n), p)
    assert r.randint(1000, 5).tolist() == [rs.randint(0, 1000) for _ in range(5)]  # stream position


@pytest.mark.parametrize("gname", ["tiny", "small"])
def test_sampler_and_shuffle_vs_reference(gname, golden_tiny, golden_small):
    g = golden_tiny if gname == "tiny" else golden_small
    r = H.Rng(2024)
    args = (g["train_user"], g["train_item"], g["pos_indptr"], g["pos_indices"], int(g["num_items"]))
    s1 = r.sample_epoch(*args)
    p1 = r.shuffle_perm(len(s1))
    s2 = r.sample_epoch(*args)
    p2 = r.shuffle_perm(len(s2))
    assert np.array_equal(s1, g["sample1"]) and np.array_equal(p1, g["perm1"])
    assert np.array_equal(s2, g["sample2"]) and np.array_equal(p2, g["perm2"])


def test_global_stream_is_shared_with_numpy(golden_small):
    g = golden_small
    np.random.seed(2024)
    with H.GlobalStream() as r:
        s1 = r.sample_epoch(g["train_user"], g["train_item"], g["pos_indptr"], g["pos_indices"], int(g["num_items"]))
    idx = np.arange(len(s1))
    np.random.shuffle(idx)  # numpy continues the very same stream
    assert np.array_equal(s1, g["sample1"]) and np.array_equal(idx, g["perm1"])


def test_sampler_skips_users_without_positives():
    # user 1 has no positives: its edges are skipped and draw nothing (data_loader.py:114-115)
    indptr = np.array([0, 2, 2, 3], dtype=np.int64)
    indices = np.array([0, 3, 1], dtype=np.int32)
    tu = np.array([0, 1, 2, 0], dtype=np.int64)
    ti = np.array([0, 2, 1, 3], dtype=np.int64)
    out = H.Rng(3).sample_epoch(tu, ti, indptr, indices, 5)
    assert out[:, 0].tolist() == [0, 2, 0]
    assert all(out[k, 2] not in {0: (0, 3), 2: (1,)}[out[k, 0]] for k in range(3))
    with pytest.raises(native.IdgError):  # a user who saw everything can never get a negative
        H.Rng(3).sample_epoch(np.array([0]), np.array([0]), np.array([0, 2]), np.array([0, 1], dtype=np.int32), 2)


@pytest.mark.parametrize("gname", ["tiny", "small"])
def test_parser_vs_reference(gname, tmp_path, golden_tiny, golden_small):
    g = golden_tiny if gname == "tiny" else golden_small
    for split in ("train", "test"):
        p = tmp_path / (split + ".txt")
        p.write_bytes(g[split + "_txt"].tobytes())
        users, items, lines, mu, mi = H.parse_ratings(p)
        assert np.array_equal(users, g[split + "_user"]) and np.array_equal(items, g[split + "_item"])
    assert mu + 1 <= int(g["num_users"]) and mi + 1 <= int(g["num_items"])


def test_parser_edge_cases(tmp_path):
    p = tmp_path / "r.txt"
    p.write_text("3 1 2\n7\n0 5\n")  # user 7 has no items: counted as a line, emits no edge
    users, items, lines, mu, mi = H.parse_ratings(p)
    assert users.tolist() == [3, 3, 0] and items.tolist() == [1, 2, 5]
    assert lines.tolist() == [3, 7, 0] and (mu, mi) == (3, 5)
    p.write_text("1 2 x\n")
    with pytest.raises(native.IdgError):
        H.parse_ratings(p)
    p.write_text("")
    users, items, lines, mu, mi = H.parse_ratings(p)
    assert len(users) == 0 and (mu, mi) == (-1, -1)


@pytest.mark.parametrize("gname", ["tiny", "small"])
def test_adjacency_vs_reference(gname, golden_tiny, golden_small):
    g = golden_tiny if gname == "tiny" else golden_small
    U, I = int(g["num_users"]), int(g["num_items"])
    ip, ix, dv = H.build_norm_adj(U, I, g["train_user"], g["train_item"])
    assert np.array_equal(ip, g["adj_indptr"]) and np.array_equal(ix, g["adj_indices"])
    assert np.array_equal(dv, g["adj_data"])  # bit-exact, incl. the duplicated pair on `tiny`
    ip, ix, dv = H.build_norm_adj(U, I, g["train_user"], g["train_item"], self_loops=True)
    assert np.array_equal(ip, g["adjself_indptr"]) and np.array_equal(ix, g["adjself_indices"])
    assert np.array_equal(dv, g["adjself_data"])
    # library-default d^-1/2 (correctly rounded) differs from numpy's SIMD power by <= 1 ulp per factor
    _, _, dv2 = H.build_norm_adj(U, I, g["train_user"], g["train_item"], numpy_power=False)
    np.testing.assert_allclose(dv2, g["adj_data"], rtol=3e-7)


def test_adjacency_isolated_nodes_and_empty():
    ip, ix, dv = H.build_norm_adj(3, 4, np.array([0, 0, 2]), np.array([1, 3, 1]))
    assert ip.tolist() == [0, 2, 2, 3, 3, 5, 5, 6]  # user 1, items 0 and 2 are isolated: empty rows
    assert np.isfinite(dv).all()
    ip, ix, dv = H.build_norm_adj(2, 2, np.array([], dtype=np.int64), np.array([], dtype=np.int64))
    assert ip.tolist() == [0, 0, 0, 0, 0] and len(ix) == 0


@pytest.mark.parametrize("n,k", [(10, 3), (10, 10), (100, 90), (1000, 5), (1000, 6), (5000, 40), (100000, 90000),
                                 (100000, 10), (1, 1), (0, 0), (77, 21), (1 << 20, 3)])
def test_py_random_sample_is_pythons_random_sample(n, k):
    """tools.create_adj_mat keeps int((1 - ssl_rate) * E) edges chosen by random.sample (tools.py:80): the native
    restatement returns the same list and leaves Python's `random` module in the same state (both branches of
    random.py: pool and rejection-by-set)."""
    import random

    for seed in (1, 2024):
        random.seed(seed)
        want, nxt = random.sample(range(n), k), random.random()
        random.seed(seed)
        got = H.py_random_sample(n, k)
        assert got.tolist() == want and random.random() == nxt
    with pytest.raises(ValueError):
        H.py_random_sample(3, 4)



def test_comm_entry_points_fail_loudly_without_rccl_or_communicator():
    """The RCCL communicator is opened at run time: a wrong path and a missing communicator are errors with a
    message, never a crash or a silent no-op."""
    from idgrec_amd import native

    rc = native.lib.idg_comm_load(b"/nonexistent/librccl.so")
    assert rc == -6 and b"dlopen" in native.lib.idg_last_error()
    rc = native.lib.idg_allreduce_f32(None, None, 4, 0, None)
    assert rc == -1 and b"communicator" in native.lib.idg_last_error()
    rc = native.lib.idg_allgather_f32(None, None, None, 4, None)
    assert rc == -1 and b"communicator" in native.lib.idg_last_error()



def test_bench_triples_are_deterministic_and_cover_both_sampling_paths():
    """synth.draw_triples: every rank of a multi-GPU bench derives the same global sequence from the seed; a run that
    consumes far fewer triples than the graph has edges samples negatives for a uniform subset of the edges only."""
    import idgrec_amd.synth as S

    U, I, E = 300, 200, 6000
    users, items = S.generate(U, I, E, seed=0)
    a = S.draw_triples(7, users, items, U, I, 2 * len(users))       # more than one epoch: whole-epoch path, twice
    b = S.draw_triples(7, users, items, U, I, 2 * len(users))
    assert np.array_equal(a[0], b[0]) and len(a[0]) >= 2 * len(users)
    c = S.draw_triples(7, users, items, U, I, 64)                    # 8 * need < E: subset path
    d = S.draw_triples(7, users, items, U, I, 64)
    assert np.array_equal(c[0], d[0]) and 64 <= len(c[0]) == 128
    pos = {(int(u), int(i)) for u, i in zip(users, items)}
    for t in (a[0], c[0]):
        assert all((int(u), int(p)) in pos for u, p, _ in t[:500])          # positives are train edges
        assert all((int(u), int(n)) not in pos for u, _, n in t[:500])      # negatives are not


def test_rating_file_numbers_beyond_64_bits_are_refused(tmp_path):
    """A token of more digits than int64 holds used to wrap (signed overflow, found under UBSan): now an I/O error naming the
    line — the reference's int() would take it, but no data set has such ids and a wrapped id is a silent wrong answer."""
    p = tmp_path / "train.txt"
    p.write_text("0 1 2\n1 99999999999999999999999999999999\n")
    with pytest.raises(Exception) as e:
        H.parse_ratings(str(p))
    assert "line 2" in str(e.value) and "64 bits" in str(e.value)
    p.write_text("0 9223372036854775807\n")  # the largest int64 itself still parses
    users, items, lines, mu, mi = H.parse_ratings(str(p))
    assert int(items[0]) == 9223372036854775807 and int(mi) == 9223372036854775807
