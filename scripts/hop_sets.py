"""Sizes of the row sets a batch's step reads (host, numpy): the touched items, the near / far users, the two- and
three-hop items of id-grec_amd/sharded.py (_agree_touched_items / _agree_two_hop_items).  usage: hop_sets.py SHAPE [B]"""
import sys
import time

import numpy as np

sys.path.insert(0, ".")
import idgrec_amd.synth as S  # noqa: E402

shape = sys.argv[1] if len(sys.argv) > 1 else "synth-1M"
B = int(sys.argv[2]) if len(sys.argv) > 2 else 1024
U, I, E = S.SHAPES[shape]
t0 = time.time()
users, items = S.generate(U, I, E, seed=0)
tri = S.draw_triples(2024, users, items, U, I, B)[0][:B]
print("%s: %d users x %d items, %d edges (drawn in %.0f s), B = %d" % (shape, U, I, len(users), time.time() - t0, B))


def flags(n, ids):
    f = np.zeros(n, dtype=bool)
    f[ids] = True
    return f


s_u, s_i = flags(U, tri[:, 0]), flags(I, np.concatenate([tri[:, 1], tri[:, 2]]))
t1 = s_i | flags(I, items[s_u[users]])
near = s_u | flags(U, users[s_i[items]])
t2 = s_i | flags(I, items[near[users]])
far = s_u | flags(U, users[t1[items]])
t3 = s_i | flags(I, items[far[users]])
for name, f, n in (("batch users", s_u, U), ("batch items", s_i, I), ("touched items", t1, I), ("near users", near, U),
                   ("two-hop items", t2, I), ("far users", far, U), ("three-hop items", t3, I)):
    print("  %-16s %9d  %6.2f %%" % (name, f.sum(), 100.0 * f.sum() / n))
for name, f, side in (("batch items", s_i, items), ("batch users", s_u, users), ("touched items", t1, items), ("two-hop items", t2, items), ("near users", near, users),
                      ("far users", far, users)):
    print("  stored entries pointing into the %-14s %5.1f %%" % (name, 100.0 * f[side].mean()))
