"""oracle/golden_io.py — TEST INFRASTRUCTURE (see oracle/oracle.py's header).  What the golden generators share:

* FROZEN INPUTS.  The datasets the goldens are computed on are data, not code: `tests/golden/inputs/<name>/train.txt`,
  `test.txt` in the reference's own text format.  Generators read them from there, so that a change of the synthetic
  generator (`idgrec_amd.synth`, not parity-relevant) can never make a committed fixture irreproducible;
  only when an input is missing is it drawn with the current generator and written there (first creation).
* DETERMINISTIC .npz files: `np.savez_compressed` stamps every member with the wall clock, so regenerating a fixture
  changes its bytes even when every array is identical.  `save_npz` writes the same format with a fixed timestamp:
  `python oracle/regen_all.py` leaves `git status` clean.
"""
import io
import os
import shutil
import zipfile

import numpy as np

HERE = os.path.dirname(os.path.abspath(__file__))
ROOT = os.path.dirname(HERE)
GOLDEN = os.path.join(ROOT, "tests", "golden")
INPUTS = os.path.join(GOLDEN, "inputs")


def out_dir():
    """Where generators write: tests/golden, or $IDG_GOLDEN_OUT (regen_all.py --check regenerates into a temp dir)."""
    d = os.environ.get("IDG_GOLDEN_OUT") or GOLDEN
    os.makedirs(d, exist_ok=True)
    return d


def save_npz(path, **arrays):
    with zipfile.ZipFile(path, "w", compression=zipfile.ZIP_DEFLATED, compresslevel=6) as zf:
        for name, a in arrays.items():
            buf = io.BytesIO()
            np.lib.format.write_array(buf, np.asanyarray(a), allow_pickle=False)
            info = zipfile.ZipInfo(name + ".npy", date_time=(1980, 1, 1, 0, 0, 0))
            info.compress_type = zipfile.ZIP_DEFLATED
            info.external_attr = 0o644 << 16
            zf.writestr(info, buf.getvalue())


def frozen_dataset(name, dst_dir, draw=None):
    """Copy tests/golden/inputs/<name>/{train,test}.txt into dst_dir (created).  draw(dst_dir) writes them with the
    current generator when the frozen copy does not exist yet; the result is then frozen."""
    src = os.path.join(INPUTS, name)
    os.makedirs(dst_dir, exist_ok=True)
    have = all(os.path.exists(os.path.join(src, f)) for f in ("train.txt", "test.txt"))
    if not have:
        if draw is None:
            raise FileNotFoundError("no frozen input %s and no generator given" % src)
        draw(dst_dir)
        os.makedirs(src, exist_ok=True)
        for f in ("train.txt", "test.txt"):
            shutil.copyfile(os.path.join(dst_dir, f), os.path.join(src, f))
        return dst_dir
    for f in ("train.txt", "test.txt"):
        shutil.copyfile(os.path.join(src, f), os.path.join(dst_dir, f))
    return dst_dir


def same_arrays(path_a, path_b):
    """[] when the two .npz files hold the same keys with equal dtype, shape and values; else what differs."""
    a, b = np.load(path_a, allow_pickle=False), np.load(path_b, allow_pickle=False)
    bad = []
    if sorted(a.keys()) != sorted(b.keys()):
        bad.append("keys: only in %s %s, only in %s %s" % (path_a, sorted(set(a.keys()) - set(b.keys())), path_b,
                                                           sorted(set(b.keys()) - set(a.keys()))))
    for k in sorted(set(a.keys()) & set(b.keys())):
        x, y = a[k], b[k]
        if x.dtype != y.dtype or x.shape != y.shape:
            bad.append("%s: %s%s vs %s%s" % (k, x.dtype, x.shape, y.dtype, y.shape))
        elif not (np.array_equal(x, y, equal_nan=True) if x.dtype.kind in "fc" else np.array_equal(x, y)):
            bad.append("%s: values differ" % k)
    return bad
