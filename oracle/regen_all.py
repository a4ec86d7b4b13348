"""oracle/regen_all.py — TEST INFRASTRUCTURE.  Runs ONLY where /root/reference exists.

The ONE entry point that regenerates every fixture under tests/golden/ from the imported reference:

    python oracle/regen_all.py            # rewrite tests/golden/*.npz in place (git status stays clean: same inputs,
                                          # same reference, deterministic .npz writer)
    python oracle/regen_all.py --check    # regenerate into a temp dir and compare with the committed files, array by
                                          # array (dtype, shape, values); exit status 1 on any difference
    ... --fast                            # skip convergence_medium.npz (a 40-epoch single-thread reference run, ~15 min)

Inputs are the frozen dataset files under tests/golden/inputs/ (golden_io.py), never the current synthetic generator.
Each generator runs in its own process (they seed global RNGs and import the reference at module level).
"""
import os
import shutil
import subprocess
import sys
import tempfile

HERE = os.path.dirname(os.path.abspath(__file__))
sys.path.insert(0, HERE)
import golden_io  # noqa: E402

FAST = [("gen_golden.py", ["graph_tiny.npz", "graph_small.npz", "misc.npz"]),
        ("gen_golden_next.py", ["next_small.npz"]),
        ("gen_golden_egcf.py", ["egcf_small.npz"]),
        ("gen_golden_wide.py", ["wide_small.npz"])]
SLOW = [("gen_golden_convergence.py", ["convergence_medium.npz"])]


def regenerate(out_dir, fast=False, quiet=False):
    """Run the generators with their output redirected to out_dir; returns the list of files written."""
    env = dict(os.environ, IDG_GOLDEN_OUT=out_dir, PYTHONDONTWRITEBYTECODE="1")
    files = []
    for script, outs in FAST + ([] if fast else SLOW):
        r = subprocess.run([sys.executable, "-B", os.path.join(HERE, script)], env=env, stdout=subprocess.PIPE,
                           stderr=subprocess.STDOUT, text=True)
        if r.returncode != 0:
            raise RuntimeError("%s failed:\n%s" % (script, r.stdout[-3000:]))
        if not quiet:
            print("%s: ok" % script, flush=True)
        files += outs
    return files


def check(fast=False, quiet=False):
    """{file: [differences]} between the committed fixtures and a fresh regeneration (empty lists = reproducible)."""
    tmp = tempfile.mkdtemp(prefix="idg_regen_")
    try:
        files = regenerate(tmp, fast=fast, quiet=quiet)
        return {f: golden_io.same_arrays(os.path.join(golden_io.GOLDEN, f), os.path.join(tmp, f)) for f in files}
    finally:
        shutil.rmtree(tmp, ignore_errors=True)


def main():
    fast = "--fast" in sys.argv
    if not os.path.isdir(os.environ.get("IDG_REFERENCE", "/root/reference")):
        sys.exit("regen_all.py needs the reference tree (IDG_REFERENCE or /root/reference)")
    if "--check" in sys.argv:
        res = check(fast=fast)
        for f, bad in res.items():
            print("%-28s %s" % (f, "reproduced" if not bad else "DIFFERS: " + "; ".join(bad[:6])))
        sys.exit(1 if any(res.values()) else 0)
    regenerate(golden_io.GOLDEN, fast=fast)


if __name__ == "__main__":
    main()
