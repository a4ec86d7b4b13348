"""Build libidgrec.so (gfx950) in-tree with hipcc.  No torch / pybind dependency: the
library is plain HIP + a C ABI (include/idgrec.h) and is loaded through ctypes.

    python id-grec_amd/build.py            # incremental
    python id-grec_amd/build.py --force
"""
import os
import shutil
import subprocess
import sys
from concurrent.futures import ThreadPoolExecutor

HERE = os.path.dirname(os.path.abspath(__file__))
ROOT = os.path.dirname(HERE)
CSRC = os.path.join(HERE, "csrc")
OBJ = os.path.join(HERE, "build")
LIBDIR = os.path.join(HERE, "lib")
LIB = os.path.join(LIBDIR, "libidgrec.so")
SOURCES = ["idg_host.cpp", "idg_comm.cpp", "idg_stream.cpp", "idg_step.cpp", "idg_graph.hip", "idg_bpr.hip", "idg_score.hip", "idg_ssl.hip", "idg_dense.hip",
           "idg_ngcf.hip", "idg_shard.hip"]
ARCH = "gfx950"


def _hipcc():
    for cand in (os.environ.get("HIPCC"), shutil.which("hipcc"), "/opt/rocm/bin/hipcc"):
        if cand and os.path.exists(cand):
            return cand
    raise RuntimeError("hipcc not found (set HIPCC or install ROCm)")


def _stale(target, deps):
    if not os.path.exists(target):
        return True
    t = os.path.getmtime(target)
    return any(os.path.getmtime(d) > t for d in deps)


def build(force=False, verbose=False):
    hipcc = _hipcc()
    os.makedirs(OBJ, exist_ok=True)
    os.makedirs(LIBDIR, exist_ok=True)
    # every header and include file of csrc/ is a dependency of every object (idg_score.hip includes its *.inc kernels)
    headers = [os.path.join(ROOT, "include", "idgrec.h")] + sorted(os.path.join(CSRC, f) for f in os.listdir(CSRC)
                                                                   if f.endswith((".h", ".inc")))
    # -ffp-contract=off: the kernels spell out every fused multiply-add they want (fmaf); left to itself the compiler
    # contracts a*b+c differently in different instantiations of the same source, and paths that must agree bit for
    # bit (single- vs multi-panel kernels, epilogue vs stand-alone perturbation) then differ in the last place
    common = [hipcc, "--offload-arch=" + ARCH, "-O3", "-std=c++17", "-fPIC", "-Wall", "-Wno-unused-function", "-ffp-contract=off",
              "-I" + os.path.join(ROOT, "include"), "-I" + CSRC] + os.environ.get("IDG_BUILD_DEFS", "").split()
    jobs = []
    objs = []
    for src in SOURCES:
        s = os.path.join(CSRC, src)
        o = os.path.join(OBJ, os.path.splitext(src)[0] + ".o")
        objs.append(o)
        if force or _stale(o, [s] + headers):
            cmd = common + (["-x", "hip"] if src.endswith(".hip") else []) + ["-c", s, "-o", o]
            jobs.append(cmd)

    def run(cmd):
        if verbose:
            print(" ".join(cmd), flush=True)
        r = subprocess.run(cmd, stdout=subprocess.PIPE, stderr=subprocess.STDOUT, text=True)
        return cmd, r.returncode, r.stdout

    if jobs:
        with ThreadPoolExecutor(max_workers=min(4, len(jobs))) as ex:
            for cmd, rc, out in ex.map(run, jobs):
                if out.strip():
                    print(out, file=sys.stderr)
                if rc != 0:
                    raise RuntimeError("compile failed: " + " ".join(cmd))
    if jobs or force or _stale(LIB, objs):
        cmd = [hipcc, "--offload-arch=" + ARCH, "-shared", "-fPIC", "-o", LIB] + objs
        _, rc, out = run(cmd)
        if out.strip():
            print(out, file=sys.stderr)
        if rc != 0:
            raise RuntimeError("link failed: " + " ".join(cmd))
    return LIB


if __name__ == "__main__":
    print(build(force="--force" in sys.argv, verbose=True))
