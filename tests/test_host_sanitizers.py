"""The library's host code (rating-file parser, adjacency builder, MT19937 sampler / shuffle / random.sample:
id-grec_amd/csrc/idg_host.cpp) under AddressSanitizer + UndefinedBehaviorSanitizer.  The GPU pool offers no sanitizers;
this is the part of the native code that can run under them, on good inputs (the frozen golden inputs) and on malformed
ones (tests/sanitize_host.cpp lists them).  Found in round 4: signed overflow in the parser on a 32-digit id."""
import os
import shutil
import subprocess

import pytest

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))


def test_host_entry_points_under_asan_and_ubsan(tmp_path):
    gxx = shutil.which("g++")
    if gxx is None:
        pytest.skip("no g++")
    exe = str(tmp_path / "sanitize_host")
    build = subprocess.run([gxx, "-std=c++17", "-g", "-O1", "-fsanitize=address,undefined", "-fno-sanitize-recover=all",
                            "-fno-omit-frame-pointer", "-I" + os.path.join(ROOT, "include"),
                            "-I" + os.path.join(ROOT, "id-grec_amd", "csrc"), os.path.join(ROOT, "tests", "sanitize_host.cpp"),
                            os.path.join(ROOT, "id-grec_amd", "csrc", "idg_host.cpp"), "-o", exe],
                           capture_output=True, text=True, timeout=600)
    if build.returncode != 0 and "sanitize" in build.stderr.lower() and "cannot find" in build.stderr.lower():
        pytest.skip("this g++ has no sanitizer runtimes")
    assert build.returncode == 0, build.stderr[-3000:]
    inputs = os.path.join(ROOT, "tests", "golden", "inputs")
    files = [os.path.join(inputs, d, f) for d in ("tiny", "small", "small_egcf") for f in ("train.txt", "test.txt")
             if os.path.exists(os.path.join(inputs, d, f))]
    assert len(files) >= 4
    scratch = tmp_path / "files"
    scratch.mkdir()
    run = subprocess.run([exe, str(scratch)] + files, capture_output=True, text=True, timeout=600,
                         env=dict(os.environ, ASAN_OPTIONS="detect_leaks=1:abort_on_error=0", UBSAN_OPTIONS="print_stacktrace=1"))
    assert run.returncode == 0, (run.stdout + run.stderr)[-4000:]
