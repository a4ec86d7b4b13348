"""Launch the ranks of a multi-process test (world_size-2 gloo workers) without the classic ways to hang a test run:
outputs go to files (a full pipe cannot block a rank that the other rank is waiting for in a collective), the ranks
are polled together (one rank dying ends the others at once instead of after their rendezvous timeout), and a failed
rendezvous — the free port found a moment ago can be taken by the time rank 0 binds it — is retried on a new port."""
import os
import socket
import subprocess
import sys
import tempfile
import time

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
RENDEZVOUS_ERRORS = ("address already in use", "EADDRINUSE", "Connection refused", "connection reset", "Broken pipe",
                     "timed out", "Timed out", "failed to connect")


def free_port():
    with socket.socket() as s:
        s.bind(("127.0.0.1", 0))
        return s.getsockname()[1]


def run_ranks(script, world, args_after_port, timeout=300, attempts=3):
    """python <script> <rank> <world> <port> <args...> for every rank; returns the ranks' outputs.  Raises
    AssertionError with the outputs when a rank fails (after `attempts` tries if the failure looks like a rendezvous
    problem) or the ranks do not finish within `timeout` seconds."""
    env = dict(os.environ, PYTHONPATH=ROOT, HSA_ENABLE_IPC_MODE_LEGACY="0")
    last = ""
    for attempt in range(attempts):
        port = free_port()
        files = [tempfile.TemporaryFile(mode="w+") for _ in range(world)]
        procs = [subprocess.Popen([sys.executable, os.path.join(ROOT, "tests", script), str(r), str(world), str(port)]
                                  + [str(a) for a in args_after_port], env=env, stdout=files[r], stderr=subprocess.STDOUT,
                                  text=True) for r in range(world)]
        t0, failed = time.time(), False
        while any(p.poll() is None for p in procs):
            if any(p.returncode not in (None, 0) for p in procs) or time.time() - t0 > timeout:
                failed = True
                break
            time.sleep(0.05)
        timed_out = failed and all(p.returncode in (None, 0) for p in procs)
        for p in procs:
            if p.poll() is None:
                p.kill()
                p.wait()
        outs = []
        for f in files:
            f.seek(0)
            outs.append(f.read())
            f.close()
        if all(p.returncode == 0 for p in procs) and not failed:
            return outs
        last = "\n".join("---- rank %d (rc %s%s) ----\n%s" % (r, procs[r].returncode, ", killed after timeout" if timed_out else "",
                                                            outs[r][-3000:]) for r in range(world))
        if timed_out or not any(e in o for o in outs for e in RENDEZVOUS_ERRORS):
            break  # a real failure (or a hang): do not paper over it
    raise AssertionError(last)
