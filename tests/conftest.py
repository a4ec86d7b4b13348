import os
import sys

import numpy as np
import pytest

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
if ROOT not in sys.path:
    sys.path.insert(0, ROOT)
GOLDEN = os.path.join(ROOT, "tests", "golden")


def pytest_configure(config):
    config.addinivalue_line("markers", "gpu: needs a real MI355X (run with -m gpu on the GPU box)")


@pytest.fixture(scope="session")
def golden_tiny():
    return dict(np.load(os.path.join(GOLDEN, "graph_tiny.npz"), allow_pickle=False))


@pytest.fixture(scope="session")
def golden_small():
    return dict(np.load(os.path.join(GOLDEN, "graph_small.npz"), allow_pickle=False))


@pytest.fixture(scope="session")
def golden_misc():
    return dict(np.load(os.path.join(GOLDEN, "misc.npz"), allow_pickle=False))


@pytest.fixture(scope="session", autouse=True)
def _build_native():
    """The shared library and the C oracle are build products; make sure they exist."""
    sys.path.insert(0, os.path.join(ROOT, "id-grec_amd"))
    import importlib.util

    spec = importlib.util.spec_from_file_location("idg_build", os.path.join(ROOT, "id-grec_amd", "build.py"))
    mod = importlib.util.module_from_spec(spec)
    spec.loader.exec_module(mod)
    mod.build()
    from oracle import oracle

    oracle.build()
    yield


def has_gpu():
    try:
        import torch

        return torch.cuda.is_available()
    except Exception:
        return False


@pytest.fixture(autouse=True)
def _topk_options_reset():
    """score_topk's knobs (idg_score_topk_option) are process-wide: a test's override does not outlive it."""
    yield
    mod = sys.modules.get("idgrec_amd.ops")
    if mod is not None:
        mod.topk_option("reset")
