import os
import sys

import pytest

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
if ROOT not in sys.path:
    sys.path.insert(0, ROOT)


def pytest_configure(config):
    config.addinivalue_line("markers", "gpu: needs a real MI355X (run with -m gpu on the GPU box)")


def pytest_collection_modifyitems(config, items):
    """A plain `pytest tests` on a box without a GPU skips the gpu-marked tests instead of erroring in them.  When the
    marker is asked for explicitly (-m gpu) nothing is skipped: on the GPU box a missing device must fail loudly."""
    if "gpu" in (config.getoption("-m") or ""):
        return
    try:
        import torch
        has = torch.cuda.is_available()
    except Exception:
        has = False
    if has:
        return
    skip = pytest.mark.skip(reason="needs a real MI355X (run with -m gpu on the GPU box)")
    for it in items:
        if "gpu" in it.keywords:
            it.add_marker(skip)


@pytest.fixture(scope="session")
def golden_dir():
    return os.path.join(ROOT, "tests", "golden")
