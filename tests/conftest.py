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


@pytest.fixture(params=["by-size", "split"])
def color_structure(request):
    """Colour-stage launch structure of the decode kernels: "by-size" = the default (fused 16-sample tiles for small launches, the
    split k_nbr_* / k_trunk_* kernels beyond 384 tiles), "split" = the split kernels at every size -- the reference fixtures are
    small, so the split kernels meet them only when forced."""
    from point_slam_amd import _lib
    L = _lib.lib()
    _lib.check(L.psl_debug_option(b"color_split", 2 if request.param == "split" else 1))
    yield request.param
    _lib.check(L.psl_debug_option(b"color_split", 1))
