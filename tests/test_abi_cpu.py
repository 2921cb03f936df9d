"""CPU: the C-ABI shared library loads and exports every function include/pointslam_hip.h declares; the parameter
table agrees with the reference state_dict keys/shapes stored in the golden fixtures; host logic (config, dist
partitioning).  No compute calls (no GPU here)."""
import ctypes
import os
import re

import pytest

from tests.helpers import ROOT, load_decoders


def _declared():
    src = open(os.path.join(ROOT, "include", "pointslam_hip.h")).read()
    src = re.sub(r"/\*.*?\*/", "", src, flags=re.S)
    return sorted(set(re.findall(r"\b(psl_[a-z_0-9]+)\s*\(", src)))


def test_library_exports_every_declared_symbol():
    from point_slam_amd import _lib
    assert os.path.exists(_lib.LIB_PATH), "build with __graft_entry__.build()"
    L = ctypes.CDLL(_lib.LIB_PATH)
    names = _declared()
    assert len(names) >= 30
    missing = [n for n in names if not hasattr(L, n)]
    assert not missing, missing
    # and the Python binding declares a signature for each of them
    unbound = [n for n in names if n not in _lib.EXPORTED_SYMBOLS]
    assert not unbound, unbound
    assert _lib.lib().psl_abi_version() == _lib.ABI_VERSION


def test_param_table_matches_reference_state_dict():
    from point_slam_amd import params
    ref = load_decoders("replica")
    tab = params.table()
    off = 0
    for name, shape, o in tab:
        assert name in ref, name
        assert tuple(ref[name].shape) == tuple(shape), (name, shape, tuple(ref[name].shape))
        assert o == off
        n = 1
        for s in shape:
            n *= s
        off += n
    assert off == params.master_floats() == 124665
    assert params.color_floats() == 108865               # 108 865 colour-decoder parameters
    blob = params.pack_master(ref)
    back = params.unpack_master(blob)
    for name, shape, o in tab:
        assert (back[name] == ref[name]).all()


def test_missing_library_fails_loudly(monkeypatch):
    from point_slam_amd import _lib
    monkeypatch.setattr(_lib, "_lib", None)
    monkeypatch.setattr(_lib, "LIB_PATH", "/nonexistent/libpointslam_hip.so")
    with pytest.raises(_lib.PslError):
        _lib.lib()


def test_config_inheritance(tmp_path):
    from point_slam_amd.config import default_config, load_config, replica_overrides
    base = tmp_path / "base.yaml"
    base.write_text("tracking:\n  pixels: 321\nmapping:\n  iters: 7\n")
    child = tmp_path / "child.yaml"
    child.write_text(f"inherit_from: {base}\ntracking:\n  iters: 9\n")
    cfg = load_config(str(child))
    assert cfg["tracking"]["pixels"] == 321 and cfg["tracking"]["iters"] == 9 and cfg["mapping"]["iters"] == 7
    assert cfg["pointcloud"]["nn_num"] == 8               # falls through to the built-in defaults
    assert replica_overrides(default_config())["mapping"]["pixels"] == 5000
