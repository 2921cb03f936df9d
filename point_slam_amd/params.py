"""Decoder parameters <-> the flat "master" blob of the C ABI (include/pointslam_hip.h).

The table of tensors (reference state_dict keys, torch shapes, offsets) is read
FROM the library (psl_param_entry) so that host and device agree by construction.
"""
from __future__ import annotations

import ctypes as C
from functools import lru_cache
from typing import Dict, List, Tuple

import torch

from . import _lib


@lru_cache(maxsize=1)
def table() -> List[Tuple[str, Tuple[int, ...], int]]:
    """[(state_dict key, torch shape, offset in floats)] in blob order."""
    L = _lib.lib()
    out = []
    buf = C.create_string_buffer(128)
    r, c, o = C.c_int(), C.c_int(), C.c_int()
    for i in range(L.psl_param_count()):
        _lib.check(L.psl_param_entry(i, buf, 128, C.byref(r), C.byref(c), C.byref(o)), "psl_param_entry")
        name = buf.value.decode()
        shape = (r.value,) if name.endswith(".bias") else (r.value, c.value)
        out.append((name, shape, o.value))
    return out


def master_floats() -> int:
    return _lib.lib().psl_param_master_floats()


def color_floats() -> int:
    t = table()
    n = _lib.lib().psl_param_color_count()
    name, shape, off = t[n]
    return off


def named_tensors(decoders) -> Dict[str, torch.Tensor]:
    """Parameters of a reference POINT module (decoder.py:452-475) by state_dict key.
    Also accepts a plain dict name -> tensor."""
    if isinstance(decoders, dict):
        return decoders
    d = dict(decoders.named_parameters())
    return d


def pack_master(decoders) -> torch.Tensor:
    """Differentiable concatenation of the decoder tensors in blob order."""
    d = named_tensors(decoders)
    parts = []
    for name, shape, off in table():
        t = d[name]
        if tuple(t.shape) != tuple(shape):
            raise ValueError(f"{name}: expected shape {shape}, got {tuple(t.shape)}")
        parts.append(t.reshape(-1))
    return torch.cat(parts).float()


def unpack_master(blob: torch.Tensor) -> Dict[str, torch.Tensor]:
    out = {}
    for name, shape, off in table():
        n = 1
        for s in shape:
            n *= s
        out[name] = blob[off:off + n].reshape(shape)
    return out


def color_embed_B(decoders) -> torch.Tensor:
    """The fixed colour Fourier matrix [3][20]: a plain attribute, not in state_dict (decoder.py:27-28,305-306)."""
    if isinstance(decoders, dict):
        return decoders["color_decoder.embedder._B"]
    return decoders.color_decoder.embedder._B
