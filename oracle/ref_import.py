"""Import the UNMODIFIED reference modules from /root/reference on CPU.

TEST INFRASTRUCTURE ONLY (see oracle/pointslam_oracle.py header).  Used by
oracle/gen_golden.py (fixture generation, this container only) and by the
optional ``cpu_baseline.kind == "reference"`` leg.  /root/reference does not
exist on the GPU box, so nothing that runs there may call ``load()``.

Stubs (sys.modules) for dependencies that are absent from the image:
  faiss            -> exact k-NN index (oracle.knn_exact): the FAISS seam is
                      parity-unpinned, see the oracle header.
  cv2              -> ``remap`` + ``INTER_LINEAR`` only (numpy, see _cv2_remap): the one cv2 call on the
                      path, the depth lookup of Mapper.get_mask_from_c2w (src/Mapper.py:149-155).  With it the
                      UNMODIFIED method runs here and pins its projection, dtype chain, edge crop, hole rule
                      and depth + 0.5 test (oracle/gen_golden_frame.py); the interpolation inside remap itself
                      stays a restatement of OpenCV's published algorithm (CV2_REMAP_RULE selects it).
  skimage, wandb, colorama, open3d, torchmetrics, pytorch_msssim,
  src.utils.datasets, src.utils.Visualizer, src.utils.Logger -> inert shells
                      (none of them is on the hot path).
Two CPU-only failures of the reference are wrapped, not edited:
  * POINT.forward builds the device string 'cuda:-1' (decoder.py:499,505);
    ``point_forward_cpu`` calls the two decoders exactly as POINT.forward does.
  * quad2rotation does ``.to(quad.get_device())`` (common.py:238).
"""
from __future__ import annotations

import os
import sys
import types

import torch

REF = os.environ.get("POINTSLAM_REFERENCE", "/root/reference")
_loaded = None
CV2_REMAP_RULE = "cv2"        # "cv2": INTER_LINEAR as OpenCV computes it (1/32-pixel fixed point); "exact": plain bilinear


def available() -> bool:
    return os.path.isdir(os.path.join(REF, "src"))


class _ExactIndex:
    """Minimal stand-in for the FAISS index API used at src/neural_point.py:37-41,
    60-64,162-164,193 -- exact search."""

    def __init__(self, *a, **k):
        self.is_trained = False
        self.nprobe = 1
        self._pts = torch.zeros(0, 3)

    @property
    def ntotal(self):
        return self._pts.shape[0]

    def train(self, x):
        self.is_trained = True

    def add(self, x):
        self._pts = torch.cat([self._pts, x.detach().float().cpu().reshape(-1, 3)], 0)

    def search(self, q, k):
        from . import pointslam_oracle as O
        return O.knn_exact(self._pts, q.detach().float().cpu(), k)


def _cv2_remap(src, map1, map2, interpolation=1, **kw):
    """Stand-in for cv2.remap(src, map1, map2, interpolation=cv2.INTER_LINEAR) with float32 maps and the default
    BORDER_CONSTANT 0, in numpy.  1-D maps are n x 1 images, as cv2 treats them (the caller takes [:, 0]).
    CV2_REMAP_RULE == "cv2": OpenCV's remap() for CV_32FC1 maps (imgproc/src/imgwarp.cpp): sx = cvRound(x * 32),
    integer part sx >> 5 saturated to int16, weights from the fractional 5 bits, taps summed row by row;
    "exact": bilinear interpolation at the unquantised coordinate."""
    import numpy as np
    assert interpolation == 1 and not kw
    img = np.asarray(src, dtype=np.float32)
    H, W = img.shape
    x = np.asarray(map1, dtype=np.float32).reshape(-1)
    y = np.asarray(map2, dtype=np.float32).reshape(-1)
    near = (np.abs(x) < 1.0e6) & (np.abs(y) < 1.0e6)              # NaN compares false
    xs, ys = np.where(near, x, np.float32(0)), np.where(near, y, np.float32(0))
    if CV2_REMAP_RULE == "cv2":
        sx = np.rint(xs * np.float32(32)).astype(np.int64)      # cvRound: round half to even
        sy = np.rint(ys * np.float32(32)).astype(np.int64)
        ix, iy = np.clip(sx >> 5, -32768, 32767), np.clip(sy >> 5, -32768, 32767)
        ax = (sx & 31).astype(np.float32) * np.float32(1.0 / 32.0)
        ay = (sy & 31).astype(np.float32) * np.float32(1.0 / 32.0)
    else:
        fx_, fy_ = np.floor(xs), np.floor(ys)
        ix, iy = fx_.astype(np.int64), fy_.astype(np.int64)
        ax, ay = xs - fx_, ys - fy_
    out = np.zeros(x.shape, dtype=np.float32)
    one = np.float32(1)
    for ky, wy in ((0, one - ay), (1, ay)):
        for kx, wx in ((0, one - ax), (1, ax)):
            xx, yy = ix + kx, iy + ky
            ok = (xx >= 0) & (xx < W) & (yy >= 0) & (yy < H)
            val = np.where(ok, img[np.clip(yy, 0, H - 1), np.clip(xx, 0, W - 1)], np.float32(0))
            out = out + val * (wy * wx)
    out = np.where(near, out, np.float32(0)).astype(np.float32)
    return out.reshape(-1, 1)


def _install_stubs():
    def mod(name, **attrs):
        m = types.ModuleType(name)
        m.__dict__.update(attrs)
        sys.modules[name] = m
        return m

    faiss = mod("faiss", StandardGpuResources=lambda: None, METRIC_L2=1,
                IndexFlatL2=lambda d: None,
                IndexIVFFlat=lambda q, d, nlist, metric: _ExactIndex(),
                index_cpu_to_gpu=lambda res, dev, idx: idx)
    faiss.contrib = mod("faiss.contrib")
    faiss.contrib.torch_utils = mod("faiss.contrib.torch_utils")
    sk = mod("skimage")
    sk.color = mod("skimage.color", rgb2gray=None)
    sk.filters = mod("skimage.filters")
    mod("cv2", remap=_cv2_remap, INTER_LINEAR=1)
    mod("wandb")
    mod("open3d")
    mod("colorama", Fore=types.SimpleNamespace(MAGENTA="", GREEN="", RED=""),
        Style=types.SimpleNamespace(RESET_ALL=""))
    tm = mod("torchmetrics")
    tm.image = mod("torchmetrics.image")
    tm.image.lpip = mod("torchmetrics.image.lpip", LearnedPerceptualImagePatchSimilarity=object)
    mod("pytorch_msssim", ms_ssim=None)
    mod("src.utils.datasets", get_dataset=None)
    mod("src.utils.Visualizer", Visualizer=object)
    mod("src.utils.Logger", Logger=object)


def load():
    """Returns a namespace with the reference modules (common, decoder, neural_point, Renderer, Tracker)."""
    global _loaded
    if _loaded is not None:
        return _loaded
    if not available():
        raise RuntimeError(f"reference tree not found at {REF}")
    _install_stubs()
    if REF not in sys.path:
        sys.path.insert(0, REF)
    from src import common
    from src.conv_onet.models import decoder
    from src import neural_point
    from src.utils import Renderer as renderer_mod
    ns = types.SimpleNamespace(common=common, decoder=decoder, neural_point=neural_point,
                               Renderer=renderer_mod.Renderer)
    try:
        from src import Tracker as tracker_mod
        ns.tracker_mod = tracker_mod
    except Exception as e:  # pragma: no cover
        ns.tracker_mod = None
        ns.tracker_err = repr(e)
    _loaded = ns
    return ns


def make_decoders(cfg, seed=1219):
    """setup_seed + get_model equivalent (run.py:31, src/conv_onet/config.py:4-21)."""
    ns = load()
    ns.common.setup_seed(seed)
    m = ns.decoder.POINT(cfg, c_dim=cfg["model"]["c_dim"],
                         pos_embedding_method=cfg["model"]["pos_embedding_method"],
                         use_view_direction=cfg["model"]["use_view_direction"])
    return m


class PointCPU(torch.nn.Module):
    """Wrapper that runs POINT.forward's two branches on CPU (decoder.py:497-518)."""

    def __init__(self, point):
        super().__init__()
        self.point = point
        self.geo_decoder = point.geo_decoder
        self.color_decoder = point.color_decoder

    def forward(self, p, npc, stage, npc_geo_feats, npc_col_feats, pts_num=16, is_tracker=False,
                cloud_pos=None, pts_views_d=None, dynamic_r_query=None, exposure_feat=None):
        geo_occ, ray_mask, point_mask = self.geo_decoder(
            p, npc, npc_geo_feats, pts_num=pts_num, is_tracker=is_tracker, cloud_pos=cloud_pos,
            dynamic_r_query=dynamic_r_query)
        if stage == "geometry":
            raw = torch.zeros(geo_occ.shape[0], 4, dtype=torch.float)
            raw[..., -1] = geo_occ
            return raw, ray_mask, point_mask
        raw = self.color_decoder(p, npc, npc_col_feats, is_tracker=is_tracker, cloud_pos=cloud_pos,
                                 pts_views_d=pts_views_d, dynamic_r_query=dynamic_r_query,
                                 exposure_feat=exposure_feat)
        raw = torch.cat([raw, geo_occ.unsqueeze(-1)], dim=-1)
        return raw, ray_mask, point_mask


def state_with_fixed_B(point) -> dict:
    """state_dict plus the non-persistent colour embedder matrix (decoder.py:27-28,305-306)."""
    sd = {k: v.detach().clone() for k, v in point.state_dict().items()}
    sd["color_decoder.embedder._B"] = point.color_decoder.embedder._B.detach().clone()
    return sd
