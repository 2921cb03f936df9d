"""Per-frame image operators in front of the render/optimise path, on the device.

Mirrors of host code the reference runs with skimage / scipy / numpy once per frame:
  * dynamic_radius_maps        <- src/Tracker.py:235-250, src/Mapper.py:686-701
  * get_selected_index_with_grad <- src/common.py:116-159
  * keyframe_overlap / keyframe_selection_overlap <- src/Mapper.py:170-235
All three call libpointslam_hip.so (psl_frame_radii, psl_topgrad_select_sync, psl_keyframe_overlap_sync); there is
no torch fallback.
"""
from __future__ import annotations

import ctypes as C

import numpy as np
import torch

from . import _lib


def dynamic_radius_maps(color: torch.Tensor, cfg: dict, with_grad_mag: bool = False):
    """color [H,W,3] float32 on the device -> (r_add [H,W] f32, r_query [H,W] f32[, grad_mag [H,W] f64])."""
    pc = cfg["pointcloud"]
    H, W = int(color.shape[0]), int(color.shape[1])
    col = color.detach().float().contiguous()
    r_add = torch.empty(H, W, device=col.device, dtype=torch.float32)
    r_query = torch.empty(H, W, device=col.device, dtype=torch.float32)
    gm = torch.empty(H, W, device=col.device, dtype=torch.float64) if with_grad_mag else None
    _lib.check(_lib.lib().psl_frame_radii(_lib.ptr(col), H, W, float(pc["color_grad_threshold"]),
                                          float(pc["radius_add_max"]), float(pc["radius_add_min"]),
                                          float(pc["radius_query_ratio"]), _lib.ptr(gm), _lib.ptr(r_add),
                                          _lib.ptr(r_query), _lib.stream_ptr()), "psl_frame_radii")
    return (r_add, r_query, gm) if with_grad_mag else (r_add, r_query)


def get_selected_index_with_grad(npc, H0, H1, W0, W1, n, image, ratio=15, gt_depth=None, depth_limit=False,
                                 grad_mag=None, cfg=None):
    """common.get_selected_index_with_grad (src/common.py:116-159): flat indices (sorted, int64, on the device) of
    the pixels among the top ratio*n by colour-gradient magnitude that lie in the region and have sensor depth;
    also returns the gradient-magnitude image (float64).  `npc` supplies the native context (scratch)."""
    H, W = int(image.shape[0]), int(image.shape[1])
    if grad_mag is None:
        col = image.detach().float().contiguous()
        grad_mag = torch.empty(H, W, device=col.device, dtype=torch.float64)
        _lib.check(_lib.lib().psl_frame_radii(_lib.ptr(col), H, W, 0.15, 0.08, 0.02, 2.0, _lib.ptr(grad_mag), None, None,
                                              _lib.stream_ptr()), "psl_frame_radii")
    k = min(int(ratio * n), H * W)
    sel = torch.empty(max(k, 1), device=grad_mag.device, dtype=torch.int32)
    n_sel = C.c_int(0)
    dep = gt_depth.detach().float().contiguous() if gt_depth is not None else None
    _lib.check(_lib.lib().psl_topgrad_select_sync(npc.handle, _lib.ptr(grad_mag), _lib.ptr(dep), H, W, k, int(H0), int(H1),
                                                  int(W0), int(W1), 5.0 if depth_limit else 0.0, _lib.ptr(sel),
                                                  C.byref(n_sel), _lib.stream_ptr()), "psl_topgrad_select_sync")
    return torch.sort(sel[:n_sel.value].long()).values, grad_mag


def keyframe_overlap(rays_o, rays_d, gt_depth, keyframe_c2w, cam: dict, n_samples=8, edge=20):
    """percent_inside per keyframe (src/Mapper.py:197-229). keyframe_c2w: list of [4,4] (or [3,4]) poses, tensors or host
    float lists.  Rays with gt_depth <= 0 are skipped by the kernel (get_samples' depth_filter, Mapper.py:190-192): the
    caller may pass the uncompacted draw."""
    n_kf = len(keyframe_c2w)
    if n_kf == 0:
        return np.zeros(0, dtype=np.float32)
    flat = []
    for c in keyframe_c2w:
        if isinstance(c, (list, tuple)):       # 12 (3x4) or 16 host floats, row-major: Frame.c2w_host() -- no device copy
            h = [float(x) for x in c]
            flat += h[:12] + [0.0, 0.0, 0.0, 1.0]
            continue
        m = torch.eye(4)
        cc = c.detach().float().cpu()
        m[:cc.shape[0], :] = cc
        flat += m.reshape(-1).tolist()
    host = (C.c_float * (16 * n_kf))(*flat)
    out = (C.c_float * n_kf)()
    ro = rays_o.detach().float().contiguous()
    rd = rays_d.detach().float().contiguous()
    gd = gt_depth.detach().float().reshape(-1).contiguous()
    intr = _lib.psl_cam_intr(H=cam["H"], W=cam["W"], fx=cam["fx"], fy=cam["fy"], cx=cam["cx"], cy=cam["cy"])
    _lib.check(_lib.lib().psl_keyframe_overlap_sync(_lib.ptr(ro), _lib.ptr(rd), _lib.ptr(gd), int(ro.shape[0]),
                                                    int(n_samples), host, n_kf, intr, float(edge), out,
                                                    _lib.stream_ptr()), "psl_keyframe_overlap_sync")
    return np.array(list(out), dtype=np.float32)


def keyframe_selection_overlap(rays_o, rays_d, gt_depth, keyframe_c2w, cam: dict, k, n_samples=8, rng=None):
    """Mapper.keyframe_selection_overlap (src/Mapper.py:170-235): keyframes with any overlap, in random order, the
    first k of them (np.random.permutation on the host, as in the reference)."""
    pct = keyframe_overlap(rays_o, rays_d, gt_depth, keyframe_c2w, cam, n_samples)
    order = sorted(range(len(pct)), key=lambda i: pct[i], reverse=True)
    sel = [i for i in order if pct[i] > 0.0]
    perm = (rng or np.random).permutation(np.array(sel, dtype=np.int64)) if sel else np.zeros(0, dtype=np.int64)
    return list(perm[:k])


def image_metrics(gt_color, gt_depth, color, depth):
    """(psnr, ms_ssim, depth_l1) of one rendered frame as the end-of-run evaluation computes them
    (src/Mapper.py:861-879).  Images [H,W,3] / [H,W] on the device."""
    H, W = int(gt_depth.shape[0]), int(gt_depth.shape[1])
    out = (C.c_double * 3)()
    a, b = gt_color.detach().float().contiguous(), color.detach().float().contiguous()
    c, d = gt_depth.detach().float().contiguous(), depth.detach().float().contiguous()
    _lib.check(_lib.lib().psl_image_metrics_sync(_lib.ptr(a), _lib.ptr(c), _lib.ptr(b), _lib.ptr(d), H, W, out,
                                                 _lib.stream_ptr()), "psl_image_metrics_sync")
    return float(out[0]), float(out[1]), float(out[2])
