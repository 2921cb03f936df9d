"""Host-side (torch, any device) glue that the reference keeps in src/common.py and inside its
Tracker/Mapper loops.  These are the CALLERS' ops around the HIP render path: they stay in torch so
that the reference loops can be pointed at HipRenderer unchanged.  The fused native
versions used by the fast path live in libpointslam_hip.so (psl_track_iters / psl_map_iters).
"""
from __future__ import annotations

import torch


def quad2rotation(quad: torch.Tensor) -> torch.Tensor:
    """src/common.py:225-248 (device-agnostic: the reference's `.to(quad.get_device())` fails on CPU)."""
    qr, qi, qj, qk = quad[:, 0], quad[:, 1], quad[:, 2], quad[:, 3]
    two_s = 2.0 / (quad * quad).sum(-1)
    R = torch.zeros(quad.shape[0], 3, 3, device=quad.device, dtype=quad.dtype)
    R[:, 0, 0] = 1 - two_s * (qj ** 2 + qk ** 2)
    R[:, 0, 1] = two_s * (qi * qj - qk * qr)
    R[:, 0, 2] = two_s * (qi * qk + qj * qr)
    R[:, 1, 0] = two_s * (qi * qj + qk * qr)
    R[:, 1, 1] = 1 - two_s * (qi ** 2 + qk ** 2)
    R[:, 1, 2] = two_s * (qj * qk - qi * qr)
    R[:, 2, 0] = two_s * (qi * qk - qj * qr)
    R[:, 2, 1] = two_s * (qj * qk + qi * qr)
    R[:, 2, 2] = 1 - two_s * (qi ** 2 + qj ** 2)
    return R


def get_camera_from_tensor(inputs: torch.Tensor) -> torch.Tensor:
    """src/common.py:251-267: (quat wxyz, T) -> [3,4]."""
    one = inputs.dim() == 1
    if one:
        inputs = inputs.unsqueeze(0)
    RT = torch.cat([quad2rotation(inputs[:, :4]), inputs[:, 4:, None]], 2)
    return RT[0] if one else RT


def get_rays_from_uv(i, j, c2w, fx, fy, cx, cy, device=None):
    """src/common.py:40-56."""
    dirs = torch.stack([(i - cx) / fx, -(j - cy) / fy, -torch.ones_like(i)], -1).reshape(-1, 1, 3)
    rays_d = torch.sum(dirs * c2w[:3, :3], -1)
    rays_o = c2w[:3, -1].expand(rays_d.shape)
    return rays_o, rays_d


def pixels_from_flat_index(idx, H0, H1, W0, W1):
    """select_uv on the cropped window (src/common.py:59-89)."""
    w = W1 - W0
    return (W0 + idx % w).float(), (H0 + torch.div(idx, w, rounding_mode="floor")).float()


def depth_inlier_mask(d):
    """src/Tracker.py:142-144 / src/Mapper.py:507-509."""
    return d <= torch.minimum(10 * d.median(), 1.2 * d.max())


def tracker_loss(depth, var, rgb, gt_depth, gt_color, handle_dynamic=True, use_color=True, w_color=0.5):
    """src/Tracker.py:159-180."""
    u = var.detach()
    nan_mask = (~torch.isnan(depth)) & (~torch.isnan(u))
    if handle_dynamic:
        tmp = torch.abs(gt_depth - depth) / torch.sqrt(u + 1e-10)
        mask = (tmp < 10 * tmp.mean()) & (gt_depth > 0)
    else:
        tmp = torch.abs(gt_depth - depth)
        mask = (tmp < 10 * tmp.median()) & (gt_depth > 0)
    mask = mask & nan_mask
    geo = torch.clamp(torch.abs(gt_depth - depth) / torch.sqrt(u + 1e-10), min=0.0, max=1e3)[mask].sum()
    col = torch.abs(gt_color - rgb)[mask].sum()
    loss = geo + w_color * col if use_color else geo
    return loss, geo, col, mask


def mapper_loss(depth, rgb, valid_ray, gt_depth, gt_color, stage, w_color=0.1):
    """src/Mapper.py:524-553 (without per-frame exposure)."""
    m = (gt_depth > 0) & valid_ray & (~torch.isnan(depth))
    geo = torch.abs(gt_depth[m] - depth[m]).sum()
    loss = geo.clone()
    col = torch.zeros((), device=depth.device)
    if stage == "color":
        col = torch.abs(gt_color[m] - rgb[m]).sum()
        loss = loss + w_color * col
    return loss, geo, col, m


def const_speed_init(prev_c2w: torch.Tensor, prev_prev_c2w: torch.Tensor = None) -> torch.Tensor:
    """Initial pose of the next frame (src/Tracker.py:283-290): delta = pre_c2w @ inv(c2w[idx-2]), estimate = delta @ pre_c2w
    (const_speed_assumption); the previous estimate itself when there is no frame idx-2."""
    prev = prev_c2w.float()
    if prev_prev_c2w is None:
        return prev
    return (prev @ prev_prev_c2w.float().inverse()) @ prev


def camera_tensor_from_c2w_device(c2w: torch.Tensor) -> torch.Tensor:
    """get_tensor_from_camera (src/common.py:270-295) without leaving the device: [quat(w,x,y,z), T] of a 4x4 (or 3x4)
    pose.  The reference goes through scipy on the host; a frame loop that feeds the tracker's own estimates back as the next
    initial pose would stall on that copy once per frame.  Shepperd's method, the branch taken by selects; the sign of the
    quaternion is free (q and -q are the same rotation; the reference flips it towards the ground-truth hemisphere,
    Tracker.py:294-295, which changes nothing in the optimisation)."""
    R = c2w[:3, :3].float()
    m00, m11, m22 = R[0, 0], R[1, 1], R[2, 2]
    tr = m00 + m11 + m22
    cands = torch.stack([
        torch.stack([1.0 + tr, R[2, 1] - R[1, 2], R[0, 2] - R[2, 0], R[1, 0] - R[0, 1]]),
        torch.stack([R[2, 1] - R[1, 2], 1.0 + m00 - m11 - m22, R[0, 1] + R[1, 0], R[0, 2] + R[2, 0]]),
        torch.stack([R[0, 2] - R[2, 0], R[0, 1] + R[1, 0], 1.0 - m00 + m11 - m22, R[1, 2] + R[2, 1]]),
        torch.stack([R[1, 0] - R[0, 1], R[0, 2] + R[2, 0], R[1, 2] + R[2, 1], 1.0 - m00 - m11 + m22])])
    k = torch.argmax(torch.stack([tr, m00, m11, m22]))          # the largest diagonal term: the best-conditioned candidate
    q = cands.index_select(0, k.reshape(1))[0]
    q = q / q.norm()
    q = torch.where(q[0] < 0, -q, q)
    return torch.cat([q, c2w[:3, 3].float()])
