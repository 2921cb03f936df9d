"""Seeded synthetic RGB-D stream: there are no datasets in the
container, so tests and bench.py render an analytic room.

Scene: axis-aligned room [0,6]x[0,4]x[0,3] m seen from inside; depth is the
analytic ray/box exit distance measured along the UNNORMALISED camera ray (the
reference's ``rays_d`` has camera-z = -1, so this equals sensor depth,
src/common.py:49-56); colour is a smooth procedural texture plus checker edges
so that Sobel magnitudes -- and therefore the dynamic radii of
src/Tracker.py:235-250 -- vary over the image.  Camera convention: +x right,
+y up, -z forward (src/utils/datasets.py:147-148).
"""
from __future__ import annotations

import math

import torch

ROOM = (6.0, 4.0, 3.0)


def intrinsics(W=640, H=480):
    s = W / 640.0
    return dict(H=H, W=W, fx=517.0 * s, fy=517.0 * s, cx=319.5 * s + (s - 1) * 0.5, cy=239.5 * s + (s - 1) * 0.5)


def pose(t: float, device="cpu") -> torch.Tensor:
    """Smooth Lissajous trajectory inside the room; c2w 4x4 (OpenGL-style)."""
    c = torch.tensor([3.0 + 1.2 * math.sin(0.05 * t), 2.0 + 0.8 * math.sin(0.035 * t + 0.7),
                      1.5 + 0.3 * math.sin(0.02 * t + 0.3)], dtype=torch.float64)
    yaw = 0.6 * math.sin(0.017 * t) + 0.01 * t
    pitch = 0.15 * math.sin(0.023 * t + 1.0)
    fwd = torch.tensor([math.cos(pitch) * math.cos(yaw), math.cos(pitch) * math.sin(yaw), math.sin(pitch)],
                       dtype=torch.float64)
    up0 = torch.tensor([0.0, 0.0, 1.0], dtype=torch.float64)
    right = torch.linalg.cross(fwd, up0)
    right = right / right.norm()
    up = torch.linalg.cross(right, fwd)
    c2w = torch.eye(4, dtype=torch.float64)
    c2w[:3, 0], c2w[:3, 1], c2w[:3, 2], c2w[:3, 3] = right, up, -fwd, c
    return c2w.float().to(device)


def texture(p: torch.Tensor) -> torch.Tensor:
    """Procedural colour in [0,1] at world points p[...,3]."""
    k = torch.tensor([[7.0, 3.0, 5.0], [2.0, 9.0, 4.0], [5.0, 6.0, 8.0]], device=p.device, dtype=p.dtype)
    ph = torch.tensor([0.3, 1.1, 2.0], device=p.device, dtype=p.dtype)
    base = 0.5 + 0.35 * torch.sin(p @ k.T + ph)
    chk = ((torch.floor(p[..., 0] * 2.0) + torch.floor(p[..., 1] * 2.0) + torch.floor(p[..., 2] * 2.0)) % 2.0)
    return (base + 0.15 * (chk[..., None] - 0.5)).clamp(0.0, 1.0)


def box_depth(o: torch.Tensor, d: torch.Tensor) -> torch.Tensor:
    """Exit parameter t of rays o + t d from inside the room (t == sensor depth)."""
    hi = torch.tensor(ROOM, device=o.device, dtype=o.dtype)
    t_hi = (hi - o) / d
    t_lo = (0.0 - o) / d
    t = torch.where(d > 0, t_hi, torch.where(d < 0, t_lo, torch.full_like(d, float("inf"))))
    return t.min(-1).values


def pixel_dirs(cam: dict, device="cpu"):
    """Camera-frame directions [(u-cx)/fx, -(v-cy)/fy, -1] for the full image -> [H,W,3]."""
    H, W = cam["H"], cam["W"]
    u = torch.arange(W, device=device, dtype=torch.float32)
    v = torch.arange(H, device=device, dtype=torch.float32)
    vv, uu = torch.meshgrid(v, u, indexing="ij")
    return torch.stack([(uu - cam["cx"]) / cam["fx"], -(vv - cam["cy"]) / cam["fy"], -torch.ones_like(uu)], -1)


def render_frame(cam: dict, c2w: torch.Tensor, noise: float = 0.0, dropout: float = 0.0, gen=None):
    """Returns depth [H,W] f32, color [H,W,3] f32 for pose c2w."""
    dev = c2w.device
    dirs = pixel_dirs(cam, dev)
    rd = (dirs[..., None, :] * c2w[:3, :3]).sum(-1)
    ro = c2w[:3, 3].expand_as(rd)
    depth = box_depth(ro, rd)
    col = texture(ro + rd * depth[..., None])
    if noise > 0:
        depth = depth * (1.0 + noise * torch.randn(depth.shape, generator=gen, device=dev))
    if dropout > 0:
        depth = torch.where(torch.rand(depth.shape, generator=gen, device=dev) < dropout,
                            torch.zeros_like(depth), depth)
    return depth.float().contiguous(), col.float().contiguous()


def sobel_mag(color: torch.Tensor) -> torch.Tensor:
    """|grad| of the luma image with skimage's Sobel normalisation (/4), reflect border.
    Stand-in for rgb2gray + filters.sobel_h/_v at src/Tracker.py:236-240."""
    g = (0.2125 * color[..., 0] + 0.7154 * color[..., 1] + 0.0721 * color[..., 2])[None, None]
    g = torch.nn.functional.pad(g, (1, 1, 1, 1), mode="reflect")
    kx = torch.tensor([[1.0, 0.0, -1.0], [2.0, 0.0, -2.0], [1.0, 0.0, -1.0]], device=color.device) / 4.0
    gx = torch.nn.functional.conv2d(g, kx[None, None])
    gy = torch.nn.functional.conv2d(g, kx.T[None, None])
    return torch.sqrt(gx * gx + gy * gy)[0, 0]


def dynamic_radii(color: torch.Tensor, cfg: dict):
    """Per-pixel (r_add, r_query) maps, src/Tracker.py:235-250: piecewise linear in
    the clipped gradient magnitude with knots [0, 0.01, thr]."""
    pc = cfg["pointcloud"]
    thr, rmax, rmin, ratio = pc["color_grad_threshold"], pc["radius_add_max"], pc["radius_add_min"], \
        pc["radius_query_ratio"]
    g = sobel_mag(color).clamp(0.0, thr)
    t = ((g - 0.01) / (thr - 0.01)).clamp(0.0, 1.0)
    r_add = rmax + (rmin - rmax) * t
    return r_add.float(), (ratio * r_add).float()


def in_hole(p: torch.Tensor, cell: float = 0.5) -> torch.Tensor:
    """3-D checkerboard of `cell`-sized cubes: every other cube is left without seed points, so that the frames of a run
    still find uncovered surface and the map GROWS (as it does in a real sequence) instead of being saturated.  (Only
    the inside of an empty cube is "new": a surface point closer than its add-radius, 2..8 cm, to a seeded cube is
    rejected by the dedupe test.)"""
    c = torch.floor((p + 0.5 * cell) / cell).long()     # half-cell offset: the room's walls lie on multiples of the cell
    return ((c[..., 0] + c[..., 1] + c[..., 2]) % 2) == 0


def seed_cloud(cam: dict, n_points: int, n_add: int = 3, n_views: int = 64, seed: int = 1219, device="cpu",
               near=0.98, far=1.02, holes: bool = False, t0: float = 0.0, dt: float = 40.0):
    """Seed ~n_points neural point positions by back-projecting a jittered pixel grid
    from n_views poses (add_neural_points geometry: n_add pts/location at
    linspace(near,far)*depth, src/neural_point.py:126-145).  No dedupe.  holes: leave the cubes of in_hole() empty.
    The views are pose(t0 + dt * v): the default walks the whole trajectory, a small (t0, dt) packs a small cloud around
    the stretch a run will track (bench.py --track-only: BASELINE config 1's fixed 50 k-point cloud)."""
    g = torch.Generator(device="cpu").manual_seed(seed)
    per_view = (n_points // n_add + n_views - 1) // n_views
    if holes:
        per_view = per_view * 2 + 16
    out = []
    t = torch.linspace(0.0, 1.0, n_add)
    for v in range(n_views):
        c2w = pose(t0 + dt * v, "cpu")
        u = torch.rand(per_view, generator=g) * (cam["W"] - 1)
        w = torch.rand(per_view, generator=g) * (cam["H"] - 1)
        dirs = torch.stack([(u - cam["cx"]) / cam["fx"], -(w - cam["cy"]) / cam["fy"], -torch.ones_like(u)], -1)
        rd = (dirs[:, None, :] * c2w[:3, :3]).sum(-1)
        ro = c2w[:3, 3].expand_as(rd)
        d = box_depth(ro, rd)
        z = near * d[:, None] * (1 - t) + far * d[:, None] * t
        trip = ro[:, None, :] + rd[:, None, :] * z[:, :, None]
        if holes:
            trip = trip[~in_hole(ro + rd * d[:, None])]
        out.append(trip.reshape(-1, 3))
    pts = torch.cat(out, 0)
    if holes and pts.shape[0] > n_points:      # keep whole views' worth of points, n_add per location
        pts = pts[:n_points // n_add * n_add]
    pts = pts[:n_points].float().contiguous()
    return pts.to(device)
