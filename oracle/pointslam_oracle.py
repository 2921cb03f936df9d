"""CPU oracle for the Point-SLAM render/optimise hot path.

TEST INFRASTRUCTURE ONLY.  Nothing under ``point_slam_amd/`` may import this
module; only ``tests/``, ``__graft_entry__.smoke()`` and the ``cpu_baseline``
leg of ``bench.py`` do, and only as the checker.

It is an independent fp32 restatement (torch on CPU, autograd for gradients)
of the reference algorithm, written function-by-function from the reference
sources cited below (paths relative to /root/reference).  It is pinned against
the reference's *own* Python, imported unmodified with a FAISS stub, by
``oracle/gen_golden.py``; the resulting vectors live in ``tests/golden/``.

Parity status: pinned for everything except the k-NN seam.  The reference's
k-NN is FAISS-GPU IVF-Flat (third-party ``faiss-gpu==1.7.2``, env.yaml:96, not
vendored, approximate, nprobe=4/nlist=400) for which the reference holds no
golden vectors: **parity unpinned at the FAISS seam**.  The oracle defines the
search as EXACT 8-NN under squared L2 ``(dx*dx+dy*dy)+dz*dz`` in fp32, ties
broken by the lower point index, ascending order (== FAISS with nprobe=nlist).
"""
from __future__ import annotations

import math
from typing import Dict, Optional, Tuple

import numpy as np
import torch
import torch.nn.functional as F

Tensor = torch.Tensor
TWO_PI_F32 = float(np.float32(2.0 * math.pi))


# --------------------------------------------------------------------------
# rays / sample placement
# --------------------------------------------------------------------------
def rays_from_uv(i: Tensor, j: Tensor, c2w: Tensor, fx, fy, cx, cy) -> Tuple[Tensor, Tensor]:
    """Pixel (i=u, j=v) -> world ray. src/common.py:40-56.

    dirs = [(u-cx)/fx, -(v-cy)/fy, -1]; rays_d[r,a] = sum_k dirs[r,k]*c2w[a,k];
    rays_o = c2w[:3,3] broadcast.  rays_d is NOT normalised.
    """
    dirs = torch.stack([(i - cx) / fx, -(j - cy) / fy, -torch.ones_like(i)], -1)
    rays_d = (dirs[:, None, :] * c2w[:3, :3]).sum(-1)
    rays_o = c2w[:3, 3].expand(rays_d.shape)
    return rays_o, rays_d


def z_samples(gt_depth: Tensor, near_surf: float, far_surf: float, n_surface: int) -> Tensor:
    """Per-ray depths of the S samples around the sensor depth.
    src/utils/Renderer.py:134-141 (rays with gt_depth > 0)."""
    t = torch.linspace(0.0, 1.0, steps=n_surface)
    d = gt_depth.reshape(-1, 1).repeat(1, n_surface)
    return near_surf * d * (1.0 - t) + far_surf * d * t


def sample_points(rays_o: Tensor, rays_d: Tensor, z: Tensor) -> Tensor:
    """pts = o + d*z (separate mul, then add). src/utils/Renderer.py:172-174."""
    return (rays_o[:, None, :] + rays_d[:, None, :] * z[:, :, None]).reshape(-1, 3)


def quat_to_rot(q: Tensor) -> Tensor:
    """quad2rotation for one quaternion (w,x,y,z), src/common.py:225-248."""
    qr, qi, qj, qk = q[0], q[1], q[2], q[3]
    two_s = 2.0 / (q * q).sum()
    return torch.stack([
        torch.stack([1 - two_s * (qj ** 2 + qk ** 2), two_s * (qi * qj - qk * qr), two_s * (qi * qk + qj * qr)]),
        torch.stack([two_s * (qi * qj + qk * qr), 1 - two_s * (qi ** 2 + qk ** 2), two_s * (qj * qk - qi * qr)]),
        torch.stack([two_s * (qi * qk - qj * qr), two_s * (qj * qk + qi * qr), 1 - two_s * (qi ** 2 + qj ** 2)]),
    ])


def camera_from_tensor(quat: Tensor, trans: Tensor) -> Tensor:
    """get_camera_from_tensor, src/common.py:251-267 -> [3,4]."""
    return torch.cat([quat_to_rot(quat), trans[:, None]], 1)


def pixels_from_flat_index(idx: Tensor, H0: int, H1: int, W0: int, W1: int):
    """get_sample_uv/select_uv, src/common.py:59-89: the flat index addresses the
    cropped [H0:H1, W0:W1] window row-major; returns float (u, v) pixel coords."""
    w = W1 - W0
    return (W0 + idx % w).float(), (H0 + torch.div(idx, w, rounding_mode="floor")).float()


# --------------------------------------------------------------------------
# k-NN (exact definition; replaces FAISS, see module docstring)
# --------------------------------------------------------------------------
def sqdist(q: Tensor, p: Tensor) -> Tensor:
    """(dx*dx + dy*dy) + dz*dz, fp32, no fma; q [...,3], p [...,3]."""
    d = p - q
    return (d[..., 0] * d[..., 0] + d[..., 1] * d[..., 1]) + d[..., 2] * d[..., 2]


_TREE_CACHE: dict = {}
KNN_WORKERS = -1


def knn_exact(cloud: Tensor, q: Tensor, k: int = 8, chunk: int = 2048) -> Tuple[Tensor, Tensor]:
    """Exact k-NN.  Returns D [n,k] f32 squared distances ascending, I [n,k] int64.
    Missing neighbours (cloud smaller than k): D=+inf, I=-1.
    Contract of NeuralPointCloud.find_neighbors_faiss, src/neural_point.py:169-197.
    """
    n, N = q.shape[0], cloud.shape[0]
    D = torch.full((n, k), float("inf"), dtype=torch.float32)
    I = torch.full((n, k), -1, dtype=torch.int64)
    if N == 0 or n == 0:
        return D, I
    use_tree = N > 20000
    if use_tree:
        from scipy.spatial import cKDTree
        key = (cloud.data_ptr(), N)
        tree = _TREE_CACHE.get(key)
        if tree is None:
            _TREE_CACHE.clear()
            tree = _TREE_CACHE[key] = cKDTree(cloud.double().numpy())
        kk = min(N, k + 8)
        _, cand = tree.query(q.double().numpy(), k=kk, workers=KNN_WORKERS)
        cand = torch.from_numpy(np.asarray(cand).reshape(n, kk).astype(np.int64))
        dc = sqdist(q[:, None, :], cloud[cand])                # [n,kk] fp32 re-evaluation
        key_order = _lexsort_rows(dc, cand)
        dc = torch.gather(dc, 1, key_order)[:, :k]
        cand = torch.gather(cand, 1, key_order)[:, :k]
        m = min(k, kk)
        D[:, :m], I[:, :m] = dc[:, :m], cand[:, :m]
        return D, I
    for s in range(0, n, chunk):
        qq = q[s:s + chunk]
        d = sqdist(qq[:, None, :], cloud[None, :, :])          # [c,N]
        m = min(k, N)
        # stable sort on distance keeps the lower index first among ties
        ds, idx = torch.sort(d, dim=1, stable=True)
        D[s:s + chunk, :m] = ds[:, :m]
        I[s:s + chunk, :m] = idx[:, :m]
    return D, I


def _lexsort_rows(d: Tensor, idx: Tensor) -> Tensor:
    """argsort rows by (d, idx) lexicographically."""
    o1 = torch.argsort(idx, dim=1, stable=True)
    d1 = torch.gather(d, 1, o1)
    o2 = torch.argsort(d1, dim=1, stable=True)
    return torch.gather(o1, 1, o2)


def neighbor_count(D: Tensor, radius) -> Tensor:
    """#(D < r^2) per query, int32. src/neural_point.py:207-213 (strict <)."""
    if torch.is_tensor(radius):
        r2 = radius.reshape(-1, 1) ** 2
    else:
        r2 = radius ** 2
    return (D < r2).sum(-1).int()


# --------------------------------------------------------------------------
# feature interpolation
# --------------------------------------------------------------------------
def idw_weights(D: Tensor, r2, weighting: str = "distance") -> Tensor:
    """w = 1/(D+1e-10) ('distance') or exp(-20 sqrt(D)) ('expo'); w[D>r2]=0; L1 normalise with eps 1e-12.
    src/conv_onet/models/decoder.py:152-160 / 362-368 (pointcloud.nn_weighting, configs/point_slam.yaml:110)."""
    w = 1.0 / (D + 1e-10) if weighting == "distance" else torch.exp(-20 * torch.sqrt(D))
    w = torch.where(D > r2, torch.zeros_like(w), w)
    return w / w.abs().sum(1, keepdim=True).clamp_min(1e-12)


def fourier(x: Tensor, B: Tensor, concat: bool) -> Tensor:
    """GaussianFourierFeatureTransform.forward, decoder.py:30-37:
    y=(2*pi*x)@B; sin(y) or [sin(y),cos(y)]."""
    y = (TWO_PI_F32 * x) @ B
    return torch.cat((torch.sin(y), torch.cos(y)), -1) if concat else torch.sin(y)


def softplus100(x: Tensor) -> Tensor:
    """torch.nn.Softplus(beta=100) with default threshold 20. decoder.py:124,231,335."""
    return F.softplus(x, beta=100.0, threshold=20.0)


def safe_gather(t: Tensor, I: Tensor) -> Tensor:
    """t[I] with I==-1 (missing neighbour) mapped to row 0; such slots always
    carry weight 0 (D=+inf).  The reference would wrap to the last row
    (decoder.py:146-147,163) -- a latent bug we do not reproduce."""
    return t[I.clamp_min(0)]


def geo_features(p, D, I, cnt, geo_feats, cloud, r2, min_nn, fallback, pts_grad, weighting="distance"):
    """MLP_geometry.get_feature_at_pos, decoder.py:130-173."""
    if pts_grad:  # is_tracker: re-evaluate D so that it carries d/dp (decoder.py:143-148)
        D = sqdist(p[:, None, :], safe_gather(cloud, I))
        D = torch.where(I < 0, torch.full_like(D, float("inf")), D)
    has_nb = cnt > (min_nn - 1)
    w = idw_weights(D, r2, weighting)
    c = (w[..., None] * safe_gather(geo_feats, I)).sum(1)
    c = torch.where(has_nb[:, None], c, fallback[None, :].expand_as(c))
    return c, has_nb


def col_features(p, D, I, cnt, col_feats, cloud, r2, min_nn, fallback, pts_grad, P: Dict[str, Tensor],
                 encode_rel_pos: bool, weighting="distance"):
    """MLP_color.get_feature_at_pos, decoder.py:341-390 (+ F_theta, decoder.py:225-240)."""
    if pts_grad:
        D = sqdist(p[:, None, :], safe_gather(cloud, I))
        D = torch.where(I < 0, torch.full_like(D, float("inf")), D)
    has_nb = cnt > (min_nn - 1)
    w = idw_weights(D, r2, weighting)
    nf = safe_gather(col_feats, I)                               # [P,K,C]
    if encode_rel_pos:
        rel = safe_gather(cloud, I) - p[:, None, :]              # decoder.py:372-373
        e = fourier(rel.reshape(-1, 3), P["color_decoder.embedder_rel_pos._B"], True)
        x = torch.cat([e.reshape(nf.shape[0], nf.shape[1], -1), nf], -1)
        h = F.linear(x, P["color_decoder.mlp_col_neighbor.linear1.weight"],
                     P["color_decoder.mlp_col_neighbor.linear1.bias"])
        h = softplus100(h)
        nf = F.linear(h, P["color_decoder.mlp_col_neighbor.linear2.weight"],
                      P["color_decoder.mlp_col_neighbor.linear2.bias"])
    c = (w[..., None] * nf).sum(1)
    c = torch.where(has_nb[:, None], c, fallback[None, :].expand_as(c))
    return c, has_nb


# --------------------------------------------------------------------------
# decoders
# --------------------------------------------------------------------------
def _trunk(prefix: str, emb: Tensor, c: Tensor, P: Dict[str, Tensor], act) -> Tensor:
    """Shared skeleton of both decoders: 5 blocks, skip after block 2.
    decoder.py:207-219 (geometry, ReLU) / 422-430 (colour, Softplus)."""
    h = emb
    for i in range(5):
        h = F.linear(h, P[f"{prefix}.pts_linears.{i}.weight"], P[f"{prefix}.pts_linears.{i}.bias"])
        h = act(h)
        h = h + F.linear(c, P[f"{prefix}.fc_c.{i}.weight"], P[f"{prefix}.fc_c.{i}.bias"])
        if i == 2:
            h = torch.cat([emb, h], -1)
    return F.linear(h, P[f"{prefix}.output_linear.weight"], P[f"{prefix}.output_linear.bias"])


def geo_mlp(p: Tensor, c: Tensor, P: Dict[str, Tensor]) -> Tensor:
    """MLP_geometry.forward trunk, decoder.py:203-222 -> occupancy logit [P]."""
    emb = fourier(p, P["geo_decoder.embedder._B"], False)
    return _trunk("geo_decoder", emb, c, P, F.relu).squeeze(-1)


def col_mlp(p: Tensor, c: Tensor, P: Dict[str, Tensor], exposure_affine: Optional[Tensor],
            apply_sigmoid: bool) -> Tensor:
    """MLP_color.forward trunk, decoder.py:411-449 -> rgb [P,3].
    exposure_affine: 12-vector (rot 3x3 row-major, trans) = mlp_exposure(exposure_feat)."""
    emb = fourier(p, P["color_decoder.embedder._B"], True)
    out = _trunk("color_decoder", emb, c, P, softplus100)
    if exposure_affine is not None:
        out = out @ exposure_affine[:9].reshape(3, 3) + exposure_affine[9:]
    return torch.sigmoid(out) if apply_sigmoid else out


def exposure_mlp(feat: Tensor, P: Dict[str, Tensor]) -> Tensor:
    """MLP_exposure.forward, decoder.py:243-258."""
    h = F.linear(feat, P["color_decoder.mlp_exposure.linear1.weight"],
                 P["color_decoder.mlp_exposure.linear1.bias"])
    return F.linear(softplus100(h), P["color_decoder.mlp_exposure.linear2.weight"],
                    P["color_decoder.mlp_exposure.linear2.bias"])


# --------------------------------------------------------------------------
# compositing
# --------------------------------------------------------------------------
def composite(raw: Tensor, z: Tensor, coef: float):
    """raw2outputs_nerf_color, src/common.py:298-336.  raw [R,S,4], z [R,S]."""
    alpha = torch.sigmoid(coef * raw[..., 3])
    T = torch.cumprod(torch.cat([torch.ones_like(alpha[:, :1]), 1.0 - alpha + 1e-10], -1), -1)[:, :-1]
    w = alpha * T
    wsum = w.sum(-1, keepdim=True) + 1e-10
    rgb = (w[..., None] * raw[..., :3]).sum(-2) / wsum
    depth = (w * z).sum(-1) / wsum.squeeze(-1)
    var = (w * (z - depth[:, None]) ** 2).sum(1)
    return depth, var, rgb, w


# --------------------------------------------------------------------------
# full render (the Seam-1 contract)
# --------------------------------------------------------------------------
def sample_near_pcl(cloud: Tensor, rays_o: Tensor, rays_d: Tensor, near: float, far: float, num: int,
                    radius_query: float, k: int = 8):
    """NeuralPointCloud.sample_near_pcl, src/neural_point.py:217-277: march 25 steps from near to far, a step
    'hits' when it has >= 1 neighbour closer than radius_query; rays with < 2 hits are invalid (uniform samples);
    the others get `num` samples between their FIRST TWO hit steps (float64 linspace, cast to f32)."""
    n = rays_o.shape[0]
    steps = 25
    z = torch.linspace(near, far, steps=steps)
    pts = (rays_o[:, None, :] + rays_d[:, None, :] * z[None, :, None]).reshape(-1, 3)
    D, _ = knn_exact(cloud, pts, k)
    hit = (neighbor_count(D, radius_query) > 0).reshape(n, steps)
    invalid = hit.sum(1) < 2
    def np_linspace(a, b, m):                  # numpy.linspace in float64: arange * step + start, endpoint forced
        a = torch.as_tensor(a, dtype=torch.float64).reshape(-1, 1)
        b = torch.as_tensor(b, dtype=torch.float64).reshape(-1, 1)
        y = torch.arange(m, dtype=torch.float64)[None, :] * ((b - a) / (m - 1)) + a
        y[:, -1] = b[:, 0]
        return y
    far = float(torch.tensor(far, dtype=torch.float32))                 # the reference passes a float32 tensor
    zs = np_linspace(near, far, steps)[0]                               # z_section (:247)
    out = np_linspace(near, far, num).repeat(n, 1)
    order = torch.argsort((~hit).int(), dim=1, stable=True)             # hit steps first, in step order
    seg = np_linspace(zs[order[:, 0]], zs[order[:, 1]], num)            # (:262-265)
    out = torch.where(invalid[:, None], out, seg)
    return out.float(), invalid


def render_z_vals(cfg: dict, cloud: Tensor, rays_o: Tensor, rays_d: Tensor, gt_depth: Tensor):
    """Per-ray sample depths incl. pixels without sensor depth, src/utils/Renderer.py:104-170.
    Returns z [R,S], near_pcl mask [R] (False only for depth-less rays far from the cloud)."""
    S = cfg["rendering"]["N_surface"]
    far = torch.minimum(5 * gt_depth.mean(), torch.max(gt_depth * 1.2))
    nz = gt_depth > 0
    z = torch.zeros(gt_depth.shape[0], S)
    z[nz] = z_samples(gt_depth[nz], cfg["rendering"]["near_end_surface"], cfg["rendering"]["far_end_surface"], S)
    near_mask = torch.ones(gt_depth.shape[0], dtype=torch.bool)
    if int(nz.sum()) < gt_depth.shape[0]:
        if cfg["rendering"]["sample_near_pcl"]:
            zz, inv = sample_near_pcl(cloud, rays_o[~nz].detach(), rays_d[~nz].detach(), cfg["rendering"]["near_end"],
                                      float(far), S, cfg["pointcloud"]["radius_query"], cfg["pointcloud"]["nn_num"])
            z[~nz] = zz
            idx = torch.nonzero(~nz).flatten()
            near_mask[idx[inv]] = False
        else:
            z[~nz] = torch.linspace(cfg["rendering"]["near_end"], float(far), steps=S).repeat(int((~nz).sum()), 1)
    return z, near_mask


def render_batch_ray(cfg: dict, P: Dict[str, Tensor], cloud: Tensor, geo_feats: Tensor, col_feats: Tensor,
                     rays_o: Tensor, rays_d: Tensor, gt_depth: Tensor, stage: str,
                     r_query: Optional[Tensor], fallback_geo: Tensor, fallback_col: Tensor,
                     pts_grad: bool = False, exposure_affine: Optional[Tensor] = None,
                     coef: float = 0.1, knn: Optional[Tuple[Tensor, Tensor]] = None):
    """Renderer.render_batch_ray, src/utils/Renderer.py:77-202, through POINT.forward (decoder.py:476-518);
    rays without sensor depth (gt_depth == 0) take the sample_near_pcl / uniform branch (:142-168).

    Returns depth[R], var[R], rgb[R,3], valid_ray[R], aux dict.
    """
    S = cfg["rendering"]["N_surface"]
    if bool((gt_depth > 0).all()):
        z = z_samples(gt_depth, cfg["rendering"]["near_end_surface"], cfg["rendering"]["far_end_surface"], S)
        near_mask = None
    else:
        z, near_mask = render_z_vals(cfg, cloud, rays_o, rays_d, gt_depth)
    pts = sample_points(rays_o, rays_d, z)
    if cfg["use_dynamic_radius"]:
        rq = r_query.reshape(-1, 1).repeat_interleave(S, 0)          # Renderer.py:179-181
        r2 = rq ** 2
    else:
        rq = cfg["pointcloud"]["radius_query"]
        r2 = rq ** 2
    K = cfg["pointcloud"]["nn_num"]
    if knn is None:
        D, I = knn_exact(cloud, pts.detach(), K)
    else:
        D, I = knn
    cnt = neighbor_count(D, rq)
    min_nn = cfg["pointcloud"]["min_nn_num"]
    weighting = cfg["pointcloud"].get("nn_weighting", "distance")
    cg, has_nb = geo_features(pts, D, I, cnt, geo_feats, cloud, r2, min_nn, fallback_geo, pts_grad, weighting)
    valid_ray = has_nb.view(-1, S).sum(1) >= int(S / 2 + 1)          # decoder.py:200-201
    occ = geo_mlp(pts, cg, P)
    if stage == "color":
        cc, _ = col_features(pts, D, I, cnt, col_feats, cloud, r2, min_nn, fallback_col, pts_grad, P,
                             cfg["model"]["encode_rel_pos_in_col"], weighting)
        sig = (not cfg["model"]["encode_exposure"]) or (exposure_affine is not None)
        rgb_pts = col_mlp(pts, cc, P, exposure_affine, sig)
    else:
        rgb_pts = torch.zeros(pts.shape[0], 3)
    # Renderer.py:189-190 writes -100 into raw IN PLACE under no_grad: the VALUE is
    # replaced but autograd still routes d/d(raw) to the decoder output unchanged
    # (straight-through).  Rays whose samples are all masked normalise by a tiny
    # sum of weights, so this gradient is not negligible (it feeds the tracker's pose).
    occ = occ + (torch.where(has_nb, occ, torch.full_like(occ, -100.0)) - occ).detach()
    raw = torch.cat([rgb_pts, occ[:, None]], -1).reshape(-1, S, 4)
    depth, var, rgb, w = composite(raw, z, coef)
    if near_mask is not None:
        valid_ray = valid_ray & near_mask                                # Renderer.py:198
        if not cfg["rendering"]["sample_near_pcl"]:
            depth = torch.where(gt_depth > 0, depth, torch.zeros_like(depth))   # Renderer.py:200-201
    return depth, var, rgb, valid_ray, {"D": D, "I": I, "cnt": cnt, "z": z, "pts": pts, "raw": raw, "w": w,
                                        "has_nb": has_nb}


# --------------------------------------------------------------------------
# losses
# --------------------------------------------------------------------------
def depth_inlier_mask(gt_depth: Tensor) -> Tensor:
    """d <= min(10*median(d), 1.2*max(d)); torch.median = LOWER median.
    src/Tracker.py:142-144, src/Mapper.py:507-509."""
    return gt_depth <= torch.minimum(10 * gt_depth.median(), 1.2 * gt_depth.max())


def tracker_loss(depth, var, rgb, gt_depth, gt_color, handle_dynamic=True, use_color=True, w_color=0.5):
    """src/Tracker.py:159-180. gt_color may be float64 (loader dtype), as in the reference."""
    u = var.detach()
    nan_mask = (~torch.isnan(depth)) & (~torch.isnan(u))
    if handle_dynamic:
        tmp = (gt_depth - depth).abs() / torch.sqrt(u + 1e-10)
        mask = (tmp < 10 * tmp.mean()) & (gt_depth > 0)
    else:
        tmp = (gt_depth - depth).abs()
        mask = (tmp < 10 * tmp.median()) & (gt_depth > 0)
    mask = mask & nan_mask
    geo = torch.clamp((gt_depth - depth).abs() / torch.sqrt(u + 1e-10), min=0.0, max=1e3)[mask].sum()
    col = (gt_color - rgb).abs()[mask].sum()
    loss = geo + w_color * col if use_color else geo
    return loss, geo, col, mask


def mapper_loss(depth, rgb, valid_ray, gt_depth, gt_color, stage: str, w_color=0.1):
    """src/Mapper.py:524-553 (no exposure)."""
    m = (gt_depth > 0) & valid_ray & (~torch.isnan(depth))
    geo = (gt_depth[m] - depth[m]).abs().sum()
    loss = geo.clone()
    col = torch.zeros(())
    if stage == "color":
        col = (gt_color[m] - rgb[m]).abs().sum()
        loss = loss + w_color * col
    return loss, geo, col, m


def tracker_iteration(cfg, P, cloud, geo_feats, col_feats, quat, trans, pix_idx, depth_img, color_img,
                      rq_img, cam, fb_geo, fb_col, edge_h, edge_w, coef=0.1):
    """One Tracker.optimize_cam_in_batch forward (src/Tracker.py:89-180) for pre-drawn
    flat pixel indices.  Returns loss, geo, col, mask and the leaf tensors' graph."""
    H, W = cam["H"], cam["W"]
    c2w = camera_from_tensor(quat, trans)
    u, v = pixels_from_flat_index(pix_idx.long(), edge_h, H - edge_h, edge_w, W - edge_w)
    ro, rd = rays_from_uv(u, v, c2w, cam["fx"], cam["fy"], cam["cx"], cam["cy"])
    ui, vi = u.long(), v.long()
    gd, gc = depth_img[vi, ui], color_img[vi, ui]
    rq = rq_img[vi, ui] if rq_img is not None else None
    keep = gd > 0                                                  # depth_filter, common.py:173-179
    ro, rd, gd, gc = ro[keep], rd[keep], gd[keep], gc[keep]
    rq = rq[keep] if rq is not None else None
    inl = depth_inlier_mask(gd)
    ro, rd, gd, gc = ro[inl], rd[inl], gd[inl], gc[inl]
    rq = rq[inl] if rq is not None else None
    depth, var, rgb, valid, aux = render_batch_ray(cfg, P, cloud, geo_feats, col_feats, ro, rd, gd, "color", rq,
                                                   fb_geo, fb_col, pts_grad=True, coef=coef)
    loss, geo, col, mask = tracker_loss(depth, var, rgb, gd, gc, cfg["tracking"]["handle_dynamic"],
                                        cfg["tracking"]["use_color_in_tracking"], cfg["tracking"]["w_color_loss"])
    return loss, geo, col, mask


# --------------------------------------------------------------------------
# Adam (torch.optim.Adam defaults; src/Mapper.py:394-402,556; src/Tracker.py:323,183)
# --------------------------------------------------------------------------
def adam_step(p: Tensor, g: Tensor, m: Tensor, v: Tensor, step: int, lr: float,
              b1=0.9, b2=0.999, eps=1e-8):
    """One dense Adam update (step is the 1-based count AFTER increment). Returns p,m,v."""
    m = b1 * m + (1 - b1) * g
    v = b2 * v + (1 - b2) * g * g
    bc1 = 1 - b1 ** step
    bc2 = 1 - b2 ** step
    denom = v.sqrt() / math.sqrt(bc2) + eps
    p = p - (lr / bc1) * (m / denom)
    return p, m, v


# --------------------------------------------------------------------------
# the iteration loops (src/Mapper.py:408-568, src/Tracker.py:296-350), replayed for PRE-DRAWN random numbers
# --------------------------------------------------------------------------
class _Adam:
    """torch.optim.Adam as the loops use it: one step counter per tensor, tensors whose gradient is None are
    skipped (their counter does not advance) -- the semantics of torch >= 2.0's zero_grad(set_to_none=True)."""

    def __init__(self, zero_grad_names=()):
        self.state = {}
        # torch 1.12 (the reference's env.yaml:61): zero_grad() leaves ZERO TENSORS on parameters that have had a gradient
        # before, and Adam steps them (the counter advances; m = v = 0 keep p where it is).  Names listed here get a
        # zero gradient instead of None -- the colour-decoder tensors from the second mapped frame on.
        self.zero_grad_names = set(zero_grad_names)

    def step(self, name: str, p: Tensor, g: Optional[Tensor], lr: float) -> Tensor:
        if g is None and name in self.zero_grad_names:
            g = torch.zeros_like(p)
        if g is None:
            return p
        st = self.state.setdefault(name, dict(step=0, m=torch.zeros_like(p), v=torch.zeros_like(p)))
        st["step"] += 1
        q, st["m"], st["v"] = adam_step(p, g, st["m"], st["v"], st["step"], lr)
        return q


def mapper_rays(frames, pix_it, cam):
    """get_samples per window frame (common.py:162-183, depth_filter) + concatenation + depth-outlier mask
    (Mapper.py:455-514).  frames: dicts depth/color/c2w/r_query; pix_it [F][ppf] flat indices into the full image.
    Returns rays_o, rays_d, gt_depth, gt_color, r_query (or None), frame id per ray."""
    H, W = cam["H"], cam["W"]
    ros, rds, gds, gcs, rqs, fids = [], [], [], [], [], []
    for f, fr in enumerate(frames):
        u, v = pixels_from_flat_index(pix_it[f].long(), 0, H, 0, W)
        ro, rd = rays_from_uv(u, v, fr["c2w"], cam["fx"], cam["fy"], cam["cx"], cam["cy"])
        ui, vi = u.long(), v.long()
        gd = fr["depth"][vi, ui]
        keep = gd > 0
        ros.append(ro[keep]); rds.append(rd[keep]); gds.append(gd[keep]); gcs.append(fr["color"][vi, ui][keep].float())
        if fr.get("r_query") is not None:
            rqs.append(fr["r_query"][vi, ui][keep])
        fids.append(torch.full((int(keep.sum()),), f, dtype=torch.long))
    ro, rd, gd, gc, fid = torch.cat(ros), torch.cat(rds), torch.cat(gds), torch.cat(gcs), torch.cat(fids)
    rq = torch.cat(rqs) if rqs else None
    inl = depth_inlier_mask(gd)
    return ro[inl], rd[inl], gd[inl], gc[inl], (rq[inl] if rq is not None else None), fid[inl]


def mapper_iterations(cfg, P, cloud, geo_feats, col_feats, sel, frames, pix, fb, n_geo_iters, cam, coef=0.1,
                      exposure_feats=None, init=False, record=None, torch1_zero_grads=False, lr_override=None,
                      train_decoder=None):
    """The joint_iter loop of Mapper.optimize_map (src/Mapper.py:408-568, no BA) for pre-drawn pixels `pix`
    [n_iters][F][ppf] and fallback vectors `fb` [n_iters][2][32].  The frustum-selected rows `sel` of both feature
    sets, the colour decoder (all of color_decoder.parameters(), :358-360) and -- ScanNet -- the current frame's
    exposure vector (lr 0.001, :399-401) are optimised; frames[-1] is the current frame.
    exposure_feats: list of [8] tensors per window frame or None.  Returns (losses, geo_feats, col_feats, P,
    exposure of the current frame)."""
    mp = cfg["mapping"]
    stage_tab = mp["init" if init else "stage"]
    sel = sel.long()
    geo_p = geo_feats[sel].detach().clone()
    col_p = col_feats[sel].detach().clone()
    P = {k: v.detach().clone() for k, v in P.items()}
    train_keys = [k for k in P if k.startswith("color_decoder.") and k != "color_decoder.embedder._B"
                  and P[k].dtype.is_floating_point] if not mp["fix_color_decoder"] else []
    if train_decoder is not None and not train_decoder:
        train_keys = []                     # colour refinement: fix_color_decoder forced (Mapper.py:717)
    exposure = cfg["model"]["encode_exposure"]
    ex_cur = exposure_feats[-1].detach().clone() if exposure else None
    # torch1_zero_grads: the decoder tensors carry zero .grad tensors from the previous mapped frame (see _Adam)
    opt = _Adam(train_keys if torch1_zero_grads else ())
    n_iters = pix.shape[0]
    losses = []
    for it in range(n_iters):
        stage = "geometry" if it <= n_geo_iters else "color"                       # Mapper.py:420-423
        lr = stage_tab[stage]
        if lr_override is not None:                                                # colour refinement, Mapper.py:427-430
            lr = lr_override
        gp, cp = geo_p.clone().requires_grad_(True), col_p.clone().requires_grad_(True)
        Pg = {k: (v.clone().requires_grad_(True) if k in train_keys else v) for k, v in P.items()}
        ex = ex_cur.clone().requires_grad_(True) if exposure else None
        g_full = geo_feats.detach().index_put((sel,), gp)                          # :413-414
        c_full = col_feats.detach().index_put((sel,), cp)
        ro, rd, gd, gc, rq, fid = mapper_rays(frames, pix[it], cam)
        depth, var, rgb, valid, _ = render_batch_ray(cfg, Pg, cloud, g_full, c_full, ro, rd, gd, stage, rq,
                                                     fb[it, 0], fb[it, 1], pts_grad=False, coef=coef)
        m = (gd > 0) & valid & (~torch.isnan(depth))                               # :524-526
        loss = (gd[m] - depth[m]).abs().sum()
        if stage == "color":
            if exposure:                                                           # :530-548, frame by frame
                rgb = rgb.clone()
                for f in range(len(frames)):
                    rows = torch.nonzero(fid == f).flatten()
                    if rows.numel() == 0:
                        continue
                    a, b = int(rows[0]), int(rows[-1]) + 1
                    aff = exposure_mlp(ex if f == len(frames) - 1 else exposure_feats[f], Pg)
                    rgb[a:b] = rgb[a:b].clone() @ aff[:9].reshape(3, 3) + aff[9:]
                rgb = torch.sigmoid(rgb)
            loss = loss + mp["w_color_loss"] * (gc[m] - rgb[m]).abs().sum()
        loss.backward()
        losses.append(float(loss.detach()))
        geo_p = opt.step("geo", geo_p, gp.grad, lr["geometry_lr"])
        col_p = opt.step("col", col_p, cp.grad, lr["color_lr"])
        for k in train_keys:
            P[k] = opt.step(k, P[k], Pg[k].grad, lr["decoders_lr"])
        if exposure:
            ex_cur = opt.step("exposure", ex_cur, ex.grad, 0.001)
        if record is not None:
            record(it, geo_p, col_p, P, ex_cur)
    geo_out, col_out = geo_feats.detach().clone(), col_feats.detach().clone()
    geo_out[sel], col_out[sel] = geo_p, col_p
    return losses, geo_out, col_out, P, ex_cur, opt


def tracker_loop(cfg, P, cloud, geo_feats, col_feats, cam0, pix, fb, depth_img, color_img, rq_img, cam, edge_h,
                 edge_w, coef=0.1, exposure_feat=None, full_image_index=False):
    """The cam_iter loop of Tracker.run (src/Tracker.py:296-350) around optimize_cam_in_batch (:89-186): Adam on
    T (lr) and the quaternion (0.2 lr, separate_LR) and -- encode_exposure -- on the frame's exposure vector and
    mlp_exposure (both lr 0.001, :305-311).  pix [n_iters][n]: flat indices into the cropped window, or into the
    FULL image when full_image_index (sample_with_color_grad, :115-128: no depth filter there).
    Returns per-iteration losses, camera tensors [n_iters][7], the lowest-loss camera tensor, the exposure vector
    and the (updated) parameter dict."""
    tr = cfg["tracking"]
    H, W = cam["H"], cam["W"]
    quat, trans = cam0[:4].detach().clone(), cam0[4:].detach().clone()
    P = {k: v.detach().clone() for k, v in P.items()}
    exposure = cfg["model"]["encode_exposure"]
    ex_keys = [k for k in P if k.startswith("color_decoder.mlp_exposure.")] if exposure else []
    ex = exposure_feat.detach().clone() if exposure else None
    lr = tr["lr"]
    lr_q = lr * 0.2 if tr["separate_LR"] else lr
    opt = _Adam()
    losses, cams = [], []
    best, best_loss = None, float("inf")
    for it in range(pix.shape[0]):
        q, t = quat.clone().requires_grad_(True), trans.clone().requires_grad_(True)
        Pg = {k: (v.clone().requires_grad_(True) if k in ex_keys else v) for k, v in P.items()}
        e = ex.clone().requires_grad_(True) if exposure else None
        c2w = camera_from_tensor(q, t)
        if full_image_index:
            vi, ui = torch.div(pix[it].long(), W, rounding_mode="floor"), pix[it].long() % W
            u, v = ui.float(), vi.float()
        else:
            u, v = pixels_from_flat_index(pix[it].long(), edge_h, H - edge_h, edge_w, W - edge_w)
            ui, vi = u.long(), v.long()
        ro, rd = rays_from_uv(u, v, c2w, cam["fx"], cam["fy"], cam["cx"], cam["cy"])
        gd, gc = depth_img[vi, ui], color_img[vi, ui]
        rq = rq_img[vi, ui] if cfg["use_dynamic_radius"] else None
        if not full_image_index:
            keep = gd > 0
            ro, rd, gd, gc = ro[keep], rd[keep], gd[keep], gc[keep]
            rq = rq[keep] if rq is not None else None
        inl = depth_inlier_mask(gd)
        ro, rd, gd, gc = ro[inl], rd[inl], gd[inl], gc[inl]
        rq = rq[inl] if rq is not None else None
        aff = exposure_mlp(e, Pg) if exposure else None
        depth, var, rgb, valid, _ = render_batch_ray(cfg, Pg, cloud, geo_feats, col_feats, ro, rd, gd, "color", rq,
                                                     fb[it, 0], fb[it, 1], pts_grad=True, exposure_affine=aff,
                                                     coef=coef)
        loss, geo, col, mask = tracker_loss(depth, var, rgb, gd, gc, tr["handle_dynamic"], tr["use_color_in_tracking"],
                                            tr["w_color_loss"])
        loss.backward()
        lv = float(loss.detach())
        if lv < best_loss:
            best_loss, best = lv, torch.cat([quat, trans]).clone()
        trans = opt.step("T", trans, t.grad, lr)
        quat = opt.step("quat", quat, q.grad, lr_q)
        if exposure:
            ex = opt.step("exposure", ex, e.grad, 0.001)
            for k in ex_keys:
                P[k] = opt.step(k, P[k], Pg[k].grad, 0.001)
        losses.append(lv)
        cams.append(torch.cat([quat, trans]).clone())
    return losses, torch.stack(cams), best, ex, P


# --------------------------------------------------------------------------
# point growth
# --------------------------------------------------------------------------
def add_points_select(cloud: Tensor, rays_o, rays_d, gt_depth, radius, n_add=3, near=0.98, far=1.02,
                      k: int = 8):
    """NeuralPointCloud.add_neural_points selection rule, src/neural_point.py:106-146.

    keep locations whose surface point has ZERO existing neighbours with D < r^2
    (dedupe only against the cloud before this batch); each kept location
    yields n_add points at linspace(near,far,n_add)*depth along the ray.
    Returns (new_pts [3*kept,3], keep_mask [n] over depth>0 rays, surface pts).
    """
    pos = gt_depth > 0
    rays_o, rays_d, gt_depth = rays_o[pos], rays_d[pos], gt_depth[pos]
    if torch.is_tensor(radius):
        radius = radius[pos]
    surf = rays_o + rays_d * gt_depth[:, None]
    if cloud.shape[0] > 0:
        D, _ = knn_exact(cloud, surf, k)
        keep = neighbor_count(D, radius) == 0
    else:
        keep = torch.ones(surf.shape[0], dtype=torch.bool)
    t = torch.linspace(0.0, 1.0, steps=n_add)
    d = gt_depth[:, None].repeat(1, n_add)
    z = near * d * (1.0 - t) + far * d * t
    pts = rays_o[:, None, :] + rays_d[:, None, :] * z[:, :, None]
    return pts[keep].reshape(-1, 3), keep, surf


# --------------------------------------------------------------------------
# frustum feature selection (src/Mapper.py:120-168).  The depth lookup is cv2.remap(..., INTER_LINEAR)
# (:149-155); cv2 is absent from the image and the reference holds no vectors for it: PARITY UNPINNED at this seam.
#   remap="cv2"   (default, round 4): OpenCV's INTER_LINEAR restated from its published source (imgproc/src/imgwarp.cpp,
#                 remap() with CV_32FC1 maps + remapBilinear<float>, 4.x): coordinates in fixed point with INTER_BITS = 5
#                 (sx = cvRound(u * 32), integer part sx >> 5 saturated to int16, fraction (sx & 31) / 32), weights
#                 w[ky][kx] = vy[ky] * vx[kx] from the float table, taps summed left to right, constant-0 border;
#   remap="exact" : plain bilinear interpolation (rounds 1-3; what oracle/gen_golden_loops.py used to select the rows of the
#                 committed mapper fixtures -- the generator keeps passing it so that the fixtures regenerate bit for bit).
# --------------------------------------------------------------------------
def frustum_select(cloud: Tensor, c2w: Tensor, depth_img: Tensor, H, W, fx, fy, cx, cy, edge: float, remap: str = "cv2"):
    w2c = torch.linalg.inv(c2w.double())
    pc = (w2c[:3, :3] @ cloud.double().T + w2c[:3, 3:4]).T
    x, y, zc = -pc[:, 0], pc[:, 1], pc[:, 2]
    z = zc + 1e-5
    u = ((fx * x + cx * zc) / z).float()
    v = ((fy * y + cy * zc) / z).float()
    d = remap_linear_cv2(depth_img, u, v) if remap == "cv2" else bilinear_zero_border(depth_img, u, v)
    inb = (u < W - edge) & (u > edge) & (v < H - edge) & (v > edge)
    d = torch.where(d == 0, d.max(), d)
    mz = (-z).float()
    return torch.nonzero(inb & (mz >= 0) & (mz <= d + 0.5)).flatten()


def bilinear_zero_border(img: Tensor, u: Tensor, v: Tensor) -> Tensor:
    H, W = img.shape
    u0, v0 = torch.floor(u), torch.floor(v)
    fu, fv = u - u0, v - v0
    out = torch.zeros_like(u)
    for dv, wv in ((0, 1 - fv), (1, fv)):
        for du, wu in ((0, 1 - fu), (1, fu)):
            uu, vv = (u0 + du).long(), (v0 + dv).long()
            ok = (uu >= 0) & (uu < W) & (vv >= 0) & (vv < H)
            val = img[vv.clamp(0, H - 1), uu.clamp(0, W - 1)]
            out = out + torch.where(ok, val, torch.zeros_like(val)) * wu * wv
    return out


def remap_linear_cv2(img: Tensor, u: Tensor, v: Tensor) -> Tensor:
    """cv2.remap(img, u, v, interpolation=cv2.INTER_LINEAR) for a float32 image and float32 coordinate maps, default
    BORDER_CONSTANT 0 (see the block comment above).  torch.round is round-half-to-even like cvRound."""
    H, W = img.shape
    img = img.float()
    far = ~((u.abs() < 1.0e6) & (v.abs() < 1.0e6))                  # cv2 saturates to int16: every tap is border (NaN too)
    uq = torch.where(far, torch.zeros_like(u), u)
    vq = torch.where(far, torch.zeros_like(v), v)
    sx = torch.round(uq.float() * 32.0).long()
    sy = torch.round(vq.float() * 32.0).long()
    ix, iy = (sx >> 5).clamp(-32768, 32767), (sy >> 5).clamp(-32768, 32767)
    fx, fy = (sx & 31).float() * (1.0 / 32.0), (sy & 31).float() * (1.0 / 32.0)
    vx, vy = (1.0 - fx, fx), (1.0 - fy, fy)
    out = torch.zeros_like(uq, dtype=torch.float32)
    for ky in range(2):
        for kx in range(2):
            xx, yy = ix + kx, iy + ky
            ok = (xx >= 0) & (xx < W) & (yy >= 0) & (yy < H)
            val = torch.where(ok, img[yy.clamp(0, H - 1), xx.clamp(0, W - 1)], torch.zeros_like(out))
            out = out + val * (vy[ky] * vx[kx])
    return torch.where(far, torch.zeros_like(out), out)
