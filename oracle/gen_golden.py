"""Generate the golden fixtures under tests/golden/ by running the UNMODIFIED
reference (imported from /root/reference through oracle/ref_import.py) on
seeded synthetic inputs, and check the oracle restatement against it.

Run in the build container only (the GPU box has no /root/reference):
    python -m oracle.gen_golden

Each fixture is an .npz holding the exact inputs and the reference's outputs
(+ autograd gradients).  tests/test_oracle_golden.py replays the inputs
through the oracle; tests/test_hip_parity.py (gpu) replays them through the
HIP path.
"""
from __future__ import annotations

import copy
import os
import sys
import types

import numpy as np
import torch
import yaml

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)

from oracle import pointslam_oracle as O  # noqa: E402
from oracle import ref_import as RI  # noqa: E402
from point_slam_amd import synthetic as syn  # noqa: E402

GOLD = os.path.join(ROOT, "tests", "golden")


def base_cfg():
    cfg = yaml.safe_load(open(os.path.join(RI.REF, "configs", "point_slam.yaml")))
    cfg["mapping"]["device"] = "cpu"
    cfg["tracking"]["device"] = "cpu"
    return cfg


def cfg_variant(name):
    cfg = base_cfg()
    if name == "replica":          # dynamic radius + per-neighbour colour MLP (base yaml)
        pass
    elif name == "tum":            # fixed radius, no rel-pos MLP (configs/TUM_RGBD/tum.yaml)
        cfg["use_dynamic_radius"] = False
        cfg["model"]["encode_rel_pos_in_col"] = False
    elif name == "replica_expo":   # pointcloud.nn_weighting = 'expo' (decoder.py:154-156, 364-366; no shipped config)
        cfg["pointcloud"]["nn_weighting"] = "expo"
    elif name == "scannet":        # exposure, rho=0.04 (configs/ScanNet/scannet.yaml)
        cfg["model"]["encode_rel_pos_in_col"] = False
        cfg["model"]["encode_exposure"] = True
        cfg["rendering"]["near_end_surface"] = 0.96
        cfg["rendering"]["far_end_surface"] = 1.04
    return cfg


def build_scene(cfg, n_pts, n_rays, seed, frame_t=3.0, sparse_frac=0.0, W=640, H=480):
    """Synthetic cloud + a batch of rays with sensor depth/colour/radius."""
    cam = syn.intrinsics(W, H)
    g = torch.Generator().manual_seed(seed)
    c2w = syn.pose(frame_t)
    depth, color = syn.render_frame(cam, c2w)
    _, r_query = syn.dynamic_radii(color, cfg)
    # cloud: back-project random pixels of nearby views (so the query frame sees neighbours)
    pts = []
    t = torch.linspace(0.0, 1.0, 3)
    for v in range(4):
        cw = syn.pose(frame_t + 1.5 * (v - 1.5))
        u = torch.rand(n_pts // 12 + 1, generator=g) * (cam["W"] - 1)
        w = torch.rand(n_pts // 12 + 1, generator=g) * (cam["H"] - 1)
        ro, rd = O.rays_from_uv(u, w, cw, cam["fx"], cam["fy"], cam["cx"], cam["cy"])
        d = syn.box_depth(ro, rd)
        z = 0.98 * d[:, None] * (1 - t) + 1.02 * d[:, None] * t
        pts.append((ro[:, None] + rd[:, None] * z[..., None]).reshape(-1, 3))
    cloud = torch.cat(pts)[:n_pts].float().contiguous()
    if sparse_frac > 0:  # carve holes so that some samples have <2 neighbours
        keep = cloud[:, 1] > cloud[:, 1].quantile(sparse_frac)
        cloud = cloud[keep].contiguous()
    N = cloud.shape[0]
    geo = torch.zeros(N, 32).normal_(0, 0.1, generator=g)
    col = torch.zeros(N, 32).normal_(0, 0.1, generator=g)
    ui = torch.randint(20, cam["W"] - 20, (n_rays,), generator=g)
    vi = torch.randint(20, cam["H"] - 20, (n_rays,), generator=g)
    gd = depth[vi, ui].clone()
    gc = color[vi, ui].clone()
    rq = r_query[vi, ui].clone()
    return dict(cam=cam, c2w=c2w, cloud=cloud, geo=geo, col=col, ui=ui.float(), vi=vi.float(), gt_depth=gd,
                gt_color=gc, r_query=rq)


def make_ref_npc(ns, cfg, cloud, geo, col):
    npc = ns.neural_point.NeuralPointCloud(cfg)
    npc._cloud_pos = cloud.tolist()
    npc._pts_num = cloud.shape[0]
    npc.geo_feats, npc.col_feats = geo.clone(), col.clone()
    npc.index.train(cloud)
    npc.index.add(cloud)
    return npc


def run_render_case(name, cfg_name, stage, is_tracker, n_pts, n_rays, seed, sparse_frac=0.0, exposure=False,
                    store_param_grads=False, zero_depth_frac=0.0, sample_near_pcl=None):
    ns = RI.load()
    cfg = cfg_variant(cfg_name)
    if sample_near_pcl is not None:
        cfg["rendering"]["sample_near_pcl"] = bool(sample_near_pcl)
    dec = RI.make_decoders(cfg)
    P = RI.state_with_fixed_B(dec)
    sc = build_scene(cfg, n_pts, n_rays, seed, sparse_frac=sparse_frac)
    if zero_depth_frac > 0:   # sensor holes: these rays take the sample_near_pcl / uniform branch
        hole = torch.rand(n_rays, generator=torch.Generator().manual_seed(seed + 5)) < zero_depth_frac
        sc["gt_depth"] = torch.where(hole, torch.zeros_like(sc["gt_depth"]), sc["gt_depth"])
    cam = sc["cam"]
    npc = make_ref_npc(ns, cfg, sc["cloud"], sc["geo"], sc["col"])
    slam = types.SimpleNamespace(**cam)
    rend = ns.Renderer(cfg, None, slam)
    rend.sigmoid_coefficient = cfg["rendering"]["sigmoid_coef_mapper"]
    wrapped = RI.PointCPU(dec)
    for p in dec.parameters():
        p.requires_grad_(True)

    c2w = sc["c2w"].clone()
    rays_o, rays_d = ns.common.get_rays_from_uv(sc["ui"], sc["vi"], c2w, cam["fx"], cam["fy"], cam["cx"],
                                               cam["cy"], "cpu")
    rays_o = rays_o.clone().requires_grad_(True)
    rays_d = rays_d.clone().requires_grad_(True)
    geo = sc["geo"].clone().requires_grad_(True)
    col = sc["col"].clone().requires_grad_(True)
    exposure_feat = None
    if exposure:
        exposure_feat = torch.zeros(cfg["model"]["exposure_dim"]).normal_(0, 0.5,
                                                                         generator=torch.Generator().manual_seed(seed))
        exposure_feat.requires_grad_(True)
    # the reference draws its two fallback vectors from the global RNG, geo first (decoder.py:170,387)
    torch.manual_seed(seed + 7)
    fb_geo = torch.zeros(32).normal_(mean=0, std=0.01)
    fb_col = torch.zeros(32).normal_(mean=0, std=0.01)
    torch.manual_seed(seed + 7)
    depth, var, rgb, valid = rend.render_batch_ray(
        npc, wrapped, rays_d, rays_o, "cpu", stage, gt_depth=sc["gt_depth"], npc_geo_feats=geo,
        npc_col_feats=col, is_tracker=is_tracker, cloud_pos=sc["cloud"],
        dynamic_r_query=sc["r_query"] if cfg["use_dynamic_radius"] else None, exposure_feat=exposure_feat)
    # scalar for gradients: fixed random cotangents
    g = torch.Generator().manual_seed(seed + 11)
    w_d = torch.randn(depth.shape, generator=g)
    w_c = torch.randn(rgb.shape, generator=g)
    w_v = torch.randn(var.shape, generator=g) * 10.0
    obj = (depth * w_d).sum() + (rgb * w_c).sum() + (var * w_v).sum()
    obj.backward()
    ref = dict(depth=depth, var=var, rgb=rgb, valid=valid, g_rays_o=rays_o.grad, g_rays_d=rays_d.grad,
               g_geo=geo.grad, g_col=col.grad if col.grad is not None else torch.zeros_like(col))
    pgrads = {k: (p.grad.clone() if p.grad is not None else torch.zeros_like(p))
              for k, p in dec.named_parameters()}
    if exposure:
        ref["g_exposure_feat"] = exposure_feat.grad

    # ---- oracle replay on identical inputs -------------------------------
    Pg = {k: v.clone().requires_grad_(True) for k, v in P.items()}
    ro2 = rays_o.detach().clone().requires_grad_(True)
    rd2 = rays_d.detach().clone().requires_grad_(True)
    geo2 = sc["geo"].clone().requires_grad_(True)
    col2 = sc["col"].clone().requires_grad_(True)
    aff = None
    if exposure:
        ef2 = exposure_feat.detach().clone().requires_grad_(True)
        aff = O.exposure_mlp(ef2, Pg)
    d2, v2, c2, val2, aux = O.render_batch_ray(
        cfg, Pg, sc["cloud"], geo2, col2, ro2, rd2, sc["gt_depth"], stage,
        sc["r_query"] if cfg["use_dynamic_radius"] else None, fb_geo, fb_col, pts_grad=is_tracker,
        exposure_affine=aff, coef=rend.sigmoid_coefficient)
    obj2 = (d2 * w_d).sum() + (c2 * w_c).sum() + (v2 * w_v).sum()
    obj2.backward()

    def rel(a, b):
        return float((a - b).abs().max() / (b.abs().max() + 1e-30))
    rep = dict(depth=rel(d2, depth), var=rel(v2, var), rgb=rel(c2, rgb),
               valid=bool((val2 == valid).all()),
               g_rays_o=rel(ro2.grad, rays_o.grad), g_rays_d=rel(rd2.grad, rays_d.grad),
               g_geo=rel(geo2.grad, geo.grad))
    if stage == "color":
        rep["g_col"] = rel(col2.grad, col.grad)
        rep["g_W"] = max(rel(Pg[k].grad, pgrads[k]) for k in pgrads
                         if Pg[k].grad is not None and pgrads[k].abs().max() > 0)
    print(f"[{name}] N={sc['cloud'].shape[0]} R={n_rays} has_nb={float(aux['has_nb'].float().mean()):.3f} "
          f"valid={float(valid.float().mean()):.3f} oracle-vs-reference:", rep)
    tol = 2e-5
    assert rep["valid"], name
    for k, v in rep.items():
        if k != "valid":
            assert v < (1e-3 if k.startswith("g_") else tol), (name, k, v)

    out = dict(cfg_name=cfg_name, stage=stage, is_tracker=is_tracker, coef=rend.sigmoid_coefficient,
               sample_near_pcl=bool(cfg["rendering"]["sample_near_pcl"]),
               cloud=sc["cloud"], geo=sc["geo"], col=sc["col"], rays_o=rays_o.detach(), rays_d=rays_d.detach(),
               gt_depth=sc["gt_depth"], gt_color=sc["gt_color"], r_query=sc["r_query"], fb_geo=fb_geo,
               fb_col=fb_col, w_d=w_d, w_c=w_c, w_v=w_v, c2w=c2w, ui=sc["ui"], vi=sc["vi"],
               I=aux["I"].int(), has_nb=aux["has_nb"])
    if exposure:
        out["exposure_feat"] = exposure_feat.detach()
    out.update({"ref_" + k: v.detach() for k, v in ref.items()})
    # feature gradients are row-sparse: store touched rows only
    for nm in ("g_geo", "g_col"):
        gfull = out.pop("ref_" + nm)
        rows = torch.nonzero(gfull.abs().sum(1) > 0).flatten()
        out["ref_" + nm + "_rows"] = rows.int()
        out["ref_" + nm + "_vals"] = gfull[rows]
    # parameter gradients: store the non-zero ones
    if store_param_grads:
        for k, v in pgrads.items():
            if v.abs().max() > 0:
                out["refgp_" + k] = v
    else:
        for k, v in pgrads.items():
            if v.abs().max() > 0:
                out["refgpnorm_" + k] = v.double().norm().float()
    save(name, out)
    return cfg, P


def save(name, d):
    arrs = {}
    for k, v in d.items():
        if torch.is_tensor(v):
            arrs[k] = v.detach().cpu().numpy()
        elif isinstance(v, (bool, int, float, str)):
            arrs[k] = np.array(v)
        else:
            arrs[k] = v
    path = os.path.join(GOLD, name + ".npz")
    np.savez_compressed(path, **arrs)
    print("  wrote", path, f"{os.path.getsize(path)/1024:.0f} KiB")


def run_tracker_case(name, n_pts, n_rays, seed, handle_dynamic=True):
    """One real Tracker.optimize_cam_in_batch iteration (src/Tracker.py:89-186) with a fake self.
    handle_dynamic=False: the median-mask branch (Tracker.py:166-168; no shipped config uses it) on the SAME scene and draws --
    only the reference's results are stored (the inputs are those of the handle_dynamic=True fixture of the same seed)."""
    ns = RI.load()
    if ns.tracker_mod is None:
        print("Tracker module not importable:", ns.tracker_err)
        return
    T = ns.tracker_mod
    cfg = cfg_variant("replica")
    dec = RI.make_decoders(cfg)
    P = RI.state_with_fixed_B(dec)
    sc = build_scene(cfg, n_pts, n_rays, seed, W=160, H=120)
    cam = sc["cam"]
    npc = make_ref_npc(ns, cfg, sc["cloud"], sc["geo"], sc["col"])
    rend = ns.Renderer(cfg, None, types.SimpleNamespace(**cam))
    rend.sigmoid_coefficient = cfg["rendering"]["sigmoid_coef_tracker"]
    depth_img, color_img = syn.render_frame(cam, sc["c2w"])
    _, rq_img = syn.dynamic_radii(color_img, cfg)

    # CPU-safe quaternion chain: same arithmetic as common.py:225-267
    def quad2rot_cpu(quad):
        qr, qi, qj, qk = quad[:, 0], quad[:, 1], quad[:, 2], quad[:, 3]
        two_s = 2.0 / (quad * quad).sum(-1)
        R = torch.zeros(quad.shape[0], 3, 3)
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
    ns.common.quad2rotation = quad2rot_cpu

    # perturbed initial pose
    cam_t = ns.common.get_tensor_from_camera(sc["c2w"])
    cam_t = cam_t + torch.tensor([0.002, -0.001, 0.0015, 0.001, 0.01, -0.008, 0.006])
    quad = cam_t[:4].clone().requires_grad_(True)
    Tt = cam_t[4:].clone().requires_grad_(True)
    camera_tensor = torch.cat([quad, Tt], 0)
    lr = cfg["tracking"]["lr"]
    opt = torch.optim.Adam([{"params": [Tt], "lr": lr}, {"params": [quad], "lr": lr * 0.2}])
    fake = types.SimpleNamespace(
        device="cpu", npc=npc, H=cam["H"], W=cam["W"], ignore_edge_W=20, ignore_edge_H=20,
        sample_with_color_grad=False, fx=cam["fx"], fy=cam["fy"], cx=cam["cx"], cy=cam["cy"],
        depth_limit=False, use_dynamic_radius=True, dynamic_r_query=rq_img, renderer=rend,
        decoders=RI.PointCPU(dec), npc_geo_feats=sc["geo"].clone(), npc_col_feats=sc["col"].clone(),
        cloud_pos=sc["cloud"], exposure_feat=None, handle_dynamic=bool(handle_dynamic), use_color_in_tracking=True,
        w_color_loss=cfg["tracking"]["w_color_loss"])
    torch.manual_seed(seed + 3)
    # record the pixel indices the reference will draw: select_uv uses torch.randint on the global RNG
    Hc, Wc = cam["H"] - 40, cam["W"] - 40
    idx = torch.randint(Hc * Wc, (n_rays,))
    torch.manual_seed(seed + 3)
    fb_state = None
    loss, closs, gloss = T.Tracker.optimize_cam_in_batch(fake, camera_tensor, color_img.double(), depth_img,
                                                        n_rays, opt)
    out = dict(cloud=sc["cloud"], geo=sc["geo"], col=sc["col"], c2w_gt=sc["c2w"], cam0=cam_t,
               pix_idx=idx.int(), depth_img=depth_img, color_img=color_img, rq_img=rq_img,
               ref_loss=np.float64(loss), ref_color_loss_pp=np.float64(closs), ref_geo_loss_pp=np.float64(gloss),
               ref_quad_after=quad.detach(), ref_T_after=Tt.detach(), seed=seed, n_rays=n_rays)
    # fallback vectors: drawn after randint from the same stream (geo then colour)
    torch.manual_seed(seed + 3)
    torch.randint(Hc * Wc, (n_rays,))
    out["fb_geo"] = torch.zeros(32).normal_(mean=0, std=0.01)
    out["fb_col"] = torch.zeros(32).normal_(mean=0, std=0.01)
    print(f"[{name}] reference tracker iteration: loss={loss:.6f} col/px={closs:.6f} geo/px={gloss:.6f}")
    # oracle replay: loss, then torch Adam on the same leaves
    q2 = cam_t[:4].clone().requires_grad_(True)
    t2 = cam_t[4:].clone().requires_grad_(True)
    cfg["tracking"]["handle_dynamic"] = bool(handle_dynamic)
    l2, g2, c2, m2 = O.tracker_iteration(cfg, P, sc["cloud"], sc["geo"], sc["col"], q2, t2, idx, depth_img,
                                         color_img.double(), rq_img, cam, out["fb_geo"], out["fb_col"], 20, 20,
                                         coef=rend.sigmoid_coefficient)
    l2.backward()
    tn, _, _ = O.adam_step(t2.detach(), t2.grad, torch.zeros(3), torch.zeros(3), 1, lr)
    qn, _, _ = O.adam_step(q2.detach(), q2.grad, torch.zeros(4), torch.zeros(4), 1, lr * 0.2)
    print("   oracle loss", float(l2), "rel", abs(float(l2) - loss) / loss,
          "dT", float((tn - Tt.detach()).abs().max()), "dq", float((qn - quad.detach()).abs().max()))
    assert abs(float(l2) - loss) / loss < 1e-5
    assert float((tn - Tt.detach()).abs().max()) < 1e-6 and float((qn - quad.detach()).abs().max()) < 1e-6
    out["ref_g_quad_sign"] = torch.sign(q2.grad)
    out["ref_mask_count"] = int(m2.sum())
    if not handle_dynamic:      # results only: the scene and the draws are the dynamic fixture's
        out = {k: v for k, v in out.items() if k.startswith("ref_") or k in ("seed", "n_rays")}
        out["handle_dynamic"] = False
    save(name, out)


def run_expo_cases():
    """pointcloud.nn_weighting = 'expo'.  The TRACKER cannot run with it in the reference: decoder.py:157 / 367 zero the output of
    torch.exp in place and autograd raises in backward as soon as the weights carry a gradient (is_tracker) -- checked here, and the
    library refuses PSL_PTS_GRAD with 'expo' for the same reason.  The mapper's stages (no gradient through the weights) work."""
    try:
        run_render_case("_expo_tracker_must_raise", "replica_expo", "color", True, 500, 16, 112)
    except RuntimeError as e:
        assert "inplace" in str(e) or "in-place" in str(e), e
        print("[expo] reference tracker backward raises as expected:", str(e)[:90])
    else:
        raise AssertionError("the reference's tracker backward was expected to raise with nn_weighting='expo'")
    run_render_case("render_expo_color_mapper", "replica_expo", "color", False, 3000, 96, 111, sparse_frac=0.35,
                    store_param_grads=True)
    run_render_case("render_expo_geometry_mapper", "replica_expo", "geometry", False, 2000, 96, 110)


def main():
    os.makedirs(GOLD, exist_ok=True)
    torch.set_num_threads(8)
    if "--tracker-static" in sys.argv:   # the handle_dynamic=False branch of the tracker loss (results only; inputs = tracker_iter_replica)
        run_tracker_case("tracker_iter_replica_static", 2000, 200, 106, handle_dynamic=False)
        return
    if "--scannet-mapper" in sys.argv:   # encode_exposure with exposure_feat=None: raw colour logits, no sigmoid (decoder.py:432-448)
        run_render_case("render_scannet_color_mapper", "scannet", "color", False, 2000, 64, 109)
        return
    if "--expo" in sys.argv:   # nn_weighting = 'expo': the mapper's two stages (feature / decoder gradients)
        run_expo_cases()
        return
    if "--zero-depth" in sys.argv:   # only the sensor-hole cases (the others are unchanged)
        run_render_case("render_holes_nearpcl_mapper", "replica", "color", False, 3000, 96, 107, sparse_frac=0.35,
                        zero_depth_frac=0.4, sample_near_pcl=True)
        run_render_case("render_holes_uniform_tracker", "replica", "color", True, 2000, 64, 108,
                        zero_depth_frac=0.4, sample_near_pcl=False)
        return
    cfg, P = run_render_case("render_replica_color_tracker", "replica", "color", True, 2000, 96, 101,
                             store_param_grads=True)
    save("decoders_seed1219_replica", {k: v for k, v in P.items()})
    run_render_case("render_replica_color_mapper", "replica", "color", False, 3000, 96, 102, sparse_frac=0.35)
    run_render_case("render_replica_geometry_mapper", "replica", "geometry", False, 2000, 96, 103)
    cfg, P = run_render_case("render_tum_color_mapper", "tum", "color", False, 2000, 96, 104)
    cfg, P = run_render_case("render_scannet_color_tracker", "scannet", "color", True, 2000, 64, 105,
                             exposure=True, store_param_grads=True)
    save("decoders_seed1219_scannet", {k: v for k, v in P.items()})
    run_tracker_case("tracker_iter_replica", 2000, 200, 106)
    run_tracker_case("tracker_iter_replica_static", 2000, 200, 106, handle_dynamic=False)
    run_render_case("render_holes_nearpcl_mapper", "replica", "color", False, 3000, 96, 107, sparse_frac=0.35,
                    zero_depth_frac=0.4, sample_near_pcl=True)
    run_render_case("render_holes_uniform_tracker", "replica", "color", True, 2000, 64, 108,
                    zero_depth_frac=0.4, sample_near_pcl=False)

    run_render_case("render_scannet_color_mapper", "scannet", "color", False, 2000, 64, 109)
    run_expo_cases()


if __name__ == "__main__":
    main()
