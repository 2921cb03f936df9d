"""GPU parity tests: the HIP path (through the C ABI / the drop-in Python mirrors) against
  (a) the golden fixtures = outputs of the UNMODIFIED reference (tests/golden/*.npz), and
  (b) the CPU oracle on fresh seeded inputs.

Tolerances (fp32; SURVEY §7 'fp32 parity through Fourier features'): a 1-ulp change of a sample
position moves a per-ray colour by up to 1e-2 and the summed loss by ~6e-5 in the reference itself,
so per-ray colour is checked at 5e-3 abs / depth at 2e-4 rel, and the LOSS-level quantity at the
1e-4 relative bound that BASELINE.json states.
"""
import json
import os

import pytest
import torch

from tests.helpers import RENDER_CASES, base_cfg, cfg_variant, fixture_cfg, load_decoders, load_npz, relerr, ROOT

pytestmark = pytest.mark.gpu

REPORT = os.path.join(ROOT, "gpurun_out", "parity_report.jsonl")


def report(**kw):
    os.makedirs(os.path.dirname(REPORT), exist_ok=True)
    with open(REPORT, "a") as f:
        f.write(json.dumps(kw) + "\n")
    print("REPORT", kw)


@pytest.fixture(scope="module")
def dev():
    assert torch.cuda.is_available(), "these tests need the MI355X"
    return torch.device("cuda:0")


def make_npc(cfg, cloud, geo, col, dev, max_points=200000):
    from point_slam_amd.neural_point import HipNeuralPointCloud
    cfg = dict(cfg)
    cfg["mapping"] = dict(cfg["mapping"], device="cuda:0")
    npc = HipNeuralPointCloud(cfg, max_points=max_points, device="cuda:0")
    npc.set_points(cloud.to(dev), geo.to(dev), col.to(dev))
    return npc


def make_renderer(cfg, coef=0.1):
    import types
    from point_slam_amd import synthetic as syn
    from point_slam_amd.renderer import HipRenderer
    cam = syn.intrinsics(640, 480)
    r = HipRenderer(cfg, None, types.SimpleNamespace(**cam))
    r.sigmoid_coefficient = coef
    return r


def make_decoders(cfg, cfg_name, dev):
    from point_slam_amd.decoders import PointDecoders
    d = PointDecoders(cfg).load_reference_state(load_decoders(cfg_name))
    return d.to(dev)


# ------------------------------------------------------------------------------ k-NN
@pytest.mark.parametrize("n_pts,nq", [(20000, 4000), (7, 50), (300000, 3000)])
def test_knn_matches_oracle(dev, n_pts, nq):
    from oracle import pointslam_oracle as O
    g = torch.Generator().manual_seed(n_pts)
    # points on a few planes (surface-like) + volume noise; queries near the surfaces
    cloud = torch.rand(n_pts, 3, generator=g) * torch.tensor([4.0, 3.0, 0.05]) + torch.tensor([0.0, 0.0, 1.0])
    cloud[::3, 2] += torch.rand((n_pts + 2) // 3, generator=g) * 2.0
    q = cloud[torch.randint(n_pts, (nq,), generator=g)] + 0.03 * torch.randn(nq, 3, generator=g)
    r = 0.04 + 0.12 * torch.rand(nq, generator=g)
    cfg = base_cfg()
    npc = make_npc(cfg, cloud, torch.zeros(n_pts, 32), torch.zeros(n_pts, 32), dev, max_points=n_pts + 10)
    D, I, cnt = npc.find_neighbors_faiss(q.to(dev), step="query", dynamic_radius=r.to(dev))
    D, I, cnt = D.cpu(), I.cpu(), cnt.cpu()
    Do, Io = O.knn_exact(cloud, q, 8)
    cnt_o = O.neighbor_count(Do, r)
    inr = Do <= (r * r)[:, None]            # slots the HIP kernel is required to fill
    Io_m = torch.where(inr, Io, torch.full_like(Io, -1))
    Do_m = torch.where(inr, Do, torch.full_like(Do, float("inf")))
    bad = (I != Io_m).any(1)
    report(test="knn", n_pts=n_pts, nq=nq, mismatched_queries=int(bad.sum()), cnt_mismatch=int((cnt != cnt_o).sum()),
           maxD=float(torch.nan_to_num((D - Do_m).abs(), nan=0.0, posinf=0.0).max()))
    assert torch.equal(cnt, cnt_o)
    assert torch.equal(I, Io_m)
    assert torch.equal(D, Do_m)             # bit-exact distances: same (dx*dx+dy*dy)+dz*dz


def test_knn_fixed_radius_and_add_step(dev):
    from oracle import pointslam_oracle as O
    g = torch.Generator().manual_seed(3)
    cloud = torch.rand(5000, 3, generator=g)
    q = torch.rand(500, 3, generator=g)
    cfg = cfg_variant("tum")
    npc = make_npc(cfg, cloud, torch.zeros(5000, 32), torch.zeros(5000, 32), dev)
    for step, rad in (("query", cfg["pointcloud"]["radius_query"]), ("add", cfg["pointcloud"]["radius_add"])):
        D, I, cnt = npc.find_neighbors_faiss(q.to(dev), step=step)
        Do, Io = O.knn_exact(cloud, q, 8)
        assert torch.equal(cnt.cpu(), O.neighbor_count(Do, rad))


def test_ray_knn_sparse_cloud_matches_oracle(dev):
    """Ray-mode k-NN on a SPARSE cloud (the expanding search needs several passes; many samples have fewer than 8 or no
    neighbours inside the radius) through the render workspace; fixed query radius."""
    import ctypes as C
    from oracle import pointslam_oracle as O
    from point_slam_amd import _lib, params as P_
    g = torch.Generator().manual_seed(17)
    n_pts, R = 1500, 333
    cloud = torch.rand(n_pts, 3, generator=g) * torch.tensor([2.0, 2.0, 0.4]) + torch.tensor([0.0, 0.0, 1.8])
    cfg = cfg_variant("tum")
    npc = make_npc(cfg, cloud, torch.zeros(n_pts, 32), torch.zeros(n_pts, 32), dev, max_points=n_pts + 8)
    dec = make_decoders(cfg, "tum", dev)
    theta = P_.pack_master(dec).detach().contiguous()
    Bcol = P_.color_embed_B(dec).to(dev).float().contiguous()
    ro = torch.zeros(R, 3) + torch.tensor([1.0, 1.0, 0.0])
    rd = torch.cat([(torch.rand(R, 2, generator=g) - 0.5) * 0.9, torch.ones(R, 1)], 1)
    gd = 1.8 + 0.4 * torch.rand(R, generator=g)
    L = _lib.lib()
    ro_d, rd_d, gd_d = ro.to(dev).contiguous(), rd.to(dev).contiguous(), gd.to(dev).contiguous()
    ws = torch.zeros(int(L.psl_render_ws_floats(R, 0)), device=dev)
    depth = torch.empty(R, device=dev); var = torch.empty(R, device=dev); rgb = torch.empty(R, 3, device=dev)
    valid = torch.empty(R, device=dev, dtype=torch.uint8)
    fb = torch.zeros(2, 32, device=dev)
    feats = torch.zeros(n_pts, 32, device=dev)
    a = _lib.psl_render_args(n_rays=R, flags=0, sigmoid_coef=0.1, rays_o=ro_d.data_ptr(), rays_d=rd_d.data_ptr(),
                             gt_depth=gd_d.data_ptr(), r_query=None, geo_feats=feats.data_ptr(), col_feats=feats.data_ptr(),
                             params=theta.data_ptr(), col_embed_B=Bcol.data_ptr(), fallback_geo=fb[0].data_ptr(),
                             fallback_col=fb[1].data_ptr(), exposure_affine=None, ws=ws.data_ptr(), depth=depth.data_ptr(),
                             var=var.data_ptr(), rgb=rgb.data_ptr(), valid_ray=valid.data_ptr())
    P = 5 * R
    Ppad = (P + 15) // 16 * 16
    q = O.sample_points(ro, rd, O.z_samples(gd, 0.98, 1.02, 5))
    rq = cfg["pointcloud"]["radius_query"]
    Do, Io = O.knn_exact(cloud, q, 8)
    Io_m = torch.where(Do <= rq * rq, Io, torch.full_like(Io, -1))
    cnt_o = O.neighbor_count(Do, rq)
    assert 0.02 < float((cnt_o == 0).float().mean()) and float((cnt_o == 8).float().mean()) < 0.9
    try:
        for ver in (0, 2, 4):    # by launch / one wavefront per ray (side-stream prefetch) / one per sample, flat enumeration
            _lib.check(L.psl_debug_option(b"knn", ver))
            ws.zero_()
            _lib.check(L.psl_render_fwd(npc.handle, C.byref(a), _lib.stream_ptr()))
            torch.cuda.synchronize()
            I = ws[:Ppad * 8].view(torch.int32).reshape(Ppad, 8)[:P].cpu().long()
            cnt = ws[Ppad * 8:Ppad * 9].view(torch.int32)[:P].cpu()
            report(test="ray_knn_sparse", kernel=ver, empty_frac=float((cnt_o == 0).float().mean()),
                   full_frac=float((cnt_o == 8).float().mean()), mismatched=int((I != Io_m).any(1).sum()))
            assert torch.equal(I, Io_m) and torch.equal(cnt, cnt_o), f"k-NN kernel {ver}"
    finally:
        _lib.check(L.psl_debug_option(b"knn", 0))


# ------------------------------------------------------------------------------ compositing
def test_composite_matches_oracle(dev):
    import ctypes as C
    from oracle import pointslam_oracle as O
    from point_slam_amd import _lib
    g = torch.Generator().manual_seed(9)
    R = 1000
    raw = torch.randn(R, 5, 4, generator=g) * 3
    raw[..., 3] *= 20
    raw[::7, :, 3] = -100.0
    z = torch.sort(torch.rand(R, 5, generator=g) * 3 + 0.5, dim=1).values
    d_o, v_o, c_o, w_o = O.composite(raw.clone(), z, 0.1)
    rd, zd = raw.to(dev).contiguous(), z.to(dev).contiguous()
    depth = torch.empty(R, device=dev); var = torch.empty(R, device=dev)
    rgb = torch.empty(R, 3, device=dev); w = torch.empty(R, 5, device=dev)
    _lib.check(_lib.lib().psl_composite_fwd(_lib.ptr(rd), _lib.ptr(zd), R, C.c_float(0.1), _lib.ptr(depth),
                                            _lib.ptr(var), _lib.ptr(rgb), _lib.ptr(w), _lib.stream_ptr()))
    torch.cuda.synchronize()
    assert relerr(depth.cpu(), d_o) < 1e-6 and relerr(rgb.cpu(), c_o) < 1e-6
    assert relerr(w.cpu(), w_o) < 1e-6 and relerr(var.cpu(), v_o) < 1e-5


# ------------------------------------------------------------------------------ render forward
def _run_case(case, dev, grads):
    fx = load_npz(case)
    cfg = fixture_cfg(fx)
    dec = make_decoders(cfg, fx["cfg_name"], dev)
    npc = make_npc(cfg, fx["cloud"], fx["geo"], fx["col"], dev)
    rend = make_renderer(cfg, fx["coef"])
    rend.fixed_fallback = (fx["fb_geo"].to(dev), fx["fb_col"].to(dev))
    ro = fx["rays_o"].to(dev).requires_grad_(grads and fx["is_tracker"])
    rd = fx["rays_d"].to(dev).requires_grad_(grads and fx["is_tracker"])
    geo = fx["geo"].to(dev).requires_grad_(grads)
    col = fx["col"].to(dev).requires_grad_(grads)
    for p in dec.parameters():
        p.requires_grad_(grads)
    ef = None
    if "exposure_feat" in fx:
        ef = fx["exposure_feat"].to(dev).requires_grad_(grads)
    out = rend.render_batch_ray(npc, dec, rd, ro, dev, fx["stage"], gt_depth=fx["gt_depth"].to(dev),
                                npc_geo_feats=geo, npc_col_feats=col, is_tracker=fx["is_tracker"],
                                dynamic_r_query=fx["r_query"].to(dev) if cfg["use_dynamic_radius"] else None,
                                exposure_feat=ef)
    return fx, cfg, dec, (ro, rd, geo, col, ef), out


@pytest.mark.parametrize("case", RENDER_CASES)
def test_render_forward_matches_reference(dev, case, color_structure):
    with torch.no_grad():
        fx, cfg, dec, _, (d, v, c, valid) = _run_case(case, dev, grads=False)
    d, v, c, valid = d.cpu(), v.cpu(), c.cpu(), valid.cpu()
    # loss-level quantities (what BASELINE.json bounds at 1e-4): mapper-style L1 sums against the sensor values
    gd, gc = fx["gt_depth"], fx["gt_color"]
    m = fx["ref_valid"]
    Ld, Ld_ref = (gd - d)[m].abs().sum(), (gd - fx["ref_depth"])[m].abs().sum()
    Lc, Lc_ref = (gc - c)[m].abs().sum(), (gc - fx["ref_rgb"])[m].abs().sum()
    rep = dict(test="render_fwd", case=case, depth_rel=relerr(d, fx["ref_depth"]),
               rgb_abs=float((c - fx["ref_rgb"]).abs().max()), var_rel=relerr(v, fx["ref_var"]),
               valid_eq=bool(torch.equal(valid, fx["ref_valid"])),
               depth_loss_rel=float((Ld - Ld_ref).abs() / Ld_ref.clamp_min(1e-12)),
               color_loss_rel=float((Lc - Lc_ref).abs() / Lc_ref.clamp_min(1e-12)) if fx["stage"] == "color" else 0.0)
    report(**rep)
    assert rep["valid_eq"]
    assert rep["depth_rel"] < 2e-4
    assert rep["rgb_abs"] < 5e-3
    assert rep["var_rel"] < 2e-3
    assert rep["depth_loss_rel"] < 1e-4
    assert rep["color_loss_rel"] < 1e-4


def test_expo_weighting_refuses_the_pose_gradient(dev):
    """pointcloud.nn_weighting = 'expo' with is_tracker=True: the reference raises in backward (decoder.py:157 / 367 zero the output of
    torch.exp in place; checked by oracle/gen_golden.py run_expo_cases) -- the library refuses the call instead of inventing a gradient."""
    from point_slam_amd import _lib
    fx = load_npz("render_expo_color_mapper")
    cfg = fixture_cfg(fx)
    assert cfg["pointcloud"]["nn_weighting"] == "expo"
    dec = make_decoders(cfg, fx["cfg_name"], dev)
    npc = make_npc(cfg, fx["cloud"], fx["geo"], fx["col"], dev)
    rend = make_renderer(cfg, fx["coef"])
    rend.fixed_fallback = (fx["fb_geo"].to(dev), fx["fb_col"].to(dev))
    ro, rd = fx["rays_o"].to(dev).requires_grad_(True), fx["rays_d"].to(dev).requires_grad_(True)
    with pytest.raises(_lib.PslError, match="expo"):
        rend.render_batch_ray(npc, dec, rd, ro, dev, "color", gt_depth=fx["gt_depth"].to(dev), npc_geo_feats=fx["geo"].to(dev),
                              npc_col_feats=fx["col"].to(dev), is_tracker=True, dynamic_r_query=fx["r_query"].to(dev))


# ------------------------------------------------------------------------------ render backward
@pytest.mark.parametrize("case", RENDER_CASES)
def test_render_backward_matches_reference(dev, case, color_structure):
    fx, cfg, dec, (ro, rd, geo, col, ef), (d, v, c, valid) = _run_case(case, dev, grads=True)
    obj = (d * fx["w_d"].to(dev)).sum() + (c * fx["w_c"].to(dev)).sum() + (v * fx["w_v"].to(dev)).sum()
    obj.backward()
    torch.cuda.synchronize()
    rep = dict(test="render_bwd", case=case)
    if fx["is_tracker"]:
        rep["g_rays_o"] = relerr(ro.grad.cpu(), fx["ref_g_rays_o"])
        rep["g_rays_d"] = relerr(rd.grad.cpu(), fx["ref_g_rays_d"])
    rows = fx["ref_g_geo_rows"].long()
    gg = geo.grad.cpu()
    rep["g_geo"] = relerr(gg[rows], fx["ref_g_geo_vals"])
    rep["g_geo_offrows"] = float(gg.abs().sum() - gg[rows].abs().sum())
    if fx["stage"] == "color":
        rows = fx["ref_g_col_rows"].long()
        gc = col.grad.cpu()
        rep["g_col"] = relerr(gc[rows], fx["ref_g_col_vals"])
        rep["g_col_offrows"] = float(gc.abs().sum() - gc[rows].abs().sum())
        pg = {k: p.grad.cpu() for k, p in dec.named_parameters() if p.grad is not None}
        worst, worst_name = 0.0, ""
        for k in fx:
            if k.startswith("refgp_"):
                name = k[len("refgp_"):]
                if name.startswith("geo_decoder"):
                    continue        # geometry decoder is frozen in every config (point_slam.yaml:47)
                e = relerr(pg[name], fx[k])
                if e > worst:
                    worst, worst_name = e, name
            if k.startswith("refgpnorm_"):
                name = k[len("refgpnorm_"):]
                if name.startswith("geo_decoder"):
                    continue
                e = abs(float(pg[name].double().norm()) - float(fx[k])) / float(fx[k])
                if e > worst:
                    worst, worst_name = e, name
        rep["g_params_worst"], rep["g_params_worst_name"] = worst, worst_name
    if ef is not None:
        rep["g_exposure"] = relerr(ef.grad.cpu(), fx["ref_g_exposure_feat"])
    report(**rep)
    for k, val in rep.items():
        if k.startswith("g_") and not k.endswith("_name") and not k.endswith("offrows"):
            assert val < 2e-3, (k, val)
    assert abs(rep["g_geo_offrows"]) < 1e-4


# ------------------------------------------------------------------------------ point growth
def test_add_points_matches_oracle(dev):
    from oracle import pointslam_oracle as O
    from point_slam_amd import synthetic as syn
    cfg = base_cfg()
    cam = syn.intrinsics(320, 240)
    g = torch.Generator().manual_seed(77)
    npc = make_npc(cfg, torch.zeros(0, 3), torch.zeros(0, 32), torch.zeros(0, 32), dev)
    cloud = torch.zeros(0, 3)
    for frame in range(3):
        c2w = syn.pose(5.0 * frame)
        depth, color = syn.render_frame(cam, c2w)
        r_add, _ = syn.dynamic_radii(color, cfg)
        n = 3000
        u = torch.randint(0, cam["W"], (n,), generator=g)
        v = torch.randint(0, cam["H"], (n,), generator=g)
        ro, rd = O.rays_from_uv(u.float(), v.float(), c2w, cam["fx"], cam["fy"], cam["cx"], cam["cy"])
        dep = depth[v, u].clone()
        dep[::17] = 0.0                                   # depth-less pixels are skipped
        rad = r_add[v, u]
        new_pts, keep_o, _ = O.add_points_select(cloud, ro, rd, dep, rad)
        kept, keep_h, n_before = npc.add_neural_points(ro.to(dev), rd.to(dev).contiguous(), dep.to(dev),
                                                       torch.zeros(n, 3, device=dev), dynamic_radius=rad.to(dev),
                                                       return_new=True)
        pos_mask = dep > 0
        assert kept == int(keep_o.sum())
        assert torch.equal(keep_h.cpu()[pos_mask], keep_o)
        assert not keep_h.cpu()[~pos_mask].any()
        cloud = torch.cat([cloud, new_pts])
        got = npc.cloud_pos().cpu()
        assert got.shape == cloud.shape and torch.equal(got, cloud)       # bit-exact positions, same order
        assert npc.get_geo_feats().shape[0] == cloud.shape[0]
    report(test="add_points", total=int(cloud.shape[0]))


# ------------------------------------------------------------------------------ Adam
def test_adam_matches_torch(dev):
    from point_slam_amd import _lib
    g = torch.Generator().manual_seed(2)
    n = 10007
    p = torch.randn(n, generator=g)
    ref = p.clone().requires_grad_(True)
    opt = torch.optim.Adam([ref], lr=0.005)
    pd = p.to(dev); m = torch.zeros(n, device=dev); v = torch.zeros(n, device=dev)
    for step in range(1, 8):
        gr = torch.randn(n, generator=g) * (0.1 if step % 2 else 3.0)
        ref.grad = gr.clone()
        opt.step()
        gd = gr.to(dev)
        _lib.check(_lib.lib().psl_adam_step(_lib.ptr(pd), _lib.ptr(gd), _lib.ptr(m), _lib.ptr(v), n, step, 0.005, 0.9,
                                            0.999, 1e-8, 1, _lib.stream_ptr()))
        assert float(gd.abs().max()) == 0.0
    err = float((pd.cpu() - ref.detach()).abs().max())
    report(test="adam", max_abs=err)
    assert err < 1e-6
    # row-indexed variant
    feats = torch.randn(500, 32, generator=g)
    rows = torch.randperm(500, generator=g)[:200].int()
    ref2 = feats[rows.long()].clone().requires_grad_(True)
    opt2 = torch.optim.Adam([ref2], lr=0.03)
    fd = feats.to(dev); rd_ = rows.to(dev)
    m2 = torch.zeros(200, 32, device=dev); v2 = torch.zeros(200, 32, device=dev)
    for step in range(1, 4):
        gr = torch.randn(200, 32, generator=g)
        ref2.grad = gr.clone(); opt2.step()
        gd = gr.to(dev).contiguous()
        _lib.check(_lib.lib().psl_adam_step_rows(_lib.ptr(fd), _lib.ptr(rd_), _lib.ptr(gd), _lib.ptr(m2), _lib.ptr(v2),
                                                 200, step, 0.03, 0.9, 0.999, 1e-8, 0, _lib.stream_ptr()))
    out = fd.cpu()
    assert float((out[rows.long()] - ref2.detach()).abs().max()) < 1e-6
    untouched = torch.ones(500, dtype=torch.bool); untouched[rows.long()] = False
    assert torch.equal(out[untouched], feats[untouched])


# ------------------------------------------------------------------------------ drop-in tracker iteration
def test_tracker_iteration_through_hip_renderer(dev, color_structure):
    """The reference's Tracker.optimize_cam_in_batch step (golden: loss + pose after one Adam step), re-run with
    HipRenderer substituted for Renderer and everything else (sampling, loss, torch Adam) as in the reference."""
    import types
    from point_slam_amd import host_ops as H, synthetic as syn
    from point_slam_amd.renderer import HipRenderer
    fx = load_npz("tracker_iter_replica")
    cfg = base_cfg()
    cam = syn.intrinsics(160, 120)
    dec = make_decoders(cfg, "replica", dev)
    for p in dec.parameters():
        p.requires_grad_(False)
    npc = make_npc(cfg, fx["cloud"], fx["geo"], fx["col"], dev)
    rend = HipRenderer(cfg, None, types.SimpleNamespace(**cam))
    rend.sigmoid_coefficient = cfg["rendering"]["sigmoid_coef_tracker"]
    rend.fixed_fallback = (fx["fb_geo"].to(dev), fx["fb_col"].to(dev))
    quad = fx["cam0"][:4].to(dev).requires_grad_(True)
    T = fx["cam0"][4:].to(dev).requires_grad_(True)
    lr = cfg["tracking"]["lr"]
    opt = torch.optim.Adam([{"params": [T], "lr": lr}, {"params": [quad], "lr": lr * 0.2}])
    c2w = H.get_camera_from_tensor(torch.cat([quad, T]))
    u, v = H.pixels_from_flat_index(fx["pix_idx"].long().to(dev), 20, cam["H"] - 20, 20, cam["W"] - 20)
    ro, rd = H.get_rays_from_uv(u, v, c2w, cam["fx"], cam["fy"], cam["cx"], cam["cy"])
    depth_img, color_img, rq_img = fx["depth_img"].to(dev), fx["color_img"].to(dev).double(), fx["rq_img"].to(dev)
    gd, gc, rq = depth_img[v.long(), u.long()], color_img[v.long(), u.long()], rq_img[v.long(), u.long()]
    keep = gd > 0
    ro, rd, gd, gc, rq = ro[keep], rd[keep], gd[keep], gc[keep], rq[keep]
    inl = H.depth_inlier_mask(gd)
    ro, rd, gd, gc, rq = ro[inl], rd[inl], gd[inl], gc[inl], rq[inl]
    d, var, rgb, _ = rend.render_batch_ray(npc, dec, rd, ro, dev, "color", gt_depth=gd, npc_geo_feats=fx["geo"].to(dev),
                                           npc_col_feats=fx["col"].to(dev), is_tracker=True, dynamic_r_query=rq)
    loss, geo, colr, mask = H.tracker_loss(d, var, rgb, gd, gc)
    loss.backward()
    opt.step()
    rel = abs(float(loss) - fx["ref_loss"]) / fx["ref_loss"]
    dq = float((quad.detach().cpu() - fx["ref_quad_after"]).abs().max())
    dT = float((T.detach().cpu() - fx["ref_T_after"]).abs().max())
    report(test="tracker_iter", loss_rel=rel, dq=dq, dT=dT)
    assert rel < 1e-4                     # BASELINE.json: render-loss rel-err <= 1e-4
    # one Adam step moves each parameter by ~lr*sign(grad): equality means every gradient SIGN matches
    assert dq < 1e-6 and dT < 1e-6


# ------------------------------------------------------------------------------ pixels without sensor depth
def _hole_scene(seed, n_pts=9000, W=64, H=48):
    from oracle import pointslam_oracle as O
    from point_slam_amd import synthetic as syn
    cam = syn.intrinsics(W, H)
    g = torch.Generator().manual_seed(seed)
    c2w = syn.pose(3.0)
    depth, color = syn.render_frame(cam, c2w)
    cfg = base_cfg()
    _, rq = syn.dynamic_radii(color, cfg)
    pts = []
    t = torch.linspace(0.0, 1.0, 3)
    for v in range(3):
        cw = syn.pose(3.0 + 1.0 * (v - 1))
        u = torch.rand(n_pts // 9 + 1, generator=g) * (W - 1)
        w = torch.rand(n_pts // 9 + 1, generator=g) * (H - 1)
        ro, rd = O.rays_from_uv(u, w, cw, cam["fx"], cam["fy"], cam["cx"], cam["cy"])
        d = syn.box_depth(ro, rd)
        z = 0.98 * d[:, None] * (1 - t) + 1.02 * d[:, None] * t
        pts.append((ro[:, None] + rd[:, None] * z[..., None]).reshape(-1, 3))
    cloud = torch.cat(pts)[:n_pts].float().contiguous()
    cloud = cloud[cloud[:, 0] > cloud[:, 0].quantile(0.3)].contiguous()      # part of the view sees no cloud
    N = cloud.shape[0]
    geo = torch.zeros(N, 32).normal_(0, 0.1, generator=g)
    col = torch.zeros(N, 32).normal_(0, 0.1, generator=g)
    hole = torch.rand(H, W, generator=g) < 0.3
    depth = torch.where(hole, torch.zeros_like(depth), depth)
    return cfg, cam, c2w, depth, color, rq, cloud, geo, col


def test_sample_near_pcl_matches_oracle(dev):
    from oracle import pointslam_oracle as O
    cfg, cam, c2w, depth, color, rq, cloud, geo, col = _hole_scene(31)
    npc = make_npc(cfg, cloud, geo, col, dev)
    g = torch.Generator().manual_seed(5)
    n = 700
    u = torch.rand(n, generator=g) * (cam["W"] - 1)
    v = torch.rand(n, generator=g) * (cam["H"] - 1)
    ro, rd = O.rays_from_uv(u, v, c2w, cam["fx"], cam["fy"], cam["cx"], cam["cy"])
    near, far_a, far_b = cfg["rendering"]["near_end"], torch.tensor(4.3), torch.tensor(3.1)
    z_o, inv_o = O.sample_near_pcl(cloud, ro, rd, near, float(far_a), 5, cfg["pointcloud"]["radius_query"])
    z_h, inv_h = npc.sample_near_pcl(ro.to(dev), rd.to(dev), near, far_a.to(dev), 5)
    assert 0.05 < float(inv_o.float().mean()) < 0.95
    assert torch.equal(inv_h.cpu(), inv_o)
    assert torch.equal(z_h.cpu(), z_o)
    # per-ray far bound (render_img: one per reference batch): two groups
    far_vec = torch.where(torch.arange(n) < n // 2, far_a, far_b)
    z_h2, inv_h2 = npc.sample_near_pcl(ro.to(dev), rd.to(dev), near, far_vec.to(dev), 5)
    h = n // 2
    z_o2, inv_o2 = O.sample_near_pcl(cloud, ro[h:], rd[h:], near, float(far_b), 5, cfg["pointcloud"]["radius_query"])
    assert torch.equal(z_h2[:h].cpu(), z_o[:h]) and torch.equal(inv_h2[:h].cpu(), inv_o[:h])
    assert torch.equal(z_h2[h:].cpu(), z_o2) and torch.equal(inv_h2[h:].cpu(), inv_o2)
    report(test="sample_near_pcl", n=n, invalid_frac=float(inv_o.float().mean()))


@pytest.mark.parametrize("near_pcl", [True, False])
def test_render_img_with_holes_matches_oracle(dev, near_pcl):
    """render_img over an image with sensor holes, against the oracle run batch-by-batch like the reference
    (Renderer.py:244-268: each 3000-ray batch has its own far bound; here batches of 500)."""
    import types
    from oracle import pointslam_oracle as O
    from point_slam_amd.renderer import HipRenderer
    cfg, cam, c2w, depth, color, rq, cloud, geo, col = _hole_scene(32)
    cfg["rendering"]["sample_near_pcl"] = near_pcl
    B = 500
    dec = make_decoders(cfg, "replica", dev)
    P = load_decoders("replica")
    npc = make_npc(cfg, cloud, geo, col, dev)
    rend = HipRenderer(cfg, None, types.SimpleNamespace(**cam), ray_batch_size=B)
    g = torch.Generator().manual_seed(8)
    fb_geo = torch.zeros(32).normal_(0, 0.01, generator=g)
    fb_col = torch.zeros(32).normal_(0, 0.01, generator=g)
    rend.fixed_fallback = (fb_geo.to(dev), fb_col.to(dev))
    d_h, u_h, c_h = rend.render_img(npc, dec, c2w.to(dev), dev, "color", gt_depth=depth.to(dev),
                                    npc_geo_feats=geo.to(dev), npc_col_feats=col.to(dev),
                                    dynamic_r_query=rq.to(dev))
    H, W = cam["H"], cam["W"]
    vv, uu = torch.meshgrid(torch.arange(H).float(), torch.arange(W).float(), indexing="ij")
    ro, rd = O.rays_from_uv(uu.reshape(-1), vv.reshape(-1), c2w, cam["fx"], cam["fy"], cam["cx"], cam["cy"])
    gd, rqf = depth.reshape(-1), rq.reshape(-1)
    d_o, c_o = [], []
    with torch.no_grad():
        for i in range(0, H * W, B):
            d, _, c, _, _ = O.render_batch_ray(cfg, P, cloud, geo, col, ro[i:i + B], rd[i:i + B], gd[i:i + B], "color",
                                               rqf[i:i + B], fb_geo, fb_col, coef=rend.sigmoid_coefficient)
            d_o.append(d); c_o.append(c)
    d_o, c_o = torch.cat(d_o), torch.cat(c_o)
    d_h, c_h = d_h.reshape(-1).float().cpu(), c_h.reshape(-1, 3).cpu()
    hole = gd <= 0
    rep = dict(test="render_img_holes", near_pcl=near_pcl, hole_frac=float(hole.float().mean()),
               depth_rel=relerr(d_h, d_o), depth_rel_holes=relerr(d_h[hole], d_o[hole]) if near_pcl else 0.0,
               rgb_abs=float((c_h - c_o).abs().max()), rgb_abs_holes=float((c_h[hole] - c_o[hole]).abs().max()))
    report(**rep)
    assert rep["depth_rel"] < 2e-4 and rep["rgb_abs"] < 5e-3
    if not near_pcl:
        assert float(d_h[hole].abs().max()) == 0.0
