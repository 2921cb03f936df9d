"""GPU parity of the fused native loops against the REFERENCE'S OWN loops (fixtures of oracle/gen_golden_loops.py:
the real Mapper.optimize_map and the real Tracker.optimize_cam_in_batch loop, every random draw recorded):

  * psl_map_iters    vs mapper_iters_replica / mapper_iters_scannet (per-frame exposure latents)
  * psl_track_iters  vs tracker_iters_tum / tracker_iters_scannet (sample_with_color_grad pixel sets, exposure groups)
  * add_neural_points vs add_points_ref (uniform + pixel-gradient batches over 3 frames, _input_pos / _input_rgb)

Tolerances: the first iteration sees identical state, so its loss is held to 1e-5 relative; later iterations inherit
Adam's sign sensitivity (an entry whose gradient is at rounding-noise level moves by +-lr in either direction, in the
reference too), so losses are held to BASELINE's 1e-4 .. a few 1e-4 and final states are compared as distributions.
"""
import pytest
import torch

from tests.helpers import load_decoders, load_npz, loop_cam, loop_cfg, mapper_frames
from tests.test_hip_parity import report

pytestmark = pytest.mark.gpu


def _slam(cfg, cam, fx, dev, cfg_name):
    from point_slam_amd.decoders import PointDecoders
    from point_slam_amd.slam import HipSLAM
    dec = PointDecoders(cfg).load_reference_state(load_decoders(cfg_name))
    s = HipSLAM(cfg, cam, device="cuda:0", max_points=100000, engine="native", decoders=dec)
    s.npc.set_points(fx["cloud"].to(dev), fx["geo"].to(dev), fx["col"].to(dev))
    return s


@pytest.mark.parametrize("case", ["mapper_iters_replica", "mapper_iters_scannet"])
def test_map_iters_native_matches_reference_loop(case, color_structure):
    from point_slam_amd import params as P_
    from point_slam_amd.slam import Frame
    dev = torch.device("cuda:0")
    fx = load_npz(case)
    cfg, cam = loop_cfg(fx), loop_cam(fx)
    exposure = cfg["model"]["encode_exposure"]
    s = _slam(cfg, cam, fx, dev, fx["cfg_name"])
    window = []
    for k, f in enumerate(mapper_frames(fx, exposure)):
        fr = Frame(k, f["depth"].to(dev), f["color"].to(dev), r_query=f["r_query"].to(dev) if cfg["use_dynamic_radius"] else None,
                   c2w=f["c2w"].to(dev), exposure=f["exposure"].to(dev) if exposure else None)
        window.append(fr)
    if exposure:
        s.exposure_feat = window[-1].exposure.clone()
    sel = fx["sel"].to(dev).int().contiguous()
    # the rows are the reference's own selection (Mapper.get_mask_from_c2w inside its optimize_map, cv2 rule): the kernel
    # finds the same ones on the same map (a point within float32 rounding of a test may fall on either side: <= 2)
    got, _ = s.frustum_select(window[-1], window[-1].c2w)
    assert len(set(got.cpu().tolist()) ^ set(fx["sel"].tolist())) <= 2
    N = s.npc.pts_num()
    row_map = torch.full((N,), -1, dtype=torch.int32, device=dev)
    row_map[sel.long()] = torch.arange(sel.shape[0], dtype=torch.int32, device=dev)
    n_iters, ppf = fx["n_iters"], fx["mapping_pixels"] // 3
    draws = (fx["pix"].to(dev).int().reshape(n_iters, 3 * ppf).contiguous(), fx["fb"].to(dev).contiguous())
    s._map_native(window, sel, row_map, n_iters, ppf, draws=draws, n_geo=fx["n_geo_iters"])
    torch.cuda.synchronize()
    ls = s.last_losses.cpu().double()
    ref = fx["ref_losses"]
    rel = ((ls[:, 0] - ref).abs() / ref.abs())
    sl = fx["sel"].long()
    geo, col = s.npc.geo_feats.cpu()[sl], s.npc.col_feats.cpu()[sl]
    dg, dc = (geo - fx["ref_geo_final"]).abs(), (col - fx["ref_col_final"]).abs()
    theta = P_.unpack_master(s.theta.cpu())
    worst, worst_name, frac_worst = 0.0, "", 0.0
    for k in fx:
        if k.startswith("refdec_") and "mlp_exposure" not in k:
            name = k[len("refdec_"):]
            if name not in theta:
                continue
            d = (theta[name] - fx[k]).abs()
            if float(d.max()) > worst:
                worst, worst_name = float(d.max()), name
            frac_worst = max(frac_worst, float((d > 1e-4).float().mean()))
    rep = dict(test="map_native_vs_reference_loop", case=case, loss_rel_first=float(rel[0]), loss_rel_max=float(rel.max()),
               losses=[float(x) for x in ls[:, 0]], ref_losses=[float(x) for x in ref],
               geo_max=float(dg.max()), geo_frac_gt_1e4=float((dg > 1e-4).float().mean()),
               col_max=float(dc.max()), col_frac_gt_1e4=float((dc > 1e-4).float().mean()),
               dec_max=worst, dec_worst=worst_name, dec_frac_gt_1e4=frac_worst)
    if exposure:
        rep["exposure_abs"] = float((s.exposure_feat.cpu() - fx["ref_exposure_final"]).abs().max())
        m = s.exposure_mlp.cpu()
        rep["exposure_mlp_abs"] = max(
            float((m[:1024].reshape(128, 8) - fx["refdec_color_decoder.mlp_exposure.linear1.weight"]).abs().max()),
            float((m[1152:2688].reshape(12, 128) - fx["refdec_color_decoder.mlp_exposure.linear2.weight"]).abs().max()),
            float((m[2688:] - fx["refdec_color_decoder.mlp_exposure.linear2.bias"]).abs().max()))
    report(**rep)
    assert rep["loss_rel_first"] < 1e-5
    assert rep["loss_rel_max"] < 5e-4
    # untouched rows: bit-identical
    mask = torch.ones(N, dtype=torch.bool); mask[sl] = False
    assert torch.equal(s.npc.geo_feats.cpu()[mask], fx["geo"][mask])
    assert rep["geo_max"] < 2e-2 and rep["geo_frac_gt_1e4"] < 1e-2
    assert rep["col_max"] < 2e-2 and rep["col_frac_gt_1e4"] < 1e-2
    assert rep["dec_max"] < 1e-2 and rep["dec_frac_gt_1e4"] < 2e-2
    if exposure:
        assert rep["exposure_abs"] < 1e-4 and rep["exposure_mlp_abs"] < 2e-3


def test_colour_refinement_native_matches_reference(color_structure):
    """HipSLAM's end-of-run refinement step (psl_map_iters with sel = all rows, geometry lr 0, colour lr / 10, decoder
    frozen, iteration 0 in stage 'geometry') vs the reference's optimize_map(color_refine=True) (fixture
    mapper_refine_replica, Mapper.py:706-720,427-430)."""
    from point_slam_amd.slam import Frame
    dev = torch.device("cuda:0")
    fx = load_npz("mapper_refine_replica")
    cfg, cam = loop_cfg(fx), loop_cam(fx)
    s = _slam(cfg, cam, fx, dev, fx["cfg_name"])
    window = [Frame(k, f["depth"].to(dev), f["color"].to(dev), r_query=f["r_query"].to(dev), c2w=f["c2w"].to(dev))
              for k, f in enumerate(mapper_frames(fx, False))]
    N = s.npc.pts_num()
    sel = torch.arange(N, dtype=torch.int32, device=dev)
    n_iters, ppf = fx["n_iters"], fx["mapping_pixels"] // 3
    draws = (fx["pix"].to(dev).int().reshape(n_iters, 3 * ppf).contiguous(), fx["fb"].to(dev).contiguous())
    st = cfg["mapping"]["stage"]["color"]
    theta0 = s.theta.clone()
    s._map_native(window, sel, sel, n_iters, ppf, draws=draws, n_geo=0, train_decoder=False,
                  lr=dict(geo_geo=0.0, geo_col=0.0, col=st["color_lr"] / 10.0, dec=st["decoders_lr"]))
    torch.cuda.synchronize()
    ls = s.last_losses.cpu().double()[:, 0]
    rel = (ls - fx["ref_losses"]).abs() / fx["ref_losses"].abs()
    dc = (s.npc.col_feats.cpu() - fx["ref_col_final"]).abs()
    rep = dict(test="colour_refinement_vs_reference", loss_rel_first=float(rel[0]), loss_rel_max=float(rel.max()),
               col_max=float(dc.max()), col_frac_gt_1e5=float((dc > 1e-5).float().mean()),
               col_moved=float((fx["ref_col_final"] - fx["col"]).abs().max()))
    report(**rep)
    assert rep["loss_rel_first"] < 1e-5 and rep["loss_rel_max"] < 1e-4
    assert torch.equal(s.npc.geo_feats.cpu(), fx["geo"])           # geometry lr 0: bit-identical rows
    assert torch.equal(s.theta, theta0)                            # frozen decoder
    assert rep["col_max"] < 1e-3 and rep["col_frac_gt_1e5"] < 5e-3


def test_refine_runs_five_passes_over_all_rows():
    """HipSLAM.refine: 5 x (2 x iters) iterations, 'global' window of twice the size, all N rows trainable, nothing added."""
    from point_slam_amd.slam import Frame, HipSLAM
    dev = torch.device("cuda:0")
    cfg, cam, frames, pts = _long_scene(dev, n_pts=30000)
    cfg["mapping"].update(iters=6, pixels=300, mapping_window_size=2)
    s = HipSLAM(cfg, cam, device="cuda:0", max_points=100000, engine="native")
    s.seed_points(pts, seed=3)
    s.keyframes = [frames[0], frames[1]]
    geo0, col0, theta0, n0 = s.npc.geo_feats.clone(), s.npc.col_feats.clone(), s.theta.clone(), s.npc.pts_num()
    total = s.refine(frames[2], frames[2].c2w)
    torch.cuda.synchronize()
    assert total == 5 * 12 and s.npc.pts_num() == n0
    assert torch.equal(s.npc.geo_feats, geo0) and torch.equal(s.theta, theta0)
    moved = (s.npc.col_feats != col0).any(1)
    assert int(moved.sum()) > 1000                                 # colour rows all over the views moved ...
    assert float((s.npc.col_feats - col0).abs().max()) < 5 * 12 * cfg["mapping"]["stage"]["color"]["color_lr"] / 10 * 1.01
    assert bool(torch.isfinite(s.last_losses).all())


@pytest.mark.parametrize("case", ["tracker_iters_tum", "tracker_iters_scannet", "tracker_iters_replica_1500"])
def test_track_iters_native_matches_reference_loop(case, color_structure):
    from point_slam_amd.slam import Frame
    dev = torch.device("cuda:0")
    fx = load_npz(case)
    cfg, cam = loop_cfg(fx), loop_cam(fx)
    cfg["tracking"].update(ignore_edge_H=fx["edge"], ignore_edge_W=fx["edge"], sample_with_color_grad=True)
    exposure = cfg["model"]["encode_exposure"]
    s = _slam(cfg, cam, fx, dev, fx["cfg_name"])
    frame = Frame(0, fx["depth_img"].to(dev), fx["color_img"].to(dev),
                  r_query=fx["rq_img"].to(dev) if cfg["use_dynamic_radius"] else None)
    if exposure:
        s.exposure_feat = fx["exposure0"].to(dev).clone()
    n_iters, n_pix = fx["n_iters"], fx["n_pix"]
    draws = (fx["pix_full"].to(dev).int().contiguous(), fx["fb"].to(dev).contiguous())
    best = s._track_native(frame, fx["cam0"], n_iters, n_pix, draws=draws)
    torch.cuda.synchronize()
    ls = s.last_losses.cpu().double()[:, 0]
    ref = fx["ref_losses"]
    rel = (ls - ref).abs() / ref.abs()
    cam_end = s.last_cam.cpu()
    rep = dict(test="track_native_vs_reference_loop", case=case, loss_rel_first=float(rel[0]), loss_rel_max=float(rel.max()),
               losses=[float(x) for x in ls], cam_abs=float((cam_end - fx["ref_cams"][-1]).abs().max()),
               best_abs=float((best.cpu() - fx["ref_best"]).abs().max()), lr=cfg["tracking"]["lr"])
    if exposure:
        rep["exposure_abs"] = float((s.exposure_feat.cpu() - fx["ref_exposure_final"]).abs().max())
        m = s.exposure_mlp.cpu()
        rep["exposure_mlp_abs"] = max(float((m[:1024].reshape(128, 8) - fx["refexp_linear1.weight"]).abs().max()),
                                      float((m[1152:2688].reshape(12, 128) - fx["refexp_linear2.weight"]).abs().max()),
                                      float((m[2688:] - fx["refexp_linear2.bias"]).abs().max()))
    report(**rep)
    assert rep["loss_rel_first"] < 1e-5
    assert rep["loss_rel_max"] < 5e-4
    assert rep["cam_abs"] < 0.5 * cfg["tracking"]["lr"] and rep["best_abs"] < 0.5 * cfg["tracking"]["lr"]
    if exposure:
        # six Adam steps of lr 0.001: weights whose gradient is at rounding-noise level take steps of noise-decided sign
        assert rep["exposure_abs"] < 5e-4 and rep["exposure_mlp_abs"] < 3e-3


def test_add_neural_points_matches_reference():
    """HipNeuralPointCloud.add_neural_points over the batches the reference's add_neural_points saw (3 frames, uniform
    and pixel-gradient batches): same kept counts, bit-identical positions in the same order, same surface points and
    colours in _input_pos / _input_rgb; then sample_near_pcl on the resulting cloud."""
    from point_slam_amd.neural_point import HipNeuralPointCloud
    from tests.helpers import base_cfg
    dev = torch.device("cuda:0")
    fx = load_npz("add_points_ref")
    cfg = base_cfg()
    cfg["mapping"] = dict(cfg["mapping"], device="cuda:0")
    npc = HipNeuralPointCloud(cfg, max_points=20000, device="cuda:0")
    for f in range(3):
        for kind in ("uni", "grad"):
            t = f"f{f}_{kind}"
            assert npc.pts_num() == fx[t + "_n_before"]
            kept = npc.add_neural_points(fx[t + "_rays_o"].to(dev), fx[t + "_rays_d"].to(dev), fx[t + "_depth"].to(dev),
                                         fx[t + "_color"].to(dev), is_pts_grad=(kind == "grad"),
                                         dynamic_radius=fx[t + "_radius"].to(dev))
            assert int(kept) == fx[t + "_kept"], t
            assert npc.pts_num() == fx[t + "_n_after"]
    got = npc.cloud_pos()
    assert not got.is_cuda and torch.equal(got, fx["ref_cloud"])
    import numpy as np
    assert np.array(npc.cloud_pos()).shape == (fx["ref_cloud"].shape[0], 3)          # Mapper.py:131,760
    assert torch.equal(torch.tensor(npc.input_pos()), fx["ref_input_pos"])
    assert torch.equal(torch.tensor(npc.input_rgb()), fx["ref_input_rgb"])
    assert npc.get_geo_feats().shape[0] == fx["ref_cloud"].shape[0]
    z, inv = npc.sample_near_pcl(fx["snp_rays_o"].to(dev), fx["snp_rays_d"].to(dev), fx["snp_near"],
                                 torch.tensor(fx["snp_far"], device=dev), 5)
    assert torch.equal(inv.cpu(), fx["ref_snp_invalid"]) and torch.equal(z.cpu(), fx["ref_snp_z"])
    report(test="add_points_vs_reference", total=int(got.shape[0]))


def test_mapper_add_step_matches_reference():
    """The point-adding calls INSIDE the reference's optimize_map run (fixture mapper_iters_replica): uniform batch
    with the per-pixel r_add map, then the pixel-gradient batch with radius_min semantics."""
    from point_slam_amd.neural_point import HipNeuralPointCloud
    dev = torch.device("cuda:0")
    fx = load_npz("mapper_iters_replica")
    cfg = loop_cfg(fx)
    cfg["mapping"] = dict(cfg["mapping"], device="cuda:0")
    npc = HipNeuralPointCloud(cfg, max_points=20000, device="cuda:0")
    n0 = fx["n0"]
    npc.set_points(fx["cloud"][:n0].to(dev), fx["geo"][:n0].to(dev), fx["col"][:n0].to(dev))
    for j in range(fx["n_adds"]):
        assert npc.pts_num() == fx[f"add{j}_n_before"]
        rad = fx[f"add{j}_radius"].to(dev) if f"add{j}_radius" in fx else None
        kept = npc.add_neural_points(fx[f"add{j}_rays_o"].to(dev), fx[f"add{j}_rays_d"].to(dev), fx[f"add{j}_depth"].to(dev),
                                     torch.zeros(fx[f"add{j}_depth"].shape[0], 3, device=dev),
                                     is_pts_grad=fx[f"add{j}_is_pts_grad"], dynamic_radius=rad)
        assert int(kept) == int(fx["add_kept"][j])
        assert npc.pts_num() == fx[f"add{j}_n_after"]
    assert torch.equal(npc.cloud_pos(), fx["cloud"])


def test_checkpoint_after_native_mapping_roundtrip(tmp_path):
    """The decoders a checkpoint holds are the ones psl_map_iters trained (not the initial nn.Module), and a reloaded
    HipSLAM renders the same image."""
    import types
    from point_slam_amd.slam import Frame, HipSLAM
    from point_slam_amd.decoders import PointDecoders
    dev = torch.device("cuda:0")
    fx = load_npz("mapper_iters_replica")
    cfg, cam = loop_cfg(fx), loop_cam(fx)
    s = _slam(cfg, cam, fx, dev, "replica")
    frs = mapper_frames(fx, False)
    window = [Frame(k, f["depth"].to(dev), f["color"].to(dev), r_query=f["r_query"].to(dev), c2w=f["c2w"].to(dev))
              for k, f in enumerate(frs)]
    sel = fx["sel"].to(dev).int().contiguous()
    row_map = torch.full((s.npc.pts_num(),), -1, dtype=torch.int32, device=dev)
    row_map[sel.long()] = torch.arange(sel.shape[0], dtype=torch.int32, device=dev)
    n_iters, ppf = fx["n_iters"], fx["mapping_pixels"] // 3
    draws = (fx["pix"].to(dev).int().reshape(n_iters, 3 * ppf).contiguous(), fx["fb"].to(dev).contiguous())
    theta0 = s.theta.clone()
    s._map_native(window, sel, row_map, n_iters, ppf, draws=draws, n_geo=fx["n_geo_iters"])
    torch.cuda.synchronize()
    assert float((s.theta - theta0).abs().max()) > 1e-4               # the colour decoder was trained
    path = str(tmp_path / "00020.tar")
    s.save_checkpoint(path, idx=20)
    ck = torch.load(path, map_location="cpu", weights_only=False)
    w = ck["decoder_state_dict"]["color_decoder.pts_linears.1.weight"]
    from point_slam_amd import params as P_
    assert torch.equal(w, P_.unpack_master(s.theta.cpu())["color_decoder.pts_linears.1.weight"])
    assert ck["geo_feats"].shape[0] == s.npc.pts_num()                # N rows, not the store's capacity
    dec2 = PointDecoders(cfg).load_reference_state(load_decoders("replica"))
    s2 = HipSLAM(cfg, cam, device="cuda:0", max_points=100000, engine="native", decoders=dec2)
    s2.load_checkpoint(path)
    assert torch.equal(s2.theta.cpu(), s.theta.cpu())
    fr = window[-1]
    fb = (torch.zeros(32, device=dev), torch.zeros(32, device=dev))
    outs = []
    for slam in (s, s2):
        slam.sync_decoders_from_theta()
        slam.renderer.fixed_fallback = fb
        d, u, c = slam.renderer.render_img(slam.npc, slam.decoders, fr.c2w, dev, "color", gt_depth=fr.depth,
                                           npc_geo_feats=slam.npc.get_geo_feats(), npc_col_feats=slam.npc.get_col_feats(),
                                           dynamic_r_query=fr.r_query)
        outs.append((d.cpu(), c.cpu()))
    assert torch.equal(outs[0][0], outs[1][0]) and torch.equal(outs[0][1], outs[1][1])


# ---------------------------------------------------------------------------------------------------------------------
# Long loops against the pinned oracle ("next" 1c/1d): >= 130 iterations, >= 1 000 pixels so that the
# k-NN prefetch block (64 iterations) is crossed twice, n_sel >= 2e4 -- what validates the lazy-Adam replay (v_rcp /
# v_sqrt) and the dense catch-up at a block end against something other than the library itself.
def _long_scene(dev, n_pts=90000, W=320, H=240):
    from point_slam_amd import synthetic as syn
    from point_slam_amd.slam import Frame
    from tests.helpers import base_cfg
    cfg = base_cfg()
    cam = syn.intrinsics(W, H)
    frames = []
    for t in (10.0, 12.0, 14.0):
        c2w = syn.pose(t, dev)
        depth, color = syn.render_frame(cam, c2w)
        r_add, r_q = syn.dynamic_radii(color, cfg)
        frames.append(Frame(int(t), depth, color, r_add, r_q, c2w))
    pts = []
    g = torch.Generator().manual_seed(4)
    tt = torch.linspace(0.0, 1.0, 3)
    for t in (9.0, 11.0, 13.0, 15.0):
        c2w = syn.pose(t)
        u = torch.rand(n_pts // 12, generator=g) * (cam["W"] - 1)
        v = torch.rand(n_pts // 12, generator=g) * (cam["H"] - 1)
        dirs = torch.stack([(u - cam["cx"]) / cam["fx"], -(v - cam["cy"]) / cam["fy"], -torch.ones_like(u)], -1)
        rd = (dirs[:, None, :] * c2w[:3, :3]).sum(-1)
        ro = c2w[:3, 3].expand_as(rd)
        d = syn.box_depth(ro, rd)
        z = 0.98 * d[:, None] * (1 - tt) + 1.02 * d[:, None] * tt
        pts.append((ro[:, None] + rd[:, None] * z[..., None]).reshape(-1, 3))
    return cfg, cam, frames, torch.cat(pts).float()


def _oracle_frames(frames):
    return [dict(depth=f.depth.cpu(), color=f.color.cpu(), c2w=f.c2w.cpu(), r_query=f.r_query.cpu()) for f in frames]


def _native_long_run(cfg, cam, frames, pts, dev, n_iters, ppf, n_geo, draws, lazy, semantics="torch2", n_mapped=0):
    from point_slam_amd import _lib
    from point_slam_amd.decoders import PointDecoders
    from point_slam_amd.slam import HipSLAM
    dec = PointDecoders(cfg).load_reference_state(load_decoders("replica"))
    s = HipSLAM(cfg, cam, device="cuda:0", max_points=200000, engine="native", decoders=dec)
    s.seed_points(pts, seed=77)
    s.adam_zero_grad_semantics = semantics
    s.n_mapped = n_mapped
    sel, row_map = s.frustum_select(frames[-1], frames[-1].c2w)
    L = _lib.lib()
    _lib.check(L.psl_debug_option(b"lazy_adam", 1 if lazy else 0))
    try:
        s._map_native(frames, sel, row_map, n_iters, ppf, draws=draws, n_geo=n_geo)
        torch.cuda.synchronize()
    finally:
        _lib.check(L.psl_debug_option(b"lazy_adam", 1))
    return s, sel


def _oracle_inputs(s):
    from tests import parity_probe as PP
    return PP.oracle_state(s)


@pytest.mark.parametrize("semantics", ["torch2", "torch1", "torch2-frozen-decoder"])
def test_map_iters_140_iterations_vs_oracle(semantics):
    """140 iterations x 1 200 pixels over a 3-frame window, n_sel ~ 5e4: psl_map_iters (lazy Adam, block prefetch) and
    the same call with the dense IEEE Adam sweep, both against O.mapper_iterations on identical draws.
    semantics='torch1': the colour-decoder group counts the geometry-stage steps (zero-tensor gradients left by torch
    1.12's zero_grad, env.yaml:61) -- psl_map_args.step0_params / HipSLAM.adam_zero_grad_semantics.
    'torch2-frozen-decoder' (mapping.fix_color_decoder=True): only the feature rows move.  Without the decoder group --
    whose entries with noise-level gradients take +-lr steps of noise-decided sign, in the reference too -- nothing
    amplifies rounding differences, and all 140 losses (two block ends, replay chains of up to 140 steps on rows that
    are touched once) stay inside BASELINE's 1e-4: the sharp test of the lazy replay and of the block-end catch-up.

    Measured [MI355X, round 3]: geometry stage (57 iterations) <= 6.2e-7 with either Adam; with the decoder training the
    colour-stage losses drift to ~1e-2 (mean 1.2e-3) for the lazy AND the dense IEEE sweep alike (ratio 0.99)."""
    from oracle import pointslam_oracle as O
    from point_slam_amd import params as P_
    dev = torch.device("cuda:0")
    cfg, cam, frames, pts = _long_scene(dev)
    frozen = semantics.endswith("frozen-decoder")
    if frozen:
        cfg["mapping"]["fix_color_decoder"] = True
        semantics = "torch2"
    # torch1 differs from torch2 only in the decoder group's step counter: 72 iterations (one prefetch-block end, 28 geometry
    # + 44 colour iterations) show it; the 140-iteration runs are the other two variants
    n_iters, ppf = (72 if semantics == "torch1" else 140), 400
    n_geo = int(n_iters * cfg["mapping"]["geo_iter_ratio"])
    g = torch.Generator().manual_seed(21)
    idx = torch.randint(cam["H"] * cam["W"], (n_iters, 3 * ppf), generator=g, dtype=torch.int32)
    fb = torch.zeros(n_iters, 2, 32).normal_(mean=0, std=0.01, generator=g)
    draws = (idx.to(dev).contiguous(), fb.to(dev).contiguous())
    torch1 = semantics == "torch1"
    runs = {}
    for lazy in ((True, False) if not torch1 else (True,)):
        s, sel = _native_long_run(cfg, cam, frames, pts, dev, n_iters, ppf, n_geo, draws, lazy, semantics,
                                  n_mapped=1 if torch1 else 0)
        runs[lazy] = (s.last_losses.cpu().double()[:, 0], s.npc.geo_feats.cpu().clone(), s.npc.col_feats.cpu().clone(),
                      P_.unpack_master(s.theta.cpu()), sel.cpu().long())
        if lazy:
            # the oracle starts from the state this run STARTED from: a fresh, identically seeded instance
            from point_slam_amd.decoders import PointDecoders
            from point_slam_amd.slam import HipSLAM
            s0 = HipSLAM(cfg, cam, device="cuda:0", max_points=200000, engine="native",
                         decoders=PointDecoders(cfg).load_reference_state(load_decoders("replica")))
            s0.seed_points(pts, seed=77)
            st = _oracle_inputs(s0)
            del s0
    sel = runs[True][4]
    assert sel.shape[0] >= 20000
    O.KNN_WORKERS = 8
    ls_o, geo_o, col_o, P_o, _, opt = O.mapper_iterations(cfg, st["P"], st["cloud"], st["geo"], st["col"], sel,
                                                          _oracle_frames(frames), idx.reshape(n_iters, 3, ppf), fb, n_geo,
                                                          cam, torch1_zero_grads=torch1)
    ref = torch.tensor(ls_o, dtype=torch.float64)
    rep = dict(test="map_140_iterations_vs_oracle", semantics=semantics, frozen_decoder=frozen, n_sel=int(sel.shape[0]), n_iters=n_iters,
               n_geo=n_geo, ref_first=float(ref[0]), ref_last=float(ref[-1]))
    for lazy, (ls, geo, col, theta, _) in runs.items():
        tag = "lazy" if lazy else "dense"
        rel = (ls - ref).abs() / ref.abs()
        dg, dc = (geo[sel] - geo_o[sel]).abs(), (col[sel] - col_o[sel]).abs()
        dd = max(float((theta[k] - P_o[k]).abs().max()) for k in theta if k.startswith("color_decoder") and k in P_o)
        rep[f"{tag}_geo_max"], rep[f"{tag}_col_max"] = float(dg.max()), float(dc.max())
        rep.update({f"{tag}_loss_rel_first20": float(rel[:20].max()), f"{tag}_loss_rel_geo_stage": float(rel[:n_geo + 1].max()),
                    f"{tag}_loss_rel_at_64": float(rel[60:70].max()), f"{tag}_loss_rel_at_128": float(rel[124:134].max()) if n_iters > 134 else None,
                    f"{tag}_loss_rel_max": float(rel.max()), f"{tag}_loss_rel_mean": float(rel.mean()),
                    f"{tag}_loss_rel_last": float(rel[-1]),
                    f"{tag}_geo_mean": float(dg.mean()), f"{tag}_geo_frac_gt_1e3": float((dg > 1e-3).float().mean()),
                    f"{tag}_col_mean": float(dc.mean()), f"{tag}_col_frac_gt_1e3": float((dc > 1e-3).float().mean()),
                    f"{tag}_dec_max": dd})
    rep["adam_steps_oracle"] = {k: v["step"] for k, v in opt.state.items() if k in ("geo", "col", "color_decoder.pts_linears.1.weight")}
    report(**rep)
    # the decoder group's step counter: n_colour iterations (torch >= 2) or all iterations (torch 1.12)
    n_col = n_iters - (n_geo + 1)
    if not frozen:
        assert opt.state["color_decoder.pts_linears.1.weight"]["step"] == (n_iters if torch1 else n_col)
    assert opt.state["geo"]["step"] == n_iters and opt.state["col"]["step"] == n_col
    for tag in (("lazy", "dense") if not torch1 else ("lazy",)):
        assert rep[f"{tag}_loss_rel_first20"] <= 1e-4          # identical state, identical arithmetic order
        assert rep[f"{tag}_loss_rel_geo_stage"] <= 1e-4        # 57 iterations, lr 0.03, one replay chain per touched row
        if frozen:
            # only feature rows move: BASELINE's bound holds for every one of the 140 iterations, across both block ends
            assert rep[f"{tag}_loss_rel_max"] <= 1e-4
            assert rep[f"{tag}_geo_mean"] < 1e-4 and rep[f"{tag}_col_mean"] < 1e-4
            continue
        # with the decoder training, Adam's sign sensitivity amplifies rounding noise (in the reference too): bounded drift
        assert rep[f"{tag}_loss_rel_max"] <= 3e-2 and rep[f"{tag}_loss_rel_mean"] <= 4e-3
        assert rep[f"{tag}_geo_mean"] < 8e-3 and rep[f"{tag}_col_mean"] < 2e-2
    if not torch1:
        # the lazy replay (hardware rcp / sqrt, block-end catch-up) is no farther from the oracle than the dense IEEE sweep
        assert rep["lazy_loss_rel_mean"] <= 3.0 * rep["dense_loss_rel_mean"] + 1e-5
        assert rep["lazy_geo_mean"] <= 3.0 * rep["dense_geo_mean"] + 1e-6
        assert rep["lazy_col_mean"] <= 3.0 * rep["dense_col_mean"] + 1e-6


def test_map_iters_expo_weighting_vs_oracle():
    """pointcloud.nn_weighting = 'expo' (decoder.py:154-156, 364-366; no shipped config uses it, the reference's tracker cannot run with it)
    through psl_map_iters: 24 iterations (9 geometry-stage ones through the one-launch geometry kernel, then the colour stage with F_theta),
    frozen colour decoder so that nothing amplifies rounding, against O.mapper_iterations on identical draws -- every loss inside 1e-4."""
    from oracle import pointslam_oracle as O
    dev = torch.device("cuda:0")
    cfg, cam, frames, pts = _long_scene(dev)
    cfg["pointcloud"]["nn_weighting"] = "expo"
    cfg["mapping"]["fix_color_decoder"] = True
    n_iters, ppf = 24, 400
    n_geo = int(n_iters * cfg["mapping"]["geo_iter_ratio"])
    g = torch.Generator().manual_seed(31)
    idx = torch.randint(cam["H"] * cam["W"], (n_iters, 3 * ppf), generator=g, dtype=torch.int32)
    fb = torch.zeros(n_iters, 2, 32).normal_(mean=0, std=0.01, generator=g)
    draws = (idx.to(dev).contiguous(), fb.to(dev).contiguous())
    s, sel = _native_long_run(cfg, cam, frames, pts, dev, n_iters, ppf, n_geo, draws, True, "torch2", n_mapped=0)
    ls = s.last_losses.cpu().double()[:, 0]
    geo, col = s.npc.geo_feats.cpu().clone(), s.npc.col_feats.cpu().clone()
    sel = sel.cpu().long()
    from point_slam_amd.decoders import PointDecoders
    from point_slam_amd.slam import HipSLAM
    s0 = HipSLAM(cfg, cam, device="cuda:0", max_points=200000, engine="native",
                 decoders=PointDecoders(cfg).load_reference_state(load_decoders("replica")))
    s0.seed_points(pts, seed=77)
    st = _oracle_inputs(s0)
    del s0
    O.KNN_WORKERS = 8
    ls_o, geo_o, col_o, _, _, _ = O.mapper_iterations(cfg, st["P"], st["cloud"], st["geo"], st["col"], sel, _oracle_frames(frames),
                                                      idx.reshape(n_iters, 3, ppf), fb, n_geo, cam, torch1_zero_grads=False)
    ref = torch.tensor(ls_o, dtype=torch.float64)
    rel = (ls - ref).abs() / ref.abs()
    # the same run with 'distance' weights must differ visibly: the option is live in every kernel of the loop
    cfg_d, _, _, _ = _long_scene(dev)
    cfg_d["mapping"]["fix_color_decoder"] = True
    s_d, _ = _native_long_run(cfg_d, cam, frames, pts, dev, n_iters, ppf, n_geo, draws, True, "torch2", n_mapped=0)
    ls_d = s_d.last_losses.cpu().double()[:, 0]
    rep = dict(test="map_iters_expo_vs_oracle", n_iters=n_iters, n_geo=n_geo, loss_rel_max=float(rel.max()), loss_rel_geo_stage=float(rel[:n_geo + 1].max()),
               geo_mean=float((geo[sel] - geo_o[sel]).abs().mean()), col_mean=float((col[sel] - col_o[sel]).abs().mean()),
               distance_vs_expo_loss_rel=float(((ls_d - ls).abs() / ls.abs()).max()))
    report(**rep)
    assert rep["loss_rel_max"] <= 1e-4 and rep["geo_mean"] < 1e-4 and rep["col_mean"] < 1e-4
    assert rep["distance_vs_expo_loss_rel"] > 1e-3
