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
def test_map_iters_native_matches_reference_loop(case):
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


@pytest.mark.parametrize("case", ["tracker_iters_tum", "tracker_iters_scannet"])
def test_track_iters_native_matches_reference_loop(case):
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
