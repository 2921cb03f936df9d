"""GPU, BASELINE.json's full sizes (1 M neural points, 640x480, up to 10 000 rays = 50 000 samples per launch, the
TUM/ScanNet mapping batch).

  * loss-level parity against the pinned oracle WHERE THE METRIC IS QUOTED: one tracker iteration (200 px), one
    geometry-stage and one colour-stage mapper iteration (1 000 px, then 10 000 px) on the 1 M-point map through
    O.render_batch_ray + O.tracker_loss / O.mapper_loss vs psl_render_fwd / psl_render_bwd (tests/parity_probe.py):
    loss rel-err <= 1e-4 (BASELINE.json's bound), per-ray depth / rgb and gradient rel-L2 reported and bounded;
  * exact kNN against the oracle on 7 500 sample points, all kernel variants;
  * size-independent properties on top: batch-splitting invariance of the render, linearity of the backward pass in
    its cotangents, idempotence of point adding."""
import pytest
import torch

from tests.test_hip_parity import report

pytestmark = pytest.mark.gpu


@pytest.fixture(scope="module")
def world():
    from point_slam_amd import synthetic as syn
    from point_slam_amd.config import default_config
    from point_slam_amd.slam import Frame, HipSLAM
    dev = torch.device("cuda:0")
    cfg = default_config()
    cam = syn.intrinsics(640, 480)
    torch.manual_seed(1219)
    s = HipSLAM(cfg, cam, device="cuda:0", max_points=1_400_000, engine="native")
    pts = syn.seed_cloud(cam, 1_000_000, n_views=64, seed=1219)
    s.seed_points(pts)
    c2w = syn.pose(200.0, dev)
    depth, color = syn.render_frame(cam, c2w)
    r_add, r_q = syn.dynamic_radii(color, cfg)
    fr = Frame(0, depth, color, r_add, r_q, c2w)
    return dict(cfg=cfg, cam=cam, slam=s, pts=pts, frame=fr, dev=dev)


def _rays(w, n, seed):
    from point_slam_amd import host_ops as H
    cam, fr, dev = w["cam"], w["frame"], w["dev"]
    g = torch.Generator(device="cpu").manual_seed(seed)
    idx = torch.randint(cam["H"] * cam["W"], (n,), generator=g).to(dev)
    u, v = H.pixels_from_flat_index(idx, 0, cam["H"], 0, cam["W"])
    ro, rd = H.get_rays_from_uv(u, v, fr.c2w, cam["fx"], cam["fy"], cam["cx"], cam["cy"])
    vi, ui = v.long(), u.long()
    return ro.contiguous(), rd.contiguous(), fr.depth[vi, ui].contiguous(), fr.color[vi, ui], fr.r_query[vi, ui].contiguous()


def test_knn_exact_at_1m_points(world):
    from oracle import pointslam_oracle as O
    w = world
    ro, rd, gd, gc, rq = _rays(w, 1500, 3)
    z = O.z_samples(gd.cpu(), 0.98, 1.02, 5)
    q = O.sample_points(ro.cpu(), rd.cpu(), z)                      # 7 500 sample points of real rays
    r = rq.cpu().repeat_interleave(5)
    D, I, cnt = w["slam"].npc.find_neighbors_faiss(q.to(w["dev"]), step="query", dynamic_radius=r.to(w["dev"]))
    Do, Io = O.knn_exact(w["pts"], q, 8)
    inr = Do <= (r * r)[:, None]
    Io_m = torch.where(inr, Io, torch.full_like(Io, -1))
    assert torch.equal(I.cpu(), Io_m)
    assert torch.equal(cnt.cpu(), O.neighbor_count(Do, r))
    report(test="fullsize_knn", queries=int(q.shape[0]), mean_cnt=float(cnt.float().mean()))


def _render(w, ro, rd, gd, rq, stage="color", grads=None):
    s = w["slam"]
    s.renderer.fixed_fallback = (torch.zeros(32, device=w["dev"]), torch.zeros(32, device=w["dev"]))
    s.renderer.sigmoid_coefficient = 0.1
    if grads is None:
        with torch.no_grad():
            return s.renderer.render_batch_ray(s.npc, s.decoders, rd, ro, w["dev"], stage, gt_depth=gd,
                                               npc_geo_feats=s.npc.geo_feats, npc_col_feats=s.npc.col_feats,
                                               dynamic_r_query=rq)
    geo = s.npc.geo_feats.detach().clone().requires_grad_(True)
    col = s.npc.col_feats.detach().clone().requires_grad_(True)
    for p in s.decoders.parameters():
        p.requires_grad_(True)
        p.grad = None
    d, v, c, valid = s.renderer.render_batch_ray(s.npc, s.decoders, rd, ro, w["dev"], stage, gt_depth=gd,
                                                 npc_geo_feats=geo, npc_col_feats=col, dynamic_r_query=rq)
    gdp, grgb = grads
    ((d * gdp).sum() + (c * grgb).sum()).backward()
    theta = torch.cat([p.grad.reshape(-1) for n, p in s.decoders.named_parameters()
                       if n.startswith("color_decoder") and p.grad is not None])
    return geo.grad, col.grad, theta


def test_render_is_invariant_to_batch_splitting(world, color_structure):
    """10 000 rays in one launch (50 000 samples: 32-slot tiles, several rounds of workgroups) == the same rays in
    launches of 5 000 / 3 000 / 2 000 (other tile geometries).  Rays are independent in the reference."""
    w = world
    ro, rd, gd, gc, rq = _rays(w, 10000, 11)
    d, v, c, valid = _render(w, ro, rd, gd, rq)
    parts = [(0, 5000), (5000, 8000), (8000, 10000)]
    dd, vv, cc, va = [], [], [], []
    for a, b in parts:
        o = _render(w, ro[a:b].contiguous(), rd[a:b].contiguous(), gd[a:b].contiguous(), rq[a:b].contiguous())
        dd.append(o[0]); vv.append(o[1]); cc.append(o[2]); va.append(o[3])
    d2, v2, c2, va2 = torch.cat(dd), torch.cat(vv), torch.cat(cc), torch.cat(va)
    rep = dict(test="fullsize_split", depth=float((d - d2).abs().max()), rgb=float((c - c2).abs().max()),
               valid_frac=float(valid.float().mean()))
    report(**rep)
    assert torch.equal(valid, va2)
    # different tile geometries sum the MFMA k-loop in a different order: fp32 noise only
    assert rep["depth"] < 1e-5 and rep["rgb"] < 2e-5
    assert float(valid.float().mean()) > 0.5


def test_backward_is_linear_in_cotangents(world, color_structure):
    w = world
    ro, rd, gd, gc, rq = _rays(w, 5000, 12)
    g = torch.Generator(device="cpu").manual_seed(1)
    a_d, a_c = torch.randn(5000, generator=g).to(w["dev"]), torch.randn(5000, 3, generator=g).to(w["dev"])
    b_d, b_c = torch.randn(5000, generator=g).to(w["dev"]), torch.randn(5000, 3, generator=g).to(w["dev"])
    ga = _render(w, ro, rd, gd, rq, grads=(a_d, a_c))
    gb = _render(w, ro, rd, gd, rq, grads=(b_d, b_c))
    gs = _render(w, ro, rd, gd, rq, grads=(a_d + 2.0 * b_d, a_c + 2.0 * b_c))
    worst = 0.0
    for x, y, z in zip(ga, gb, gs):
        ref = x + 2.0 * y
        worst = max(worst, float((z - ref).abs().max() / ref.abs().max().clamp_min(1e-20)))
    report(test="fullsize_linearity", worst_rel=worst)
    assert worst < 5e-4          # float atomics in the scatter + fp32 sums over 25 000 samples


def test_add_points_is_idempotent(world):
    """Surface points of a frame are added once; presenting the same pixels again adds nothing (dedupe radius,
    neural_point.py:118-121) -- at 1 M points."""
    from point_slam_amd.slam import Frame
    w = world
    s, fr0 = w["slam"], w["frame"]
    # the seeded cloud is denser than the normal add radius everywhere: shrink the radius map so that the first
    # pass really adds points
    fr = Frame(0, fr0.depth, fr0.color, fr0.r_add * 0.1, fr0.r_query, fr0.c2w)
    n0 = s.npc.pts_num()
    torch.manual_seed(5)
    st = torch.cuda.get_rng_state()         # add_points draws its pixels on the device generator
    a1 = s.add_points(fr, fr.c2w, n_pixels=6000)
    n1 = s.npc.pts_num()
    torch.cuda.set_rng_state(st)
    a2 = s.add_points(fr, fr.c2w, n_pixels=6000)
    report(test="fullsize_add", added_first=a1, added_second=a2, points=n1)
    assert a1 > 500 and n1 == n0 + 3 * a1 and a2 == 0 and s.npc.pts_num() == n1
    assert s.npc.get_geo_feats().shape[0] == n1


@pytest.mark.parametrize("n_rays,seed", [(1500, 4), (37, 5)])
def test_ray_knn_matches_oracle_at_1m_points(world, n_rays, seed):
    """The ray-mode k-NN inside psl_render_fwd (one wavefront per ray, five samples share the candidate scan): neighbour
    lists and counts read back from the render workspace == exact 8-NN of the oracle, bit for bit."""
    import ctypes as C
    from oracle import pointslam_oracle as O
    from point_slam_amd import _lib
    w = world
    s, dev = w["slam"], w["dev"]
    ro, rd, gd, gc, rq = _rays(w, n_rays, seed)
    L = _lib.lib()
    R = n_rays
    ws = torch.zeros(int(L.psl_render_ws_floats(R, 0)), device=dev)
    depth = torch.empty(R, device=dev); var = torch.empty(R, device=dev); rgb = torch.empty(R, 3, device=dev)
    valid = torch.empty(R, device=dev, dtype=torch.uint8)
    fb = torch.zeros(2, 32, device=dev)
    a = _lib.psl_render_args(n_rays=R, flags=0, sigmoid_coef=0.1, rays_o=ro.data_ptr(), rays_d=rd.data_ptr(),
                             gt_depth=gd.data_ptr(), r_query=rq.data_ptr(), geo_feats=s.npc.geo_feats.data_ptr(),
                             col_feats=s.npc.col_feats.data_ptr(), params=s.theta.data_ptr(),
                             col_embed_B=s.Bcol.data_ptr(), fallback_geo=fb[0].data_ptr(), fallback_col=fb[1].data_ptr(),
                             exposure_affine=None, ws=ws.data_ptr(), depth=depth.data_ptr(), var=var.data_ptr(),
                             rgb=rgb.data_ptr(), valid_ray=valid.data_ptr())
    P = 5 * R
    Ppad = (P + 15) // 16 * 16
    got = {}
    # the two kernels that exist: one wavefront per ray (the mapper's side-stream prefetch), one per sample -- the latter with its two
    # round-6 mechanisms (start pass chosen from the row lengths, eighth-best bound carried into the next pass) off everywhere (0) and
    # on at every launch size (15; the default, 3, applies them below 5 000 queries only)
    for ver, hint in ((2, 3), (4, 0), (4, 15)):
        _lib.check(L.psl_debug_option(b"knn", ver))
        _lib.check(L.psl_debug_option(b"knn_start_hint", hint))
        ws.zero_()
        _lib.check(L.psl_render_fwd(s.npc.handle, C.byref(a), _lib.stream_ptr()))
        torch.cuda.synchronize()
        got[(ver, hint)] = (ws[:Ppad * 8].view(torch.int32).reshape(Ppad, 8)[:P].cpu().long().clone(),
                            ws[Ppad * 8:Ppad * 9].view(torch.int32)[:P].cpu().clone())
    _lib.check(L.psl_debug_option(b"knn_start_hint", 3))
    assert torch.equal(got[(4, 0)][0], got[(4, 15)][0]) and torch.equal(got[(4, 0)][1], got[(4, 15)][1])
    got = {2: got[(2, 3)], 4: got[(4, 15)]}
    I, cnt = got[2]
    z = O.z_samples(gd.cpu(), 0.98, 1.02, 5)
    q = O.sample_points(ro.cpu(), rd.cpu(), z)
    r = rq.cpu().repeat_interleave(5)
    cloud = s.npc.cloud_pos()           # the module's cloud as it is NOW (an earlier test may have added points)
    Do, Io = O.knn_exact(cloud, q, 8)
    inr = Do <= (r * r)[:, None]
    Io_m = torch.where(inr, Io, torch.full_like(Io, -1))
    bad = int((I != Io_m).any(1).sum())
    bad1 = int((got[4][0] != Io_m).any(1).sum())
    assert torch.equal(got[4][1], got[2][1])
    diag = []
    for p_ in torch.nonzero((I != Io_m).any(1)).flatten()[:4].tolist():
        diag.append(dict(sample=p_, s=p_ % 5, r=float(r[p_]), want=Io_m[p_].tolist(), got=I[p_].tolist(),
                         v4=got[4][0][p_].tolist(), D=[float(x) for x in Do[p_]]))
    report(test="fullsize_ray_knn", rays=R, mismatched_samples=bad, mismatched_v1=bad1, mean_cnt=float(cnt.float().mean()),
           diag=diag)
    _lib.check(L.psl_debug_option(b"knn", 0))       # back to the default (kernel chosen by launch size)
    assert bad1 == 0
    assert bad == 0
    assert torch.equal(cnt, O.neighbor_count(Do, r))


# ---------------------------------------------------------------------------------------------------------------------
# loss-level parity against the oracle at the headline configuration ("next" 1a)
LOSS_PARITY_CASES = [("tracker", 200), ("map_geometry", 1000), ("map_color", 1000), ("map_color", 10000),
                     ("map_geometry", 10000), ("tracker", 5000)]


@pytest.fixture(scope="module")
def oracle_world(world):
    from tests import parity_probe as PP
    return PP.oracle_state(world["slam"])       # 1 M points: 268 MB to the host, kd-tree built once by the first probe


@pytest.mark.parametrize("kind,n_pix", LOSS_PARITY_CASES)
def test_loss_parity_vs_oracle_at_1m_points(world, oracle_world, kind, n_pix, color_structure):
    """|L_hip - L_oracle| / |L_oracle| <= 1e-4 at 1 M points, for the geometry and the colour term separately, plus
    per-ray outputs and every gradient the iteration produces (SURVEY.md 8d parity protocol)."""
    from tests import parity_probe as PP
    w = world
    rep = PP.probe(w["slam"], w["cfg"], w["cam"], w["frame"], kind, n_pix, seed=100 + n_pix, state=oracle_world)
    report(test="fullsize_loss_parity", **rep)
    assert rep["points"] >= 1_000_000 and rep["rays"] > 0.9 * n_pix and rep["valid_frac"] > 0.5
    assert rep["mask_mismatch"] == 0 and rep["valid_mismatch"] == 0
    assert rep["loss_rel"] <= 1e-4 and rep["geo_loss_rel"] <= 1e-4 and rep["col_loss_rel"] <= 1e-4
    assert rep["depth_rel_max"] < 2e-4 and rep["rgb_abs_max"] < 5e-3
    if kind == "tracker":
        assert rep["g_rays_o_rel_l2"] < 2e-3 and rep["g_rays_d_rel_l2"] < 2e-3
        assert rep["g_rays_o_cos"] > 0.99999 and rep["g_rays_d_cos"] > 0.99999
    else:
        assert rep["g_geo_rel_l2"] < 1e-3 and rep["g_geo_cos"] > 0.99999
        if kind == "map_color":
            assert rep["g_col_rel_l2"] < 1e-3 and rep["g_col_cos"] > 0.99999
            assert rep["g_params_rel_l2"] < 1e-3 and rep["g_params_cos"] > 0.99999


@pytest.mark.parametrize("kind,n_pix", [("tracker", 5000), ("map_color", 10000)])
def test_loss_parity_on_the_noisy_depth_stream(world, oracle_world, kind, n_pix):
    """BASELINE config 3 ("TUM fr1_desk ... noisy-depth path with 5k pixels/iter"): the TUM-like stream of SURVEY.md 8d --
    sensor depth with 0.5 % multiplicative noise and 2 % of the pixels dropped to 0 -- at the TUM pixel budgets on the 1 M-point
    map.  The holes go through the depth > 0 compaction, the noisy depths move the five samples of a ray off the map's
    surfaces (fewer neighbours inside the radius, more invalid rays); same bounds as the clean stream."""
    from point_slam_amd import synthetic as syn
    from point_slam_amd.slam import Frame
    from tests import parity_probe as PP
    w = world
    g = torch.Generator(device=w["dev"]).manual_seed(17)
    fr0 = w["frame"]
    depth, color = syn.render_frame(w["cam"], fr0.c2w, noise=0.005, dropout=0.02, gen=g)
    fr = Frame(0, depth, color, fr0.r_add, fr0.r_query, fr0.c2w)
    holes = float((depth == 0).float().mean())
    rep = PP.probe(w["slam"], w["cfg"], w["cam"], fr, kind, n_pix, seed=300 + n_pix, state=oracle_world)
    report(test="fullsize_loss_parity_noisy_depth", holes=holes, **rep)
    assert 0.015 < holes < 0.025
    assert rep["points"] >= 1_000_000 and 0.95 * n_pix < rep["rays"] < 0.995 * n_pix        # the holes are gone from the batch
    assert rep["mask_mismatch"] == 0 and rep["valid_mismatch"] == 0
    assert rep["loss_rel"] <= 1e-4 and rep["geo_loss_rel"] <= 1e-4 and rep["col_loss_rel"] <= 1e-4
    assert rep["depth_rel_max"] < 2e-4 and rep["rgb_abs_max"] < 5e-3
    if kind == "tracker":
        assert rep["g_rays_o_rel_l2"] < 2e-3 and rep["g_rays_d_rel_l2"] < 2e-3
    else:
        assert rep["g_geo_rel_l2"] < 1e-3 and rep["g_col_rel_l2"] < 1e-3 and rep["g_params_rel_l2"] < 1e-3


# ---------------------------------------------------------------------------------------------------------------------
# BASELINE.json configs[4]: synthetic 1280x960 stream, 2 M seeded neural points (the grid index works with <= 2^22 cells:
# untested at that density before round 3)
@pytest.fixture(scope="module")
def world_cfg5():
    from point_slam_amd import synthetic as syn
    from point_slam_amd.config import default_config
    from point_slam_amd.slam import Frame, HipSLAM
    dev = torch.device("cuda:0")
    cfg = default_config()
    cam = syn.intrinsics(1280, 960)
    torch.manual_seed(1219)
    s = HipSLAM(cfg, cam, device="cuda:0", max_points=2_400_000, engine="native")
    pts = syn.seed_cloud(cam, 2_000_000, n_views=64, seed=1219)
    s.seed_points(pts)
    c2w = syn.pose(200.0, dev)
    depth, color = syn.render_frame(cam, c2w)
    r_add, r_q = syn.dynamic_radii(color, cfg)
    fr = Frame(0, depth, color, r_add, r_q, c2w)
    return dict(cfg=cfg, cam=cam, slam=s, pts=pts, frame=fr, dev=dev)


def test_cfg5_knn_exact_at_2m_points(world_cfg5):
    """Free-query and ray-mode k-NN (every kernel variant) at 2 M points / 1280x960: bit-exact against the oracle."""
    from oracle import pointslam_oracle as O
    w = world_cfg5
    assert w["slam"].npc.pts_num() == 2_000_000
    ro, rd, gd, gc, rq = _rays(w, 1200, 21)
    z = O.z_samples(gd.cpu(), 0.98, 1.02, 5)
    q = O.sample_points(ro.cpu(), rd.cpu(), z)
    r = rq.cpu().repeat_interleave(5)
    D, I, cnt = w["slam"].npc.find_neighbors_faiss(q.to(w["dev"]), step="query", dynamic_radius=r.to(w["dev"]))
    Do, Io = O.knn_exact(w["pts"], q, 8)
    inr = Do <= (r * r)[:, None]
    Io_m = torch.where(inr, Io, torch.full_like(Io, -1))
    assert torch.equal(I.cpu(), Io_m) and torch.equal(cnt.cpu(), O.neighbor_count(Do, r))
    report(test="cfg5_knn", points=2_000_000, queries=int(q.shape[0]), mean_cnt=float(cnt.float().mean()))


def test_cfg5_render_split_invariance_and_loss_parity(world_cfg5):
    """2 M points, 1280x960: the render of 6 000 rays equals the same rays in three launches, and one tracker / one
    colour-stage mapper iteration agree with the oracle to BASELINE's 1e-4 at the loss level."""
    from tests import parity_probe as PP
    w = world_cfg5
    ro, rd, gd, gc, rq = _rays(w, 6000, 22)
    d, v, c, valid = _render(w, ro, rd, gd, rq)
    dd, cc = [], []
    for a, b in [(0, 2500), (2500, 4000), (4000, 6000)]:
        o = _render(w, ro[a:b].contiguous(), rd[a:b].contiguous(), gd[a:b].contiguous(), rq[a:b].contiguous())
        dd.append(o[0]); cc.append(o[2])
    e_d, e_c = float((d - torch.cat(dd)).abs().max()), float((c - torch.cat(cc)).abs().max())
    assert e_d < 1e-5 and e_c < 2e-5 and float(valid.float().mean()) > 0.5
    st = PP.oracle_state(w["slam"])
    reps = [PP.probe(w["slam"], w["cfg"], w["cam"], w["frame"], kind, n, seed=300 + n, state=st)
            for kind, n in (("tracker", 200), ("map_color", 1000))]
    for rep in reps:
        report(test="cfg5_loss_parity", **rep)
        assert rep["points"] == 2_000_000 and rep["mask_mismatch"] == 0 and rep["valid_mismatch"] == 0
        assert rep["loss_rel"] <= 1e-4 and rep["geo_loss_rel"] <= 1e-4 and rep["col_loss_rel"] <= 1e-4
    report(test="cfg5_split", depth=e_d, rgb=e_c, valid_frac=float(valid.float().mean()))


# ---------------------------------------------------------------------------------------------------------------------
# outcome-level parity of a LONG tracker run ("next" 4)
LONG_TRACKER = [(2000, 60)] + ([(5000, 200)] if __import__("os").environ.get("PSL_LONG_TESTS") == "1" else [])


@pytest.fixture(scope="module")
def trained_world():
    """A map the tracker can actually converge on: 300 k seeded points, three keyframes mapped at their true poses (the
    fullsize `world` carries its random initial features: on it the render loss has no minimum at the true pose and a long
    tracker run -- the oracle's as much as the kernels' -- random-walks away from it)."""
    from point_slam_amd import synthetic as syn
    from point_slam_amd.config import default_config
    from point_slam_amd.slam import Frame, HipSLAM
    from tests import parity_probe as PP
    dev = torch.device("cuda:0")
    cfg = default_config()
    cam = syn.intrinsics(640, 480)
    torch.manual_seed(77)
    s = HipSLAM(cfg, cam, device="cuda:0", max_points=500_000, engine="native")
    s.seed_points(syn.seed_cloud(cam, 300_000, n_views=48, seed=77))

    def frame_at(t, idx, **kw):
        c2w = syn.pose(t, dev)
        depth, color = syn.render_frame(cam, c2w, **kw)
        r_add, r_q = syn.dynamic_radii(color, cfg)
        return Frame(idx, depth, color, r_add, r_q, c2w)
    for k, t in enumerate((196.0, 200.0, 204.0, 198.0, 202.0)):
        kf = frame_at(t, k)
        s.map(kf, kf.c2w, n_iters=150, fixed_iters=True)
        s.keyframes.append(kf)
    torch.cuda.synchronize()
    g = torch.Generator(device=dev).manual_seed(23)
    fr = frame_at(201.0, 9, noise=0.005, dropout=0.02, gen=g)         # the TUM-like stream: 0.5 % depth noise, 2 % holes
    return dict(cfg=cfg, cam=cam, slam=s, frame=fr, dev=dev, state=PP.oracle_state(s))


@pytest.mark.parametrize("n_pix,n_iters", LONG_TRACKER)
def test_long_tracker_run_ends_where_the_oracle_ends(trained_world, n_pix, n_iters):
    """Per-iteration parity (first-iteration loss, gradients) says nothing about what 200 chained Adam steps do: best-pose
    selection, the Adam moments across the run and the large-batch launch structure (ten launches per iteration, the split
    decode kernels) could hide a slow drift.  ONE frame of the TUM-like stream on a trained map, the tracker started ~1.5 cm
    off the truth, identical draws: psl_track_iters against O.tracker_loop from the same initial pose.  Yardstick: the oracle
    against ITSELF from an initial pose moved by one ulp -- Adam normalises gradient components at the rounding-noise level
    next to the optimum to steps of +-lr, so two exact runs end a few steps apart.  The HIP run must end inside a small
    multiple of that band, and both must end closer to the truth than they started.
    (2 000 px x 60 it always; the TUM yaml's 5 000 px x 200 it under PSL_LONG_TESTS=1 -- minutes of oracle time --, the round's
    run is in profiles/r06_long_tracker_outcome.json.)"""
    from oracle import pointslam_oracle as O
    from point_slam_amd.slam import camera_tensor_from_c2w
    w = trained_world
    s, cfg, cam, dev, st, fr = w["slam"], w["cfg"], w["cam"], w["dev"], w["state"], w["frame"]
    tr = cfg["tracking"]
    eh, ew = tr["ignore_edge_H"], tr["ignore_edge_W"]
    truth = camera_tensor_from_c2w(fr.c2w).cpu()
    cam0 = truth.clone()
    cam0[4:] += torch.tensor([0.009, -0.008, 0.010])
    cam0[:4] += torch.tensor([0.0, 0.0012, -0.0009, 0.0008])
    gi = torch.Generator(device="cpu").manual_seed(41)
    hi = (cam["H"] - 2 * eh) * (cam["W"] - 2 * ew)
    pix = torch.randint(hi, (n_iters, n_pix), generator=gi, dtype=torch.int32)
    fb = torch.zeros(n_iters, 2, 32).normal_(mean=0, std=0.01, generator=gi)
    full0 = tr.get("sample_with_color_grad", False)
    tr["sample_with_color_grad"] = False
    try:
        best = s._track_native(fr, cam0, n_iters, n_pix, draws=(pix.to(dev).contiguous(), fb.to(dev).contiguous())).cpu()
        torch.cuda.synchronize()
        hip_losses = s.last_losses.cpu().double()[:, 0]
        hip_end = s.last_cam.cpu()
        O.KNN_WORKERS = 16
        kw = dict(coef=cfg["rendering"]["sigmoid_coef_tracker"])
        od, oc, orq = fr.depth.cpu(), fr.color.cpu(), fr.r_query.cpu()
        ls, cams, obest, _, _ = O.tracker_loop(cfg, st["P"], st["cloud"], st["geo"], st["col"], cam0, pix, fb, od, oc, orq, cam, eh, ew, **kw)
        cam0_ulp = torch.nextafter(cam0, torch.full_like(cam0, 10.0))
        ls_u, cams_u, obest_u, _, _ = O.tracker_loop(cfg, st["P"], st["cloud"], st["geo"], st["col"], cam0_ulp, pix, fb, od, oc, orq, cam, eh, ew, **kw)
    finally:
        tr["sample_with_color_grad"] = full0
    ref = torch.tensor(ls, dtype=torch.float64)
    rel = (hip_losses - ref).abs() / ref.abs()
    self_rel = (torch.tensor(ls_u, dtype=torch.float64) - ref).abs() / ref.abs()      # the oracle against itself, one ulp apart
    band_best = float((obest_u - obest).abs().max())
    band_end = float((cams_u[-1] - cams[-1]).abs().max())
    err = lambda c: float((c[4:] - truth[4:]).norm() * 100.0)        # translation error against the truth, cm
    rep = dict(test="long_tracker_outcome", n_pix=n_pix, n_iters=n_iters, loss_rel_first=float(rel[0]), loss_rel_first10=float(rel[:10].max()),
               loss_rel_max=float(rel.max()), loss_first=float(ref[0]), loss_last_oracle=float(ref[-1]), loss_last_hip=float(hip_losses[-1]),
               best_abs=float((best - obest).abs().max()), end_abs=float((hip_end - cams[-1]).abs().max()),
               oracle_self_1ulp_best=band_best, oracle_self_1ulp_end=band_end, oracle_self_1ulp_loss_rel_first10=float(self_rel[:10].max()),
               oracle_self_1ulp_loss_rel_max=float(self_rel.max()), lr=tr["lr"],
               err_start_cm=err(cam0), err_hip_cm=err(best), err_oracle_cm=err(obest), err_oracle_ulp_cm=err(obest_u))
    report(**rep)
    # the first loss is a parity statement; the next nine already carry the run's own sensitivity (the trained map differs from run to
    # run -- float atomics in its training -- and so does how fast two exact runs part: 3e-4 and 2.6e-3 were both measured with
    # bit-identical kernels, tools/track_structs_probe.py), so they are held to the oracle's own one-ulp divergence over the same
    # iterations, never tighter than 5e-3
    assert rep["loss_rel_first"] < 1e-4
    assert rep["loss_rel_first10"] < max(5e-3, 4.0 * rep["oracle_self_1ulp_loss_rel_first10"]), rep
    # the end pose inside a small multiple of the oracle's own sensitivity (never tighter than a few Adam steps)
    band = max(band_best, band_end, 3.0 * tr["lr"])
    assert rep["best_abs"] <= 4.0 * band and rep["end_abs"] <= 4.0 * band
    # and the run does what tracking is for: both end closer to the truth than they started, HIP as close as the oracle to within the band
    assert rep["err_hip_cm"] < rep["err_start_cm"] and rep["err_oracle_cm"] < rep["err_start_cm"]
    assert abs(rep["err_hip_cm"] - rep["err_oracle_cm"]) <= 100.0 * 4.0 * band
