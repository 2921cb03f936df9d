"""Per-frame image operators: oracle known-answer tests (CPU) and HIP-vs-oracle parity (GPU).

The skimage pieces are parity-unpinned (skimage is not installed, the reference has no vectors for them): the oracle
restates skimage 0.19's published definition with the same scipy call; the CPU tests pin it to known answers."""
import numpy as np
import pytest
import torch

from tests.helpers import base_cfg


def _frame(W=160, H=120, t=3.0):
    from point_slam_amd import synthetic as syn
    cam = syn.intrinsics(W, H)
    c2w = syn.pose(t)
    depth, color = syn.render_frame(cam, c2w)
    return cam, c2w, depth, color


# ------------------------------------------------------------------------------------------------ CPU: the oracle
def test_oracle_sobel_known_answers():
    from oracle import frame_oracle as F
    img = np.zeros((9, 12, 3))
    img[...] = (np.arange(12) * 0.01)[None, :, None]            # horizontal ramp, slope 0.01 / pixel
    gm = F.grad_magnitude(img)
    assert np.allclose(gm[:, 1:-1], 0.02, atol=1e-15)           # [1,0,-1] x [1,2,1]/4 on a ramp = 2 * slope
    assert np.allclose(gm[:, 0], 0.01, atol=1e-15)              # 'reflect' border duplicates the edge pixel
    assert np.allclose(F.rgb2gray(np.ones((2, 2, 3))), 1.0)
    flat = F.grad_magnitude(np.full((5, 5, 3), 0.3))
    assert np.all(np.abs(flat) < 1e-15)


def test_oracle_radius_map_knots():
    from oracle import frame_oracle as F
    cfg = base_cfg()
    img = np.zeros((6, 40, 3))
    img[...] = (np.arange(40) ** 2 * 0.0012)[None, :, None]     # growing slope: sweeps the whole gradient range
    r_add, r_query, gm = F.dynamic_radius_maps(img, cfg)
    assert r_add.max() == 0.08 and abs(r_add.min() - 0.02) < 1e-15
    assert np.allclose(r_query, 2.0 * r_add, atol=1e-15)
    lo = gm <= 0.01
    assert np.all(r_add[lo] == 0.08)
    mid = (gm > 0.01) & (gm < 0.15)
    assert np.allclose(r_add[mid], 0.08 + (0.02 - 0.08) * (gm[mid] - 0.01) / 0.14, atol=1e-15)
    assert np.all(np.abs(r_add[gm >= 0.15] - 0.02) < 1e-15)


def test_oracle_topk_is_the_topk():
    from oracle import frame_oracle as F
    rng = np.random.default_rng(3)
    gm = rng.random((30, 40))
    depth = (rng.random((30, 40)) > 0.2).astype(np.float32) * 2.0
    sel = F.selected_index_with_grad(5, 25, 5, 35, 10, gm, ratio=15, gt_depth=depth)
    thr = np.sort(gm.ravel())[-150]
    ih, iw = np.unravel_index(np.arange(gm.size), gm.shape)
    want = np.nonzero((gm.ravel() >= thr) & (ih >= 5) & (ih < 25) & (iw >= 5) & (iw < 35) & (depth.ravel() > 0))[0]
    assert np.array_equal(sel, want)


def test_oracle_keyframe_overlap_extremes():
    from oracle import frame_oracle as F
    from oracle import pointslam_oracle as O
    cam, c2w, depth, color = _frame()
    g = torch.Generator().manual_seed(1)
    u = torch.randint(25, cam["W"] - 25, (100,), generator=g).float()
    v = torch.randint(25, cam["H"] - 25, (100,), generator=g).float()
    ro, rd = O.rays_from_uv(u, v, c2w, cam["fx"], cam["fy"], cam["cx"], cam["cy"])
    gd = depth[v.long(), u.long()]
    c4 = torch.eye(4); c4[:3] = c2w[:3]
    away = c4.clone(); away[:3, :3] = c4[:3, :3] @ torch.diag(torch.tensor([-1.0, 1.0, -1.0]))   # looks backwards
    pct = F.keyframe_overlap(ro.numpy(), rd.numpy(), gd.numpy(), [c4.numpy(), away.numpy()], cam["H"], cam["W"],
                             cam["fx"], cam["fy"], cam["cx"], cam["cy"])
    assert pct[0] == 1.0 and pct[1] == 0.0          # same pose sees all of its own frustum, the reversed one none


def test_oracle_image_metrics_known_answers():
    from oracle import eval_oracle as E
    cam, c2w, depth, color = _frame(256, 192)
    psnr, ms, l1 = E.image_metrics(color, depth, (color + 0.1).clone(), depth + 0.25)
    assert abs(psnr - 20.0) < 1e-4 and abs(l1 - 0.25) < 1e-6          # uniform error e -> -20 log10 e
    assert abs(E.image_metrics(color, depth, color.clone(), depth)[1] - 1.0) < 1e-6
    assert abs(float(E.gauss_1d().sum()) - 1.0) < 1e-6
    g = torch.Generator().manual_seed(0)
    noisy = (color + 0.1 * torch.randn(color.shape, generator=g)).clamp(0, 1)
    assert 0.3 < E.image_metrics(color, depth, noisy, depth)[1] < 0.99


# ------------------------------------------------------------------------------------------------ GPU: HIP vs oracle
@pytest.mark.gpu
def test_frame_radii_match_oracle():
    from oracle import frame_oracle as F
    from point_slam_amd import frame_ops as FO
    cfg = base_cfg()
    dev = torch.device("cuda:0")
    cam, c2w, depth, color = _frame(640, 480)
    color = (color + 0.02 * torch.randn(color.shape, generator=torch.Generator().manual_seed(2))).clamp(0, 1)
    ra_o, rq_o, gm_o = F.dynamic_radius_maps(color.numpy().astype(np.float64), cfg)
    ra, rq, gm = FO.dynamic_radius_maps(color.to(dev), cfg, with_grad_mag=True)
    gm_err = float(np.abs(gm.cpu().numpy() - gm_o).max())
    assert gm_err < 1e-14
    # the radii leave as float32: equal to the rounded float64 oracle up to one ulp (fp64 summation order differs)
    assert float(np.abs(ra.cpu().numpy() - ra_o.astype(np.float32)).max()) <= 8e-9
    assert float(np.abs(rq.cpu().numpy() - rq_o.astype(np.float32)).max()) <= 1.5e-8
    assert ra_o.min() < 0.03 and ra_o.max() == 0.08      # the frame exercises both ends of the map


@pytest.mark.gpu
@pytest.mark.parametrize("depth_limit", [False, True])
def test_topgrad_select_matches_oracle(depth_limit):
    from oracle import frame_oracle as F
    from point_slam_amd import frame_ops as FO
    from tests.test_hip_parity import make_npc
    cfg = base_cfg()
    dev = torch.device("cuda:0")
    cam, c2w, depth, color = _frame(640, 480)
    g = torch.Generator().manual_seed(4)
    color = (color + 0.02 * torch.randn(color.shape, generator=g)).clamp(0, 1)
    depth = torch.where(torch.rand(depth.shape, generator=g) < 0.1, torch.zeros_like(depth), depth)
    depth = torch.where(torch.rand(depth.shape, generator=g) < 0.1, torch.full_like(depth, 6.0), depth)
    npc = make_npc(cfg, torch.zeros(0, 3), torch.zeros(0, 32), torch.zeros(0, 32), dev)
    _, _, gm = FO.dynamic_radius_maps(color.to(dev), cfg, with_grad_mag=True)
    n = 1000
    sel, _ = FO.get_selected_index_with_grad(npc, 20, 460, 20, 620, n, color.to(dev), ratio=15, gt_depth=depth.to(dev),
                                             depth_limit=depth_limit, grad_mag=gm)
    # oracle on the SAME gradient image (the selection is exact on it; ties at the threshold are measure-zero here)
    want = F.selected_index_with_grad(20, 460, 20, 620, n, gm.cpu().numpy(), ratio=15, gt_depth=depth.numpy(),
                                      depth_limit=depth_limit)
    assert sel.shape[0] == want.shape[0] and np.array_equal(sel.cpu().numpy(), want)
    assert 5000 < want.shape[0] < 15000
    # all-equal image: k of the tied pixels, whichever
    flat = torch.zeros(48, 64, dtype=torch.float64, device=dev)
    s2, _ = FO.get_selected_index_with_grad(npc, 0, 48, 0, 64, 10, None if False else torch.zeros(48, 64, 3, device=dev),
                                            ratio=15, grad_mag=flat)
    assert s2.shape[0] == 150 and len(set(s2.tolist())) == 150


@pytest.mark.gpu
def test_keyframe_overlap_matches_oracle():
    from oracle import frame_oracle as F
    from oracle import pointslam_oracle as O
    from point_slam_amd import frame_ops as FO
    from point_slam_amd import synthetic as syn
    dev = torch.device("cuda:0")
    cam, c2w, depth, color = _frame(640, 480, t=12.0)
    g = torch.Generator().manual_seed(7)
    n = 200
    u = torch.randint(0, cam["W"], (n,), generator=g).float()
    v = torch.randint(0, cam["H"], (n,), generator=g).float()
    ro, rd = O.rays_from_uv(u, v, c2w, cam["fx"], cam["fy"], cam["cx"], cam["cy"])
    gd = depth[v.long(), u.long()]
    kfs = []
    for t in (12.0, 15.0, 22.0, 40.0, 90.0, 170.0):
        c4 = torch.eye(4); c4[:3] = syn.pose(t)[:3]
        kfs.append(c4)
    want = F.keyframe_overlap(ro.numpy(), rd.numpy(), gd.numpy(), [k.numpy() for k in kfs], cam["H"], cam["W"],
                              cam["fx"], cam["fy"], cam["cx"], cam["cy"])
    got = FO.keyframe_overlap(ro.to(dev), rd.to(dev), gd.to(dev), kfs, cam)
    # a point within float32 rounding of the 20-pixel border may fall on either side: at most 2 of 1600 samples
    assert np.abs(got - want).max() <= 2.0 / (n * 8) + 1e-7
    assert want[0] > 0.5 and want.min() < 0.2
    sel = FO.keyframe_selection_overlap(ro.to(dev), rd.to(dev), gd.to(dev), kfs, cam, k=3,
                                        rng=np.random.default_rng(0))
    assert len(sel) <= 3 and all(want[i] > 0 for i in sel)


@pytest.mark.gpu
def test_keyframe_selection_overlap_matches_reference_fixture():
    """psl_keyframe_overlap_sync + the host permutation against the UNMODIFIED Mapper.keyframe_selection_overlap
    (src/Mapper.py:170-235; tests/golden/keyframe_overlap_ref.npz from oracle/gen_golden_frame.py): percent_inside of the
    eight keyframes (a sample within float32 rounding of the 20-pixel border may fall on either side: <= 2 of 920), the same
    keyframes with and without overlap, and -- np.random seeded as the generator seeded it -- the very list returned."""
    from point_slam_amd import frame_ops as FO
    from tests.helpers import load_npz
    dev = torch.device("cuda:0")
    k = load_npz("keyframe_overlap_ref")
    cam = dict(H=k["H"], W=k["W"], fx=k["fx"], fy=k["fy"], cx=k["cx"], cy=k["cy"])
    kfs = [c for c in k["kf_c2w"]]
    want = k["percent_inside"].numpy()
    ro, rd, gd = k["rays_o"].to(dev), k["rays_d"].to(dev), k["ray_depth"].to(dev)
    got = FO.keyframe_overlap(ro, rd, gd, kfs, cam)
    n = ro.shape[0] * 8
    assert np.abs(got - want).max() <= 2.0 / n + 1e-7
    assert np.array_equal(got > 0, want > 0)
    sel = FO.keyframe_selection_overlap(ro, rd, gd, kfs, cam, k=k["k"], rng=np.random.RandomState(k["seed"]))
    assert [int(i) for i in sel] == k["selected"].tolist()


@pytest.mark.gpu
@pytest.mark.parametrize("size", [(640, 480), (333, 201)])
def test_image_metrics_match_oracle(size):
    from oracle import eval_oracle as E
    from point_slam_amd import frame_ops as FO
    dev = torch.device("cuda:0")
    cam, c2w, depth, color = _frame(*size)
    g = torch.Generator().manual_seed(11)
    noisy = (color + 0.05 * torch.randn(color.shape, generator=g)).clamp(0, 1)
    d2 = depth * (1 + 0.01 * torch.randn(depth.shape, generator=g))
    gt_depth = torch.where(torch.rand(depth.shape, generator=g) < 0.1, torch.zeros_like(depth), depth)
    want = E.image_metrics(color, gt_depth, noisy, d2)
    got = FO.image_metrics(color.to(dev), gt_depth.to(dev), noisy.to(dev), d2.to(dev))
    assert abs(got[0] - want[0]) < 1e-4 * abs(want[0])      # PSNR (dB)
    assert abs(got[1] - want[1]) < 2e-5                     # MS-SSIM: float32 maps, different summation order
    assert abs(got[2] - want[2]) < 1e-6 * max(want[2], 1.0)
    assert 0.5 < want[1] < 0.999
