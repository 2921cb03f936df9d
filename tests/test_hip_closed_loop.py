"""Closed-loop parity: track -> map -> track over several frames through HipSLAM(engine="native"), every stage checked
against the pinned oracle composed the way the reference composes its loops, with identical random draws.

Every loop of the path is pinned in isolation elsewhere (tests/test_hip_loops.py: the reference's own optimize_map /
tracker loop / add_neural_points fixtures).  What nothing else checks is the ORCHESTRATION between them
(src/Tracker.py:283-290,379-380 <-> src/Mapper.py:263-330,404-406,642-783): the pose a tracked frame hands to the mapper, the
points the mapper adds at THAT pose, the frustum rows it then selects, the data-dependent iteration count, the trained
rows and decoder the next tracked frame renders against, the constant-speed initial pose built from two estimated poses.

Protocol.  ONE closed-loop HIP run over five frames (map every 2nd: three mapped, four tracked; seven
frames until round 5 -- shortened for the suite's wall time, the stages covered are the same), which RECORDS every host-side random draw (pixel
indices and fallback vectors of every tracking / mapping call, the add-pixels, the keyframe window, the N(0, 0.1^2) initial
features of the new points) and snapshots its state (cloud, both feature sets, decoder, estimated poses) in front of every
frame.  The oracle then replays every STAGE from the HIP run's own hand-over -- the state and the poses the previous stages
left -- and must arrive where the HIP run arrived: a stale pose, a wrong window, a selection made before the add, an
iteration count from the wrong `added`, a tracker rendering against yesterday's decoder would each separate the two at
the stage where it happens.  The stages are NOT chained on the oracle side, for a measured reason: a free-running oracle
copy separates from ANY other implementation within two frames (first run of this test, profiles/r04_closed_loop_free_running.json:
the colour-stage of mapped frame 0 -- decoder training, Adam steps of noise-decided sign, the drift tests/test_hip_loops.py
documents -- leaves the two decoders 2e-3 apart in loss, the tracker of frame 1 then starts 8e-4 apart and its 20 Adam
steps of +-lr, whose signs are decided by gradient components at the noise level next to the optimum, end 8 steps apart).
That is the sensitivity of the reference's own objective (yardstick below), not an orchestration error, and it would mask
one.
"""
import pytest
import torch

from tests.helpers import base_cfg, load_decoders
from tests.test_hip_parity import report

pytestmark = pytest.mark.gpu

N_FRAMES, MAP_EVERY = 5, 2
TRACK_ITERS, TRACK_PIX = 20, 200
MAP_ITERS, MAP_PIX, ADD_PIX = 20, 600, 1500


def _cfg():
    cfg = base_cfg()
    cfg["tracking"].update(iters=TRACK_ITERS, pixels=TRACK_PIX)
    cfg["mapping"].update(iters=MAP_ITERS, pixels=MAP_PIX, pixels_adding=ADD_PIX, every_frame=MAP_EVERY,
                          mapping_window_size=4, keyframe_selection_method="global")
    return cfg


def _scene(dev, n_pts=50000, W=320, H=240):
    """~50 k seeded points seen from around the trajectory (with unseeded stripes, so that every mapped frame ADDS
    points), N_FRAMES frames 1 trajectory unit (~2.8 cm, 0.2 degrees) apart."""
    from point_slam_amd import synthetic as syn
    from point_slam_amd.slam import Frame
    cfg = _cfg()
    cam = syn.intrinsics(W, H)
    frames = []
    for i in range(N_FRAMES):
        c2w = syn.pose(10.0 + 1.0 * i, dev)
        depth, color = syn.render_frame(cam, c2w)
        r_add, r_q = syn.dynamic_radii(color, cfg)
        frames.append(Frame(i, depth, color, r_add, r_q, c2w))
    pts = []
    g = torch.Generator().manual_seed(4)
    tt = torch.linspace(0.0, 1.0, 3)
    for t in (8.0, 11.0, 14.0, 17.0):
        c2w = syn.pose(t)
        u = torch.rand(n_pts // 12, generator=g) * (cam["W"] - 1)
        v = torch.rand(n_pts // 12, generator=g) * (cam["H"] - 1)
        keep = (torch.floor(u / 40) % 4) != 1                      # vertical stripes left empty
        u, v = u[keep], v[keep]
        dirs = torch.stack([(u - cam["cx"]) / cam["fx"], -(v - cam["cy"]) / cam["fy"], -torch.ones_like(u)], -1)
        rd = (dirs[:, None, :] * c2w[:3, :3]).sum(-1)
        ro = c2w[:3, 3].expand_as(rd)
        d = syn.box_depth(ro, rd)
        z = 0.98 * d[:, None] * (1 - tt) + 1.02 * d[:, None] * tt
        pts.append((ro[:, None] + rd[:, None] * z[..., None]).reshape(-1, 3))
    return cfg, cam, frames, torch.cat(pts).float()


def _c2w_from_cam(best, dev):
    from point_slam_amd import host_ops as H
    c34 = H.get_camera_from_tensor(best.to(dev))
    return torch.cat([c34, torch.tensor([[0.0, 0.0, 0.0, 1.0]], device=dev)], 0)


class _Recorder:
    """Wraps a HipSLAM so that every host-side draw is kept for the replay."""

    def __init__(self, slam):
        self.s = slam
        self.draws, self.add_idx, self.windows, self.init_feats = [], [], [], []
        orig_draws, orig_add, orig_win, orig_add_points = slam._draws, slam._add_batch, slam.select_window, slam.add_points

        def draws(n_iters, n_idx, hi):
            d = orig_draws(n_iters, n_idx, hi)
            self.draws.append((d[0].cpu().clone(), d[1].cpu().clone()))
            return d

        def add_batch(frame, c2w, idx, is_pts_grad):
            self.add_idx.append(idx.cpu().clone())
            return orig_add(frame, c2w, idx, is_pts_grad)

        def select_window(frame, c2w=None, size=None, method=None):
            w = orig_win(frame, c2w, size, method)
            self.windows.append([f.idx for f in w])
            return w

        def add_points(frame, c2w, n_pixels=None, first=False):
            # the N(0, 0.1^2) features of the new points (neural_point.py:152) are drawn on the device inside the add:
            # read them back BEFORE the mapper trains them -- a recorded draw like the others
            n0 = slam.npc.pts_num()
            added = orig_add_points(frame, c2w, n_pixels, first)
            n1 = slam.npc.pts_num()
            self.init_feats.append((slam.npc.get_geo_feats()[n0:n1].cpu().clone(), slam.npc.get_col_feats()[n0:n1].cpu().clone()))
            return added

        slam._draws, slam._add_batch, slam.select_window, slam.add_points = draws, add_batch, select_window, add_points


def test_track_map_track_closed_loop_matches_oracle():
    from oracle import pointslam_oracle as O
    from point_slam_amd import host_ops as H
    from point_slam_amd.decoders import PointDecoders
    from point_slam_amd.slam import HipSLAM, camera_tensor_from_c2w
    from tests import parity_probe as PP
    dev = torch.device("cuda:0")
    cfg, cam, frames, pts = _scene(dev)
    tr, mp = cfg["tracking"], cfg["mapping"]
    eh, ew = tr["ignore_edge_H"], tr["ignore_edge_W"]

    # ------------------------------------------------------------------------------------------------ HIP, recording
    s = HipSLAM(cfg, cam, device="cuda:0", max_points=300000, engine="native",
                decoders=PointDecoders(cfg).load_reference_state(load_decoders("replica")))
    s.seed_points(pts, seed=77)
    st0 = PP.oracle_state(s)                                         # the state both runs start from
    rec = _Recorder(s)
    torch.manual_seed(123)
    hip = []                                                         # per frame: dict of what happened
    est = []                                                         # estimated poses (device tensors)
    for i, fr in enumerate(frames):
        out = dict(idx=i, state=PP.oracle_state(s), est=[e.cpu().clone() for e in est], kf=[f.idx for f in s.keyframes])
        if i == 0:
            c2w = fr.c2w.clone()                                     # idx 0: ground-truth pose (Tracker.py:278-279)
        else:
            cam0 = s.init_pose(est).to(dev)
            out["track_draw"] = len(rec.draws)
            best = s.track(fr, cam0)
            torch.cuda.synchronize()
            out.update(cam0=cam0.cpu(), track_losses=s.last_losses[:, 0].cpu().double().clone(), best=best.cpu().clone(),
                       last_cam=s.last_cam.cpu().clone())
            c2w = _c2w_from_cam(best, dev)
        est.append(c2w)
        if i % MAP_EVERY == 0:
            n0 = s.npc.pts_num()
            n_draws0 = len(rec.draws)
            added, n_sel = s.map(fr, c2w)
            torch.cuda.synchronize()
            sel, _ = s.frustum_select(fr, c2w)                       # the selection map() trained (same inputs, pure function)
            n1 = s.npc.pts_num()
            out.update(mapped=True, added=added, n_sel=n_sel, sel=sel.cpu().long().clone(), n_pts=n1, n_new=n1 - n0,
                       map_losses=s.last_losses[:, 0].cpu().double().clone(), n_iters=int(s.last_losses.shape[0]),
                       map_draw=n_draws0)
            s.keyframes.append(fr)
        hip.append(out)
    final = dict(cloud=s.npc.cloud_pos().float(), geo=s.npc.get_geo_feats().cpu().clone(), col=s.npc.get_col_feats().cpu().clone())

    # ------------------------------------------------------------------------------------------------ oracle, stage by stage
    O.KNN_WORKERS = 8
    est_hip = [e.cpu() for e in est]                                 # the poses the HIP run estimated, frame by frame
    oframes = [dict(depth=f.depth.cpu(), color=f.color.cpu(), r_query=f.r_query.cpu(), r_add=f.r_add.cpu(), idx=f.idx,
                    c2w=est_hip[f.idx]) for f in frames]             # keyframes carry their ESTIMATED pose (Mapper.py:755-760)
    add_i, win_i = 0, 0
    step = tr["lr"]
    per_frame = []
    for i, of in enumerate(oframes):
        h = hip[i]
        st = h["state"]
        P, cloud, geo, col = st["P"], st["cloud"], st["geo"], st["col"]
        row = dict(idx=i, n_pts_before=int(cloud.shape[0]))
        if i > 0:
            # ---- tracker stage: initial pose from the two poses the previous stages handed over (Tracker.py:283-290)
            c0 = H.const_speed_init(h["est"][-1], h["est"][-2] if len(h["est"]) >= 2 else None)
            cam0 = camera_tensor_from_c2w(c0)
            pix, fb = rec.draws[h["track_draw"]]
            kw = dict(coef=cfg["rendering"]["sigmoid_coef_tracker"])
            ls, cams, best, _, _ = O.tracker_loop(cfg, P, cloud, geo, col, cam0, pix, fb, of["depth"], of["color"], of["r_query"],
                                                  cam, eh, ew, **kw)
            # yardstick (tests/test_hip_slam.py:78): the same oracle loop from an initial pose moved by ONE ulp
            noise = 0.0
            for target in (10.0,):
                cam0_ulp = torch.nextafter(cam0, torch.full_like(cam0, target))
                _, _, best_ulp, _, _ = O.tracker_loop(cfg, P, cloud, geo, col, cam0_ulp, pix, fb, of["depth"], of["color"], of["r_query"],
                                                      cam, eh, ew, **kw)
                noise = max(noise, float((best_ulp - best).abs().max()))
            ref = torch.tensor(ls, dtype=torch.float64)
            rel = (h["track_losses"] - ref).abs() / ref.abs()
            row.update(cam0_abs=float((h["cam0"] - cam0).abs().max()), track_loss_rel_first=float(rel[0]),
                       track_loss_rel_first5=float(rel[:5].max()), track_loss_rel_max=float(rel.max()),
                       pose_abs=float((h["best"] - best).abs().max()), oracle_self_noise_1ulp=noise)
        if h.get("mapped"):
            c2w = est_hip[i]                                         # the hand-over: what the tracker stage returned for this frame
            window_ids = rec.windows[win_i]; win_i += 1
            # ---- point adding at that pose (Mapper.py:303-330; uniform batch only: pixels_based_on_color_grad = 0)
            idx = rec.add_idx[add_i]; init_geo, init_col = rec.init_feats[add_i]; add_i += 1
            u, v = O.pixels_from_flat_index(idx.long(), 0, cam["H"], 0, cam["W"])
            ro, rd = O.rays_from_uv(u, v, c2w, cam["fx"], cam["fy"], cam["cx"], cam["cy"])
            gd = of["depth"][v.long(), u.long()]
            new_pts, keep, _ = O.add_points_select(cloud, ro, rd, gd, of["r_add"][v.long(), u.long()])
            added_o = int(keep.sum())
            n0 = int(cloud.shape[0])
            row.update(added=h["added"], added_oracle=added_o, n_pts=h["n_pts"], n_pts_oracle=n0 + int(new_pts.shape[0]))
            assert h["n_new"] == int(new_pts.shape[0]) == 3 * h["added"], row
            row["new_pts_abs"] = float((final["cloud"][n0:n0 + new_pts.shape[0]] - new_pts).abs().max()) if new_pts.shape[0] else 0.0
            cloud = torch.cat([cloud, new_pts])
            geo, col = torch.cat([geo, init_geo]), torch.cat([col, init_col])
            # ---- iteration count (Mapper.py:404-406), frustum rows (:120-168), window (:263-276) from the oracle's own numbers
            lo = int(mp.get("min_iter_ratio", 0.95) * mp["iters"])
            n_iters = int(min(max(int(mp["iters"] * added_o / 300), lo), 2 * mp["iters"]))
            n_geo = int(n_iters * mp["geo_iter_ratio"])
            sel_o = O.frustum_select(cloud, c2w, of["depth"], cam["H"], cam["W"], cam["fx"], cam["fy"], cam["cx"], cam["cy"],
                                     float(mp["frustum_edge"]))
            a, b = set(h["sel"].tolist()), set(sel_o.tolist())
            row.update(n_iters=h["n_iters"], n_iters_oracle=n_iters, n_sel=len(a), n_sel_oracle=len(b), sel_sym_diff=len(a ^ b),
                       window=window_ids)
            assert h["n_iters"] == n_iters, row
            # the window: the last keyframe and the current frame close it, the others are earlier keyframes (:263-276)
            assert window_ids[-1] == i and (not h["kf"] or window_ids[-2] == h["kf"][-1]) and set(window_ids[:-1]) <= set(h["kf"]), row
            # ---- the joint iterations over the recorded window
            win = [oframes[k] for k in window_ids]
            pix, fb = rec.draws[h["map_draw"]]
            ppf = mp["pixels"] // len(win)
            ls, geo_o, col_o, P_o, _, _ = O.mapper_iterations(cfg, P, cloud, geo, col, sel_o, win, pix.reshape(n_iters, len(win), ppf), fb,
                                                              n_geo, cam, coef=cfg["rendering"]["sigmoid_coef_mapper"])
            ref = torch.tensor(ls, dtype=torch.float64)
            rel = (h["map_losses"] - ref).abs() / ref.abs()
            # where the HIP run arrived: the state snapshot in front of the NEXT frame (or the final state)
            nxt = hip[i + 1]["state"] if i + 1 < len(hip) else dict(geo=final["geo"], col=final["col"])
            so = sel_o.long()
            row.update(map_loss_rel_first=float(rel[0]), map_loss_rel_geo_stage=float(rel[:n_geo + 1].max()),
                       map_loss_rel_first_colour=float(rel[n_geo + 1]), map_loss_rel_max=float(rel.max()),
                       rows_geo_mean=float((nxt["geo"][so] - geo_o[so]).abs().mean()), rows_col_mean=float((nxt["col"][so] - col_o[so]).abs().mean()))
        per_frame.append(row)
    report(test="closed_loop_track_map_track", frames=per_frame, final_pts=int(final["cloud"].shape[0]), pose_step=step)
    mapped = [r for r in per_frame if "n_sel" in r]
    assert len(mapped) == (N_FRAMES + MAP_EVERY - 1) // MAP_EVERY
    for r in mapped:
        assert r["added"] == r["added_oracle"] and r["n_pts"] == r["n_pts_oracle"], r
        assert r["added"] > 0, r                                      # every mapped frame grows the map: the add path is in the loop
        assert r["new_pts_abs"] <= 2e-6, r                            # the new points sit where the handed-over pose puts them
        assert r["sel_sym_diff"] <= max(2, r["n_sel"] // 10000), r    # frustum rows: the same set (border flips: see report)
        assert r["map_loss_rel_first"] <= 1e-4, r                     # BASELINE.json: render-loss rel-err <= 1e-4
        assert r["map_loss_rel_geo_stage"] <= 1e-4, r                 # the whole geometry stage (decoders frozen: not chaotic)
        assert r["map_loss_rel_first_colour"] <= 1e-4, r
        # where the stage ARRIVED: the rows after 19-40 iterations (38-80 when first measured), the colour stage among them (decoder training: the drift the
        # 140-iteration test documents; measured 4e-6 .. 1.9e-3 mean over four runs).  Training the wrong rows, or against the
        # wrong window / decoder, leaves O(0.1).
        assert r["rows_geo_mean"] <= 1e-2 and r["rows_col_mean"] <= 1e-2, r
    for r in per_frame[1:]:
        assert r["cam0_abs"] <= 2e-6, r                               # constant-speed init from the two handed-over poses
        assert r["track_loss_rel_first"] <= 1e-4, r                   # the tracker renders against the map the mapper left
        assert r["pose_abs"] <= 20 * step, r                          # hard cap: 20 Adam steps cannot carry a correct loop further apart
    # The lowest-loss pose of 20 Adam steps is NOT a parity quantity: next to the optimum the signs of Adam's +-lr steps are decided by
    # gradient components at the rounding level, and the oracle separates from ITSELF by 1e-5 .. 1.3e-2 (6 steps) when its
    # initial pose moves by one ulp (first measured run: HIP-vs-oracle 1e-4 .. 1.0e-2 on the same frames).  What can be
    # asserted is that the two spreads are of one size -- over the tracked frames, against as many perturbed oracle runs:
    worst_pose = max(r["pose_abs"] for r in per_frame[1:])
    worst_noise = max(r["oracle_self_noise_1ulp"] for r in per_frame[1:])
    assert worst_pose <= max(0.25 * step, 4 * worst_noise), (worst_pose, worst_noise)
