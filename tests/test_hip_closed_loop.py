"""Closed-loop parity: track -> map -> track over several frames, once through HipSLAM(engine="native") and once through
the pinned oracle COMPOSED the same way, with identical random draws.

Every loop of the path is pinned in isolation elsewhere (tests/test_hip_loops.py: the reference's own optimize_map /
tracker loop / add_neural_points fixtures).  What nothing else checks is the ORCHESTRATION between them
(src/Tracker.py:259-270,379-380 <-> src/Mapper.py:263-330,404-406,642-783): the pose a tracked frame hands to the mapper, the
points the mapper adds at THAT pose, the frustum rows it then selects, the data-dependent iteration count, the trained
rows and decoder the next tracked frame renders against, the constant-speed initial pose built from two estimated poses.
An error in any hand-over would show up here and in no other test.

Protocol.  The HIP run goes first and RECORDS every host-side random draw (pixel indices and fallback vectors of every
tracking / mapping call, the add-pixels, the keyframe window, the N(0, 0.1^2) initial features of the new points).  The
oracle run replays them: O.tracker_loop -> O.add_points_select -> O.frustum_select -> O.mapper_iterations, carrying ITS OWN
state (cloud, features, decoder, poses) from frame to frame -- the two runs share inputs and draws, nothing else.
"""
import pytest
import torch

from tests.helpers import base_cfg, load_decoders
from tests.test_hip_parity import report

pytestmark = pytest.mark.gpu

N_FRAMES, MAP_EVERY = 7, 2
TRACK_ITERS, TRACK_PIX = 20, 200
MAP_ITERS, MAP_PIX, ADD_PIX = 40, 600, 1500


def _cfg():
    cfg = base_cfg()
    cfg["tracking"].update(iters=TRACK_ITERS, pixels=TRACK_PIX)
    cfg["mapping"].update(iters=MAP_ITERS, pixels=MAP_PIX, pixels_adding=ADD_PIX, every_frame=MAP_EVERY,
                          mapping_window_size=4, keyframe_selection_method="global")
    return cfg


def _scene(dev, n_pts=50000, W=320, H=240):
    """~50 k seeded points seen from around the trajectory (with unseeded stripes, so that every mapped frame ADDS
    points), seven frames 1 trajectory unit (~2.8 cm, 0.2 degrees) apart."""
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
        out = dict(idx=i)
        if i == 0:
            c2w = fr.c2w.clone()                                     # idx 0: ground-truth pose (Tracker.py:254-255)
        else:
            cam0 = s.init_pose(est).to(dev)
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

    # ------------------------------------------------------------------------------------------------ oracle, replaying
    O.KNN_WORKERS = 8
    P, cloud, geo, col = st0["P"], st0["cloud"], st0["geo"], st0["col"]
    oframes = [dict(depth=f.depth.cpu(), color=f.color.cpu(), r_query=f.r_query.cpu(), r_add=f.r_add.cpu(), c2w=None, idx=f.idx)
               for f in frames]
    est_o = []
    draw_i, add_i, win_i = 0, 0, 0
    worst = dict(track_first=0.0, track_all=0.0, map_first=0.0, map_all=0.0, pose=0.0, sel_diff=0, n_pts_diff=0)
    per_frame = []
    for i, of in enumerate(oframes):
        h = hip[i]
        row = dict(idx=i)
        if i == 0:
            c2w = frames[0].c2w.cpu().clone()
        else:
            c0 = H.const_speed_init(est_o[-1], est_o[-2] if len(est_o) >= 2 else None)
            cam0 = camera_tensor_from_c2w(c0)
            pix, fb = rec.draws[draw_i]; draw_i += 1
            ls, cams, best, _, _ = O.tracker_loop(cfg, P, cloud, geo, col, cam0, pix, fb, of["depth"], of["color"], of["r_query"],
                                                  cam, eh, ew, coef=cfg["rendering"]["sigmoid_coef_tracker"])
            ref = torch.tensor(ls, dtype=torch.float64)
            rel = (h["track_losses"] - ref).abs() / ref.abs()
            dpose = float((h["best"] - best).abs().max())
            row.update(track_loss_rel_first=float(rel[0]), track_loss_rel_max=float(rel.max()), pose_abs=dpose,
                       cam0_abs=float((h["cam0"] - cam0).abs().max()))
            worst["track_first"] = max(worst["track_first"], float(rel[0])); worst["track_all"] = max(worst["track_all"], float(rel.max()))
            worst["pose"] = max(worst["pose"], dpose)
            c34 = H.get_camera_from_tensor(best)
            c2w = torch.cat([c34, torch.tensor([[0.0, 0.0, 0.0, 1.0]])], 0)
        est_o.append(c2w)
        of["c2w"] = c2w
        if i % MAP_EVERY == 0:
            window_ids = rec.windows[win_i]; win_i += 1
            # ---- point adding at the ORACLE's pose (Mapper.py:303-330; uniform batch only: pixels_based_on_color_grad = 0)
            idx = rec.add_idx[add_i]; add_i += 1
            u, v = O.pixels_from_flat_index(idx.long(), 0, cam["H"], 0, cam["W"])
            ro, rd = O.rays_from_uv(u, v, c2w, cam["fx"], cam["fy"], cam["cx"], cam["cy"])
            gd = of["depth"][v.long(), u.long()]
            new_pts, keep, _ = O.add_points_select(cloud, ro, rd, gd, of["r_add"][v.long(), u.long()])
            added_o = int(keep.sum())
            init_geo, init_col = rec.init_feats[add_i - 1]
            row.update(added=h["added"], added_oracle=added_o, n_pts=h["n_pts"], n_pts_oracle=int(cloud.shape[0]) + int(new_pts.shape[0]))
            worst["n_pts_diff"] = max(worst["n_pts_diff"], abs(h["n_new"] - int(new_pts.shape[0])))
            if h["n_new"] != int(new_pts.shape[0]):
                per_frame.append(row)
                break                                                # a different number of points: the runs have separated
            cloud = torch.cat([cloud, new_pts])
            geo, col = torch.cat([geo, init_geo]), torch.cat([col, init_col])
            row["new_pts_abs"] = float((final["cloud"][cloud.shape[0] - new_pts.shape[0]:cloud.shape[0]] - new_pts).abs().max()) if new_pts.shape[0] else 0.0
            # ---- iteration count (Mapper.py:404-406) and frustum rows (:120-168) from the oracle's own numbers
            lo = int(mp.get("min_iter_ratio", 0.95) * mp["iters"])
            n_iters = int(min(max(int(mp["iters"] * added_o / 300), lo), 2 * mp["iters"]))
            n_geo = int(n_iters * mp["geo_iter_ratio"])
            sel_o = O.frustum_select(cloud, c2w, of["depth"], cam["H"], cam["W"], cam["fx"], cam["fy"], cam["cx"], cam["cy"],
                                     float(mp["frustum_edge"]))
            a, b = set(h["sel"].tolist()), set(sel_o.tolist())
            row.update(n_iters=h["n_iters"], n_iters_oracle=n_iters, n_sel=len(a), n_sel_oracle=len(b), sel_sym_diff=len(a ^ b))
            worst["sel_diff"] = max(worst["sel_diff"], len(a ^ b))
            if h["n_iters"] != n_iters:
                per_frame.append(row)
                break
            # ---- the joint iterations over the recorded window
            win = [oframes[k] for k in window_ids]
            pix, fb = rec.draws[h["map_draw"]]; draw_i = h["map_draw"] + 1
            ppf = mp["pixels"] // len(win)
            ls, geo, col, P, _, _ = O.mapper_iterations(cfg, P, cloud, geo, col, sel_o, win, pix.reshape(n_iters, len(win), ppf), fb,
                                                        n_geo, cam, coef=cfg["rendering"]["sigmoid_coef_mapper"])
            ref = torch.tensor(ls, dtype=torch.float64)
            rel = (h["map_losses"] - ref).abs() / ref.abs()
            row.update(map_loss_rel_first=float(rel[0]), map_loss_rel_geo_stage=float(rel[:n_geo + 1].max()), map_loss_rel_max=float(rel.max()))
            worst["map_first"] = max(worst["map_first"], float(rel[0])); worst["map_all"] = max(worst["map_all"], float(rel.max()))
        per_frame.append(row)
    # ------------------------------------------------------------------------------------------------ the end states
    n = min(cloud.shape[0], final["cloud"].shape[0])
    d_geo, d_col = (final["geo"][:n] - geo[:n]).abs(), (final["col"][:n] - col[:n]).abs()
    s.sync_decoders_from_theta()
    P_hip = {k: v.detach().cpu() for k, v in s.decoders.state_dict().items()}
    d_dec = max(float((P_hip[k] - P[k]).abs().max()) for k in P_hip if k.startswith("color_decoder") and k in P and P[k].dtype.is_floating_point)
    # yardstick (SURVEY 7 / tests/test_hip_slam.py:78): the sensitivity of the tracker's own objective -- one Adam step of the
    # pose is lr (0.002 translation, 0.0004 quaternion); two correct implementations agree to a small fraction of ONE step
    step = tr["lr"]
    report(test="closed_loop_track_map_track", frames=per_frame, **{"worst_" + k: v for k, v in worst.items()},
           final_pts=int(final["cloud"].shape[0]), final_pts_oracle=int(cloud.shape[0]), rows_geo_max=float(d_geo.max()),
           rows_geo_mean=float(d_geo.mean()), rows_col_max=float(d_col.max()), rows_col_mean=float(d_col.mean()), decoder_max=d_dec,
           pose_step=step)
    assert len(per_frame) == N_FRAMES, per_frame[-1]                  # no structural separation (point / iteration counts)
    mapped = [r for r in per_frame if "n_sel" in r]
    assert len(mapped) == (N_FRAMES + MAP_EVERY - 1) // MAP_EVERY
    for r in mapped:
        assert r["added"] == r["added_oracle"] and r["n_pts"] == r["n_pts_oracle"] and r["n_iters"] == r["n_iters_oracle"], r
        assert r["added"] > 0, r                                      # every mapped frame grows the map: the add path is in the loop
        assert r["new_pts_abs"] <= 2e-5, r                            # new points sit where the oracle puts them (poses agree to ~1e-6)
        # frustum rows: identical sets.  A point ON the frustum border or the depth band can flip with the ~1e-6 pose
        # difference of the two runs; allow two such rows in ~3e4 (measured: see the report)
        assert r["sel_sym_diff"] <= 2, r
        assert r["map_loss_rel_first"] <= 1e-4, r                     # BASELINE.json: render-loss rel-err <= 1e-4
    for r in per_frame[1:]:
        assert r["track_loss_rel_first"] <= 1e-4, r
        assert r["cam0_abs"] <= 0.05 * step, r                        # constant-speed init from two ESTIMATED poses
        assert r["pose_abs"] <= 0.25 * step, r                        # best pose after 20 Adam steps: a fraction of one step
    # trained rows of the whole run (4 mapped frames x ~40 iterations, rows trained up to 4 times)
    assert float(d_geo.mean()) <= 2e-5 and float(d_col.mean()) <= 2e-5
    assert final["cloud"].shape[0] == cloud.shape[0]
