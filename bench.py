#!/usr/bin/env python
"""Headline benchmark: mapping+tracking FPS @640x480 with 1 M neural points (BASELINE.json).

A "step" is one RGB-D frame of the serial SLAM loop on synthetic data: 20 tracking iterations x 200 pixels
every frame and, every 5th frame, point adding (6000 pixels) + frustum feature selection + 400 mapping
iterations x 1000 pixels over a 5-frame window -- the iteration mix of the reference's base config
(configs/point_slam.yaml:35-36,42,60,64).  All inputs (frames, point cloud, decoders) are resident in HBM
before the timed region.  N>1: frame-parallel, one process per GPU (frame t on rank t mod N), each rank a full
map replica, RCCL all-gather of newly added neural points every `--exchange-every` mapped frames.

Prints ONE JSON line (rank 0).  `roofline` is computed from HIP-event timings of the dominant kernel class
recorded on the launch stream during the timed region; `cpu_baseline` times the CPU oracle (oracle/, a port of
the reference path) on a bounded sample of the same workload.
"""
from __future__ import annotations

import argparse
import json
import math
import os
import sys
import time

ROOT = os.path.dirname(os.path.abspath(__file__))
if ROOT not in sys.path:
    sys.path.insert(0, ROOT)

PEAK_F32_MFMA_TFLOPS = 157.3      # /opt/skills/guides/MI355X_MICROARCH.md: f32-in MFMA dense peak
PEAK_HBM_GBS = 8000.0             # HBM3E spec peak


def parse():
    ap = argparse.ArgumentParser()
    ap.add_argument("--gpus", type=int, default=1)
    ap.add_argument("--steps", type=int, default=20)
    ap.add_argument("--warmup", type=int, default=5)
    ap.add_argument("--points", type=int, default=1_000_000)
    ap.add_argument("--engine", default="native", choices=["native", "dropin"])
    ap.add_argument("--mix", default="base", choices=["base", "replica", "tum", "scannet"],
                    help="iteration mix of the reference config: base = configs/point_slam.yaml (BASELINE headline), "
                         "replica / tum / scannet = the dataset yaml on top of it (BASELINE configs 2-4)")
    ap.add_argument("--saturated-map", action="store_true",
                    help="seed the cloud without holes: (almost) no point growth during the run, as in round 1")
    ap.add_argument("--width", type=int, default=640)
    ap.add_argument("--height", type=int, default=480)
    ap.add_argument("--exchange-every", type=int, default=None,
                    help="mapped frames between exchanges (N>1); default: derived from --exchange-every-keyframes")
    ap.add_argument("--exchange-every-keyframes", type=int, default=10,
                    help="BASELINE config 4: 'RCCL neural-point all-gather every 10 keyframes' -> one exchange every "
                         "10 * keyframe_every / every_frame mapped frames of a rank; a run shorter than that still performs ONE "
                         "exchange inside the timed region (at its last mapped frame), so that its cost is in `value`")
    ap.add_argument("--track-only", action="store_true",
                    help="BASELINE config 1: tracking only on a FIXED cloud (no mapping, no point growth); every frame starts "
                         "from the constant-speed extrapolation of the tracker's own previous estimates (Tracker.py:283-290). "
                         "Quoted with --points 50000 --width 1200 --height 680 --mix replica --steps 200")
    ap.add_argument("--depth-noise", type=float, default=None,
                    help="multiplicative Gaussian sensor-depth noise (sigma as a fraction of the depth); default: 0.005 for --mix tum "
                         "(SURVEY.md 8d: the TUM-like stream, 'sigma = 0.5 % d + 2 % dropout'), 0 otherwise")
    ap.add_argument("--depth-dropout", type=float, default=None,
                    help="fraction of pixels whose sensor depth is dropped to 0 (holes); default: 0.02 for --mix tum, 0 otherwise")
    ap.add_argument("--closed-loop", action="store_true",
                    help="frame i starts from the constant-speed extrapolation of the tracker's OWN two previous estimates "
                         "(Tracker.py:283-290, psl_pose_const_speed on the device), the mapper maps at the tracker's estimate; "
                         "config.ate_rmse_cm in the line.  One GPU.  Default: open loop (ground-truth pose + a perturbation of the size "
                         "the constant-speed model leaves) -- measured in round 5, the base yaml's tracker budget (200 px x 20 it, "
                         "<= 4 cm of correction per frame) does not hold this stream's 5.6 cm / 0.4 deg per frame in closed loop (ATE "
                         "76 cm after 40 frames, the run then maps junk), the Replica yaml's does (0.5 cm): see DESIGN.md")
    ap.add_argument("--units-per-frame", type=float, default=None,
                    help="trajectory units the camera advances per frame (default 2.0 = ~5.6 cm / 0.4 deg, SURVEY.md 8d; 0.5 = ~1.4 cm, "
                         "a Replica-like camera speed; --track-only uses 0.5)")
    ap.add_argument("--pretrain-keyframes", type=int, default=8,
                    help="closed loop: keyframes mapped at their true poses before the run starts (untimed set-up), one every "
                         "mapping.every_frame frames back along the trajectory -- the map a run that reached the first frame would "
                         "hold.  (Open loop: the four earlier keyframes of rounds 1-4, points added, features untrained.)")
    ap.add_argument("--different-frames", action="store_true",
                    help="rounds 1-4: the event-timed pass runs on the NEXT --steps frames.  Default since round 5: the map, the "
                         "poses and the RNGs are snapshotted after the warm-up and restored, so that `value` (plain pass) and the "
                         "kernel classes (event-carrying pass) are measured on the SAME frames")
    ap.add_argument("--expandable-segments", action="store_true",
                    help="N > 1: PYTORCH_HIP_ALLOC_CONF=expandable_segments:True for every rank (removes the rare 60-75 ms exchange "
                         "stalls measured on one GPU, profiles/r04_exchange_timing_*; not the default because it has never run next "
                         "to RCCL across GPUs and the first multi-GPU run should not depend on it)")
    ap.add_argument("--merge", default="mean", choices=["mean", "owner"],
                    help="N > 1: rule for feature rows several ranks trained (point_slam_amd/dist.py)")
    ap.add_argument("--fps-window", type=int, default=0,
                    help="long runs: frames/s of every window of this many frames of the timed pass (one device synchronisation per "
                         "window), next to the device-memory high-water mark -- the steady-state evidence behind the 20-frame headline")
    ap.add_argument("--no-sub-records", action="store_true",
                    help="skip config.closed_loop / config.steady_state: two further runs of this script (100 closed-loop frames at "
                         "the camera speed the base tracker holds; 500 open-loop frames with the frames/s of every 100-frame window), "
                         "made after the timed passes so that the one line the driver keeps carries them")
    ap.add_argument("--no-cpu-baseline", action="store_true")
    ap.add_argument("--no-kernel-timing", action="store_true")
    ap.add_argument("--event-stride", type=int, default=1,
                    help="kernel-class timing on one launch in N of every class (1 = every launch, what the rocprofv3 "
                         "trace of the same command averages over)")
    return ap.parse_args()


def build_world(args, rank, world, dev):
    import torch
    from point_slam_amd import synthetic as syn
    from point_slam_amd.config import MIXES, default_config
    from point_slam_amd.slam import Frame, HipSLAM, camera_tensor_from_c2w
    cfg = MIXES[args.mix](default_config())
    for k_, v_ in dict(closed_loop=False, depth_noise=None, depth_dropout=None, pretrain_keyframes=8, different_frames=False,
                       units_per_frame=None, no_kernel_timing=False, track_only=False).items():
        if not hasattr(args, k_):       # the measurement tools call build_world with a hand-made namespace
            setattr(args, k_, v_)
    args.open_loop = not args.closed_loop
    cam = syn.intrinsics(args.width, args.height)
    torch.manual_seed(cfg["setup_seed"] + rank)
    slam = HipSLAM(cfg, cam, device=str(dev), max_points=int(args.points * 1.3) + 300_000, engine=args.engine)
    # one cube in four of a 25 cm checker is left unseeded: every mapped frame still finds uncovered surface and ADDS
    # points (thousands at first, fewer as the holes fill), like a real sequence; --saturated-map restores round 1
    track_only = getattr(args, "track_only", False)
    if track_only:
        # the fixed cloud of config 1: all of it around the stretch of the trajectory the run tracks (t = 200 + 2 i)
        # 0.5 trajectory units per frame (~1.4 cm, 0.1 degrees: a Replica-like camera speed; 50 k points cover the stretch
        # 405 frames see about as densely as the headline's 1 M cover the whole room)
        n_fr = args.warmup + 2 * args.steps
        args._unit_per_frame = 0.5
        pts = syn.seed_cloud(cam, args.points, n_views=48, seed=cfg["setup_seed"], t0=190.0, dt=(0.5 * n_fr + 20.0) / 47.0)
        args._train_span = 0.5 * n_fr + 10.0
    else:
        pts = syn.seed_cloud(cam, args.points, n_views=64, seed=cfg["setup_seed"], holes=not args.saturated_map)
    slam.seed_points(pts)
    if track_only:
        # A fixed cloud with RANDOM features localises nothing: tracked in closed loop the pose runs away within ten frames
        # (tools/track_probe.py --train-frames 0: 1 m after 10 frames, 5.7 m after 30) and the tracker's rays leave the map, which
        # is not config 1's workload.  Untimed set-up therefore: the features (and the colour decoder) are TRAINED by mapping a
        # keyframe every 5 trajectory units at its true pose, no point adding (the cloud stays as seeded); afterwards the closed
        # loop holds the trajectory to millimetres (same probe: 0.05-0.41 cm over 60 frames; this run: config.ate_rmse_cm).
        n_train = int(args._train_span / 5.0) + 2
        for k in range(n_train):
            c2w = syn.pose(195.0 + 5.0 * k, dev)
            depth, color = syn.render_frame(cam, c2w)
            r_add, r_q = syn.dynamic_radii(color, cfg)
            kf = Frame(-1 - k, depth, color, r_add, r_q, c2w)
            slam.map(kf, c2w, n_iters=cfg["mapping"]["iters"], add=False, fixed_iters=True)
            slam.keyframes.append(kf)
            if len(slam.keyframes) > cfg["mapping"]["mapping_window_size"]:
                slam.keyframes.pop(0)
        torch.cuda.synchronize()
        args._trained_keyframes = n_train
    every = cfg["mapping"]["every_frame"]
    # BASELINE config 3 ("TUM fr1_desk ... noisy-depth path"): the TUM-like stream of SURVEY.md 8d -- depth noise of 0.5 % of the
    # depth and 2 % of the pixels dropped to 0 -- so that the timed loop runs the sensor-hole filters (depth > 0 compaction of
    # the tracker / mapper batches, the add step's depth filter, the frustum selection's hole rule) and the depth-outlier mask
    noise = args.depth_noise if args.depth_noise is not None else (0.005 if args.mix == "tum" else 0.0)
    dropout = args.depth_dropout if args.depth_dropout is not None else (0.02 if args.mix == "tum" else 0.0)
    args._depth_noise, args._depth_dropout = noise, dropout
    g_noise = torch.Generator(device=dev).manual_seed(4242 + rank)
    n_total = args.warmup + args.steps * (2 if (not args.no_kernel_timing and (args.different_frames or world > 1)) else 1)
    # frame-parallel partition: local step i is global frame rank + world*i
    frames, cams0 = [], []
    g = torch.Generator().manual_seed(1000 + rank)
    upf = args.units_per_frame if args.units_per_frame is not None else getattr(args, "_unit_per_frame", 2.0)
    args.open_loop = not args.closed_loop
    n_early = 4 if (args.open_loop or track_only) else max(args.pretrain_keyframes, 2)
    for i in range(-n_early, n_total):
        # 2 trajectory units per frame = ~5.6 cm and ~0.4 degrees ("5 cm / 1 deg per frame"): every mapped frame sees new surface
        if i >= 0:
            t = 200.0 + float(rank + world * i) * upf
        elif args.open_loop or track_only:
            t = 170.0 + float(-3 * (i + 5))                  # rounds 1-4: four earlier keyframes ~1 m back
        else:
            t = 200.0 + float(i) * upf * every * world       # closed loop: the frames a run would have mapped on its way here
        c2w = syn.pose(t, dev)
        depth, color = syn.render_frame(cam, c2w, noise=noise, dropout=dropout, gen=g_noise)
        r_add, r_q = syn.dynamic_radii(color, cfg)
        fr = Frame(i, depth, color, r_add, r_q, c2w)
        if i < 0:
            if track_only:
                continue                                             # fixed cloud, no keyframes
            # earlier keyframes: their views were MAPPED when they were taken, i.e. points were added where they saw
            # uncovered surface (untimed set-up; otherwise half of their samples would query empty space for ever)
            if args.open_loop:
                slam.add_points(fr, c2w)
            else:
                # closed loop: ... and their features TRAINED, as a run that reached this frame would have left them (a
                # tracker that starts on random features has nothing to localise against and the loop runs away)
                slam.map(fr, c2w)
            slam.keyframes.append(fr)
        else:
            frames.append(fr)
            fr.gt_c2w = c2w
            fr.gt_cam = camera_tensor_from_c2w(c2w).to(dev)
            # open loop: initial pose = ground truth + a perturbation of the size the constant-speed model leaves
            cam0 = camera_tensor_from_c2w(c2w) + torch.randn(7, generator=g) * torch.tensor([1e-3] * 4 + [5e-3] * 3)
            cams0.append(cam0.to(dev))
    if world > 1 and not args.open_loop:
        # the set-up trained the early keyframes on every rank with the same draws, but float atomics make the trained rows
        # differ in their last bits from rank to rank: the replicas START identical (rank 0's), as the exchange keeps them
        import torch.distributed as dist
        N = slam.npc.pts_num()
        n_all = torch.tensor([N], device=dev)
        dist.all_reduce(n_all, op=dist.ReduceOp.MAX)
        if int(n_all.item()) != N:
            raise SystemExit(f"rank {rank}: {N} points after the set-up, another rank has {int(n_all.item())}")
        for t_ in (slam.npc.get_geo_feats(), slam.npc.get_col_feats(), slam.theta):
            dist.broadcast(t_, src=0)
        if slam.encode_exposure:
            dist.broadcast(slam.exposure_mlp, src=0); dist.broadcast(slam.exposure_feat, src=0)
        torch.cuda.synchronize()
    return cfg, cam, slam, frames, cams0, every


def run_step(i, slam, frames, cams0, every, cfg, world, args, state):
    import torch
    from point_slam_amd import host_ops as H
    fr = frames[i]
    if cfg["use_dynamic_radius"]:
        # the per-frame radius maps are part of the reference's frame loop (Tracker.py:235-250): one kernel here
        from point_slam_amd import frame_ops
        fr.r_add, fr.r_query = frame_ops.dynamic_radius_maps(fr.color, cfg)
    if getattr(args, "track_only", False):
        # closed loop on the tracker's own estimates, no host copy of a pose anywhere: frames 0 and 1 take the ground truth
        # (Tracker.py:278-279), frame i >= 2 starts from delta @ pre_c2w, delta = pre_c2w @ inv(c2w[i-2]) (:283-288) --
        # psl_pose_const_speed, camera tensors in and out
        hist = state.setdefault("cam_hist", [])
        if len(hist) < 2:
            hist.append(fr.gt_cam.clone())
            return
        best = slam.track(fr, slam.init_pose_device(hist[-1], hist[-2]))
        hist.append(best.clone())
        hist.pop(0)
        state.setdefault("traj", []).append((i, hist[-1][4:7]))
        return
    if args.open_loop:
        cam0 = cams0[i]
    else:
        # closed loop (Tracker.py:278-290): the first two frames of a rank start from the ground truth, every later one from the
        # constant-speed extrapolation of the tracker's OWN two previous estimates -- one launch on the device, no pose copy
        hist = state.setdefault("cam_hist", [])
        cam0 = fr.gt_cam if len(hist) < 2 else slam.init_pose_device(hist[-1], hist[-2])
    best = slam.track(fr, cam0)
    if not args.open_loop:
        hist.append(best.clone())
        if len(hist) > 2:
            hist.pop(0)
    state.setdefault("traj", []).append((i, best[4:7].clone()))      # translation of the pose the tracker settled on
    if i % every == 0:
        c2w34 = H.get_camera_from_tensor(best)
        if "row4" not in state:
            state["row4"] = torch.tensor([[0.0, 0.0, 0.0, 1.0]], device=c2w34.device)
        c2w = torch.cat([c2w34, state["row4"]], 0)
        n_base = slam.npc.pts_num()
        slam.map(fr, c2w)
        state["mapped"] += 1
        state["added"] += slam.npc.pts_num() - n_base
        state.setdefault("map_log", []).append(dict(slam.last_map, frame=i))
        if world > 1 and (state["mapped"] % state["exchange_every"] == 0 or i in state.get("force_exchange_at_all", ())):
            # new points (cross-rank dedupe), features of shared rows and the colour decoder are reconciled
            # timed by two events on the stream, read after the pass: no device synchronisation around the exchange (the host
            # goes on enqueueing the next frame's tracking while the collectives run).  The exchange itself
            # still synchronises where it must learn the sizes of the ragged blocks (dist.py).
            ev = (torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True))
            ev[0].record()
            state["sync"].exchange(slam.npc, slam.theta)
            ev[1].record()
            state.setdefault("exchange_ev", []).append(ev)
            state.setdefault("exchange_host_ms", []).append(dict(getattr(state["sync"], "last_host_ms", {})))   # host time per phase
        if (i // every) % max(cfg["mapping"]["keyframe_every"] // every, 1) == 0:
            slam.keyframes.append(fr)
            if len(slam.keyframes) > 40:        # the reference keeps every keyframe on the CPU; bounded here
                slam.keyframes.pop(0)


def kernel_profile(slam):
    import ctypes as C
    from point_slam_amd import _lib
    L = _lib.lib()
    n = L.psl_profile_classes()
    ms = (C.c_double * n)(); cnt = (C.c_int * n)(); work = (C.c_double * n)()
    _lib.check(L.psl_profile_read(slam.npc.handle, ms, cnt, work, n))
    return {L.psl_profile_name(i).decode(): dict(ms=ms[i], launches=cnt[i], work=work[i]) for i in range(n)}


MFMA_CLASSES = ("decode_fwd", "decode_bwd", "dw_gemm", "decode_fwd_geo", "decode_bwd_geo", "decode_fwd_track",
                "decode_bwd_track", "geo_iter")


def pmc_traffic(mix):
    """HBM bytes per launch of each kernel class from the PMC pass of the same command (tools/pmc_traffic.sh: separate
    rocprofv3 --pmc runs for FETCH_SIZE and WRITE_SIZE, gfx950 correction of MI355X_MICROARCH.md applied there);
    rocprofv3 cannot run inside the timed process, so the figures are read from the committed summary."""
    for rnd in ("r06", "r05", "r04", "r03", "r02"):
        path = os.path.join(ROOT, "profiles", f"{rnd}_pmc_traffic_{mix}.json")
        if not os.path.exists(path):
            continue
        try:
            d = json.load(open(path))
        except Exception:
            continue
        meta = d.pop("_meta", {}) if isinstance(d, dict) else {}
        meta.update(file=os.path.relpath(path, ROOT), kind="offline PMC pass (separate rocprofv3 --pmc runs), not measured "
                                                           "in this process")
        d["_source"] = meta
        return d
    return {}


def algorithmic_bytes(cfg):
    """SURVEY.md 8(d) algorithmic HBM bytes per LAUNCH of the decode classes: per sample, query 12 + neighbour positions 96 +
    8 feature rows of 128 B per feature set (2 156 B colour stage, 1 132 B geometry stage) gathered by the forward and again
    by the backward, + the read-modify-write of the gradient rows (4 096 / 2 048 B) where features are trained (mapper)."""
    tr, mp = cfg["tracking"], cfg["mapping"]
    pt, pm = 5 * tr["pixels"], 5 * mp["pixels"]
    return {"decode_fwd": 2156 * pm, "decode_bwd": (2156 + 4096) * pm, "decode_fwd_geo": 1132 * pm, "decode_bwd_geo": (1132 + 2048) * pm,
            "geo_iter": (1132 + 2048) * pm, "decode_fwd_track": 2156 * pt, "decode_bwd_track": 2156 * pt}


def roofline_of(prof, traffic=None, algo=None):
    if not prof:
        return None, {}
    per = {}
    for k, v in prof.items():
        if v["launches"] == 0 or v["ms"] <= 0:
            continue
        sec = v["ms"] * 1e-3
        if k in MFMA_CLASSES:
            per[k] = dict(bound="mfma", achieved=v["work"] / sec / 1e12, peak=PEAK_F32_MFMA_TFLOPS, unit="TFLOP/s",
                          avg_us=v["ms"] * 1e3 / v["launches"], launches=v["launches"], total_ms=v["ms"])
        else:
            per[k] = dict(bound="hbm", achieved=v["work"] / sec / 1e9, peak=PEAK_HBM_GBS, unit="GB/s",
                          avg_us=v["ms"] * 1e3 / v["launches"], launches=v["launches"], total_ms=v["ms"])
        per[k]["frac"] = per[k]["achieved"] / per[k]["peak"]
    if not per:
        return None, {}
    # "knn_side_stream": the mapper's k-NN block prefetch, throttled to two workgroups per CU on a second stream while
    # the decode kernels of the previous block run -- its (stretched) duration is not on the critical path
    dom = max((k for k in per if k not in ("misc", "knn_side_stream")), key=lambda k: per[k]["total_ms"])
    r = per[dom]
    tr = (traffic or {}).get(dom)
    roof = dict(kernel=dom, bound=r["bound"], achieved=round(r["achieved"], 4), peak=r["peak"], unit=r["unit"],
                frac=round(r["frac"], 5), traffic=tr.get("bytes_per_launch") if tr else None,
                avg_launch_us=round(r["avg_us"], 2), launches=r["launches"])
    if algo and dom in algo:
        roof["algorithmic_bytes_per_launch"] = int(algo[dom])
        if tr and tr.get("bytes_per_launch"):
            roof["traffic_over_algorithmic"] = round(tr["bytes_per_launch"] / algo[dom], 2)
    if tr:
        roof["traffic_detail"] = tr
    roof["traffic_source"] = (traffic or {}).get("_source") if tr else None
    return roof, per


def parity_vs_oracle(slam, cfg, cam, frame):
    """Part of the cpu_baseline leg (the only place bench.py touches oracle/): ONE tracker iteration, one geometry-stage and
    one colour-stage mapper iteration of the configured pixel budgets on the map AS THE RUN LEFT IT (>= 1 M points, trained
    features and decoder), evaluated by the HIP path through the C ABI and by the pinned oracle on identical inputs --
    outside every timed region.  The float that goes into config.render_loss_rel_err_vs_reference is the largest of the
    three loss-level relative errors (BASELINE.json: <= 1e-4)."""
    from tests import parity_probe as PP
    st = PP.oracle_state(slam)
    tr, mp = cfg["tracking"], cfg["mapping"]
    cases = [("tracker", tr["pixels"]), ("map_geometry", mp["pixels"]), ("map_color", mp["pixels"])]
    detail = [PP.probe(slam, cfg, cam, frame, k, n, seed=31 + j, state=st) for j, (k, n) in enumerate(cases)]
    worst = max(max(d["loss_rel"], d["geo_loss_rel"], d["col_loss_rel"]) for d in detail)
    keep = ("kind", "n_pix", "rays", "points", "loss", "loss_ref", "loss_rel", "geo_loss_rel", "col_loss_rel",
            "depth_rel_max", "rgb_abs_max", "g_rays_o_rel_l2", "g_rays_d_rel_l2", "g_geo_rel_l2", "g_col_rel_l2",
            "g_params_rel_l2")
    return worst, dict(checker="oracle/pointslam_oracle.py (pinned to the unmodified reference by tests/test_oracle_golden.py)",
                       cases=[{k: (float(f"{v:.4g}") if isinstance(v, float) else v) for k, v in d.items() if k in keep}
                              for d in detail])


def cpu_baseline_track_only(cfg, cam, n_points, pts):
    """BASELINE config 1 beside the GPU line: the CPU oracle's tracker iteration (exact cKDTree 8-NN + torch fp32 decoders +
    autograd through the pose, Tracker.py:89-186) on the SAME fixed cloud and image size, a bounded sample of iterations
    (~10-20 s), extrapolated to tracking.iters per frame."""
    import torch
    from oracle import pointslam_oracle as O
    from point_slam_amd import synthetic as syn
    from point_slam_amd.decoders import PointDecoders
    n_thr = min(os.cpu_count(), 16)
    torch.set_num_threads(n_thr)
    O.KNN_WORKERS = n_thr
    torch.manual_seed(cfg["setup_seed"])
    dec = PointDecoders(cfg)
    P = {k: v.detach() for k, v in dec.state_dict().items()}
    P["color_decoder.embedder._B"] = dec.color_decoder.embedder._B
    pts = pts.cpu().float()
    g = torch.Generator().manual_seed(7)
    geo = torch.zeros(pts.shape[0], 32).normal_(0, 0.1, generator=g)
    col = torch.zeros(pts.shape[0], 32).normal_(0, 0.1, generator=g)
    c2w = syn.pose(230.0)
    depth, color = syn.render_frame(cam, c2w)
    _, rq_img = syn.dynamic_radii(color, cfg)
    tr = cfg["tracking"]
    eh, ew = tr["ignore_edge_H"], tr["ignore_edge_W"]
    O.knn_exact(pts, pts[:8], 8)
    fb = torch.zeros(32)

    def one_iter():
        idx = torch.randint((cam["H"] - 2 * eh) * (cam["W"] - 2 * ew), (tr["pixels"],), generator=g)
        u, v = O.pixels_from_flat_index(idx, eh, cam["H"] - eh, ew, cam["W"] - ew)
        ro, rd = O.rays_from_uv(u, v, c2w, cam["fx"], cam["fy"], cam["cx"], cam["cy"])
        gd, gc, rq = depth[v.long(), u.long()], color[v.long(), u.long()], rq_img[v.long(), u.long()]
        ro = ro.clone().requires_grad_(True); rd = rd.clone().requires_grad_(True)
        d, var, rgb, valid, _ = O.render_batch_ray(cfg, P, pts, geo, col, ro, rd, gd, "color", rq, fb, fb, True)
        loss, *_ = O.tracker_loss(d, var, rgb, gd, gc)
        loss.backward()
    one_iter()
    n = 24
    t0 = time.perf_counter()
    for _ in range(n):
        one_iter()
    t_it = (time.perf_counter() - t0) / n
    return dict(value=round(1.0 / (tr["iters"] * t_it), 5), unit="frames/s", cores=n_thr, kind="port",
                sample=f"kind=port (/root/reference does not exist on the GPU box); oracle tracker iteration (cKDTree exact 8-NN + "
                       f"torch fp32 + autograd to the pose), N={pts.shape[0]} points, {cam['W']}x{cam['H']}, {tr['pixels']} px: "
                       f"{n} iterations of {t_it * 1e3:.0f} ms each, extrapolated to {tr['iters']} iterations per frame")


def cpu_baseline(cfg, cam, n_points):
    """The CPU oracle (port of the reference path: exact cKDTree 8-NN + torch fp32 decoders + autograd + Adam) on a
    bounded sample: 20 tracking + 20 geometry-stage + 40 colour-stage mapping iterations of the base mix over the same
    synthetic cloud (~10 s); FPS is extrapolated with the per-frame iteration counts."""
    import torch
    from oracle import pointslam_oracle as O
    from point_slam_amd import params as P_, synthetic as syn
    from point_slam_amd.decoders import PointDecoders
    n_thr = min(os.cpu_count(), 16)     # more threads only add contention on these small ops
    torch.set_num_threads(n_thr)
    O.KNN_WORKERS = n_thr
    torch.manual_seed(cfg["setup_seed"])
    dec = PointDecoders(cfg)
    P = {k: v.detach() for k, v in dec.state_dict().items()}
    P["color_decoder.embedder._B"] = dec.color_decoder.embedder._B
    pts = syn.seed_cloud(cam, n_points, n_views=64, seed=cfg["setup_seed"])
    g = torch.Generator().manual_seed(7)
    geo = torch.zeros(n_points, 32).normal_(0, 0.1, generator=g)
    col = torch.zeros(n_points, 32).normal_(0, 0.1, generator=g)
    c2w = syn.pose(200.0)
    depth, color = syn.render_frame(cam, c2w)
    _, rq_img = syn.dynamic_radii(color, cfg)
    O.knn_exact(pts, pts[:8], 8)         # builds (and caches) the kd-tree outside the timed sample
    fb = torch.zeros(32)

    def one_iter(n_pix, tracker, stage="color"):
        idx = torch.randint(cam["H"] * cam["W"], (n_pix,), generator=g)
        u, v = (idx % cam["W"]).float(), torch.div(idx, cam["W"], rounding_mode="floor").float()
        ro, rd = O.rays_from_uv(u, v, c2w, cam["fx"], cam["fy"], cam["cx"], cam["cy"])
        gd, gc, rq = depth[v.long(), u.long()], color[v.long(), u.long()], rq_img[v.long(), u.long()]
        if tracker:
            ro = ro.clone().requires_grad_(True); rd = rd.clone().requires_grad_(True)
            d, var, rgb, valid, _ = O.render_batch_ray(cfg, P, pts, geo, col, ro, rd, gd, "color", rq, fb, fb, True)
            loss, *_ = O.tracker_loss(d, var, rgb, gd, gc)
            loss.backward()
        else:
            gp, cp = geo.clone().requires_grad_(True), col.clone().requires_grad_(True)
            train = stage == "color"
            Pg = {k: (t.clone().requires_grad_(True) if train and k.startswith("color_decoder")
                      and t.dtype.is_floating_point and k != "color_decoder.embedder._B" else t) for k, t in P.items()}
            d, var, rgb, valid, _ = O.render_batch_ray(cfg, Pg, pts, gp, cp, ro, rd, gd, stage, rq, fb, fb, False)
            loss, *_ = O.mapper_loss(d, rgb, valid, gd, gc, stage)
            loss.backward()
            # dense Adam over a frustum-sized selection (~5 % of the cloud, as in the GPU run) + decoder
            n_sel = n_points * 5 // 100
            grads = (gp.grad[:n_sel], cp.grad[:n_sel]) if train else (gp.grad[:n_sel],)
            for t_ in grads:
                O.adam_step(torch.zeros_like(t_), t_, torch.zeros_like(t_), torch.zeros_like(t_), 1, 0.005)

    tr, mp = cfg["tracking"], cfg["mapping"]

    def timed(n, *a):
        one_iter(*a)                                   # untimed: first-touch / allocator warm-up
        t0 = time.perf_counter()
        for _ in range(n):
            one_iter(*a)
        return (time.perf_counter() - t0) / n
    n_t, n_g, n_c = 20, 20, 40                        # ~10 s of CPU work on the GPU box's host
    t_track = timed(n_t, tr["pixels"], True)
    t_geo = timed(n_g, mp["pixels"], False, "geometry")
    t_col = timed(n_c, mp["pixels"], False, "color")
    r = mp["geo_iter_ratio"]
    per_frame = tr["iters"] * t_track + mp["iters"] / mp["every_frame"] * (r * t_geo + (1.0 - r) * t_col)
    calib = None
    try:      # the imported reference against this port, per iteration, measured in the build container (oracle/calibrate_cpu_baseline.py)
        cf = next(f for f in ("r05_cpu_calibration.json", "r03_cpu_calibration.json") if os.path.exists(os.path.join(ROOT, "profiles", f)))
        cj = json.load(open(os.path.join(ROOT, "profiles", cf)))
        calib = dict(reference_over_port_per_frame=cj["reference_over_port_per_frame"], threads=cj["threads"],
                     file="profiles/" + cf)
    except Exception:
        pass
    return dict(value=round(1.0 / per_frame, 5), unit="frames/s", cores=n_thr, kind="port", calibration_vs_imported_reference=calib,
                sample=f"kind=port because /root/reference does not exist on the GPU box (the imported reference can only "
                       f"run in the build container, which has no GPU to compare with); "
                       f"oracle (cKDTree exact 8-NN + torch fp32 + autograd + Adam), N={n_points}: {n_t} tracking iters "
                       f"({t_track*1e3:.0f} ms each) + {n_g} geometry-stage ({t_geo*1e3:.0f} ms) + {n_c} colour-stage "
                       f"({t_col*1e3:.0f} ms) mapping iters, extrapolated to {tr['iters']} track + "
                       f"{mp['iters']}/{mp['every_frame']} map iters ({r:.0%} geometry stage) per frame")


def take_snapshot(slam, state, dev):
    """Everything a timed pass changes, so that the event-carrying pass can run on the SAME frames as the plain one: the map
    (point count, both feature stores, surface-point lists), the decoder blob and exposure state, the keyframe list with the
    poses / latents of its frames, the tracker's pose history and the random generators."""
    import torch
    npc = slam.npc
    N = npc.pts_num()
    return dict(N=N, geo=npc.get_geo_feats()[:N].clone(), col=npc.get_col_feats()[:N].clone(), theta=slam.theta.clone(),
                keyframes=list(slam.keyframes), kf=[(f, f.c2w, f.exposure) for f in slam.keyframes], n_mapped=slam.n_mapped,
                map_step=dict(slam.map_step), n_in=(len(npc._input_pos), len(npc._input_rgb)),
                exposure=(slam.exposure_feat.clone(), slam.exposure_mlp.clone()) if slam.encode_exposure else None,
                hist=[c.clone() for c in state.get("cam_hist", [])], traj=len(state.get("traj", [])),
                counts=(state["mapped"], state["added"]), map_log=len(state.get("map_log", [])),
                rng=(torch.get_rng_state(), torch.cuda.get_rng_state(dev)))


def restore_snapshot(slam, state, snap, dev):
    import torch
    npc = slam.npc
    npc.truncate(snap["N"])
    npc.get_geo_feats().copy_(snap["geo"])
    npc.get_col_feats().copy_(snap["col"])
    npc._build()
    del npc._input_pos[snap["n_in"][0]:], npc._input_rgb[snap["n_in"][1]:]
    slam.theta.copy_(snap["theta"])
    slam.keyframes[:] = snap["keyframes"]
    for f, c2w, ex in snap["kf"]:
        f.c2w, f.exposure = c2w, ex
    slam.n_mapped, slam.map_step = snap["n_mapped"], dict(snap["map_step"])
    if snap["exposure"] is not None:
        slam.exposure_feat, slam.exposure_mlp = snap["exposure"][0].clone(), snap["exposure"][1].clone()
    state["cam_hist"] = [c.clone() for c in snap["hist"]]
    del state.setdefault("traj", [])[snap["traj"]:]
    state["mapped"], state["added"] = snap["counts"]
    torch.set_rng_state(snap["rng"][0])
    torch.cuda.set_rng_state(snap["rng"][1], dev)
    torch.cuda.synchronize()


def ate_of(traj, frames):
    """Translation error (cm) of the closed loop's estimates against the ground truth: rmse, max, n."""
    import torch
    if not traj:
        return None
    est_t = torch.stack([t for _, t in traj]).cpu()
    gt_t = torch.stack([frames[k].gt_c2w[:3, 3] for k, _ in traj]).cpu()
    err = (est_t - gt_t).norm(dim=1) * 100.0
    return dict(rmse_cm=round(float((err ** 2).mean().sqrt()), 4), max_cm=round(float(err.max()), 4), frames=len(traj))


def self_launch(n):
    import socket
    import subprocess
    sk = socket.socket(); sk.bind(("127.0.0.1", 0)); port = sk.getsockname()[1]; sk.close()
    env = dict(os.environ)
    env.setdefault("HSA_ENABLE_IPC_MODE_LEGACY", "0")
    env.setdefault("OMP_NUM_THREADS", "4")
    if "--expandable-segments" in sys.argv:
        env.setdefault("PYTORCH_HIP_ALLOC_CONF", "expandable_segments:True")
    cmd = [sys.executable, "-m", "torch.distributed.run", "--nnodes=1", f"--nproc-per-node={n}", "--master-addr", "127.0.0.1",
           "--master-port", str(port), os.path.abspath(__file__)] + sys.argv[1:]
    return subprocess.call(cmd, env=env)


def pin_rank_to_cpus(local_rank, local_world, gpu_index):
    """One process per GPU, eight Python hosts on one node: each rank keeps to its own cores -- those of its GPU's NUMA node
    when the kernel says which that is (/sys/class/drm/card*/device/numa_node), split evenly among the ranks that share the
    node; otherwise an even slice of what the process may run on.  Returns what was done (printed per rank in the line)."""
    allowed = sorted(os.sched_getaffinity(0))
    how, cpus = "even slice of the allowed cores", None
    try:
        if gpu_index is not None:
            import glob
            import torch
            bdf = torch.cuda.get_device_properties(gpu_index).pci_bus_id if hasattr(torch.cuda.get_device_properties(gpu_index), "pci_bus_id") else None
            nodes = {}
            for d in glob.glob("/sys/class/drm/card*/device"):
                try:
                    nodes[os.path.basename(os.path.realpath(d)).lower()] = int(open(os.path.join(d, "numa_node")).read())
                except Exception:
                    pass
            node = None
            if bdf:
                node = next((v for k, v in nodes.items() if k.endswith(str(bdf).lower()[-7:])), None)
            if node is not None and node >= 0:
                txt = open(f"/sys/devices/system/node/node{node}/cpulist").read().strip()
                ncpus = []
                for part in txt.split(","):
                    a, _, b = part.partition("-")
                    ncpus += list(range(int(a), int(b or a) + 1))
                ncpus = [c for c in ncpus if c in allowed]
                n_nodes = max(len({v for v in nodes.values() if v >= 0}), 1)
                per_node = max(local_world // n_nodes, 1)
                k = local_rank % per_node
                share = max(len(ncpus) // per_node, 1)
                cpus, how = ncpus[k * share:(k + 1) * share], f"NUMA node {node} of the GPU, slice {k} of {per_node}"
    except Exception:
        cpus = None
    if not cpus:
        share = max(len(allowed) // max(local_world, 1), 1)
        cpus = allowed[local_rank * share:(local_rank + 1) * share] or allowed
    try:
        os.sched_setaffinity(0, cpus)
    except Exception as e:
        how += f" (sched_setaffinity failed: {e!r})"
    return dict(cpus=[cpus[0], cpus[-1]] if cpus else None, n=len(cpus), how=how)


def sub_records(args):
    """Two further runs of this script, after the timed passes (the headline stream is open loop and 20
    frames long; the line should also say what the loop does closed, and what a long run sustains):
      closed_loop   100 frames, frame i starts from the constant-speed extrapolation of the tracker's own two previous estimates,
                    mapping at the tracker's estimate, at the camera speed the base yaml's tracker budget holds (0.5 trajectory
                    units = 1.4 cm per frame; at the headline's 5.6 cm per frame the 200 px x 20 it tracker diverges);
      steady_state  500 open-loop frames of the headline stream: frames/s over the run and of every 100-frame window.
    Each is a fresh process (its own world, the same seeds); neither touches `value`."""
    import subprocess
    base = [sys.executable, os.path.abspath(__file__), "--no-cpu-baseline", "--no-kernel-timing", "--no-sub-records", "--mix", args.mix,
            "--points", str(args.points), "--width", str(args.width), "--height", str(args.height), "--warmup", str(args.warmup)]

    def run(extra, pick):
        try:
            r = subprocess.run(base + extra, capture_output=True, text=True, timeout=420)
            d = json.loads(r.stdout.strip().splitlines()[-1])
            return pick(d)
        except Exception as e:
            return {"error": repr(e)}
    closed = run(["--closed-loop", "--units-per-frame", "0.5", "--steps", "100"],
                 lambda d: dict(frames=d["steps"], fps=d["value"], ate_rmse_cm=d["config"]["ate_rmse_cm"], ate_max_cm=d["config"]["ate_max_cm"],
                                camera_speed="0.5 trajectory units (~1.4 cm) per frame", pose_loop=d["config"]["pose_loop"][:6]))
    steady = run(["--steps", "500", "--fps-window", "100"],
                 lambda d: dict(frames=d["steps"], fps=d["value"], fps_per_window=[w["fps"] for w in (d["config"]["fps_per_window"] or [])],
                                fps_window_min=min([w["fps"] for w in (d["config"]["fps_per_window"] or [])], default=None),
                                points_end=d["config"]["points_end"], device_used_mb=d["config"]["device_memory"]["device_used_mb"]))
    return closed, steady


def held_out_render_loss(slam, cfg, cam, frame, n_pix=4000, seed=5):
    """Mean |depth error| (m) and mean |colour error| of a render of `frame` at its ground-truth pose through the product path
    (HipRenderer.render_batch_ray -> psl_render_fwd): what the map looks like after an exchange, under either merge rule."""
    import torch
    from point_slam_amd import host_ops as H
    dev = frame.depth.device
    g = torch.Generator(device="cpu").manual_seed(seed)
    idx = torch.randint(cam["H"] * cam["W"], (n_pix,), generator=g).to(dev)
    u, v = H.pixels_from_flat_index(idx, 0, cam["H"], 0, cam["W"])
    ro, rd = H.get_rays_from_uv(u, v, frame.gt_c2w, cam["fx"], cam["fy"], cam["cx"], cam["cy"])
    ui, vi = u.long(), v.long()
    gd, gc = frame.depth[vi, ui].contiguous(), frame.color[vi, ui]
    rq = frame.r_query[vi, ui].contiguous() if frame.r_query is not None else None
    slam.sync_decoders_from_theta()
    r = slam.renderer
    old = (r.fixed_fallback, r.sigmoid_coefficient)
    r.fixed_fallback = (torch.zeros(32, device=dev), torch.zeros(32, device=dev))
    r.sigmoid_coefficient = cfg["rendering"]["sigmoid_coef_mapper"]
    with torch.no_grad():
        d, _, c, valid = r.render_batch_ray(slam.npc, slam.decoders, rd.contiguous(), ro.contiguous(), dev, "color", gt_depth=gd,
                                            npc_geo_feats=slam.npc.get_geo_feats(), npc_col_feats=slam.npc.get_col_feats(),
                                            dynamic_r_query=rq)
    r.fixed_fallback, r.sigmoid_coefficient = old
    m = valid & (gd > 0)
    return dict(depth_l1_m=round(float((d - gd).abs()[m].mean()), 6), colour_l1=round(float((c - gc).abs()[m].mean()), 6),
                valid_frac=round(float(m.float().mean()), 4))


def main():
    args = parse()
    if args.expandable_segments and (int(os.environ.get("WORLD_SIZE", "1")) > 1):
        os.environ.setdefault("PYTORCH_HIP_ALLOC_CONF", "expandable_segments:True")
    import torch
    rank = int(os.environ.get("RANK", "0"))
    world = int(os.environ.get("WORLD_SIZE", "1"))
    local_rank = int(os.environ.get("LOCAL_RANK", "0"))
    if args.gpus > 1 and "WORLD_SIZE" not in os.environ:
        # `python bench.py --gpus N` (how the driver's scaling run may call it): re-exec under torch.distributed.run,
        # one rank per GPU; rank 0 of the children prints the JSON line on the inherited stdout
        raise SystemExit(self_launch(args.gpus))
    if args.gpus != world:
        raise SystemExit(f"--gpus {args.gpus} but WORLD_SIZE={world}")
    if not torch.cuda.is_available():
        raise SystemExit("bench.py needs the MI355X (no CPU fallback for the product path)")
    # PSL_BENCH_SHARE_GPU=1 (debug only): all ranks on cuda:0 with the gloo backend, to exercise the N>1 code path
    # on a one-GPU box; RCCL refuses two ranks on one device
    share = os.environ.get("PSL_BENCH_SHARE_GPU") == "1"
    if share:
        local_rank = 0
    elif local_rank >= torch.cuda.device_count():
        raise SystemExit(f"rank {rank}: local rank {local_rank} but only {torch.cuda.device_count()} GPU(s) visible "
                         f"(PSL_BENCH_SHARE_GPU=1 runs all ranks on cuda:0 over gloo, for one-GPU boxes)")
    torch.cuda.set_device(local_rank)
    dev = torch.device(f"cuda:{local_rank}")
    if os.environ.get("PSL_POISON") == "1":
        # debug: leave 0x7F7F7F7F in the caching allocator's blocks so that a read of an uninitialised torch.empty
        # buffer shows up deterministically (the library poisons its own allocations under the same variable)
        junk = [torch.full((256 << 20,), 0x7F7F7F7F, dtype=torch.int32, device=dev) for _ in range(6)]
        torch.cuda.synchronize()
        del junk
    affinity = None
    if world > 1:
        affinity = pin_rank_to_cpus(local_rank, int(os.environ.get("LOCAL_WORLD_SIZE", world)), None if share else local_rank)
        torch.set_num_threads(max(1, min(4, affinity["n"])))
    if world > 1:
        import torch.distributed as dist
        os.environ.setdefault("HSA_ENABLE_IPC_MODE_LEGACY", "0")
        if share:
            dist.init_process_group("gloo", rank=rank, world_size=world)
        else:
            dist.init_process_group("nccl", rank=rank, world_size=world, device_id=dev)

    if args.closed_loop and world > 1:
        raise SystemExit("--closed-loop: one GPU (frame-parallel ranks see every N-th frame: a constant-speed extrapolation across "
                         "the stride is not the reference's pose chain)")
    if args.track_only and (args.warmup < 2 or world > 1):
        raise SystemExit("--track-only: one GPU, and --warmup >= 2 (frames 0 and 1 take the ground-truth pose, Tracker.py:278-279)")
    cfg, cam, slam, frames, cams0, every = build_world(args, rank, world, dev)
    state = dict(mapped=0, added=0)
    mpc = cfg["mapping"]
    per_kf = max(mpc["keyframe_every"] // mpc["every_frame"], 1)               # mapped frames per keyframe
    state["exchange_every"] = args.exchange_every or max(args.exchange_every_keyframes * per_kf, 1)
    mapped_in_timed = len([i for i in range(args.warmup, args.warmup + args.steps) if i % every == 0])
    if world > 1 and mapped_in_timed < state["exchange_every"]:
        # shorter than the cadence: one exchange at the last mapped frame of each timed pass
        last = [max([i for i in range(a, a + args.steps) if i % every == 0], default=-1)
                for a in (args.warmup, args.warmup + args.steps)]
        state["force_exchange_at_all"] = last
    if world > 1:
        from point_slam_amd import params as P_
        from point_slam_amd.dist import FrameParallelSync
        state["sync"] = slam.sync = FrameParallelSync(slam.npc, slam.theta, n_color=P_.color_floats(), merge=args.merge)

    def barrier():
        torch.cuda.synchronize()
        if world > 1:
            dist.barrier()
            torch.cuda.synchronize()

    points_start = slam.npc.pts_num()
    for i in range(args.warmup):
        run_step(i, slam, frames, cams0, every, cfg, world, args, state)
    from point_slam_amd import _lib

    mark = torch.zeros(1, device=dev) if os.environ.get("PSL_BENCH_MARK") == "1" else None

    windows = []

    def timed(first, record_windows=False):
        barrier()
        if mark is not None:        # a kernel no other code launches: tools/rocpd_window.py cuts the rocprofv3 trace at these
            mark.erfinv_()
            torch.cuda.synchronize()
        t0 = tw = time.perf_counter()
        for i in range(first, first + args.steps):
            run_step(i, slam, frames, cams0, every, cfg, world, args, state)
            if record_windows and args.fps_window > 0 and (i - first + 1) % args.fps_window == 0:
                torch.cuda.synchronize()
                now = time.perf_counter()
                windows.append(dict(frames=f"{i - args.fps_window + 1}..{i}", fps=round(args.fps_window / (now - tw), 2),
                                    points=slam.npc.pts_num(), mapped=state["mapped"]))
                tw = now
        barrier()
        d = time.perf_counter() - t0
        if mark is not None:
            mark.erfinv_()
            torch.cuda.synchronize()
        if world > 1:
            tt = torch.tensor([d], device=dev, dtype=torch.float64)
            dist.all_reduce(tt, op=dist.ReduceOp.MAX)
            d = float(tt.item())
        return d

    # (1) the timed region proper: EXACTLY `steps` frames, no instrumentation -> `value`
    same_frames = world == 1 and not args.different_frames and not args.no_kernel_timing
    snap = take_snapshot(slam, state, dev) if same_frames else None
    n_log0, n_traj0 = len(state.get("map_log", [])), len(state.get("traj", []))
    dt = timed(args.warmup, record_windows=True)
    pass1 = dict(map_log=state.get("map_log", [])[n_log0:], ate=ate_of(state.get("traj", [])[n_traj0:], frames),
                 points_end=slam.npc.pts_num(), mapped=state["mapped"], added=state["added"])
    # (2) the same work again with a HIP start/stop event pair on every kernel launch of the hot classes
    #     (hipExtLaunchKernelGGL stamps the pair with the dispatch's own begin/end -- the timestamps rocprofv3 reports)
    #     -> `roofline`.  Since round 5 on the SAME frames: the map, the poses and the RNGs are restored to what they
    #     were in front of pass (1) (--different-frames: the next `steps` frames, as rounds 1-4 did).  Event-carrying
    #     launches cost wall time; the profiled wall time is reported as profiled_ms_per_step.
    prof, dt_prof, pass2 = {}, None, None
    if not args.no_kernel_timing:
        first2 = args.warmup + (0 if same_frames else args.steps)
        if same_frames:
            restore_snapshot(slam, state, snap, dev)
            snap = None
        n_log0, n_traj0 = len(state.get("map_log", [])), len(state.get("traj", []))
        _lib.check(_lib.lib().psl_profile_enable(slam.npc.handle, max(1, args.event_stride)))
        dt_prof = timed(first2)
        prof = kernel_profile(slam)
        _lib.check(_lib.lib().psl_profile_enable(slam.npc.handle, 0))
        pass2 = dict(map_log=state.get("map_log", [])[n_log0:], ate=ate_of(state.get("traj", [])[n_traj0:], frames),
                     first_frame=first2)

    points_end = slam.npc.pts_num()
    mapped_total = max(state["mapped"], 1)
    per_rank = None
    if world > 1:
        # a last, untimed exchange: afterwards every replica must hold the same map
        t0 = time.perf_counter()
        state["sync"].exchange(slam.npc, slam.theta)
        torch.cuda.synchronize()
        t_ex = time.perf_counter() - t0
        state["exchange_s"] = [e0.elapsed_time(e1) * 1e-3 for e0, e1 in state.get("exchange_ev", [])]
        ex_ms = sorted(round(x * 1e3, 3) for x in state.get("exchange_s", []))
        try:
            loss_after = held_out_render_loss(slam, cfg, cam, frames[args.warmup + args.steps - 1])
        except Exception as e:
            loss_after = {"error": repr(e)}
        mine = dict(rank=rank, device=str(dev), cpu_affinity=affinity, host_threads=torch.get_num_threads(), points_end=points_end, points_after_final_exchange=slam.npc.pts_num(),
                    added=state["added"], mapped=state["mapped"], final_exchange_ms=round(t_ex * 1e3, 3),
                    exchange_ms=[round(x * 1e3, 3) for x in state.get("exchange_s", [])],
                    exchange_host_ms=state.get("exchange_host_ms", []),       # where the calling thread spent it: rows / decoder / new_points
                    exchange_ms_p50=ex_ms[len(ex_ms) // 2] if ex_ms else None, exchange_ms_p100=ex_ms[-1] if ex_ms else None,
                    ate_rmse_cm=(ate_of(state.get("traj", []), frames) or {}).get("rmse_cm"),
                    render_loss_after_final_exchange=loss_after,
                    feat_checksum=float(slam.npc.get_geo_feats().double().sum()))
        per_rank = [None] * world
        dist.all_gather_object(per_rank, mine)
    # (3) SURVEY.md 8(d): tracking-only and mapping-only rates next to the combined one (single GPU; after the measured
    #     regions, on frames already seen: 10 tracked frames, then 2 mapped frames at their true poses)
    split = None
    if world == 1 and not args.no_kernel_timing and not args.track_only:
        from point_slam_amd import frame_ops
        ids = list(range(max(len(frames) - 10, 0), len(frames)))
        torch.cuda.synchronize()
        t0 = time.perf_counter()
        for i in ids:
            if cfg["use_dynamic_radius"]:
                frames[i].r_add, frames[i].r_query = frame_ops.dynamic_radius_maps(frames[i].color, cfg)
            slam.track(frames[i], cams0[i])
        torch.cuda.synchronize()
        t_track = (time.perf_counter() - t0) / len(ids)
        t_maps, map_detail = [], []
        for i in ids[:3]:
            # a mapping call always follows the tracking of its frame (Point_SLAM's loop; every_frame >= 1); two mapping calls
            # directly after one another -- which no run of the loop produces -- cost the second one a one-off ~130 ms
            # (tools/map_repeat_probe.py: 43 / 176 / 43 / 43 ms; 43 / 43 / 43 with a tracked frame in between)
            slam.track(frames[i], cams0[i])
            torch.cuda.synchronize()
            t0 = time.perf_counter()
            slam.map(frames[i], frames[i].gt_c2w, n_iters=cfg["mapping"]["iters"], fixed_iters=True)   # the yaml's count, not the data-dependent one
            torch.cuda.synchronize()
            t_maps.append(time.perf_counter() - t0)
            map_detail.append(dict(slam.last_map, ms=round(t_maps[-1] * 1e3, 3)))
        # the MEDIAN of three calls: one call of a process pays a one-off of 60-130 ms somewhere (unattributed, DESIGN.md section 4;
        # the per-call figures are in map_calls)
        t_map = sorted(t_maps)[len(t_maps) // 2]
        split = {"track_ms_per_frame": round(t_track * 1e3, 3), "track_only_fps": round(1.0 / t_track, 2),
                 "map_ms_per_mapped_frame": round(t_map * 1e3, 3),
                 "map_only_fps": round(cfg["mapping"]["every_frame"] / t_map, 2),
                 # what the three mapping calls of this split ran (frustum rows, window, iterations: the cost of a call follows them)
                 "map_calls": map_detail}

    if rank == 0:
        pmc_key = "cfg5" if (args.points >= 2_000_000 and args.width >= 1280) else args.mix
        roof, per = roofline_of(prof, pmc_traffic(pmc_key), algorithmic_bytes(cfg))
        def iters_of(log):
            return dict(mapped_frames=len(log), map_iters=[m["n_iters"] for m in log], geo_iters=[m["n_geo"] for m in log],
                        n_sel=[m["n_sel"] for m in log], locations_added=[m["added"] for m in log])
        if roof is not None:
            # `value` comes from pass (1) (no instrumentation); the kernel classes -- and therefore `roofline` -- from pass (2):
            # the SAME frames from the restored snapshot (default), or the next `steps` frames (--different-frames, N > 1).
            f2 = pass2["first_frame"]
            roof["measured_on"] = (f"pass 2: frames {f2}..{f2 + args.steps - 1}" + (" (the SAME frames as pass 1: map, poses and RNGs restored "
                                   "from a snapshot taken after the warm-up)" if same_frames else " (the frames AFTER pass 1)")
                                   + f", event-carrying launches, {round(dt_prof / args.steps * 1e3, 3)} ms/step; `value` is pass 1: frames "
                                   f"{args.warmup}..{args.warmup + args.steps - 1}, no instrumentation, {round(dt / args.steps * 1e3, 3)} ms/step")
            # reconciliation of pass 2: wall time = sum of the classes on the main stream + what no class accounts for (device idle
            # while the host works between launches and at its synchronising calls, launches of torch / the runtime that carry no
            # event pair).  knn_side_stream runs on a second stream UNDER the decode launches: not part of the sum.
            cls = sum(v["total_ms"] for k, v in per.items() if k != "knn_side_stream") / args.steps
            wall2 = dt_prof / args.steps * 1e3
            roof["class_sum_ms_per_step"] = round(cls, 3)
            roof["unclassified_ms_per_step"] = round(wall2 - cls, 3)
            roof["unclassified_frac_of_wall"] = round((wall2 - cls) / wall2, 4)
            roof["instrumentation_overhead_ms_per_step"] = round(wall2 - dt / args.steps * 1e3, 3) if same_frames else None
            # what no class accounts for in `value` itself (same frames): device idle at the host's synchronising calls and between
            # dependent launches + torch's small kernels; the rocprofv3 trace cut at the pass markers measures the same thing
            # without any instrumentation (profiles/r05_window.txt: device busy 96 % of pass 1)
            wall1 = dt / args.steps * 1e3
            roof["value_pass_unclassified_ms_per_step"] = round(wall1 - cls, 3) if same_frames else None
            roof["value_pass_unclassified_frac"] = round((wall1 - cls) / wall1, 4) if same_frames else None
            roof["pass1_work"] = iters_of(pass1["map_log"])
            roof["pass2_work"] = iters_of(pass2["map_log"])
        tr, mp = cfg["tracking"], cfg["mapping"]
        out = {
            "metric": f"mapping+tracking FPS @{args.width}x{args.height}, {args.points / 1e6:g}M neural points",
            "value": round(world * args.steps / dt, 4), "unit": "frames/s", "n_gpus": world, "steps": args.steps,
            "warmup": args.warmup, "ms_per_step": round(dt / args.steps * 1e3, 3), "higher_is_better": True,
            "scaling": "weak", "vs_baseline": None, "dtype": "f32", "data": "synthetic",
            "config": {"workload": f"synthetic {args.width}x{args.height} RGB-D room, {args.points} seeded neural points, "
                                   f"{args.mix} iteration mix: track {tr['pixels']}px x {tr['iters']}it per frame, map "
                                   f"{mp['pixels']}px x {mp['iters']}it + {mp['pixels_adding']} add-pixels every "
                                   f"{mp['every_frame']} frames, window {mp['mapping_window_size']}"
                                   + (f"; sensor depth noise {args._depth_noise:g} x depth, {args._depth_dropout:g} of the pixels dropped to 0"
                                      if (args._depth_noise or args._depth_dropout) else ""),
                       "engine": args.engine, "points_start": points_start, "points_end": points_end,
                       "points_added_per_mapped_frame": round(state["added"] / mapped_total, 1),
                       "mapped_frames": state["mapped"],
                       # what the timed pass actually ran (the mapping iteration count is data dependent, Mapper.py:404-406)
                       "timed_pass": dict(iters_of(pass1["map_log"]), points_end=pass1["points_end"]),
                       "fps_per_window": windows or None,
                       "device_memory": dict(torch_allocated_peak_mb=round(torch.cuda.max_memory_allocated(dev) / 2 ** 20, 1),
                                             torch_reserved_mb=round(torch.cuda.memory_reserved(dev) / 2 ** 20, 1),
                                             device_used_mb=round((torch.cuda.mem_get_info(dev)[1] - torch.cuda.mem_get_info(dev)[0]) / 2 ** 20, 1)),
                       "pose_loop": ("open: every frame starts from the ground-truth pose + noise (rounds 1-4)" if args.open_loop else
                                     "closed: frame i starts from the constant-speed extrapolation of the tracker's own two previous "
                                     "estimates (psl_pose_const_speed, Tracker.py:283-290); mapping at the tracker's estimate"),
                       # translation error of the tracker's lowest-loss pose against the ground truth over the timed frames: the
                       # trajectory error of the closed loop, or (open loop) what is left of the initial perturbation
                       "ate_rmse_cm": pass1["ate"]["rmse_cm"] if pass1["ate"] else None,
                       "ate_max_cm": pass1["ate"]["max_cm"] if pass1["ate"] else None,
                       "ate_event_pass": pass2["ate"] if pass2 else None,
                       "parallelism": f"frame-parallel x{world}" if world > 1 else "single GPU",
                       "exchange_cadence": (f"every {state['exchange_every']} mapped frames of a rank"
                                            + (f" (= {args.exchange_every_keyframes} keyframes, BASELINE config 4)" if not args.exchange_every else "")
                                            + ("; this run is shorter: ONE exchange forced at the last mapped frame of the timed region"
                                               if state.get("force_exchange_at_all") else "")) if world > 1 else None,
                       "keyframes_kept": "last 40 (the reference keeps every keyframe on the CPU)",
                       # measured in this run by the cpu_baseline leg (parity_vs_oracle); null when that leg is off
                       "render_loss_rel_err_vs_reference": None},
            "roofline": roof,
            "profiled_ms_per_step": round(dt_prof / args.steps * 1e3, 3) if dt_prof else None,
            "event_timing": (f"hipExtLaunchKernelGGL start/stop events on 1 launch in {max(1, args.event_stride)} of each kernel "
                             "class; class totals = mean of the timed launches x launches") if dt_prof else None,
            "split": split,
            "kernels": {k: {kk: (round(vv, 5) if isinstance(vv, float) else vv) for kk, vv in v.items()}
                        for k, v in per.items()},
        }
        if args.track_only:
            # the trajectory the closed loop produced against the ground truth (translation, cm), both passes
            a = ate_of(state.get("traj", []), frames)
            if a:
                out["config"]["ate_rmse_cm"], out["config"]["ate_max_cm"], out["config"]["tracked_frames"] = a["rmse_cm"], a["max_cm"], a["frames"]
            out["metric"] = (f"tracking FPS @{args.width}x{args.height}, fixed {args.points / 1e3:g}k-point cloud "
                             f"(BASELINE config 1: tracking only)")
            out["config"]["workload"] = (f"synthetic {args.width}x{args.height} RGB-D room, FIXED cloud of {args.points} neural points, "
                                         f"features trained beforehand (untimed) by mapping {getattr(args, '_trained_keyframes', 0)} keyframes at their "
                                         f"true poses, no point adding; tracking only: {tr['pixels']}px x {tr['iters']}it per frame ({args.mix} mix), every frame "
                                         f"initialised by constant-speed extrapolation of the tracker's own two previous estimates")
            for k in ("points_added_per_mapped_frame", "mapped_frames", "keyframes_kept", "timed_pass", "ate_event_pass"):
                out["config"].pop(k, None)
        if per_rank is not None:
            from point_slam_amd.dist import transport_name
            out["config"]["rccl_ranks"] = world if dist.get_backend() == "nccl" else 0
            out["config"]["process_group_backend"] = dist.get_backend()
            out["config"]["exchange_transport"] = transport_name(state["sync"].transport)
            out["config"]["merge_rule"] = args.merge
            out["config"]["allocator"] = os.environ.get("PYTORCH_HIP_ALLOC_CONF", "default")
            out["config"]["per_rank"] = per_rank
            out["config"]["replicas_identical_after_exchange"] = (
                len({r["points_after_final_exchange"] for r in per_rank}) == 1 and
                len({r["feat_checksum"] for r in per_rank}) == 1)
        if world == 1 and not args.no_cpu_baseline and args.track_only:
            try:
                from tests import parity_probe as PP
                d = PP.probe(slam, cfg, cam, frames[args.warmup + args.steps - 1], "tracker", tr["pixels"], seed=31)
                out["config"]["render_loss_rel_err_vs_reference"] = float(f"{max(d['loss_rel'], d['geo_loss_rel'], d['col_loss_rel']):.4g}")
            except Exception as e:
                out["config"]["render_loss_parity"] = {"error": repr(e)}
            try:
                out["cpu_baseline"] = cpu_baseline_track_only(cfg, cam, args.points, slam.npc.cloud_pos())
            except Exception as e:
                out["cpu_baseline"] = {"error": repr(e)}
        elif world == 1 and not args.no_cpu_baseline:
            try:
                worst, detail = parity_vs_oracle(slam, cfg, cam, frames[args.warmup + args.steps - 1])
                out["config"]["render_loss_rel_err_vs_reference"] = float(f"{worst:.4g}")
                out["config"]["render_loss_parity"] = detail
            except Exception as e:
                out["config"]["render_loss_parity"] = {"error": repr(e)}
            try:
                out["cpu_baseline"] = cpu_baseline(cfg, cam, args.points)
            except Exception as e:      # the baseline must never take the measured line down with it
                out["cpu_baseline"] = {"error": repr(e)}
        # only in the full default invocation (what the driver runs): development calls pass --no-cpu-baseline
        if (world == 1 and not args.no_sub_records and not args.no_cpu_baseline and not args.track_only and not args.closed_loop
                and not args.no_kernel_timing):
            out["config"]["closed_loop"], out["config"]["steady_state"] = sub_records(args)
        print(json.dumps(out))
    if world > 1:
        dist.barrier()
        dist.destroy_process_group()


if __name__ == "__main__":
    main()
