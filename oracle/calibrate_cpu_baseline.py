"""Calibration of bench.py's `cpu_baseline` (kind "port"): the UNMODIFIED reference's render + loss + backward, imported
from /root/reference (oracle/ref_import.py), timed per iteration against the oracle port on the SAME inputs, in the build
container (no GPU there; /root/reference does not exist on the GPU box, which is why the bench line can only time the
port).  Writes profiles/r05_cpu_calibration.json (round 3's file: r03_cpu_calibration.json): the ratio reference / port per iteration kind says how far the port's
frames/s is from what the real reference would show on the same cores.

Both legs use the same exact k-NN (oracle.knn_exact on a cKDTree) -- FAISS is absent from the image -- so the ratio isolates
the decoders / renderer / autograd graph of the reference against their restatement.

Run in the build container only:   python -m oracle.calibrate_cpu_baseline [n_points]
"""
from __future__ import annotations

import json
import os
import sys
import time
import types

import torch

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)

from oracle import pointslam_oracle as O  # noqa: E402
from oracle import ref_import as RI  # noqa: E402
from oracle.gen_golden import base_cfg, make_ref_npc  # noqa: E402
from point_slam_amd import synthetic as syn  # noqa: E402


def main():
    n_points = int(sys.argv[1]) if len(sys.argv) > 1 else 1_000_000
    n_thr = min(os.cpu_count(), 16)
    torch.set_num_threads(n_thr)
    O.KNN_WORKERS = n_thr
    ns = RI.load()
    cfg = base_cfg()
    dec = RI.make_decoders(cfg)
    P = RI.state_with_fixed_B(dec)
    cam = syn.intrinsics(640, 480)
    pts = syn.seed_cloud(cam, n_points, n_views=64, seed=cfg["setup_seed"])
    g = torch.Generator().manual_seed(7)
    geo = torch.zeros(n_points, 32).normal_(0, 0.1, generator=g)
    col = torch.zeros(n_points, 32).normal_(0, 0.1, generator=g)
    c2w = syn.pose(200.0)
    depth, color = syn.render_frame(cam, c2w)
    _, rq_img = syn.dynamic_radii(color, cfg)
    npc = make_ref_npc(ns, cfg, pts, geo, col)
    rend = ns.Renderer(cfg, None, types.SimpleNamespace(**cam))
    rend.sigmoid_coefficient = 0.1
    wrapped = RI.PointCPU(dec)
    O.knn_exact(pts, pts[:8], 8)                    # kd-tree built outside the timed region (shared by both legs)
    fb = torch.zeros(32)

    def rays(n_pix):
        idx = torch.randint(cam["H"] * cam["W"], (n_pix,), generator=g)
        u, v = (idx % cam["W"]).float(), torch.div(idx, cam["W"], rounding_mode="floor").float()
        ro, rd = O.rays_from_uv(u, v, c2w, cam["fx"], cam["fy"], cam["cx"], cam["cy"])
        return ro, rd, depth[v.long(), u.long()], color[v.long(), u.long()], rq_img[v.long(), u.long()]

    def ref_iter(n_pix, tracker, stage):
        ro, rd, gd, gc, rq = rays(n_pix)
        for p in dec.parameters():
            p.requires_grad_(not tracker and stage == "color" and True)
            p.grad = None
        if tracker:
            ro, rd = ro.clone().requires_grad_(True), rd.clone().requires_grad_(True)
            gp, cp = geo, col
        else:
            gp, cp = geo.clone().requires_grad_(True), col.clone().requires_grad_(True)
        d, var, rgb, valid = rend.render_batch_ray(npc, wrapped, rd, ro, "cpu", stage, gt_depth=gd, npc_geo_feats=gp,
                                                   npc_col_feats=cp, is_tracker=tracker, cloud_pos=pts, dynamic_r_query=rq)
        if tracker:
            loss, *_ = O.tracker_loss(d, var, rgb, gd, gc)
        else:
            loss, *_ = O.mapper_loss(d, rgb, valid, gd, gc, stage)
        loss.backward()
        return float(loss)

    def port_iter(n_pix, tracker, stage):
        ro, rd, gd, gc, rq = rays(n_pix)
        if tracker:
            ro, rd = ro.clone().requires_grad_(True), rd.clone().requires_grad_(True)
            d, var, rgb, valid, _ = O.render_batch_ray(cfg, P, pts, geo, col, ro, rd, gd, "color", rq, fb, fb, True)
            loss, *_ = O.tracker_loss(d, var, rgb, gd, gc)
        else:
            gp, cp = geo.clone().requires_grad_(True), col.clone().requires_grad_(True)
            train = stage == "color"
            Pg = {k: (t.clone().requires_grad_(True) if train and k.startswith("color_decoder") and t.dtype.is_floating_point
                      and k != "color_decoder.embedder._B" else t) for k, t in P.items()}
            d, var, rgb, valid, _ = O.render_batch_ray(cfg, Pg, pts, gp, cp, ro, rd, gd, stage, rq, fb, fb, False)
            loss, *_ = O.mapper_loss(d, rgb, valid, gd, gc, stage)
        loss.backward()
        return float(loss)

    def timed(fn, n, *a):
        fn(*a)
        t0 = time.perf_counter()
        for _ in range(n):
            fn(*a)
        return (time.perf_counter() - t0) / n

    tr, mp = cfg["tracking"], cfg["mapping"]
    out = dict(points=n_points, threads=n_thr, torch=torch.__version__,
               note="render + loss + backward per iteration, identical inputs and k-NN; Adam excluded on both sides; median of 5 "
                    "interleaved repeats per leg; the build container has 8 cores (the GPU box's host runs the port on 16 threads: "
                    "the RATIO is what transfers, and it is quoted with the thread count it was measured at)", cases=[])
    # interleaved repeats (reference, port, reference, port ...): the median of each leg, so that a noisy neighbour on the
    # shared host does not land on one leg only (round 3 timed each leg once, 8-12 iterations)
    for name, n, (n_pix, tracker, stage) in (("tracker", 12, (tr["pixels"], True, "color")),
                                              ("map_geometry", 8, (mp["pixels"], False, "geometry")),
                                              ("map_color", 8, (mp["pixels"], False, "color"))):
        refs, ports = [], []
        for _ in range(5):
            refs.append(timed(ref_iter, n, n_pix, tracker, stage))
            ports.append(timed(port_iter, n, n_pix, tracker, stage))
        t_ref, t_port = sorted(refs)[2], sorted(ports)[2]
        out["cases"].append(dict(kind=name, n_pix=n_pix, reference_ms=round(t_ref * 1e3, 2), port_ms=round(t_port * 1e3, 2),
                                 reference_over_port=round(t_ref / t_port, 3)))
        print(out["cases"][-1])
    r = mp["geo_iter_ratio"]
    c = {d["kind"]: d for d in out["cases"]}

    def frame(key):
        return tr["iters"] * c["tracker"][key] + mp["iters"] / mp["every_frame"] * (r * c["map_geometry"][key] + (1 - r) * c["map_color"][key])
    out["frames_per_s_reference"] = round(1e3 / frame("reference_ms"), 4)
    out["frames_per_s_port"] = round(1e3 / frame("port_ms"), 4)
    out["reference_over_port_per_frame"] = round(frame("reference_ms") / frame("port_ms"), 3)
    path = os.path.join(ROOT, "profiles", "r05_cpu_calibration.json")
    json.dump(out, open(path, "w"), indent=1)
    print("wrote", path, out["reference_over_port_per_frame"])


if __name__ == "__main__":
    main()
