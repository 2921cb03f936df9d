"""Loss-level parity of ONE tracker / mapper iteration at any map size: the HIP path (HipRenderer.render_batch_ray ->
psl_render_fwd / psl_render_bwd through the C ABI) against the pinned CPU oracle (oracle/pointslam_oracle.py) on
identical inputs -- the same rays, sensor depths, query radii, features, decoder parameters and fallback vectors.

What SURVEY.md 8(d) "parity protocol" asks for: |L_hip - L_ref| / |L_ref| for the geometry and colour losses
separately, max / mean per-ray |d depth| / depth and |d rgb|, gradient rel-L2 (and cosine) for features, decoder
parameters and the pose (ray) gradients.

Test infrastructure: imported by tests/test_hip_fullsize.py and by bench.py's `cpu_baseline` leg only (the oracle is
the checker, never the thing measured).
"""
from __future__ import annotations

import torch

from oracle import pointslam_oracle as O
from point_slam_amd import host_ops as H
from point_slam_amd import params as P_


def oracle_state(slam):
    """CPU copies of everything the oracle needs: cloud [N,3], both feature sets, the decoder tensors."""
    slam.sync_decoders_from_theta()
    dec = slam.decoders
    P = {k: v.detach().cpu().clone() for k, v in dec.state_dict().items()}
    P["color_decoder.embedder._B"] = dec.color_decoder.embedder._B.detach().cpu().clone()
    return dict(cloud=slam.npc.cloud_pos().float(), geo=slam.npc.get_geo_feats().cpu().clone(),
                col=slam.npc.get_col_feats().cpu().clone(), P=P)


def draw_rays(cam, frame, n_pix, seed, edge=0):
    """n_pix uniformly drawn pixels of `frame` with sensor depth, after the depth-outlier mask (common.py:162-183,
    Tracker.py:142-149): rays_o, rays_d, gt_depth, gt_color, r_query -- on the frame's device."""
    dev = frame.depth.device
    g = torch.Generator(device="cpu").manual_seed(seed)
    Hh, Ww = cam["H"], cam["W"]
    idx = torch.randint((Hh - 2 * edge) * (Ww - 2 * edge), (n_pix,), generator=g).to(dev)
    u, v = H.pixels_from_flat_index(idx, edge, Hh - edge, edge, Ww - edge)
    ro, rd = H.get_rays_from_uv(u, v, frame.c2w, cam["fx"], cam["fy"], cam["cx"], cam["cy"])
    ui, vi = u.long(), v.long()
    gd, gc = frame.depth[vi, ui], frame.color[vi, ui]
    rq = frame.r_query[vi, ui] if frame.r_query is not None else None
    keep = gd > 0
    ro, rd, gd, gc = ro[keep], rd[keep], gd[keep], gc[keep]
    rq = rq[keep] if rq is not None else None
    inl = H.depth_inlier_mask(gd)
    ro, rd, gd, gc = ro[inl].contiguous(), rd[inl].contiguous(), gd[inl].contiguous(), gc[inl].contiguous()
    rq = rq[inl].contiguous() if rq is not None else None
    return ro, rd, gd, gc, rq


def _rel(a, b):
    return float(abs(a - b) / max(abs(b), 1e-30))


def _grad_metrics(got, ref):
    got, ref = got.detach().double().cpu().reshape(-1), ref.detach().double().reshape(-1)
    n = float(ref.norm())
    rel_l2 = float((got - ref).norm() / max(n, 1e-30))
    cos = float((got @ ref) / max(float(got.norm()) * n, 1e-30))
    return rel_l2, cos


def probe(slam, cfg, cam, frame, kind, n_pix, seed=0, state=None, knn_cache=None):
    """kind: 'tracker' (colour stage, pose gradients, Tracker.py:89-186), 'map_geometry' / 'map_color'
    (feature (+ colour-decoder) gradients, Mapper.py:408-568).  Returns a dict of plain floats."""
    dev = frame.depth.device
    st = state or oracle_state(slam)
    tracker = kind == "tracker"
    stage = "geometry" if kind == "map_geometry" else "color"
    edge = cfg["tracking"]["ignore_edge_H"] if tracker else 0
    ro, rd, gd, gc, rq = draw_rays(cam, frame, n_pix, seed, edge)
    g = torch.Generator(device="cpu").manual_seed(seed + 77)
    fb = torch.zeros(2, 32).normal_(mean=0, std=0.01, generator=g)
    tr, mp = cfg["tracking"], cfg["mapping"]

    # ---------------------------------------------------------------- HIP (C ABI through the drop-in autograd function)
    r = slam.renderer
    old = (r.fixed_fallback, r.sigmoid_coefficient, r.skip_decoder_grads)
    r.fixed_fallback = (fb[0].to(dev), fb[1].to(dev))
    r.sigmoid_coefficient = cfg["rendering"]["sigmoid_coef_tracker" if tracker else "sigmoid_coef_mapper"]
    slam.sync_decoders_from_theta()
    for p in slam.decoders.parameters():
        p.requires_grad_(False)
        p.grad = None
    geo_h, col_h = slam.npc.get_geo_feats(), slam.npc.get_col_feats()
    dec_names = []
    if tracker:
        r.skip_decoder_grads = True
        ro_h, rd_h = ro.clone().requires_grad_(True), rd.clone().requires_grad_(True)
    else:
        r.skip_decoder_grads = False
        ro_h, rd_h = ro, rd
        geo_h = geo_h.clone().requires_grad_(True)
        col_h = col_h.clone().requires_grad_(True)
        if stage == "color" and not mp["fix_color_decoder"]:
            for n_, p in slam.decoders.color_decoder.named_parameters():
                if "mlp_exposure" not in n_:
                    p.requires_grad_(True)
                    dec_names.append("color_decoder." + n_)
    d, v, c, valid = r.render_batch_ray(slam.npc, slam.decoders, rd_h, ro_h, dev, stage, gt_depth=gd,
                                        npc_geo_feats=geo_h, npc_col_feats=col_h, is_tracker=tracker,
                                        dynamic_r_query=rq)
    if tracker:
        L, Lg, Lc, m = H.tracker_loss(d, v, c, gd, gc, tr["handle_dynamic"], tr["use_color_in_tracking"], tr["w_color_loss"])
    else:
        L, Lg, Lc, m = H.mapper_loss(d, c, valid, gd, gc, stage, mp["w_color_loss"])
    L.backward()
    torch.cuda.synchronize()
    r.fixed_fallback, r.sigmoid_coefficient, r.skip_decoder_grads = old

    # ---------------------------------------------------------------- oracle (CPU)
    P = st["P"]
    ro_c, rd_c, gd_c, gc_c = ro.cpu(), rd.cpu(), gd.cpu(), gc.cpu()
    rq_c = rq.cpu() if rq is not None else None
    if tracker:
        ro_o, rd_o = ro_c.clone().requires_grad_(True), rd_c.clone().requires_grad_(True)
        geo_o, col_o, Pg = st["geo"], st["col"], P
    else:
        ro_o, rd_o = ro_c, rd_c
        geo_o, col_o = st["geo"].clone().requires_grad_(True), st["col"].clone().requires_grad_(True)
        Pg = {k: (t.clone().requires_grad_(True) if k in dec_names else t) for k, t in P.items()}
    do, vo, co, valid_o, aux = O.render_batch_ray(cfg, Pg, st["cloud"], geo_o, col_o, ro_o, rd_o, gd_c, stage, rq_c,
                                                  fb[0], fb[1], pts_grad=tracker,
                                                  coef=cfg["rendering"]["sigmoid_coef_tracker" if tracker else "sigmoid_coef_mapper"])
    if tracker:
        Lo, Lgo, Lco, mo = O.tracker_loss(do, vo, co, gd_c, gc_c, tr["handle_dynamic"], tr["use_color_in_tracking"],
                                          tr["w_color_loss"])
    else:
        Lo, Lgo, Lco, mo = O.mapper_loss(do, co, valid_o, gd_c, gc_c, stage, mp["w_color_loss"])
    Lo.backward()

    out = dict(kind=kind, n_pix=int(n_pix), rays=int(ro.shape[0]), points=int(st["cloud"].shape[0]),
               loss=float(L), loss_ref=float(Lo), loss_rel=_rel(float(L), float(Lo)),
               geo_loss_rel=_rel(float(Lg), float(Lgo)),
               col_loss_rel=_rel(float(Lc), float(Lco)) if stage == "color" else 0.0,
               mask_mismatch=int((m.cpu() != mo).sum()), valid_mismatch=int((valid.cpu() != valid_o).sum()),
               valid_frac=float(valid.float().mean()),
               depth_rel_max=float(((d.detach().cpu() - do.detach()).abs() / gd_c).max()),
               depth_rel_mean=float(((d.detach().cpu() - do.detach()).abs() / gd_c).mean()),
               rgb_abs_max=float((c.detach().cpu() - co.detach()).abs().max()),
               rgb_abs_mean=float((c.detach().cpu() - co.detach()).abs().mean()))
    if tracker:
        out["g_rays_o_rel_l2"], out["g_rays_o_cos"] = _grad_metrics(ro_h.grad, ro_o.grad)
        out["g_rays_d_rel_l2"], out["g_rays_d_cos"] = _grad_metrics(rd_h.grad, rd_o.grad)
    else:
        out["g_geo_rel_l2"], out["g_geo_cos"] = _grad_metrics(geo_h.grad, geo_o.grad)
        if stage == "color":
            out["g_col_rel_l2"], out["g_col_cos"] = _grad_metrics(col_h.grad, col_o.grad)
            if dec_names:
                named = dict(slam.decoders.named_parameters())
                gh = torch.cat([named[n_].grad.reshape(-1).cpu() for n_ in dec_names])
                go = torch.cat([Pg[n_].grad.reshape(-1) for n_ in dec_names])
                out["g_params_rel_l2"], out["g_params_cos"] = _grad_metrics(gh, go)
    for p in slam.decoders.parameters():
        p.requires_grad_(False)
        p.grad = None
    return out
