"""HipRenderer: drop-in for the reference's Renderer (src/utils/Renderer.py:6-283).

`render_batch_ray` keeps the reference signature and return tuple
(Renderer.py:77-79,202) and is autograd-differentiable w.r.t. rays_o / rays_d
(pose), npc_geo_feats / npc_col_feats, the decoder parameters and
exposure_feat, but the whole chain -- 5 samples per ray, 8-NN lookup,
interpolation, both MLP decoders, compositing -- runs in the HIP kernels of
libpointslam_hip.so through psl_render_fwd / psl_render_bwd.  There is no
torch/CPU fallback.
"""
from __future__ import annotations

import ctypes as C

import torch

from . import _lib
from . import params as P_


class _RenderFn(torch.autograd.Function):
    @staticmethod
    def forward(ctx, rays_o, rays_d, geo_feats, col_feats, theta, affine, m):
        L = _lib.lib()
        R = rays_o.shape[0]
        dev = rays_o.device
        rays_o = rays_o.detach().float().contiguous()
        rays_d = rays_d.detach().float().contiguous()
        geo_feats = geo_feats.detach().float().contiguous()
        col_feats = col_feats.detach().float().contiguous() if col_feats is not None else None
        theta_c = theta.detach().float().contiguous()
        affine_c = affine.detach().float().contiguous() if affine is not None else None
        flags = m["flags"]
        ws = torch.empty(int(L.psl_render_ws_floats(R, flags)), device=dev, dtype=torch.float32)
        depth = torch.empty(R, device=dev, dtype=torch.float32)
        var = torch.empty(R, device=dev, dtype=torch.float32)
        rgb = torch.empty(R, 3, device=dev, dtype=torch.float32)
        valid = torch.empty(R, device=dev, dtype=torch.uint8)
        a = _lib.psl_render_args(
            n_rays=R, flags=flags, sigmoid_coef=float(m["coef"]),
            rays_o=rays_o.data_ptr(), rays_d=rays_d.data_ptr(), gt_depth=m["gt_depth"].data_ptr(),
            r_query=m["r_query"].data_ptr() if m["r_query"] is not None else None,
            geo_feats=geo_feats.data_ptr(), col_feats=col_feats.data_ptr() if col_feats is not None else None,
            params=theta_c.data_ptr(), col_embed_B=m["Bcol"].data_ptr(),
            fallback_geo=m["fb_geo"].data_ptr(), fallback_col=m["fb_col"].data_ptr(),
            exposure_affine=affine_c.data_ptr() if affine_c is not None else None,
            ws=ws.data_ptr(), depth=depth.data_ptr(), var=var.data_ptr(), rgb=rgb.data_ptr(),
            valid_ray=valid.data_ptr(), z_vals=m["z_vals"].data_ptr() if m.get("z_vals") is not None else None)
        _lib.check(L.psl_render_fwd(m["handle"], C.byref(a), _lib.stream_ptr()), "psl_render_fwd")
        ctx.m = m
        ctx.args = a
        ctx.keep = (rays_o, rays_d, geo_feats, col_feats, theta_c, affine_c, ws, depth, var, rgb, valid)
        ctx.n_feat = geo_feats.shape[0]
        valid_b = valid.bool()
        ctx.mark_non_differentiable(valid_b)
        return depth, var, rgb, valid_b

    @staticmethod
    def backward(ctx, g_depth, g_var, g_rgb, _g_valid):
        L = _lib.lib()
        m, a = ctx.m, ctx.args
        flags = m["flags"]
        dev = g_depth.device
        R = a.n_rays
        need = ctx.needs_input_grad
        g_depth = g_depth.float().contiguous()
        g_var = g_var.float().contiguous() if g_var is not None else None
        g_rgb = g_rgb.float().contiguous() if g_rgb is not None else None
        g_o = g_d = g_geo = g_col = g_theta = g_aff = None
        if flags & _lib.PTS_GRAD:
            g_o = torch.empty(R, 3, device=dev)
            g_d = torch.empty(R, 3, device=dev)
        if flags & _lib.FEAT_GRAD:
            g_geo = torch.zeros(ctx.n_feat, 32, device=dev)
            # stage 'geometry' never touches the colour features (decoder.py:497-505): their gradient must stay
            # None, not zeros -- torch.optim.Adam skips None but would advance its step count on zeros
            if flags & _lib.STAGE_COLOR:
                g_col = torch.zeros(ctx.n_feat, 32, device=dev)
        if flags & _lib.PARAM_GRAD:
            g_theta = torch.empty(P_.master_floats(), device=dev)
        if flags & _lib.HAS_AFFINE:
            g_aff = torch.empty(12, device=dev)
        g = _lib.psl_render_grads(
            g_depth=g_depth.data_ptr(), g_var=g_var.data_ptr() if g_var is not None else None,
            g_rgb=g_rgb.data_ptr() if g_rgb is not None else None,
            g_geo_feats=g_geo.data_ptr() if g_geo is not None else None,
            g_col_feats=g_col.data_ptr() if g_col is not None else None, feat_row_map=None,
            g_params=g_theta.data_ptr() if g_theta is not None else None,
            g_rays_o=g_o.data_ptr() if g_o is not None else None,
            g_rays_d=g_d.data_ptr() if g_d is not None else None,
            g_exposure_affine=g_aff.data_ptr() if g_aff is not None else None)
        _lib.check(L.psl_render_bwd(m["handle"], C.byref(a), C.byref(g), _lib.stream_ptr()), "psl_render_bwd")
        return (g_o if need[0] else None, g_d if need[1] else None, g_geo if need[2] else None,
                g_col if need[3] else None, g_theta if need[4] else None, g_aff if need[5] else None, None)


class HipRenderer(object):
    """Same constructor and attributes as the reference Renderer (Renderer.py:6-21)."""

    def __init__(self, cfg, args, slam, points_batch_size=500000, ray_batch_size=3000):
        self.ray_batch_size = ray_batch_size
        self.points_batch_size = points_batch_size
        self.N_surface = cfg['rendering']['N_surface']
        self.near_end_surface = cfg['rendering']['near_end_surface']
        self.far_end_surface = cfg['rendering']['far_end_surface']
        self.sample_near_pcl = cfg['rendering']['sample_near_pcl']
        self.near_end = cfg['rendering']['near_end']
        self.use_dynamic_radius = cfg['use_dynamic_radius']
        self.crop_edge = 0 if cfg['cam']['crop_edge'] is None else cfg['cam']['crop_edge']
        self.encode_exposure = cfg['model']['encode_exposure']
        self.H, self.W, self.fx, self.fy, self.cx, self.cy = slam.H, slam.W, slam.fx, slam.fy, slam.cx, slam.cy
        self.sigmoid_coefficient = cfg['rendering']['sigmoid_coef_mapper']   # set by callers (Tracker.py:36)
        # The reference tracker back-propagates into every decoder weight although its optimiser only holds the
        # pose (Tracker.py:305-311,182).  Set True (HipTracker does) to skip that dead work.
        self.skip_decoder_grads = False
        self.geo_decoder_trainable = not cfg['mapping'].get('fix_geo_decoder', True)
        # fallback vectors may be pinned for reproducible tests; default: drawn per call like the reference
        self.fixed_fallback = None

    def _fallbacks(self, device):
        if self.fixed_fallback is not None:
            return self.fixed_fallback
        # drawn on every call, geometry first (decoder.py:170-171, 387-388)
        fb_geo = torch.zeros([32], device=device).normal_(mean=0, std=0.01)
        fb_col = torch.zeros([32], device=device).normal_(mean=0, std=0.01)
        return fb_geo, fb_col

    def _linspace_rows(self, near, far, steps, device):
        """rows of torch.linspace(near, far_i, steps) for a per-ray far bound (one linspace per distinct value)."""
        far_u, row = torch.unique(far.reshape(-1), return_inverse=True)
        tab = torch.stack([torch.linspace(near, f, steps=steps, device=device) for f in far_u])
        return tab[row]

    def sample_depths(self, npc, rays_o, rays_d, gt_depth, far):
        """Per-ray sample depths for a batch that holds pixels without sensor depth (Renderer.py:126-170).
        `far`: 0-dim tensor (one batch) or [R] (render_img: per 3000-ray batch).  Returns z_vals [R,S] and the
        mask of rays close to the cloud (False only for depth-less rays that sample_near_pcl rejects)."""
        S = self.N_surface
        dev = gt_depth.device
        R = gt_depth.shape[0]
        t = torch.linspace(0.0, 1.0, steps=S, device=dev)
        gd = gt_depth.reshape(-1, 1).repeat(1, S)
        z = self.near_end_surface * gd * (1. - t) + self.far_end_surface * gd * t
        near_mask = torch.ones(R, device=dev, dtype=torch.bool)
        idx = torch.nonzero(gt_depth <= 0).flatten()
        if idx.numel():
            far_h = far[idx] if far.dim() else far
            if self.sample_near_pcl:
                zz, inv = npc.sample_near_pcl(rays_o[idx].detach().clone(), rays_d[idx].detach().clone(),
                                              self.near_end, far_h, S)
                near_mask[idx[inv]] = False
            elif far.dim():
                zz = self._linspace_rows(self.near_end, far_h, S, dev)
            else:
                zz = torch.linspace(self.near_end, far_h, steps=S, device=dev).repeat(idx.numel(), 1)
            z[idx] = zz
        return z.contiguous(), near_mask

    def render_batch_ray(self, npc, decoders, rays_d, rays_o, device, stage, gt_depth=None,
                         npc_geo_feats=None, npc_col_feats=None, is_tracker=False, cloud_pos=None,
                         dynamic_r_query=None, exposure_feat=None, far=None):
        """Renderer.render_batch_ray (Renderer.py:77-202).  `cloud_pos` is accepted for signature
        compatibility; positions live in the npc's device index.  `far` (not in the reference signature) lets
        render_img hand over the per-ray far bound of the reference's 3000-ray batches."""
        N_rays = rays_o.shape[0]
        if gt_depth is None:               # render over 10 m when no depth is available at all (:123-127)
            gt_depth = torch.zeros(N_rays, device=rays_o.device)
            far = torch.tensor(10.0, device=rays_o.device) if far is None else far
        gt_depth = gt_depth.detach().reshape(-1).float().contiguous()
        z_vals = near_mask = None
        if N_rays and not bool((gt_depth > 0).all()):
            if far is None:
                far = torch.minimum(5 * gt_depth.mean(), torch.max(gt_depth * 1.2)).float()
            z_vals, near_mask = self.sample_depths(npc, rays_o, rays_d, gt_depth, far)
        color = stage == 'color'
        flags = _lib.STAGE_COLOR if color else 0
        theta = P_.pack_master(decoders)
        pts_grad = bool(is_tracker and (rays_o.requires_grad or rays_d.requires_grad))
        if not is_tracker and (rays_o.requires_grad or rays_d.requires_grad):
            raise NotImplementedError("ray gradients are produced only with is_tracker=True (as every reference "
                                      "caller does: Tracker.py:151-157, Mapper.py:515-521 with BA)")
        if pts_grad:
            flags |= _lib.PTS_GRAD
        if npc_geo_feats.requires_grad or (npc_col_feats is not None and npc_col_feats.requires_grad):
            flags |= _lib.FEAT_GRAD
        if color and theta.requires_grad and not self.skip_decoder_grads:
            # colour-decoder gradients only: the geometry decoder is frozen in every shipped config
            # (mapping.fix_geo_decoder, point_slam.yaml:47); its gradient is returned as zeros.
            if self.geo_decoder_trainable:
                raise NotImplementedError("geometry-decoder parameter gradients are not produced "
                                          "(fix_geo_decoder=True in all reference configs)")
            flags |= _lib.PARAM_GRAD
        affine = None
        if color and self.encode_exposure:
            if exposure_feat is not None:
                # MLP_exposure: 8 -> 128 -> 12, one vector per batch (decoder.py:243-258,432-438): stays in torch
                affine = decoders.color_decoder.mlp_exposure(exposure_feat)
                flags |= _lib.HAS_AFFINE
            else:
                flags |= _lib.NO_SIGMOID     # mapper applies the per-frame affine outside (Mapper.py:530-548)
        rq = None
        if self.use_dynamic_radius:
            rq = dynamic_r_query.detach().reshape(-1).float().contiguous()
        fb_geo, fb_col = self._fallbacks(rays_o.device)
        m = dict(npc=npc,  # keeps the native context alive until backward has run
                 handle=npc.handle, gt_depth=gt_depth, r_query=rq, Bcol=P_.color_embed_B(decoders).to(rays_o.device)
                 .float().contiguous(), fb_geo=fb_geo.float().contiguous(), fb_col=fb_col.float().contiguous(),
                 flags=flags, coef=self.sigmoid_coefficient, z_vals=z_vals)
        if npc_col_feats is None:
            npc_col_feats = npc_geo_feats
        depth, var, rgb, valid = _RenderFn.apply(rays_o, rays_d, npc_geo_feats, npc_col_feats, theta, affine, m)
        if near_mask is not None:
            valid = valid & near_mask                                     # Renderer.py:198
            if not self.sample_near_pcl:
                depth = torch.where(gt_depth > 0, depth, torch.zeros_like(depth))   # Renderer.py:200-201
        return depth, var, rgb, valid

    def render_img(self, npc, decoders, c2w, device, stage, gt_depth=None,
                   npc_geo_feats=None, npc_col_feats=None,
                   dynamic_r_query=None, cloud_pos=None, exposure_feat=None):
        """Renderer.render_img (Renderer.py:204-283).  The reference renders 3000-ray batches, and the far bound
        of depth-less pixels is a statistic of their batch (:108-112): it is computed per such batch here and the
        image is then rendered in a few large launches."""
        with torch.no_grad():
            H, W = self.H, self.W
            u = torch.arange(W, device=device, dtype=torch.float32)
            v = torch.arange(H, device=device, dtype=torch.float32)
            vv, uu = torch.meshgrid(v, u, indexing="ij")
            dirs = torch.stack([(uu - self.cx) / self.fx, -(vv - self.cy) / self.fy, -torch.ones_like(uu)], -1)
            rays_d = torch.sum(dirs[..., None, :] * c2w[:3, :3], -1).reshape(-1, 3)
            rays_o = c2w[:3, -1].expand(rays_d.shape)
            n = H * W
            B = self.ray_batch_size
            if gt_depth is None:
                gd = torch.zeros(n, device=device)
                far = torch.full((n,), 10.0, device=device)
            else:
                gd = gt_depth.reshape(-1).float()
                far = torch.empty(n, device=device)
                for i in range(0, n, B):                     # the reference's batch statistics (:108-112)
                    g = gd[i:i + B]
                    far[i:i + B] = torch.minimum(5 * g.mean(), torch.max(g * 1.2))
            depth = torch.zeros(n, device=device, dtype=torch.float64)
            unc = torch.zeros(n, device=device, dtype=torch.float64)
            color = torch.zeros(n, 3, device=device)
            rq = dynamic_r_query.reshape(-1) if self.use_dynamic_radius else None
            big = B * 16
            for i in range(0, n, big):
                s = slice(i, min(i + big, n))
                d, u_, c, _ = self.render_batch_ray(
                    npc, decoders, rays_d[s], rays_o[s].contiguous(), device, stage, gt_depth=gd[s],
                    npc_geo_feats=npc_geo_feats, npc_col_feats=npc_col_feats, cloud_pos=cloud_pos,
                    dynamic_r_query=rq[s] if rq is not None else None, exposure_feat=exposure_feat, far=far[s])
                depth[s], unc[s], color[s] = d.double(), u_.double(), c
            return depth.reshape(H, W), unc.reshape(H, W), color.reshape(H, W, 3)
