"""Per-frame tracking / mapping loops on top of the HIP hot path.

`HipSLAM` plays the role of the reference's Tracker.run / Mapper.run bodies
(src/Tracker.py:203-394, src/Mapper.py:237-640) for ONE device: same
iteration counts, pixel budgets, stage / learning-rate schedule, point adding
and frustum feature selection, driven from Python but with two engines:

  engine="native": psl_track_iters / psl_map_iters -- the whole iteration loop runs as
                   back-to-back HIP kernels with no host synchronisation (fast path);
  engine="dropin": the reference's own loop structure in torch (sampling, loss, torch.optim.Adam)
                   calling HipRenderer.render_batch_ray -- what a user gets by swapping only the
                   Renderer / NeuralPointCloud classes (parity path).

Around the loops `HipSLAM.map` reproduces the per-mapped-frame logic of Mapper.optimize_map / Mapper.run:
uniform + colour-gradient point adding (Mapper.py:306-330), frustum feature selection (:120-168), keyframe
selection by overlap or at random (:170-235,263-276), the first-frame schedule (`iters_first`, `geo_iter_first`,
the `init` LR table, :420,424), the data-dependent iteration count (:404-406) and, for `model.encode_exposure`,
the per-frame exposure latents (:399-401,530-548; Tracker.py:305-311).  BA is not built (off in every config).
"""
from __future__ import annotations

import ctypes as C
from typing import List, Optional

import torch

from . import _lib, host_ops as H, params as P_
from .decoders import PointDecoders
from .neural_point import HipNeuralPointCloud
from .renderer import HipRenderer


class Frame:
    """One RGB-D frame resident on the device."""

    def __init__(self, idx, depth, color, r_add=None, r_query=None, c2w=None, exposure=None):
        self.idx = idx
        self.depth = depth.float().contiguous()
        self.color = color.float().contiguous()
        self.r_add = r_add.float().contiguous() if r_add is not None else None
        self.r_query = r_query.float().contiguous() if r_query is not None else None
        self.c2w = c2w            # [4,4] estimated pose (device tensor)
        self.exposure = exposure  # [8] exposure latent of a keyframe (model.encode_exposure)
        self.grad_mag = None      # [H,W] f64 colour-gradient magnitude, filled on demand

    def c2w_host(self) -> list:
        """The pose as 12 host floats (row-major 3x4): ONE device-to-host copy per pose, shared by everything a mapped frame
        needs it for on the host side (psl_frame_view of the mapping window, the frustum selection, the keyframe overlap test)
        -- each of those used to copy it by itself, and every copy stalls the host on all the work queued so far.  The cache
        holds a REFERENCE to the tensor it copied (an id() alone can be recycled by a new tensor at the same address, version 0
        again) plus its version counter; tensors without one (inference mode) are copied every time."""
        try:
            ver = self.c2w._version
        except Exception:
            ver = None
        if ver is None or getattr(self, "_c2w_ref", None) is not self.c2w or self._c2w_ver != ver:
            self._c2w_host = self.c2w[:3, :4].detach().float().cpu().reshape(-1).tolist()
            self._c2w_ref, self._c2w_ver = self.c2w, ver
        return self._c2w_host

    def view(self, pose: bool = True) -> _lib.psl_frame_view:
        """pose=False: the tracker reads the pose from its camera tensor on the device (psl_frame_view.c2w is "mapping
        only"); the device-to-host copy of c2w would stall the host on everything queued so far, once per frame."""
        v = _lib.psl_frame_view()
        v.depth = self.depth.data_ptr()
        v.color = self.color.data_ptr()
        v.r_query = self.r_query.data_ptr() if self.r_query is not None else None
        if pose and self.c2w is not None:
            h = self.c2w_host()
            for i in range(12):
                v.c2w[i] = h[i]
        return v


def camera_tensor_from_c2w(c2w: torch.Tensor) -> torch.Tensor:
    """get_tensor_from_camera (src/common.py:270-295): [quat(w,x,y,z), T]; host-side (scipy in the reference)."""
    from scipy.spatial.transform import Rotation
    import numpy as np
    m = c2w.detach().cpu().double().numpy()
    q = np.roll(Rotation.from_matrix(m[:3, :3]).as_quat(), 1)
    return torch.from_numpy(np.concatenate([q, m[:3, 3]])).float()


class HipSLAM:
    def __init__(self, cfg, cam: dict, device="cuda:0", max_points=2_500_000, engine="native", decoders=None):
        self.cfg, self.cam, self.device, self.engine = cfg, cam, torch.device(device), engine
        self.encode_exposure = bool(cfg["model"].get("encode_exposure", False))
        if self.encode_exposure and engine != "native":
            raise NotImplementedError("HipSLAM(engine='dropin') does not carry the per-frame exposure latents; use the "
                                      "native engine, or HipRenderer inside the reference's own Tracker/Mapper")
        cfgd = dict(cfg)
        cfgd["mapping"] = dict(cfg["mapping"], device=str(device))
        self.npc = HipNeuralPointCloud(cfgd, max_points=max_points, device=str(device))
        import types
        self.renderer = HipRenderer(cfg, None, types.SimpleNamespace(**cam))
        if decoders is None:
            torch.manual_seed(cfg["setup_seed"])
            decoders = PointDecoders(cfg)
        self.decoders = decoders.to(self.device)
        self.theta = P_.pack_master(self.decoders).detach().clone().contiguous()       # native engine's master blob
        self.Bcol = P_.color_embed_B(self.decoders).to(self.device).float().contiguous()
        self.keyframes: List[Frame] = []
        self.cam_intr = _lib.psl_cam_intr(H=cam["H"], W=cam["W"], fx=cam["fx"], fy=cam["fy"], cx=cam["cx"],
                                          cy=cam["cy"])
        self._ws_track = None
        self._ws_map = None
        self.last_losses = None
        self.map_step = dict(geo=0, col=0)
        self.n_mapped = 0
        self.sync = None          # FrameParallelSync of the multi-GPU mode: told which rows a mapped frame is about to train
        # torch 1.12 (the reference's env.yaml) keeps zero .grad tensors after zero_grad(): from the second mapped frame
        # on its Adam counts the geometry-stage iterations for the colour decoder as well.  "torch2" (None gradients,
        # what the fixtures of this repo were generated with) is the default; see psl_map_args.step0_params.
        self.adam_zero_grad_semantics = "torch2"
        if self.encode_exposure:
            # MLP_exposure weights live outside the master blob (decoder.py:243-258); shared latent as in
            # Point_SLAM.py:85-87 (N(0, 0.01^2))
            ex = self.decoders.color_decoder.mlp_exposure
            self.exposure_mlp = torch.cat([ex.linear1.weight.reshape(-1), ex.linear1.bias, ex.linear2.weight.reshape(-1),
                                           ex.linear2.bias]).detach().float().clone().contiguous()
            assert self.exposure_mlp.numel() == _lib.EXPOSURE_MLP_FLOATS
            self.exposure_feat = torch.zeros(cfg["model"]["exposure_dim"], device=self.device).normal_(0, 0.01)

    # ------------------------------------------------------------------ state
    def seed_points(self, pos: torch.Tensor, seed=1219):
        g = torch.Generator(device="cpu").manual_seed(seed)
        n = pos.shape[0]
        geo = torch.zeros(n, 32).normal_(0, 0.1, generator=g)
        col = torch.zeros(n, 32).normal_(0, 0.1, generator=g)
        self.npc.set_points(pos.to(self.device), geo.to(self.device), col.to(self.device))

    def sync_decoders_from_theta(self):
        """Write the native blob back into the nn.Module (checkpoint surface, Logger.py:22-40)."""
        with torch.no_grad():
            named = dict(self.decoders.named_parameters())
            for name, t in P_.unpack_master(self.theta).items():
                named[name].copy_(t)
            if self.encode_exposure:
                ex, m = self.decoders.color_decoder.mlp_exposure, self.exposure_mlp
                ex.linear1.weight.copy_(m[:1024].reshape(128, 8)); ex.linear1.bias.copy_(m[1024:1152])
                ex.linear2.weight.copy_(m[1152:2688].reshape(12, 128)); ex.linear2.bias.copy_(m[2688:2700])

    def sync_theta_from_decoders(self):
        """The opposite direction (after load_state_dict): repack the native blob and the colour Fourier matrix."""
        self.theta = P_.pack_master(self.decoders).detach().clone().contiguous()
        self.Bcol = P_.color_embed_B(self.decoders).to(self.device).float().contiguous()
        if self.encode_exposure:
            ex = self.decoders.color_decoder.mlp_exposure
            self.exposure_mlp = torch.cat([ex.linear1.weight.reshape(-1), ex.linear1.bias, ex.linear2.weight.reshape(-1),
                                           ex.linear2.bias]).detach().float().clone().contiguous()

    def save_checkpoint(self, path, idx=0, **kw):
        """Logger.log schema (src/utils/Logger.py:20-40) with the decoders AS TRAINED by the native loops."""
        from . import checkpoint as CK
        self.sync_decoders_from_theta()
        torch.cuda.synchronize()
        CK.save_checkpoint(path, self.npc, self.decoders, idx=idx, **kw)

    def load_checkpoint(self, path_or_dict):
        from . import checkpoint as CK
        ck = torch.load(path_or_dict, map_location=self.device, weights_only=False) if isinstance(path_or_dict, str) \
            else path_or_dict
        CK.load_neural_point_cloud(self.npc, ck)
        CK.load_decoders(self.decoders, ck)
        self.sync_theta_from_decoders()
        return ck

    # ------------------------------------------------------------------ tracking
    def init_pose(self, est_c2w: list) -> torch.Tensor:
        """Camera tensor the tracker starts frame len(est_c2w) from (Tracker.py:283-290): constant-speed extrapolation of
        the last two ESTIMATED poses."""
        c2w = H.const_speed_init(est_c2w[-1], est_c2w[-2] if len(est_c2w) >= 2 else None)
        return camera_tensor_from_c2w(c2w)

    def init_pose_device(self, cam_prev: torch.Tensor, cam_prev2: Optional[torch.Tensor] = None) -> torch.Tensor:
        """The same on the device, camera tensors in, camera tensor out (psl_pose_const_speed): one launch, no host copy of
        a pose -- what a closed track -> track loop calls once per frame."""
        out = torch.empty(7, device=self.device)
        # the kernel dereferences these pointers: bring host tensors (what init_pose / camera_tensor_from_c2w return) and tensors
        # of another device onto this one, and refuse anything that is not a camera tensor [qw qx qy qz tx ty tz]
        a = cam_prev.detach().to(self.device, torch.float32).contiguous()
        b = cam_prev2.detach().to(self.device, torch.float32).contiguous() if cam_prev2 is not None else None
        if a.numel() != 7 or (b is not None and b.numel() != 7):
            raise ValueError("init_pose_device: camera tensors have 7 elements (quaternion + translation)")
        _lib.check(_lib.lib().psl_pose_const_speed(_lib.ptr(a), _lib.ptr(b), _lib.ptr(out), _lib.stream_ptr()),
                   "psl_pose_const_speed")
        self._keep_pose = (a, b)
        return out

    def track(self, frame: Frame, cam0: torch.Tensor, n_iters=None, n_pix=None) -> torch.Tensor:
        """Optimise the pose of `frame` from the initial camera tensor cam0 [7]; returns the lowest-loss
        camera tensor (candidate_cam_tensor, Tracker.py:347-350)."""
        tr = self.cfg["tracking"]
        n_iters = n_iters or tr["iters"]
        n_pix = n_pix or tr["pixels"]
        if self.engine == "native":
            return self._track_native(frame, cam0, n_iters, n_pix)
        return self._track_dropin(frame, cam0, n_iters, n_pix)

    def _draws(self, n_iters, n_idx, hi):
        idx = torch.randint(hi, (n_iters, n_idx), device=self.device, dtype=torch.int32)
        fb = torch.zeros(n_iters, 2, 32, device=self.device).normal_(mean=0, std=0.01)
        return idx, fb

    def grad_mag(self, frame: Frame):
        if frame.grad_mag is None:
            from . import frame_ops
            _, _, frame.grad_mag = frame_ops.dynamic_radius_maps(frame.color, self.cfg, with_grad_mag=True)
        return frame.grad_mag

    def color_grad_draws(self, frame: Frame, n_iters, n_pix):
        """sample_with_color_grad (Tracker.py:281-285,115-128): the top 15*n_pix gradient pixels inside the border with
        sensor depth, then n_pix of them WITHOUT replacement per iteration (np.random.choice(..., replace=False))."""
        from . import frame_ops
        tr, cam = self.cfg["tracking"], self.cam
        eh, ew = tr["ignore_edge_H"], tr["ignore_edge_W"]
        sel, _ = frame_ops.get_selected_index_with_grad(self.npc, eh, cam["H"] - eh, ew, cam["W"] - ew, n_pix, frame.color,
                                                        gt_depth=frame.depth, depth_limit=bool(tr.get("depth_limit", False)),
                                                        grad_mag=self.grad_mag(frame))
        if sel.numel() < n_pix:
            raise RuntimeError(f"sample_with_color_grad: only {sel.numel()} candidate pixels for {n_pix} samples")
        pick = torch.rand(n_iters, sel.numel(), device=self.device).topk(n_pix, dim=1).indices
        return sel[pick].to(torch.int32).contiguous()

    def _exposure_block(self, feats, lr_mlp, lr_feat=0.001):
        """psl_exposure_args over the shared MLP blob and `feats` ([8] or [F,8]); returns (struct, keep-alive)."""
        adam = torch.zeros(2, _lib.EXPOSURE_MLP_FLOATS + _lib.EXPOSURE_DIM, device=self.device)
        e = _lib.psl_exposure_args(mlp=self.exposure_mlp.data_ptr(), feats=feats.data_ptr(), adam=adam.data_ptr(),
                                   lr_mlp=lr_mlp, lr_feat=lr_feat, step0=0)
        return e, (adam, feats)

    def _track_native(self, frame, cam0, n_iters, n_pix, draws=None):
        L = _lib.lib()
        tr, cam = self.cfg["tracking"], self.cam
        eh, ew = tr["ignore_edge_H"], tr["ignore_edge_W"]
        full = bool(tr.get("sample_with_color_grad", False))
        if draws is not None:
            idx, fb = draws
        elif full:
            idx = self.color_grad_draws(frame, n_iters, n_pix)
            fb = torch.zeros(n_iters, 2, 32, device=self.device).normal_(mean=0, std=0.01)
        else:
            idx, fb = self._draws(n_iters, n_pix, (cam["H"] - 2 * eh) * (cam["W"] - 2 * ew))
        need = int(L.psl_track_ws_floats(n_pix))
        if self._ws_track is None or self._ws_track.numel() < need:
            self._ws_track = torch.empty(need, device=self.device)
        cam_t = cam0.to(self.device).float().clone().contiguous()
        adam = torch.zeros(14, device=self.device)
        losses = torch.empty(n_iters, 4, device=self.device)
        best = torch.empty(8, device=self.device)
        a = _lib.psl_track_args()
        a.cam = self.cam_intr
        a.edge_h, a.edge_w, a.n_iters, a.n_pix = eh, ew, n_iters, n_pix
        a.pix_idx, a.fallback = idx.data_ptr(), fb.data_ptr()
        a.frame = frame.view(pose=False)
        a.cam_tensor, a.adam_state, a.step0 = cam_t.data_ptr(), adam.data_ptr(), 0
        a.lr_T = tr["lr"]
        a.lr_quat = tr["lr"] * 0.2 if tr["separate_LR"] else tr["lr"]
        a.w_color, a.handle_dynamic, a.use_color = tr["w_color_loss"], int(tr["handle_dynamic"]), \
            int(tr["use_color_in_tracking"])
        a.sigmoid_coef = self.cfg["rendering"]["sigmoid_coef_tracker"]
        a.geo_feats, a.col_feats = self.npc.geo_feats.data_ptr(), self.npc.col_feats.data_ptr()
        a.params, a.col_embed_B = self.theta.data_ptr(), self.Bcol.data_ptr()
        a.ws, a.loss_out, a.best_out = self._ws_track.data_ptr(), losses.data_ptr(), best.data_ptr()
        a.pix_full_image = 1 if full else 0
        keep_ex = None
        if self.encode_exposure:
            # Tracker.py:269-271: latent cloned from the shared one.  BOTH optimiser branches add the latent and
            # mlp_exposure.parameters() with lr 0.001 (separate_LR: Tracker.py:307-311; single camera tensor: :316-320)
            ex_feat = self.exposure_feat.clone().contiguous()
            ex, keep_ex = self._exposure_block(ex_feat, 0.001)
            a.exposure = C.pointer(ex)
        _lib.check(L.psl_track_iters(self.npc.handle, C.byref(a), _lib.stream_ptr()), "psl_track_iters")
        if self.encode_exposure:
            self.exposure_feat = ex_feat                 # exposure_feat_shared[0] = ... (Tracker.py:385-387)
        self._keep = (idx, fb, cam_t, adam, keep_ex)     # alive until the stream has consumed them
        self.last_losses = losses
        self.last_cam = cam_t
        return best[:7]

    def _track_dropin(self, frame, cam0, n_iters, n_pix, draws=None):
        """Tracker.run inner loop (Tracker.py:296-350) with HipRenderer in place of Renderer."""
        tr, cam, dev = self.cfg["tracking"], self.cam, self.device
        eh, ew = tr["ignore_edge_H"], tr["ignore_edge_W"]
        idx, fb = draws if draws is not None else self._draws(n_iters, n_pix, (cam["H"] - 2 * eh) * (cam["W"] - 2 * ew))
        cam_t = cam0.to(dev).float()
        quad = cam_t[:4].clone().requires_grad_(True)
        T = cam_t[4:].clone().requires_grad_(True)
        lr = tr["lr"]
        opt = torch.optim.Adam([{"params": [T], "lr": lr}, {"params": [quad], "lr": lr * 0.2 if tr["separate_LR"] else lr}])
        self.renderer.sigmoid_coefficient = self.cfg["rendering"]["sigmoid_coef_tracker"]
        self.renderer.skip_decoder_grads = True
        best, best_loss, losses = None, float("inf"), []
        for it in range(n_iters):
            camera_tensor = torch.cat([quad, T], 0)
            opt.zero_grad()
            c2w = H.get_camera_from_tensor(camera_tensor)
            u, v = H.pixels_from_flat_index(idx[it].long(), eh, cam["H"] - eh, ew, cam["W"] - ew)
            ro, rd = H.get_rays_from_uv(u, v, c2w, cam["fx"], cam["fy"], cam["cx"], cam["cy"])
            ui, vi = u.long(), v.long()
            gd, gc = frame.depth[vi, ui], frame.color[vi, ui]
            rq = frame.r_query[vi, ui] if self.cfg["use_dynamic_radius"] else None
            keep = gd > 0
            ro, rd, gd, gc = ro[keep], rd[keep], gd[keep], gc[keep]
            rq = rq[keep] if rq is not None else None
            inl = H.depth_inlier_mask(gd)
            ro, rd, gd, gc = ro[inl], rd[inl], gd[inl], gc[inl]
            rq = rq[inl] if rq is not None else None
            self.renderer.fixed_fallback = (fb[it, 0], fb[it, 1])
            d, var, rgb, _ = self.renderer.render_batch_ray(
                self.npc, self.decoders, rd, ro, dev, "color", gt_depth=gd, npc_geo_feats=self.npc.geo_feats,
                npc_col_feats=self.npc.col_feats, is_tracker=True, dynamic_r_query=rq)
            loss, geo, col, mask = H.tracker_loss(d, var, rgb, gd, gc, tr["handle_dynamic"],
                                                  tr["use_color_in_tracking"], tr["w_color_loss"])
            loss.backward()
            opt.step()
            lv = float(loss)
            losses.append(lv)
            if lv < best_loss:
                best_loss, best = lv, camera_tensor.detach().clone()
        self.renderer.fixed_fallback = None
        self.last_losses = losses
        self.last_cam = torch.cat([quad, T]).detach()
        return best

    # ------------------------------------------------------------------ mapping
    def _add_batch(self, frame, c2w, idx, is_pts_grad):
        cam = self.cam
        u, v = H.pixels_from_flat_index(idx, 0, cam["H"], 0, cam["W"])
        ro, rd = H.get_rays_from_uv(u, v, c2w, cam["fx"], cam["fy"], cam["cx"], cam["cy"])
        ui, vi = u.long(), v.long()
        gd, gc = frame.depth[vi, ui], frame.color[vi, ui]
        # depth_filter=True (Mapper.py:311-313): the native admission test drops rays without sensor depth itself (keep =
        # depth > 0 and no point inside the radius, in ray order), so the batch goes down uncompacted -- four boolean-index
        # operations, each a host synchronisation, per batch
        rad = frame.r_add[vi, ui] if self.cfg["use_dynamic_radius"] else None
        return int(self.npc.add_neural_points(ro.contiguous(), rd.contiguous(), gd, gc,
                                              is_pts_grad=is_pts_grad, dynamic_radius=rad))

    def add_points(self, frame: Frame, c2w: torch.Tensor, n_pixels=None, first=False):
        """Point adding of Mapper.optimize_map (Mapper.py:303-330): `pixels_adding` uniformly drawn pixels (scaled by
        (median depth / 2.5)^2, clamped to [1,3]x, on the first frame) against the per-pixel r_add map, then
        `pixels_based_on_color_grad` pixels from the top-gradient set against radius_min
        (get_samples_with_pixel_grad, common.py:186-222).  Returns the number of new LOCATIONS."""
        mp, cam, dev = self.cfg["mapping"], self.cam, self.device
        n = n_pixels or mp["pixels_adding"]
        if first:
            # Mapper.py:304-306: gt_depth.median() over the WHOLE image, sensor holes (zeros) included
            scale = float((frame.depth.median() / 2.5) ** 2)
            n = int(min(max(n * scale, n), 3 * n))
        idx = torch.randint(cam["H"] * cam["W"], (n,), device=dev)
        added = self._add_batch(frame, c2w, idx, False)
        n_grad = int(mp.get("pixels_based_on_color_grad", 0))
        if n_grad > 0:
            # get_sample_uv_with_grad (common.py:92-113): n_grad of the top 5*n_grad gradient pixels, without replacement
            from . import frame_ops
            sel, _ = frame_ops.get_selected_index_with_grad(self.npc, 0, cam["H"], 0, cam["W"], n_grad, frame.color,
                                                            ratio=5, grad_mag=self.grad_mag(frame))
            if sel.numel() >= n_grad:
                pick = torch.randperm(sel.numel(), device=dev)[:n_grad]
                added += self._add_batch(frame, c2w, sel[pick], True)
        return added

    def frustum_select(self, frame: Frame, c2w: torch.Tensor):
        """Mapper.get_mask_from_c2w (Mapper.py:120-168) on the device: returns (sel int32[n_sel], row_map int32[N]).
        Points whose bilinear depth lookup is 0 take the maximum over the PER-POINT lookups (:161-162), reduced on the
        device (depth_max = -1)."""
        L = _lib.lib()
        N = self.npc.pts_num()
        sel = torch.empty(N, dtype=torch.int32, device=self.device)
        row_map = torch.empty(N, dtype=torch.int32, device=self.device)
        if c2w is frame.c2w:
            h = frame.c2w_host() + [0.0, 0.0, 0.0, 1.0]         # the copy the frame already holds (one per pose)
        else:
            h = c2w.detach().float().cpu().reshape(-1).tolist()
        c2w_h = (C.c_float * 16)(*h)
        n_sel = C.c_int(0)
        _lib.check(L.psl_frustum_select_sync(self.npc.handle, c2w_h, self.cam_intr, _lib.ptr(frame.depth), -1.0,
                                             float(self.cfg["mapping"]["frustum_edge"]), _lib.ptr(sel),
                                             _lib.ptr(row_map), C.byref(n_sel), _lib.stream_ptr()),
                   "psl_frustum_select_sync")
        return sel[:n_sel.value], row_map

    def select_window(self, frame: Frame, c2w=None, size=None, method=None) -> List[Frame]:
        """Keyframe selection of Mapper.optimize_map (:263-276): mapping_window_size-2 keyframes among all but the last
        one -- at random ('global', random_select) or among those that overlap the current view ('overlap',
        keyframe_selection_overlap :170-235: 200 pixels x 8 frustum samples projected into every keyframe, the
        overlapping ones in random order) -- then the last keyframe and the current frame."""
        mp = self.cfg["mapping"]
        k = (size or mp["mapping_window_size"]) - 2
        method = method or mp.get("keyframe_selection_method", "overlap")
        win: List[Frame] = []
        if len(self.keyframes) > 0:
            older = self.keyframes[:-1]
            if older and k > 0:
                if method == "overlap" and c2w is not None:
                    from . import frame_ops
                    cam = self.cam
                    idx = torch.randint(cam["H"] * cam["W"], (200,), device=self.device)
                    u, v = H.pixels_from_flat_index(idx, 0, cam["H"], 0, cam["W"])
                    ro, rd = H.get_rays_from_uv(u, v, c2w, cam["fx"], cam["fy"], cam["cx"], cam["cy"])
                    gd = frame.depth[v.long(), u.long()]
                    # depth_filter=True (Mapper.py:190-192): pixels without sensor depth carry no samples -- the kernel skips
                    # rays with depth <= 0 and counts over the others, no compaction (and no host sync) here
                    ids = frame_ops.keyframe_selection_overlap(ro.contiguous(), rd.contiguous(), gd.contiguous(),
                                                               [f.c2w_host() for f in older], cam, k)
                    win += [older[int(i)] for i in ids]
                else:
                    perm = torch.randperm(len(older))[:k].tolist()
                    win += [older[i] for i in perm]
            win.append(self.keyframes[-1])
        win.append(frame)
        return win

    def mapping_iters(self, n_iters, pts_added, first):
        """Mapper.py:404-406: clip(iters * pts_added / 300, min_iter_ratio * iters, 2 * iters) after the first frame."""
        if first:
            return n_iters
        lo = int(self.cfg["mapping"].get("min_iter_ratio", 0.95) * n_iters)
        return int(min(max(int(n_iters * pts_added / 300), lo), 2 * n_iters))

    def map(self, frame: Frame, c2w: torch.Tensor, n_iters=None, add=True, first=False, fixed_iters=False):
        """One Mapper.optimize_map call.  first: the idx == 0 schedule (`iters_first`, `geo_iter_first`, `init` LRs)."""
        mp = self.cfg["mapping"]
        frame.c2w = c2w
        window = self.select_window(frame, c2w)
        added = self.add_points(frame, c2w, first=first) if add else 0
        n_iters = n_iters or (mp["iters_first"] if first else mp["iters"])
        if not fixed_iters:
            n_iters = self.mapping_iters(n_iters, added, first)
        n_geo = mp["geo_iter_first"] if first else int(n_iters * mp["geo_iter_ratio"])
        sel, row_map = self.frustum_select(frame, c2w)
        pix_per_frame = mp["pixels"] // len(window)
        if self.sync is not None:
            self.sync.note_rows(self.npc, sel)
        if self.engine == "native":
            self._map_native(window, sel, row_map, n_iters, pix_per_frame, n_geo=n_geo, first=first)
        else:
            self._map_dropin(window, sel, n_iters, pix_per_frame)
        self.n_mapped += 1
        self.last_map = dict(frame=frame.idx, added=int(added), n_sel=int(sel.shape[0]), n_iters=int(n_iters), n_geo=int(n_geo),
                             window=len(window), points=self.npc.pts_num())
        return added, int(sel.shape[0])

    def refine(self, frame: Frame, c2w: torch.Tensor, n_outer=5, n_iters=None):
        """End-of-run colour refinement (Mapper.run, Mapper.py:706-720,740-747; optimize_map :316,:346-360,:427-430):
        five optimize_map calls with twice the iterations over a window of twice the size drawn at random from ALL
        keyframes ('global'), no point adding, no frustum selection (every row of both feature sets sits in the optimiser),
        colour decoder frozen, geo_iter_ratio 0 (iteration 0 still runs stage 'geometry', :420-421), geometry lr 0 and
        colour lr = color_lr / 10 in both stages.  Native engine: psl_map_iters with sel = all rows."""
        if self.engine != "native":
            raise NotImplementedError("HipSLAM.refine runs on the native engine")
        mp = self.cfg["mapping"]
        frame.c2w = c2w
        n_iters = n_iters or 2 * mp["iters"]
        N = self.npc.pts_num()
        sel = torch.arange(N, dtype=torch.int32, device=self.device)
        if self.sync is not None:
            self.sync.note_rows(self.npc, sel)      # every row is trained: all of them belong to the next exchange
        lr = dict(geo_geo=0.0, geo_col=0.0, col=mp["stage"]["color"]["color_lr"] / 10.0, dec=mp["stage"]["color"]["decoders_lr"])
        for _ in range(n_outer):
            window = self.select_window(frame, None, size=2 * mp["mapping_window_size"], method="global")
            ppf = mp["pixels"] // len(window)
            self._map_native(window, sel, sel, n_iters, ppf, n_geo=0, lr=lr, train_decoder=False)
        return n_outer * n_iters

    def _map_native(self, window, sel, row_map, n_iters, ppf, draws=None, n_geo=None, first=False, lr=None,
                    train_decoder=None):
        L = _lib.lib()
        mp, cam, dev = self.cfg["mapping"], self.cam, self.device
        W = len(window)
        n = W * ppf
        idx, fb = draws if draws is not None else self._draws(n_iters, n, cam["H"] * cam["W"])
        need = int(L.psl_map_ws_floats(n, W))
        if self._ws_map is None or self._ws_map.numel() < need:
            self._ws_map = torch.empty(need, device=dev)
        n_sel = int(sel.shape[0])
        ncol = P_.color_floats()
        g_geo = torch.zeros(n_sel, 32, device=dev)
        g_col = torch.zeros(n_sel, 32, device=dev)
        adam_geo = torch.zeros(2, n_sel, 32, device=dev)
        adam_col = torch.zeros(2, n_sel, 32, device=dev)
        adam_par = torch.zeros(2, ncol, device=dev)
        losses = torch.empty(n_iters, 4, device=dev)
        views = (_lib.psl_frame_view * W)(*[f.view() for f in window])
        st = mp["init" if first else "stage"]
        a = _lib.psl_map_args()
        a.cam = self.cam_intr
        a.n_frames, a.pix_per_frame, a.n_iters = W, ppf, n_iters
        a.n_geo_iters = int(n_iters * mp["geo_iter_ratio"]) if n_geo is None else int(n_geo)
        a.frames = views
        a.pix_idx, a.fallback = idx.data_ptr(), fb.data_ptr()
        a.geo_feats, a.col_feats = self.npc.geo_feats.data_ptr(), self.npc.col_feats.data_ptr()
        a.params, a.col_embed_B = self.theta.data_ptr(), self.Bcol.data_ptr()
        a.sel_rows, a.row_map, a.n_sel = sel.data_ptr(), row_map.data_ptr(), n_sel
        a.g_geo, a.g_col = g_geo.data_ptr(), g_col.data_ptr()
        a.adam_geo, a.adam_col, a.adam_params = adam_geo.data_ptr(), adam_col.data_ptr(), adam_par.data_ptr()
        a.step0_geo, a.step0_col = 0, 0
        a.step0_params = (a.n_geo_iters + 1) if (self.adam_zero_grad_semantics == "torch1" and self.n_mapped > 0) else 0
        a.train_decoder = (0 if mp["fix_color_decoder"] else 1) if train_decoder is None else int(bool(train_decoder))
        a.lr_geo_geo_stage, a.lr_geo_color_stage = st["geometry"]["geometry_lr"], st["color"]["geometry_lr"]
        a.lr_col, a.lr_decoder = st["color"]["color_lr"], st["color"]["decoders_lr"]
        if lr is not None:          # colour refinement (Mapper.py:427-430)
            a.lr_geo_geo_stage, a.lr_geo_color_stage, a.lr_col, a.lr_decoder = lr["geo_geo"], lr["geo_col"], lr["col"], lr["dec"]
        a.w_color, a.sigmoid_coef = mp["w_color_loss"], self.cfg["rendering"]["sigmoid_coef_mapper"]
        a.ws, a.loss_out = self._ws_map.data_ptr(), losses.data_ptr()
        keep_ex = None
        if self.encode_exposure:
            # one latent per window frame; the current frame's is cloned from the shared one, optimised (lr 0.001) and
            # written back (Mapper.py:341-343,399-401,624-626)
            feats = torch.stack([(f.exposure if f.exposure is not None else self.exposure_feat).to(dev).float()
                                 for f in window[:-1]] + [self.exposure_feat.float()]).contiguous()
            ex, keep_ex = self._exposure_block(feats, st["color"]["decoders_lr"])
            a.exposure = C.pointer(ex)
        _lib.check(L.psl_map_iters(self.npc.handle, C.byref(a), _lib.stream_ptr()), "psl_map_iters")
        if self.encode_exposure:
            self.exposure_feat = feats[-1].clone()
            window[-1].exposure = feats[-1].clone()
        self._keep_map = (idx, fb, g_geo, g_col, adam_geo, adam_col, adam_par, views, sel, row_map, keep_ex)
        self.last_losses = losses

    def _map_dropin(self, window, sel, n_iters, ppf, draws=None):
        """Mapper.optimize_map joint_iter loop (Mapper.py:408-568) in torch, HipRenderer for the render."""
        mp, cam, dev = self.cfg["mapping"], self.cam, self.device
        W = len(window)
        idx, fb = draws if draws is not None else self._draws(n_iters, W * ppf, cam["H"] * cam["W"])
        sel_l = sel.long()
        npc_geo, npc_col = self.npc.get_geo_feats(), self.npc.get_col_feats()
        geo_p = npc_geo[sel_l].detach().clone().requires_grad_(True)
        col_p = npc_col[sel_l].detach().clone().requires_grad_(True)
        for p in self.decoders.parameters():
            p.requires_grad_(False)
        dec_params = [p for n, p in self.decoders.color_decoder.named_parameters() if "mlp_exposure" not in n]
        if not mp["fix_color_decoder"]:
            for p in dec_params:
                p.requires_grad_(True)
        opt = torch.optim.Adam([{"params": dec_params if not mp["fix_color_decoder"] else [], "lr": 0},
                                {"params": [geo_p], "lr": 0}, {"params": [col_p], "lr": 0}])
        self.renderer.sigmoid_coefficient = self.cfg["rendering"]["sigmoid_coef_mapper"]
        self.renderer.skip_decoder_grads = False
        n_geo = int(n_iters * mp["geo_iter_ratio"])
        losses = []
        for it in range(n_iters):
            npc_geo = npc_geo.detach().index_put((sel_l,), geo_p)
            npc_col = npc_col.detach().index_put((sel_l,), col_p)
            stage = "geometry" if it <= n_geo else "color"
            st = mp["stage"][stage]
            opt.param_groups[0]["lr"], opt.param_groups[1]["lr"], opt.param_groups[2]["lr"] = \
                st["decoders_lr"], st["geometry_lr"], st["color_lr"]
            opt.zero_grad()
            ros, rds, gds, gcs, rqs = [], [], [], [], []
            for f, fr in enumerate(window):
                u, v = H.pixels_from_flat_index(idx[it, f * ppf:(f + 1) * ppf].long(), 0, cam["H"], 0, cam["W"])
                ro, rd = H.get_rays_from_uv(u, v, fr.c2w.to(dev), cam["fx"], cam["fy"], cam["cx"], cam["cy"])
                ui, vi = u.long(), v.long()
                gd = fr.depth[vi, ui]
                keep = gd > 0
                ros.append(ro[keep]); rds.append(rd[keep]); gds.append(gd[keep]); gcs.append(fr.color[vi, ui][keep])
                if self.cfg["use_dynamic_radius"]:
                    rqs.append(fr.r_query[vi, ui][keep])
            ro, rd, gd, gc = torch.cat(ros), torch.cat(rds), torch.cat(gds), torch.cat(gcs)
            rq = torch.cat(rqs) if rqs else None
            inl = H.depth_inlier_mask(gd)
            ro, rd, gd, gc = ro[inl], rd[inl], gd[inl], gc[inl]
            rq = rq[inl] if rq is not None else None
            self.renderer.fixed_fallback = (fb[it, 0], fb[it, 1])
            d, var, rgb, valid = self.renderer.render_batch_ray(
                self.npc, self.decoders, rd, ro, dev, stage, gt_depth=gd, npc_geo_feats=npc_geo, npc_col_feats=npc_col,
                is_tracker=False, dynamic_r_query=rq)
            loss, geo, col, m = H.mapper_loss(d, rgb, valid, gd, gc, stage, mp["w_color_loss"])
            loss.backward()
            opt.step()
            losses.append(float(loss.detach()))
        self.renderer.fixed_fallback = None
        self.npc.update_geo_feats(geo_p, indices=sel_l)
        self.npc.update_col_feats(col_p, indices=sel_l)
        self.theta = P_.pack_master(self.decoders).detach().clone().contiguous()
        self.last_losses = losses
