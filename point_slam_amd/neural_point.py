"""HipNeuralPointCloud: drop-in for the reference's NeuralPointCloud
(src/neural_point.py:9-277) -- same method names, argument meaning and return
conventions -- backed by the uniform-grid index of libpointslam_hip.so instead
of FAISS-GPU, with device-resident positions instead of Python lists.

Deviations, all documented in DESIGN.md:
  * the search is exact (FAISS IVF nprobe=4 is approximate); `nlist`/`nprobe` are ignored;
  * find_neighbors_faiss returns (inf, -1) in slots beyond the query radius;
  * cloud_pos() returns a [N,3] float32 CPU tensor instead of a list of lists: every reference caller wraps it
    at once -- np.array(...) (Mapper.py:131,760), torch.tensor(..., device=...) (Mapper.py:336-337,
    Tracker.py:198-199) -- or stores it (Logger.py:25), and all of these accept a CPU tensor.
    cloud_pos_device() is the no-copy-to-host variant the native loops and tests use;
  * features live in ONE pre-allocated [max_points, 32] store per set; geo_feats / col_feats are views of its first
    N rows, so appending points writes the new rows in place instead of re-concatenating 128 MB tensors
    (neural_point.py:155-159 does torch.cat per call).
"""
from __future__ import annotations

import ctypes as C

import torch

from . import _lib


def _kept_indices(keep_u8: torch.Tensor, n_kept: int) -> torch.Tensor:
    """Indices of the set flags, their number known on the host: no device-to-host synchronisation."""
    if n_kept == 0:
        return torch.zeros(0, dtype=torch.int64, device=keep_u8.device)
    try:
        return torch.nonzero_static(keep_u8, size=n_kept).reshape(-1)
    except (RuntimeError, NotImplementedError, AttributeError):
        return torch.nonzero(keep_u8).reshape(-1)


class _HostCount(int):
    """Return value of add_neural_points: the reference returns a 0-d tensor or an int (neural_point.py:91-92, callers use it
    in arithmetic and int()); the count is already on the host, so it is an int that also answers .item() / int() / float()
    without a device round trip."""

    def __new__(cls, v, device=None):
        o = super().__new__(cls, int(v))
        o.device = device
        return o

    def item(self):
        return int(self)


class HipNeuralPointCloud(object):
    def __init__(self, cfg, max_points: int = 4_000_000, device=None):
        self.cfg = cfg
        self.c_dim = cfg['model']['c_dim']
        self.device = device or cfg['mapping']['device']
        self.use_dynamic_radius = cfg['use_dynamic_radius']
        self.nn_num = cfg['pointcloud']['nn_num']
        self.radius_add = cfg['pointcloud']['radius_add']
        self.radius_min = cfg['pointcloud']['radius_min']
        self.radius_query = cfg['pointcloud']['radius_query']
        self.fix_interval_when_add_along_ray = cfg['pointcloud']['fix_interval_when_add_along_ray']
        if self.fix_interval_when_add_along_ray:
            raise NotImplementedError("fix_interval_when_add_along_ray=True is not used by any shipped config")
        self.N_surface = cfg['rendering']['N_surface']
        self.N_add = cfg['pointcloud']['N_add']
        if self.N_add != 3:
            raise NotImplementedError("N_add must be 3")
        self.near_end_surface = cfg['pointcloud']['near_end_surface']
        self.far_end_surface = cfg['pointcloud']['far_end_surface']
        self._input_pos = []          # device tensors [k,3], one per add call (neural_point.py:123)
        self._input_rgb = []          # device tensors [k,3], colour * 255 (neural_point.py:109,124)
        self.geo_feats = None
        self.col_feats = None
        self._geo_store = None        # [max_points, c_dim] backing stores; geo_feats / col_feats are views [:N]
        self._col_store = None
        self._rad_store = None        # [max_points] add-radius of the location a point belongs to (multi-GPU dedupe)
        self._max_points = int(max_points)
        self.keyframe_dict = []

        pc = cfg['pointcloud']
        max_r = pc['radius_add_max'] * pc['radius_query_ratio'] if self.use_dynamic_radius else pc['radius_query']
        max_r = max(max_r, pc['radius_add'], pc['radius_query'] if not self.use_dynamic_radius else 0.0)
        c = _lib.psl_config(n_surface=self.N_surface, nn_num=self.nn_num, c_dim=self.c_dim,
                            min_nn_num=pc['min_nn_num'],
                            near_end_surface=cfg['rendering']['near_end_surface'],
                            far_end_surface=cfg['rendering']['far_end_surface'],
                            radius_query=self.radius_query, max_query_radius=max_r,
                            encode_rel_pos=1 if cfg['model']['encode_rel_pos_in_col'] else 0,
                            max_points=max_points,
                            nn_weighting={"distance": 0, "expo": 1}[pc.get('nn_weighting', 'distance')])       # decoder.py:89,288
        dev = torch.device(self.device)
        self._dev_index = dev.index if dev.index is not None else 0
        h = C.c_void_p()
        _lib.check(_lib.lib().psl_create(self._dev_index, C.byref(c), C.byref(h)), "psl_create")
        self._h = h
        self._trained = False
        self._cfgs = c
        torch.manual_seed(cfg["setup_seed"])  # neural_point.py:42 (setup_seed) -- RNG for the feature init

    def __del__(self):
        h = getattr(self, "_h", None)
        if h is not None and _lib is not None:
            try:
                _lib.lib().psl_destroy(h)
            except Exception:
                pass
            self._h = None

    # ---- getters (neural_point.py:44-73) -------------------------------------------------
    @property
    def handle(self):
        return self._h

    def __reduce__(self):
        # the reference shares its cloud between the tracker and mapper PROCESSES through a BaseManager proxy
        # (src/Point_SLAM.py:93-97,189-207); a native context (device buffers, grid index) belongs to one process
        raise TypeError("HipNeuralPointCloud cannot be pickled or sent to another process: run tracker and mapper in one "
                        "process per GPU (INTEGRATION.md section 2)")

    def cloud_pos_device(self, first=0, count=None):
        """Positions [count,3] of points [first, first+count) as a device tensor (no host copy)."""
        n = self.pts_num()
        count = n - first if count is None else count
        out = torch.empty(count, 3, device=self.device, dtype=torch.float32)
        if count:
            _lib.check(_lib.lib().psl_points_download_range(self._h, int(first), int(count), _lib.ptr(out),
                                                            _lib.stream_ptr()), "psl_points_download_range")
        return out

    def cloud_pos(self, index=None):
        """neural_point.py:44-47.  A CPU float32 tensor [N,3] (see the module docstring): np.array(...),
        torch.tensor(..., device=...), .tolist() and indexing all behave as on the reference's list of lists."""
        if index is not None:
            return self.cloud_pos_device(int(index), 1)[0].cpu()
        return self.cloud_pos_device().cpu()

    def input_pos(self):
        """neural_point.py:49-50: surface points of every added location, list of [x,y,z]."""
        return torch.cat(self._input_pos).cpu().tolist() if self._input_pos else []

    def input_rgb(self):
        return torch.cat(self._input_rgb).cpu().tolist() if self._input_rgb else []

    def pts_num(self):
        return _lib.lib().psl_points_count(self._h)

    def index_train(self, xb):
        assert torch.is_tensor(xb), 'use tensor to train FAISS index'
        self._build()
        return True

    def index_ntotal(self):
        return self.pts_num()

    def get_radius_query(self):
        return self.radius_query

    def get_geo_feats(self):
        # a detached VIEW of the store (no copy): in-place writes by a caller reach the cloud -- as with the reference's
        # shared tensor -- but autograd history never attaches to the store itself (the reference's BaseManager proxy
        # hands out a detached copy, neural_point.py:64-67)
        return self.geo_feats.detach() if self.geo_feats is not None else None

    def get_col_feats(self):
        return self.col_feats.detach() if self.col_feats is not None else None

    def update_geo_feats(self, feats, indices=None):
        assert torch.is_tensor(feats), 'use tensor to update features'
        if indices is not None:
            self.geo_feats[indices] = feats.detach().clone()
        else:
            assert feats.shape[0] == self.geo_feats.shape[0], 'feature shape[0] mismatch'
            self.geo_feats.copy_(feats.detach())          # stays a view of the pre-allocated store

    def update_col_feats(self, feats, indices=None):
        assert torch.is_tensor(feats), 'use tensor to update features'
        if indices is not None:
            self.col_feats[indices] = feats.detach().clone()
        else:
            assert feats.shape[0] == self.col_feats.shape[0], 'feature shape[0] mismatch'
            self.col_feats.copy_(feats.detach())

    def _append_feats(self, gnew, cnew, radius=None):
        """Write new feature rows behind the current ones in the pre-allocated stores (no torch.cat of the map)."""
        n_old = 0 if self.geo_feats is None else self.geo_feats.shape[0]
        k = gnew.shape[0]
        if self._geo_store is None:
            self._geo_store = torch.empty(self._max_points, self.c_dim, device=self.device, dtype=torch.float32)
            self._col_store = torch.empty(self._max_points, self.c_dim, device=self.device, dtype=torch.float32)
            self._rad_store = torch.empty(self._max_points, device=self.device, dtype=torch.float32)
        if n_old + k > self._max_points:
            raise _lib.PslError(f"feature capacity {self._max_points} exceeded ({n_old} + {k})")
        self._geo_store[n_old:n_old + k] = gnew
        self._col_store[n_old:n_old + k] = cnew
        self._rad_store[n_old:n_old + k] = self.radius_add if radius is None else radius
        self.geo_feats = self._geo_store[:n_old + k]
        self.col_feats = self._col_store[:n_old + k]

    def point_radius(self, first=0, count=None):
        """Add-radius of the location each point was created for (dynamic r_add of its pixel, radius_min for
        gradient pixels, radius_add otherwise): what the cross-rank dedupe of the multi-GPU exchange tests with."""
        n = self.pts_num()
        count = n - first if count is None else count
        return self._rad_store[first:first + count]

    def locations_free(self, loc, radius):
        """True where a surface point has NO existing neural point strictly inside its radius: the admission test
        of add_neural_points (neural_point.py:116-121) for externally supplied locations."""
        _, _, cnt = self.find_neighbors_faiss(loc, step='add', dynamic_radius=radius)
        return cnt == 0

    def count_within(self, loc, radius, idx_limit=None):
        """Number of points with index < idx_limit (default: all) strictly inside radius_i of loc_i (psl_dedupe_count):
        the admission test of add_neural_points restricted to a prefix of the cloud -- the multi-GPU merge tests foreign
        locations against the base map on the index as it stands."""
        q = loc.detach().float().contiguous()
        rad = radius.detach().float().reshape(-1).contiguous()
        n = q.shape[0]
        cnt = torch.zeros(n, device=q.device, dtype=torch.int32)
        lim = self.pts_num() if idx_limit is None else int(idx_limit)
        _lib.check(_lib.lib().psl_dedupe_count(self._h, _lib.ptr(q), _lib.ptr(rad), float(self.radius_add), n, lim,
                                               _lib.ptr(cnt), _lib.stream_ptr()), "psl_dedupe_count")
        return cnt

    def dedupe_blocks(self, rec, pos_col, rad_col, block_first, keep):
        """Cross-rank half of the admission test (psl_dedupe_blocks): `rec` [3 L, W] fp32 holds one record per point (xyz
        at columns pos_col.., radius at rad_col), locations in rank blocks block_first[b]..block_first[b+1]; `keep` uint8
        [L] is updated in place: a location loses its flag when a point of a still-kept location of an earlier block lies
        strictly inside its radius."""
        assert rec.dtype == torch.float32 and rec.is_contiguous() and keep.dtype == torch.uint8 and keep.is_contiguous()
        import ctypes as C
        bf = (C.c_int32 * len(block_first))(*[int(b) for b in block_first])
        w = rec.shape[1]
        _lib.check(_lib.lib().psl_dedupe_blocks(self._h, rec.data_ptr() + 4 * pos_col, w, rec.data_ptr() + 4 * rad_col, w, bf,
                                                len(block_first) - 1, _lib.ptr(keep), _lib.stream_ptr()), "psl_dedupe_blocks")
        return keep

    # ---- state upload (checkpoint / multi-GPU merge) ------------------------------------
    def set_points(self, pos: torch.Tensor, geo_feats: torch.Tensor = None, col_feats: torch.Tensor = None):
        """Replace the cloud by `pos` [N,3] (device tensor) and rebuild the index."""
        L = _lib.lib()
        _lib.check(L.psl_points_reset(self._h))
        if geo_feats is not None:
            self.geo_feats = self.col_feats = None
        self.append_points(pos, geo_feats, col_feats)

    def truncate(self, n: int):
        """Keep the first n points (positions and features); used by the multi-GPU merge.  O(1): views shrink."""
        _lib.check(_lib.lib().psl_points_truncate(self._h, int(n)), "psl_points_truncate")
        if self.geo_feats is not None:
            self.geo_feats = self._geo_store[:n]
            self.col_feats = self._col_store[:n]

    def append_points(self, pos, geo_feats=None, col_feats=None, build=True, radius=None):
        pos = pos.to(self.device, torch.float32).contiguous()
        n = pos.shape[0]
        if n:
            _lib.check(_lib.lib().psl_points_append(self._h, _lib.ptr(pos), n, _lib.stream_ptr()), "append")
        if geo_feats is not None:
            self._append_feats(geo_feats.to(self.device).float(), col_feats.to(self.device).float(), radius)
        if build:
            self._build()

    def _build(self):
        _lib.check(_lib.lib().psl_index_build(self._h, _lib.stream_ptr()), "psl_index_build")
        self._trained = True

    # ---- add_neural_points (neural_point.py:91-167) ---------------------------------------
    def add_neural_points(self, batch_rays_o, batch_rays_d, batch_gt_depth, batch_gt_color,
                          train=False, is_pts_grad=False, dynamic_radius=None, return_new=False):
        n = batch_rays_o.shape[0]
        if not n:
            return 0
        ro = batch_rays_o.detach().float().contiguous()
        rd = batch_rays_d.detach().float().contiguous()
        dep = batch_gt_depth.detach().float().contiguous()
        rad = dynamic_radius.detach().float().contiguous() if dynamic_radius is not None else None
        r_scalar = self.radius_min if is_pts_grad else self.radius_add      # neural_point.py:199-205
        keep = torch.empty(n, dtype=torch.uint8, device=ro.device)
        kept = C.c_int(0)
        n_before = self.pts_num()
        _lib.check(_lib.lib().psl_add_points_sync(self._h, _lib.ptr(ro), _lib.ptr(rd), _lib.ptr(dep), _lib.ptr(rad),
                                                  float(r_scalar), n, float(self.near_end_surface),
                                                  float(self.far_end_surface), _lib.ptr(keep), C.byref(kept),
                                                  _lib.stream_ptr()), "psl_add_points_sync")
        n_new = 3 * kept.value
        # features ~ N(0, 0.1^2), geometry first then colour (neural_point.py:149-159)
        gnew = torch.zeros([n_new, self.c_dim], device=self.device).normal_(mean=0, std=0.1)
        cnew = torch.zeros([n_new, self.c_dim], device=self.device).normal_(mean=0, std=0.1)
        # rows of the kept locations: the count is already on the host (psl_add_points_sync), so the index list is built
        # without another synchronisation (boolean indexing would stall the host once per use: three times here)
        kidx = _kept_indices(keep, kept.value)
        rad_new = (rad.index_select(0, kidx) if rad is not None else torch.full((kept.value,), float(r_scalar), device=ro.device))
        self._append_feats(gnew, cnew, rad_new.repeat_interleave(3))
        # surface point and colour*255 of every kept location (neural_point.py:109,113,123-124; exported by
        # Mapper.py:757-758 and the checkpoint)
        self._input_pos.append((ro + rd * dep[:, None]).index_select(0, kidx))
        self._input_rgb.append((batch_gt_color.detach().to(ro.device) * 255).index_select(0, kidx).float())
        if kept.value:
            self._build()
        if return_new:
            return kept.value, keep.bool(), n_before
        return _HostCount(kept.value, self.device)

    # ---- find_neighbors_faiss (neural_point.py:169-215) -----------------------------------
    def find_neighbors_faiss(self, pos, step='add', retrain=False, is_pts_grad=False, dynamic_radius=None):
        assert step in ['add', 'query']
        q = pos.detach().float().contiguous()
        n = q.shape[0]
        if step == 'query':
            radius = self.radius_query
        else:
            radius = self.radius_min if is_pts_grad else self.radius_add
        rad = None
        if dynamic_radius is not None:
            assert pos.shape[0] == dynamic_radius.shape[0], 'shape mis-match for input points and dynamic radius'
            rad = dynamic_radius.detach().float().reshape(-1).contiguous()
        D = torch.empty(n, self.nn_num, device=q.device, dtype=torch.float32)
        I = torch.empty(n, self.nn_num, device=q.device, dtype=torch.int64)
        cnt = torch.empty(n, device=q.device, dtype=torch.int32)
        _lib.check(_lib.lib().psl_knn(self._h, _lib.ptr(q), _lib.ptr(rad), float(radius), n, _lib.ptr(D), _lib.ptr(I),
                                      _lib.ptr(cnt), _lib.stream_ptr()), "psl_knn")
        return D, I, cnt

    def sample_near_pcl(self, rays_o, rays_d, near, far, num):
        """NeuralPointCloud.sample_near_pcl (src/neural_point.py:217-277) for pixels without sensor depth: march
        25 steps from `near` to `far`, a step hits when a neural point lies inside radius_query
        (psl_near_pcl_hits); a ray with >= 2 hits is sampled `num` times between its first two hit steps, the
        others uniformly in [near, far] and flagged invalid.  `far` is a scalar / 0-dim tensor as in the
        reference, or a per-ray [n] tensor (render_img: one far bound per 3000-ray batch).

        Returns z_vals [n, num] float32 and invalid_mask [n] bool, both on the rays' device."""
        dev = rays_o.device
        n = rays_o.shape[0]
        steps = 25
        ro = rays_o.detach().float().contiguous()
        rd = rays_d.detach().float().contiguous()
        far_t = torch.as_tensor(far, dtype=torch.float32, device=dev)
        if far_t.dim() == 0:
            far_u, row = far_t.reshape(1), None
        else:
            far_u, row = torch.unique(far_t.reshape(-1), return_inverse=True)
            row = row.to(torch.int32).contiguous()
        z_steps = torch.stack([torch.linspace(near, f, steps=steps, device=dev) for f in far_u]).contiguous()
        hits = torch.empty(n, steps, device=dev, dtype=torch.uint8)
        _lib.check(_lib.lib().psl_near_pcl_hits(self._h, _lib.ptr(ro), _lib.ptr(rd), n, _lib.ptr(z_steps),
                                                _lib.ptr(row), steps, float(self.radius_query), _lib.ptr(hits),
                                                _lib.stream_ptr()), "psl_near_pcl_hits")
        hit = hits.bool()
        invalid = hit.sum(1) < 2
        # np.linspace(near, far, 25) in float64 (:247): arange * step + start, endpoint forced
        far64 = (far_u.double()[row.long()] if row is not None else far_u.double().expand(n))[:, None]
        ar = torch.arange(steps, device=dev, dtype=torch.float64)[None, :]
        z_sec = ar * ((far64 - near) / (steps - 1)) + near
        z_sec[:, -1] = far64[:, 0]
        order = torch.argsort((~hit).to(torch.int8), dim=1, stable=True)      # hit steps first, in step order
        a = torch.gather(z_sec, 1, order[:, 0:1])
        b = torch.gather(z_sec, 1, order[:, 1:2])
        an = torch.arange(num, device=dev, dtype=torch.float64)[None, :]
        seg = an * ((b - a) / (num - 1)) + a
        seg[:, -1] = b[:, 0]
        uni = an * ((far64 - near) / (num - 1)) + near
        uni[:, -1] = far64[:, 0]
        z = torch.where(invalid[:, None], uni, seg)
        return z.float(), invalid
