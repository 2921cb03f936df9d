"""Small fixed workload for PMC collection (rocprofv3 --pmc serialises every dispatch, ~30 ms each, so the full
bench is far too long): seeded points, 2 tracking iterations and 4 mapping iterations (2 geometry-stage + 2 colour-stage)
at the launch sizes of the chosen iteration mix -- the same launches bench.py times.

usage: python tools/pmc_probe.py [--mix base|replica|tum|scannet] [--points N] [--width W] [--height H]
Writes gpurun_out/pmc_probe_meta.json: the work-item counts of the tracker's and the mapper's decode launches, which is how
tools/pmc_traffic.py tells the two uses of one kernel symbol apart."""
import argparse, json, os, sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
from point_slam_amd import synthetic as syn
from point_slam_amd.config import MIXES, default_config
from point_slam_amd.slam import Frame, HipSLAM, camera_tensor_from_c2w

ap = argparse.ArgumentParser()
ap.add_argument("--mix", default="base")
ap.add_argument("--points", type=int, default=1_000_000)
ap.add_argument("--width", type=int, default=640)
ap.add_argument("--height", type=int, default=480)
ap.add_argument("--calibrate", action="store_true",
                help="only the known-traffic kernels (psl_selftest_traffic): what FETCH_SIZE / WRITE_SIZE report for a streaming "
                     "read / write, a 128-byte row gather and a row-atomic scatter of known size, tables far beyond the L2")
a = ap.parse_args()
if a.calibrate:
    from point_slam_amd import _lib
    L = _lib.lib()
    dev = torch.device("cuda:0")
    n_rows = 8_000_000                                  # 1 GiB table of 128-byte rows: beyond the 256-MiB Infinity Cache
    table = torch.zeros(n_rows, 32, device=dev)
    scratch = torch.zeros(1024, device=dev)
    g = torch.Generator(device="cpu").manual_seed(3)
    n_g = 2_000_000
    rows_unique = torch.randperm(n_rows, generator=g)[:n_g].int().to(dev)              # every row once: no reuse in any cache
    # the hot path's reuse: ~40 k (sample, neighbour) pairs over ~20 k distinct rows of a 128-MB table that fits the Infinity Cache
    rows_hot = torch.randint(0, 1_000_000, (n_g,), generator=g).int().to(dev)
    st = _lib.stream_ptr()
    for rep in range(3):
        _lib.check(L.psl_selftest_traffic(0, _lib.ptr(table), None, _lib.ptr(scratch), n_rows * 8, st))
        _lib.check(L.psl_selftest_traffic(1, _lib.ptr(table), None, None, n_rows * 8, st))
        _lib.check(L.psl_selftest_traffic(2, _lib.ptr(table), _lib.ptr(rows_unique), _lib.ptr(scratch), n_g, st))
        _lib.check(L.psl_selftest_traffic(3, _lib.ptr(table), _lib.ptr(rows_unique), None, n_g, st))
        _lib.check(L.psl_selftest_traffic(2 | 0x100, _lib.ptr(table), _lib.ptr(rows_hot), _lib.ptr(scratch), n_g, st))     # grid 4064: "hot"
        _lib.check(L.psl_selftest_traffic(3 | 0x100, _lib.ptr(table), _lib.ptr(rows_hot), None, n_g, st))
    torch.cuda.synchronize()
    known = dict(stream_read_bytes=n_rows * 128, stream_write_bytes=n_rows * 128, gather_bytes=n_g * 128,
                 scatter_rmw_bytes=n_g * 128, grids={"1048576": "every row once, 1-GiB table", "1040384": "random rows of a 1 M-row (128-MB) table: the hot path's reuse"}, repeats=3)
    os.makedirs("gpurun_out", exist_ok=True)
    json.dump(known, open("gpurun_out/pmc_calibration_known.json", "w"))
    print(json.dumps(known))
    sys.exit(0)
dev = torch.device("cuda:0")
cfg = MIXES[a.mix](default_config())
cam = syn.intrinsics(a.width, a.height)
torch.manual_seed(1219)
s = HipSLAM(cfg, cam, device="cuda:0", max_points=int(a.points * 1.3), engine="native")
s.seed_points(syn.seed_cloud(cam, a.points, n_views=64, seed=1219))
tr, mp = cfg["tracking"], cfg["mapping"]
W = min(mp["mapping_window_size"], 5 if a.mix == "base" else 3 + mp["mapping_window_size"] // 2)
frames = []
for i in range(W):
    t = 200.0 - 15.0 * (W - 1 - i)
    c2w = syn.pose(t, dev)
    d, c = syn.render_frame(cam, c2w)
    ra, rq = syn.dynamic_radii(c, cfg)
    frames.append(Frame(int(t), d, c, ra, rq, c2w))
s.keyframes = frames[:-1]
fr = frames[-1]
s.track(fr, camera_tensor_from_c2w(fr.c2w).to(dev), n_iters=2, n_pix=tr["pixels"])
sel, row_map = s.frustum_select(fr, fr.c2w)
ppf = mp["pixels"] // W
s._map_native(frames, sel, row_map, 4, ppf)
torch.cuda.synchronize()
tiles = lambda n_rays: (5 * n_rays + 15) // 16
meta = dict(mix=a.mix, points=a.points, track_rays=tr["pixels"], map_rays=ppf * W, window=W,
            track_fwd_items=2 * tiles(tr["pixels"]) * 512, map_fwd_items=2 * tiles(ppf * W) * 512, n_sel=int(sel.shape[0]))
os.makedirs("gpurun_out", exist_ok=True)
json.dump(meta, open("gpurun_out/pmc_probe_meta.json", "w"))
print(json.dumps(meta))
