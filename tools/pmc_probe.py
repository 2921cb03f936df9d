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
a = ap.parse_args()
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
