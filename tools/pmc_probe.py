"""Small fixed workload for PMC collection (rocprofv3 --pmc serialises every dispatch, ~30 ms each, so the full
bench is far too long): 1 M seeded points, 2 tracking iterations (200 px) and 4 mapping iterations (1000 px,
2 geometry-stage + 2 colour-stage) of the base mix -- the same launches bench.py times."""
import os, sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
from point_slam_amd import synthetic as syn
from point_slam_amd.config import default_config
from point_slam_amd.slam import Frame, HipSLAM, camera_tensor_from_c2w

dev = torch.device("cuda:0")
cfg = default_config()
cam = syn.intrinsics(640, 480)
torch.manual_seed(1219)
s = HipSLAM(cfg, cam, device="cuda:0", max_points=1_300_000, engine="native")
s.seed_points(syn.seed_cloud(cam, 1_000_000, n_views=64, seed=1219))
frames = []
for t in (170.0, 185.0, 200.0):
    c2w = syn.pose(t, dev)
    d, c = syn.render_frame(cam, c2w)
    ra, rq = syn.dynamic_radii(c, cfg)
    frames.append(Frame(int(t), d, c, ra, rq, c2w))
s.keyframes = frames[:2]
fr = frames[2]
s.track(fr, camera_tensor_from_c2w(fr.c2w).to(dev), n_iters=2, n_pix=200)
sel, row_map = s.frustum_select(fr, fr.c2w)
s._map_native(frames, sel, row_map, 4, 333)
torch.cuda.synchronize()
print("n_sel", int(sel.shape[0]))
