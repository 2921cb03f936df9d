"""Per-phase cycle counts of the decode kernels (PSL_DEBUG_PHASES=1)."""
import os, sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
os.environ["PSL_DEBUG_PHASES"] = "1"
import torch
from tests.test_hip_slam import _scene, _slam
from point_slam_amd.slam import camera_tensor_from_c2w
dev = torch.device("cuda:0")
cfg, cam, frames, pts = _scene(dev, n_pts=120000)
s = _slam(cfg, cam, "native", dev)
s.seed_points(pts)
cam0 = camera_tensor_from_c2w(frames[1].c2w)
s.track(frames[1], cam0, n_iters=3, n_pix=200)
torch.cuda.synchronize()
s.keyframes = [frames[0]]
sel, row_map = s.frustum_select(frames[2], frames[2].c2w)
s._map_native([frames[0], frames[1], frames[2]], sel, row_map, 6, 333)
torch.cuda.synchronize()
