"""psl_track_iters under launch structures 0 / 2 / 3 on the trained 300 k-point world of tests/test_hip_fullsize.py (noisy depth with holes):
per-iteration losses and end poses must agree bit for bit (same draws, same process).  usage: python tools/track_structs_probe.py [n_pix n_iters]"""
import os, sys, json
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
from point_slam_amd import _lib, synthetic as syn
from point_slam_amd.config import default_config
from point_slam_amd.slam import Frame, HipSLAM, camera_tensor_from_c2w

n_pix = int(sys.argv[1]) if len(sys.argv) > 1 else 2000
n_iters = int(sys.argv[2]) if len(sys.argv) > 2 else 60
dev = torch.device("cuda:0")
cfg = default_config()
cam = syn.intrinsics(640, 480)
torch.manual_seed(77)
s = HipSLAM(cfg, cam, device="cuda:0", max_points=500_000, engine="native")
s.seed_points(syn.seed_cloud(cam, 300_000, n_views=48, seed=77))


def frame_at(t, idx, **kw):
    c2w = syn.pose(t, dev)
    depth, color = syn.render_frame(cam, c2w, **kw)
    r_add, r_q = syn.dynamic_radii(color, cfg)
    return Frame(idx, depth, color, r_add, r_q, c2w)


for k, t in enumerate((196.0, 200.0, 204.0)):
    kf = frame_at(t, k)
    s.map(kf, kf.c2w, n_iters=100, fixed_iters=True)
    s.keyframes.append(kf)
g = torch.Generator(device=dev).manual_seed(23)
fr = frame_at(201.0, 9, noise=0.005, dropout=0.02, gen=g)
tr = cfg["tracking"]
tr["sample_with_color_grad"] = False
eh, ew = tr["ignore_edge_H"], tr["ignore_edge_W"]
truth = camera_tensor_from_c2w(fr.c2w).cpu()
cam0 = truth.clone()
cam0[4:] += torch.tensor([0.009, -0.008, 0.010])
cam0[:4] += torch.tensor([0.0, 0.0012, -0.0009, 0.0008])
gi = torch.Generator(device="cpu").manual_seed(41)
hi = (cam["H"] - 2 * eh) * (cam["W"] - 2 * ew)
pix = torch.randint(hi, (n_iters, n_pix), generator=gi, dtype=torch.int32).to(dev).contiguous()
fb = torch.zeros(n_iters, 2, 32).normal_(mean=0, std=0.01, generator=gi).to(dev).contiguous()
L = _lib.lib()
out = {}
for ver in (0, 2, 3, 3, 2):
    _lib.check(L.psl_debug_option(b"track_fused", ver))
    best = s._track_native(fr, cam0, n_iters, n_pix, draws=(pix, fb)).cpu()
    torch.cuda.synchronize()
    out.setdefault(ver, []).append((best.clone(), s.last_cam.cpu().clone(), s.last_losses.cpu().clone()))
_lib.check(L.psl_debug_option(b"track_fused", 3))
ref = out[0][0]
rep = {}
for ver, runs in out.items():
    for i, r in enumerate(runs):
        dl = (r[2][:, 0] - ref[2][:, 0]).abs() / ref[2][:, 0].abs()
        rep[f"v{ver}_{i}"] = dict(loss_rel_max=float(dl.max()), first_diff_iter=int((dl > 0).nonzero()[0]) if (dl > 0).any() else -1,
                                  end_abs=float((r[1] - ref[1]).abs().max()), best_abs=float((r[0] - ref[0]).abs().max()),
                                  n_active_diff=float((r[2][:, 3] - ref[2][:, 3]).abs().max()))
print(json.dumps(dict(n_pix=n_pix, n_iters=n_iters, **rep), indent=1))
