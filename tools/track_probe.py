"""Does the tracker hold a trajectory in closed loop on the synthetic scene?  (bench.py --track-only set-up question.)
Fixed cloud of N points, optionally trained first by mapping a set of keyframes at their true poses, then `n` frames tracked from
the constant-speed extrapolation of the tracker's own estimates; prints the translation error every 10 frames."""
import argparse, os, sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
from point_slam_amd import host_ops as H, synthetic as syn
from point_slam_amd.config import MIXES, default_config
from point_slam_amd.slam import Frame, HipSLAM

ap = argparse.ArgumentParser()
ap.add_argument("--points", type=int, default=50000)
ap.add_argument("--width", type=int, default=1200)
ap.add_argument("--height", type=int, default=680)
ap.add_argument("--mix", default="replica")
ap.add_argument("--train-frames", type=int, default=24)
ap.add_argument("--train-iters", type=int, default=300)
ap.add_argument("--frames", type=int, default=60)
ap.add_argument("--dt", type=float, default=2.0)
a = ap.parse_args()
dev = torch.device("cuda:0")
cfg = MIXES[a.mix](default_config())
cam = syn.intrinsics(a.width, a.height)
torch.manual_seed(1219)
s = HipSLAM(cfg, cam, device="cuda:0", max_points=a.points * 2 + 100000, engine="native")
n_fr = a.frames + 5
s.seed_points(syn.seed_cloud(cam, a.points, n_views=48, seed=1219, t0=190.0, dt=(a.dt * n_fr + 20.0) / 47.0))


def frame(i, t):
    c2w = syn.pose(t, dev)
    d, c = syn.render_frame(cam, c2w)
    ra, rq = syn.dynamic_radii(c, cfg)
    return Frame(i, d, c, ra, rq, c2w)


if a.train_frames:
    for k in range(a.train_frames):
        fr = frame(-1 - k, 195.0 + k * (a.dt * n_fr + 10.0) / max(a.train_frames - 1, 1))
        s.map(fr, fr.c2w, n_iters=a.train_iters, add=False, fixed_iters=True)
        s.keyframes.append(fr)
        if len(s.keyframes) > 12:
            s.keyframes.pop(0)
        if k % 6 == 0:
            torch.cuda.synchronize()
            print("train", k, "last losses", [round(float(x), 2) for x in s.last_losses[-3:, 0]], flush=True)
row4 = torch.tensor([[0.0, 0.0, 0.0, 1.0]], device=dev)
est = []
for i in range(a.frames):
    fr = frame(i, 200.0 + a.dt * i)
    if len(est) < 2:
        est.append(fr.c2w.clone()); continue
    cam0 = H.camera_tensor_from_c2w_device(H.const_speed_init(est[-1], est[-2]))
    best = s.track(fr, cam0)
    c2w = torch.cat([H.get_camera_from_tensor(best), row4], 0)
    est.append(c2w)
    if i % 10 == 0 or i == a.frames - 1:
        torch.cuda.synchronize()
        print(f"frame {i}: |t_est - t_gt| = {float((c2w[:3, 3] - fr.c2w[:3, 3]).norm()) * 100:.2f} cm, init error "
              f"{float((H.get_camera_from_tensor(cam0)[:3, 3] - fr.c2w[:3, 3]).norm()) * 100:.2f} cm, first/last loss {float(s.last_losses[0, 0]):.1f} / {float(s.last_losses[-1, 0]):.1f}", flush=True)
