"""Wall time of consecutive HipSLAM.map calls (no tracking in between) on the bench's world: is a mapping call slower when it
follows another one directly?  usage: python tools/map_repeat_probe.py [n_calls]"""
import os, sys, time, types
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
import bench as B

args = types.SimpleNamespace(gpus=1, steps=20, warmup=5, points=1_000_000, engine="native", mix=os.environ.get("MIX", "base"), width=640,
                             height=480, saturated_map=False, exchange_every=None, exchange_every_keyframes=10)
dev = torch.device("cuda:0")
cfg, cam, slam, frames, cams0, every = B.build_world(args, 0, 1, dev)
n = int(sys.argv[1]) if len(sys.argv) > 1 else 6
state = dict(mapped=0, added=0)
for i in range(5):
    B.run_step(i, slam, frames, cams0, every, cfg, 1, args, state)
torch.cuda.synchronize()
for k in range(n):
    fr = frames[6 + k]
    t0 = time.perf_counter()
    slam.map(fr, fr.gt_c2w, n_iters=cfg["mapping"]["iters"], fixed_iters=True)
    torch.cuda.synchronize()
    print(f"call {k}: {1e3 * (time.perf_counter() - t0):8.2f} ms  {slam.last_map}", flush=True)
    if os.environ.get("TRACK_BETWEEN") == "1":
        slam.track(frames[6 + k], cams0[6 + k]); torch.cuda.synchronize()
