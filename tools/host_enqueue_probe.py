"""How far ahead of the device is the host inside psl_map_iters / psl_track_iters?  Wall time of the CALL (it returns once
everything is enqueued) against the time until the stream has drained, on the bench world (1 M points, base mix).
If the two are close the loop is launch-bound on the host and the device idles between launches."""
import os, sys, json, time, types
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
import bench as B

args = types.SimpleNamespace(gpus=1, steps=4, warmup=0, points=1_000_000, engine="native", mix="base", width=640,
                             height=480, exchange_every=2, no_cpu_baseline=True, no_kernel_timing=True, saturated_map=False)
dev = torch.device("cuda:0")
cfg, cam, slam, frames, cams0, every = B.build_world(args, 0, 1, dev)
fr = frames[0]
window = slam.keyframes[-4:] + [fr]
sel, row_map = slam.frustum_select(fr, fr.c2w)
out = {}
for name, fn in (("map_400x1000px", lambda: slam._map_native(window, sel, row_map, 400, 200)),
                 ("track_20x200px", lambda: slam.track(fr, cams0[0]))):
    fn(); torch.cuda.synchronize()
    res = []
    for _ in range(3):
        t0 = time.perf_counter(); fn(); t1 = time.perf_counter(); torch.cuda.synchronize(); t2 = time.perf_counter()
        res.append((round((t1 - t0) * 1e3, 3), round((t2 - t0) * 1e3, 3)))
    out[name] = dict(call_returns_ms=[r[0] for r in res], stream_drained_ms=[r[1] for r in res])
print(json.dumps(out))
