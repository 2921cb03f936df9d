"""Host wall time of the stages of HipSLAM.map (no synchronisation added): where the calling thread spends a mapped frame before
psl_map_iters is enqueued.  Bench world, base mix; every stage's own synchronisations are inside its number."""
import os, sys, json, time, types
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
import bench as B

args = types.SimpleNamespace(gpus=1, steps=4, warmup=0, points=1_000_000, engine="native", mix="base", width=640,
                             height=480, exchange_every=2, no_cpu_baseline=True, no_kernel_timing=True, saturated_map=False)
dev = torch.device("cuda:0")
cfg, cam, slam, frames, cams0, every = B.build_world(args, 0, 1, dev)
T = {}


def timed(name, fn):
    def w(*a, **k):
        t0 = time.perf_counter()
        r = fn(*a, **k)
        T.setdefault(name, []).append((time.perf_counter() - t0) * 1e3)
        return r
    return w


for nm in ("select_window", "add_points", "mapping_iters", "frustum_select", "_map_native", "_draws"):
    setattr(slam, nm, timed(nm, getattr(slam, nm)))
rows = []
for i in range(5, 45):
    fr = frames[i % len(frames)]
    slam.track(fr, cams0[i % len(cams0)])
    if i % every == 0:
        t0 = time.perf_counter()
        slam.map(fr, fr.c2w)
        t1 = time.perf_counter()
        torch.cuda.synchronize()
        t2 = time.perf_counter()
        rows.append(dict(map_call_ms=round((t1 - t0) * 1e3, 3), drained_ms=round((t2 - t0) * 1e3, 3)))
print(json.dumps(dict(stages_ms={k: [round(x, 3) for x in v[-6:]] for k, v in T.items()}, calls=rows[-6:]), indent=1))
