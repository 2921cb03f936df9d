"""k-NN roofline: candidates examined (16-byte sorted-position records, psl_knn_candidates) per launch and per query,
16 B x candidates / kernel time against the L2 bandwidth, for the tracker-size launch (200 rays, one wavefront per
sample) and the mapper's block prefetch (64 iterations x 1 000 rays, one wavefront per ray), on the bench world."""
import os, sys, json, types
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
import bench as B
from point_slam_amd import _lib

L2_TBS = 34.5      # /opt/skills/guides/MI355X_MICROARCH.md: aggregate L2 bandwidth
args = types.SimpleNamespace(gpus=1, steps=4, warmup=0, points=1_000_000, engine="native", mix="base", width=640,
                             height=480, exchange_every=2, no_cpu_baseline=True, no_kernel_timing=True,
                             saturated_map="--saturated-map" in sys.argv)
dev = torch.device("cuda:0")
cfg, cam, slam, frames, cams0, every = B.build_world(args, 0, 1, dev)
L = _lib.lib()
out = []


def measure(label, fn, queries_per_launch, cls="knn"):
    fn()                                   # warm
    torch.cuda.synchronize()
    L.psl_knn_candidates(slam.npc.handle)
    _lib.check(L.psl_profile_enable(slam.npc.handle, 1))
    fn()
    torch.cuda.synchronize()
    prof = B.kernel_profile(slam)[cls]
    _lib.check(L.psl_profile_enable(slam.npc.handle, 0))
    cand = int(L.psl_knn_candidates(slam.npc.handle))
    n = prof["launches"]
    us = prof["ms"] * 1e3 / max(n, 1)
    row = dict(launch=label, launches=n, avg_us=round(us, 1), queries_per_launch=queries_per_launch,
               candidates_per_query=round(cand / max(n * queries_per_launch, 1), 1),
               l2_gbs=round(16.0 * cand / max(prof["ms"] * 1e-3, 1e-9) / 1e9, 1))
    row["frac_of_l2"] = round(row["l2_gbs"] / (L2_TBS * 1e3), 5)
    out.append(row)
    print(json.dumps(row), flush=True)


fr = frames[0]
measure("tracker 200 rays x 20 iterations", lambda: slam.track(fr, cams0[0]), 1000)
window = slam.keyframes[-4:] + [fr]
sel, row_map = slam.frustum_select(fr, fr.c2w)
measure("mapper prefetch 64 iterations x 1000 rays", lambda: slam._map_native(window, sel, row_map, 64, 200), 64 * 1000 * 5,
        cls="knn_prefetch")       # one block: looked up on the main stream, unthrottled
os.makedirs("gpurun_out", exist_ok=True)
json.dump(out, open("gpurun_out/knn_roofline.json", "w"), indent=1)
