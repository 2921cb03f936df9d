"""Decode / dW kernel time and roofline fraction against launch size (samples per launch), colour stage, 1 M points.
The headline workload launches 1 000-5 000 samples (latency regime); this shows where the kernels saturate."""
import os, sys, json, types
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
import bench as B
from point_slam_amd import _lib

args = types.SimpleNamespace(gpus=1, steps=1, warmup=0, points=1_000_000, engine="native", mix="base", width=640,
                             height=480, exchange_every=2, no_cpu_baseline=True, no_kernel_timing=True, saturated_map=True)
dev = torch.device("cuda:0")
cfg, cam, slam, frames, cams0, every = B.build_world(args, 0, 1, dev)
cfg["mapping"]["geo_iter_ratio"] = 0.0
fr = frames[0]
window = slam.keyframes[-4:] + [fr]
sel, row_map = slam.frustum_select(fr, fr.c2w)
out = []
for ppf in [40, 200, 1000, 2000, 5000, 13000]:
    n_it = 9
    slam._map_native(window, sel, row_map, 3, ppf)       # warm
    torch.cuda.synchronize()
    _lib.check(_lib.lib().psl_profile_enable(slam.npc.handle, 1))
    slam._map_native(window, sel, row_map, n_it, ppf)
    torch.cuda.synchronize()
    prof = B.kernel_profile(slam)
    _lib.check(_lib.lib().psl_profile_enable(slam.npc.handle, 0))
    P = 5 * ppf * len(window)
    row = dict(samples=P)
    for k in ("decode_fwd", "decode_bwd", "dw_gemm", "knn", "adam"):
        v = prof[k]
        if v["launches"]:
            us = v["ms"] * 1e3 / v["launches"]
            row[k + "_us"] = round(us, 1)
            if k in B.MFMA_CLASSES:
                row[k + "_frac"] = round(v["work"] / (v["ms"] * 1e-3) / 1e12 / B.PEAK_F32_MFMA_TFLOPS, 4)
    out.append(row)
    print(json.dumps(row), flush=True)
os.makedirs("gpurun_out", exist_ok=True)
json.dump(out, open("gpurun_out/roofline_sweep.json", "w"), indent=1)

# ---- the tracker's instantiations (pose gradient: k_decode_bwd2<true, true>), rays per launch = tracking.pixels of the mixes
from point_slam_amd.slam import camera_tensor_from_c2w
out_t = []
cam0 = camera_tensor_from_c2w(fr.c2w).to(dev)
for n_pix in [200, 1500, 5000, 10000]:
    slam.track(fr, cam0, n_iters=3, n_pix=n_pix)     # warm
    torch.cuda.synchronize()
    _lib.check(_lib.lib().psl_profile_enable(slam.npc.handle, 1))
    slam.track(fr, cam0, n_iters=9, n_pix=n_pix)
    torch.cuda.synchronize()
    prof = B.kernel_profile(slam)
    _lib.check(_lib.lib().psl_profile_enable(slam.npc.handle, 0))
    row = dict(samples=5 * n_pix, tracker=True)
    for k in ("decode_fwd_track", "decode_bwd_track", "knn"):
        v = prof[k]
        if v["launches"]:
            row[k + "_us"] = round(v["ms"] * 1e3 / v["launches"], 1)
            if k in B.MFMA_CLASSES:
                row[k + "_frac"] = round(v["work"] / (v["ms"] * 1e-3) / 1e12 / B.PEAK_F32_MFMA_TFLOPS, 4)
    out_t.append(row)
    print(json.dumps(row), flush=True)
json.dump(dict(mapper=out, tracker=out_t), open("gpurun_out/roofline_sweep.json", "w"), indent=1)
