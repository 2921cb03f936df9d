"""Forward decode time with and without the saved-activation stores (drop-in render_batch_ray under no_grad / with gradients),
25 000 samples per launch, 1 M points: how much of the colour-stage forward is the 16 KB per sample of saved rows."""
import os, sys, json, types
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
import bench as B
from point_slam_amd import _lib
from tests.test_hip_fullsize import _rays, _render

args = types.SimpleNamespace(gpus=1, steps=1, warmup=0, points=1_000_000, engine="native", mix="base", width=640,
                             height=480, exchange_every=2, no_cpu_baseline=True, no_kernel_timing=True, saturated_map=True)
dev = torch.device("cuda:0")
cfg, cam, slam, frames, cams0, every = B.build_world(args, 0, 1, dev)
w = dict(cfg=cfg, cam=cam, slam=slam, frame=frames[0], dev=dev)
import os as _os
MODES = tuple(_os.environ.get("SAVE_PROBE_MODES", "no_grad,grad").split(","))
for n in (1000, 5000, 25000):
    ro, rd, gd, gc, rq = _rays(w, n, 5)
    for mode in MODES:
        for rep in range(2):
            if rep == 1: _lib.check(_lib.lib().psl_profile_enable(slam.npc.handle, 1))
            for k in range(5):
                if mode == "no_grad": _render(w, ro, rd, gd, rq)
                else: _render(w, ro, rd, gd, rq, grads=(torch.ones(n, device=dev), torch.ones(n, 3, device=dev)))
            torch.cuda.synchronize()
        prof = B.kernel_profile(slam)
        _lib.check(_lib.lib().psl_profile_enable(slam.npc.handle, 0))
        row = dict(samples=5 * n, mode=mode)
        for k in ("decode_fwd", "decode_bwd", "dw_gemm"):
            v = prof[k]
            if v["launches"]: row[k + "_us"] = round(v["ms"] * 1e3 / v["launches"], 1)
        print(json.dumps(row), flush=True)
