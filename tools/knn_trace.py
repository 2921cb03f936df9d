#!/usr/bin/env python
"""Per-query cost distribution of the tracker's k-NN launches (PSL_KNN_TRACE=1): shader cycles, candidates and passes of
every query of 20 tracking iterations on the bench world; printed by the library to stderr."""
import os
import sys
import types

os.environ["PSL_KNN_TRACE"] = "1"
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch                                  # noqa: E402
import bench as B                              # noqa: E402
from point_slam_amd import _lib                # noqa: E402

args = types.SimpleNamespace(gpus=1, steps=4, warmup=0, points=1_000_000, engine="native", mix="base", width=640,
                             height=480, exchange_every=2, no_cpu_baseline=True, no_kernel_timing=True,
                             saturated_map=False)
dev = torch.device("cuda:0")
cfg, cam, slam, frames, cams0, every = B.build_world(args, 0, 1, dev)
L = _lib.lib()
fr = frames[0]
slam.track(fr, cams0[0])
torch.cuda.synchronize()
_lib.check(L.psl_debug_option(b"knn_trace_dump", 1))      # discard the warm-up
slam.track(fr, cams0[0])
torch.cuda.synchronize()
_lib.check(L.psl_debug_option(b"knn_trace_dump", 1))
rq = fr.r_query
print("r_query of the frame: min %.4f mean %.4f max %.4f; grid cell = max_query_radius / 4 = %.4f" %
      (float(rq.min()), float(rq.mean()), float(rq.max()), float(cfg["pointcloud"].get("radius_query_max", rq.max())) / 4),
      file=sys.stderr)
