"""GPU: `python bench.py --gpus 2` starts by itself (no torchrun on the command line) and runs the frame-parallel path --
two ranks, the periodic exchange of new points / touched feature rows / the colour decoder -- to the end.  On a one-GPU
box both ranks share cuda:0 over gloo (PSL_BENCH_SHARE_GPU=1; RCCL refuses two ranks on one device); on a multi-GPU node
the same command runs one rank per GPU over RCCL."""
import json
import os
import subprocess
import sys

import pytest
import torch

from tests.helpers import ROOT
from tests.test_hip_parity import report

pytestmark = pytest.mark.gpu


@pytest.mark.parametrize("merge", ["mean", "owner"])
def test_bench_self_launches_two_ranks(merge):
    env = dict(os.environ)
    if torch.cuda.device_count() < 2:
        env["PSL_BENCH_SHARE_GPU"] = "1"
    cmd = [sys.executable, os.path.join(ROOT, "bench.py"), "--gpus", "2", "--steps", "6", "--warmup", "5", "--points", "200000",
           "--exchange-every", "1", "--no-kernel-timing", "--merge", merge]
    p = subprocess.run(cmd, env=env, capture_output=True, text=True, timeout=900, cwd=ROOT)
    assert p.returncode == 0, p.stderr[-4000:]
    lines = [ln for ln in p.stdout.splitlines() if ln.startswith("{")]
    assert len(lines) == 1, p.stdout[-2000:]
    out = json.loads(lines[0])
    assert out["n_gpus"] == 2 and out["scaling"] == "weak" and out["steps"] == 6
    assert out["config"]["parallelism"] == "frame-parallel x2"
    pr = out["config"]["per_rank"]
    assert [r["rank"] for r in pr] == [0, 1]
    assert pr[0]["mapped"] >= 2 and pr[0]["added"] > 0 and pr[1]["added"] > 0
    # after the last exchange both replicas hold the same map (count and a checksum of the geometry features)
    assert pr[0]["points_after_final_exchange"] == pr[1]["points_after_final_exchange"]
    assert pr[0]["feat_checksum"] == pr[1]["feat_checksum"]
    assert out["config"]["replicas_identical_after_exchange"] is True
    assert out["value"] > 0
    # the line describes the multi-rank run by itself: backend, transport actually used, merge rule, per-rank exchange times
    # and the held-out render loss after the last exchange (identical replicas: same map,
    # each rank renders its own last frame)
    c = out["config"]
    assert c["merge_rule"] == merge and c["process_group_backend"] in ("gloo", "nccl")
    assert c["rccl_ranks"] == (2 if c["process_group_backend"] == "nccl" else 0)
    assert "torch.distributed" in c["exchange_transport"]             # native RCCL is opt-in (PSL_NATIVE_RCCL=1)
    for r in pr:
        assert r["exchange_ms_p50"] is not None and r["exchange_ms_p100"] >= r["exchange_ms_p50"]
        la = r["render_loss_after_final_exchange"]
        assert "error" not in la and la["valid_frac"] > 0.5 and la["depth_l1_m"] < 0.1
    report(test="bench_two_ranks", merge=merge, value=out["value"], ms_per_step=out["ms_per_step"], per_rank=pr,
           shared_gpu=env.get("PSL_BENCH_SHARE_GPU") == "1")
