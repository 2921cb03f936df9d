#!/bin/bash
# round 3, call U: host enqueue time against device time of the fused loops
mkdir -p gpurun_out
cd /tmp && export TMPDIR=/tmp && cd "$GRAFT_REPO_ROOT"
timeout 300 python tools/host_enqueue_probe.py 2>/dev/null | tail -1 | tee gpurun_out/r03_host_enqueue.json
