#!/bin/bash
# round 3, call H: one-launch geometry-stage iteration -- loop parity (reference fixtures, 140-iteration oracle anchors,
# switch agreement), bench A/B
mkdir -p gpurun_out
cd /tmp && export TMPDIR=/tmp && cd "$GRAFT_REPO_ROOT"
rm -f gpurun_out/parity_report.jsonl
timeout 1200 python -m pytest tests/test_hip_loops.py tests/test_hip_slam.py -q -m gpu --durations=3 -x 2>&1 | tail -30 > gpurun_out/pytest_r3h.log; tail -8 gpurun_out/pytest_r3h.log
for v in 1 0 1 0; do
  PSL_GEO_FUSED=$v timeout 300 python bench.py --no-cpu-baseline 2>gpurun_out/r03_bench_h$v.err | tail -1 > gpurun_out/r03_bench_h$v.json
  echo "geo_fused=$v"; python tools/show_bench.py gpurun_out/r03_bench_h$v.json | grep -E "FPS|geo|composite_fwd|adam "
done
