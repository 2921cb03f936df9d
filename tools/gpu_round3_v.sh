#!/bin/bash
# round 3, call V: lazy Adam without dense catch-ups at the prefetch-block ends (work list across the boundary + round-robin slices)
mkdir -p gpurun_out
cd /tmp && export TMPDIR=/tmp && cd "$GRAFT_REPO_ROOT"
rm -f gpurun_out/parity_report.jsonl
timeout 1200 python -m pytest tests/test_hip_slam.py tests/test_hip_loops.py -q -m gpu -x 2>&1 | tail -4
for v in 1 0 1 0; do
  PSL_ADAM_SLICES=$v timeout 300 python bench.py --no-cpu-baseline 2>/dev/null | tail -1 > gpurun_out/r03_bench_v$v.json
  echo "adam_slices=$v"; python tools/show_bench.py gpurun_out/r03_bench_v$v.json | grep -E "FPS|adam"
done
grep "map_140\|scheduling_switch" gpurun_out/parity_report.jsonl | cut -c1-420
