#!/bin/bash
# round 3, call W: grid of the lazy Adam's work-list launch (workgroups per group; each walks the list with that stride)
mkdir -p gpurun_out
cd /tmp && export TMPDIR=/tmp && cd "$GRAFT_REPO_ROOT"
for v in 2048 1024 3072 4096 6144 2048; do
  PSL_ADAM_GRID=$v timeout 300 python bench.py --no-cpu-baseline --steps 15 2>/dev/null | tail -1 > gpurun_out/r03_bench_w$v.json
  echo "adam_grid=$v"; python tools/show_bench.py gpurun_out/r03_bench_w$v.json | grep -E "FPS|adam  "
done
