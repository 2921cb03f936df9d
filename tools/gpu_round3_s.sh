#!/bin/bash
# round 3, call S: flat per-sample k-NN for tracker launches of 1 500 (Replica yaml) and 5 000 rays (TUM / ScanNet)?
mkdir -p gpurun_out
cd /tmp && export TMPDIR=/tmp && cd "$GRAFT_REPO_ROOT"
for mix in replica scannet; do
for v in 1024 100000; do
  PSL_KNN_SMALL_MAX=$v timeout 300 python bench.py --no-cpu-baseline --mix $mix --steps 10 2>/dev/null | tail -1 > gpurun_out/r03_bench_s_${mix}_$v.json
  echo "mix=$mix small_max=$v"; python tools/show_bench.py gpurun_out/r03_bench_s_${mix}_$v.json | grep -E "FPS|knn   "
done; done
