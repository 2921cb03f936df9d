#!/bin/bash
# round 3, call T: flat per-sample k-NN up to 8 192 rays (all tracker configs) vs also for the mapper's main-stream prefetch
mkdir -p gpurun_out
cd /tmp && export TMPDIR=/tmp && cd "$GRAFT_REPO_ROOT"
timeout 600 python -m pytest tests/test_hip_parity.py tests/test_hip_fullsize.py -q -m gpu -x -k "knn" 2>&1 | tail -2
for mix in base replica tum scannet; do
for v in 8192 10000000; do
  PSL_KNN_SMALL_MAX=$v timeout 300 python bench.py --no-cpu-baseline --mix $mix --steps 10 2>/dev/null | tail -1 > gpurun_out/r03_bench_t_${mix}_$v.json
  echo "mix=$mix small_max=$v"; python tools/show_bench.py gpurun_out/r03_bench_t_${mix}_$v.json | grep -E "FPS|knn   |knn_prefetch"
done; done
