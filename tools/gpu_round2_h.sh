#!/bin/bash
# round 2, call H: full gpu suite; k-NN (row-parallel per-sample kernel) A/B; bwd2 phases; growth
mkdir -p gpurun_out
cd /tmp && export TMPDIR=/tmp && cd "$GRAFT_REPO_ROOT"
timeout 900 python -m pytest tests -q -m gpu 2>&1 | tail -30 > gpurun_out/pytest_h.log
grep -E "passed|failed|FAILED|Error" gpurun_out/pytest_h.log | head -20
timeout 300 python tools/phase_probe.py 2>&1 | grep "psl bwd2" | sort | uniq -c | sort -rn | head -6 > gpurun_out/phases_h.log
cat gpurun_out/phases_h.log
for v in 1 2; do
  PSL_KNN=$v timeout 300 python bench.py --steps 10 --warmup 5 --no-cpu-baseline 2>gpurun_out/bench_h_knn$v.err | tail -1 > gpurun_out/bench_h_knn$v.json
  echo "PSL_KNN=$v"; python tools/show_bench.py gpurun_out/bench_h_knn$v.json | head -3
  python -c "import json;d=json.load(open('gpurun_out/bench_h_knn$v.json'));print(d['config']['points_start'],d['config']['points_end'],d['config']['points_added_per_mapped_frame'],d['split'])"
done
