#!/bin/bash
# round 2, call I: full gpu suite; bwd2 with pipelined F_theta backward + lazy Adam; k-NN A/B on the growing map
mkdir -p gpurun_out
cd /tmp && export TMPDIR=/tmp && cd "$GRAFT_REPO_ROOT"
timeout 900 python -m pytest tests -q -m gpu 2>&1 | tail -30 > gpurun_out/pytest_i.log
grep -E "passed|failed|FAILED|Error" gpurun_out/pytest_i.log | head -20
timeout 300 python tools/phase_probe.py 2>&1 | grep "psl bwd2" | sort | uniq -c | sort -rn | head -6 > gpurun_out/phases_i.log
cat gpurun_out/phases_i.log
for v in 1 2; do
  PSL_KNN=$v timeout 300 python bench.py --steps 10 --warmup 5 --no-cpu-baseline 2>gpurun_out/bench_i_knn$v.err | tail -1 > gpurun_out/bench_i_knn$v.json
  echo "PSL_KNN=$v"; python tools/show_bench.py gpurun_out/bench_i_knn$v.json | head -14
  python -c "import json;d=json.load(open('gpurun_out/bench_i_knn$v.json'));print(d['config']['points_start'],d['config']['points_end'],d['config']['points_added_per_mapped_frame'],d['split'])"
done
