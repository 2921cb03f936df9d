#!/bin/bash
# round 2, call F: per-ray k-NN parity + A/B; growth-enabled bench; TUM / ScanNet / Replica mixes; cfg5 size
mkdir -p gpurun_out
cd /tmp && export TMPDIR=/tmp && cd "$GRAFT_REPO_ROOT"
timeout 900 python -m pytest tests -q -m gpu 2>&1 | tail -40 > gpurun_out/pytest_f.log
grep -E "passed|failed|FAILED|Error" gpurun_out/pytest_f.log | head -30
for v in 2 1; do
  PSL_KNN=$v timeout 300 python bench.py --steps 10 --warmup 5 --no-cpu-baseline 2>gpurun_out/bench_f_knn$v.err | tail -1 > gpurun_out/bench_f_knn$v.json
  echo "PSL_KNN=$v"; python tools/show_bench.py gpurun_out/bench_f_knn$v.json | head -3
  python -c "import json;d=json.load(open('gpurun_out/bench_f_knn$v.json'));print(d['config']['points_start'],d['config']['points_end'],d['config']['points_added_per_mapped_frame'],d['split'])"
done
for mix in replica tum scannet; do
  timeout 600 python bench.py --mix $mix --steps 6 --warmup 2 --no-cpu-baseline 2>gpurun_out/bench_f_$mix.err | tail -1 > gpurun_out/bench_f_$mix.json
  echo "mix $mix"; tail -2 gpurun_out/bench_f_$mix.err; python tools/show_bench.py gpurun_out/bench_f_$mix.json | head -14
done
timeout 600 python bench.py --points 2000000 --width 1280 --height 960 --steps 10 --warmup 5 --no-cpu-baseline 2>gpurun_out/bench_f_cfg5.err | tail -1 > gpurun_out/bench_f_cfg5.json
echo cfg5; tail -2 gpurun_out/bench_f_cfg5.err; python tools/show_bench.py gpurun_out/bench_f_cfg5.json | head -4
