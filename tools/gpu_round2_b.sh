#!/bin/bash
# round 2, call B: new forward kernel (v2) vs the fixtures, then timing A/B against the round-1 kernel
mkdir -p gpurun_out
timeout 600 python -m pytest tests/test_hip_parity.py tests/test_hip_loops.py tests/test_hip_slam.py tests/test_hip_fullsize.py -q -m gpu -x 2>&1 | tail -40 > gpurun_out/pytest_b.log
tail -6 gpurun_out/pytest_b.log
for v in 2 1; do
  PSL_DECODE=$v timeout 300 python bench.py --steps 10 --warmup 5 --no-cpu-baseline 2>gpurun_out/bench_b$v.err | tail -1 > gpurun_out/bench_b$v.json
  echo "PSL_DECODE=$v"; python tools/show_bench.py gpurun_out/bench_b$v.json | head -6
done
