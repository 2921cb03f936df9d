#!/bin/bash
# round 2, call A: full gpu test suite (all failures shown) + a short bench line
mkdir -p gpurun_out
python -m pytest tests -q -m gpu --deselect tests/test_hip_dist.py 2>&1 | tail -30 > gpurun_out/pytest_a.log
python -m pytest tests/test_hip_dist.py -q -m gpu 2>&1 | tail -15 > gpurun_out/pytest_dist.log
python bench.py --steps 10 --warmup 5 --no-cpu-baseline 2>gpurun_out/bench_a.err | tail -1 > gpurun_out/bench_a.json
tail -5 gpurun_out/pytest_a.log; tail -3 gpurun_out/pytest_dist.log; python tools/show_bench.py gpurun_out/bench_a.json | head -12
