#!/bin/bash
# round 2, call D: phase stamps of forward v2; backward v2 parity; A/B timing
mkdir -p gpurun_out
cd /tmp && export TMPDIR=/tmp && cd "$GRAFT_REPO_ROOT"
PSL_DECODE_BWD=1 timeout 300 python tools/phase_probe.py 2>&1 | grep "psl fwd2" | sort | uniq -c | sort -rn | head -12 > gpurun_out/phases_d.log
cat gpurun_out/phases_d.log
timeout 900 python -m pytest tests/test_hip_parity.py tests/test_hip_loops.py tests/test_hip_slam.py tests/test_hip_fullsize.py -q -m gpu 2>&1 | tail -60 > gpurun_out/pytest_d.log
grep -E "passed|failed|FAILED|Error" gpurun_out/pytest_d.log | head -30
for v in 2 1; do
  PSL_DECODE_BWD=$v timeout 300 python bench.py --steps 10 --warmup 5 --no-cpu-baseline 2>gpurun_out/bench_d$v.err | tail -1 > gpurun_out/bench_d$v.json
  echo "PSL_DECODE_BWD=$v"; python tools/show_bench.py gpurun_out/bench_d$v.json | head -8
done
