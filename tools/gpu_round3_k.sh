#!/bin/bash
# round 3, call K: coalesced gradient scatter (LDS transpose), fc_c prefetch in the backward trunk, ray stage inside the
# colour-stage backward -- parity, bench A/B, phase stamps
mkdir -p gpurun_out
cd /tmp && export TMPDIR=/tmp && cd "$GRAFT_REPO_ROOT"
rm -f gpurun_out/parity_report.jsonl
timeout 1500 python -m pytest tests/test_hip_parity.py tests/test_hip_fullsize.py tests/test_hip_loops.py tests/test_hip_slam.py -q -m gpu --durations=3 -x 2>&1 | tail -30 > gpurun_out/pytest_r3k.log; tail -8 gpurun_out/pytest_r3k.log
for v in 1 0 1 0; do
  PSL_RAY_IN_BWD=$v timeout 300 python bench.py --no-cpu-baseline 2>gpurun_out/r03_bench_k$v.err | tail -1 > gpurun_out/r03_bench_k$v.json
  echo "ray_in_bwd=$v"; python tools/show_bench.py gpurun_out/r03_bench_k$v.json | grep -E "FPS|decode_bwd |geo_iter|composite_fwd|adam "
done
PSL_DEBUG_PHASES=1 timeout 300 python tools/phase_probe.py 2>&1 | grep "psl geo_iter\|psl bwd2 colour" | sort | uniq -c | sort -rn | head -9
