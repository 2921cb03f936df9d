#!/bin/bash
# round 3, call J: feature rows of the lazy Adam inside the dW launch; phase stamps of the one-launch geometry iteration
mkdir -p gpurun_out
cd /tmp && export TMPDIR=/tmp && cd "$GRAFT_REPO_ROOT"
rm -f gpurun_out/parity_report.jsonl
timeout 1200 python -m pytest tests/test_hip_loops.py tests/test_hip_slam.py -q -m gpu --durations=3 -x 2>&1 | tail -30 > gpurun_out/pytest_r3j.log; tail -8 gpurun_out/pytest_r3j.log
for v in 1 0 1 0; do
  PSL_ROWS_IN_DW=$v timeout 300 python bench.py --no-cpu-baseline 2>gpurun_out/r03_bench_j$v.err | tail -1 > gpurun_out/r03_bench_j$v.json
  echo "rows_in_dw=$v"; python tools/show_bench.py gpurun_out/r03_bench_j$v.json | grep -E "FPS|dw_gemm|adam "
done
PSL_DEBUG_PHASES=1 timeout 300 python tools/phase_probe.py 2>&1 | grep "psl geo_iter\|psl fwd2 colour\|psl bwd2 colour" | sort | uniq -c | sort -rn | head -12
