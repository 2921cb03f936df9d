#!/bin/bash
# round 3, call X: k-NN kernels emit (neighbour - sample, index) records; decode set-up phases read them (no position gather)
mkdir -p gpurun_out
cd /tmp && export TMPDIR=/tmp && cd "$GRAFT_REPO_ROOT"
rm -f gpurun_out/parity_report.jsonl
timeout 1200 python -m pytest tests/test_hip_parity.py tests/test_hip_fullsize.py tests/test_hip_slam.py tests/test_hip_loops.py -q -m gpu -x -k "not 140" 2>&1 | tail -4
for v in 0 1; do
  timeout 300 python bench.py --no-cpu-baseline 2>/dev/null | tail -1 > gpurun_out/r03_bench_x$v.json
  python tools/show_bench.py gpurun_out/r03_bench_x$v.json | grep -E "FPS|decode|geo_iter|knn  "
done
PSL_DEBUG_PHASES=1 timeout 300 python tools/phase_probe.py 2>&1 | grep "psl geo_iter\|psl bwd2 colour P=4995\|psl fwd2 colour P=4995" | sort | uniq -c | sort -rn | head -6 | cut -c1-330
