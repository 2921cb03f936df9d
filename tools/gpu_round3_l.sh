#!/bin/bash
# round 3, call L: ray stage back in its own launch, n_x prefetch in the F' rel-pos part, backward-fragment prefetch in the
# one-launch geometry iteration -- full GPU suite, bench, phase stamps
mkdir -p gpurun_out
cd /tmp && export TMPDIR=/tmp && cd "$GRAFT_REPO_ROOT"
rm -f gpurun_out/parity_report.jsonl
timeout 1500 python -m pytest tests -q -m gpu --durations=3 -x 2>&1 | tail -30 > gpurun_out/pytest_r3l.log; tail -8 gpurun_out/pytest_r3l.log
for v in 0 1; do
  timeout 300 python bench.py --no-cpu-baseline 2>gpurun_out/r03_bench_l$v.err | tail -1 > gpurun_out/r03_bench_l$v.json
  python tools/show_bench.py gpurun_out/r03_bench_l$v.json | grep -E "FPS|decode_bwd |decode_fwd |geo_iter|dw_gemm|adam "
done
PSL_DEBUG_PHASES=1 timeout 300 python tools/phase_probe.py 2>&1 | grep "psl geo_iter\|psl bwd2 colour" | sort | uniq -c | sort -rn | head -9
