#!/bin/bash
# round 3, call Z: the mapper's ray stage inside the colour-stage decode backward (second attempt: per-sample evaluation by the
# 16 d(logits) threads, loss sums spread over 32 slots per iteration)
mkdir -p gpurun_out
cd /tmp && export TMPDIR=/tmp && cd "$GRAFT_REPO_ROOT"
rm -f gpurun_out/parity_report.jsonl
timeout 1200 python -m pytest tests/test_hip_slam.py -q -m gpu -x -k "switch" 2>&1 | tail -2
for v in 1 0 1 0; do
  PSL_RAY_IN_BWD=$v timeout 300 python bench.py --no-cpu-baseline 2>gpurun_out/r03_bench_z$v.err | tail -1 > gpurun_out/r03_bench_z$v.json; tail -3 gpurun_out/r03_bench_z$v.err
  echo "ray_in_bwd=$v"; python tools/show_bench.py gpurun_out/r03_bench_z$v.json | grep -E "FPS|decode_bwd |composite_fwd|decode_fwd "
done
PSL_DEBUG_PHASES=1 timeout 300 python tools/phase_probe.py 2>&1 | grep "psl bwd2 colour P=4995" | sort | uniq -c | sort -rn | head -3 | cut -c1-330
grep "scheduling_switch" gpurun_out/parity_report.jsonl | grep "ray_stage" | cut -c1-330
