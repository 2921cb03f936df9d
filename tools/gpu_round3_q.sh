#!/bin/bash
# round 3, call Q: class timing through hipExtLaunchKernelGGL's start/stop events (no barrier markers); split kernels deleted
mkdir -p gpurun_out
cd /tmp && export TMPDIR=/tmp && cd "$GRAFT_REPO_ROOT"
rm -f gpurun_out/parity_report.jsonl
timeout 1500 python -m pytest tests -q -m gpu --durations=3 -x 2>&1 | tail -30 > gpurun_out/pytest_r3q.log; tail -6 gpurun_out/pytest_r3q.log
for v in 5 1; do
  timeout 300 python bench.py --no-cpu-baseline --event-stride $v 2>gpurun_out/r03_bench_q$v.err | tail -1 > gpurun_out/r03_bench_q$v.json
  echo "stride=$v"; python tools/show_bench.py gpurun_out/r03_bench_q$v.json | head -16
  python -c "
import json; d=json.loads(open('gpurun_out/r03_bench_q$v.json').read()); print('ms/step', d['ms_per_step'], 'profiled', d['profiled_ms_per_step'])"
done
timeout 600 rocprofv3 --kernel-trace --stats -d gpurun_out/prof_q -o q -- python bench.py --no-cpu-baseline > gpurun_out/r03_bench_under_rocprof_q.json 2> gpurun_out/rocprof_q.err
python tools/rocpd_stats.py gpurun_out/prof_q/q_results.db --csv gpurun_out/r03_q_kernel_trace_stats.csv | head -16
rm -rf gpurun_out/prof_q
