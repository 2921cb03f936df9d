#!/bin/bash
# round 2, call E: forward/backward v2 with spread geometry role; parity + phases + kernel trace
mkdir -p gpurun_out
cd /tmp && export TMPDIR=/tmp && cd "$GRAFT_REPO_ROOT"
timeout 900 python -m pytest tests -q -m gpu 2>&1 | tail -40 > gpurun_out/pytest_e.log
grep -E "passed|failed|FAILED|Error" gpurun_out/pytest_e.log | head -30
timeout 300 python tools/phase_probe.py 2>&1 | grep "psl fwd2" | sort | uniq -c | sort -rn | head -9 > gpurun_out/phases_e.log
cat gpurun_out/phases_e.log
timeout 300 python bench.py --steps 10 --warmup 5 --no-cpu-baseline 2>gpurun_out/bench_e.err | tail -1 > gpurun_out/bench_e.json
python tools/show_bench.py gpurun_out/bench_e.json | head -14
timeout 600 rocprofv3 --kernel-trace --stats -d gpurun_out/prof_e -o e -- python bench.py --steps 10 --warmup 5 --no-cpu-baseline --no-kernel-timing > gpurun_out/bench_e_rocprof.json 2> gpurun_out/rocprof_e.err
python tools/rocpd_stats.py gpurun_out/prof_e/e_results.db --csv gpurun_out/r02_e_kernel_trace_stats.csv | head -16
rm -rf gpurun_out/prof_e
