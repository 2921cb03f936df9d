#!/bin/bash
# round 2, call C: forward v2 (pipelined) parity + rocprofv3 kernel trace of the bench
mkdir -p gpurun_out
cd /tmp && export TMPDIR=/tmp && cd "$GRAFT_REPO_ROOT"
timeout 600 python -m pytest tests/test_hip_parity.py tests/test_hip_loops.py tests/test_hip_slam.py tests/test_hip_fullsize.py -q -m gpu -x 2>&1 | tail -30 > gpurun_out/pytest_c.log
tail -4 gpurun_out/pytest_c.log
timeout 300 python bench.py --steps 10 --warmup 5 --no-cpu-baseline 2>gpurun_out/bench_c.err | tail -1 > gpurun_out/bench_c.json
python tools/show_bench.py gpurun_out/bench_c.json | head -10
timeout 600 rocprofv3 --kernel-trace --stats -d gpurun_out/prof_c -o c -- python bench.py --steps 10 --warmup 5 --no-cpu-baseline --no-kernel-timing > gpurun_out/bench_c_rocprof.json 2> gpurun_out/rocprof_c.err
python tools/rocpd_stats.py gpurun_out/prof_c/c_results.db --csv gpurun_out/r02_c_kernel_trace_stats.csv | head -24
rm -rf gpurun_out/prof_c
