#!/bin/bash
# kernel trace of the default bench + the bench line itself
cd /tmp && export TMPDIR=/tmp && cd "$GRAFT_REPO_ROOT"
python bench.py > gpurun_out/r01_bench.json 2> gpurun_out/r01_bench.err
tail -1 gpurun_out/r01_bench.err
rocprofv3 --kernel-trace --stats -d gpurun_out/prof_r01 -o r01 -- python bench.py --no-cpu-baseline > gpurun_out/r01_bench_under_rocprof.json 2> gpurun_out/r01_rocprof.err
python tools/rocpd_stats.py gpurun_out/prof_r01/r01_results.db --csv gpurun_out/r01_kernel_trace_stats.csv | head -24
rm -rf gpurun_out/prof_r01
python tools/show_bench.py gpurun_out/r01_bench.json
