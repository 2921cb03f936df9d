#!/bin/bash
# round 3, call G: flat k-NN v2 (all-pass ranges in one trip, sphere-trimmed rows): exactness, bench, trace; launch gaps
mkdir -p gpurun_out
cd /tmp && export TMPDIR=/tmp && cd "$GRAFT_REPO_ROOT"
rm -f gpurun_out/parity_report.jsonl
timeout 900 python -m pytest tests/test_hip_parity.py tests/test_hip_fullsize.py "tests/test_hip_loops.py::test_track_iters_native_matches_reference_loop" tests/test_hip_slam.py -q -m gpu --durations=3 -x 2>&1 | tail -30 > gpurun_out/pytest_r3g.log; tail -8 gpurun_out/pytest_r3g.log
for v in 4 4; do
  PSL_KNN_SMALL=$v timeout 300 python bench.py --no-cpu-baseline 2>gpurun_out/r03_bench_g$v.err | tail -1 > gpurun_out/r03_bench_g$v.json
  echo "knn_small=$v"; python tools/show_bench.py gpurun_out/r03_bench_g$v.json | grep -E "FPS|knn  |adam |composite_fwd"
done
timeout 300 python tools/knn_trace.py 2>&1 | grep "knn trace\|r_query" | tail -4
timeout 600 rocprofv3 --kernel-trace --stats -d gpurun_out/prof_g -o g -- python bench.py --no-cpu-baseline --no-kernel-timing --steps 10 > gpurun_out/r03_bench_under_rocprof_g.json 2> gpurun_out/rocprof_g.err
python tools/rocpd_stats.py gpurun_out/prof_g/g_results.db --csv gpurun_out/r03_g_kernel_trace_stats.csv | head -24
python tools/rocpd_gaps.py gpurun_out/prof_g/g_results.db | tee gpurun_out/r03_g_launch_gaps.txt | head -30
rm -rf gpurun_out/prof_g
