#!/bin/bash
# round 3, call F: flat-enumeration k-NN for the tracker's launches, fixed-grid lazy Adam; exactness tests, bench A/B, trace
mkdir -p gpurun_out
cd /tmp && export TMPDIR=/tmp && cd "$GRAFT_REPO_ROOT"
rm -f gpurun_out/parity_report.jsonl
timeout 900 python -m pytest tests/test_hip_parity.py tests/test_hip_fullsize.py tests/test_hip_slam.py "tests/test_hip_loops.py::test_map_iters_native_matches_reference_loop" "tests/test_hip_loops.py::test_track_iters_native_matches_reference_loop" "tests/test_hip_loops.py::test_map_iters_140_iterations_vs_oracle[torch2-frozen-decoder]" -q -m gpu --durations=5 -x 2>&1 | tail -30 > gpurun_out/pytest_r3f.log; tail -12 gpurun_out/pytest_r3f.log
for v in 4 1 4 1; do
  PSL_KNN_SMALL=$v timeout 300 python bench.py --no-cpu-baseline 2>gpurun_out/r03_bench_f$v.err | tail -1 > gpurun_out/r03_bench_f$v.json
  echo "knn_small=$v"; python tools/show_bench.py gpurun_out/r03_bench_f$v.json | grep -E "FPS|knn  |adam |composite_fwd"
done
timeout 300 python tools/knn_trace.py 2>&1 | grep "knn trace\|r_query" | tail -4
