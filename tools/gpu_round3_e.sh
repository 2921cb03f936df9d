#!/bin/bash
# round 3, call E: F_theta weight fragments in LDS (fused and split kernels) -- parity, bench A/B on one box, traces
mkdir -p gpurun_out
cd /tmp && export TMPDIR=/tmp && cd "$GRAFT_REPO_ROOT"
rm -f gpurun_out/parity_report.jsonl gpurun_out/r03_blocks_e.jsonl
timeout 900 python -m pytest tests/test_hip_parity.py tests/test_hip_fullsize.py tests/test_hip_slam.py "tests/test_hip_loops.py::test_map_iters_native_matches_reference_loop" "tests/test_hip_loops.py::test_track_iters_native_matches_reference_loop" tests/test_hip_loops.py::test_colour_refinement_native_matches_reference tests/test_hip_loops.py::test_refine_runs_five_passes_over_all_rows -q -m gpu --durations=5 2>&1 | tail -30 > gpurun_out/pytest_r3e.log; tail -12 gpurun_out/pytest_r3e.log
for v in 0 1 0 1; do
  PSL_DECODE_SPLIT=$v timeout 300 python bench.py --no-cpu-baseline 2>gpurun_out/r03_bench_e$v.err | tail -1 > gpurun_out/r03_bench_e$v.json
  echo "split=$v"; python tools/show_bench.py gpurun_out/r03_bench_e$v.json | grep -E "FPS|decode_fwd |decode_bwd |decode_fwd_track|decode_bwd_track|knn  "
done
PSL_DECODE_SPLIT=0 PSL_DEBUG_BLOCKS=gpurun_out/r03_blocks_e.jsonl timeout 300 python bench.py --steps 6 --warmup 5 --no-cpu-baseline --no-kernel-timing > /dev/null 2> gpurun_out/r03_blocks_e.err
python tools/block_trace.py gpurun_out/r03_blocks_e.jsonl > gpurun_out/r03_block_trace_e.txt
grep -A5 "flags=0x1000d\|P=1000" gpurun_out/r03_block_trace_e.txt | head -40
timeout 300 python tools/knn_trace.py 2>&1 | grep "knn trace\|r_query" | tail -8
