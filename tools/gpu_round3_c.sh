#!/bin/bash
# round 3, call C: the O(new + touched) exchange (dist tests incl. the library's own RCCL communicator, 2-rank bench) and the
# per-workgroup trace of the colour-stage mapper launches
mkdir -p gpurun_out
cd /tmp && export TMPDIR=/tmp && cd "$GRAFT_REPO_ROOT"
rm -f gpurun_out/parity_report.jsonl gpurun_out/r03_blocks.jsonl
timeout 900 python -m pytest tests/test_hip_dist.py tests/test_hip_bench_multi.py tests/test_hip_slam.py -q -m gpu --durations=5 2>&1 | tail -30 > gpurun_out/pytest_r3c.log; tail -22 gpurun_out/pytest_r3c.log
PSL_DEBUG_BLOCKS=gpurun_out/r03_blocks.jsonl timeout 300 python bench.py --steps 6 --warmup 5 --no-cpu-baseline --no-kernel-timing > gpurun_out/r03_blocks_bench.json 2> gpurun_out/r03_blocks.err
python tools/block_trace.py gpurun_out/r03_blocks.jsonl > gpurun_out/r03_block_trace_summary.txt
grep -A7 "flags=0x1000d" gpurun_out/r03_block_trace_summary.txt | head -70
