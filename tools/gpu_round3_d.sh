#!/bin/bash
# round 3, call D: split decode kernels (F_theta / trunk) -- parity suite, per-workgroup traces, bench A/B on one box
mkdir -p gpurun_out
cd /tmp && export TMPDIR=/tmp && cd "$GRAFT_REPO_ROOT"
rm -f gpurun_out/parity_report.jsonl gpurun_out/r03_blocks_split.jsonl
timeout 1500 python -m pytest tests -q -m gpu -x --durations=5 2>&1 | tail -30 > gpurun_out/pytest_r3d.log; tail -14 gpurun_out/pytest_r3d.log
for v in 1 0 1 0; do
  PSL_DECODE_SPLIT=$v timeout 300 python bench.py --no-cpu-baseline 2>gpurun_out/r03_bench_split$v.err | tail -1 > gpurun_out/r03_bench_split$v.json
  echo "split=$v"; python tools/show_bench.py gpurun_out/r03_bench_split$v.json | grep -E "FPS|decode_fwd |decode_bwd |dw_gemm|adam "
done
PSL_DEBUG_BLOCKS=gpurun_out/r03_blocks_split.jsonl timeout 300 python bench.py --steps 6 --warmup 5 --no-cpu-baseline --no-kernel-timing > /dev/null 2> gpurun_out/r03_blocks_split.err
python tools/block_trace.py gpurun_out/r03_blocks_split.jsonl > gpurun_out/r03_block_trace_split.txt
grep -A5 "ftheta\|trunk" gpurun_out/r03_block_trace_split.txt | head -60
