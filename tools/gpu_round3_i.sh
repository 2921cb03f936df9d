#!/bin/bash
# round 3, call I: checkpoint -- whole GPU suite (incl. cfg5 at 2 M points), bench line with the CPU leg, dW item-count sweep,
# per-workgroup trace of the one-launch geometry iteration
mkdir -p gpurun_out
cd /tmp && export TMPDIR=/tmp && cd "$GRAFT_REPO_ROOT"
rm -f gpurun_out/parity_report.jsonl gpurun_out/r03_blocks_i.jsonl
timeout 1500 python -m pytest tests -q -m gpu --durations=6 2>&1 | tail -30 > gpurun_out/pytest_r3i.log; tail -12 gpurun_out/pytest_r3i.log
cp gpurun_out/parity_report.jsonl gpurun_out/r03_parity_report_i.jsonl 2>/dev/null
timeout 600 python bench.py 2>gpurun_out/r03_bench_i.err | tail -1 > gpurun_out/r03_bench_i.json
python tools/show_bench.py gpurun_out/r03_bench_i.json | head -20
for n in 1000 1500 2000 3000; do
  PSL_DW_ITEMS=$n timeout 300 python bench.py --no-cpu-baseline --steps 10 2>/dev/null | tail -1 > gpurun_out/r03_bench_dw$n.json
  echo "dw_items=$n"; python tools/show_bench.py gpurun_out/r03_bench_dw$n.json | grep -E "FPS|dw_gemm|adam "
done
PSL_DEBUG_BLOCKS=gpurun_out/r03_blocks_i.jsonl timeout 300 python bench.py --steps 6 --warmup 5 --no-cpu-baseline --no-kernel-timing > /dev/null 2> gpurun_out/r03_blocks_i.err
python tools/block_trace.py gpurun_out/r03_blocks_i.jsonl > gpurun_out/r03_block_trace_i.txt
grep -A4 "geo_iter" gpurun_out/r03_block_trace_i.txt | head -24
