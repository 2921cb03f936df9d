#!/bin/bash
# round 3, call B: the long-loop tests (three Adam semantics) + the remaining GPU suite; per-workgroup trace of the decode
# launches at 1 000 / 5 000 samples (where does a 43 us forward spend its time: dispatch skew, tile latency, CU sharing?)
mkdir -p gpurun_out
cd /tmp && export TMPDIR=/tmp && cd "$GRAFT_REPO_ROOT"
rm -f gpurun_out/parity_report.jsonl gpurun_out/r03_blocks.jsonl
timeout 1500 python -m pytest tests -q -m gpu --durations=8 2>&1 | tail -30 > gpurun_out/pytest_r3b.log; tail -22 gpurun_out/pytest_r3b.log
cp gpurun_out/parity_report.jsonl gpurun_out/r03_parity_report_b.jsonl 2>/dev/null
PSL_DEBUG_BLOCKS=gpurun_out/r03_blocks.jsonl timeout 300 python bench.py --steps 6 --warmup 5 --no-cpu-baseline --no-kernel-timing > gpurun_out/r03_blocks_bench.json 2> gpurun_out/r03_blocks.err
python tools/block_trace.py gpurun_out/r03_blocks.jsonl | tee gpurun_out/r03_block_trace_summary.txt | head -80
