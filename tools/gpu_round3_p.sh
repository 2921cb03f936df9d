#!/bin/bash
# round 3, call P: sampled event bracketing (1 launch in 5 per class) vs every launch; exchange timing on the RCCL backend
mkdir -p gpurun_out
cd /tmp && export TMPDIR=/tmp && cd "$GRAFT_REPO_ROOT"
timeout 300 python tools/exchange_timing.py --out gpurun_out/r03_exchange_timing.json > gpurun_out/r03_exchange.log 2>&1; tail -3 gpurun_out/r03_exchange.log | cut -c1-600
for v in 5 1; do
  timeout 300 python bench.py --no-cpu-baseline --event-stride $v 2>gpurun_out/r03_bench_p$v.err | tail -1 > gpurun_out/r03_bench_p$v.json
  echo "stride=$v"; python tools/show_bench.py gpurun_out/r03_bench_p$v.json | head -16
done
