#!/bin/bash
# timing experiment: how much of the decode forward is the saving of activations for the backward pass
for v in 0 1 3 7 15; do
  PSL_DEBUG_NOSAVE=$v python bench.py --steps 6 --warmup 3 --no-cpu-baseline 2>/dev/null | tail -1 > gpurun_out/bench_ns$v.json
  echo "NOSAVE=$v"; python tools/show_bench.py gpurun_out/bench_ns$v.json | grep -E "FPS|decode_fwd"
done
