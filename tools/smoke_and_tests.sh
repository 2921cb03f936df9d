#!/bin/bash
python -c "import __graft_entry__ as g; g.smoke()" 2>&1 | tail -2
python -m pytest tests -q -m gpu 2>&1 | tail -1
python bench.py --steps 10 --warmup 5 --no-cpu-baseline 2>/dev/null | tail -1 > gpurun_out/bench_last.json
python tools/show_bench.py gpurun_out/bench_last.json | head -3
