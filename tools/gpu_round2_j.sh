#!/bin/bash
# round 2, call J: tests; bench (auto k-NN, coarse early-out); PMC passes -> traffic json; roofline sweep
mkdir -p gpurun_out
cd /tmp && export TMPDIR=/tmp && cd "$GRAFT_REPO_ROOT"
timeout 900 python -m pytest tests -q -m gpu 2>&1 | tail -30 > gpurun_out/pytest_j.log
grep -E "passed|failed|FAILED|Error" gpurun_out/pytest_j.log | head -20
timeout 300 python bench.py --steps 10 --warmup 5 --no-cpu-baseline 2>gpurun_out/bench_j.err | tail -1 > gpurun_out/bench_j.json
python tools/show_bench.py gpurun_out/bench_j.json | head -14
python -c "import json;d=json.load(open('gpurun_out/bench_j.json'));print(d['config']['points_start'],d['config']['points_end'],d['config']['points_added_per_mapped_frame'],d['split'])"
bash tools/pmc_run.sh r02 > gpurun_out/pmc_run.log 2>&1
tail -3 gpurun_out/pmc_run.log
python tools/pmc_traffic.py gpurun_out/pmc_r02/fetch.csv gpurun_out/pmc_r02/write.csv gpurun_out/r02_pmc_traffic_base.json | head -40
timeout 600 python tools/roofline_sweep.py 2>gpurun_out/sweep.err | tail -8
