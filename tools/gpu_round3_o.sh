#!/bin/bash
# round 3, call O: no pose copy per tracked frame (host runs ahead); exchange timing on the RCCL backend
mkdir -p gpurun_out
cd /tmp && export TMPDIR=/tmp && cd "$GRAFT_REPO_ROOT"
timeout 300 python tools/exchange_timing.py 2>gpurun_out/r03_exchange.err | tail -1 > gpurun_out/r03_exchange_timing.json; python -c "
import json; d=json.load(open('gpurun_out/r03_exchange_timing.json')); print({k:v for k,v in d.items() if k!='per_call'}); [print(p) for p in d['per_call']]"; tail -2 gpurun_out/r03_exchange.err
for v in 0 1; do
  timeout 300 python bench.py --no-cpu-baseline 2>gpurun_out/r03_bench_o$v.err | tail -1 > gpurun_out/r03_bench_o$v.json
  python tools/show_bench.py gpurun_out/r03_bench_o$v.json | grep -E "FPS"
done
timeout 600 python -m pytest tests/test_hip_slam.py tests/test_hip_loops.py -q -m gpu -x -k "not 140" 2>&1 | tail -3
timeout 600 rocprofv3 --kernel-trace --stats -d gpurun_out/prof_o -o o -- python bench.py --no-cpu-baseline --no-kernel-timing --steps 20 > gpurun_out/r03_bench_under_rocprof_o.json 2> gpurun_out/rocprof_o.err
python tools/rocpd_timeline.py gpurun_out/prof_o/o_results.db 0.5 | tee gpurun_out/r03_o_timeline.txt | head -8
rm -rf gpurun_out/prof_o
