#!/bin/bash
# round 3, call N: psl_dedupe_blocks; exchange wall time again; device timeline of the steady state
mkdir -p gpurun_out
cd /tmp && export TMPDIR=/tmp && cd "$GRAFT_REPO_ROOT"
timeout 600 python -m pytest tests/test_hip_dist.py -q -m gpu -x 2>&1 | tail -5
timeout 300 python tools/exchange_timing.py 2>gpurun_out/r03_exchange.err | tail -1 > gpurun_out/r03_exchange_timing.json; python -c "
import json; d=json.load(open('gpurun_out/r03_exchange_timing.json')); print({k:v for k,v in d.items() if k!='per_call'}); [print(p) for p in d['per_call']]"
timeout 600 rocprofv3 --kernel-trace --stats -d gpurun_out/prof_n -o n -- python bench.py --no-cpu-baseline --no-kernel-timing --steps 20 > gpurun_out/r03_bench_under_rocprof_n.json 2> gpurun_out/rocprof_n.err
python tools/rocpd_timeline.py gpurun_out/prof_n/n_results.db 0.5 | tee gpurun_out/r03_n_timeline.txt
rm -rf gpurun_out/prof_n
