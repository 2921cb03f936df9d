#!/bin/bash
# round 2, run M: kernel trace of a short bench run (optionally with extra environment: tools/gpu_round2_m.sh tag VAR=1 ...)
cd /tmp && export TMPDIR=/tmp && cd "$GRAFT_REPO_ROOT"
mkdir -p gpurun_out
tag=${1:-m}; shift
env "$@" timeout 600 rocprofv3 --kernel-trace --stats -d gpurun_out/prof_$tag -o $tag -- python bench.py --no-cpu-baseline --steps 10 --warmup 5 > gpurun_out/${tag}_bench.json 2> gpurun_out/${tag}_rocprof.err
python tools/rocpd_stats.py gpurun_out/prof_$tag/${tag}_results.db --csv gpurun_out/${tag}_kernel_trace_stats.csv | head -${LINES_SHOWN:-14}
rm -rf gpurun_out/prof_$tag
