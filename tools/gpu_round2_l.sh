#!/bin/bash
# round 2, run L: lazy exact Adam -- parity suite, then the bench line and a kernel trace
cd "$GRAFT_REPO_ROOT" || exit 1
mkdir -p gpurun_out
export TMPDIR=/tmp
timeout 1500 python -m pytest tests -m gpu -x -q > gpurun_out/l_pytest.log 2>&1; echo "pytest rc=$?" >> gpurun_out/l_pytest.log
tail -5 gpurun_out/l_pytest.log
timeout 600 python bench.py --no-cpu-baseline > gpurun_out/l_bench.json 2> gpurun_out/l_bench.err; tail -c 1500 gpurun_out/l_bench.json
PSL_DECODE_BWD=1 timeout 600 python bench.py --no-cpu-baseline --steps 10 > gpurun_out/l_bench_dense_adam.json 2>/dev/null; tail -c 400 gpurun_out/l_bench_dense_adam.json
for mix in tum scannet; do timeout 600 python bench.py --no-cpu-baseline --mix $mix --steps 5 --warmup 2 > gpurun_out/l_bench_$mix.json 2>/dev/null; tail -c 300 gpurun_out/l_bench_$mix.json; done
