#!/bin/bash
# round 2, call S: final artefacts -- parity suite, bench line with cpu baseline, kernel trace of the same command,
# k-NN roofline, the three other iteration mixes, cfg5
mkdir -p gpurun_out
cd /tmp && export TMPDIR=/tmp && cd "$GRAFT_REPO_ROOT"
timeout 1500 python -m pytest tests -q -m gpu 2>&1 | tail -5 > gpurun_out/pytest_s.log; cat gpurun_out/pytest_s.log
timeout 600 python bench.py 2>gpurun_out/r02_bench.err | tail -1 > gpurun_out/r02_bench.json
python tools/show_bench.py gpurun_out/r02_bench.json | head -18
timeout 600 rocprofv3 --kernel-trace --stats -d gpurun_out/prof_s -o s -- python bench.py --no-cpu-baseline > gpurun_out/r02_bench_under_rocprof.json 2> gpurun_out/rocprof_s.err
python tools/rocpd_stats.py gpurun_out/prof_s/s_results.db --csv gpurun_out/r02_kernel_trace_stats.csv | head -26
rm -rf gpurun_out/prof_s
timeout 300 python tools/knn_roofline.py 2>gpurun_out/knn_roofline.err | tail -3
for mix in replica tum scannet; do
  timeout 600 python bench.py --mix $mix --steps 6 --warmup 2 --no-cpu-baseline 2>gpurun_out/r02_bench_$mix.err | tail -1 > gpurun_out/r02_bench_$mix.json
  echo "mix $mix"; python tools/show_bench.py gpurun_out/r02_bench_$mix.json | head -1
done
timeout 600 python bench.py --points 2000000 --width 1280 --height 960 --no-cpu-baseline 2>gpurun_out/r02_bench_cfg5.err | tail -1 > gpurun_out/r02_bench_cfg5.json
echo cfg5; python tools/show_bench.py gpurun_out/r02_bench_cfg5.json | head -1
