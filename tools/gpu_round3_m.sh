#!/bin/bash
# round 3, call M: device busy/idle timeline of the bench under rocprofv3, exchange wall time at 1 M points / 8 blocks,
# the other mixes, cfg5 and the two-rank share-GPU line
mkdir -p gpurun_out
cd /tmp && export TMPDIR=/tmp && cd "$GRAFT_REPO_ROOT"
timeout 300 python tools/exchange_timing.py 2>gpurun_out/r03_exchange.err | tail -1 > gpurun_out/r03_exchange_timing.json; cat gpurun_out/r03_exchange_timing.json; tail -3 gpurun_out/r03_exchange.err
timeout 600 rocprofv3 --kernel-trace --stats -d gpurun_out/prof_m -o m -- python bench.py --no-cpu-baseline --no-kernel-timing --steps 20 > gpurun_out/r03_bench_under_rocprof_m.json 2> gpurun_out/rocprof_m.err
python tools/rocpd_stats.py gpurun_out/prof_m/m_results.db --csv gpurun_out/r03_m_kernel_trace_stats.csv | head -16
python tools/rocpd_timeline.py gpurun_out/prof_m/m_results.db 0.3 | tee gpurun_out/r03_m_timeline.txt
rm -rf gpurun_out/prof_m
for mix in replica tum scannet; do
  timeout 300 python bench.py --no-cpu-baseline --mix $mix 2>gpurun_out/r03_bench_$mix.err | tail -1 > gpurun_out/r03_bench_$mix.json
  echo "mix=$mix"; python tools/show_bench.py gpurun_out/r03_bench_$mix.json | grep -E "FPS"
done
timeout 400 python bench.py --no-cpu-baseline --points 2000000 --width 1280 --height 960 2>gpurun_out/r03_bench_cfg5.err | tail -1 > gpurun_out/r03_bench_cfg5.json
echo cfg5; python tools/show_bench.py gpurun_out/r03_bench_cfg5.json | grep -E "FPS"
PSL_BENCH_SHARE_GPU=1 timeout 400 python bench.py --no-cpu-baseline --gpus 2 --steps 10 2>gpurun_out/r03_bench_x2.err | tail -1 > gpurun_out/r03_bench_frame_parallel_x2_shared_gpu.json
echo x2; python tools/show_bench.py gpurun_out/r03_bench_frame_parallel_x2_shared_gpu.json | grep -E "FPS"
