#!/bin/bash
# round 3, final artefacts: PMC traffic first (bench.py reads profiles/r03_pmc_traffic_base.json), full GPU suite, the bench
# line, the rocprofv3 kernel trace of the same command, roofline sweep, k-NN roofline, other mixes, cfg5, two ranks
mkdir -p gpurun_out
cd /tmp && export TMPDIR=/tmp && cd "$GRAFT_REPO_ROOT"
COMMIT=${1:-unknown}
bash tools/pmc_run.sh r03 > gpurun_out/pmc_r03.log 2>&1
python tools/pmc_traffic.py gpurun_out/pmc_r03/fetch.csv gpurun_out/pmc_r03/write.csv profiles/r03_pmc_traffic_base.json $COMMIT > /dev/null && cp profiles/r03_pmc_traffic_base.json gpurun_out/
rm -f gpurun_out/parity_report.jsonl
timeout 1500 python -m pytest tests -q -m gpu --durations=5 2>&1 | tail -40 > gpurun_out/r03_pytest_gpu.log; tail -3 gpurun_out/r03_pytest_gpu.log
cp gpurun_out/parity_report.jsonl gpurun_out/r03_parity_report.jsonl
timeout 900 python bench.py 2>gpurun_out/r03_bench.err | tail -1 > gpurun_out/r03_bench.json
python tools/show_bench.py gpurun_out/r03_bench.json
timeout 900 rocprofv3 --kernel-trace --stats -d gpurun_out/prof_r03 -o r03 -- python bench.py > gpurun_out/r03_bench_under_rocprof.json 2> gpurun_out/r03_rocprof.err
python tools/rocpd_stats.py gpurun_out/prof_r03/r03_results.db --csv gpurun_out/r03_kernel_trace_stats.csv | head -20
python tools/rocpd_timeline.py gpurun_out/prof_r03/r03_results.db 0.5 > gpurun_out/r03_timeline.txt
rm -rf gpurun_out/prof_r03
timeout 600 python tools/roofline_sweep.py > gpurun_out/r03_sweep.log 2>&1; cp gpurun_out/roofline_sweep.json gpurun_out/r03_roofline_sweep.json
timeout 300 python tools/knn_roofline.py > gpurun_out/r03_knn_roofline.log 2>&1; cp gpurun_out/knn_roofline.json gpurun_out/r03_knn_roofline.json
for mix in replica tum scannet; do
  timeout 300 python bench.py --no-cpu-baseline --mix $mix 2>gpurun_out/r03_bench_$mix.err | tail -1 > gpurun_out/r03_bench_$mix.json
  echo "mix=$mix"; python tools/show_bench.py gpurun_out/r03_bench_$mix.json | grep -E "FPS"
  # round 2 quoted these mixes on 6 steps after 2 warm-up frames: the same command for a like-for-like comparison
  timeout 300 python bench.py --no-cpu-baseline --mix $mix --steps 6 --warmup 2 2>/dev/null | tail -1 > gpurun_out/r03_bench_${mix}_6steps.json
  echo "mix=$mix, 6 steps"; python tools/show_bench.py gpurun_out/r03_bench_${mix}_6steps.json | grep -E "FPS"
done
timeout 400 python bench.py --no-cpu-baseline --points 2000000 --width 1280 --height 960 2>gpurun_out/r03_bench_cfg5.err | tail -1 > gpurun_out/r03_bench_cfg5.json
echo cfg5; python tools/show_bench.py gpurun_out/r03_bench_cfg5.json | grep -E "FPS"
PSL_BENCH_SHARE_GPU=1 timeout 400 python bench.py --no-cpu-baseline --gpus 2 --steps 10 2>gpurun_out/r03_bench_x2.err | tail -1 > gpurun_out/r03_bench_frame_parallel_x2_shared_gpu.json
echo x2; python tools/show_bench.py gpurun_out/r03_bench_frame_parallel_x2_shared_gpu.json | grep -E "FPS"
timeout 300 python tools/exchange_timing.py --out gpurun_out/r03_exchange_timing_1m_8blocks.json > gpurun_out/r03_exchange.log 2>&1
