#!/bin/bash
# round 3, call R: why did the replica / scannet mixes get slower than round 2?  kernel trace of the replica mix
mkdir -p gpurun_out
cd /tmp && export TMPDIR=/tmp && cd "$GRAFT_REPO_ROOT"
timeout 600 rocprofv3 --kernel-trace --stats -d gpurun_out/prof_rr -o rr -- python bench.py --no-cpu-baseline --no-kernel-timing --mix replica --steps 10 > gpurun_out/r03_bench_replica_under_rocprof.json 2> gpurun_out/rocprof_rr.err
python tools/rocpd_stats.py gpurun_out/prof_rr/rr_results.db --csv gpurun_out/r03_replica_kernel_trace_stats.csv | head -24
python tools/rocpd_timeline.py gpurun_out/prof_rr/rr_results.db 0.5 | head -12
rm -rf gpurun_out/prof_rr
for v in 0 1; do
PSL_GEO_FUSED=$v timeout 300 python bench.py --no-cpu-baseline --mix replica --steps 10 2>/dev/null | tail -1 > gpurun_out/r03_bench_replica_geo$v.json
echo "geo_fused=$v"; python tools/show_bench.py gpurun_out/r03_bench_replica_geo$v.json | grep -E "FPS|geo|knn  "
done
