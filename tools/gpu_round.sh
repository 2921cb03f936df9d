#!/bin/bash
# One parametrised script for every `gpurun` call of a round (replaces the per-call tools/gpu_round{2,3}_*.sh records).
#   gpurun --timeout 1500 -- 'bash tools/gpu_round.sh <tag> <step> [<step> ...]'
# <tag> prefixes every artefact under gpurun_out/ (r04a, r04b ...); steps run in the order given; each step is bounded by
# its own `timeout`, so a hung kernel costs one step, not the box.  Steps (arguments after ':' are comma-separated):
#   tests[:<pytest -k expr>]      pytest -m gpu (-x), tail of the log -> <tag>_pytest_gpu.log
#   bench[:<name>[:<args>]]       python bench.py <args> (args with ',' for ' ') -> <tag>_bench_<name>.json
#   eb:<name>:<ENV=V,...>;<args>  bench.py --no-cpu-baseline <args> under the given environment -> <tag>_bench_<name>.json
#   ab:<name>:<args>              the same bench line with PSL_LIB=_old/libpointslam_hip_${AB_BASE:-r03}.so (the "r03" column) and with this
#                                 build, twice each, on this one box
#   sweep                         tools/roofline_sweep.py -> <tag>_roofline_sweep.json
#   phases                        tools/phase_probe.py (PSL_DEBUG_PHASES) -> <tag>_phases.log
#   blocks                        per-workgroup trace of the decode launches -> <tag>_block_trace.txt
#   trace[:<args>[:<ENV=V,..>]]   rocprofv3 --kernel-trace --stats of bench.py <args> -> <tag>_kernel_trace_stats.csv (+ timeline)
#   pmc:<mix>                     PMC passes (FETCH_SIZE / WRITE_SIZE / SQ groups) on tools/pmc_probe.py --mix <mix>
#                                 -> <tag>_pmc_<mix>/*.csv and profiles-ready <tag>_pmc_traffic_<mix>.json
#   pmccal                        FETCH_SIZE / WRITE_SIZE on kernels of known traffic (tools/pmc_probe.py --calibrate) -> <tag>_pmc_calibration.json
#   x8[:<steps>[:<points>]]       bench.py --gpus 8 with all ranks sharing the one GPU (gloo) -> <tag>_bench_frame_parallel_x8_shared_gpu.json
#   knn                           tools/knn_roofline.py -> <tag>_knn_roofline.json
#   exchange                      tools/exchange_timing.py -> <tag>_exchange_timing.json
#   py:<script>[:<args>]          python <script> <args> -> <tag>_<basename>.log
mkdir -p gpurun_out
cd /tmp && export TMPDIR=/tmp && cd "$GRAFT_REPO_ROOT"
TAG=$1; shift
COMMIT=$(cat .commit_id 2>/dev/null || echo unknown)
O=gpurun_out
sp() { echo "${1//,/ }"; }
for step in "$@"; do
  IFS=':' read -r kind a1 a2 <<< "$step"
  t0=$(date +%s)
  case $kind in
    tests)
      rm -f $O/parity_report.jsonl
      if [ -n "$a1" ]; then timeout 1500 python -m pytest tests -q -m gpu -k "$(sp "$a1")" --durations=8 2>&1 | tail -40 > $O/${TAG}_pytest_gpu.log
      else timeout 1500 python -m pytest tests -q -m gpu --durations=8 2>&1 | tail -40 > $O/${TAG}_pytest_gpu.log; fi
      tail -3 $O/${TAG}_pytest_gpu.log; grep -E "^(FAILED|ERROR)|Error|assert" $O/${TAG}_pytest_gpu.log | head -10
      [ -f $O/parity_report.jsonl ] && cp $O/parity_report.jsonl $O/${TAG}_parity_report.jsonl ;;
    bench)
      name=${a1:-base}
      timeout 900 python bench.py $(sp "$a2") 2> $O/${TAG}_bench_$name.err | tail -1 > $O/${TAG}_bench_$name.json
      echo "bench $name: $(python tools/show_bench.py $O/${TAG}_bench_$name.json 2>&1 | grep -E 'FPS|frames' | head -2 | tr '\n' ' ')" ;;
    eb)   # bench with environment: eb:<name>:<ENV=V,ENV=V>;<args>
      envs="${a2%%;*}"; bargs="${a2#*;}"
      env $(sp "$envs") timeout 900 python bench.py --no-cpu-baseline $(sp "$bargs") 2> $O/${TAG}_bench_$a1.err | tail -1 > $O/${TAG}_bench_$a1.json
      echo "bench $a1 [$envs]: $(python tools/show_bench.py $O/${TAG}_bench_$a1.json 2>&1 | grep -E 'FPS|adam ' | tr '\n' ' ')" ;;
    ab)
      for rep in 1 2; do
        PSL_LIB=$PWD/_old/libpointslam_hip_${AB_BASE:-r03}.so timeout 600 python bench.py --no-cpu-baseline $(sp "$a2") 2>/dev/null | tail -1 > $O/${TAG}_ab_${a1}_r03_$rep.json
        timeout 600 python bench.py --no-cpu-baseline $(sp "$a2") 2>/dev/null | tail -1 > $O/${TAG}_ab_${a1}_new_$rep.json
        python - <<EOF
import json
for w in ("r03", "new"):
    try:
        d = json.load(open("$O/${TAG}_ab_${a1}_%s_$rep.json" % w)); print("ab $a1 rep $rep", w, d["value"], d["unit"], d["ms_per_step"], "ms/step")
    except Exception as e: print("ab $a1", w, "failed", e)
EOF
      done ;;
    sweep)
      timeout 600 python tools/roofline_sweep.py > $O/${TAG}_sweep.log 2>&1; cp $O/roofline_sweep.json $O/${TAG}_roofline_sweep.json; grep samples $O/${TAG}_sweep.log ;;
    phases)
      timeout 300 python tools/phase_probe.py > $O/${TAG}_phases.log 2>&1; grep "psl " $O/${TAG}_phases.log | tail -8 ;;
    blocks)
      rm -f $O/${TAG}_blocks.raw
      PSL_DEBUG_BLOCKS=$PWD/$O/${TAG}_blocks.raw timeout 300 python bench.py --no-cpu-baseline --no-kernel-timing --steps 5 --warmup 5 > /dev/null 2>&1
      python tools/block_trace.py $O/${TAG}_blocks.raw > $O/${TAG}_block_trace.txt 2>&1; rm -f $O/${TAG}_blocks.raw
      grep -A4 "P=5000 flags=0x1000d" $O/${TAG}_block_trace.txt | tail -24 ;;
    trace)
      env PSL_BENCH_MARK=1 $(sp "$a2") timeout 900 rocprofv3 --kernel-trace --stats -d $O/prof_$TAG -o $TAG -- python bench.py $(sp "$a1") > $O/${TAG}_bench_under_rocprof.json 2> $O/${TAG}_rocprof.err
      python tools/rocpd_stats.py $O/prof_$TAG/${TAG}_results.db --csv $O/${TAG}_kernel_trace_stats.csv --by-grid adam_lazy | grep -E "by-grid|^kernel|^[^,]*,[0-9]+,[0-9.]+,[0-9.]+,[0-9.]+,[0-9.]+,[0-9.]+$" | head -26
      python tools/rocpd_timeline.py $O/prof_$TAG/${TAG}_results.db 0.5 > $O/${TAG}_timeline.txt 2>&1
      python tools/rocpd_window.py $O/prof_$TAG/${TAG}_results.db --json $O/${TAG}_window.json > $O/${TAG}_window.txt 2>&1; grep "== pass" $O/${TAG}_window.txt
      rm -rf $O/prof_$TAG ;;
    pmc)
      mix=${a1:-base}; P=$O/${TAG}_pmc_$mix; mkdir -p $P
      pargs="--mix $mix"; pname=$mix
      if [ "$mix" = "cfg5" ]; then pargs="--mix base --points 2000000 --width 1280 --height 960"; fi
      run() { n=$1; shift; timeout 300 rocprofv3 --pmc "$@" -d $P -o $n -- python tools/pmc_probe.py $pargs > $P/$n.log 2>&1; python tools/rocpd_pmc.py $P/${n}_results.db > $P/$n.csv 2>&1; rm -f $P/${n}_results.db; }
      run fetch FETCH_SIZE
      run write WRITE_SIZE
      if [ "$a2" = "sq" ]; then
        run sq1 SQ_WAVES SQ_BUSY_CYCLES SQ_WAVE_CYCLES SQ_INSTS_VALU SQ_INSTS_MFMA SQ_INSTS_LDS SQ_INSTS_VMEM_RD SQ_INSTS_VMEM_WR
        run sq2 SQ_WAIT_ANY SQ_WAIT_INST_ANY SQ_ACTIVE_INST_ANY SQ_ACTIVE_INST_VALU SQ_VALU_MFMA_BUSY_CYCLES SQ_INSTS_VALU_TRANS_F32 SQ_LDS_BANK_CONFLICT SQ_ACTIVE_INST_LDS
      fi
      cp $O/pmc_probe_meta.json $P/probe_meta.json
      python tools/pmc_traffic.py $P/fetch.csv $P/write.csv $O/${TAG}_pmc_traffic_$mix.json $COMMIT $P/probe_meta.json | python -c "import json,sys; d=json.load(sys.stdin); [print(k, v.get('bytes_per_launch')) for k,v in d.items() if k!='_meta']" ;;
    pmccal)
      P=$O/${TAG}_pmccal; mkdir -p $P
      for c in FETCH_SIZE WRITE_SIZE; do
        n=$(echo $c | tr A-Z a-z | cut -d_ -f1)
        timeout 300 rocprofv3 --pmc $c -d $P -o $n -- python tools/pmc_probe.py --calibrate > $P/$n.log 2>&1
        python tools/rocpd_pmc.py $P/${n}_results.db > $P/$n.csv 2>&1; rm -f $P/${n}_results.db
      done
      python tools/pmc_calibration.py $P/fetch.csv $P/write.csv $O/pmc_calibration_known.json $O/${TAG}_pmc_calibration.json $COMMIT | grep -E "over_known|\"[a-z_0-9A-Z /-]*\": \{" | head -30 ;;
    x8)
      # N = 8 ranks on the ONE GPU over gloo (RCCL refuses two ranks per device): launcher, rank-order merge, ragged / empty
      # blocks and the replica check at N = 8 -- functional evidence, not a scaling number
      PSL_BENCH_SHARE_GPU=1 timeout 900 python bench.py --no-cpu-baseline --no-kernel-timing --gpus 8 --steps ${a1:-6} --points ${a2:-300000} 2> $O/${TAG}_bench_x8.err | tail -1 > $O/${TAG}_bench_frame_parallel_x8_shared_gpu.json
      python -c "import json; d=json.load(open('$O/${TAG}_bench_frame_parallel_x8_shared_gpu.json')); c=d['config']; print('x8', d['value'], d['unit'], 'identical', c.get('replicas_identical_after_exchange'), c.get('exchange_cadence')); print([ (r['rank'], r['added'], r['exchange_ms']) for r in c['per_rank']])" ;;
    knn)
      timeout 300 python tools/knn_roofline.py > $O/${TAG}_knn_roofline.log 2>&1; cp $O/knn_roofline.json $O/${TAG}_knn_roofline.json; tail -3 $O/${TAG}_knn_roofline.log ;;
    exchange)
      timeout 300 python tools/exchange_timing.py --out $O/${TAG}_exchange_timing.json > $O/${TAG}_exchange.log 2>&1
      PYTORCH_HIP_ALLOC_CONF=expandable_segments:True timeout 300 python tools/exchange_timing.py --out $O/${TAG}_exchange_timing_expandable.json > $O/${TAG}_exchange2.log 2>&1
      python tools/exchange_timing.py --summary $O/${TAG}_exchange_timing.json $O/${TAG}_exchange_timing_expandable.json ;;
    py)
      b=$(basename "$a1" .py)
      timeout 900 python $a1 $(sp "$a2") > $O/${TAG}_$b.log 2>&1; tail -15 $O/${TAG}_$b.log ;;
    *) echo "unknown step $step" ;;
  esac
  echo "== $step: $(( $(date +%s) - t0 )) s"
done
