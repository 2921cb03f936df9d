#!/bin/bash
# PMC counters of the hot-path kernels on tools/pmc_probe.py, one rocprofv3 pass per counter group
# (gpurun refuses --pmc combined with traces; FETCH_SIZE and WRITE_SIZE do not fit one pass).
# usage (GPU box, repo root): bash tools/pmc_run.sh <tag>
TAG=${1:-r01}
cd /tmp && export TMPDIR=/tmp && cd "$GRAFT_REPO_ROOT"
OUT=gpurun_out/pmc_$TAG
mkdir -p $OUT
run() { name=$1; shift; timeout 150 rocprofv3 --pmc "$@" -d $OUT -o $name -- python tools/pmc_probe.py > $OUT/$name.log 2>&1; python tools/rocpd_pmc.py $OUT/${name}_results.db > $OUT/$name.csv 2>&1; rm -f $OUT/${name}_results.db; }
run fetch FETCH_SIZE
run write WRITE_SIZE
run sq1 SQ_WAVES SQ_BUSY_CYCLES SQ_WAVE_CYCLES SQ_INSTS_VALU SQ_INSTS_MFMA SQ_INSTS_LDS SQ_INSTS_VMEM_RD SQ_INSTS_VMEM_WR
run sq2 SQ_WAIT_ANY SQ_WAIT_INST_ANY SQ_ACTIVE_INST_ANY SQ_ACTIVE_INST_VALU SQ_VALU_MFMA_BUSY_CYCLES SQ_INSTS_VALU_TRANS_F32 SQ_LDS_BANK_CONFLICT SQ_ACTIVE_INST_LDS
ls -la $OUT; head -30 $OUT/fetch.csv
