#!/bin/bash
# PMC counters for the bench kernels, one pass per counter group (gpurun refuses --pmc together with traces).
# usage (on the GPU box, from the repo root): bash tools/pmc_run.sh <tag>
set -e
TAG=${1:-r1}
cd /tmp && export TMPDIR=/tmp && cd "$GRAFT_REPO_ROOT"
CMD="python bench.py --steps 5 --warmup 5 --no-cpu-baseline --no-kernel-timing"
rocprofv3 --pmc SQ_WAVES SQ_BUSY_CYCLES SQ_WAVE_CYCLES SQ_INSTS_VALU SQ_INSTS_MFMA SQ_INSTS_LDS SQ_INSTS_VMEM_RD SQ_INSTS_VMEM_WR -d gpurun_out/pmc_$TAG -o sq1 -- $CMD > /dev/null 2> gpurun_out/pmc_$TAG.err1 || true
rocprofv3 --pmc SQ_WAIT_ANY SQ_WAIT_INST_ANY SQ_ACTIVE_INST_ANY SQ_ACTIVE_INST_VALU SQ_VALU_MFMA_BUSY_CYCLES SQ_INSTS_VALU_TRANS_F32 SQ_LDS_BANK_CONFLICT SQ_ACTIVE_INST_LDS -d gpurun_out/pmc_$TAG -o sq2 -- $CMD > /dev/null 2> gpurun_out/pmc_$TAG.err2 || true
rocprofv3 --pmc FETCH_SIZE -d gpurun_out/pmc_$TAG -o fetch -- $CMD > /dev/null 2> gpurun_out/pmc_$TAG.err3 || true
rocprofv3 --pmc WRITE_SIZE -d gpurun_out/pmc_$TAG -o write -- $CMD > /dev/null 2> gpurun_out/pmc_$TAG.err4 || true
ls -la gpurun_out/pmc_$TAG
