#!/bin/bash
# rocprofv3 --kernel-trace of an arbitrary command, per-kernel summary: bash tools/trace_cmd.sh <tag> <command...>
cd /tmp && export TMPDIR=/tmp && cd "$GRAFT_REPO_ROOT"
TAG=$1; shift
O=gpurun_out; mkdir -p $O
timeout 600 rocprofv3 --kernel-trace --stats -d $O/prof_$TAG -o $TAG -- "$@" > $O/${TAG}_cmd.log 2>&1
python tools/rocpd_stats.py $O/prof_$TAG/${TAG}_results.db --csv $O/${TAG}_kernel_trace_stats.csv | grep -E "^kernel|nbr|trunk|k_dw|decode" | cut -c1-160 | head -20
rm -rf $O/prof_$TAG
