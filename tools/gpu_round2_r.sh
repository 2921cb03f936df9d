#!/bin/bash
# round 2, run R: does the bench still fault?  (async, as the driver runs it; 1 in ~20 runs faulted before the fix)
cd "$GRAFT_REPO_ROOT" || exit 1
N=${1:-40}
bad=0
for rep in $(seq 1 $N); do
  timeout 300 python bench.py --no-cpu-baseline --no-kernel-timing > /tmp/r.json 2> /tmp/r_err.log
  rc=$?
  if [ $rc -ne 0 ]; then bad=$((bad+1)); grep -h "Memory access fault\|Error\|error" /tmp/r_err.log | head -2; fi
done
echo "bench runs failed: $bad / $N"
