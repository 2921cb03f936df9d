#!/bin/bash
# round 2, run R: which k-NN kernel does the rare memory fault follow?
cd "$GRAFT_REPO_ROOT" || exit 1
N=${1:-30}
count() { # dir, label, env...
  dir=$1; label=$2; shift; shift
  bad=0
  for rep in $(seq 1 $N); do
    (cd $dir && env "$@" timeout 300 python bench.py --no-cpu-baseline --no-kernel-timing > /tmp/r.json 2> /tmp/r_err.log)
    rc=$?
    if [ $rc -ne 0 ]; then bad=$((bad+1)); grep -h "Memory access fault\|Error\|error" /tmp/r_err.log | head -2; fi
  done
  echo "$label: $bad / $N failed"
}
count . knn1_everywhere PSL_KNN=1
count . knn2_everywhere PSL_KNN=2
