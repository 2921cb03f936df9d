#!/bin/bash
# round 2, run N: A/B on one box: tools/gpu_round2_n.sh VAR  -> bench with VAR=1 / VAR=0, two repetitions each
cd "$GRAFT_REPO_ROOT" || exit 1
mkdir -p gpurun_out
V=${1:-PSL_KNN_OVERLAP}
show() { python - "$1" <<'PY'
import json, sys
d = json.loads(open(sys.argv[1]).read().strip().splitlines()[-1])
print(sys.argv[1], d['value'], d['ms_per_step'])
PY
}
for rep in 1 2; do
  for on in 1 0; do
    env $V=$on timeout 600 python bench.py --no-cpu-baseline --no-kernel-timing > gpurun_out/n_${on}_$rep.json 2>>gpurun_out/n_err.log; show gpurun_out/n_${on}_$rep.json
  done
done
