#!/bin/bash
# round 2, run N: A/B on one box -- lazy Adam and tracker fusion on/off
cd "$GRAFT_REPO_ROOT" || exit 1
mkdir -p gpurun_out
show() { python - "$1" <<'PY'
import json, sys
d = json.loads(open(sys.argv[1]).read().strip().splitlines()[-1])
print(sys.argv[1], d['value'], d['ms_per_step'], d.get('split'))
PY
}
rm -f gpurun_out/n_err.log
for rep in 1 2; do
PSL_LAZY_ADAM=1 PSL_DW_FUSED=1 timeout 600 python bench.py --no-cpu-baseline > gpurun_out/n_l1d1_$rep.json 2>>gpurun_out/n_err.log; show gpurun_out/n_l1d1_$rep.json
PSL_LAZY_ADAM=0 PSL_DW_FUSED=1 timeout 600 python bench.py --no-cpu-baseline > gpurun_out/n_l0d1_$rep.json 2>>gpurun_out/n_err.log; show gpurun_out/n_l0d1_$rep.json
PSL_LAZY_ADAM=1 PSL_DW_FUSED=0 timeout 600 python bench.py --no-cpu-baseline > gpurun_out/n_l1d0_$rep.json 2>>gpurun_out/n_err.log; show gpurun_out/n_l1d0_$rep.json
PSL_LAZY_ADAM=0 PSL_DW_FUSED=0 timeout 600 python bench.py --no-cpu-baseline > gpurun_out/n_l0d0_$rep.json 2>>gpurun_out/n_err.log; show gpurun_out/n_l0d0_$rep.json
done
