#!/bin/bash
# round 3, call Y (timing experiment, results of the decode are wrong with PSL_EXP != 0): what does F_theta's first layer wait for?
# PSL_EXP bit 0: no store of the hidden activations n_h1; bit 1: relu instead of softplus; bit 2: no store of the input rows n_x
mkdir -p gpurun_out
cd /tmp && export TMPDIR=/tmp && cd "$GRAFT_REPO_ROOT"
for v in 0 1 2 4 7; do
echo "PSL_EXP=$v"
PSL_EXP=$v PSL_DEBUG_PHASES=1 timeout 300 python tools/phase_probe.py 2>&1 | grep "psl fwd2 colour P=4995" | sort | uniq -c | sort -rn | head -3 | cut -c1-200
done
