#!/bin/bash
# round 3, call A: the new parity tests (1 M-point loss parity vs the oracle, 140-iteration loops, self-launching 2-rank
# bench) + the whole GPU suite + the default bench line with the measured parity float
mkdir -p gpurun_out
cd /tmp && export TMPDIR=/tmp && cd "$GRAFT_REPO_ROOT"
rm -f gpurun_out/parity_report.jsonl
timeout 1500 python -m pytest tests -q -m gpu -x --durations=15 2>&1 | tail -40 > gpurun_out/pytest_r3a.log; tail -25 gpurun_out/pytest_r3a.log
cp gpurun_out/parity_report.jsonl gpurun_out/r03_parity_report_a.jsonl 2>/dev/null
timeout 600 python bench.py 2>gpurun_out/r03_bench_a.err | tail -1 > gpurun_out/r03_bench_a.json
python tools/show_bench.py gpurun_out/r03_bench_a.json | head -18
python - <<'PY'
import json
d = json.load(open("gpurun_out/r03_bench_a.json"))
print("parity:", d["config"].get("render_loss_rel_err_vs_reference"), json.dumps(d["config"].get("render_loss_parity"))[:1500])
print("cpu_baseline:", d.get("cpu_baseline"))
PY
tail -5 gpurun_out/r03_bench_a.err
