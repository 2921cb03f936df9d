#!/bin/bash
# round 2, call G: k-NN diagnostics (per-ray vs per-sample kernel), growth check, scannet composite fix
mkdir -p gpurun_out
cd /tmp && export TMPDIR=/tmp && cd "$GRAFT_REPO_ROOT"
timeout 600 python -m pytest tests/test_hip_fullsize.py -q -m gpu -k "ray_knn" 2>&1 | tail -5
grep "fullsize_ray_knn" gpurun_out/parity_report.jsonl | tail -2 > gpurun_out/knn_diag.jsonl
python - <<'PY'
import json
for line in open('gpurun_out/knn_diag.jsonl'):
    d = json.loads(line)
    print(d['rays'], 'mismatched v2', d['mismatched_samples'], 'v1', d['mismatched_v1'])
    for x in d['diag']:
        print(' s', x['s'], 'r', round(x['r'],4), '\n   want', x['want'], '\n   got ', x['got'], '\n   v1  ', x['v1'], '\n   D', [round(v,6) for v in x['D']])
PY
PSL_KNN=1 timeout 300 python bench.py --steps 10 --warmup 5 --no-cpu-baseline 2>gpurun_out/bench_g.err | tail -1 > gpurun_out/bench_g.json
python tools/show_bench.py gpurun_out/bench_g.json | head -3
python -c "import json;d=json.load(open('gpurun_out/bench_g.json'));print(d['config']['points_start'],d['config']['points_end'],d['config']['points_added_per_mapped_frame'],d['split'])"
PSL_KNN=1 timeout 600 python bench.py --mix scannet --steps 6 --warmup 2 --no-cpu-baseline 2>gpurun_out/bench_g_scannet.err | tail -1 > gpurun_out/bench_g_scannet.json
python tools/show_bench.py gpurun_out/bench_g_scannet.json | head -5
