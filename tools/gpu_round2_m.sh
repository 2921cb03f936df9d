#!/bin/bash
# round 2, run M: where does the lazy Adam spend its time (kernel trace of a short bench run)
cd /tmp && export TMPDIR=/tmp && cd "$GRAFT_REPO_ROOT"
mkdir -p gpurun_out
timeout 600 rocprofv3 --kernel-trace --stats -d gpurun_out/prof_m -o m -- python bench.py --no-cpu-baseline --steps 10 --warmup 5 > gpurun_out/m_bench.json 2> gpurun_out/m_rocprof.err
python tools/rocpd_stats.py gpurun_out/prof_m/m_results.db --csv gpurun_out/m_kernel_trace_stats.csv | head -12
python - <<'PY'
import sqlite3, glob
db = sqlite3.connect(glob.glob('gpurun_out/prof_m/*.db')[0])
tabs = [r[0] for r in db.execute("select name from sqlite_master where type='table'")]
kd = [t for t in tabs if 'kernel_dispatch' in t][0]
ks = [t for t in tabs if 'kernel_symbol' in t][0]
rows = db.execute(f"select s.kernel_name, d.end - d.start from {kd} d join {ks} s on d.kernel_id = s.id order by d.start").fetchall()
ad = [(i, dur) for i, (n, dur) in enumerate(rows) if 'k_map_adam' in n]
print('adam launches', len(ad))
durs = [d for _, d in ad]
import statistics
print('median us', statistics.median(durs) / 1e3, 'mean', statistics.mean(durs) / 1e3, 'max', max(durs) / 1e3)
big = sorted(durs)[-20:]
print('largest 20 (us):', [round(x / 1e3) for x in big])
# sequence sample
print('first 140 (us):', [round(d / 1e3) for d in durs[:140]])
PY
rm -rf gpurun_out/prof_m
