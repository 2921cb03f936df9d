#!/bin/bash
# SQ counter groups of the decode kernels on tools/pmc_probe.py (separate --pmc passes): bash tools/pmc_sq.sh <tag> [mix]
cd /tmp && export TMPDIR=/tmp && cd "$GRAFT_REPO_ROOT"
TAG=$1; MIX=${2:-base}; P=gpurun_out/${TAG}_sq_$MIX; mkdir -p $P
run() { n=$1; shift; timeout 300 rocprofv3 --pmc "$@" -d $P -o $n -- python tools/pmc_probe.py --mix $MIX > $P/$n.log 2>&1; python tools/rocpd_pmc.py $P/${n}_results.db > $P/$n.csv 2>&1; rm -f $P/${n}_results.db; }
run sq1 SQ_WAVES SQ_BUSY_CYCLES SQ_WAVE_CYCLES SQ_INSTS_VALU SQ_INSTS_MFMA SQ_INSTS_LDS SQ_INSTS_VMEM_RD SQ_INSTS_VMEM_WR
run sq2 SQ_WAIT_ANY SQ_WAIT_INST_ANY SQ_ACTIVE_INST_ANY SQ_ACTIVE_INST_VALU SQ_VALU_MFMA_BUSY_CYCLES SQ_INSTS_VALU_TRANS_F32 SQ_LDS_BANK_CONFLICT SQ_ACTIVE_INST_LDS
run sq3 SQ_INST_CYCLES_VMEM SQ_ACTIVE_INST_VMEM SQ_WAIT_INST_LDS SQ_ACTIVE_INST_MISC SQ_INSTS_SALU SQ_INSTS_SMEM SQ_ACTIVE_INST_SCA SQ_BUSY_CU_CYCLES
python - <<PY
import csv, collections
d = collections.defaultdict(dict)
for n in ("sq1", "sq2", "sq3"):
    try:
        for r in csv.reader(open("$P/%s.csv" % n)):
            if len(r) == 5 and r[0] != "kernel": d[r[0]][r[1]] = float(r[3])
    except Exception as e: print(n, e)
for k, v in d.items():
    if not any(x in k for x in ("nbr", "trunk", "decode", "k_dw", "geo_iter", "adam")): continue
    print(k[-70:])
    print("   ", {c: round(x) for c, x in sorted(v.items())})
    if v.get("SQ_BUSY_CYCLES"):
        b = v["SQ_BUSY_CYCLES"]
        print("    mfma_busy/busy %.3f  valu_active/busy %.3f  lds_active/busy %.3f  valu/mfma insts %.2f" % (
            v.get("SQ_VALU_MFMA_BUSY_CYCLES", 0) / b, v.get("SQ_ACTIVE_INST_VALU", 0) / b, v.get("SQ_ACTIVE_INST_LDS", 0) / b,
            v.get("SQ_INSTS_VALU", 0) / max(v.get("SQ_INSTS_MFMA", 1), 1)))
PY
