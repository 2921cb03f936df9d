"""Summarise a PSL_DEBUG_BLOCKS trace (JSON lines written by libpointslam_hip.so): for every traced launch, where its
workgroups ran (XCC / SE / CU from HW_ID + XCC_ID), when they started relative to the first one (100 MHz wall clock) and how
long they took -- i.e. how much of a launch's duration is dispatch skew, single-tile latency, or two tiles sharing a CU."""
import json
import statistics as st
import sys
from collections import Counter, defaultdict


def cu_of(hw):
    hwid, xcc = hw & 0xFFFFFFFF, (hw >> 32) & 0xF
    return (xcc, (hwid >> 13) & 7, (hwid >> 12) & 1, (hwid >> 8) & 15)


def q(v, p):
    v = sorted(v)
    return v[min(len(v) - 1, int(p * len(v)))]


def main(path, limit=None):
    for n, line in enumerate(open(path)):
        d = json.loads(line)
        B = d["blocks"]
        ct = d["color_tiles"]
        t0 = min(b[0] for b in B if b[0])
        span = (max(b[1] for b in B) - t0) / 100.0
        print(f"== {d['kernel']} P={d['P']} flags={d['flags']:#x} grid={d['grid']} colour tiles={ct} threads={d['threads']}: "
              f"first start -> last end {span:.1f} us")
        # colour tiles first (fused kernels, trunk kernels); ct < 0: -ct geometry-role workgroups FIRST (k_nbr_fwd / k_nbr_bwd)
        roles = (("colour", B[:ct]), ("geometry", B[ct:])) if ct >= 0 else (("geometry", B[:-ct]), ("f_theta", B[-ct:]))
        for role, blocks in roles:
            blocks = [b for b in blocks if b[0]]      # work-list workgroups of a launch leave no record
            if not blocks:
                continue
            cus = Counter(cu_of(b[2]) for b in blocks)
            starts = [(b[0] - t0) / 100.0 for b in blocks]
            durs = [(b[1] - b[0]) / 100.0 for b in blocks]
            cyc = [b[3] for b in blocks]
            ends = [(b[1] - t0) / 100.0 for b in blocks]
            print(f"  {role:8s}: {len(blocks)} workgroups on {len(cus)} CUs (per CU: {dict(Counter(cus.values()))}); "
                  f"start us min/med/p90/max {min(starts):.1f}/{st.median(starts):.1f}/{q(starts, .9):.1f}/{max(starts):.1f}; "
                  f"duration us min/med/p90/max {min(durs):.1f}/{st.median(durs):.1f}/{q(durs, .9):.1f}/{max(durs):.1f}; "
                  f"cycles med/max {int(st.median(cyc))}/{max(cyc)}; end us med/max {st.median(ends):.1f}/{max(ends):.1f}")
            by_n = defaultdict(list)
            for b in blocks:
                by_n[cus[cu_of(b[2])]].append((b[1] - b[0]) / 100.0)
            print("            duration by workgroups sharing the CU: " +
                  ", ".join(f"{k}: med {st.median(v):.1f} max {max(v):.1f} us (n={len(v)})" for k, v in sorted(by_n.items())))
        xcc = Counter(cu_of(b[2])[0] for b in B if b[0])
        print(f"  workgroups per XCC: {dict(sorted(xcc.items()))}")
        if limit and n + 1 >= limit:
            break


if __name__ == "__main__":
    main(sys.argv[1], int(sys.argv[2]) if len(sys.argv) > 2 else None)
