#!/usr/bin/env python
"""Busy / idle split of the device over the steady-state tail of a rocprofv3 rocpd (.db) kernel trace of bench.py: the union
of kernel intervals over all streams is "busy"; the holes are binned by length and the long ones are attributed to the
kernel that ends them (a host-side stall shows up as a long hole in front of the first kernel the host launches next).
Usage: python tools/rocpd_timeline.py x_results.db [tail_fraction=0.4]   (tail of the span in which k_dw runs)"""
import sqlite3
import sys
from collections import defaultdict


def main():
    db = sqlite3.connect(sys.argv[1])
    frac = float(sys.argv[2]) if len(sys.argv) > 2 else 0.4
    cur = db.cursor()
    sym_cols = [r[1] for r in cur.execute("pragma table_info(rocpd_info_kernel_symbol)")]
    name_col = "kernel_name" if "kernel_name" in sym_cols else ("display_name" if "display_name" in sym_cols else sym_cols[-1])
    rows = cur.execute(f"""select s.{name_col}, d.start, d.end from rocpd_kernel_dispatch d
                           join rocpd_info_kernel_symbol s on d.kernel_id = s.id order by d.start""").fetchall()
    # the mapper loop is the only caller of k_dw: the window is the last `frac` of the span between its first and last launch
    dw = [r for r in rows if "k_dw" in r[0]]
    t0, t1 = (dw[0][1], dw[-1][2]) if dw else (rows[0][1], rows[-1][2])
    cut = t1 - frac * (t1 - t0)
    rows = [r for r in rows if cut <= r[1] <= t1]
    bins = [(0, 3), (3, 20), (20, 200), (200, 1e12)]
    hole_sum, hole_n = [0.0] * len(bins), [0] * len(bins)
    by_next = defaultdict(lambda: [0.0, 0])
    busy = 0.0
    cur_s, cur_e = rows[0][1], rows[0][2]
    for name, st, en in rows[1:]:
        if st > cur_e:
            busy += (cur_e - cur_s) / 1e3
            g = (st - cur_e) / 1e3
            for b, (lo, hi) in enumerate(bins):
                if lo <= g < hi:
                    hole_sum[b] += g
                    hole_n[b] += 1
            if g >= 20.0:
                k = name.split("(")[0][-56:]
                by_next[k][0] += g
                by_next[k][1] += 1
            cur_s, cur_e = st, en
        else:
            cur_e = max(cur_e, en)
    busy += (cur_e - cur_s) / 1e3
    span = (rows[-1][2] - rows[0][1]) / 1e3
    print(f"window {span/1e3:.1f} ms ({len(rows)} kernels): busy {busy/1e3:.1f} ms = {100*busy/span:.1f} %")
    for (lo, hi), s, n in zip(bins, hole_sum, hole_n):
        print(f"  holes {lo:>4g}..{hi if hi < 1e11 else 'inf':>4} us: {n:6d} holes, {s/1e3:8.2f} ms = {100*s/span:5.1f} %")
    print("long holes (>= 20 us) by the kernel that ends them: kernel,holes,total_ms,avg_us")
    for k in sorted(by_next, key=lambda k: -by_next[k][0])[:16]:
        s, n = by_next[k]
        print(f"  {k},{n},{s/1e3:.2f},{s/n:.1f}")


if __name__ == "__main__":
    main()
