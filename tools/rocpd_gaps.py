#!/usr/bin/env python
"""Idle time between consecutive kernels of a rocprofv3 rocpd (.db) kernel trace: for every kernel symbol, how long the
device sat idle between the end of the previous kernel (any stream) and its start, on average and in total -- what a
launch-latency-sensitive loop (400 iterations x 5 dependent launches per mapped frame) pays for being five launches.
Usage: python tools/rocpd_gaps.py x_results.db [min_busy_window_us]"""
import sqlite3
import sys
from collections import defaultdict


def main():
    db = sqlite3.connect(sys.argv[1])
    cur = db.cursor()
    sym_cols = [r[1] for r in cur.execute("pragma table_info(rocpd_info_kernel_symbol)")]
    name_col = "kernel_name" if "kernel_name" in sym_cols else ("display_name" if "display_name" in sym_cols else sym_cols[-1])
    rows = cur.execute(f"""select s.{name_col}, d.start, d.end from rocpd_kernel_dispatch d
                           join rocpd_info_kernel_symbol s on d.kernel_id = s.id order by d.start""").fetchall()
    gap_sum, gap_n, dur_sum = defaultdict(float), defaultdict(int), defaultdict(float)
    busy_end = None
    total_gap = total_busy = 0.0
    for name, st, en in rows:
        short = name.split("(")[0][-48:]
        dur_sum[short] += (en - st) / 1e3
        if busy_end is not None:
            g = (st - busy_end) / 1e3
            if 0 < g < 200.0:               # longer pauses are host-side (frame set-up, syncs), not launch gaps
                gap_sum[short] += g
                total_gap += g
            gap_n[short] += 1
        busy_end = en if busy_end is None else max(busy_end, en)
    span = (rows[-1][2] - rows[0][1]) / 1e3
    total_busy = sum(dur_sum.values())
    print(f"kernels {len(rows)}, span {span/1e3:.1f} ms, sum of durations {total_busy/1e3:.1f} ms, launch gaps (< 200 us) {total_gap/1e3:.1f} ms")
    print("kernel,calls,avg_us,avg_gap_before_us,total_gap_ms")
    for k in sorted(gap_sum, key=lambda k: -gap_sum[k])[:24]:
        n = max(gap_n[k], 1)
        print(f"{k},{gap_n[k]},{dur_sum[k]/n:.1f},{gap_sum[k]/n:.2f},{gap_sum[k]/1e3:.2f}")


if __name__ == "__main__":
    main()
