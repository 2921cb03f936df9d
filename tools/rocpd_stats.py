#!/usr/bin/env python
"""Summarise a rocprofv3 rocpd (.db) kernel trace: per-kernel calls, total/avg/min/max duration.
Usage: python tools/rocpd_stats.py gpurun_out/prof/x_results.db [--csv out.csv] [--by-grid <substring of the kernel name>]
--by-grid splits the launches of the matching kernels by grid size (e.g. the Adam launches of the geometry stage -- feature rows
only -- from those of the colour stage, which also carry the decoder parameters)."""
import sqlite3
import sys


def main():
    db = sqlite3.connect(sys.argv[1])
    cur = db.cursor()
    cols = [r[1] for r in cur.execute("pragma table_info(rocpd_kernel_dispatch)")]
    sym_cols = [r[1] for r in cur.execute("pragma table_info(rocpd_info_kernel_symbol)")]
    name_col = "kernel_name" if "kernel_name" in sym_cols else ("display_name" if "display_name" in sym_cols else sym_cols[-1])
    q = f"""select s.{name_col}, count(*), sum(d.end-d.start), avg(d.end-d.start), min(d.end-d.start), max(d.end-d.start)
            from rocpd_kernel_dispatch d join rocpd_info_kernel_symbol s on d.kernel_id = s.id
            group by s.{name_col} order by 3 desc"""
    rows = cur.execute(q).fetchall()
    tot = sum(r[2] for r in rows) or 1
    lines = ["kernel,calls,total_us,avg_us,min_us,max_us,pct"]
    for n, c, t, a, mn, mx in rows:
        short = n.split("(")[0][-110:]      # long enough to keep the template arguments of k_decode_bwd2 apart
        lines.append(f"{short},{c},{t/1e3:.1f},{a/1e3:.2f},{mn/1e3:.2f},{mx/1e3:.2f},{100*t/tot:.1f}")
    if "--by-grid" in sys.argv:
        pat = sys.argv[sys.argv.index("--by-grid") + 1]
        gcol = next((c for c in cols if "grid" in c and c.endswith("x")), None)
        if not gcol:
            print("by-grid: no grid column among", cols, file=sys.stderr)
        else:
            q = f"""select s.{name_col}, d.{gcol}, count(*), avg(d.end-d.start), min(d.end-d.start), max(d.end-d.start)
                    from rocpd_kernel_dispatch d join rocpd_info_kernel_symbol s on d.kernel_id = s.id
                    where s.{name_col} like ? group by s.{name_col}, d.{gcol} order by 3 desc"""
            for n, g, c, a, mn, mx in cur.execute(q, (f"%{pat}%",)).fetchall()[:12]:
                lines.append(f"by-grid {n.split('(')[0][-60:]} grid_x={g},{c},,{a/1e3:.2f},{mn/1e3:.2f},{mx/1e3:.2f},")
    out = "\n".join(lines)
    if "--csv" in sys.argv:
        open(sys.argv[sys.argv.index("--csv") + 1], "w").write(out + "\n")
    print(out)


if __name__ == "__main__":
    main()
