#!/usr/bin/env python
"""Per-kernel mean of every PMC counter in a rocprofv3 rocpd database.
usage: python tools/rocpd_pmc.py file.db [more.db ...]"""
import sqlite3
import sys


def main():
    for path in sys.argv[1:]:
        db = sqlite3.connect(path)
        cur = db.cursor()
        cols = [r[1] for r in cur.execute("pragma table_info(rocpd_pmc_event)")]
        pcols = [r[1] for r in cur.execute("pragma table_info(rocpd_info_pmc)")]
        kcols = [r[1] for r in cur.execute("pragma table_info(rocpd_kernel_dispatch)")]
        scols = [r[1] for r in cur.execute("pragma table_info(rocpd_info_kernel_symbol)")]
        name_col = "kernel_name" if "kernel_name" in scols else scols[-1]
        pname = "name" if "name" in pcols else ("symbol" if "symbol" in pcols else pcols[-1])
        # launches of one kernel at different sizes (tracker batch / mapper batch) are different classes: keep the grid
        gcol = next((c for c in ("grid_size_x", "grid_x", "grid_size") if c in kcols), None)
        gsel = f"d.{gcol}" if gcol else "0"
        q = f"""select s.{name_col} || '@' || {gsel}, p.{pname}, count(*), avg(e.value), sum(e.value)
                from rocpd_pmc_event e join rocpd_info_pmc p on e.pmc_id = p.id
                join rocpd_kernel_dispatch d on e.event_id = d.event_id
                join rocpd_info_kernel_symbol s on d.kernel_id = s.id
                group by s.{name_col}, {gsel}, p.{pname} order by s.{name_col}"""
        try:
            rows = cur.execute(q).fetchall()
        except Exception as ex:
            print(path, "query failed:", ex, "| pmc_event cols:", cols, "| dispatch cols:", kcols)
            continue
        print("#", path)
        print("kernel,counter,dispatches,mean,sum")
        for n, c, k, m, sm in rows:
            if "psl" not in n:
                continue
            base, _, grid = n.rpartition("@")
            print(f"{base.split('(')[0][-110:]}@{grid},{c},{k},{m:.1f},{sm:.0f}")


if __name__ == "__main__":
    main()
