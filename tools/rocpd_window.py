#!/usr/bin/env python
"""Where a timed pass of bench.py goes, from a rocprofv3 rocpd (.db) kernel trace of `PSL_BENCH_MARK=1 python bench.py ...`:
bench.py launches a marker kernel (torch.erfinv_, used by nothing else) right before and right after each timed pass;
for every window between an opening and a closing marker this prints
  * wall span, device-busy time (union of kernel intervals over all streams) and idle time,
  * kernel time by symbol (sum of durations; `overlapped` = time during which another kernel was also running),
  * the idle holes binned by length, the long ones attributed to the kernel that ENDS them and the kernel BEFORE them
    (a host-side stall shows up as a long hole in front of the first kernel the host launches next).
Usage: python tools/rocpd_window.py x_results.db [--marker erfinv] [--top 30] [--json out.json]"""
import json
import sqlite3
import sys
from collections import defaultdict


def short(name):
    n = name.split("(")[0]
    for key in ("k_decode_fwd2ILb1", "k_decode_fwd2ILb0", "k_decode_bwd2ILb0ELb1", "k_decode_bwd2ILb1ELb1", "k_decode_bwd2ILb0ELb0",
                "k_decode_bwd2ILb1ELb0"):
        if key in n:
            return key.replace("ILb", "<").replace("ELb", ",") + ">"
    for key in ("k_nbr_fwdILb1E", "k_nbr_fwdILb0E", "k_trunk_fwd_w", "k_trunk_fwdE", "k_trunk_bwdILb0E", "k_trunk_bwdILb1E", "k_nbr_bwdILb0ELb1E",
                "k_nbr_bwdILb0ELb0E", "k_nbr_bwdILb1ELb1E", "k_nbr_bwdILb1ELb0E"):       # the split colour stage (round 6)
        if key in n:
            return key.rstrip("E").replace("ILb", "<").replace("ELb", ",") + (">" if "ILb" in key else "")
    if "psl" in n:
        i = n.find("psl")
        return n[i:i + 40]
    return n[-56:]


def load(path):
    db = sqlite3.connect(path)
    cur = db.cursor()
    sym_cols = [r[1] for r in cur.execute("pragma table_info(rocpd_info_kernel_symbol)")]
    name_col = "kernel_name" if "kernel_name" in sym_cols else ("display_name" if "display_name" in sym_cols else sym_cols[-1])
    return cur.execute(f"""select s.{name_col}, d.start, d.end from rocpd_kernel_dispatch d
                           join rocpd_info_kernel_symbol s on d.kernel_id = s.id order by d.start""").fetchall()


def analyse(rows, top=30):
    span = (rows[-1][2] - rows[0][1]) / 1e3
    dur, cnt = defaultdict(float), defaultdict(int)
    for n, s, e in rows:
        k = short(n)
        dur[k] += (e - s) / 1e3
        cnt[k] += 1
    bins = [(0, 3), (3, 10), (10, 30), (30, 100), (100, 1000), (1000, 1e12)]
    hole_sum, hole_n = [0.0] * len(bins), [0] * len(bins)
    by_next, by_prev = defaultdict(lambda: [0.0, 0]), defaultdict(lambda: [0.0, 0])
    busy = 0.0
    cur_s, cur_e, last = rows[0][1], rows[0][2], short(rows[0][0])
    for n, st, en in rows[1:]:
        if st > cur_e:
            busy += (cur_e - cur_s) / 1e3
            g = (st - cur_e) / 1e3
            for b, (lo, hi) in enumerate(bins):
                if lo <= g < hi:
                    hole_sum[b] += g
                    hole_n[b] += 1
            if g >= 10.0:
                by_next[short(n)][0] += g; by_next[short(n)][1] += 1
                by_prev[last][0] += g; by_prev[last][1] += 1
            cur_s, cur_e, last = st, en, short(n)
        else:
            if en > cur_e:
                cur_e, last = en, short(n)
    busy += (cur_e - cur_s) / 1e3
    # outliers: launches that took more than 8x the median of their symbol (and more than 100 us), with what ran around them
    by_sym = defaultdict(list)
    for i, (n, st, en) in enumerate(rows):
        by_sym[short(n)].append((en - st) / 1e3)
    med = {k: sorted(v)[len(v) // 2] for k, v in by_sym.items()}
    outliers = []
    for i, (n, st, en) in enumerate(rows):
        k, d = short(n), (en - st) / 1e3
        if d > 100.0 and d > 8.0 * med[k]:
            conc = sorted({short(m) for (m, s2, e2) in rows[max(0, i - 40):i + 40] if s2 < en and e2 > st and (m, s2, e2) != (n, st, en)})
            outliers.append(dict(kernel=k, us=round(d, 1), median_us=round(med[k], 1), at_ms=round((st - rows[0][1]) / 1e6, 3),
                                 prev=short(rows[i - 1][0]) if i else None, concurrent=conc[:6]))
    total = sum(dur.values())
    out = dict(span_ms=span / 1e3, busy_ms=busy / 1e3, idle_ms=(span - busy) / 1e3, kernel_sum_ms=total / 1e3,
               overlapped_ms=(total - busy) / 1e3, kernels=len(rows),
               by_kernel={k: dict(calls=cnt[k], total_ms=round(dur[k] / 1e3, 3), avg_us=round(dur[k] / cnt[k], 2))
                          for k in sorted(dur, key=lambda k: -dur[k])[:top]},
               outliers=outliers[:40],
               holes=[dict(lo_us=lo, hi_us=(hi if hi < 1e11 else None), n=n, total_ms=round(s / 1e3, 3))
                      for (lo, hi), s, n in zip(bins, hole_sum, hole_n)],
               long_holes_by_next={k: dict(n=v[1], total_ms=round(v[0] / 1e3, 3)) for k, v in
                                   sorted(by_next.items(), key=lambda kv: -kv[1][0])[:16]},
               long_holes_by_prev={k: dict(n=v[1], total_ms=round(v[0] / 1e3, 3)) for k, v in
                                   sorted(by_prev.items(), key=lambda kv: -kv[1][0])[:16]})
    return out


def main():
    a = sys.argv[1:]
    path = a[0]
    marker = a[a.index("--marker") + 1] if "--marker" in a else "erfinv"
    top = int(a[a.index("--top") + 1]) if "--top" in a else 30
    rows = load(path)
    marks = [i for i, r in enumerate(rows) if marker in r[0]]
    if len(marks) < 2:
        raise SystemExit(f"fewer than two '{marker}' kernels in the trace: run bench.py with PSL_BENCH_MARK=1")
    res = []
    for w in range(0, len(marks) - 1, 2):
        seg = rows[marks[w] + 1:marks[w + 1]]
        if not seg:
            continue
        r = analyse(seg, top)
        r["pass"] = w // 2 + 1
        r["marker_to_marker_ms"] = (rows[marks[w + 1]][1] - rows[marks[w]][2]) / 1e6
        res.append(r)
        print(f"== pass {r['pass']}: marker to marker {r['marker_to_marker_ms']:.2f} ms, first to last kernel {r['span_ms']:.2f} ms, "
              f"{r['kernels']} kernels; device busy {r['busy_ms']:.2f} ms ({100 * r['busy_ms'] / r['span_ms']:.1f} %), idle "
              f"{r['idle_ms']:.2f} ms; sum of kernel durations {r['kernel_sum_ms']:.2f} ms (overlapped {r['overlapped_ms']:.2f} ms)")
        print("kernel,calls,total_ms,avg_us,pct_of_span")
        for k, v in r["by_kernel"].items():
            print(f"  {k},{v['calls']},{v['total_ms']},{v['avg_us']},{100 * v['total_ms'] / r['span_ms']:.1f}")
        for o in r["outliers"]:
            print(f"  OUTLIER {o['kernel']}: {o['us']} us (median {o['median_us']}) at +{o['at_ms']} ms, after {o['prev']}, concurrent with {o['concurrent']}")
        print("idle holes: range_us,n,total_ms")
        for h in r["holes"]:
            print(f"  {h['lo_us']}..{h['hi_us']},{h['n']},{h['total_ms']}")
        print("holes >= 10 us by the kernel that ends them: kernel,n,total_ms")
        for k, v in r["long_holes_by_next"].items():
            print(f"  {k},{v['n']},{v['total_ms']}")
        print("holes >= 10 us by the kernel before them: kernel,n,total_ms")
        for k, v in r["long_holes_by_prev"].items():
            print(f"  {k},{v['n']},{v['total_ms']}")
    # outliers over the WHOLE trace (set-up included): which launches took far longer than their symbol's median, and when
    whole = analyse(rows, top)
    t_first_mark = rows[marks[0]][1]
    print(f"== whole trace: {len(rows)} kernels; outliers (offset relative to the first marker):")
    for o in whole["outliers"]:
        print(f"  OUTLIER {o['kernel']}: {o['us']} us (median {o['median_us']}) at {o['at_ms'] - (t_first_mark - rows[0][1]) / 1e6:+.1f} ms, "
              f"after {o['prev']}, concurrent with {o['concurrent']}")
    if "--json" in a:
        json.dump(res, open(a[a.index("--json") + 1], "w"), indent=1)


if __name__ == "__main__":
    main()
