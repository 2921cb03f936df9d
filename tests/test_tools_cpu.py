"""CPU: the measurement tools that turn rocprofv3 databases and PMC tables into the files under profiles/ -- run on small
synthetic inputs with the schema this image's rocprofv3 writes (rocpd sqlite), so that a tool edit cannot silently break
the artefacts the bench line reads (`roofline.traffic` comes from tools/pmc_traffic.py's output)."""
import json
import os
import sqlite3
import subprocess
import sys

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
TOOLS = os.path.join(ROOT, "tools")

FWD = "_ZN3psl13k_decode_fwd2ILb1EEEvNS_10DecodeArgsEPKfi.kd"
BWD = "_ZN3psl13k_decode_bwd2ILb0ELb1EEEvNS_10DecodeArgsENS_7Bwd2OutEPKfiNS_7RayFuseENS_12AdamWorklistEi.kd"
BWD_T = "_ZN3psl13k_decode_bwd2ILb1ELb1EEEvNS_10DecodeArgsENS_7Bwd2OutEPKfiNS_7RayFuseENS_12AdamWorklistEi.kd"
DW = "_ZN3psl4k_dwENS_6DwArgsE.kd"


def _trace_db(path):
    db = sqlite3.connect(path)
    db.execute("create table rocpd_info_kernel_symbol (id integer primary key, kernel_name text)")
    db.execute("create table rocpd_kernel_dispatch (id integer primary key, kernel_id integer, start integer, end integer, grid_size_x integer default 0)")
    for i, n in enumerate((FWD, BWD, BWD_T, DW)):
        db.execute("insert into rocpd_info_kernel_symbol values (?, ?)", (i, n))
    t = 1_000_000
    for it in range(40):                     # fwd 48 us, hole 1 us, bwd 49 us, dW 26 us; every 10th iteration a 60-us host stall
        for kid, dur in ((0, 48_000), (1, 49_000), (3, 26_000)):
            db.execute("insert into rocpd_kernel_dispatch (kernel_id, start, end) values (?, ?, ?)", (kid, t, t + dur))
            t += dur + 1_000
        if it % 10 == 9:
            t += 60_000
    db.execute("insert into rocpd_kernel_dispatch (kernel_id, start, end) values (2, ?, ?)", (t, t + 25_000))
    db.commit(); db.close()


def _run(tool, *args):
    p = subprocess.run([sys.executable, os.path.join(TOOLS, tool), *args], capture_output=True, text=True, timeout=120)
    assert p.returncode == 0, p.stderr
    return p.stdout


def test_rocpd_stats_keeps_the_backward_instantiations_apart(tmp_path):
    db = str(tmp_path / "t_results.db")
    _trace_db(db)
    out = _run("rocpd_stats.py", db, "--csv", str(tmp_path / "s.csv"))
    rows = {l.split(",")[0]: l.split(",") for l in out.splitlines()[1:]}
    assert len(rows) == 4                    # mapper and tracker backward are two rows (60-char names used to merge them)
    fwd = next(v for k, v in rows.items() if "k_decode_fwd2ILb1E" in k)
    assert int(fwd[1]) == 40 and abs(float(fwd[3]) - 48.0) < 1e-6
    bwd_t = next(v for k, v in rows.items() if "bwd2ILb1ELb1E" in k)
    assert int(bwd_t[1]) == 1
    assert open(tmp_path / "s.csv").read().strip() == out.strip()


def test_rocpd_stats_splits_a_kernel_by_grid_size(tmp_path):
    """--by-grid: the Adam launches of the geometry stage (one row group) and of the colour stage (two groups + decoder parameters)
    are one symbol with two grid sizes; the split is what profiles/r04_adam_lanes_per_row.txt records."""
    db = str(tmp_path / "t_results.db")
    _trace_db(db)
    con = sqlite3.connect(db)
    con.execute("insert into rocpd_info_kernel_symbol values (9, '_ZN3psl15k_map_adam_lazyILi16EEEvNS_11AdamRowsSegE.kd')")
    for i in range(6):
        g, dur = ((524288, 13_000) if i % 3 else (1919744, 21_000))
        con.execute("insert into rocpd_kernel_dispatch (kernel_id, start, end, grid_size_x) values (9, ?, ?, ?)", (10**9 + i * 10**5, 10**9 + i * 10**5 + dur, g))
    con.commit(); con.close()
    out = _run("rocpd_stats.py", db, "--by-grid", "adam_lazy")
    by = [l for l in out.splitlines() if l.startswith("by-grid")]
    assert len(by) == 2
    small = next(l for l in by if "grid_x=524288" in l).split(",")
    big = next(l for l in by if "grid_x=1919744" in l).split(",")
    assert int(small[1]) == 4 and abs(float(small[3]) - 13.0) < 1e-6
    assert int(big[1]) == 2 and abs(float(big[3]) - 21.0) < 1e-6


def test_rocpd_timeline_and_gaps(tmp_path):
    db = str(tmp_path / "t_results.db")
    _trace_db(db)
    out = _run("rocpd_timeline.py", db, "1.0")
    head = out.splitlines()[0]
    busy = float(head.split("busy")[1].split("ms")[0])
    assert abs(busy - 40 * (48 + 49 + 26) / 1e3) < 0.2           # window = span of the k_dw launches
    long_holes = [l for l in out.splitlines() if l.strip().startswith("holes") and "20.." in l][0]
    assert int(long_holes.split(":")[1].split("holes")[0]) == 3  # the host stalls inside the window (the fourth ends it)
    assert "k_decode_fwd2" in out.split("long holes")[1]         # ... are charged to the kernel that follows them
    gaps = _run("rocpd_gaps.py", db)
    assert "kernels 121" in gaps.splitlines()[0]


def test_rocpd_window_cuts_the_trace_at_the_bench_markers(tmp_path):
    """tools/rocpd_window.py: the kernels between bench.py's marker launches (PSL_BENCH_MARK=1), device-busy union, holes and
    their attribution, per pass."""
    db = str(tmp_path / "t_results.db")
    _trace_db(db)
    con = sqlite3.connect(db)
    con.execute("insert into rocpd_info_kernel_symbol values (7, 'void at::native::erfinv_kernel_something(int).kd')")
    # markers: before iteration 0, after iteration 19 (pass 1), before iteration 20, after the last kernel (pass 2)
    per_it = 48_000 + 49_000 + 26_000 + 3_000
    t_mid = 1_000_000 + 20 * per_it + 2 * 60_000
    for st in (999_000, t_mid - 900, t_mid - 600, 1_000_000 + 40 * per_it + 4 * 60_000 + 26_000):
        con.execute("insert into rocpd_kernel_dispatch (kernel_id, start, end) values (7, ?, ?)", (st, st + 100))
    con.commit(); con.close()
    js = str(tmp_path / "w.json")
    out = _run("rocpd_window.py", db, "--json", js)
    res = json.load(open(js))
    assert len(res) == 2 and res[0]["kernels"] == 60 and res[1]["kernels"] == 61
    assert abs(res[0]["busy_ms"] - 20 * 0.123) < 1e-6
    # pass 1: one 61-us host stall inside (after iteration 9; the one after iteration 19 lies behind the last kernel)
    assert sum(h["n"] for h in res[0]["holes"] if h["lo_us"] >= 30) == 1
    assert "k_decode_fwd2<1>" in res[0]["long_holes_by_next"] and "psl4k_dw" in "".join(res[0]["long_holes_by_prev"])
    assert "== pass 2" in out


def test_pmc_traffic_classes_and_fetch_correction(tmp_path):
    def table(counter, rows):
        p = tmp_path / f"{counter}.csv"
        p.write_text("# x\nkernel,counter,dispatches,mean,sum\n" +
                     "".join(f"{n}@{g},{counter},{k},{m},{m * k}\n" for n, g, k, m in rows))
        return str(p)
    f = table("FETCH_SIZE", [(FWD, 320512, 2, 9000.0), (FWD, 64512, 2, 3500.0), (BWD, 320512, 2, 28000.0), (DW, 64768, 2, 50000.0),
                             ("_ZN3psl10k_geo_iterENS_10DecodeArgsE.kd", 41344, 2, 6000.0)])
    w = table("WRITE_SIZE", [(FWD, 320512, 2, 65000.0), (FWD, 64512, 2, 9000.0), (BWD, 320512, 2, 58000.0), (DW, 64768, 2, 12000.0),
                             ("_ZN3psl10k_geo_iterENS_10DecodeArgsE.kd", 41344, 2, 3800.0)])
    out = str(tmp_path / "traffic.json")
    _run("pmc_traffic.py", f, w, out, "abc1234")
    d = json.load(open(out))
    assert d["_meta"]["commit"] == "abc1234"
    assert set(d) >= {"decode_fwd", "decode_fwd_track", "decode_bwd", "dw_gemm", "geo_iter"}
    # FETCH_SIZE (KiB) doubled for the 16-B/lane streams, WRITE_SIZE as reported (MI355X_MICROARCH.md)
    assert d["decode_fwd"]["read_bytes"] == round(9000.0 * 1024 * 2) and d["decode_fwd"]["write_bytes"] == round(65000.0 * 1024)
    assert d["decode_fwd_track"]["read_bytes"] == round(3500.0 * 1024 * 2)       # the tracker's launches are a class of their own
    assert d["decode_bwd"]["bytes_per_launch"] == round(28000.0 * 1024 * 2 + 58000.0 * 1024)


def test_block_trace_skips_untraced_workgroups(tmp_path):
    rec = {"kernel": "bwd2", "P": 32, "flags": 0, "grid": 6, "color_tiles": 2, "threads": 512,
           "blocks": [[100, 3000, 0x1000, 7000], [120, 3100, 0x2100, 7100], [110, 500, 0x1200, 900], [115, 520, 0x3300, 950],
                      [0, 0, 0, 0], [0, 0, 0, 0]]}
    p = tmp_path / "b.jsonl"
    p.write_text(json.dumps(rec) + "\n")
    out = _run("block_trace.py", str(p))
    assert "first start -> last end 30.0 us" in out and "colour  : 2 workgroups" in out and "geometry: 2 workgroups" in out


def test_decode_kernels_fit_their_register_budget_without_scratch():
    """tools/kernel_resources.py cross-compiles the decode kernels for gfx950 and reads hipcc's own metadata: every one of
    them free of scratch, and the two colour-stage kernels that must run TWO 512-thread workgroups per CU (4 waves per SIMD)
    within 128 VGPRs -- round 3 shipped the forward with 19 spilled registers and the tracker's backward at 189 VGPRs (one
    workgroup per CU, 16.5 % of peak)."""
    import os
    import shutil
    import pytest
    if not shutil.which("/opt/rocm/bin/hipcc"):
        pytest.skip("hipcc not installed")
    from tools import kernel_resources as KR
    seen = {}
    for f in ("psl_decode_fwd2.hip", "psl_decode_bwd2.hip", "psl_decode_geo.hip", "psl_trunk_wave.hip"):
        for k in KR.resources(os.path.join(KR.CSRC, f)):
            seen[k["name"]] = k
            assert k["vgpr_spill_count"] == 0 and k["private_segment_fixed_size"] == 0, k
    two_per_cu = [k for n, k in seen.items() if ("k_decode_fwd2ILb1E" in n or "k_decode_bwd2ILb0ELb1E" in n or "k_decode_bwd2ILb1ELb1E" in n)]
    assert len(two_per_cu) == 3
    for k in two_per_cu:
        assert k["vgpr_count"] + k["agpr_count"] <= 128 and k["max_flat_workgroup_size"] == 512, k
    # the split colour stage (round 6): three four-wavefront workgroups per CU for the F_theta kernels and three wavefronts per
    # SIMD for the wave-per-tile trunk (<= 168 registers), one 512-thread trunk workgroup per CU (<= 256)
    three = [k for n, k in seen.items() if ("k_nbr_fwd" in n or "k_nbr_bwd" in n or "k_trunk_fwd_w" in n)]
    assert len(three) == 7 and all(k["vgpr_count"] + k["agpr_count"] <= 168 for k in three), [(k["name"], k["vgpr_count"]) for k in three]
    one = [k for n, k in seen.items() if ("k_trunk_fwdE" in n or "k_trunk_bwdILb" in n)]
    assert len(one) == 3 and all(k["vgpr_count"] + k["agpr_count"] <= 256 for k in one)


def test_kernel_resources_parses_metadata():
    from tools import kernel_resources as KR
    text = """
amdhsa.kernels:
  - .agpr_count:     4
    .args: []
    .group_segment_fixed_size: 1024
    .max_flat_workgroup_size: 256
    .name:           _Z3fooPf
    .private_segment_fixed_size: 16
    .sgpr_count:     20
    .sgpr_spill_count: 0
    .vgpr_count:     33
    .vgpr_spill_count: 2
  - .agpr_count:     0
    .group_segment_fixed_size: 0
    .max_flat_workgroup_size: 64
    .name:           _Z3barPf
    .private_segment_fixed_size: 0
    .sgpr_count:     8
    .sgpr_spill_count: 0
    .vgpr_count:     7
    .vgpr_spill_count: 0
amdhsa.target:   amdgcn-amd-amdhsa--gfx950
"""
    ks = KR.parse_asm(text)
    assert [k["name"] for k in ks] == ["_Z3fooPf", "_Z3barPf"]
    assert ks[0]["vgpr_count"] == 33 and ks[0]["agpr_count"] == 4 and ks[0]["vgpr_spill_count"] == 2 and ks[0]["private_segment_fixed_size"] == 16
    assert ks[1]["max_flat_workgroup_size"] == 64
