#!/usr/bin/env python3
"""Markdown tables of DESIGN.md section 6 straight from the committed artefacts under profiles/ (so that the document cannot
drift from the files it cites).  usage: python tools/design_tables.py [round-tag, default r05] > fragment.md"""
import json
import os
import sys

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
P = os.path.join(ROOT, "profiles")
TAG = sys.argv[1] if len(sys.argv) > 1 else "r05"
PREV = {"base": 94.92, "base_100steps": 76.65, "replica": 31.46, "replica_100steps": 22.49, "tum": "9.33 on CLEAN depth", "scannet": 12.81,
        "cfg5": 88.36, "cfg1": 130.3}      # round 4 (DESIGN_HISTORY.md section 10.1)


def load(name):
    try:
        return json.load(open(os.path.join(P, f"{TAG}_{name}.json")))
    except Exception:
        return None


def bench_table():
    rows = [("base", "cfg 2 sizes, base iteration mix, 1 M points, 640x480", "`python bench.py`"),
            ("base_100steps", "same, 100 timed frames", "`--steps 100`"),
            ("base_500steps", "same, 500 timed frames (steady state; per-window rates below)", "`--steps 500 --fps-window 100`"),
            ("replica", "Replica yaml (map 5 000 px x 300, track 1 500 px x 40)", "`--mix replica`"),
            ("replica_100steps", "same, 100 timed frames", "`--mix replica --steps 100`"),
            ("replica_500steps", "same, 500 timed frames", "`--mix replica --steps 500 --fps-window 100`"),
            ("replica_6steps", "same, 6 frames after 2 (round 2's command)", "`--mix replica --steps 6 --warmup 2`"),
            ("tum", "TUM yaml (track 5 000 px x 200, map 10 000 px x 150 every 2), NOISY depth: 0.5 % noise, 2 % holes", "`--mix tum`"),
            ("tum_6steps", "same, 6 frames after 2", "`--mix tum --steps 6 --warmup 2`"),
            ("tum_200steps", "same, 200 timed frames", "`--mix tum --steps 200 --fps-window 50`"),
            ("scannet", "ScanNet yaml (exposure latents; track 5 000 px x 100, map 10 000 px x 300)", "`--mix scannet`"),
            ("scannet_200steps", "same, 200 timed frames", "`--mix scannet --steps 200 --fps-window 50`"),
            ("scannet_6steps", "same, 6 frames after 2", "`--mix scannet --steps 6 --warmup 2`"),
            ("cfg5", "cfg 5: 2 M points, 1280x960", "`--points 2000000 --width 1280 --height 960`"),
            ("cfg5_500steps", "same, 500 timed frames", "`--points 2000000 --width 1280 --height 960 --steps 500 --fps-window 100`"),
            ("cfg1", "cfg 1: tracking only, fixed 50 k cloud, 1200x680, 200 frames", "`--track-only --points 50000 --width 1200 --height 680 --mix replica --steps 200`"),
            ("closed_loop_0p5u_100steps", "base mix, CLOSED loop at a Replica-like camera speed (1.4 cm / frame), 100 frames", "`--closed-loop --units-per-frame 0.5 --steps 100`"),
            ("replica_closed_loop_40steps", "Replica yaml, CLOSED loop at full speed (5.6 cm / frame), 40 frames", "`--mix replica --closed-loop --steps 40`"),
            ("replica_closed_loop_100steps", "same, 100 frames", "`--mix replica --closed-loop --steps 100`"),
            ("closed_loop_diverges", "base mix, CLOSED loop at full speed, 40 frames: DIVERGES (tracker budget too small, section 5)", "`--closed-loop --steps 40`")]
    print("| config (BASELINE.json) | command | frames/s | ms/frame | dominant class | HBM traffic / launch (PMC) vs algorithmic | file |")
    print("|---|---|---|---|---|---|---|")
    for key, what, cmd in rows:
        d = load(f"bench_{key}")
        if not d:
            continue
        r = d.get("roofline") or {}
        prev = f" (round 4: {PREV[key]})" if key in PREV else ""
        c = d["config"]
        if not r:
            r = {}
        ate = c.get("ate_rmse_cm")
        loop = "closed" if "closed" in key else ("closed" if key == "cfg1" else "open")
        tr = r.get("traffic")
        traffic = f"{tr / 1e6:.0f} MB = {r.get('traffic_over_algorithmic')} x" if tr else "n/a"
        extra = f"; {loop}-loop pose error {ate} cm rmse / {c.get('ate_max_cm')} cm max" if (ate is not None and key != "cfg1") else ""
        if key == "cfg1":
            extra = f"; trajectory error {c.get('ate_rmse_cm')} cm rmse / {c.get('ate_max_cm')} cm max over {c.get('tracked_frames')} closed-loop frames; CPU oracle {d.get('cpu_baseline', {}).get('value')} frames/s"
        dom = (f"`{r.get('kernel')}` {100 * r.get('frac', 0):.1f} % of {'fp32-MFMA' if r.get('bound') == 'mfma' else 'HBM'} peak "
               f"({r.get('avg_launch_us')} us x {r.get('launches')})") if r.get("kernel") else "(no event pass)"
        print(f"| {what} | {cmd} | **{d['value']:.4g}**{prev} | {d['ms_per_step']:.4g} | {dom}{extra} | {traffic} | `{TAG}_bench_{key}.json` |")


def class_table():
    d = load("bench_base")
    if not d:
        return
    print("\n| class (base mix, event-timed pass) | us / launch | launches | ms / 20 frames | of peak |")
    print("|---|---|---|---|---|")
    for k, v in sorted(d["kernels"].items(), key=lambda kv: -kv[1]["total_ms"]):
        print(f"| {k} | {v['avg_us']:.1f} | {v['launches']} | {v['total_ms']:.1f} | {100 * v['frac']:.1f} % ({v['unit']}) |")
    r = d.get("roofline") or {}
    print(f"\nreconciliation (same frames in both passes): class sum {r.get('class_sum_ms_per_step')} ms/step; pass 2 wall "
          f"{d.get('profiled_ms_per_step')} = class sum + unclassified {r.get('unclassified_ms_per_step')} (of which event-pair overhead "
          f"{r.get('instrumentation_overhead_ms_per_step')}); pass 1 (`value`) wall {d['ms_per_step']} = class sum + "
          f"{r.get('value_pass_unclassified_ms_per_step')} ({r.get('value_pass_unclassified_frac')} of it)")
    s = d.get("split") or {}
    print(f"\nsplit: tracking {s.get('track_ms_per_frame')} ms per frame, mapping {s.get('map_ms_per_mapped_frame')} ms per mapped frame; "
          f"`config.render_loss_rel_err_vs_reference` {d['config'].get('render_loss_rel_err_vs_reference')}; cpu_baseline {d.get('cpu_baseline', {}).get('value')} frames/s "
          f"({d.get('cpu_baseline', {}).get('cores')} threads)")


def sweep_table():
    d = load("roofline_sweep")
    if not d:
        return
    m = d["mapper"] if isinstance(d, dict) else d
    print("\n| samples / launch (mapper, colour stage with F_theta) | forward | backward (dX, ray stage inside below 10 000) | dW GEMM |")
    print("|---|---|---|---|")
    for r in m:
        print(f"| {r['samples']} | {100 * r.get('decode_fwd_frac', 0):.1f} % ({r.get('decode_fwd_us')} us) | {100 * r.get('decode_bwd_frac', 0):.1f} % ({r.get('decode_bwd_us')} us) | "
              f"{100 * r.get('dw_gemm_frac', 0):.1f} % ({r.get('dw_gemm_us')} us) |")
    if isinstance(d, dict) and d.get("tracker"):
        print("\n| samples / launch (tracker: pose gradient) | forward | backward (`k_decode_bwd2<true, true>`) | k-NN |")
        print("|---|---|---|---|")
        for r in d["tracker"]:
            print(f"| {r['samples']} | {100 * r.get('decode_fwd_track_frac', 0):.1f} % ({r.get('decode_fwd_track_us')} us) | "
                  f"{100 * r.get('decode_bwd_track_frac', 0):.1f} % ({r.get('decode_bwd_track_us')} us) | {r.get('knn_us')} us |")


def traffic_table():
    print("\n| class | " + " | ".join(["base", "replica", "tum", "scannet", "cfg5"]) + " |")
    print("|---|---|---|---|---|---|")
    tabs = {m: load(f"pmc_traffic_{m}") or {} for m in ("base", "replica", "tum", "scannet", "cfg5")}
    classes = sorted({k for t in tabs.values() for k in t if not k.startswith("_")})
    for c in classes:
        cells = []
        for m in ("base", "replica", "tum", "scannet", "cfg5"):
            v = tabs[m].get(c)
            cells.append(f"{v['bytes_per_launch'] / 1e6:.1f} ({v['read_bytes'] / 1e6:.0f} r + {v['write_bytes'] / 1e6:.0f} w)" if v else "")
        print(f"| {c} | " + " | ".join(cells) + " |")
    print("\n(MB per launch; launch sizes of each mix as in `profiles/pmc_r05/<mix>/probe_meta.json`)")


def window_table():
    for key in ("base_500steps", "replica_500steps", "tum_200steps", "scannet_200steps", "cfg5_500steps"):
        d = load(f"bench_{key}")
        if not d or not d["config"].get("fps_per_window"):
            continue
        w = d["config"]["fps_per_window"]
        tp = d["config"].get("timed_pass", {})
        it = tp.get("map_iters") or [0]
        print(f"\n`{TAG}_bench_{key}.json`: " + ", ".join(f"frames {x['frames']}: **{x['fps']}** frames/s ({x['points']} points)" for x in w)
              + f"; mapping iterations per mapped frame min / mean / max {min(it)} / {sum(it) / len(it):.0f} / {max(it)}, frustum rows mean "
              f"{sum(tp.get('n_sel', [0])) / max(len(it), 1):.0f}; device memory {d['config'].get('device_memory')}")


if __name__ == "__main__":
    bench_table()
    window_table()
    class_table()
    sweep_table()
    traffic_table()
