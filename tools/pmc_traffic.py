#!/usr/bin/env python
"""HBM traffic per launch of every kernel class from the two PMC passes of tools/pmc_run.sh (FETCH_SIZE and WRITE_SIZE,
separate rocprofv3 --pmc runs on tools/pmc_probe.py: the same launches bench.py times, 1 M points).

Units and corrections as /opt/skills/guides/MI355X_MICROARCH.md prescribes: both counters are reported in KiB;
on gfx950 FETCH_SIZE tallies the 128-byte requests of wide (16 B/lane) streaming reads at 64 B, so it is DOUBLED for
the kernels whose reads are such streams (the weight-fragment and activation streams of the register-chained decode
kernels, the dW GEMM, Adam); WRITE_SIZE is taken as reported (uncalibrated there).

usage: python tools/pmc_traffic.py gpurun_out/pmc_<tag>/fetch.csv gpurun_out/pmc_<tag>/write.csv out.json [commit [probe_meta.json]]
"""
import json
import sys

# (substring of the (possibly left-truncated) mangled kernel name, predicate on the launch's work-item count, class).
# One symbol serves the tracker's and the mapper's colour-stage forward: they are told apart by the work-item count of the
# TRACKER's launch, which tools/pmc_probe.py writes to gpurun_out/pmc_probe_meta.json (base mix: 126 x 512 = 64 512).
TRACK_ITEMS = [64512]


def _is_track(g):
    return g == TRACK_ITEMS[0]


CLASS_OF = [
    ("decode_fwd2ILb1", lambda g: not _is_track(g), "decode_fwd"),
    ("decode_fwd2ILb1", lambda g: True, "decode_fwd_track"),
    ("decode_fwd2ILb0", lambda g: True, "decode_fwd_geo"),
    ("2ILb0ELb1EEEvNS_10DecodeArgsENS_7Bwd2Out", lambda g: True, "decode_bwd"),
    ("2ILb1ELb1EEEvNS_10DecodeArgsENS_7Bwd2Out", lambda g: True, "decode_bwd_track"),
    ("2ILb0ELb0EEEvNS_10DecodeArgsENS_7Bwd2Out", lambda g: True, "decode_bwd_geo"),
    ("2ILb1ELb0EEEvNS_10DecodeArgsENS_7Bwd2Out", lambda g: True, "decode_bwd_geo_track"),
    ("k_geo_iter", lambda g: True, "geo_iter"),
    ("k_dwE", lambda g: True, "dw_gemm"),
    ("AdamRowsSeg", lambda g: True, "adam"),
    ("SA_SA_SA_SA_ffffiPiSB_Py", lambda g: True, "knn"),
    ("k_map_ray_fused", lambda g: True, "map_ray"),
]
DOUBLE_FETCH = {"geo_iter", "decode_fwd", "decode_fwd_track", "decode_fwd_geo", "decode_bwd", "decode_bwd_track", "decode_bwd_geo",
                "dw_gemm", "adam"}


def read(path, counter):
    out = {}
    for line in open(path):
        if line.startswith("#") or line.startswith("kernel,"):
            continue
        parts = line.strip().split(",")
        if len(parts) != 5 or parts[1] != counter:
            continue
        name, _, grid = parts[0].rpartition("@")
        try:
            g = int(float(grid))
        except ValueError:
            g = 0
        for sub, pred, cls in CLASS_OF:
            if sub in name and pred(g):
                n, mean = int(parts[2]), float(parts[3])
                a = out.setdefault(cls, [0, 0.0])
                a[0] += n
                a[1] += mean * n
                break
    return {k: v[1] / v[0] for k, v in out.items() if v[0]}


def main():
    meta_probe = None
    if len(sys.argv) > 5:
        try:
            meta_probe = json.load(open(sys.argv[5]))
            TRACK_ITEMS[0] = int(meta_probe["track_fwd_items"])
        except Exception:
            meta_probe = None
    fetch, write = read(sys.argv[1], "FETCH_SIZE"), read(sys.argv[2], "WRITE_SIZE")
    res = {}
    for cls in sorted(set(fetch) | set(write)):
        f_kib, w_kib = fetch.get(cls, 0.0), write.get(cls, 0.0)
        f_b = f_kib * 1024 * (2 if cls in DOUBLE_FETCH else 1)
        w_b = w_kib * 1024
        res[cls] = dict(bytes_per_launch=round(f_b + w_b), read_bytes=round(f_b), write_bytes=round(w_b),
                        fetch_size_kib_raw=round(f_kib, 1), write_size_kib_raw=round(w_kib, 1),
                        fetch_doubled=cls in DOUBLE_FETCH,
                        source="rocprofv3 --pmc FETCH_SIZE / WRITE_SIZE, separate passes, tools/pmc_probe.py")
    res["_meta"] = dict(commit=sys.argv[4] if len(sys.argv) > 4 else None, command="bash tools/gpu_round.sh <tag> pmc:<mix> (tools/pmc_probe.py)",
                        passes=["--pmc FETCH_SIZE", "--pmc WRITE_SIZE"], probe=meta_probe)
    json.dump(res, open(sys.argv[3], "w"), indent=1)
    print(json.dumps(res, indent=1))


if __name__ == "__main__":
    main()
