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
TRACK_TILES = [63]         # 16-sample tiles of the tracker's launch (pmc_probe_meta.json: track_rays)


def _is_track(g):
    return g == TRACK_ITEMS[0]


def _split_items(tiles):
    """work-item counts of the split colour-stage kernels for a launch of `tiles` 16-sample tiles (psl_decode_fwd2.hip)"""
    nbr = ((tiles + 3) // 4 + (tiles * 8 + 3) // 4) * 256
    if tiles <= 256:
        trunk = tiles * 512
    elif tiles <= 512:
        trunk = 256 * 512
    else:
        trunk = (tiles // 2 + (tiles & 1)) * 512
    return dict(nbr=nbr, trunk=trunk, trunk_w=((tiles + 3) // 4) * 256)


def _trk(kind):
    return lambda g: g == _split_items(TRACK_TILES[0])[kind]


# (substring, predicate on the work-item count, class, member): a class launch of the SPLIT colour stage is two kernels (members
# "nbr" and "trunk"); its bytes per launch are the SUM of its members' means
CLASS_OF = [
    ("decode_fwd2ILb1", lambda g: not _is_track(g), "decode_fwd", "fused"),
    ("decode_fwd2ILb1", lambda g: True, "decode_fwd_track", "fused"),
    ("decode_fwd2ILb0", lambda g: True, "decode_fwd_geo", "fused"),
    ("2ILb0ELb1EEEvNS_10DecodeArgsENS_7Bwd2Out", lambda g: True, "decode_bwd", "fused"),
    ("2ILb1ELb1EEEvNS_10DecodeArgsENS_7Bwd2Out", lambda g: True, "decode_bwd_track", "fused"),
    ("2ILb0ELb0EEEvNS_10DecodeArgsENS_7Bwd2Out", lambda g: True, "decode_bwd_geo", "fused"),
    ("2ILb1ELb0EEEvNS_10DecodeArgsENS_7Bwd2Out", lambda g: True, "decode_bwd_geo_track", "fused"),
    ("k_nbr_fwd", _trk("nbr"), "decode_fwd_track", "nbr"),
    ("k_nbr_fwd", lambda g: True, "decode_fwd", "nbr"),
    ("k_trunk_fwd_w", _trk("trunk_w"), "decode_fwd_track", "trunk"),
    ("k_trunk_fwd_w", lambda g: True, "decode_fwd", "trunk"),
    ("k_trunk_fwdE", _trk("trunk"), "decode_fwd_track", "trunk"),
    ("k_trunk_fwdE", lambda g: True, "decode_fwd", "trunk"),
    ("k_trunk_bwdILb1E", lambda g: True, "decode_bwd_track", "trunk"),
    ("k_trunk_bwdILb0E", lambda g: True, "decode_bwd", "trunk"),
    ("k_nbr_bwdILb1E", lambda g: True, "decode_bwd_track", "nbr"),
    ("k_nbr_bwdILb0E", lambda g: True, "decode_bwd", "nbr"),
    ("k_geo_iter", lambda g: True, "geo_iter", "fused"),
    ("k_dwE", lambda g: True, "dw_gemm", "fused"),
    ("AdamRowsSeg", lambda g: True, "adam", "fused"),
    ("SA_SA_SA_SA_ffffiPiSB_Py", lambda g: True, "knn", "fused"),
    ("SB_SB_SB_SB_ffffiPiSC_Py", lambda g: True, "knn", "fused"),       # k_knn_rays_flat<U, MINW, POSE> (TrackPose argument, round 6)
    ("k_map_ray_fused", lambda g: True, "map_ray", "fused"),
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
        for sub, pred, cls, member in CLASS_OF:
            if sub in name and pred(g):
                n, mean = int(parts[2]), float(parts[3])
                a = out.setdefault(cls, {}).setdefault(member, [0, 0.0])
                a[0] += n
                a[1] += mean * n
                break
    # bytes per CLASS launch: the members' means added up (a member is launched once per class launch)
    return {k: sum(v[1] / v[0] for v in m.values() if v[0]) for k, m in out.items()}


def main():
    meta_probe = None
    if len(sys.argv) > 5:
        try:
            meta_probe = json.load(open(sys.argv[5]))
            TRACK_ITEMS[0] = int(meta_probe["track_fwd_items"])
            TRACK_TILES[0] = (5 * int(meta_probe["track_rays"]) + 15) // 16
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
