"""Wall time of one frame-parallel exchange (point_slam_amd/dist.py) at the headline map size: 1 M points, 8 rank blocks.
One process, one GPU: the seven other ranks are SIMULATED by a transport that returns this rank's records followed by
seven prepared blocks (new points observed from neighbouring frames -- the local block shifted by a fraction of the add
radius, so the cross-rank dedupe has real work -- and as many trained rows as this rank sends), i.e. everything the
exchange does on the device except the wire time of the two all-gathers.  Prints one JSON line.
usage (GPU box): python tools/exchange_timing.py [--points 1000000] [--blocks 8] [--new-locations 1000] [--rows 40000]"""
import argparse, json, os, sys, time
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
import torch.distributed as dist
from point_slam_amd import synthetic as syn
from point_slam_amd.config import default_config
from point_slam_amd.dist import FrameParallelSync, REC, REC_ROW
from point_slam_amd.neural_point import HipNeuralPointCloud


class SimTransport:
    """allgather_v of a world of `blocks` ranks of which this process is rank 0."""
    def __init__(self, blocks):
        self.blocks = blocks
        self.pending = None          # list of (blocks - 1) tensors for the next call

    def allgather_v(self, rec):
        other = self.pending or []
        self.pending = None
        return torch.cat([rec] + other, 0), [rec.shape[0]] + [o.shape[0] for o in other]


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--points", type=int, default=1_000_000)
    ap.add_argument("--blocks", type=int, default=8)
    ap.add_argument("--new-locations", type=int, default=1000)
    ap.add_argument("--rows", type=int, default=40000)
    ap.add_argument("--repeats", type=int, default=63)
    ap.add_argument("--warmup", type=int, default=3, help="calls left out of the percentiles (first-touch: RCCL communicator, allocator)")
    ap.add_argument("--backend", default="nccl")
    ap.add_argument("--out", default=None)
    a = ap.parse_args()
    os.environ.setdefault("MASTER_ADDR", "127.0.0.1"); os.environ.setdefault("MASTER_PORT", "29541")
    dev = torch.device("cuda:0")
    torch.cuda.set_device(dev)
    dist.init_process_group(a.backend, rank=0, world_size=1)    # nccl = RCCL: the decoder all-reduce is a device collective
    cfg = default_config()
    cam = syn.intrinsics(640, 480)
    npc = HipNeuralPointCloud(cfg, max_points=a.points + 400_000 + 12_000 * a.repeats, device="cuda:0")
    base = syn.seed_cloud(cam, a.points, n_views=64, seed=1219).to(dev)
    g = torch.Generator(device="cpu").manual_seed(5)
    npc.set_points(base, torch.randn(a.points, 32, generator=g).to(dev), torch.randn(a.points, 32, generator=g).to(dev))
    theta = torch.zeros(70000, device=dev)
    sync = FrameParallelSync(npc, theta, n_color=60000)
    tr = SimTransport(a.blocks)
    sync.transport = tr
    r_add = float(cfg["pointcloud"]["radius_add_max"]) if "radius_add_max" in cfg["pointcloud"] else 0.08
    times, parts = [], []
    for it in range(a.repeats):
        n_base = npc.pts_num()
        # this rank's new surface locations: off the existing surface so they pass the local add test, three points each
        gi = torch.Generator(device="cpu").manual_seed(100 + it)
        pick = torch.randint(0, a.points, (a.new_locations,), generator=gi).to(dev)
        surf = base[pick] + torch.tensor([0.0, 0.0, 3.0 + it], device=dev)
        trip = torch.stack([surf - 0.01, surf, surf + 0.01], 1).reshape(-1, 3)
        rad = torch.full((trip.shape[0],), r_add, device=dev)
        npc.append_points(trip, torch.randn(trip.shape[0], 32, generator=gi).to(dev), torch.randn(trip.shape[0], 32, generator=gi).to(dev),
                          radius=rad, build=True)
        rows = torch.randperm(n_base, generator=gi)[:a.rows].to(dev)
        sync.note_rows(npc, rows)
        npc.get_geo_feats()[rows] += 0.01
        npc.get_col_feats()[rows] += 0.01
        theta[:60000] += 0.001
        # the other ranks' blocks: the same locations seen from neighbouring frames (half of them within the add radius of
        # this rank's, half displaced beyond it), and their trained rows (half of them rows this rank also trained)
        pts_blocks, row_blocks = [], []
        for k in range(1, a.blocks):
            shift = torch.where(torch.arange(a.new_locations, device=dev) % 2 == 0, 0.3 * r_add, 2.5 * r_add * k)
            t2 = (trip.reshape(-1, 3, 3) + torch.stack([shift, torch.zeros_like(shift), torch.zeros_like(shift)], 1)[:, None, :]).reshape(-1, 3)
            rec = torch.empty(t2.shape[0], REC, device=dev)
            rec[:, :3] = t2; rec[:, 3:67] = 0.5; rec[:, 67] = r_add
            pts_blocks.append(rec)
            rk = torch.cat([rows[: a.rows // 2], torch.randperm(n_base, generator=gi)[: a.rows - a.rows // 2].to(dev)]).unique()
            rr = torch.empty(rk.shape[0], REC_ROW, device=dev)
            rr[:, 0] = rk.to(torch.int32).view(torch.float32); rr[:, 1:] = 0.02
            row_blocks.append(rr)
        torch.cuda.synchronize()
        ms0 = torch.cuda.memory_stats()
        # exchange = _reconcile_rows (first all-gather) + decoder all-reduce + merge_new_points (second all-gather)
        t0 = time.perf_counter()
        tr.pending = row_blocks
        sync._reconcile_rows(npc)
        torch.cuda.synchronize(); t1 = time.perf_counter()
        d = theta[:sync.n_color] - sync.snap_theta
        dist.all_reduce(d)
        theta[:sync.n_color] = sync.snap_theta + d / a.blocks
        torch.cuda.synchronize(); t2_ = time.perf_counter()
        from point_slam_amd.dist import merge_new_points
        tr.pending = pts_blocks
        counts = merge_new_points(npc, sync.n_base, None, True, tr)
        sync.snap_theta = theta[:sync.n_color].clone()
        sync.n_base = npc.pts_num()
        torch.cuda.synchronize(); t3 = time.perf_counter()
        times.append((t3 - t0) * 1e3)
        ms1 = torch.cuda.memory_stats()
        # what the caching allocator did during the call: new segments = hipMalloc calls, freed segments = hipFree (a
        # device-wide synchronisation)
        parts.append(dict(rows_ms=round((t1 - t0) * 1e3, 3), decoder_ms=round((t2_ - t1) * 1e3, 3), points_ms=round((t3 - t2_) * 1e3, 3),
                          contributed=sum(counts), admitted=npc.pts_num() - n_base, rows_received=sync.last_stats.get("rows_received"),
                          segments_allocated=ms1["segment.all.allocated"] - ms0["segment.all.allocated"],
                          segments_freed=ms1["segment.all.freed"] - ms0["segment.all.freed"],
                          alloc_retries=ms1["num_alloc_retries"] - ms0["num_alloc_retries"],
                          reserved_mb=round(ms1["reserved_bytes.all.current"] / 2**20)))
    steady = sorted(times[a.warmup:])
    res = json.dumps(dict(metric="frame-parallel exchange wall time (device work + host, no wire)", unit="ms", points=a.points,
                          blocks=a.blocks, new_locations_per_block=a.new_locations, trained_rows_per_block=a.rows,
                          median_ms=round(steady[len(steady) // 2], 3), min_ms=round(steady[0], 3), max_ms=round(steady[-1], 3),
                          p50_ms=round(steady[len(steady) // 2], 3), p90_ms=round(steady[int(0.9 * (len(steady) - 1))], 3), p100_ms=round(steady[-1], 3),
                          calls=len(steady), warmup_calls_ms=[round(t, 3) for t in times[:a.warmup]],
                          slow_calls=[dict(call=i, ms=round(times[i], 3), **parts[i]) for i in range(a.warmup, len(times)) if times[i] > 5.0],
                          allocator_conf=os.environ.get("PYTORCH_HIP_ALLOC_CONF") or os.environ.get("PYTORCH_CUDA_ALLOC_CONF"),
                          first_call_ms=round(times[0], 3), per_call=parts[a.warmup:],
                          backend=a.backend,
                          note="world of one process: the two all-gathers return this rank's records plus seven prepared blocks "
                               "(no wire time); the decoder all-reduce (240 KB) runs on the one-rank communicator"))
    if a.out:
        open(a.out, "w").write(res + "\n")
    print(res, flush=True)
    dist.destroy_process_group()


def summary(files):
    for f in files:
        try:
            d = json.loads(open(f).read())
            print(os.path.basename(f), "p50", d["p50_ms"], "p90", d["p90_ms"], "p100", d["p100_ms"], "allocator", d.get("allocator_conf"), "slow calls",
                  [(c["call"], c["ms"], "rows", c["rows_ms"], "segs +%d -%d" % (c["segments_allocated"], c["segments_freed"])) for c in d["slow_calls"]])
        except Exception as e:
            print(f, "failed", e)


if __name__ == "__main__":
    if len(sys.argv) > 1 and sys.argv[1] == "--summary":
        summary(sys.argv[2:])
    else:
        main()
