"""CPU, world_size=2, gloo: the multi-GPU exchange step -- all-gather-v of newly added neural points, the cross-rank
dedupe in rank order, the rank-ordered rebuild, and the reconciliation of features / colour decoder that several
ranks optimised -- the N>1 path that the driver runs on RCCL at round end.  (The same code on the real
HipNeuralPointCloud with two ranks sharing cuda:0: tests/test_hip_dist.py, -m gpu.)"""
import os
import socket

import torch
import torch.distributed as dist
import torch.multiprocessing as mp


class FakeCloud:
    """Host stand-in with the HipNeuralPointCloud methods the exchange touches (brute-force neighbour test)."""

    def __init__(self, pos, geo, col, rad=None):
        self.pos, self.geo, self.col = pos, geo, col
        self.rad = rad if rad is not None else torch.full((pos.shape[0],), 0.04)

    def pts_num(self): return self.pos.shape[0]
    def cloud_pos_device(self, first=0, count=None): return self.pos[first:(None if count is None else first + count)]
    def get_geo_feats(self): return self.geo
    def get_col_feats(self): return self.col
    def point_radius(self, first=0, count=None): return self.rad[first:(None if count is None else first + count)]

    index_ok = True          # mirrors the native index: stale after truncate / append until the next build

    def count_within(self, loc, radius, idx_limit=None):
        assert self.index_ok, "neighbour test on a stale index"
        pos = self.pos if idx_limit is None else self.pos[:idx_limit]
        d2 = ((loc[:, None, :] - pos[None]) ** 2).sum(-1)
        return (d2 < (radius * radius)[:, None]).sum(1)

    def dedupe_blocks(self, rec, pos_col, rad_col, block_first, keep):
        """psl_dedupe_blocks restated with torch on the host: block b against the kept locations of the blocks before it."""
        pts = rec[:, pos_col:pos_col + 3].reshape(-1, 3, 3)
        rad = rec[:, rad_col].reshape(-1, 3)[:, 1]
        for b in range(1, len(block_first) - 1):
            a, e = block_first[b], block_first[b + 1]
            prev = pts[:a][keep[:a].bool()].reshape(-1, 3)
            if e == a or prev.shape[0] == 0:
                continue
            d = pts[a:e, 1, None, :] - prev[None]
            d2 = (d[..., 0] * d[..., 0] + d[..., 1] * d[..., 1]) + d[..., 2] * d[..., 2]
            keep[a:e] &= (~(d2 < (rad[a:e] * rad[a:e])[:, None]).any(1)).to(torch.uint8)
        return keep

    def truncate(self, n):
        if n != self.pos.shape[0]:
            self.index_ok = False
        self.pos, self.geo, self.col, self.rad = self.pos[:n], self.geo[:n], self.col[:n], self.rad[:n]

    def append_points(self, p, g, c, build=True, radius=None):
        r = radius if radius is not None else torch.full((p.shape[0],), 0.04)
        self.pos, self.geo, self.col, self.rad = (torch.cat([self.pos, p]), torch.cat([self.geo, g]),
                                                  torch.cat([self.col, c]), torch.cat([self.rad, r]))
        self.index_ok = bool(build)
        self.builds = getattr(self, "builds", 0) + (1 if build else 0)


def _triplets(centres):
    """3 points per location at 0.98 / 1.00 / 1.02 of the way from the origin (N_add = 3)."""
    return (centres[:, None, :] * torch.tensor([0.98, 1.0, 1.02])[None, :, None]).reshape(-1, 3)


def _np(x):
    """tensors travel through the queue BY VALUE (numpy): torch's fd-sharing needs the sender alive at receive time"""
    if torch.is_tensor(x):
        return x.detach().cpu().numpy()
    return x


def _t(x):
    import numpy as np
    return torch.from_numpy(x) if isinstance(x, np.ndarray) else x


def _worker(rank, world, port, q):
    os.environ.update(MASTER_ADDR="127.0.0.1", MASTER_PORT=str(port))
    dist.init_process_group("gloo", rank=rank, world_size=world)
    from point_slam_amd.dist import FrameParallelSync, frames_of_rank, merge_new_points
    g = torch.Generator().manual_seed(100)
    base = torch.rand(10, 3, generator=g) + 5.0
    bg, bc = torch.rand(10, 32, generator=g), torch.rand(10, 32, generator=g)
    gr = torch.Generator().manual_seed(rank + 1)
    # ragged blocks: 1 location on rank 0, 2 on rank 1; rank 1's FIRST location sits 1 cm from rank 0's -> duplicate
    c0 = torch.tensor([[1.0, 1.0, 1.0]])
    centres = c0 if rank == 0 else torch.cat([c0 + 0.01, torch.tensor([[2.0, 2.0, 2.0]])])
    new = _triplets(centres)
    n_new = new.shape[0]
    cloud = FakeCloud(torch.cat([base, new]), torch.cat([bg, torch.rand(n_new, 32, generator=gr)]),
                      torch.cat([bc, torch.rand(n_new, 32, generator=gr)]))
    own = cloud.pos[10:].clone()
    counts = merge_new_points(cloud, 10)
    n_after_dedupe = cloud.pts_num()
    # second exchange where one rank has nothing to contribute, no dedupe
    n0 = cloud.pts_num()
    if rank == 0:
        cloud.append_points(torch.ones(3, 3) * 9, torch.ones(3, 32), torch.ones(3, 32))
    counts2 = merge_new_points(cloud, n0, dedupe=False)
    # third exchange: rank 0 contributes NOTHING, rank 1 a location -- the first non-empty block is not block 0 (the
    # stale-index case of round 2's advisor finding); and the dedupe test runs on a valid index on both ranks
    n1 = cloud.pts_num()
    builds0 = cloud.builds
    if rank == 1:
        cloud.append_points(_triplets(torch.tensor([[3.0, 3.0, 3.0]])), torch.ones(3, 32), torch.ones(3, 32))
        builds0 = cloud.builds
    counts2b = merge_new_points(cloud, n1)
    assert counts2b == [0, 3] and cloud.pts_num() == n1 + 3 and cloud.index_ok
    assert cloud.builds == builds0 + 1          # ONE rebuild per exchange, however many blocks

    # features / decoder reconciliation: rank 0 changes rows 0,1; rank 1 changes rows 1,2; both change theta
    theta = torch.arange(8, dtype=torch.float32)
    sync = FrameParallelSync(cloud, theta, n_color=6)
    geo_before = cloud.geo.clone()
    # the rows a mapped frame is about to train are announced first (HipSLAM.map -> note_rows); row 5 is announced but
    # never changes, row 2 is announced twice on rank 1
    sync.note_rows(cloud, torch.tensor([0, 1, 5]) if rank == 0 else torch.tensor([1, 2]))
    if rank == 1:
        sync.note_rows(cloud, torch.tensor([2, 6]))
    if rank == 0:
        cloud.geo[0] += 1.0; cloud.geo[1] += 2.0; theta[:6] += 1.0; theta[6:] += 100.0
    else:
        cloud.geo[1] += 4.0; cloud.geo[2] += 8.0; theta[:6] += 3.0
    counts3 = sync.exchange(cloud, theta)
    q.put(tuple(_np(x) for x in (rank, counts, counts2, counts3, n_after_dedupe, cloud.pos.clone(), cloud.geo.clone(), own, geo_before,
           theta.clone(), frames_of_rank(7, rank, world))))
    dist.barrier()
    dist.destroy_process_group()


def test_exchange_new_points_world2():
    s = socket.socket(); s.bind(("127.0.0.1", 0)); port = s.getsockname()[1]; s.close()
    ctx = mp.get_context("spawn")
    q = ctx.Queue()
    procs = [ctx.Process(target=_worker, args=(r, 2, port, q)) for r in range(2)]
    for p in procs:
        p.start()
    res = sorted([tuple(_t(x) for x in q.get(timeout=120)) for _ in range(2)], key=lambda x: x[0])
    for p in procs:
        p.join(timeout=60)
        assert p.exitcode == 0
    (r0, c0, c0b, c0c, n0, pos0, geo0, own0, gb0, th0, fr0), (r1, c1, c1b, c1c, n1, pos1, geo1, own1, gb1, th1, fr1) = res
    assert c0 == c1 == [3, 6] and c0b == c1b == [3, 0] and c0c == c1c == [0, 0]
    # rank 1's duplicate location was dropped on BOTH ranks: 10 base + 3 (rank 0) + 3 (rank 1's second location)
    assert n0 == n1 == 16
    assert torch.equal(pos0, pos1) and torch.equal(geo0, geo1)          # identical replicas, identical order
    assert pos0.shape[0] == 16 + 3 + 3
    assert torch.equal(pos0[10:13], own0) and torch.equal(pos0[13:16], own1[3:6])   # rank order, kept block
    # min-distance invariant across ranks: no surface point of a later block inside 4 cm of an earlier block's points
    assert float(((pos0[14] - pos0[10:13]) ** 2).sum(-1).min()) > 0.04 ** 2
    # averaged CHANGES: row 0 only rank 0 (+1), row 1 both ((2+4)/2), row 2 only rank 1 (+8), others untouched
    d = geo0[:3, 0] - gb0[:3, 0]
    assert torch.allclose(d, torch.tensor([1.0, 3.0, 8.0]))
    assert torch.equal(geo0[3:], gb0[3:])
    # colour-decoder group: mean of the changes (+1, +3 -> +2); the geometry group is not exchanged
    assert torch.allclose(th0[:6], torch.arange(6, dtype=torch.float32) + 2.0) and torch.equal(th0[:6], th1[:6])
    assert th0[6] == 106.0 and th1[6] == 6.0
    assert fr0 == [0, 2, 4, 6] and fr1 == [1, 3, 5]


# ---------------------------------------------------------------------------------------------------------------------
# world 4: ragged blocks with TWO empty ranks and a rank whose WHOLE block is deduped away by an
# earlier rank's block; then an exchange in which nobody contributes; features touched on a subset of ranks.
def _worker4(rank, world, port, q):
    os.environ.update(MASTER_ADDR="127.0.0.1", MASTER_PORT=str(port))
    dist.init_process_group("gloo", rank=rank, world_size=world)
    from point_slam_amd.dist import FrameParallelSync
    g = torch.Generator().manual_seed(100)
    base = torch.rand(12, 3, generator=g) + 5.0
    bg, bc = torch.rand(12, 32, generator=g), torch.rand(12, 32, generator=g)
    cloud = FakeCloud(base.clone(), bg.clone(), bc.clone())
    theta = torch.arange(8, dtype=torch.float32)
    sync = FrameParallelSync(cloud, theta, n_color=6)
    # rank 0: nothing.  rank 1: locations A, B.  rank 2: nothing.  rank 3: A + 5 mm and B - 5 mm -> both inside rank 1's
    # add-radius (4 cm): its whole block goes.
    A, B = torch.tensor([[1.0, 1.0, 1.0]]), torch.tensor([[2.0, 1.0, 0.5]])
    gr = torch.Generator().manual_seed(rank + 1)
    if rank == 1:
        new = _triplets(torch.cat([A, B]))
    elif rank == 3:
        new = _triplets(torch.cat([A + 0.005, B - 0.005]))
    else:
        new = torch.zeros(0, 3)
    if new.shape[0]:
        cloud.append_points(new, torch.rand(new.shape[0], 32, generator=gr), torch.rand(new.shape[0], 32, generator=gr))
    builds0 = getattr(cloud, "builds", 0)
    # rows: rank 0 trains rows 0,1; rank 2 trains row 1; ranks 1,3 announce row 3 but leave it unchanged
    sync.note_rows(cloud, torch.tensor([0, 1]) if rank == 0 else (torch.tensor([1]) if rank == 2 else torch.tensor([3])))
    if rank == 0:
        cloud.geo[0] += 1.0; cloud.geo[1] += 2.0
    if rank == 2:
        cloud.geo[1] += 6.0
    counts = sync.exchange(cloud, theta)
    stats = dict(sync.last_stats)
    n_after = cloud.pts_num()
    builds = cloud.builds - builds0
    # second exchange: nobody has anything (no rows, no points): the replica must stand as it is, no rebuild needed
    pos_before = cloud.pos.clone()
    counts2 = sync.exchange(cloud, theta)
    same = bool(torch.equal(pos_before, cloud.pos)) and cloud.index_ok
    q.put(tuple(_np(x) for x in (rank, counts, counts2, n_after, builds, cloud.pos.clone(), cloud.geo.clone(), bg, stats["rows_sent"],
                                 stats["rows_received"], same)))
    dist.barrier()
    dist.destroy_process_group()


def test_exchange_world4_empty_ranks_and_a_block_deduped_away():
    s = socket.socket(); s.bind(("127.0.0.1", 0)); port = s.getsockname()[1]; s.close()
    ctx = mp.get_context("spawn")
    q = ctx.Queue()
    procs = [ctx.Process(target=_worker4, args=(r, 4, port, q)) for r in range(4)]
    for p in procs:
        p.start()
    res = sorted([tuple(_t(x) for x in q.get(timeout=180)) for _ in range(4)], key=lambda x: x[0])
    for p in procs:
        p.join(timeout=60)
        assert p.exitcode == 0
    for r in res:
        rank, counts, counts2, n_after, builds, pos, geo, bg, sent, recv, same = r
        assert counts == [0, 6, 0, 6] and counts2 == [0, 0, 0, 0]
        assert n_after == 12 + 6                      # rank 1's two locations; rank 3's block is gone entirely
        assert builds == 1 and same                   # one rebuild per exchange; an empty exchange changes nothing
        assert torch.equal(pos, res[0][5]) and torch.equal(geo, res[0][6])          # identical replicas
        assert recv == 3                              # rows 0, 1 from rank 0 and row 1 from rank 2; the untouched row 3 is not sent
        assert sent == {0: 2, 1: 0, 2: 1, 3: 0}[rank]
    geo, bg = res[0][6], res[0][7]
    d = geo[:4, 0] - bg[:4, 0]
    assert torch.allclose(d, torch.tensor([1.0, 4.0, 0.0, 0.0]))                    # row 1: mean of +2 and +6


def _worker_owner(rank, world, port, q):
    os.environ.update(MASTER_ADDR="127.0.0.1", MASTER_PORT=str(port))
    dist.init_process_group("gloo", rank=rank, world_size=world)
    from point_slam_amd.dist import FrameParallelSync
    g = torch.Generator().manual_seed(5)
    base = torch.rand(6, 3, generator=g) + 5.0
    cloud = FakeCloud(base.clone(), torch.zeros(6, 32), torch.zeros(6, 32))
    sync = FrameParallelSync(cloud, None, merge="owner")
    # interval 1: every rank creates one location far from the others' (three points each) -> rows 6..8 belong to rank 0,
    # 9..11 to rank 1, 12..14 to rank 2 on every replica
    cloud.append_points(_triplets(torch.tensor([[1.0 + rank, 1.0, 1.0]])), torch.zeros(3, 32), torch.zeros(3, 32))
    sync.exchange(cloud)
    owner = sync._owner[:cloud.pts_num()].clone()
    # interval 2: row 10 (created by rank 1) is changed by ranks 0, 1, 2 -> rank 1's change wins;
    #             row 13 (created by rank 2) by ranks 0 and 1 only -> the lowest contributor (rank 0) wins;
    #             row 2 (seed map) by ranks 1 and 2 -> rank 1 wins; row 7 by its creator alone
    rows = {0: [10, 13, 7], 1: [10, 13, 2], 2: [10, 2]}[rank]
    sync.note_rows(cloud, torch.tensor(rows))
    for r in rows:
        cloud.geo[r] += float(10 ** rank) * (1 + r)         # 1x, 10x, 100x (1 + row)
        cloud.col[r] -= float(10 ** rank)
    sync.exchange(cloud)
    q.put((rank, _np(owner), _np(cloud.geo[:, 0].clone()), _np(cloud.col[:, 0].clone())))
    dist.barrier()
    dist.destroy_process_group()


def test_owner_writes_merge_rule_world3():
    """merge='owner' (SURVEY.md 8e): the creating rank's change wins when it is among the contributors, otherwise the lowest
    contributing rank's; creators are learnt at the exchange that admits the points; replicas identical."""
    s = socket.socket(); s.bind(("127.0.0.1", 0)); port = s.getsockname()[1]; s.close()
    ctx = mp.get_context("spawn")
    q = ctx.Queue()
    procs = [ctx.Process(target=_worker_owner, args=(r, 3, port, q)) for r in range(3)]
    for p in procs:
        p.start()
    res = sorted([tuple(_t(x) for x in q.get(timeout=300)) for _ in range(3)], key=lambda x: x[0])
    for p in procs:
        p.join(timeout=60)
        assert p.exitcode == 0
    for r in res[1:]:
        assert torch.equal(r[1], res[0][1]) and torch.equal(r[2], res[0][2]) and torch.equal(r[3], res[0][3])
    owner, geo, col = res[0][1], res[0][2], res[0][3]
    assert owner.tolist() == [-1] * 6 + [0] * 3 + [1] * 3 + [2] * 3
    assert geo[10] == 10.0 * 11 and col[10] == -10.0          # creator (rank 1) among three contributors
    assert geo[13] == 1.0 * 14 and col[13] == -1.0            # creator absent: lowest contributor (rank 0)
    assert geo[2] == 10.0 * 3 and col[2] == -10.0             # seed-map row: lowest contributor (rank 1)
    assert geo[7] == 1.0 * 8                                  # its creator alone
    assert float(geo[[0, 1, 3, 4, 5, 6, 8, 9, 11, 12, 14]].abs().sum()) == 0.0


# ---------------------------------------------------------------------------------------------------------------------
# world 8: the rank count of BASELINE config 4 at realistic RATIOS -- every rank trains a few thousand rows that
# overlap its neighbours', adds tens of locations of which some collide across ranks, and the colour decoder moves on all of them.  The
# replicas must come out identical, and the HOST time of each phase of the exchange is recorded per rank (FrameParallelSync.last_host_ms):
# the record the 8-GPU run will be read against (on this box the eight ranks share eight cores and gloo's TCP loopback).
def _worker8(rank, world, port, q):
    os.environ.update(MASTER_ADDR="127.0.0.1", MASTER_PORT=str(port))
    torch.set_num_threads(1)
    dist.init_process_group("gloo", rank=rank, world_size=world)
    from point_slam_amd.dist import FrameParallelSync
    g = torch.Generator().manual_seed(7)
    n0 = 6000
    base = torch.rand(n0, 3, generator=g) * 4.0 + 5.0
    bg, bc = torch.rand(n0, 32, generator=g), torch.rand(n0, 32, generator=g)
    cloud = FakeCloud(base.clone(), bg.clone(), bc.clone())
    theta = torch.arange(64, dtype=torch.float32)
    sync = FrameParallelSync(cloud, theta, n_color=48)
    host = []
    for interval in range(2):
        gr = torch.Generator().manual_seed(1000 * interval + rank)
        # rows: a window of 1 500 rows per rank, half of it shared with the next rank; every other announced row really changes
        first = (rank * 750 + 300 * interval) % (n0 - 1500)
        rows = torch.arange(first, first + 1500)
        sync.note_rows(cloud, rows)
        ch = rows[::2]
        cloud.geo[ch] += torch.rand(ch.shape[0], 32, generator=gr)
        cloud.col[ch] -= torch.rand(ch.shape[0], 32, generator=gr)
        theta[:48] += float(rank + 1)
        # new locations: 24 per rank on a lattice far from the seed map; ranks r and r + 4 propose the SAME lattice cell shifted by 5 mm
        cell = torch.arange(24, dtype=torch.float32)
        centres = torch.stack([1.0 + 0.2 * cell, torch.full((24,), 1.0 + 0.3 * (rank % 4)), torch.full((24,), 1.0 + 0.5 * interval)], 1)
        if rank >= 4:
            centres = centres + 0.005
        new = _triplets(centres)
        cloud.append_points(new, torch.rand(new.shape[0], 32, generator=gr), torch.rand(new.shape[0], 32, generator=gr))
        counts = sync.exchange(cloud, theta)
        host.append(dict(sync.last_host_ms, **{k: int(v) for k, v in sync.last_stats.items()}))
    q.put(tuple(_np(x) for x in (rank, counts, cloud.pts_num(), cloud.pos.clone(), cloud.geo.clone(), cloud.col.clone(), theta.clone(), host)))
    dist.barrier()
    dist.destroy_process_group()


def test_exchange_world8_replicas_identical_and_host_time_recorded():
    import json
    s = socket.socket(); s.bind(("127.0.0.1", 0)); port = s.getsockname()[1]; s.close()
    ctx = mp.get_context("spawn")
    q = ctx.Queue()
    procs = [ctx.Process(target=_worker8, args=(r, 8, port, q)) for r in range(8)]
    for p in procs:
        p.start()
    res = sorted([tuple(_t(x) for x in q.get(timeout=300)) for _ in range(8)], key=lambda x: x[0])
    for p in procs:
        p.join(timeout=60)
        assert p.exitcode == 0
    r0 = res[0]
    for r in res:
        rank, counts, n_after, pos, geo, col, theta, host = r
        # (counts = what every rank proposed) ranks 4..7 lose every location to ranks 0..3: same cells, 5 mm apart, inside the 4-cm add radius
        assert counts == [72] * 8
        assert n_after == 6000 + 2 * 4 * 72
        assert torch.equal(pos, r0[3]) and torch.equal(geo, r0[4]) and torch.equal(col, r0[5]) and torch.equal(theta, r0[6])
        for h in host:
            assert h["rows_noted"] == 1500 and h["rows_sent"] == 750 and h["rows_received"] == 8 * 750
            assert set(("rows", "decoder", "new_points")) <= set(h) and all(h[k] >= 0.0 for k in ("rows", "decoder", "new_points"))
    # colour decoder: mean of the changes, (1 + ... + 8) / 8 = 4.5 per interval
    assert torch.allclose(r0[6][:48], torch.arange(48, dtype=torch.float32) + 9.0) and torch.equal(r0[6][48:], torch.arange(48, 64, dtype=torch.float32))
    rep = dict(test="exchange_world8_host_ms", world=8, backend="gloo (CPU tensors, eight ranks on this box's cores)",
               per_rank=[dict(rank=r[0], intervals=r[7]) for r in res],
               worst_total_ms=max(sum(h[k] for k in ("rows", "decoder", "new_points")) for r in res for h in r[7]))
    out = os.path.join(os.path.dirname(os.path.dirname(os.path.abspath(__file__))), "gpurun_out")
    try:
        os.makedirs(out, exist_ok=True)
        with open(os.path.join(out, "exchange_world8_host_ms.json"), "w") as f:
            json.dump(rep, f, indent=1)
    except OSError:
        pass
    print("REPORT", json.dumps(rep)[:400])
