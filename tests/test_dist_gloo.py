"""CPU, world_size=2, gloo: the multi-GPU exchange step (all-gather-v of newly added neural points and the
rank-ordered rebuild) -- the N>1 path that the driver runs on RCCL at round end."""
import os
import socket

import torch
import torch.distributed as dist
import torch.multiprocessing as mp


class FakeCloud:
    """Host stand-in with the HipNeuralPointCloud methods merge_new_points touches."""

    def __init__(self, pos, geo, col):
        self.pos, self.geo, self.col = pos, geo, col

    def pts_num(self): return self.pos.shape[0]
    def cloud_pos(self): return self.pos
    def get_geo_feats(self): return self.geo
    def get_col_feats(self): return self.col

    def truncate(self, n):
        self.pos, self.geo, self.col = self.pos[:n], self.geo[:n], self.col[:n]

    def append_points(self, p, g, c):
        self.pos, self.geo, self.col = torch.cat([self.pos, p]), torch.cat([self.geo, g]), torch.cat([self.col, c])


def _worker(rank, world, port, q):
    os.environ.update(MASTER_ADDR="127.0.0.1", MASTER_PORT=str(port))
    dist.init_process_group("gloo", rank=rank, world_size=world)
    from point_slam_amd.dist import exchange_new_points, frames_of_rank, merge_new_points
    g = torch.Generator().manual_seed(100)
    base = torch.rand(10, 3, generator=g)
    bg, bc = torch.rand(10, 32, generator=g), torch.rand(10, 32, generator=g)
    gr = torch.Generator().manual_seed(rank + 1)
    n_new = 3 * (rank + 1) if rank < 2 else 0                    # ragged: 3, 6 ; third exchange empty on rank 1
    cloud = FakeCloud(torch.cat([base, torch.rand(n_new, 3, generator=gr)]),
                      torch.cat([bg, torch.rand(n_new, 32, generator=gr)]),
                      torch.cat([bc, torch.rand(n_new, 32, generator=gr)]))
    own = cloud.pos[10:].clone()
    counts = merge_new_points(cloud, 10)
    # second exchange where one rank has nothing to contribute
    n0 = cloud.pts_num()
    if rank == 0:
        cloud.append_points(torch.ones(2, 3), torch.ones(2, 32), torch.ones(2, 32))
    counts2 = merge_new_points(cloud, n0)
    q.put((rank, counts, counts2, cloud.pos.clone(), cloud.geo.clone(), own, frames_of_rank(7, rank, world)))
    dist.barrier()
    dist.destroy_process_group()


def test_exchange_new_points_world2():
    s = socket.socket(); s.bind(("127.0.0.1", 0)); port = s.getsockname()[1]; s.close()
    ctx = mp.get_context("spawn")
    q = ctx.Queue()
    procs = [ctx.Process(target=_worker, args=(r, 2, port, q)) for r in range(2)]
    for p in procs:
        p.start()
    res = sorted([q.get(timeout=120) for _ in range(2)], key=lambda x: x[0])
    for p in procs:
        p.join(timeout=60)
        assert p.exitcode == 0
    (r0, c0, c0b, pos0, geo0, own0, fr0), (r1, c1, c1b, pos1, geo1, own1, fr1) = res
    assert c0 == c1 == [3, 6] and c0b == c1b == [2, 0]
    assert torch.equal(pos0, pos1) and torch.equal(geo0, geo1)          # identical replicas, identical order
    assert pos0.shape[0] == 10 + 9 + 2
    assert torch.equal(pos0[10:13], own0) and torch.equal(pos0[13:19], own1)   # rank order
    assert fr0 == [0, 2, 4, 6] and fr1 == [1, 3, 5]
