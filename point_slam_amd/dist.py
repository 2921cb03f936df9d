"""Frame-parallel multi-GPU mode: one process per GPU (torch.distributed; backend "nccl" = RCCL over xGMI on
ROCm, "gloo" in the CPU tests), a full map replica per rank, and ONE exchange step: a periodic all-gather of the
neural points each rank has added since the previous exchange.  The reference has no distributed
code at all (no NCCL / torch.distributed call anywhere in /root/reference); this mode is new functionality
specified by BASELINE.json's north_star.

Design for point-to-point xGMI: one fused, padded all-gather of [n_max, 67] fp32 records (position 3 + geometry
feature 32 + colour feature 32 = 268 B per point) instead of three collectives per tensor -- at <= 18 k new
locations x 3 points per rank (14.5 MB) the exchange is latency-, not bandwidth-bound on 7 x 153 GB/s links.
Every rank then rebuilds its cloud as  base points + rank-0 block + rank-1 block + ...  so that point indices
(and therefore feature rows) are identical on all ranks.  Features of PRE-EXISTING points are not exchanged
(owner-writes policy, v1).
"""
from __future__ import annotations

from typing import List, Tuple

import torch
import torch.distributed as dist

REC = 3 + 32 + 32


def exchange_new_points(pos: torch.Tensor, geo: torch.Tensor, col: torch.Tensor, group=None
                        ) -> Tuple[torch.Tensor, torch.Tensor, torch.Tensor, List[int]]:
    """All-gather-v of this rank's new points.  Returns (pos, geo, col) of ALL ranks concatenated in rank order
    and the per-rank counts.  Works on any backend/device torch.distributed supports."""
    world = dist.get_world_size(group)
    dev = pos.device
    n = torch.tensor([pos.shape[0]], device=dev, dtype=torch.int64)
    counts_t = [torch.zeros_like(n) for _ in range(world)]
    dist.all_gather(counts_t, n, group=group)
    counts = [int(c.item()) for c in counts_t]
    n_max = max(max(counts), 1)
    rec = torch.zeros(n_max, REC, device=dev, dtype=torch.float32)
    if pos.shape[0]:
        rec[:pos.shape[0], :3] = pos
        rec[:pos.shape[0], 3:35] = geo
        rec[:pos.shape[0], 35:] = col
    out = [torch.empty_like(rec) for _ in range(world)]
    dist.all_gather(out, rec, group=group)
    blocks = [o[:c] for o, c in zip(out, counts)]
    allrec = torch.cat(blocks, 0) if blocks else rec[:0]
    return allrec[:, :3].contiguous(), allrec[:, 3:35].contiguous(), allrec[:, 35:].contiguous(), counts


def merge_new_points(npc, n_base: int, group=None) -> List[int]:
    """Exchange the points npc gained since it had n_base points and rebuild the replica in global rank order."""
    n_now = npc.pts_num()
    pos = npc.cloud_pos()[n_base:n_now]
    geo = npc.get_geo_feats()[n_base:n_now]
    col = npc.get_col_feats()[n_base:n_now]
    p_all, g_all, c_all, counts = exchange_new_points(pos, geo, col, group)
    npc.truncate(n_base)
    npc.append_points(p_all, g_all, c_all)
    return counts


def frames_of_rank(n_frames: int, rank: int, world: int) -> List[int]:
    """Frame t is processed by rank t mod world (SURVEY.md §8e partitioning)."""
    return [t for t in range(n_frames) if t % world == rank]
