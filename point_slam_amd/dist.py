"""Frame-parallel multi-GPU mode: one process per GPU (torch.distributed; backend "nccl" = RCCL over xGMI on
ROCm, "gloo" in the CPU tests), a full map replica per rank, and ONE exchange step every few mapped frames
(SURVEY.md §8e).  The reference has no distributed code at all (no NCCL / torch.distributed call anywhere in
/root/reference); this mode is new functionality specified by BASELINE.json's north_star.

What one exchange does (`FrameParallelSync.exchange`), identically on every rank:

 1. NEW POINTS.  One fused, padded all-gather of [n_max, 68] fp32 records (position 3 + geometry feature 32 +
    colour feature 32 + the add-radius of the point's location = 272 B per point) instead of one collective per
    tensor: at <= 18 k new locations x 3 points per rank (14.7 MB) the exchange is latency-, not bandwidth-bound on
    7 x 153 GB/s point-to-point links.  Only the tail [n_base, N) leaves the device structures (O(new), not O(N)).
 2. CROSS-RANK DEDUPE.  Frames t and t+1 run on different ranks and see almost the same surfaces; each rank deduped
    only against its own replica.  The gathered blocks are therefore re-admitted in rank order: a location (its 3
    points) of block k is kept iff its surface point has no neighbour within its add-radius among
    base + kept blocks 0..k-1 -- the reference's own rule (src/neural_point.py:116-121) applied across ranks, so the
    min-distance invariant of the single-GPU map holds for the merged map.  Every rank runs the same deterministic
    procedure on the same data: replicas end up identical (same points, same order, same feature rows).
 3. FEATURES OF EXISTING POINTS.  Rows optimised by several ranks since the last exchange are reconciled by
    averaging their CHANGES: new = snapshot + sum_k (feats_k - snapshot) / #{k : row changed on k}
    (two all-reduces over the [N_base, 64] matrix: 256 MB at 1 M points, ~ms on xGMI, every ~50 frames).
 4. COLOUR DECODER.  The same rule on the trainable colour-decoder blob (all-reduce mean of the per-rank changes):
    features gathered from rank k were trained against rank k's decoder, so the decoders must not drift apart.

The collectives are issued through torch.distributed (which owns the RCCL communicator); the C ABI provides the
pack / unpack ends (psl_points_download_range, psl_points_append, psl_knn for the dedupe test).
"""
from __future__ import annotations

from typing import List, Optional, Tuple

import torch
import torch.distributed as dist

REC = 3 + 32 + 32 + 1


def exchange_new_points(pos: torch.Tensor, geo: torch.Tensor, col: torch.Tensor, radius: Optional[torch.Tensor] = None,
                        group=None) -> Tuple[torch.Tensor, torch.Tensor, torch.Tensor, torch.Tensor, List[int]]:
    """All-gather-v of this rank's new points.  Returns (pos, geo, col, radius) of ALL ranks concatenated in rank
    order and the per-rank counts.  Works on any backend/device torch.distributed supports."""
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
        rec[:pos.shape[0], 35:67] = col
        if radius is not None:
            rec[:pos.shape[0], 67] = radius
    out = [torch.empty_like(rec) for _ in range(world)]
    dist.all_gather(out, rec, group=group)
    blocks = [o[:c] for o, c in zip(out, counts)]
    allrec = torch.cat(blocks, 0) if blocks else rec[:0]
    return (allrec[:, :3].contiguous(), allrec[:, 3:35].contiguous(), allrec[:, 35:67].contiguous(),
            allrec[:, 67].contiguous(), counts)


def merge_new_points(npc, n_base: int, group=None, dedupe: bool = True) -> List[int]:
    """Exchange the points `npc` gained since it had n_base points and rebuild the replica in global rank order
    (steps 1-2 of the module docstring).  Returns the number of points every rank CONTRIBUTED; the number admitted
    after the cross-rank dedupe is npc.pts_num() - n_base."""
    n_now = npc.pts_num()
    pos = npc.cloud_pos_device(n_base, n_now - n_base)          # tail only
    geo = npc.get_geo_feats()[n_base:n_now]
    col = npc.get_col_feats()[n_base:n_now]
    rad = npc.point_radius(n_base, n_now - n_base)
    p_all, g_all, c_all, r_all, counts = exchange_new_points(pos, geo, col, rad, group)
    npc.truncate(n_base)
    if n_now != n_base and hasattr(npc, "_build"):
        # truncating invalidates the index; a rank whose own block is empty (or comes later) would otherwise run the
        # dedupe test of the first foreign block against a stale index while its peers carry on -> desynchronised ranks
        npc._build()
    off = 0
    for k, c in enumerate(counts):
        pk, gk, ck, rk = p_all[off:off + c], g_all[off:off + c], c_all[off:off + c], r_all[off:off + c]
        off += c
        if c == 0:
            continue
        if dedupe and k > 0 and npc.pts_num() > 0:
            # locations are triplets (N_add = 3, neural_point.py:126-143); the middle point is the surface point
            loc = pk.reshape(-1, 3, 3)[:, 1, :].contiguous()
            keep = npc.locations_free(loc, rk.reshape(-1, 3)[:, 1].contiguous())
            keep3 = keep[:, None].expand(-1, 3).reshape(-1)
            pk, gk, ck, rk = pk[keep3], gk[keep3], ck[keep3], rk[keep3]
        npc.append_points(pk, gk, ck, radius=rk, build=True)    # the next block is tested against this one too
    return counts


class FrameParallelSync:
    """State of the periodic reconciliation (steps 3-4): a snapshot of the feature rows and of the colour-decoder
    blob as of the last exchange."""

    def __init__(self, npc, theta: Optional[torch.Tensor] = None, n_color: int = 0, group=None):
        self.group = group
        self.n_color = n_color
        self.snap_geo = npc.get_geo_feats().clone()
        self.snap_col = npc.get_col_feats().clone()
        self.snap_theta = theta[:n_color].clone() if theta is not None else None
        self.n_base = npc.pts_num()

    @staticmethod
    def _avg_changes(cur: torch.Tensor, snap: torch.Tensor, group) -> torch.Tensor:
        delta = cur - snap
        changed = (delta != 0).any(dim=-1, keepdim=True).to(torch.float32) if delta.dim() > 1 else \
            (delta != 0).to(torch.float32)
        dist.all_reduce(delta, op=dist.ReduceOp.SUM, group=group)
        dist.all_reduce(changed, op=dist.ReduceOp.SUM, group=group)
        return snap + delta / changed.clamp_min(1.0)

    def exchange(self, npc, theta: Optional[torch.Tensor] = None, dedupe: bool = True) -> List[int]:
        """One exchange (all four steps).  `theta`: the master parameter blob, updated in place (colour group)."""
        nb = self.n_base
        # 3. features of the points that existed at the last exchange
        if nb:
            geo = self._avg_changes(npc.get_geo_feats()[:nb], self.snap_geo[:nb], self.group)
            col = self._avg_changes(npc.get_col_feats()[:nb], self.snap_col[:nb], self.group)
            npc.get_geo_feats()[:nb] = geo
            npc.get_col_feats()[:nb] = col
        # 4. colour decoder
        if theta is not None and self.snap_theta is not None:
            d = theta[:self.n_color] - self.snap_theta
            dist.all_reduce(d, op=dist.ReduceOp.SUM, group=self.group)
            theta[:self.n_color] = self.snap_theta + d / dist.get_world_size(self.group)
        # 1-2. new points
        counts = merge_new_points(npc, nb, self.group, dedupe)
        self.snap_geo = npc.get_geo_feats().clone()
        self.snap_col = npc.get_col_feats().clone()
        if theta is not None:
            self.snap_theta = theta[:self.n_color].clone()
        self.n_base = npc.pts_num()
        return counts


def frames_of_rank(n_frames: int, rank: int, world: int) -> List[int]:
    """Frame t is processed by rank t mod world (SURVEY.md §8e partitioning)."""
    return [t for t in range(n_frames) if t % world == rank]
