"""Frame-parallel multi-GPU mode: one process per GPU (torch.distributed; backend "nccl" = RCCL over xGMI on
ROCm, "gloo" in the CPU tests), a full map replica per rank, and ONE exchange step every few mapped frames
(SURVEY.md §8e).  The reference has no distributed code at all (no NCCL / torch.distributed call anywhere in
/root/reference); this mode is new functionality specified by BASELINE.json's north_star.

What one exchange does (`FrameParallelSync.exchange`), identically on every rank -- all of it O(new + touched), nothing
scales with the size of the map:

 1. NEW POINTS.  One fused all-gather-v of [n, 68] fp32 records (position 3 + geometry feature 32 + colour feature 32 +
    the add-radius of the point's location = 272 B per point) instead of one collective per tensor: at <= 18 k new
    locations x 3 points per rank (14.7 MB) the exchange is latency-, not bandwidth-bound on 7 x 153 GB/s point-to-point
    links.  Only the tail [n_base, N) leaves the device structures.
 2. CROSS-RANK DEDUPE.  Frames t and t+1 run on different ranks and see almost the same surfaces; each rank deduped
    only against its own replica.  The gathered blocks are therefore re-admitted in rank order: a location (its 3
    points) of block k is kept iff its surface point has no neighbour strictly inside its add-radius among
    base + kept blocks 0..k-1 -- the reference's own rule (src/neural_point.py:116-121) applied across ranks, so the
    min-distance invariant of the single-GPU map holds for the merged map.  The test against the BASE map is ONE
    `psl_dedupe_count` launch over all foreign locations on the index as it stands (restricted to point indices
    < n_base: no rebuild before the test); the test against the few thousand points of the earlier kept blocks is a
    brute-force distance matrix; the grid index is rebuilt ONCE, after every block has been admitted.  Every rank runs
    the same deterministic procedure on the same data: replicas end up identical (same points, same order, same rows).
 3. FEATURES OF EXISTING POINTS.  A mapped frame trains only its frustum-selected rows (Mapper.py:346-360): the rows it
    is ABOUT to train are snapshotted first (`note_rows`, compact: row ids + values, each row once per exchange
    interval).  At the exchange a rank sends [n_changed, 1 + 64] records (row id, change of both feature rows) of the
    rows whose values really moved; every rank then applies  new = snapshot + sum_k change_k / #{k : row changed on k}
    in rank order (no atomics, no rank-dependent summation order: bit-identical replicas).
 4. COLOUR DECODER.  The same rule on the trainable colour-decoder blob (all-reduce of the per-rank changes, 109 k
    floats): features gathered from rank k were trained against rank k's decoder, so the decoders must not drift apart.

Transport: the all-gather-v's go through torch.distributed collectives by default -- on the "nccl" backend that IS RCCL
over xGMI (what a multi-GPU node runs), on gloo the CPU tests / ranks sharing one GPU.  The same exchange inside
libpointslam_hip.so (`psl_allgather_new_points` on the library's own RCCL communicator, `psl_comm_init`; the 128-byte
ncclUniqueId travels over torch.distributed once) is OPT-IN: `transport="native"` / PSL_NATIVE_RCCL=1.  It has only ever run
on a one-rank communicator (no multi-GPU node has been available to any session), so it is not the path a first 8-GPU run
takes by default (advisor, round 4); tests/test_hip_dist.py::test_exchange_two_gpus_nccl runs the default transport with two
ranks wherever two GPUs are visible (the native one under PSL_TEST_NATIVE_RCCL=1).  Before any rank enters ncclCommInitRank the ranks agree (all-reduce over
torch.distributed) that every one of them could load librccl and reserve its device buffers: a rank-local failure can no
longer leave the others blocked inside the communicator's rendezvous.

Merge rule for the features of points several ranks trained (`merge=`): "mean" (default) = snapshot + mean of the per-rank
changes; "owner" = SURVEY.md 8e's owner-writes policy: the rank that CREATED the point wins if it changed the row (points of
the seed map and rows their creator did not touch: the lowest contributing rank).  Both are deterministic and
rank-invariant; bench.py --merge reports the render loss after the exchange under either.
"""
from __future__ import annotations

import ctypes as C
import os
from typing import List, Optional, Tuple

import time

import torch
import torch.distributed as dist

REC = 3 + 32 + 32 + 1          # new-point record: xyz, geometry feature, colour feature, add-radius
REC_ROW = 1 + 64               # touched-row record: row id (int32 bits), change of the geometry + colour feature row


# ------------------------------------------------------------------------------------------------------- transport
class _TorchTransport:
    """all-gather-v through torch.distributed: counts, then one padded all-gather."""

    def __init__(self, group=None):
        self.group = group
        self.world = dist.get_world_size(group)

    def allgather_v(self, rec: torch.Tensor) -> Tuple[torch.Tensor, List[int]]:
        dev, width = rec.device, rec.shape[1]
        n = torch.tensor([rec.shape[0]], device=dev, dtype=torch.int64)
        counts_t = [torch.zeros_like(n) for _ in range(self.world)]
        dist.all_gather(counts_t, n, group=self.group)
        counts = [int(c.item()) for c in counts_t]
        n_max = max(counts)
        if n_max == 0:
            return rec[:0], counts
        send = torch.zeros(n_max, width, device=dev, dtype=torch.float32)
        send[:rec.shape[0]] = rec
        out = [torch.empty_like(send) for _ in range(self.world)]
        dist.all_gather(out, send, group=self.group)
        return torch.cat([o[:c] for o, c in zip(out, counts)], 0), counts


class _NativeTransport:
    """all-gather-v inside libpointslam_hip.so (psl_allgather_new_points on the library's RCCL communicator)."""

    def __init__(self, npc, group=None):
        """Collective.  Raises RuntimeError ON EVERY RANK when the communicator cannot be brought up on any of them (the
        ranks agree through torch.distributed before anybody returns), so that a caller may fall back as one."""
        from . import _lib
        self._lib, self.npc, self.group = _lib, npc, group
        self.world, rank = dist.get_world_size(group), dist.get_rank(group)
        L = _lib.lib()
        dev = npc.get_geo_feats().device
        flag_dev = dev if dev.type == "cuda" and dist.get_backend(group) == "nccl" else "cpu"

        def all_ok(ok: bool) -> bool:
            t = torch.tensor([1 if ok else 0], device=flag_dev, dtype=torch.int32)
            dist.all_reduce(t, op=dist.ReduceOp.MIN, group=group)
            return int(t.item()) == 1
        # pre-init agreement (advisor r4): EVERY rank does everything psl_comm_init does in front of ncclCommInitRank -- dlopen of
        # librccl + the resolution of its symbols and the device buffers of the counts phase, all inside psl_comm_reserve --
        # and the ranks compare notes BEFORE anybody enters the communicator's rendezvous.  Only rank 0 ever creates a unique
        # id (advisor r5: every ncclGetUniqueId starts a bootstrap root -- a listening socket and a thread -- that is never
        # torn down, and needs a usable network interface, i.e. can fail for reasons that have nothing to do with this rank)
        rc0 = L.psl_comm_reserve(npc.handle, self.world)
        if not all_ok(rc0 >= 0):
            raise RuntimeError("native RCCL transport: librccl / device buffers unavailable on some rank"
                               + (": " + L.psl_last_error().decode() if rc0 < 0 else ""))
        ident = C.create_string_buffer(128)
        box = [None]
        if rank == 0:
            if L.psl_comm_unique_id(ident) >= 0:
                box = [bytes(ident.raw)]
            else:
                box = [None]                                     # tell the others instead of leaving them waiting
        dist.broadcast_object_list(box, src=0, group=group)      # the only use of torch.distributed on the data path set-up
        if box[0] is None:
            raise RuntimeError("psl_comm_unique_id failed on rank 0: " + (L.psl_last_error().decode() if rank == 0 else "(see rank 0)"))
        ident = C.create_string_buffer(box[0], 128)
        rc = L.psl_comm_init(npc.handle, ident, rank, self.world)
        ok = torch.tensor([1 if rc >= 0 else 0], device=flag_dev, dtype=torch.int32)
        dist.all_reduce(ok, op=dist.ReduceOp.MIN, group=group)
        if int(ok.item()) == 0:
            if rc >= 0:
                L.psl_comm_destroy(npc.handle)
            raise RuntimeError("psl_comm_init failed on some rank" + (": " + L.psl_last_error().decode() if rc < 0 else ""))
        self.buf = None

    def allgather_v(self, rec: torch.Tensor) -> Tuple[torch.Tensor, List[int]]:
        _lib, L = self._lib, self._lib.lib()
        dev, width = rec.device, rec.shape[1]
        rec = rec.contiguous()
        counts = (C.c_int32 * self.world)()
        if self.buf is None or self.buf.shape[1] != width:
            # first guess from this rank's own block; from then on the buffer grows through the CAPACITY round below only
            self.buf = torch.empty(max(4 * self.world * max(rec.shape[0], 1024), 65536), width, device=dev, dtype=torch.float32)
        while True:
            rc = L.psl_allgather_new_points(self.npc.handle, None, 0, _lib.ptr(rec), rec.shape[0], width, _lib.ptr(self.buf),
                                            self.buf.shape[0], counts, _lib.stream_ptr())
            if rc == -3:
                # PSL_ERR_CAPACITY is decided on the gathered (rows, capacity) pairs -- total > the SMALLEST buffer of any
                # rank -- so every rank is here together, before the records collective, with the same counts: all grow to
                # the same size and repeat the same sequence (buffers of different size on different ranks are fine)
                self.buf = torch.empty(2 * sum(counts) + 1024, width, device=dev, dtype=torch.float32)
                continue
            total = _lib.check(rc, "psl_allgather_new_points")
            return self.buf[:total].clone(), [int(c) for c in counts]


def make_transport(npc, group=None, kind: Optional[str] = None):
    """`kind` None: torch.distributed collectives (backend "nccl" = RCCL over xGMI on a multi-GPU node; gloo in the CPU tests
    and for ranks sharing one GPU).  "native" / PSL_NATIVE_RCCL=1: the library's own RCCL all-gather-v -- opt-in until it has
    run with more than one rank (module docstring)."""
    forced = kind is not None
    if kind is None:
        env = os.environ.get("PSL_NATIVE_RCCL")
        kind, forced = ("native", True) if env == "1" else ("torch", env == "0")
    if kind == "native":
        try:
            return _NativeTransport(npc, group)
        except RuntimeError as e:
            if forced:
                raise
            # the default choice could not be brought up (on all ranks alike, see _NativeTransport.__init__): the same
            # exchange through torch.distributed's collectives -- said loudly, the path is not silently swapped
            import sys
            print(f"[point_slam_amd.dist] native RCCL transport unavailable ({e}); using torch.distributed collectives",
                  file=sys.stderr, flush=True)
    return _TorchTransport(group)


def transport_name(tr) -> str:
    return "native (psl_allgather_new_points on the library's RCCL communicator)" if isinstance(tr, _NativeTransport) \
        else f"torch.distributed all_gather ({dist.get_backend(tr.group)})"


# ------------------------------------------------------------------------------------------------------- new points
def merge_new_points(npc, n_base: int, group=None, dedupe: bool = True, transport=None, creators: Optional[list] = None) -> List[int]:
    """Exchange the points `npc` gained since it had n_base points and rebuild the replica in global rank order
    (steps 1-2 of the module docstring).  Returns the number of points every rank CONTRIBUTED; the number admitted
    after the cross-rank dedupe is npc.pts_num() - n_base.  creators: a list that receives ONE int16 tensor, the
    contributing rank of every admitted point in append order (the owner-writes merge rule needs it)."""
    tr = transport or _TorchTransport(group)
    n_now = npc.pts_num()
    n_new = n_now - n_base
    dev = npc.get_geo_feats().device
    rec = torch.empty(n_new, REC, device=dev, dtype=torch.float32)
    if n_new:
        rec[:, :3] = npc.cloud_pos_device(n_base, n_new)          # tail only
        rec[:, 3:35] = npc.get_geo_feats()[n_base:n_now]
        rec[:, 35:67] = npc.get_col_feats()[n_base:n_now]
        rec[:, 67] = npc.point_radius(n_base, n_new)
    allrec, counts = tr.allgather_v(rec)
    if sum(counts) == 0:
        return counts                                             # nobody added anything: the replica stands as it is
    offs = [0]
    for c in counts:
        offs.append(offs[-1] + c)
    keep = torch.ones(allrec.shape[0], dtype=torch.bool, device=dev)
    if dedupe:
        # locations are triplets (N_add = 3, neural_point.py:126-143); the middle point is the surface point
        loc = allrec[:, :3].reshape(-1, 3, 3)[:, 1, :].contiguous()
        rad = allrec[:, 67].reshape(-1, 3)[:, 1].contiguous()
        first = next(k for k, c in enumerate(counts) if c > 0)    # the first non-empty block is admitted whole
        lo = offs[first + 1] // 3
        keep_loc = torch.ones(loc.shape[0], dtype=torch.uint8, device=dev)
        if lo < loc.shape[0] and n_base > 0:
            # vs the BASE map: one launch on the index as it stands (it still covers this rank's own tail; indices
            # >= n_base are ignored) -- no rebuild before the test
            keep_loc[lo:] = (npc.count_within(loc[lo:], rad[lo:], n_base) == 0).to(torch.uint8)
        # vs the blocks before it, in rank order (one launch per block, flags final when the next block reads them)
        npc.dedupe_blocks(allrec, 0, 67, [o // 3 for o in offs], keep_loc)
        keep_loc = keep_loc.bool()
        keep = keep_loc[:, None].expand(-1, 3).reshape(-1)
    kept = allrec[keep]
    if creators is not None:
        blk = torch.repeat_interleave(torch.arange(len(counts), dtype=torch.int16, device=dev),
                                      torch.tensor(counts, device=dev))
        creators.append(blk[keep])
    npc.truncate(n_base)
    # ONE append and ONE index rebuild for all blocks (also when nothing was kept: truncate left the index stale)
    npc.append_points(kept[:, :3].contiguous(), kept[:, 3:35].contiguous(), kept[:, 35:67].contiguous(),
                      radius=kept[:, 67].contiguous(), build=True)
    return counts


# ------------------------------------------------------------------------------------------------------- reconciliation
class FrameParallelSync:
    """State of the periodic reconciliation: compact snapshots of the feature rows trained since the last exchange
    (`note_rows`) and of the colour-decoder blob."""

    def __init__(self, npc, theta: Optional[torch.Tensor] = None, n_color: int = 0, group=None,
                 transport: Optional[str] = None, merge: str = "mean"):
        if merge not in ("mean", "owner"):
            raise ValueError("merge must be 'mean' or 'owner'")
        self.group = group
        self.merge = merge
        self._owner = None          # int16 [capacity]: rank that created a point, -1 = the seed map (owner-writes rule)
        self.n_color = n_color
        self.transport = make_transport(npc, group, transport)
        self.snap_theta = theta[:n_color].clone() if theta is not None else None
        self.n_base = npc.pts_num()
        self._slot = None           # int32 [capacity]: snapshot slot of a row, -1 = none (O(touched) to reset)
        self._rows: List[torch.Tensor] = []
        self._vals: List[torch.Tensor] = []
        self._n_slots = 0
        self.last_stats = {}

    def note_rows(self, npc, rows: torch.Tensor):
        """Rows about to be trained (HipSLAM.map calls this with the frustum selection before psl_map_iters): those that
        have no snapshot yet in this exchange interval get one.  O(len(rows))."""
        geo, col = npc.get_geo_feats(), npc.get_col_feats()
        rows = rows.long()
        rows = rows[rows < self.n_base]
        if rows.numel() == 0:
            return
        cap = max(getattr(npc, "_max_points", 0) or 0, geo.shape[0])
        if self._slot is None or self._slot.shape[0] < cap:
            old = self._slot
            self._slot = torch.full((cap,), -1, dtype=torch.int32, device=geo.device)
            if old is not None:
                self._slot[:old.shape[0]] = old
        new = rows[self._slot[rows] < 0]
        if new.numel() == 0:
            return
        new = torch.unique(new)
        self._slot[new] = torch.arange(self._n_slots, self._n_slots + new.numel(), dtype=torch.int32, device=geo.device)
        self._n_slots += int(new.numel())
        self._rows.append(new)
        self._vals.append(torch.cat([geo[new], col[new]], 1))

    def _reconcile_rows(self, npc):
        geo, col = npc.get_geo_feats(), npc.get_col_feats()
        dev = geo.device
        if self._rows:
            rows, snap = torch.cat(self._rows), torch.cat(self._vals)
            delta = torch.cat([geo[rows], col[rows]], 1) - snap
            ch = torch.nonzero((delta != 0).any(1)).flatten()        # the one host sync of this half
            rows_c, delta_c, snap_c = rows[ch], delta[ch], snap[ch]
        else:
            rows_c = torch.zeros(0, dtype=torch.int64, device=dev)
            delta_c = snap_c = torch.zeros(0, 64, device=dev)
        rec = torch.empty(rows_c.shape[0], REC_ROW, device=dev, dtype=torch.float32)
        rec[:, 0] = rows_c.to(torch.int32).view(torch.float32)      # the id travels as raw bits (collectives only copy)
        rec[:, 1:] = delta_c
        allrec, counts = self.transport.allgather_v(rec)
        self.last_stats = dict(rows_noted=int(self._n_slots), rows_sent=int(rows_c.shape[0]), rows_received=int(sum(counts)))
        if sum(counts):
            ids = allrec[:, 0].contiguous().view(torch.int32).long()
            uniq, inv = torch.unique(ids, return_inverse=True)       # sorted: the same order on every rank
            delta_all = allrec[:, 1:]
            allrec[:, 0] = 1.0                                       # column 0 now counts the contributors of a row
            tot = torch.zeros(uniq.shape[0], REC_ROW, device=dev)
            off, mine = 0, None
            rank = dist.get_rank(self.group)
            for k, c in enumerate(counts):                           # rank order; ids are unique inside a block: plain adds
                if c:
                    sl = inv[off:off + c]
                    tot.index_add_(0, sl, allrec[off:off + c])
                    if k == rank:
                        mine = sl
                off += c
            base = torch.cat([geo[uniq], col[uniq]], 1)              # rows this rank did not change still hold the snapshot
            if mine is not None:
                base[mine] = snap_c
            if self.merge == "owner":
                # SURVEY.md 8e owner-writes: the change of the rank that created the point if it is among the contributors,
                # else the lowest contributing rank's (seed-map points have no creator) -- one contributor's change, whole
                if self._owner is not None and self._owner.shape[0] < geo.shape[0]:       # the store grew since the table was sized
                    self._owner = torch.cat([self._owner, torch.full((geo.shape[0] - self._owner.shape[0],), -1, dtype=torch.int16, device=dev)])
                own = self._owner[uniq] if self._owner is not None else torch.full_like(uniq, -1, dtype=torch.int16)
                best = torch.full((uniq.shape[0],), 1 << 20, dtype=torch.int64, device=dev)
                win = torch.zeros(uniq.shape[0], 64, device=dev)
                off = 0
                for k, c in enumerate(counts):
                    if c:
                        sl = inv[off:off + c]
                        prio = torch.where(own[sl] == k, torch.full_like(sl, -1), torch.full_like(sl, k))
                        better = prio < best[sl]
                        best[sl[better]] = prio[better]
                        win[sl[better]] = delta_all[off:off + c][better]
                    off += c
                new = base + win
            else:
                new = base + tot[:, 1:] / tot[:, :1]
            geo[uniq] = new[:, :32]
            col[uniq] = new[:, 32:]
        if self._rows:
            self._slot[torch.cat(self._rows)] = -1
        self._rows, self._vals, self._n_slots = [], [], 0

    def exchange(self, npc, theta: Optional[torch.Tensor] = None, dedupe: bool = True) -> List[int]:
        """One exchange (all four steps).  `theta`: the master parameter blob, updated in place (colour group)."""
        # HOST time of the three phases (what the calling thread spends inside each: enqueueing, the collectives' host side and the
        # host synchronisations they contain -- on a GPU the device work overlaps the next phase's enqueue): `last_host_ms`
        t0 = time.perf_counter()
        # 3. features of the points that existed at the last exchange
        self._reconcile_rows(npc)
        t1 = time.perf_counter()
        # 4. colour decoder
        if theta is not None and self.snap_theta is not None:
            d = theta[:self.n_color] - self.snap_theta
            dist.all_reduce(d, op=dist.ReduceOp.SUM, group=self.group)
            theta[:self.n_color] = self.snap_theta + d / dist.get_world_size(self.group)
        t2 = time.perf_counter()
        # 1-2. new points
        creators: list = []
        counts = merge_new_points(npc, self.n_base, self.group, dedupe, self.transport, creators if self.merge == "owner" else None)
        t3 = time.perf_counter()
        self.last_host_ms = dict(rows=round((t1 - t0) * 1e3, 3), decoder=round((t2 - t1) * 1e3, 3), new_points=round((t3 - t2) * 1e3, 3))
        if self.merge == "owner" and creators and creators[0].numel():     # the creator table only serves the owner-writes rule
            geo = npc.get_geo_feats()
            cap = max(getattr(npc, "_max_points", 0) or 0, geo.shape[0], self.n_base + int(creators[0].numel()))
            if self._owner is None or self._owner.shape[0] < cap:
                old = self._owner
                self._owner = torch.full((cap,), -1, dtype=torch.int16, device=geo.device)
                if old is not None:
                    self._owner[:old.shape[0]] = old
            self._owner[self.n_base:self.n_base + creators[0].numel()] = creators[0]
        if theta is not None:
            self.snap_theta = theta[:self.n_color].clone()
        self.n_base = npc.pts_num()
        return counts


def frames_of_rank(n_frames: int, rank: int, world: int) -> List[int]:
    """Frame t is processed by rank t mod world (SURVEY.md §8e partitioning)."""
    return [t for t in range(n_frames) if t % world == rank]
