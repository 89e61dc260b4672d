"""Frame-level sharding of the hot path across the GPUs of one node (one process per GPU).

Frames are independent in the hot path: the only shared state is the read-only quantizer
(transfer-function table + six scalars).  So the multi-GPU mode is
  * rank r owns the contiguous block ``shard_range(nframes, r, world)`` of the stream (block, not
    round-robin, so each GPU's output is already in stream order for the sequential VP9 consumer);
  * rank 0 builds the table on its host and broadcasts table + parameters once (RCCL over xGMI when the
    process group is "nccl"; the same code runs on "gloo" for the CPU tests);
  * no collective on the data path; an optional all_gather of per-frame digests / mean luminances at the end
    for verification and in-order reassembly bookkeeping.
Nothing here computes pixels.
"""
from __future__ import annotations

import numpy as np
import torch
import torch.distributed as dist


def shard_range(nframes: int, rank: int, world: int) -> range:
    """contiguous block of frame indices owned by `rank`; the first nframes % world ranks get one more"""
    if world <= 0 or not (0 <= rank < world):
        raise ValueError("bad rank/world")
    base, extra = divmod(nframes, world)
    start = rank * base + min(rank, extra)
    return range(start, start + base + (1 if rank < extra else 0))


def owner_of(frame: int, nframes: int, world: int) -> int:
    base, extra = divmod(nframes, world)
    edge = extra * (base + 1)
    if frame < edge:
        return frame // (base + 1)
    return extra + (frame - edge) // base if base else world - 1


def _world():
    return dist.get_world_size() if dist.is_available() and dist.is_initialized() else 1


def _rank():
    return dist.get_rank() if dist.is_available() and dist.is_initialized() else 0


def broadcast_quantizer(cfg, lut, device, src: int = 0):
    """cfg = (ptf, bitdepth, cs, bitdepthC, maxLum, minLum, preScaling, profile) and the table, valid on rank
    `src` (None elsewhere).  Returns (cfg, lut ndarray) on every rank.  Two tiny broadcasts: an 8-float
    parameter block, then the 2^bitdepth-float table (4-64 KiB)."""
    p = torch.zeros(8, dtype=torch.float64, device=device)
    if _rank() == src:
        p.copy_(torch.tensor([float(x) for x in cfg], dtype=torch.float64))
    if _world() > 1:
        dist.broadcast(p, src=src)
    v = p.cpu().tolist()
    cfg = (int(v[0]), int(v[1]), int(v[2]), int(v[3]), float(np.float32(v[4])), float(np.float32(v[5])),
           float(np.float32(v[6])), int(v[7]))
    t = torch.zeros(1 << cfg[1], dtype=torch.float32, device=device)
    if _rank() == src:
        t.copy_(torch.from_numpy(np.ascontiguousarray(lut, dtype=np.float32)))
    if _world() > 1:
        dist.broadcast(t, src=src)
    return cfg, t.cpu().numpy()


def gather_in_stream_order(local_values, nframes: int, device):
    """every rank contributes one int64 per owned frame (digest, byte count, ...); returns the nframes values
    in stream order on every rank"""
    world, rank = _world(), _rank()
    mine = shard_range(nframes, rank, world)
    assert len(local_values) == len(mine)
    cap = (nframes + world - 1) // world
    buf = torch.zeros(cap, dtype=torch.int64, device=device)
    if len(mine):
        buf[:len(mine)] = torch.tensor([int(x) & 0x7FFFFFFFFFFFFFFF for x in local_values], dtype=torch.int64)
    if world > 1:
        parts = [torch.zeros_like(buf) for _ in range(world)]
        dist.all_gather(parts, buf)
    else:
        parts = [buf]
    out = []
    for r in range(world):
        n = len(shard_range(nframes, r, world))
        out.extend(parts[r][:n].cpu().tolist())
    return out


class ShardedStream:
    """Drives `process(frame_index) -> int` (e.g. encode the frame on this rank's GPU and return a digest of its
    planes) over this rank's block, after broadcasting the quantizer.  `make_worker(cfg, lut)` builds the
    rank-local state (a lumahdrv_amd.Context in production; the CPU tests inject the oracle)."""

    def __init__(self, nframes: int, device, cfg=None, lut=None, src: int = 0):
        self.nframes = nframes
        self.device = device
        self.cfg, self.lut = broadcast_quantizer(cfg, lut, device, src)
        self.frames = shard_range(nframes, _rank(), _world())

    def run(self, make_worker, process):
        worker = make_worker(self.cfg, self.lut)
        local = [process(worker, f) for f in self.frames]
        return gather_in_stream_order(local, self.nframes, self.device)
