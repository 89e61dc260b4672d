"""The timed region of bench.py: W warm-up steps, then EXACTLY K steps between barrier + synchronize, MAX over ranks."""
from __future__ import annotations

import time

import numpy as np
import torch
import torch.distributed as dist


def ranks_seen(dev):
    """every rank adds one on ITS device: what comes back is the number of ranks the backend (RCCL on the GPUs, gloo in the CPU
    tests) actually reduced over -- bench.py reports it as `rccl_ranks_seen` and refuses a line whose --gpus says otherwise"""
    ones = torch.ones(1, dtype=torch.int32, device=dev)
    dist.all_reduce(ones)
    return int(ones.item())


class Timer:
    """W warm-up steps, then EXACTLY K steps between barrier + synchronize, wall-clock MAX over ranks; that K-step region
    is repeated (warm-up only before the first) until min_seconds of device time is accumulated.  The kernels run on
    torch's current stream (ctx.set_stream) or, with `lanes`, in an unordered section that forks from and joins into it, so
    the torch.cuda.Event pair around the K launches is a hipEvent pair on the launch stream bracketing all of them:
    `dev_ms` = device time of each region.  The wall clock is read after the device has drained and BEFORE the trailing
    barrier (a RCCL barrier is a kernel launch plus a host sync that only N > 1 would pay); the MAX over ranks is taken from
    the per-rank times afterwards."""

    def __init__(self, K, Wm, use_dist, dev, min_seconds, max_repeats, ctx=None, lanes=0):
        self.K, self.Wm, self.use_dist, self.dev = K, Wm, use_dist, dev
        self.min_seconds, self.max_repeats = min_seconds, max_repeats
        self.ctx, self.lanes = ctx, lanes
        self.cuda = torch.device(dev).type == "cuda"      # (the gloo tests drive the same loop on the CPU: wall clock only)

    def _sync(self):
        if self.cuda:
            torch.cuda.synchronize()

    def region(self, fn, first, lanes):
        self._sync()
        if self.use_dist:
            dist.barrier()
        self._sync()
        e0 = e1 = None
        if self.cuda:
            e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
        t0 = time.perf_counter()
        if self.cuda:
            e0.record()
        if lanes:
            self.ctx.begin_unordered(lanes)
        for i in range(self.K):
            fn(first + i)
        if lanes:
            self.ctx.end_unordered()
        if self.cuda:
            e1.record()
        self._sync()
        wall = time.perf_counter() - t0
        if self.use_dist:
            dist.barrier()
        return wall, (e0.elapsed_time(e1) if self.cuda else 1e3 * wall)

    def run(self, fn, lanes=None):
        lanes = self.lanes if lanes is None else lanes
        # warm-up in the mode that is timed: the lane streams of an unordered section are created, and get their first launch, here
        # (the first launch of a large-LDS kernel on a fresh stream has been seen to take seconds, once)
        if lanes and self.Wm > 0:
            self.ctx.begin_unordered(lanes)
        for i in range(self.Wm):
            fn(i)
        if lanes and self.Wm > 0:
            self.ctx.end_unordered()
        walls, devs = [], []
        step = self.Wm
        while True:
            wall, dev_ms = self.region(fn, step, lanes)
            step += self.K
            walls.append(wall)
            devs.append(dev_ms)
            # every rank must take the same decision: rank 0 decides
            more = torch.tensor([1 if (sum(devs) * 1e-3 < self.min_seconds and len(devs) < self.max_repeats) else 0],
                                dtype=torch.int32, device=self.dev)
            if self.use_dist:
                dist.broadcast(more, src=0)
            if int(more.item()) == 0:
                break
        mine = float(np.median(walls))                           # this rank's median region
        t = torch.tensor(walls, dtype=torch.float64, device=self.dev)
        ranks = [mine]
        if self.use_dist:
            dist.all_reduce(t, op=dist.ReduceOp.MAX)          # per region: the slowest rank
            g = [torch.zeros(1, dtype=torch.float64, device=self.dev) for _ in range(dist.get_world_size())]
            dist.all_gather(g, torch.tensor([mine], dtype=torch.float64, device=self.dev))
            ranks = [float(x.item()) for x in g]
        walls = t.cpu().numpy()
        return {"wall_median": float(np.median(walls)), "wall_min": float(walls.min()), "wall_max": float(walls.max()),
                "dev_ms_median": float(np.median(devs)), "repeats": len(devs), "seconds": float(walls.sum()),
                "rank_wall_medians": ranks}
