"""The resident stream of a plan in HBM.

`ResidentStream` holds EXACTLY the `nbatch` batches the plan (benchlib/plan.py) says: as many of them as the chunk pool has
chunks for are placed (input / Y / U+V / decoded output in different HBM region groups), the others
live in plain allocations.  Short memory never shrinks the stream: it raises `StreamDoesNotFit` (bench.py: exit code 3)
unless the caller allowed a short stream, and then `.degraded` is set.  Addresses only -- results never depend on them.
"""
from __future__ import annotations

import sys

import torch

from .plan import PACKED_RING, pool_request


class StreamDoesNotFit(RuntimeError):
    pass


def make_small_pool(L, dev, local_rank, n_float=2, n_y=1, n_uv=1):
    """lumahip_pool_create_small: a few chunks for a caller that keeps a few GB resident and shares the GPU (None: not available)"""
    try:
        from lumahdrv_amd.placement import HbmChunkPool
        ctx = L.Context(local_rank)
        ctx.set_stream(torch.cuda.current_stream().cuda_stream)
        ctx.set_quantizer(L.PTF_PQ, 11, L.CS_LUV, 8, 1e4, 0.005, L.build_lut(L.PTF_PQ, 11, 1e4, 0.005))
        pool = HbmChunkPool(ctx, dev, n_float, n_y, n_uv, 0, small=True)
        ctx.close()
        if len(pool.float) >= n_float and pool.y and pool.uv:
            return pool
        pool.close()
    except Exception as e:
        sys.stderr.write("bench.py: small chunk pool unavailable (%r)\n" % (e,))
        torch.cuda.empty_cache()
    return None


def make_pool(L, args, dev, local_rank, w, h, B, nbatches=None, with_output=True):
    """--placement auto: the chunk pool (C ABI lumahip_pool_*) the resident streams are carved from (None: plain allocations)"""
    if args.placement != "auto":
        return None
    try:
        from lumahdrv_amd.placement import HbmChunkPool
        req = pool_request(w, h, B, args.decode_layout, nbatches, with_output)
        if req is None:
            return None
        n_float, n_y, n_uv, n_striped = req["n_float"], req["n_y"], req["n_uv"], req["n_striped"]
        ctx = L.Context(local_rank)
        ctx.set_stream(torch.cuda.current_stream().cuda_stream)
        ctx.set_quantizer(L.PTF_PQ, 11, L.CS_LUV, 8, 1e4, 0.005, L.build_lut(L.PTF_PQ, 11, 1e4, 0.005))
        pool = HbmChunkPool(ctx, dev, n_float, n_y, n_uv, n_striped)
        ctx.close()
        if pool.float and pool.y and pool.uv:
            return pool
        pool.close()
        return None
    except Exception as e:      # placement is an optimisation, never a reason to fail the bench
        sys.stderr.write("bench.py: chunk pool unavailable (%r), plain allocations\n" % (e,))
        torch.cuda.empty_cache()
        return None


class ResidentStream:
    """`nbatch` batches of `B` frames of w x h: float input frames, decoded output frames, coded planes.

    ptrs(b) -> (input pointer, [R, G, B output plane pointers of the batch's first frame], [Y, U, V plane pointers]);
    `out_fs` = frame stride of the decoded output in floats (n1 when the R, G, B planes of a batch live in three buffers --
    lumahip_decode_frames_device_planar -- else n3, the packed LumaFrame)."""

    def __init__(self, dev, w, h, B, nbatch, psz, pool=None, want_output=True, striped_ok=True, allow_short=False):
        from lumahdrv_amd.placement import CHUNK_BYTES, slots
        self.dev, self.B, self.nbatch, self.pool = dev, B, nbatch, pool
        self.n1 = n1 = w * h
        self.n3 = n3 = 3 * n1
        self.psz = psz
        self.degraded = False
        self.src_c, self.out_c, self.y_c, self.uv_c, self.rgb_c = [], [], [], [], None
        ypc, self.yslot = slots(CHUNK_BYTES, B * psz[0])
        uvpc, self.uvslot = slots(CHUNK_BYTES, B * psz[1] + (1 << 20) + B * psz[2])
        spc, self.sslot = slots(CHUNK_BYTES, B * n1 * 4)       # one colour plane of one batch per slot
        self.ypc, self.uvpc, self.spc = ypc, uvpc, spc
        self.vo = (B * psz[1] + (1 << 20) - 1) // (1 << 20) * (1 << 20)
        if pool is not None and (B * n3 * 4 > CHUNK_BYTES or ypc < 1 or uvpc < 1):
            pool = self.pool = None
        # float frames: one chunk per batch of input; decoded output either packed (one chunk per batch) or, when the pool kept
        # chunks of three region groups for it, with the R, G and B planes of a batch in three different groups
        # (lumahip_decode_frames_device_planar); Y planes of `ypc` batches per chunk of one group; U and V planes of `uvpc`
        # batches per chunk of another (lumahdrv_amd/csrc/lumahip_pool.hip)
        self.striped = bool(pool is not None and want_output and striped_ok and spc >= 1 and min(len(g) for g in pool.striped) >= 1)
        placed = 0
        if pool is not None:
            per_float = 1 if (self.striped or not want_output) else 2
            placed = nbatch
            while placed > 0 and (placed * per_float > len(pool.float) or -(-placed // uvpc) > len(pool.uv) or -(-placed // ypc) > len(pool.y)
                                  or (self.striped and -(-placed // spc) > min(len(g) for g in pool.striped))):
                placed -= 1
        self.placed = placed
        if placed:
            self.src_c = pool.take_float(placed)                   # fastest first: the input gets the best chunks
            self.out_c = pool.take_float(placed) if (want_output and not self.striped) else []
            self.rgb_c = pool.take_striped(-(-placed // spc)) if self.striped else None
            self.uv_c = pool.take_uv(-(-placed // uvpc))
            self.y_c = pool.take_y(-(-placed // ypc))
            for c in self.uv_c + self.y_c:
                c.zero_()
        # every batch the pool has no chunks for: plain allocations, the SAME frames
        rest = nbatch - placed
        self.src = self.out = self.planes = None
        self.out_planar = self.striped                          # plain batches keep the layout of the placed ones
        while rest > 0:
            try:
                self.src = torch.empty(rest * B * n3, dtype=torch.float32, device=dev)
                if not want_output:
                    self.out = None
                elif self.out_planar:
                    self.out = [torch.empty(rest * B * n1, dtype=torch.float32, device=dev) for _ in range(3)]
                else:
                    self.out = torch.empty(rest * B * n3, dtype=torch.float32, device=dev)
                self.planes = [torch.zeros(rest * B * psz[p], dtype=torch.uint8, device=dev) for p in range(3)]
                break
            except torch.OutOfMemoryError:
                self.src = self.out = self.planes = None
                torch.cuda.empty_cache()
                free = torch.cuda.mem_get_info(dev)[0]
                if not allow_short or rest <= 1 and placed == 0:
                    self.close()
                    raise StreamDoesNotFit(
                        "the resident stream of this configuration (%d frames of %dx%d = %d batches of %d; %.1f GB with decoded output and "
                        "planes) does not fit this GPU's free HBM (%.1f GB free after %d placed batches).  The workload is a function of "
                        "the arguments, not of free memory: free the GPU, or pass --allow-short-stream for a shorter stream "
                        "(reported as config_degraded)" % (nbatch * B, w, h, nbatch, B, nbatch * B * (n3 * 4 * (2 if want_output else 1) + sum(psz)) / 1e9,
                                                           free / 1e9, placed))
                rest -= 1
                self.nbatch -= 1
                self.degraded = True
        self.out_fs = n1 if self.out_planar else n3
        self.nframes = self.nbatch * B

    def ptrs(self, b):
        n1, n3, B, psz = self.n1, self.n3, self.B, self.psz
        if b < self.placed:
            u = self.uv_c[b // self.uvpc].data_ptr() + (b % self.uvpc) * self.uvslot
            if self.striped:
                o = [self.rgb_c[k][b // self.spc].data_ptr() + (b % self.spc) * self.sslot for k in range(3)]
            elif self.out_c:
                o = [self.out_c[b].data_ptr() + k * n1 * 4 for k in range(3)]
            else:
                o = None
            return (self.src_c[b].data_ptr(), o, [self.y_c[b // self.ypc].data_ptr() + (b % self.ypc) * self.yslot, u, u + self.vo])
        r = b - self.placed
        if self.out is None:
            o = None
        elif self.out_planar:
            o = [self.out[k].data_ptr() + r * B * n1 * 4 for k in range(3)]
        else:
            o = [self.out.data_ptr() + r * B * n3 * 4 + k * n1 * 4 for k in range(3)]
        return (self.src.data_ptr() + r * B * n3 * 4, o, [self.planes[p].data_ptr() + r * B * psz[p] for p in range(3)])

    def placement_report(self):
        return {"batches": self.nbatch, "batches_in_pool_chunks": self.placed, "batches_in_plain_allocations": self.nbatch - self.placed,
                "decode_output": "R, G, B planes of a batch in three buffers" if self.out_planar else "packed LumaFrame"}

    def close(self):
        if self.pool is not None and self.placed:
            self.pool.give_back(self.src_c + self.out_c, self.y_c, self.uv_c, self.rgb_c)
        self.src_c, self.out_c, self.y_c, self.uv_c, self.rgb_c, self.placed = [], [], [], [], None, 0
        self.src = self.out = self.planes = None
        torch.cuda.empty_cache()


__all__ = ["ResidentStream", "StreamDoesNotFit", "make_pool", "make_small_pool", "PACKED_RING"]
