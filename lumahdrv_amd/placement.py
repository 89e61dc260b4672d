"""HBM placement for bench.py and the tests: a torch-tensor view of the C ABI's chunk pool (include/lumahip.h
`lumahip_pool_*`, lumahdrv_amd/csrc/lumahip_pool.hip -- the measurements, the grouping and the choice of chunks all happen
there; DESIGN.md section 2 says why it exists).  This module only wraps the chunks the pool hands out as uint8 tensors so
that the bench can slice, zero and digest them, and keeps the slot arithmetic of the resident streams.

Nothing here touches results: the pool decides addresses only.
"""
from __future__ import annotations

CHUNK_BYTES = 2 << 30


def plane_slots(chunk_bytes, batch_plane_bytes, align=1 << 20):
    """how many batches' Y/U/V planes fit one chunk, the slot size, and the byte offsets (Y, U, V) inside a slot"""
    offs, o = [], 0
    for b in batch_plane_bytes:
        offs.append(o)
        o = (o + b + align - 1) // align * align
    slot = (o + (64 << 20) - 1) // (64 << 20) * (64 << 20)
    per_chunk = max(1, chunk_bytes // slot) if slot <= chunk_bytes else 0
    return per_chunk, slot, offs


def slots(chunk_bytes, nbytes, align=64 << 20):
    """(how many buffers of nbytes fit one chunk, the slot size)"""
    slot = (nbytes + align - 1) // align * align
    return (chunk_bytes // slot if slot <= chunk_bytes else 0), slot


def find_groups(n, probe):
    """the pool's grouping step with a Python probe(i, r) (measurement tools, CPU tests): (groups | None, fastest, probes)"""
    from . import capi
    return capi.pool_find_groups(n, probe)


class _Raw:
    """a raw device pointer dressed up for torch.as_tensor (CUDA array interface v2)"""

    def __init__(self, ptr, nbytes):
        self.__cuda_array_interface__ = {"shape": (nbytes,), "typestr": "|u1", "data": (int(ptr), False), "version": 2}


def as_tensor(ptr, nbytes, dev):
    """uint8 tensor aliasing `nbytes` of device memory at `ptr` (owned by someone else: here, the pool)"""
    import torch
    return torch.as_tensor(_Raw(ptr, nbytes), device=dev)


class HbmChunkPool:
    """`n_float` chunks for float frames, `n_y` for Y planes, `n_uv` for U / V planes and `n_striped` chunks from each of the
    first three region groups (channel-strided decode output), chosen by lumahip_pool_create through `ctx` (needs a quantizer
    set; probing overwrites the chunks).  Lists of uint8 tensors: .float, .y, .uv, .striped[g]; fastest first."""

    def __init__(self, ctx, dev, n_float, n_y, n_uv, n_striped=0, keep_free=6 << 30, iters=2, small=False):
        from . import capi
        self.dev = dev
        self.pool = capi.Pool(ctx, n_float, n_y, n_uv, n_striped, CHUNK_BYTES, keep_free, 0, iters, small=small)
        self.stats = self.pool.stats()
        self._ptr = {}

        self.float = self._take_all(capi.POOL_FLOAT)
        self.y = self._take_all(capi.POOL_Y)
        self.uv = self._take_all(capi.POOL_UV)
        self.striped = [self._take_all(capi.POOL_STRIPED, g) for g in range(3)]

    def _wrap(self, p):
        t = as_tensor(p, CHUNK_BYTES, self.dev)
        self._ptr[t.data_ptr()] = p
        return t

    def _take_all(self, kind, group=-1):
        out = []
        while self.pool.available(kind, group) > 0:
            out.append(self._wrap(self.pool.alloc(kind, group)))
        return out

    def group_of(self, t) -> int:
        return self.pool.group_of(t.data_ptr())

    @staticmethod
    def _take(lst, n, what):
        if n > len(lst):
            raise RuntimeError("HbmChunkPool: %d %s chunks wanted, %d left" % (n, what, len(lst)))
        return lst[:n], lst[n:]

    def take_float(self, n):
        got, self.float = self._take(self.float, n, "float")
        return got

    def take_y(self, n):
        got, self.y = self._take(self.y, n, "Y")
        return got

    def take_uv(self, n):
        got, self.uv = self._take(self.uv, n, "U/V")
        return got

    def take_striped(self, n):
        """n chunks from each of the three groups: [[R chunks], [G chunks], [B chunks]]"""
        out = []
        for g in range(3):
            got, self.striped[g] = self._take(self.striped[g], n, "striped (group %d)" % g)
            out.append(got)
        return out

    def take_rotating(self, n):
        """n chunks for consecutive batches of PACKED frames through the C pool's own allocation mode (include/lumahip.h
        LUMAHIP_POOL_ROTATING: consecutive allocations walk the region groups 0, 1, 2, 0, ...) -- the caller does not look at groups"""
        from . import capi
        for g in range(3):                       # what this wrapper holds of the striped chunks goes back into the C pool first
            for t in self.striped[g]:
                self.pool.release(self._ptr[t.data_ptr()])
        self.striped = [[], [], []]
        if self.pool.available(capi.POOL_ROTATING) < n:
            self.striped = [self._take_all(capi.POOL_STRIPED, g) for g in range(3)]
            raise RuntimeError("HbmChunkPool: %d rotating chunks wanted, fewer left" % n)
        got = [self._wrap(self.pool.alloc(capi.POOL_ROTATING, 0 if i == 0 else -1)) for i in range(n)]
        self.striped = [self._take_all(capi.POOL_STRIPED, g) for g in range(3)]
        return got

    def give_back_rotating(self, chunks):
        for t in chunks:
            self.striped[self.group_of(t)].insert(0, t)

    def give_back(self, floats, y, uv, striped=None):
        self.float = list(floats) + self.float
        self.y = list(y) + self.y
        self.uv = list(uv) + self.uv
        if striped:
            for g in range(3):
                self.striped[g] = list(striped[g]) + self.striped[g]

    def close(self):
        """frees every chunk: tensors taken from the pool must not be used afterwards"""
        self.float, self.y, self.uv, self.striped = [], [], [], [[], [], []]
        if self.pool is not None:
            self.pool.close()
            self.pool = None
