"""HBM placement: which device-memory regions the frames and the coded planes of a resident stream live in.

Why this exists (profiles/r02_placement.txt; tools/placement_pairs.py, tools/placement_combos.py): on MI355X (SPX / NPS1,
ROCm 7.2) device memory falls into a few GROUPS of multi-GiB regions, and the rate of a launch depends on which groups
the streams it reads and writes CONCURRENTLY live in.  One 20-frame launch of the encode traffic (12 B/pixel read from the
float frames, 2 B/pixel written to Y, 1 B/pixel to U and V), same box, same minute:

    input, Y, U, V all in regions of one group      0.464 ms      (the encode kernel itself: 0.475 ms)
    input in group A, Y U V together in group B     0.432 ms                                   0.443 ms
    input in A, Y in B, U V in A or in a third C     0.397 ms                                   0.412 - 0.417 ms

The relation is symmetric in the read / write roles, reproducible to three digits, independent of offsets inside a region
and not a property of a single region (every region is fast with some partners and slow with others); presumably the
regions of a group share DRAM resources on which concurrent streams collide -- the mechanism is not visible from user
space, the groups are.  A plain 50 GB allocation pairs its buffers at random, which is where the 4-8 % run-to-run spread
of the bench came from (and its three levels 0.400 / 0.437 / 0.465 ms per launch).

A resident-stream application owns its buffers for a long time, so it can afford to look first: take the device memory
in 2 GiB chunks, find the groups with a few traffic-only launches per chunk, put the Y planes into chunks of ONE group and
everything else (float frames, U and V planes) into chunks of the OTHER groups, and hand the rest back.

Nothing here touches results: the pool only decides which addresses the buffers live at.
"""
from __future__ import annotations

CHUNK_BYTES = 2 << 30
PROBE_W, PROBE_H, PROBE_FRAMES, PROBE_PROFILE = 3840, 2160, 20, 2     # 1.99 GB read + 0.50 GB written per probe launch
SAME_GROUP_PENALTY = 1.035     # a pair counts as "same group" when it is this much slower than the fastest pair seen
                               # (measured: +7 %; repeatability of one measurement: 0.5 %)


def plane_slots(chunk_bytes, batch_plane_bytes, align=1 << 20):
    """how many batches' Y/U/V planes fit one chunk, the slot size, and the byte offsets (Y, U, V) inside a slot"""
    offs, o = [], 0
    for b in batch_plane_bytes:
        offs.append(o)
        o = (o + b + align - 1) // align * align
    slot = (o + (64 << 20) - 1) // (64 << 20) * (64 << 20)
    per_chunk = max(1, chunk_bytes // slot) if slot <= chunk_bytes else 0
    return per_chunk, slot, offs


def find_groups(n, probe, max_groups=16):
    """Partition chunks 0..n-1 into groups; probe(i, r) = time of a launch that reads chunk i and writes chunk r.
    Round k takes the first unclassified chunk r as reference and times every other unclassified chunk against it: the
    slow ones share r's group.  Returns (groups, fastest time seen, number of probes), or (None, ...) when the first
    round shows no contrast (one group, or a machine without the effect)."""
    todo = list(range(n))
    groups, fast, probes = [], None, 0
    while todo and len(groups) < max_groups:
        r, others = todo[0], todo[1:]
        if not others:
            groups.append([r])
            todo = []
            break
        t = {i: probe(i, r) for i in others}
        probes += len(others)
        lo, hi = min(t.values()), max(t.values())
        if fast is None:
            if hi <= lo * SAME_GROUP_PENALTY:
                return None, lo, probes
            fast = lo
        fast = min(fast, lo)      # (a round whose chunks all share the reference's group has no fast pair: min keeps `fast`)
        same = [i for i in others if t[i] > fast * SAME_GROUP_PENALTY]
        groups.append([r] + same)
        todo = [i for i in others if i not in set(same)]
    if todo:                      # more groups than max_groups: the remainder becomes one last group
        groups.append(todo)
    return groups, fast, probes


def slots(chunk_bytes, nbytes, align=64 << 20):
    """(how many buffers of nbytes fit one chunk, the slot size)"""
    slot = (nbytes + align - 1) // align * align
    return (chunk_bytes // slot if slot <= chunk_bytes else 0), slot


class HbmChunkPool:
    """Takes the free device memory of `dev` in chunks of CHUNK_BYTES (leaving `keep_free` bytes) and chooses, by
    measurement, `n_y` chunks for Y planes, `n_uv` chunks for U / V planes and `n_float` chunks for float frames; the rest
    goes back to the driver.  Every measurement is the traffic-only launch of `ctx` (lumahip_probe_encode_traffic_device:
    the loads and stores of the 4:2:0 encode kernel, no arithmetic); `ctx` needs a quantizer set, probing overwrites the
    chunks.  Steps: (1) find the groups (find_groups); (2) Y candidates = the smallest group that is large enough, U / V
    candidates = another group, reference float chunk = first chunk of the largest remaining group; (3) keep the Y and
    U / V candidates that run fastest with the reference; (4) rank ALL remaining chunks as float chunks by their time
    with the chosen planes chunks and keep the fastest -- so an imperfect grouping costs probes, not bandwidth."""

    def __init__(self, ctx, dev, n_float, n_y, n_uv, keep_free=6 << 30, iters=2):
        import statistics
        import torch
        from . import capi
        self.dev = dev
        free, _ = torch.cuda.mem_get_info(dev)
        chunks = []
        for _ in range(max(0, int((free - keep_free) // CHUNK_BYTES))):
            try:
                chunks.append(torch.empty(CHUNK_BYTES, dtype=torch.uint8, device=dev))
            except RuntimeError:        # out of memory: use what we have
                break
        n = len(chunks)
        self.stats = {"chunk_GiB": CHUNK_BYTES / 2 ** 30, "chunks": n, "float_chunks": n_float, "y_chunks": n_y, "uv_chunks": n_uv}
        self.float, self.y, self.uv = [], [], []
        if n < n_float + n_y + n_uv or n < 4:
            self.stats["grouped"] = False
            self.stats["note"] = "not enough device memory for the chunk pool"
            chunks = None
            torch.cuda.empty_cache()
            return
        w, h, B, profile = PROBE_W, PROBE_H, PROBE_FRAMES, PROBE_PROFILE
        n3 = 3 * w * h
        _, hs, st, _ = capi.plane_geometry(w, h, profile)
        psz = [hs[p] * st[p] for p in range(3)]
        _, _, offs = plane_slots(CHUNK_BYTES, [B * x for x in psz])
        assert B * n3 * 4 <= CHUNK_BYTES
        nprobe = [0]

        def probe4(i, y, u, v):
            nprobe[0] += 1
            pl = [chunks[y].data_ptr() + offs[0], chunks[u].data_ptr() + offs[1], chunks[v].data_ptr() + offs[2]]
            return ctx.probe_encode_traffic(chunks[i].data_ptr(), n3, B, w, h, pl, st, psz, iters=iters)

        probe4(1, 0, 0, 0)                                       # warm-up (first touch of the code object)
        groups, fast, _ = find_groups(n, lambda i, r: probe4(i, r, r, r))
        sizes = [len(g) for g in groups] if groups else None
        self.stats["groups"] = sizes
        gy = None
        if groups and len(groups) >= 2:
            order = sorted(range(len(groups)), key=lambda g: sizes[g])
            fit = [g for g in order if sizes[g] >= n_y + 1]
            gy = fit[0] if fit else None
        if gy is None:
            # no contrast or no usable group: plain choice (the first chunks), reported as such
            self.stats["grouped"] = False
            self.float = chunks[:n_float]
            self.y = chunks[n_float:n_float + n_y]
            self.uv = chunks[n_float + n_y:n_float + n_y + n_uv]
        else:
            self.stats["grouped"] = True
            rest_groups = [g for g in sorted(range(len(groups)), key=lambda g: -sizes[g]) if g != gy]
            g_in = rest_groups[0]                                # the largest other group holds the reference float chunk
            g_uv = rest_groups[-1] if sizes[rest_groups[-1]] >= n_uv + 1 else g_in
            cref = groups[g_in][0]
            ycand = groups[gy][:max(n_y + 6, 12)]
            uvcand = [i for i in groups[g_uv] if i != cref][:max(n_uv + 5, 8)]
            # (3) planes chunks that run fastest with the reference float chunk
            ty = {k: probe4(cref, k, uvcand[0], uvcand[0]) for k in ycand}
            ysel = sorted(ycand, key=lambda k: ty[k])[:n_y]
            tu = {k: probe4(cref, ysel[0], k, k) for k in uvcand}
            uvsel = sorted(uvcand, key=lambda k: tu[k])[:n_uv]
            # (4) every other chunk as a float chunk against the chosen planes chunks
            taken = set(ysel) | set(uvsel)
            cand = [i for i in range(n) if i not in taken]
            tf = {i: probe4(i, ysel[0], uvsel[0], uvsel[0]) for i in cand}
            fsel = sorted(cand, key=lambda i: tf[i])[:n_float]
            self.y = [chunks[i] for i in ysel]
            self.uv = [chunks[i] for i in uvsel]
            self.float = [chunks[i] for i in fsel]
            same = groups[gy][:2]
            tsel = [tf[i] for i in fsel]
            self.stats["y_group"], self.stats["uv_group"] = gy, g_uv
            self.stats["probe_ms"] = {
                "input_and_planes_in_one_group": round(probe4(same[0], same[1], same[1], same[1]), 4),
                "planes_together_in_another_group": round(fast, 4),
                "float_chunks_kept_fastest": round(min(tsel), 4), "float_chunks_kept_median": round(statistics.median(tsel), 4),
                "float_chunks_kept_slowest": round(max(tsel), 4),
                "float_chunks_rejected_median": round(statistics.median([tf[i] for i in cand if i not in set(fsel)] or [0.0]), 4),
                "y_chunks_kept_slowest": round(max(ty[k] for k in ysel), 4), "uv_chunks_kept_slowest": round(max(tu[k] for k in uvsel), 4)}
        torch.cuda.synchronize(dev)
        self.stats["probes"] = nprobe[0]
        chunks = None
        torch.cuda.empty_cache()

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

    def give_back(self, floats, y, uv):
        self.float = list(floats) + self.float
        self.y = list(y) + self.y
        self.uv = list(uv) + self.uv

    def close(self):
        import torch
        self.float, self.y, self.uv = [], [], []
        torch.cuda.empty_cache()
