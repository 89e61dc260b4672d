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


def choose_roles(group_sizes, n_other, n_y):
    """which group holds the Y planes: the smallest group that has n_y chunks while the OTHER groups together have n_other.
    Returns the group index or None."""
    total = sum(group_sizes)
    best = None
    for g, sz in enumerate(group_sizes):
        if sz >= n_y and total - sz >= n_other and (best is None or sz < group_sizes[best]):
            best = g
    return best


def slots(chunk_bytes, nbytes, align=64 << 20):
    """(how many buffers of nbytes fit one chunk, the slot size)"""
    slot = (nbytes + align - 1) // align * align
    return (chunk_bytes // slot if slot <= chunk_bytes else 0), slot


class HbmChunkPool:
    """Takes the free device memory of `dev` in chunks of CHUNK_BYTES (leaving `keep_free` bytes), finds the groups with the
    traffic-only launch of `ctx` (lumahip_probe_encode_traffic_device: the loads and stores of the 4:2:0 encode kernel, no
    arithmetic), keeps `n_y` chunks of one group for Y planes and `n_other` chunks of the other groups for everything else,
    and returns the rest to the driver.  `ctx` needs a quantizer set; probing overwrites the chunks."""

    def __init__(self, ctx, dev, n_other, n_y, keep_free=6 << 30, iters=2):
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
        self.stats = {"chunk_GiB": CHUNK_BYTES / 2 ** 30, "chunks": n, "y_chunks": n_y, "other_chunks": n_other}
        self.other, self.y = [], []
        if n < n_other + n_y or n < 3:
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

        def probe4(i, y, u, v):
            pl = [y.data_ptr() + offs[0], u.data_ptr() + offs[1], v.data_ptr() + offs[2]]
            return ctx.probe_encode_traffic(i.data_ptr(), n3, B, w, h, pl, st, psz, iters=iters)

        def probe(i, r):                                        # reads chunk i, writes all three planes into chunk r
            return probe4(chunks[i], chunks[r], chunks[r], chunks[r])

        probe(1, 0)                                              # warm-up (first touch of the code object)
        groups, fast, probes = find_groups(n, probe)
        torch.cuda.synchronize(dev)
        self.stats["probes"] = probes
        gy = choose_roles([len(g) for g in groups], n_other, n_y) if groups else None
        if gy is None:
            # no contrast, or no group layout that fits: plain choice (the first chunks), reported as such
            self.stats["grouped"] = False
            self.stats["groups"] = [len(g) for g in groups] if groups else None
            self.other, self.y = chunks[:n_other], chunks[n_other:n_other + n_y]
        else:
            self.stats["grouped"] = True
            self.stats["groups"] = [len(g) for g in groups]
            self.stats["y_group"] = gy
            self.y = [chunks[i] for i in groups[gy][:n_y]]
            rest = [i for g, grp in enumerate(groups) if g != gy for i in grp]
            self.other = [chunks[i] for i in rest[:n_other]]
            # the three layouts of the module docstring, measured on THIS box with the chunks just chosen
            o0, o1 = self.other[0], self.other[-1]
            same = [chunks[i] for i in groups[gy][:2]] if len(groups[gy]) >= 2 else None
            self.stats["probe_ms"] = {
                "input_and_planes_in_one_group": round(probe4(same[0], same[1], same[1], same[1]), 4) if same else None,
                "planes_together_in_another_group": round(fast, 4),
                "chosen_layout_y_apart": round(probe4(o0, self.y[0], o1, o1), 4)}
        chunks = None
        torch.cuda.empty_cache()

    def take_other(self, n):
        if n > len(self.other):
            raise RuntimeError("HbmChunkPool: %d chunks wanted, %d left" % (n, len(self.other)))
        got, self.other = self.other[:n], self.other[n:]
        return got

    def take_y(self, n):
        if n > len(self.y):
            raise RuntimeError("HbmChunkPool: %d Y chunks wanted, %d left" % (n, len(self.y)))
        got, self.y = self.y[:n], self.y[n:]
        return got

    def give_back(self, other, y):
        self.other = list(other) + self.other
        self.y = list(y) + self.y

    def close(self):
        import torch
        self.other, self.y = [], []
        torch.cuda.empty_cache()
