#!/usr/bin/env python3
"""tools/experiments/decode_interleaved.py -- experiment: a batch of three "tall" frames (each 3840 x 12960 = six 4K frames stacked)
placed in three 2 GiB chunks of three different region groups (equally spaced addresses, so one frame stride reaches
them), decoded / encoded with the frame-sequential tile order and with the frame-interleaved one
(LUMAHIP_TILE_INTERLEAVE_DEC / _ENC), against the same batch packed into chunks of one group."""
import os
import statistics
import sys

ROOT = os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
sys.path.insert(0, ROOT)
os.environ["LUMAHIP_TUNING"] = "1"   # the LUMAHIP_* overrides are honoured only under this gate
os.environ["LUMAHIP_ALLOW_ALIASED_FRAMES"] = "1"
import torch  # noqa: E402

import lumahdrv_amd as L  # noqa: E402
from lumahdrv_amd.placement import CHUNK_BYTES, find_groups, plane_slots  # noqa: E402


def ctx_with(**env):
    for k, v in env.items():
        os.environ[k] = str(v)
    c = L.Context(0)
    for k in env:
        os.environ.pop(k, None)
    c.set_quantizer(L.PTF_PQ, 11, L.CS_LUV, 8, 1e4, 0.005, L.build_lut(L.PTF_PQ, 11, 1e4, 0.005))
    c.set_stream(torch.cuda.current_stream().cuda_stream)
    return c


def main():
    n = int(sys.argv[1]) if len(sys.argv) > 1 else 70
    w, h, profile = 3840, 2160 * 6, 2
    n3 = 3 * w * h
    dev = torch.device("cuda:0")
    _, hs, st, _ = L.plane_geometry(w, h, profile)
    psz = [hs[p] * st[p] for p in range(3)]
    assert n3 * 4 <= CHUNK_BYTES
    chunks = [torch.zeros(CHUNK_BYTES, dtype=torch.uint8, device=dev) for _ in range(n)]
    torch.cuda.synchronize()
    va = [c.data_ptr() for c in chunks]
    base = ctx_with()
    # groups, with the standard 4K probe
    w0, h0, B0 = 3840, 2160, 20
    _, hs0, st0, _ = L.plane_geometry(w0, h0, 2)
    psz0 = [hs0[p] * st0[p] for p in range(3)]
    _, _, offs0 = plane_slots(CHUNK_BYTES, [B0 * x for x in psz0])
    probe = lambda i, r: base.probe_encode_traffic(va[i], 3 * w0 * h0, B0, w0, h0, [va[r] + o for o in offs0], st0, psz0, iters=2)
    probe(1, 0)
    groups, _, _ = find_groups(n, probe)
    gid = {i: g for g, grp in enumerate(groups) for i in grp}
    print("group sizes", [len(g) for g in groups])
    trip = None
    for k in range(1, n // 2):
        for i in range(n - 2 * k):
            a, b, c = i + 2 * k, i + k, i
            if va[b] - va[a] == va[c] - va[b] and len({gid[a], gid[b], gid[c]}) == 3:
                trip = (a, b, c)
                break
        if trip:
            break
    same = None
    for k in range(1, n // 2):
        for i in range(n - 2 * k):
            a, b, c = i + 2 * k, i + k, i
            if va[b] - va[a] == va[c] - va[b] and len({gid[a], gid[b], gid[c]}) == 1 and not ({a, b, c} & set(trip)):
                same = (a, b, c)
                break
        if same:
            break
    print("frames in chunks", trip, "(groups", [gid[i] for i in trip], ") / in one group:", same, [gid[i] for i in same])
    used = set(trip) | set(same)
    free = [i for i in range(n) if i not in used]
    # planes: Y of the three frames in three chunks of the groups OTHER than the frame's, U / V likewise
    def planes_for(frames):
        pls = []
        for i in frames:
            g = gid[i]
            yc = next(j for j in free if gid[j] != g and j not in pls)
            pls.append(yc)
        return pls
    res = {}
    for label, frames in (("frames in three groups", trip), ("frames in one group", same)):
        stride_f = (va[frames[1]] - va[frames[0]]) // 4
        pchunks = planes_for(frames)
        # planes of frame f at plane base + f * pfs: needs equally spaced plane chunks too -> put all planes in ONE chunk of a
        # group different from ... (simplification: one planes chunk per launch, in a group that is none of frame 0's)
        pc = next(j for j in free if gid[j] != gid[frames[0]])
        yo, uo, vo = 0, 3 * psz[0] + (4 << 20), 3 * psz[0] + 3 * psz[1] + (8 << 20)
        pl = [va[pc] + yo, va[pc] + uo, va[pc] + vo]
        base.synth_frames_device(va[frames[0]], stride_f, 3, w, h, 11, 0)
        base.encode_frames_device(va[frames[0]], stride_f, 3, w, h, 1.0, profile, pl, st, psz)
        torch.cuda.synchronize()
        for mode, env in (("sequential", {}), ("interleaved", {"LUMAHIP_TILE_INTERLEAVE_DEC": 1, "LUMAHIP_TILE_INTERLEAVE_ENC": 1})):
            c = ctx_with(**env)
            td = [c.time_launches(1, 3, va[frames[0]], stride_f, 3, w, h, 1.0, profile, pl, st, psz) for _ in range(7)]
            te = [c.time_launches(0, 3, va[frames[0]], stride_f, 3, w, h, 1.0, profile, pl, st, psz) for _ in range(7)]
            res[(label, mode)] = (statistics.median(td), statistics.median(te))
            c.set_stream(None)
            c.close()
    px = 3 * w * h
    for (label, mode), (td, te) in res.items():
        print("%-24s %-12s decode %.4f ms (%.0f Gpixel/s)   encode %.4f ms (%.0f Gpixel/s)" % (label, mode, td, px / td / 1e6, te, px / te / 1e6))


if __name__ == "__main__":
    main()
