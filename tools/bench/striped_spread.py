#!/usr/bin/env python3
"""tools/bench/striped_spread.py -- why did the decoded-planes-in-three-groups layout gain 11 % on one box and 1.2 % on another in
round 2 (profiles/r02_placement.txt sections 4 and 6)?  One box, one process, ordered single launches of the 4:2:0 16-bit
decode kernel over one 20-frame 4K batch, for many random choices of WHICH chunks of the groups carry the streams:
  packed : decoded frames in one chunk of group a, Y planes in a chunk of group b, U / V planes in a chunk of group c
  striped: R, G, B planes in chunks of groups a, b, c; Y and U / V planes as above
for every assignment (a, b, c) of the three region groups to the roles and several random chunk draws each.
-> profiles/r03_striped_spread.txt"""
import itertools
import os
import random
import sys

ROOT = os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
sys.path.insert(0, ROOT)
import numpy as np  # noqa: E402
import torch  # noqa: E402

import lumahdrv_amd as L  # noqa: E402
from lumahdrv_amd.placement import HbmChunkPool  # noqa: E402


def main():
    dev = torch.device("cuda:0")
    w, h, B, profile = 3840, 2160, 20, 2
    n1, n3 = w * h, 3 * w * h
    _, hs, st, _ = L.plane_geometry(w, h, profile)
    psz = [hs[p] * st[p] for p in range(3)]
    uvs = (B * psz[1] + (1 << 20) - 1) // (1 << 20) * (1 << 20)
    ctx = L.Context(0)
    ctx.set_stream(torch.cuda.current_stream().cuda_stream)
    ctx.set_quantizer(L.PTF_PQ, 11, L.CS_LUV, 8, 1e4, 0.005, L.build_lut(L.PTF_PQ, 11, 1e4, 0.005))
    pool = HbmChunkPool(ctx, dev, 1, 1, 1, n_striped=24)
    print("pool:", {k: pool.stats[k] for k in ("chunks", "groups", "grouped")}, flush=True)
    if not pool.stats.get("grouped"):
        print("no region groups on this box")
        return
    G = [list(g) for g in pool.striped]
    rng = random.Random(7)
    # real codes in every chunk that may serve as planes: encode one synthetic batch into scratch, copy around
    src = pool.float[0]
    ctx.synth_frames_device(src.data_ptr(), n3, B, w, h, 7, 0)

    def fill_planes(yc, uc):
        ctx.encode_frames_device(src.data_ptr(), n3, B, w, h, 1.0, profile, [yc.data_ptr(), uc.data_ptr(), uc.data_ptr() + uvs], st, psz)

    def time_decode(f, fs, pl):
        ts = []
        for _ in range(9):
            torch.cuda.synchronize()
            e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
            e0.record()
            ctx.decode_frames_device_planar(pl, st, psz, B, w, h, profile, 1.0, f, fs)
            e1.record()
            torch.cuda.synchronize()
            ts.append(e0.elapsed_time(e1))
        return float(np.median(ts[2:]))

    gains_all = []
    for (a, b, c) in (itertools.permutations(range(3)) if os.environ.get("SPREAD_SKIP_PERMS") != "1" else ()):
        gains = []
        row = []
        for trial in range(5):
            ca, cb, cc = rng.sample(G[a], 2), rng.sample(G[b], 2), rng.sample(G[c], 2)
            yc, uc = cb[1], cc[1]
            fill_planes(yc, uc)
            pl = [yc.data_ptr(), uc.data_ptr(), uc.data_ptr() + uvs]
            packed = time_decode([ca[0].data_ptr() + k * n1 * 4 for k in range(3)], n3, pl)
            striped = time_decode([ca[0].data_ptr(), cb[0].data_ptr(), cc[0].data_ptr()], n1, pl)
            gains.append(packed / striped - 1.0)
            row.append("%.4f/%.4f (%+.1f %%)" % (packed, striped, 100 * gains[-1]))
        gains_all += gains
        print("  out %s, Y %s, UV %s | R G B in %s %s %s:  %s" % ("ABC"[a], "ABC"[b], "ABC"[c], "ABC"[a], "ABC"[b], "ABC"[c], "  ".join(row)), flush=True)
    g = sorted(gains_all) or [0.0]
    print("gain of the striped over the packed decode output, %d draws: min %+.1f %%, median %+.1f %%, max %+.1f %%"
          % (len(g), 100 * g[0], 100 * g[len(g) // 2], 100 * g[-1]))
    # the same chunks, striped decode output, but the coded planes (a) all in ONE group, (b) spread A, B, C as in round 2's 11 % case
    print("coded planes: together in one group / in two groups / Y, U, V in three groups (striped output A B C):")
    for trial in range(4):
        ca, cb, cc = rng.sample(G[0], 3), rng.sample(G[1], 3), rng.sample(G[2], 3)
        out = [ca[0].data_ptr(), cb[0].data_ptr(), cc[0].data_ptr()]
        res = []
        for (yc, uc, vc) in ((ca[1], ca[2], ca[2]), (cb[1], cc[1], cc[1]), (ca[1], cb[1], cc[1])):
            ctx.encode_frames_device(src.data_ptr(), n3, B, w, h, 1.0, profile,
                                     [yc.data_ptr(), uc.data_ptr(), vc.data_ptr() + (uvs if vc is uc else 0)], st, psz)
            pl = [yc.data_ptr(), uc.data_ptr(), vc.data_ptr() + (uvs if vc is uc else 0)]
            res.append(time_decode(out, n1, pl))
        print("   %.4f / %.4f / %.4f ms" % tuple(res), flush=True)
    # round 2 measured +11 % BEFORE the 4:2:0 16-bit decode kernels got their 5-workgroups-per-CU rule (8 per CU then) and +1.2 %
    # after it: the same comparison under both launch geometries
    print("packed / striped decode output by persistent workgroups per CU (out A, Y B, UV C):")
    draws = [(rng.sample(G[0], 1)[0], rng.sample(G[1], 2), rng.sample(G[2], 2)) for _ in range(4)]
    for pcu in (8, 6, 5, 4):
        ctx.tune("blocks_per_cu", pcu)
        row = []
        for (ca, cb, cc) in draws:
            yc, uc = cb[1], cc[1]
            ctx.tune("blocks_per_cu", 0)
            fill_planes(yc, uc)
            ctx.tune("blocks_per_cu", pcu)
            pl = [yc.data_ptr(), uc.data_ptr(), uc.data_ptr() + uvs]
            packed = time_decode([ca.data_ptr() + k * n1 * 4 for k in range(3)], n3, pl)
            striped = time_decode([ca.data_ptr(), cb[0].data_ptr(), cc[0].data_ptr()], n1, pl)
            row.append("%.4f/%.4f (%+.1f %%)" % (packed, striped, 100 * (packed / striped - 1)))
        print("   %d per CU: %s" % (pcu, "  ".join(row)), flush=True)
    ctx.tune("blocks_per_cu", 0)
    pool.close()


if __name__ == "__main__":
    main()
