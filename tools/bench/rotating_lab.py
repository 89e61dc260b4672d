#!/usr/bin/env python3
"""tools/bench/rotating_lab.py -- packed frames of a batch rotating over three chunks (lumahip_decode_frames_device_rotating) against the
batch in one chunk, ordered (one launch at a time) and in two lanes, interleaved in one process.  (Round 5 also ran the mirror-image
ENCODE entry point through this tool -- input frames rotating over three chunks -- found it slower in every arrangement and removed
it again: profiles/r05_rotating_lab.txt has both tables.)
Input / output frames: (a) batch in ONE float chunk of the pool (the layout of bench.py's `value`); (b) frames rotating over three
chunks of three region groups (striped chunks); (c) frames rotating over three float chunks of ONE group.  Planes in Y / UV chunks
throughout.  20 x 3840x2160 per launch, PQ-11 Lu'v'.  -> profiles/r05_rotating_lab.txt"""
import os
import sys

ROOT = os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
sys.path.insert(0, ROOT)
import numpy as np  # noqa: E402
import torch  # noqa: E402

import lumahdrv_amd as L  # noqa: E402
from lumahdrv_amd.placement import HbmChunkPool  # noqa: E402


def main():
    dev = torch.device("cuda:0")
    ctx = L.Context(0)
    ctx.set_stream(torch.cuda.current_stream().cuda_stream)
    ctx.set_quantizer(L.PTF_PQ, 11, L.CS_LUV, 8, 1e4, 0.005, L.build_lut(L.PTF_PQ, 11, 1e4, 0.005))
    w, h, B, nb = 3840, 2160, 20, 6
    n1, n3 = w * h, 3 * w * h
    _, hs, st, _ = L.plane_geometry(w, h, 2)
    psz = [hs[p] * st[p] for p in range(3)]
    pool = HbmChunkPool(ctx, dev, n_float=2 * nb, n_y=2, n_uv=1, n_striped=2)
    print("pool:", pool.stats.get("groups"), "grouped", pool.stats.get("grouped"))
    one = pool.take_float(nb)                                  # (a)
    same = pool.take_float(nb)                                 # (c): six chunks of the float group(s), used three at a time
    strp = pool.take_striped(2)                                # (b): [[g0 x 2], [g1 x 2], [g2 x 2]]
    y_c, uv_c = pool.take_y(2), pool.take_uv(1)
    per = -(-B // 3)

    def planes(b):
        return [y_c[b // 5].data_ptr() + (b % 5) * (384 << 20), uv_c[0].data_ptr() + b * (192 << 20), uv_c[0].data_ptr() + b * (192 << 20) + (88 << 20)]

    def bases(kind, b):
        if kind == "three_groups":
            return [strp[g][b // 3].data_ptr() + (b % 3) * per * n3 * 4 for g in range(3)]
        return [same[3 * (b // 3) + g].data_ptr() + (b % 3) * per * n3 * 4 for g in range(3)]

    # the same synthetic frames in every layout
    for b in range(nb):
        ctx.synth_frames_device(one[b].data_ptr(), n3, B, w, h, 20250929, b * B)
        for kind in ("three_groups", "one_group"):
            bs = bases(kind, b)
            for f in range(B):
                dst = bs[f % 3] + (f // 3) * n3 * 4
                ctx.synth_frames_device(dst, n3, 1, w, h, 20250929, b * B + f)
    torch.cuda.synchronize()

    def enc(kind, b):
        if kind == "one_chunk":
            ctx.encode_frames_device(one[b].data_ptr(), n3, B, w, h, 1.0, 2, planes(b), st, psz)
        else:      # (no rotating encode entry point: see the docstring) -- the frames of buffer g, every third plane slot
            for g, base in enumerate(bases(kind, b)):
                n = len(range(g, B, 3))
                pl = [p + g * psz[i] for i, p in enumerate(planes(b))]
                ctx.encode_frames_device(base, n3, n, w, h, 1.0, 2, pl, st, [3 * x for x in psz])

    def dec(kind, b):
        if kind == "one_chunk":
            ctx.decode_frames_device(planes(b), st, psz, B, w, h, 2, 1.0, one[b].data_ptr(), n3)
        else:
            ctx.decode_frames_device_rotating(planes(b), st, psz, B, w, h, 2, 1.0, bases(kind, b), n3)

    def timed(fn, kind, lanes, reps=4):
        e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
        out = []
        for _ in range(reps):
            torch.cuda.synchronize()
            e0.record()
            if lanes:
                ctx.begin_unordered(lanes)
            for k in range(3 * nb):
                fn(kind, k % nb)
            if lanes:
                ctx.end_unordered()
            e1.record()
            torch.cuda.synchronize()
            out.append(e0.elapsed_time(e1) / (3 * nb))
        return float(np.median(out))

    ref = None
    print("direction lanes | one_chunk | three_groups | one_group   (ms per 20 x 4K launch; fraction of 8 TB/s)")
    for name, fn in (("encode", enc), ("decode", dec)):
        for lanes in (0, 2):
            row = []
            for rep in range(2):                                    # interleaved: two rounds, best of the medians
                for kind in ("one_chunk", "three_groups", "one_group"):
                    t = timed(fn, kind, lanes)
                    row.append((kind, t))
            best = {k: min(t for kk, t in row if kk == k) for k in ("one_chunk", "three_groups", "one_group")}
            print("%-6s %d | " % (name, lanes) + " | ".join("%.4f (%.3f)" % (best[k], 15.0 * B * n1 / (best[k] * 1e-3) / 8e12) for k in best), flush=True)
        if name == "encode":       # the planes of every layout's last encode are the same bytes
            torch.cuda.synchronize()
    pool.close()


if __name__ == "__main__":
    main()
