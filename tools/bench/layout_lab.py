#!/usr/bin/env python3
"""tools/bench/layout_lab.py -- which HBM region groups the streams of concurrent launches should live in (-> profiles/r03_layout_lab.txt).

The chunk pool classifies device memory into three region groups A, B, C.  For a 20-frame 4K launch this script composes
layouts per batch b (lane = b mod lanes inside an unordered section): the group of the float frames (input of encode, output of
decode -- packed, or R / G / B striped over the three groups), of the Y planes and of the U / V planes, and times K launches
ordered (one stream) and in unordered sections of 2 / 3 lanes with several per-lane grid sizes; ms per launch = hipEvent window / K.
"""
import os
import sys
import time

ROOT = os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
sys.path.insert(0, ROOT)
import numpy as np  # noqa: E402
import torch  # noqa: E402

import lumahdrv_amd as L  # noqa: E402
from lumahdrv_amd.placement import CHUNK_BYTES, HbmChunkPool, slots  # noqa: E402


def main():
    dev = torch.device("cuda:0")
    w, h, B, profile = 3840, 2160, 20, 2
    n1, n3 = w * h, 3 * w * h
    _, hs, st, _ = L.plane_geometry(w, h, profile)
    psz = [hs[p] * st[p] for p in range(3)]
    ctx = L.Context(0)
    ctx.set_stream(torch.cuda.current_stream().cuda_stream)
    ctx.set_quantizer(L.PTF_PQ, 11, L.CS_LUV, 8, 1e4, 0.005, L.build_lut(L.PTF_PQ, 11, 1e4, 0.005))
    NB = int(os.environ.get("LAB_BATCHES", "9"))
    K = int(os.environ.get("LAB_K", "27"))
    pool = HbmChunkPool(ctx, dev, 3, 1, 1, n_striped=int(os.environ.get("LAB_STRIPED", "30")))
    print("pool:", {k: pool.stats[k] for k in ("chunks", "groups", "grouped")}, flush=True)
    if not pool.stats.get("grouped"):
        print("no region groups on this box; nothing to compare")
        return
    G = [list(g) for g in pool.striped]          # chunks (uint8 tensors) by group
    nxt = [0, 0, 0]

    def take(g):
        t = G[g][nxt[g]]
        nxt[g] += 1
        return t

    _, yslot = slots(CHUNK_BYTES, B * psz[0])
    uvs = (B * psz[1] + (1 << 20) - 1) // (1 << 20) * (1 << 20)
    _, sslot = slots(CHUNK_BYTES, B * n1 * 4)

    class Layout:
        """per batch: input float chunk (encode) / output (decode), Y chunk, UV chunk"""

        def __init__(self, name, fgroup, ygroup, uvgroup, striped=False):
            self.name, self.striped = name, striped
            self.b = []
            for b in range(NB):
                fg, yg, ug = fgroup(b), ygroup(b), uvgroup(b)
                yc, uc = take(yg), take(ug)
                if striped:
                    f = [take(k).data_ptr() for k in range(3)]
                    fs = n1
                else:
                    fc = take(fg)
                    f = [fc.data_ptr() + k * n1 * 4 for k in range(3)]
                    fs = n3
                self.b.append((f, fs, [yc.data_ptr(), uc.data_ptr(), uc.data_ptr() + uvs]))

    def reset():
        for k in range(3):
            nxt[k] = 0

    def run(lay, direction, lanes, lane_grid=0):
        ctx.tune("lane_grid_enc", lane_grid)
        ctx.tune("lane_grid_dec", lane_grid)
        for b in range(NB):           # fill inputs / planes once (the same chunks are reused across layouts)
            f, fs, pl = lay.b[b]
            if direction == 0 and not lay.striped:
                ctx.synth_frames_device(f[0], n3, B, w, h, 7, b * B)

        def launch(i):
            f, fs, pl = lay.b[i % NB]
            if direction == 0:
                ctx.encode_frames_device_planar(f, fs, B, w, h, 1.0, profile, pl, st, psz)
            else:
                ctx.decode_frames_device_planar(pl, st, psz, B, w, h, profile, 1.0, f, fs)
        for i in range(3):
            launch(i)
        ts = []
        for rep in range(7):
            torch.cuda.synchronize()
            e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
            e0.record()
            if lanes:
                ctx.begin_unordered(lanes)
            for i in range(K):
                launch(i)
            if lanes:
                ctx.end_unordered()
            e1.record()
            torch.cuda.synchronize()
            ts.append(e0.elapsed_time(e1) / K)
        return float(np.median(ts))

    A, Bg, C = 0, 1, 2
    rot = lambda k: (lambda b: (b + k) % 3)          # noqa: E731
    const = lambda g: (lambda b: g)                  # noqa: E731
    enc_layouts = [
        ("in A, Y B, UV C (every batch)", const(A), const(Bg), const(C)),
        ("batch b: in g(b), Y g(b+1), UV g(b+2)", rot(0), rot(1), rot(2)),
        ("in A, Y A, UV A (one group)", const(A), const(A), const(A)),
    ]
    dec_layouts = [
        ("out A packed, Y B, UV C", const(A), const(Bg), const(C), False),
        ("out striped A/B/C, Y B, UV C", const(A), const(Bg), const(C), True),
        ("batch b: out g(b) packed, Y g(b+1), UV g(b+2)", rot(0), rot(1), rot(2), False),
        ("out striped A/B/C, Y A, UV A", const(A), const(A), const(A), True),
        ("out A, Y A, UV A (one group)", const(A), const(A), const(A), False),
    ]
    px = B * w * h
    for direction, lays, grids in ((0, enc_layouts, (640, 768, 896)), (1, dec_layouts, (1024, 1280, 1536, 2048))):
        print("\n== %s: ms per 20-frame launch (Gpixel/s, fraction of 8 TB/s) ==" % ("ENCODE" if direction == 0 else "DECODE"), flush=True)
        for spec in lays:
            reset()
            lay = Layout(spec[0], spec[1], spec[2], spec[3], *(spec[4:] or ()))
            row = []
            for lanes in (0, 2):
                for g in ((0,) if lanes == 0 else grids):
                    ms = run(lay, direction, lanes, g)
                    row.append("%s%s %.4f (%.0f, %.3f)" % ("ordered" if not lanes else "%d lanes" % lanes,
                                                           "" if not g else " x%d" % g, ms, px / ms / 1e6, 15.0 * px / (ms * 1e-3) / 8e12))
            print("  %-48s %s" % (spec[0], " | ".join(row)), flush=True)
    pool.close()


if __name__ == "__main__":
    main()
