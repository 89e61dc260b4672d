#!/usr/bin/env python3
"""tools/bench/placed_grid_sweep.py -- persistent workgroups per CU of the headline kernels WITH the buffers placed by the chunk pool
(input in group A, Y planes in B, U / V planes in C; decoded R, G, B planes striped over A, B, C), ordered and in two lanes.
The launch rules of lumahip_launch.hip were first found on plainly allocated buffers; this checks them on the layout bench.py
and a resident-stream caller actually use.  20 x 3840x2160 PQ-11 Lu'v' per launch, K = 24 launches over 8 distinct batches,
median of 7 windows, all settings interleaved in one process.  -> profiles/r03_placed_grid_sweep.txt"""
import os
import sys

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
    NB, K = 8, 24
    ypc, yslot = slots(CHUNK_BYTES, B * psz[0])
    uvpc, uvslot = slots(CHUNK_BYTES, B * psz[1] + (1 << 20) + B * psz[2])
    spc, sslot = slots(CHUNK_BYTES, B * n1 * 4)
    pool = HbmChunkPool(ctx, dev, NB, -(-NB // ypc), -(-NB // uvpc), -(-NB // spc))
    print("pool:", {k: pool.stats[k] for k in ("chunks", "groups", "grouped")}, flush=True)
    src = pool.take_float(NB)
    yc, uc, sc3 = pool.take_y(-(-NB // ypc)), pool.take_uv(-(-NB // uvpc)), pool.take_striped(-(-NB // spc))
    vo = (B * psz[1] + (1 << 20) - 1) // (1 << 20) * (1 << 20)

    def ptrs(b):
        u = uc[b // uvpc].data_ptr() + (b % uvpc) * uvslot
        return (src[b].data_ptr(), [sc3[k][b // spc].data_ptr() + (b % spc) * sslot for k in range(3)],
                [yc[b // ypc].data_ptr() + (b % ypc) * yslot, u, u + vo])
    for b in range(NB):
        ctx.synth_frames_device(ptrs(b)[0], n3, B, w, h, 7, b * B)
        ctx.encode_frames_device(ptrs(b)[0], n3, B, w, h, 1.0, profile, ptrs(b)[2], st, psz)
    torch.cuda.synchronize()
    cands = (0, 2, 3, 4, 5, 6, 8)
    px = B * w * h
    for direction in (0, 1):
        def launch(i):
            s, o, pl = ptrs(i % NB)
            if direction == 0:
                ctx.encode_frames_device(s, n3, B, w, h, 1.0, profile, pl, st, psz)
            else:
                ctx.decode_frames_device_planar(pl, st, psz, B, w, h, profile, 1.0, o, n1)
        res = {(lanes, c): [] for lanes in (0, 2) for c in cands}
        for rep in range(8):
            for lanes in (0, 2):
                for c in cands:
                    ctx.tune("blocks_per_cu", c)
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
                    if rep:
                        res[(lanes, c)].append(e0.elapsed_time(e1) / K)
        ctx.tune("blocks_per_cu", 0)
        print("== %s: ms per launch (fraction of 8 TB/s); 'rule' = what lumahip_launch.hip picks ==" % ("ENCODE (in A, Y B, UV C)" if direction == 0 else "DECODE (R G B striped A B C, Y B, UV C)"))
        for lanes in (0, 2):
            row = []
            for c in cands:
                ms = float(np.median(res[(lanes, c)]))
                row.append("%s %.4f (%.3f)" % ("rule" if c == 0 else "%d/CU" % c, ms, 15.0 * px / (ms * 1e-3) / 8e12))
            print("  %-8s %s" % ("ordered" if not lanes else "%d lanes" % lanes, " | ".join(row)), flush=True)
    pool.close()


if __name__ == "__main__":
    main()
