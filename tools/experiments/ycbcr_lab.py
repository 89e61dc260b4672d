#!/usr/bin/env python3
"""tools/experiments/ycbcr_lab.py -- does the LDS traffic of the powf tables limit the YCbCr kernels?  (-> profiles/r03_ycbcr_lab.txt)
The wide log2 table (pow_glibc.hpp) is read with one ds_read_b128 per powf at an address that depends on the argument's binary
exponent and top four mantissa bits.  The benchmark's synthetic frames are log-uniform over 24 exponents -- the worst case for
bank conflicts.  This script times the same kernels on frames whose pixels (a) are the synthetic ones, (b) share ONE binary
exponent (random mantissas: 16 distinct table rows -> conflict-free), (c) are all equal (every lane reads the same address)."""
import os
import sys

ROOT = os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
sys.path.insert(0, ROOT)
import numpy as np  # noqa: E402
import torch  # noqa: E402

import lumahdrv_amd as L  # noqa: E402


def main():
    dev = torch.device("cuda:0")
    w, h, B, profile = 3840, 2160, 20, 2
    n3 = 3 * w * h
    _, hs, st, _ = L.plane_geometry(w, h, profile)
    psz = [hs[p] * st[p] for p in range(3)]
    ctx = L.Context(0)
    ctx.set_stream(torch.cuda.current_stream().cuda_stream)
    ctx.set_quantizer(L.PTF_PQ, 10, L.CS_YCBCR, 10, 1000.0, 0.01, L.build_lut(L.PTF_PQ, 10, 1000.0, 0.01))
    NB = 4
    planes = [torch.zeros(NB * B * psz[p], dtype=torch.uint8, device=dev) for p in range(3)]
    out = torch.empty(NB * B * n3, dtype=torch.float32, device=dev)
    src = torch.empty(NB * B * n3, dtype=torch.float32, device=dev)

    def fill(kind):
        if kind == "synthetic":
            ctx.synth_frames_device(src.data_ptr(), n3, NB * B, w, h)
        elif kind == "one exponent":
            src.uniform_(1.0, 2.0)
        else:
            src.fill_(1.337)
        torch.cuda.synchronize()

    def timeit(fn):
        for i in range(2):
            fn(i)
        ts = []
        for rep in range(5):
            torch.cuda.synchronize()
            e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
            e0.record()
            for i in range(8):
                fn(i)
            e1.record()
            torch.cuda.synchronize()
            ts.append(e0.elapsed_time(e1) / 8)
        return float(np.median(ts))

    px = B * w * h
    for kind in ("synthetic", "one exponent", "one value"):
        fill(kind)
        enc = timeit(lambda i: ctx.encode_frames_device(src.data_ptr() + (i % NB) * B * n3 * 4, n3, B, w, h, 20.0, profile,
                                                        [planes[p].data_ptr() + (i % NB) * B * psz[p] for p in range(3)], st, psz))
        dec = timeit(lambda i: ctx.decode_frames_device([planes[p].data_ptr() + (i % NB) * B * psz[p] for p in range(3)], st, psz, B, w, h,
                                                        profile, 20.0, out.data_ptr() + (i % NB) * B * n3 * 4, n3))
        print("%-14s encode %.4f ms (%.1f Gpx/s)   decode %.4f ms (%.1f Gpx/s)" % (kind, enc, px / enc / 1e6, dec, px / dec / 1e6), flush=True)


if __name__ == "__main__":
    main()
