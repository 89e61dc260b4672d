#!/usr/bin/env python3
"""Host-fed (drop-in, PCIe-inclusive) throughput of lumahip_encode_frame_host / lumahip_decode_frame_host on one
3840x2160 frame: pageable numpy memory vs memory pinned with lumahip_host_register.  Reported in DESIGN.md;
never the bench's `value`."""
import ctypes as C
import os
import sys
import time

ROOT = os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
sys.path.insert(0, ROOT)
import numpy as np  # noqa: E402

import lumahdrv_amd as L  # noqa: E402
from lumahdrv_amd.capi import _arr3  # noqa: E402


def main():
    w, h = 3840, 2160
    ctx = L.Context(0)
    ctx.set_quantizer(L.PTF_PQ, 11, L.CS_LUV, 8, 1e4, 0.005, L.build_lut(L.PTF_PQ, 11))
    rng = np.random.default_rng(1)
    f = np.exp(rng.uniform(-6, 9, (3, h, w))).astype(np.float32)
    _, hs, st, _ = L.plane_geometry(w, h, 2)
    planes = [np.zeros((hs[p], st[p]), dtype=np.uint8) for p in range(3)]
    out = np.empty_like(f)
    pp = _arr3(C.c_void_p, [p.ctypes.data for p in planes])
    ss = _arr3(C.c_int, st)
    mean = C.c_float()

    def enc():
        ctx._chk(ctx.L.lumahip_encode_frame_host(ctx.h, f.ctypes.data, w, h, 1.0, 2, pp, ss, C.byref(mean), None))

    def dec():
        ctx._chk(ctx.L.lumahip_decode_frame_host(ctx.h, pp, ss, w, h, 2, 1.0, out.ctypes.data))

    def t(fn, n=8):
        fn()
        t0 = time.perf_counter()
        for _ in range(n):
            fn()
        return (time.perf_counter() - t0) / n

    nb = 12
    fr = [f.copy() for _ in range(nb)]
    outs = [np.empty_like(f) for _ in range(nb)]
    pls = [[np.zeros((hs[p], st[p]), dtype=np.uint8) for p in range(3)] for _ in range(nb)]
    fp = (C.c_void_p * nb)(*[a.ctypes.data for a in fr])
    op = (C.c_void_p * nb)(*[a.ctypes.data for a in outs])
    bp = (C.c_void_p * (3 * nb))(*[pl.ctypes.data for tri in pls for pl in tri])

    def encb():
        ctx._chk(ctx.L.lumahip_encode_frames_host(ctx.h, fp, nb, w, h, 1.0, 2, bp, ss, None))

    def decb():
        ctx._chk(ctx.L.lumahip_decode_frames_host(ctx.h, bp, ss, nb, w, h, 2, 1.0, op))

    for label in ("pageable", "pinned (lumahip_host_register)"):
        if label.startswith("pinned"):
            for a in [f, out] + planes + fr + outs + [pl for tri in pls for pl in tri]:
                ctx.host_register(a)
        te, td = t(enc), t(dec)
        tbe, tbd = t(encb, 3) / nb, t(decb, 3) / nb
        print("%-32s pipelined batch of %d: encode %.2f ms/frame = %.0f Mpixel/s | decode %.2f ms/frame = %.0f Mpixel/s"
              % (label, nb, tbe * 1e3, w * h / tbe / 1e6, tbd * 1e3, w * h / tbd / 1e6), flush=True)
        print("%-32s encode %.2f ms/frame = %.0f Mpixel/s (%.1f GB/s over PCIe) | decode %.2f ms/frame = %.0f Mpixel/s"
              % (label, te * 1e3, w * h / te / 1e6, 15.0 * w * h / te / 1e9, td * 1e3, w * h / td / 1e6), flush=True)


if __name__ == "__main__":
    main()
