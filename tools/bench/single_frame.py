#!/usr/bin/env python3
"""Kernel time of ONE device-resident frame per launch (what a per-frame caller of LumaEncoder::encode / LumaDecoder::decode
sees once data is on the GPU), encode and decode, against the launch-geometry candidates of short launches: workgroup size,
workgroups per CU.  Every figure is the median over 72 launches (24
distinct frames x 3) of a hipEvent pair around ONE launch; the same pair around an EMPTY kernel is printed first -- that floor
is part of every figure and is not the kernels'.  With LUMAHIP_LIB=<another build> the same table for that build.
-> profiles/r05_single_frame.txt (whose two experimental builds -- first loads before the staging barrier, even shares of tiles per
workgroup -- were slower and are gone)

usage: single_frame.py [rule|sweep] [cfg]     cfg: pq11_luv (default) | pq10_ycbcr | log12_luv"""
import os
import sys

ROOT = os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
sys.path.insert(0, ROOT)
import torch  # noqa: E402

import lumahdrv_amd as L  # noqa: E402

CFGS = {"pq11_luv": (L.PTF_PQ, 11, L.CS_LUV, 8, 1e4, 0.005, 1.0), "pq10_ycbcr": (L.PTF_PQ, 10, L.CS_YCBCR, 10, 1000.0, 0.01, 20.0),
        "log12_luv": (L.PTF_LOG, 12, L.CS_LUV, 8, 1e4, 0.005, 1.0)}


def main():
    mode = sys.argv[1] if len(sys.argv) > 1 else "rule"
    cfg = CFGS[sys.argv[2] if len(sys.argv) > 2 else "pq11_luv"]
    dev = torch.device("cuda:0")
    sizes = ((640, 360), (1280, 720), (1920, 1080), (3840, 2160), (7680, 4320))
    print("library: %s   config: %s" % (os.environ.get("LUMAHIP_LIB", "lumahdrv_amd/lib (default build)"), sys.argv[2] if len(sys.argv) > 2 else "pq11_luv"))
    if mode == "rule":
        combos = ((0, 0),)
    else:
        combos = [(0, 0)] + [(b, p) for b in (256, 512, 1024) for p in (1, 2, 3, 4, 6) if b * p <= 2048 or p <= 2]
    for direction in (0, 1):
        print("== %s: median us per single-frame launch (hipEvent pair around one launch) ==" % ("encode" if direction == 0 else "decode"))
        for blk, pcu in combos:
            ctx = L.Context(0)
            ctx.tune("block", blk)
            ctx.tune("blocks_per_cu", pcu)
            ctx.set_stream(torch.cuda.current_stream().cuda_stream)
            ctx.set_quantizer(cfg[0], cfg[1], cfg[2], cfg[3], cfg[4], cfg[5], L.build_lut(cfg[0], cfg[1], cfg[4], cfg[5]))
            if direction == 0 and (blk, pcu) == combos[0]:
                dummy = torch.zeros(64, dtype=torch.uint8, device=dev)
                fl = sorted(ctx.time_launches(2, 1, 0, 0, 1, 64, 2, 1.0, 2, [dummy.data_ptr()] * 3, [64, 32, 32], [0, 0, 0]) for _ in range(200))
                print("floor: an empty kernel between the same two events: median %.1f us, min %.1f us" % (fl[100] * 1e3, fl[0] * 1e3))
            line = "block %4s wg/CU %4s:" % (blk or "rule", pcu or "rule")
            for (w, h) in sizes:
                n3 = 3 * w * h
                nb = 24
                _, hs, st, _ = L.plane_geometry(w, h, 2)
                psz = [hs[p] * st[p] for p in range(3)]
                src = torch.empty(nb * n3, dtype=torch.float32, device=dev)
                planes = [torch.zeros(nb * psz[p], dtype=torch.uint8, device=dev) for p in range(3)]
                ctx.synth_frames_device(src.data_ptr(), n3, nb, w, h)
                ms = []
                for rep in range(3):
                    for b in range(nb):
                        pl = [planes[p].data_ptr() + b * psz[p] for p in range(3)]
                        ms.append(ctx.time_launches(direction, 1, src.data_ptr() + b * n3 * 4, n3, 1, w, h, cfg[6], 2, pl, st, psz))
                m = sorted(ms)[len(ms) // 2]
                line += "  %dx%d %.1f (%.0f Gpx/s)" % (w, h, m * 1e3, w * h / m / 1e6)
                del src, planes
            print(line, flush=True)
            ctx.close()


if __name__ == "__main__":
    main()
