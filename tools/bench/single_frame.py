#!/usr/bin/env python3
"""Kernel time of ONE device-resident frame per launch (what a per-frame caller sees once data is on the GPU), encode and
decode, for the launch-geometry candidates of short launches: workgroups per CU, workgroup size, and the luminance-search
records staged in LDS (the default) against read from global memory / L2 (no per-workgroup staging) -> profiles/r03_single_frame.txt"""
import os
import sys

ROOT = os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
sys.path.insert(0, ROOT)
import torch  # noqa: E402

import lumahdrv_amd as L  # noqa: E402


def main():
    dev = torch.device("cuda:0")
    sizes = ((640, 360), (1280, 720), (1920, 1080), (3840, 2160), (7680, 4320))
    for direction in (0, 1):
        print("== %s: median us per single-frame launch (hipEvent pair around one launch) ==" % ("encode" if direction == 0 else "decode"))
        for blk, pcu, ldskb in ((0, 0, -1), (0, 2, -1), (0, 4, -1), (512, 2, -1), (0, 0, 0), (0, 4, 0), (0, 2, 0)):
            ctx = L.Context(0)
            ctx.tune("block", blk)
            ctx.tune("blocks_per_cu", pcu)
            ctx.tune("lds_table_max_kb", ldskb)
            ctx.set_stream(torch.cuda.current_stream().cuda_stream)
            ctx.set_quantizer(L.PTF_PQ, 11, L.CS_LUV, 8, 1e4, 0.005, L.build_lut(L.PTF_PQ, 11))
            line = "block %4s wg/CU %s tables %s:" % (blk or "rule", pcu or "rule", "LDS" if ldskb < 0 else "global")
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
                        ms.append(ctx.time_launches(direction, 1, src.data_ptr() + b * n3 * 4, n3, 1, w, h, 1.0, 2, pl, st, psz))
                m = sorted(ms)[len(ms) // 2]
                line += "  %dx%d %.1f (%.0f Gpx/s)" % (w, h, m * 1e3, w * h / m / 1e6)
                del src, planes
            print(line, flush=True)
            ctx.close()


if __name__ == "__main__":
    main()
