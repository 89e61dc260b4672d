#!/usr/bin/env python3
"""Sweep launch geometry (threads per workgroup x workgroups per CU) for the encode / decode kernels on
device-resident synthetic 4K frames; prints avg kernel ms, Gpixel/s and fraction of the 8 TB/s roofline."""
import itertools
import os
import sys

ROOT = os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
sys.path.insert(0, ROOT)
import torch  # noqa: E402

import lumahdrv_amd as L  # noqa: E402


def main():
    w, h, B, nb = 3840, 2160, 20, 4
    n3 = 3 * w * h
    dev = torch.device("cuda:0")
    _, hs, st, _ = L.plane_geometry(w, h, 2)
    psz = [hs[p] * st[p] for p in range(3)]
    src = torch.empty(nb * B * n3, dtype=torch.float32, device=dev)
    planes = [torch.zeros(nb * B * psz[p], dtype=torch.uint8, device=dev) for p in range(3)]
    lut = L.build_lut(L.PTF_PQ, 11)
    blocks = [int(x) for x in os.environ.get("TUNE_BLOCKS", "256,512,1024").split(",")]
    percu = [int(x) for x in os.environ.get("TUNE_PERCU", "0,1,2,3,4,6,8").split(",")]
    first = True
    for bt, pc in itertools.product(blocks, percu):
        if pc * bt > 2048:
            continue
        os.environ["LUMAHIP_TUNING"] = "1"   # the LUMAHIP_* overrides are honoured only under this gate
        os.environ["LUMAHIP_BLOCK"] = str(bt)
        os.environ["LUMAHIP_BLOCKS_PER_CU"] = str(pc)
        ctx = L.Context(0)
        ctx.set_stream(torch.cuda.current_stream().cuda_stream)
        ctx.set_quantizer(L.PTF_PQ, 11, L.CS_LUV, 8, 1e4, 0.005, lut)
        if first:
            ctx.synth_frames_device(src.data_ptr(), n3, nb * B, w, h)
            first = False
        res = []
        for d in (0, 1):
            ms = []
            for rep in range(3):
                for b in range(nb):
                    ms.append(ctx.time_launches(d, 1, src.data_ptr() + b * B * n3 * 4, n3, B, w, h, 1.0, 2,
                                                [planes[p].data_ptr() + b * B * psz[p] for p in range(3)], st, psz))
            ms = sorted(ms)[len(ms) // 2]
            res.append((ms, B * w * h / ms / 1e6, 15 * B * w * h / ms / 1e6 / 8000))
        print("block %4d  wg/CU %d : enc %.4f ms %.1f Gpx/s %.3f | dec %.4f ms %.1f Gpx/s %.3f" %
              (bt, pc, res[0][0], res[0][1], res[0][2], res[1][0], res[1][1], res[1][2]), flush=True)
        ctx.close()


if __name__ == "__main__":
    main()
