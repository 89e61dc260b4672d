#!/usr/bin/env python3
"""tools/experiments/placement_pairs.py [n] -- traffic-only probe for every (input chunk i, planes chunk j) pair of the first n
2 GiB chunks: is the rate a property of a region or of the PAIR of regions read and written together?"""
import os
import sys

ROOT = os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
sys.path.insert(0, ROOT)
import torch  # noqa: E402

import lumahdrv_amd as L  # noqa: E402
from lumahdrv_amd.placement import CHUNK_BYTES, plane_slots  # noqa: E402


def main():
    n = int(sys.argv[1]) if len(sys.argv) > 1 else 24
    w, h, B, profile = 3840, 2160, 20, 2
    n3 = 3 * w * h
    dev = torch.device("cuda:0")
    _, hs, st, _ = L.plane_geometry(w, h, profile)
    psz = [hs[p] * st[p] for p in range(3)]
    _, _, offs = plane_slots(CHUNK_BYTES, [B * x for x in psz])
    ctx = L.Context(0)
    ctx.set_quantizer(L.PTF_PQ, 11, L.CS_LUV, 8, 1e4, 0.005, L.build_lut(L.PTF_PQ, 11, 1e4, 0.005))
    ctx.set_stream(torch.cuda.current_stream().cuda_stream)
    chunks = [torch.zeros(CHUNK_BYTES, dtype=torch.uint8, device=dev) for _ in range(n)]
    torch.cuda.synchronize()
    ctx.probe_encode_traffic(chunks[0].data_ptr(), n3, B, w, h, [chunks[1].data_ptr() + o for o in offs], st, psz, iters=2)
    print("rows: input chunk i; columns: planes chunk j; us per launch (i == j: planes in the slack behind the input)")
    for i in range(n):
        row = []
        for j in range(n):
            if i == j:
                row.append("  . ")
                continue
            pl = [chunks[j].data_ptr() + o for o in offs]
            row.append("%4.0f" % (1000 * ctx.probe_encode_traffic(chunks[i].data_ptr(), n3, B, w, h, pl, st, psz, iters=2)))
        print("%2d: %s" % (i, " ".join(row)), flush=True)


if __name__ == "__main__":
    main()
