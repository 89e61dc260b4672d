#!/usr/bin/env python3
"""tools/experiments/placement_fine.py -- traffic-only probe over 2-frame (200 MB) windows stepping through ONE 48 GiB allocation
(planes fixed): at what granularity does the memory speed change?"""
import os
import sys

ROOT = os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
sys.path.insert(0, ROOT)
import torch  # noqa: E402

import lumahdrv_amd as L  # noqa: E402


def main():
    w, h, B, profile = 3840, 2160, 2, 2
    n3 = 3 * w * h
    dev = torch.device("cuda:0")
    _, hs, st, _ = L.plane_geometry(w, h, profile)
    psz = [hs[p] * st[p] for p in range(3)]
    ctx = L.Context(0)
    ctx.set_quantizer(L.PTF_PQ, 11, L.CS_LUV, 8, 1e4, 0.005, L.build_lut(L.PTF_PQ, 11, 1e4, 0.005))
    ctx.set_stream(torch.cuda.current_stream().cuda_stream)
    nwin = 240
    big = torch.zeros(nwin * B * n3, dtype=torch.float32, device=dev)
    planes = [torch.zeros(B * psz[p], dtype=torch.uint8, device=dev) for p in range(3)]
    pl = [p.data_ptr() for p in planes]
    print("big @ 0x%x (%.1f GiB)" % (big.data_ptr(), big.numel() * 4 / 2**30))
    for p_ in range(2):
        row = []
        for i in range(nwin):
            row.append(ctx.probe_encode_traffic(big.data_ptr() + i * B * n3 * 4, n3, B, w, h, pl, st, psz, iters=8))
        print("pass %d (us):" % p_, " ".join("%.0f" % (v * 1000) for v in row), flush=True)


if __name__ == "__main__":
    main()
