#!/usr/bin/env python3
"""tools/experiments/placement_batches.py -- traffic-only probe per 2 GB batch of one resident 500-frame allocation, three passes:
are the slow batches the same ones every pass (a property of where the memory is) or random?"""
import os
import sys

ROOT = os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
sys.path.insert(0, ROOT)
import torch  # noqa: E402

import lumahdrv_amd as L  # noqa: E402


def main():
    w, h, B, nb, profile = 3840, 2160, 20, 25, 2
    n3 = 3 * w * h
    dev = torch.device("cuda:0")
    _, hs, st, _ = L.plane_geometry(w, h, profile)
    psz = [hs[p] * st[p] for p in range(3)]
    ctx = L.Context(0)
    ctx.set_quantizer(L.PTF_PQ, 11, L.CS_LUV, 8, 1e4, 0.005, L.build_lut(L.PTF_PQ, 11, 1e4, 0.005))
    ctx.set_stream(torch.cuda.current_stream().cuda_stream)
    src = torch.empty(nb * B * n3, dtype=torch.float32, device=dev)
    planes = [torch.zeros(nb * B * psz[p], dtype=torch.uint8, device=dev) for p in range(3)]
    for b in range(nb):
        ctx.synth_frames_device(src.data_ptr() + b * B * n3 * 4, n3, B, w, h, 20250929, b * B)
    torch.cuda.synchronize()
    mode = sys.argv[1] if len(sys.argv) > 1 else "same"
    for p_ in range(3):
        row = []
        for b in range(nb):
            # "same": batch b of the input with batch b of the planes; "fixedplanes": every input batch writes plane batch 0;
            # "fixedsrc": input batch 0 writes every plane batch
            sb = 0 if mode == "fixedsrc" else b
            pb = 0 if mode == "fixedplanes" else b
            pl = [planes[p].data_ptr() + pb * B * psz[p] for p in range(3)]
            row.append(ctx.probe_encode_traffic(src.data_ptr() + sb * B * n3 * 4, n3, B, w, h, pl, st, psz, iters=3))
        print(mode, "pass %d:" % p_, " ".join("%.3f" % v for v in row), flush=True)


if __name__ == "__main__":
    main()
