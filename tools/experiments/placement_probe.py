#!/usr/bin/env python3
"""tools/experiments/placement_probe.py [trials] -- does the traffic-only probe's rate depend on WHERE the buffers were allocated?
One process; per trial: allocate the 500-frame stream's buffers anew (previous ones freed, allocator cache emptied, and a
differently sized spacer allocation in front to shift the placement), fill them, time the traffic-only probe and the
encode kernel over all 25 batches (isolated launches), print both."""
import os
import sys

ROOT = os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
sys.path.insert(0, ROOT)
import statistics  # noqa: E402

import torch  # noqa: E402

import lumahdrv_amd as L  # noqa: E402


def main():
    trials = int(sys.argv[1]) if len(sys.argv) > 1 else 6
    w, h, B, nb, profile = 3840, 2160, 20, 25, 2
    n3 = 3 * w * h
    dev = torch.device("cuda:0")
    _, hs, st, _ = L.plane_geometry(w, h, profile)
    psz = [hs[p] * st[p] for p in range(3)]
    ctx = L.Context(0)
    ctx.set_quantizer(L.PTF_PQ, 11, L.CS_LUV, 8, 1e4, 0.005, L.build_lut(L.PTF_PQ, 11, 1e4, 0.005))
    ctx.set_stream(torch.cuda.current_stream().cuda_stream)
    for t in range(trials):
        spacer = torch.empty((t * 3 + 1) * (1 << 28), dtype=torch.uint8, device=dev)   # 0.25, 1, 1.75 ... GB
        src = torch.empty(nb * B * n3, dtype=torch.float32, device=dev)
        planes = [torch.zeros(nb * B * psz[p], dtype=torch.uint8, device=dev) for p in range(3)]
        for b in range(nb):
            ctx.synth_frames_device(src.data_ptr() + b * B * n3 * 4, n3, B, w, h, 20250929, b * B)
        torch.cuda.synchronize()
        pr, en = [], []
        for rep in range(2):
            for b in range(nb):
                pl = [planes[p].data_ptr() + b * B * psz[p] for p in range(3)]
                s = src.data_ptr() + b * B * n3 * 4
                en.append(ctx.time_launches(0, 1, s, n3, B, w, h, 1.0, profile, pl, st, psz))
                pr.append(ctx.probe_encode_traffic(s, n3, B, w, h, pl, st, psz))
        print("trial %d  src @ 0x%x  probe median %.4f ms (min %.4f max %.4f)  encode median %.4f ms" % (
            t, src.data_ptr(), statistics.median(pr), min(pr), max(pr), statistics.median(en)), flush=True)
        del src, planes, spacer
        torch.cuda.empty_cache()


if __name__ == "__main__":
    main()
