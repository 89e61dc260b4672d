#!/usr/bin/env python3
"""tools/bench/half_miss_rate.py -- what the half-input YCbCr encode kernel costs when its inputs are NOT binary16 values.
20 x 3840x2160 frames per launch, HDR10 recipe; a fraction p of the pixels gets random low mantissa bits in all three channels
(no longer halves), so the lanes that own them take the general path inside the launch (luma_device.hpp ycbcr_fwd_half_n).
Median us per launch with the table always (lumahip_tune half_table 2), with half_table 0 (PQenc per pixel for every input),
and the MEAN of 24 consecutive launches in the default mode (half_table 1: launches that report float data send the next 16,
32, ... to the per-pixel kernels), interleaved in one process on the same buffers.  -> profiles/r04_half_miss_rate.txt"""
import os
import sys

ROOT = os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
sys.path.insert(0, ROOT)
import torch  # noqa: E402

import lumahdrv_amd as L  # noqa: E402


def main():
    dev = torch.device("cuda:0")
    ptf, bits, cs, bitsC, mx, mn, sc = L.PTF_PQ, 10, L.CS_YCBCR, 10, 1000.0, 0.01, 20.0
    lut = L.build_lut(ptf, bits, mx, mn)
    ctxs = {}
    for key in ("table", "off", "auto"):
        c = L.Context(0)
        c.tune("half_table", {"table": 2, "off": 0, "auto": 1}[key])
        c.set_stream(torch.cuda.current_stream().cuda_stream)
        c.set_quantizer(ptf, bits, cs, bitsC, mx, mn, lut)
        ctxs[key] = c
    w, h, B, nb = 3840, 2160, 20, 6
    n3 = 3 * w * h
    _, hs, st, _ = L.plane_geometry(w, h, 2)
    psz = [hs[p] * st[p] for p in range(3)]
    src = torch.empty(nb * B * n3, dtype=torch.float32, device=dev)
    planes = {k: [torch.zeros(nb * B * psz[p], dtype=torch.uint8, device=dev) for p in range(3)] for k in ctxs}
    g = torch.Generator(device=dev)
    g.manual_seed(1)
    print("fraction of pixels that are not halves | us per launch with the table | table off | ratio | default mode, mean of 24 launches (launches on the table) | planes equal")
    for p in (0.0, 1e-6, 1e-5, 1e-4, 1e-3, 1e-2, 0.1, 1.0):
        ctxs["table"].synth_frames_device(src.data_ptr(), n3, nb * B, w, h)
        if p > 0:
            v = src.view(torch.int32).view(nb * B, 3, h * w)
            for f in range(nb * B):      # per frame: keeps the temporaries small
                m = (torch.rand(h * w, device=dev, generator=g) < p)
                noise = torch.randint(1, 1 << 13, (3, h * w), device=dev, dtype=torch.int32, generator=g)
                v[f] |= noise * m.to(torch.int32)[None]
        torch.cuda.synchronize()
        ms = {k: [] for k in ctxs}
        i0 = ctxs["auto"].half_table_info(sc)
        for rep in range(4):
            for b in range(nb):
                for k, c in ctxs.items():
                    t = c.time_launches(0, 1, src.data_ptr() + b * B * n3 * 4, n3, B, w, h, sc, 2,
                                        [planes[k][q].data_ptr() + b * B * psz[q] for q in range(3)], st, psz)
                    if rep > 0 or k == "auto":
                        ms[k].append(t)
        med = {k: sorted(v)[len(v) // 2] for k, v in ms.items()}
        i1 = ctxs["auto"].half_table_info(sc)
        same = all(torch.equal(planes["table"][q], planes[k][q]) for q in range(3) for k in ("off", "auto"))
        print("%8.0e | %8.1f | %8.1f | %.2f | %8.1f (%d of %d) | %s" % (
            p, 1e3 * med["table"], 1e3 * med["off"], med["table"] / med["off"], 1e3 * sum(ms["auto"]) / len(ms["auto"]),
            i1["table_launches"] - i0["table_launches"], 4 * nb, same), flush=True)


if __name__ == "__main__":
    main()
