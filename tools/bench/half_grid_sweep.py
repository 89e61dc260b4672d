#!/usr/bin/env python3
"""tools/bench/half_grid_sweep.py -- persistent workgroups per CU for the half-input YCbCr encode kernel (k_encode<CS_YCBCR, ., ., 6>:
1024 threads and ~140 KiB of LDS per workgroup, so ONE is resident per CU and every further one re-stages the 124 KiB table).
{1280x720, 1920x1080, 3840x2160, 7680x4320} x {1, 2, 4, 8, 20, 50 frames per launch}, HDR10 recipe (PQ-10 YCbCr, preScaling 20),
isolated launches on distinct device-resident batches (>= 1.2 GB), every setting interleaved in one process as in
tools/bench/launch_rules_sweep.py.  Columns: the rule (blocks_per_cu = 0), each fixed setting, and the same launch with the table
turned off (lumahip_tune half_table 0: PQenc per pixel).  -> profiles/r04_half_table_grid.txt"""
import os
import sys

ROOT = os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
sys.path.insert(0, ROOT)
import torch  # noqa: E402

import lumahdrv_amd as L  # noqa: E402

CANDIDATES = (1, 2, 3, 4, 6, 8, 12, 16, 32)
SIZES = ((1280, 720), (1920, 1080), (3840, 2160), (7680, 4320))
BATCHES = (1, 2, 4, 8, 20, 50)


def main():
    dev = torch.device("cuda:0")
    ptf, bits, cs, bitsC, mx, mn, sc = L.PTF_PQ, 10, L.CS_YCBCR, 10, 1000.0, 0.01, 20.0
    lut = L.build_lut(ptf, bits, mx, mn)
    ctxs = {}
    for key in ("rule",) + CANDIDATES + ("off",):
        c = L.Context(0)
        if key == "off":
            c.tune("half_table", 0)
        elif key != "rule":
            c.tune("blocks_per_cu", key)
        c.set_stream(torch.cuda.current_stream().cuda_stream)
        c.set_quantizer(ptf, bits, cs, bitsC, mx, mn, lut)
        ctxs[key] = c
    assert ctxs["rule"].half_table_info(sc)["used"] and not ctxs["off"].half_table_info(sc)["used"]
    print("median us per launch, encode, 4:2:0 16-bit; columns: " + " ".join(str(k) for k in ctxs))
    for (w, h) in SIZES:
        for B in BATCHES:
            n3 = 3 * w * h
            _, hs, st, _ = L.plane_geometry(w, h, 2)
            psz = [hs[p] * st[p] for p in range(3)]
            batch_bytes = B * (n3 * 4 + sum(psz))
            if batch_bytes > 24e9:
                continue
            nb = int(max(3, min(48, -(-1.2e9 // batch_bytes))))
            src = torch.empty(nb * B * n3, dtype=torch.float32, device=dev)
            planes = [torch.zeros(nb * B * psz[p], dtype=torch.uint8, device=dev) for p in range(3)]
            ctxs["rule"].synth_frames_device(src.data_ptr(), n3, nb * B, w, h)
            torch.cuda.synchronize()
            reps = max(2, min(6, 72 // nb))
            ms = {k: [] for k in ctxs}
            i = 0
            for rep in range(reps + 1):
                for b in range(nb):
                    for k, c in ctxs.items():
                        bb = (b + i) % nb
                        i += 1
                        t = c.time_launches(0, 1, src.data_ptr() + bb * B * n3 * 4, n3, B, w, h, sc, 2,
                                            [planes[p].data_ptr() + bb * B * psz[p] for p in range(3)], st, psz)
                        if rep > 0:
                            ms[k].append(t)
            med = {k: sorted(v)[len(v) // 2] for k, v in ms.items()}
            best = min(CANDIDATES, key=lambda k: med[k])
            px = w * h * B
            print("%4dx%-4d x%-2d | %s | best %2d (%.1f us, %.0f Gpx/s); rule loses %.1f %%; table off x%.2f" % (
                w, h, B, " ".join("%7.1f" % (1e3 * med[k]) for k in ctxs), best, 1e3 * med[best], px / med[best] / 1e6,
                100 * (med["rule"] / med[best] - 1), med["off"] / med["rule"]), flush=True)
            del src, planes
            torch.cuda.empty_cache()


if __name__ == "__main__":
    main()
