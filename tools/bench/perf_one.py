#!/usr/bin/env python3
"""tools/bench/perf_one.py <workload> [reps] -- median / min kernel ms (hipEvents, isolated launches over distinct batches) of
encode and decode for one workload, 20 x 3840x2160 device-resident frames per launch.  Meant for same-box A/B runs of
two library builds:  for r in 1 2 3; do for l in a.so b.so; do LUMAHIP_LIB=$l python tools/bench/perf_one.py pq10_ycbcr; done; done"""
import os
import sys

ROOT = os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
sys.path.insert(0, ROOT)
import torch  # noqa: E402

import lumahdrv_amd as L  # noqa: E402

CFG = {"pq11_luv": (L.PTF_PQ, 11, L.CS_LUV, 8, 1e4, 0.005, 1.0), "pq10_ycbcr": (L.PTF_PQ, 10, L.CS_YCBCR, 10, 1000.0, 0.01, 20.0),
       "log12_luv": (L.PTF_LOG, 12, L.CS_LUV, 8, 1e4, 0.005, 1.0), "pq11_luv444": (L.PTF_PQ, 11, L.CS_LUV, 8, 1e4, 0.005, 1.0),
       "pq11_xyz444": (L.PTF_PQ, 11, L.CS_XYZ, 8, 1e4, 0.005, 1.0)}


def main():
    wl = sys.argv[1] if len(sys.argv) > 1 else "pq11_luv"
    reps = int(sys.argv[2]) if len(sys.argv) > 2 else 24
    ptf, bits, cs, bitsC, mx, mn, sc = CFG[wl]
    profile = 3 if wl.endswith("444") else 2
    w, h, B, nb = 3840, 2160, 20, 4
    n3 = 3 * w * h
    dev = torch.device("cuda:0")
    _, hs, st, _ = L.plane_geometry(w, h, profile)
    psz = [hs[p] * st[p] for p in range(3)]
    src = torch.empty(nb * B * n3, dtype=torch.float32, device=dev)
    planes = [torch.zeros(nb * B * psz[p], dtype=torch.uint8, device=dev) for p in range(3)]
    ctx = L.Context(0)
    ctx.set_stream(torch.cuda.current_stream().cuda_stream)
    ctx.set_quantizer(ptf, bits, cs, bitsC, mx, mn, L.build_lut(ptf, bits, mx, mn))
    ctx.synth_frames_device(src.data_ptr(), n3, nb * B, w, h)
    out = []
    for d in (0, 1):
        ms = []
        for r in range(reps):
            b = r % nb
            ms.append(ctx.time_launches(d, 1, src.data_ptr() + b * B * n3 * 4, n3, B, w, h, sc, profile,
                                        [planes[p].data_ptr() + b * B * psz[p] for p in range(3)], st, psz))
        ms = sorted(ms[2:])
        out.append("%s med %.4f min %.4f ms (%.1f Gpx/s)" % ("enc" if d == 0 else "dec", ms[len(ms) // 2], ms[0], B * w * h / ms[len(ms) // 2] / 1e6))
    print("%-12s %-28s %s" % (wl, os.path.basename(os.environ.get("LUMAHIP_LIB", "liblumahip.so")), " | ".join(out)), flush=True)


if __name__ == "__main__":
    main()
