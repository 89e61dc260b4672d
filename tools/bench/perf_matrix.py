#!/usr/bin/env python3
"""Throughput of every colour space x VP9 profile (and of the fallback search modes) on device-resident 4K frames:
a sanity sweep that no supported configuration sits on a pathologically slow path."""
import os
import sys

ROOT = os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
sys.path.insert(0, ROOT)
import torch  # noqa: E402

import lumahdrv_amd as L  # noqa: E402


def main():
    w, h, B = 3840, 2160, 8
    n3 = 3 * w * h
    dev = torch.device("cuda:0")
    src = torch.empty(B * n3, dtype=torch.float32, device=dev)
    out = torch.empty(B * n3, dtype=torch.float32, device=dev)
    names = {L.CS_LUV: "Lu'v'", L.CS_RGB: "RGB", L.CS_YCBCR: "YCbCr", L.CS_XYZ: "XYZ"}
    rows = []
    cases = [(L.PTF_PQ, 11, cs, 8, None) for cs in (L.CS_LUV, L.CS_RGB, L.CS_YCBCR, L.CS_XYZ)]
    cases += [(L.PTF_LOG, 12, L.CS_LUV, 8, None), (L.PTF_PSI, 11, L.CS_LUV, 8, None), (L.PTF_PQ, 11, L.CS_LUV, 8, "literal"),
              (L.PTF_PQ, 13, L.CS_LUV, 8, None),
              # PTF_LINEAR: value-keyed records in LDS (search mode 7) against the float-bit records in global memory (mode 4)
              (L.PTF_LINEAR, 12, L.CS_LUV, 8, None), (L.PTF_LINEAR, 12, L.CS_LUV, 8, "no_lin_index"),
              (L.PTF_LINEAR, 14, L.CS_LUV, 8, None), (L.PTF_LINEAR, 14, L.CS_LUV, 8, "no_lin_index"), (L.PTF_LINEAR, 12, L.CS_RGB, 8, None)]
    for ptf, bits, cs, bitsC, force in cases:
        ctx = L.Context(0)
        if force == "literal":
            ctx.tune("force_literal", 1)
        if force == "no_lin_index":
            ctx.tune("lin_index", 0)
        ctx.set_stream(torch.cuda.current_stream().cuda_stream)
        ctx.set_quantizer(ptf, bits, cs, bitsC, 1e4, 0.005, L.build_lut(ptf, bits))
        info = ctx.quantizer_info()
        ctx.synth_frames_device(src.data_ptr(), n3, B, w, h)
        for profile in (2, 3, 0, 1):
            if profile < 2 and bits > 8:
                continue  # 8-bit containers take 8-bit tables
            _, hs, st, _ = L.plane_geometry(w, h, profile)
            psz = [hs[p] * st[p] for p in range(3)]
            planes = [torch.zeros(B * psz[p], dtype=torch.uint8, device=dev) for p in range(3)]
            pl = [p.data_ptr() for p in planes]
            me = sorted(ctx.time_launches(0, 1, src.data_ptr(), n3, B, w, h, 1.0, profile, pl, st, psz) for _ in range(5))[2]
            md = sorted(ctx.time_launches(1, 1, out.data_ptr(), n3, B, w, h, 1.0, profile, pl, st, psz) for _ in range(5))[2]
            px = B * w * h
            rows.append("ptf %d bits %2d %-6s profile %d  search mode %d : encode %7.1f Gpx/s  decode %7.1f Gpx/s"
                        % (ptf, bits, names[cs], profile, info["mode"], px / me / 1e6, px / md / 1e6))
            print(rows[-1], flush=True)
        ctx.close()
    # 8-bit profiles with an 8-bit table
    ctx = L.Context(0)
    ctx.set_stream(torch.cuda.current_stream().cuda_stream)
    ctx.set_quantizer(L.PTF_PQ, 8, L.CS_LUV, 8, 1e4, 0.005, L.build_lut(L.PTF_PQ, 8))
    for profile in (0, 1):
        _, hs, st, _ = L.plane_geometry(w, h, profile)
        psz = [hs[p] * st[p] for p in range(3)]
        planes = [torch.zeros(B * psz[p], dtype=torch.uint8, device=dev) for p in range(3)]
        pl = [p.data_ptr() for p in planes]
        me = sorted(ctx.time_launches(0, 1, src.data_ptr(), n3, B, w, h, 1.0, profile, pl, st, psz) for _ in range(5))[2]
        md = sorted(ctx.time_launches(1, 1, out.data_ptr(), n3, B, w, h, 1.0, profile, pl, st, psz) for _ in range(5))[2]
        print("ptf 1 bits  8 Lu'v'  profile %d  (8-bit samples)            : encode %7.1f Gpx/s  decode %7.1f Gpx/s"
              % (profile, B * w * h / me / 1e6, B * w * h / md / 1e6), flush=True)


if __name__ == "__main__":
    main()
