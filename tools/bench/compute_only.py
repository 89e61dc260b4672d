#!/usr/bin/env python3
"""How long does the encode / decode kernel take when its pixel stream comes from cache instead of HBM?
All `nframes` frames alias ONE resident frame (frame stride 0), so the working set (25 MB in + 6 MB out for
1080p) sits in the 256 MB Infinity Cache: the time that remains is the kernel's compute + LDS + issue time.
Compare with the HBM-fed time of the same pixel count (tools/bench/tune.py) and the traffic-only time
(tools/experiments/membench2.hip)."""
import os
import sys

ROOT = os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
sys.path.insert(0, ROOT)
import torch  # noqa: E402

import lumahdrv_amd as L  # noqa: E402


def main():
    os.environ["LUMAHIP_TUNING"] = "1"   # the LUMAHIP_* overrides are honoured only under this gate
    os.environ["LUMAHIP_ALLOW_ALIASED_FRAMES"] = "1"   # frame stride 0 is refused otherwise (rows of different frames overlap)
    dev = torch.device("cuda:0")
    ctx = L.Context(0)
    ctx.set_stream(torch.cuda.current_stream().cuda_stream)
    ctx.set_quantizer(L.PTF_PQ, 11, L.CS_LUV, 8, 1e4, 0.005, L.build_lut(L.PTF_PQ, 11))
    for (w, h, nf) in ((1920, 1080, 80), (3840, 2160, 20)):
        n3 = 3 * w * h
        _, hs, st, _ = L.plane_geometry(w, h, 2)
        psz = [hs[p] * st[p] for p in range(3)]
        src = torch.empty(nf * n3, dtype=torch.float32, device=dev)
        out = torch.empty(n3, dtype=torch.float32, device=dev)
        planes = [torch.zeros(nf * psz[p], dtype=torch.uint8, device=dev) for p in range(3)]
        ctx.synth_frames_device(src.data_ptr(), n3, nf, w, h)
        pl = [p.data_ptr() for p in planes]
        for label, fs, pfs in (("HBM-fed (distinct frames)", n3, psz), ("cache-fed (all frames alias frame 0)", 0, [0, 0, 0])):
            ms_e = sorted(ctx.time_launches(0, 1, src.data_ptr(), fs, nf, w, h, 1.0, 2, pl, st, pfs) for _ in range(9))[4]
            ms_d = sorted(ctx.time_launches(1, 1, out.data_ptr() if fs == 0 else src.data_ptr(), fs, nf, w, h, 1.0, 2, pl, st, pfs)
                          for _ in range(9))[4]
            px = nf * w * h
            print("%dx%d x%d  %-38s enc %.4f ms (%.0f Gpx/s)   dec %.4f ms (%.0f Gpx/s)" %
                  (w, h, nf, label, ms_e, px / ms_e / 1e6, ms_d, px / ms_d / 1e6), flush=True)


if __name__ == "__main__":
    main()
