#!/usr/bin/env python3
"""tools/prof_driver.py [n workload width height frames] -- small fixed workload for rocprofv3: N encode + N decode launches
over `frames` resident synthetic frames each (distinct batch per launch; default 20 x 3840x2160), then the traffic-only probes
of both directions.  Prints nothing but a one-line summary; timing and counters come from the profiler."""
import os
import sys

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
import torch  # noqa: E402

import lumahdrv_amd as L  # noqa: E402


def main():
    n = int(sys.argv[1]) if len(sys.argv) > 1 else 4
    wl = sys.argv[2] if len(sys.argv) > 2 else "pq11_luv"
    cfgs = {"pq11_luv": (L.PTF_PQ, 11, L.CS_LUV, 8, 1e4, 0.005, 1.0),
            "pq10_ycbcr": (L.PTF_PQ, 10, L.CS_YCBCR, 10, 1000.0, 0.01, 20.0),
            "log12_luv": (L.PTF_LOG, 12, L.CS_LUV, 8, 1e4, 0.005, 1.0)}
    ptf, bits, cs, bitsC, mx, mn, sc = cfgs[wl]
    w = int(sys.argv[3]) if len(sys.argv) > 3 else 3840
    h = int(sys.argv[4]) if len(sys.argv) > 4 else 2160
    B = int(sys.argv[5]) if len(sys.argv) > 5 else 20
    n3 = 3 * w * h
    dev = torch.device("cuda:0")
    _, hs, st, _ = L.plane_geometry(w, h, 2)
    psz = [hs[p] * st[p] for p in range(3)]
    src = torch.empty(n * B * n3, dtype=torch.float32, device=dev)
    out = torch.empty(n * B * n3, dtype=torch.float32, device=dev)
    planes = [torch.zeros(n * B * psz[p], dtype=torch.uint8, device=dev) for p in range(3)]
    ctx = L.Context(0)
    ctx.set_stream(torch.cuda.current_stream().cuda_stream)
    ctx.set_quantizer(ptf, bits, cs, bitsC, mx, mn, L.build_lut(ptf, bits, mx, mn))
    ctx.synth_frames_device(src.data_ptr(), n3, n * B, w, h)
    torch.cuda.synchronize()
    for rep in range(2):
        for b in range(n):
            pl = [planes[p].data_ptr() + b * B * psz[p] for p in range(3)]
            ctx.encode_frames_device(src.data_ptr() + b * B * n3 * 4, n3, B, w, h, sc, 2, pl, st, psz)
        for b in range(n):
            pl = [planes[p].data_ptr() + b * B * psz[p] for p in range(3)]
            ctx.decode_frames_device(pl, st, psz, B, w, h, 2, sc, out.data_ptr() + b * B * n3 * 4, n3)
    if cs == L.CS_YCBCR:
        # the per-pixel encode kernel (k_encode<CS_YCBCR, ., ., LM = 5>: what a stream of full-precision floats runs after the
        # half-input policy has backed off) on its own input class, for bench.py's float_inputs VALU roofline: a second stream
        # whose values all carry random low mantissa bits, with the half-input table switched off
        fsrc = src.clone()
        g = torch.Generator(device=dev)
        g.manual_seed(1)
        v = fsrc.view(torch.int32).view(n * B * 3, w * h)
        for i in range(v.shape[0]):
            v[i] |= torch.randint(1, 1 << 13, (w * h,), device=dev, dtype=torch.int32, generator=g)
        ctx.tune("half_table", 0)
        for rep in range(2):
            for b in range(n):
                pl = [planes[p].data_ptr() + b * B * psz[p] for p in range(3)]
                ctx.encode_frames_device(fsrc.data_ptr() + b * B * n3 * 4, n3, B, w, h, sc, 2, pl, st, psz)
        ctx.tune("half_table", 1)
        torch.cuda.synchronize()
        del fsrc
    # the traffic-only probes (same loads and stores, no arithmetic): known byte counts in the kernels' own access patterns,
    # i.e. the calibration of FETCH_SIZE / WRITE_SIZE the microarchitecture guide asks for (they overwrite planes / frames)
    for b in range(n):
        pl = [planes[p].data_ptr() + b * B * psz[p] for p in range(3)]
        o = out.data_ptr() + b * B * n3 * 4
        ctx.probe_decode_traffic(pl, st, psz, B, w, h, [o + k * w * h * 4 for k in range(3)], n3, 1)
        ctx.probe_encode_traffic(src.data_ptr() + b * B * n3 * 4, n3, B, w, h, pl, st, psz, 1)
    torch.cuda.synchronize()
    print("prof_driver done: %d launches each of encode/decode, %d pixels per launch" % (2 * n, B * w * h))


if __name__ == "__main__":
    main()
