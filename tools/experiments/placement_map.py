#!/usr/bin/env python3
"""tools/experiments/placement_map.py [GiB] -- speed map of device memory in 2 GiB chunks (allocated one by one, in order).
Per chunk: traffic-only probe time with that chunk as the INPUT of one 20-frame batch (planes in chunk 0), and with that
chunk holding the PLANES (input in chunk 0)."""
import os
import sys

ROOT = os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
sys.path.insert(0, ROOT)
import torch  # noqa: E402

import lumahdrv_amd as L  # noqa: E402


def main():
    total = int(sys.argv[1]) if len(sys.argv) > 1 else 240
    w, h, B, profile = 3840, 2160, 20, 2
    n3 = 3 * w * h
    dev = torch.device("cuda:0")
    _, hs, st, _ = L.plane_geometry(w, h, profile)
    psz = [hs[p] * st[p] for p in range(3)]
    ctx = L.Context(0)
    ctx.set_quantizer(L.PTF_PQ, 11, L.CS_LUV, 8, 1e4, 0.005, L.build_lut(L.PTF_PQ, 11, 1e4, 0.005))
    ctx.set_stream(torch.cuda.current_stream().cuda_stream)
    chunks = [torch.empty(2 << 30, dtype=torch.uint8, device=dev) for _ in range(total // 2)]
    print("chunk VAs:", " ".join("%x" % (c.data_ptr() >> 30) for c in chunks[:8]), "...")
    ctx.synth_frames_device(chunks[0].data_ptr(), n3, B, w, h, 20250929, 0)
    torch.cuda.synchronize()

    def planes_in(c):
        base = c.data_ptr()
        return [base, base + B * psz[0] + (1 << 20), base + B * (psz[0] + psz[1]) + (2 << 20)]

    rd, wr = [], []
    for i, c in enumerate(chunks):
        if i:
            c.copy_(chunks[0])
        torch.cuda.synchronize()
        rd.append(ctx.probe_encode_traffic(c.data_ptr(), n3, B, w, h, planes_in(chunks[1 if i == 0 else 0]), st, psz, iters=3))
    for i, c in enumerate(chunks):
        wr.append(ctx.probe_encode_traffic(chunks[1 if i == 0 else 0].data_ptr(), n3, B, w, h, planes_in(c), st, psz, iters=3))
    print("input in chunk i :", " ".join("%.3f" % v for v in rd))
    print("planes in chunk i:", " ".join("%.3f" % v for v in wr))


if __name__ == "__main__":
    main()
