#!/usr/bin/env python3
"""tools/experiments/placement_sweep.py -- traffic-only probe time of ONE 20-frame batch as a function of buffer offsets inside one
big arena: (a) the input base shifted in 4 KiB ... 2 MiB steps with the planes fixed, (b) the planes shifted with the
input fixed, (c) the frame stride padded.  Prints ms per launch (median of 5)."""
import os
import statistics
import sys

ROOT = os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
sys.path.insert(0, ROOT)
import torch  # noqa: E402

import lumahdrv_amd as L  # noqa: E402


def main():
    w, h, B, profile = 3840, 2160, 20, 2
    n3 = 3 * w * h
    dev = torch.device("cuda:0")
    _, hs, st, _ = L.plane_geometry(w, h, profile)
    psz = [hs[p] * st[p] for p in range(3)]
    ctx = L.Context(0)
    ctx.set_quantizer(L.PTF_PQ, 11, L.CS_LUV, 8, 1e4, 0.005, L.build_lut(L.PTF_PQ, 11, 1e4, 0.005))
    ctx.set_stream(torch.cuda.current_stream().cuda_stream)
    slack = 64 << 20
    src = torch.empty(B * n3 * 4 + 2 * slack, dtype=torch.uint8, device=dev)
    planes = [torch.zeros(B * psz[p] + 2 * slack, dtype=torch.uint8, device=dev) for p in range(3)]
    ctx.synth_frames_device(src.data_ptr(), n3, B, w, h, 20250929, 0)
    torch.cuda.synchronize()
    print("src 0x%x  Y 0x%x  U 0x%x  V 0x%x" % (src.data_ptr(), planes[0].data_ptr(), planes[1].data_ptr(), planes[2].data_ptr()))

    def t(so=0, po=(0, 0, 0), fs=n3, pfs=None, enc=False):
        pl = [planes[p].data_ptr() + po[p] for p in range(3)]
        v = []
        for _ in range(5):
            if enc:
                v.append(ctx.time_launches(0, 1, src.data_ptr() + so, fs, B, w, h, 1.0, profile, pl, st, pfs or psz))
            else:
                v.append(ctx.probe_encode_traffic(src.data_ptr() + so, fs, B, w, h, pl, st, pfs or psz))
        return statistics.median(v)

    for _ in range(3):
        t()
    print("baseline probe %.4f  encode %.4f" % (t(), t(enc=True)))
    print("(a) input base offset (planes fixed)")
    for off in [0, 256, 1024, 4096, 8192, 16384, 32768, 65536, 1 << 17, 1 << 18, 1 << 19, 1 << 20, 3 << 19, 1 << 21, 1 << 22, 1 << 23, 1 << 24]:
        print("  +%8d B : %.4f" % (off, t(so=off)), flush=True)
    print("(b) Y plane base offset (input, U, V fixed)")
    for off in [0, 256, 1024, 4096, 8192, 16384, 32768, 65536, 1 << 17, 1 << 18, 1 << 19, 1 << 20, 1 << 21, 1 << 22]:
        print("  +%8d B : %.4f" % (off, t(po=(off, 0, 0))), flush=True)
    print("(c) frame stride padded by (floats)")
    for pad in [0, 64, 256, 1024, 4096, 16384, 65536, 1 << 18, 1 << 20]:
        if (B - 1) * (n3 + pad) * 4 + n3 * 4 <= B * n3 * 4 + 2 * slack:
            print("  +%8d floats : %.4f" % (pad, t(fs=n3 + pad)), flush=True)
    print("(d) repeat baseline %.4f %.4f %.4f" % (t(), t(), t()))


if __name__ == "__main__":
    main()
