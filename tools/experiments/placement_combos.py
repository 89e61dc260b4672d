#!/usr/bin/env python3
"""tools/experiments/placement_combos.py -- after finding the region groups (lumahdrv_amd.placement.find_groups), time the traffic-only
launch for different assignments of the four streams (input, Y, U, V) to groups."""
import os
import sys

ROOT = os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
sys.path.insert(0, ROOT)
import torch  # noqa: E402

import lumahdrv_amd as L  # noqa: E402
from lumahdrv_amd.placement import CHUNK_BYTES, find_groups, plane_slots  # noqa: E402


def main():
    n = int(sys.argv[1]) if len(sys.argv) > 1 else 60
    w, h, B, profile = 3840, 2160, 20, 2
    n3 = 3 * w * h
    dev = torch.device("cuda:0")
    _, hs, st, _ = L.plane_geometry(w, h, profile)
    psz = [hs[p] * st[p] for p in range(3)]
    _, _, offs = plane_slots(CHUNK_BYTES, [B * x for x in psz])
    ctx = L.Context(0)
    ctx.set_quantizer(L.PTF_PQ, 11, L.CS_LUV, 8, 1e4, 0.005, L.build_lut(L.PTF_PQ, 11, 1e4, 0.005))
    ctx.set_stream(torch.cuda.current_stream().cuda_stream)
    chunks = [torch.zeros(CHUNK_BYTES, dtype=torch.uint8, device=dev) for _ in range(n)]
    torch.cuda.synchronize()

    def t4(i, y, u, v, enc=False):
        pl = [chunks[y].data_ptr() + offs[0], chunks[u].data_ptr() + offs[1], chunks[v].data_ptr() + offs[2]]
        if enc:
            return ctx.time_launches(0, 3, chunks[i].data_ptr(), n3, B, w, h, 1.0, profile, pl, st, psz)
        return ctx.probe_encode_traffic(chunks[i].data_ptr(), n3, B, w, h, pl, st, psz, iters=3)

    t4(1, 0, 0, 0)
    groups, fast, probes = find_groups(n, lambda i, r: t4(i, r, r, r))
    print("groups:", groups, "fast pair %.4f ms, %d probes" % (fast, probes))
    if not groups or len(groups) < 2:
        return
    gs = sorted(groups, key=len, reverse=True)
    a, b = gs[0], gs[1]
    c = gs[2] if len(gs) > 2 else None
    A0, A1, A2, A3 = a[0], a[1], a[2], a[3]
    B0, B1, B2 = b[0], b[1], b[min(2, len(b) - 1)]
    rows = [("input A, Y U V in one chunk of A (same group)", (A0, A1, A1, A1)),
            ("input A, Y U V in one chunk of B", (A0, B0, B0, B0)),
            ("input A, Y U V in three chunks of B", (A0, B0, B1, B2)),
            ("input A, Y in B, U V in A", (A0, B0, A1, A2)),
            ("input A, Y in A, U V in B", (A0, A1, B0, B1))]
    if c:
        C0, C1 = c[0], c[min(1, len(c) - 1)]
        rows += [("input A, Y in B, U V in C", (A0, B0, C0, C1)), ("input A, Y in B, U in C, V in A", (A0, B0, C0, A1)),
                 ("input C, Y U V in one chunk of B", (C0, B0, B0, B0))]
    for name, (i, y, u, v) in rows:
        print("%-48s probe %.4f ms   encode %.4f ms" % (name, t4(i, y, u, v), t4(i, y, u, v, True)), flush=True)
    # decode direction: planes read, floats written -- same relation?
    pl = [chunks[B0].data_ptr() + o for o in offs]
    ctx.encode_frames_device(chunks[A0].data_ptr(), n3, B, w, h, 1.0, profile, pl, st, psz)
    for name, o in (("decode: planes B -> output A", A1), ("decode: planes B -> output B", B1)):
        print("%-48s %.4f ms" % (name, ctx.time_launches(1, 3, chunks[o].data_ptr(), n3, B, w, h, 1.0, profile, pl, st, psz)))


if __name__ == "__main__":
    main()
