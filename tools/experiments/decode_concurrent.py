#!/usr/bin/env python3
"""tools/experiments/decode_concurrent.py -- experiment: three decode launches whose outputs live in three different region groups,
run CONCURRENTLY on three streams with a third of the persistent workgroups each, against the same three launches one after
the other (each with the full grid)."""
import os
import sys
import time

ROOT = os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
sys.path.insert(0, ROOT)
import torch  # noqa: E402

import lumahdrv_amd as L  # noqa: E402
from lumahdrv_amd.placement import CHUNK_BYTES, find_groups, plane_slots  # noqa: E402


def make_ctx(grid=None):
    c = L.Context(0)
    if grid:
        c.tune("grid_enc", grid)
        c.tune("grid_dec", grid)
    c.set_quantizer(L.PTF_PQ, 11, L.CS_LUV, 8, 1e4, 0.005, L.build_lut(L.PTF_PQ, 11, 1e4, 0.005))
    return c


def main():
    n = int(sys.argv[1]) if len(sys.argv) > 1 else 60
    w, h, B, profile = 3840, 2160, 20, 2
    n3 = 3 * w * h
    dev = torch.device("cuda:0")
    _, hs, st, _ = L.plane_geometry(w, h, profile)
    psz = [hs[p] * st[p] for p in range(3)]
    _, _, offs = plane_slots(CHUNK_BYTES, [B * x for x in psz])
    ctx = make_ctx()
    ctx.set_stream(torch.cuda.current_stream().cuda_stream)
    chunks = [torch.zeros(CHUNK_BYTES, dtype=torch.uint8, device=dev) for _ in range(n)]
    torch.cuda.synchronize()

    def probe(i, r):
        pl = [chunks[r].data_ptr() + o for o in offs]
        return ctx.probe_encode_traffic(chunks[i].data_ptr(), n3, B, w, h, pl, st, psz, iters=2)

    probe(1, 0)
    groups, fast, _ = find_groups(n, probe)
    gs = sorted(groups, key=len, reverse=True)[:3]
    print("group sizes", [len(g) for g in groups])
    A, Bg, Cg = gs
    # three batches: planes in (A, B, C) chunks resp., produced by an encode from a float chunk of another group
    src = chunks[A[0]]
    ctx.synth_frames_device(src.data_ptr(), n3, B, w, h, 1, 0)
    layouts = {"outputs in A, B, C (planes in B, C, A)": ([A[1], Bg[1], Cg[1]], [Bg[2], Cg[2], A[2]]),
               "outputs all in A (planes in B)": ([A[1], A[3], A[4]], [Bg[2], Bg[3], Bg[4]])}
    streams = [torch.cuda.Stream(device=dev) for _ in range(3)]
    for name, (outs, pls) in layouts.items():
        plp = [[chunks[p].data_ptr() + o for o in offs] for p in pls]
        for k in range(3):
            ctx.encode_frames_device(src.data_ptr(), n3, B, w, h, 1.0, profile, plp[k], st, psz)
        torch.cuda.synchronize()
        res = {}
        for mode, grid in (("one after the other, full grid", None), ("concurrent, 427 workgroups each", 427), ("concurrent, 640 each", 640)):
            cs = [make_ctx(grid) for _ in range(3)]
            for k in range(3):
                cs[k].set_stream(streams[k].cuda_stream if grid else torch.cuda.current_stream().cuda_stream)
            ts = []
            for rep in range(7):
                torch.cuda.synchronize()
                t0 = time.perf_counter()
                for it in range(4):
                    for k in range(3):
                        cs[k].decode_frames_device(plp[k], st, psz, B, w, h, profile, 1.0, chunks[outs[k]].data_ptr(), n3)
                torch.cuda.synchronize()
                ts.append((time.perf_counter() - t0) / 12 * 1e3)
            res[mode] = sorted(ts)[len(ts) // 2]
            for c in cs:
                c.set_stream(None)
                c.close()
        print(name)
        for m, v in res.items():
            print("    %-36s %.4f ms per 20-frame launch  (%.0f Gpixel/s)" % (m, v, B * w * h / v / 1e6))
    encode_part(chunks, groups, offs, n3, B, w, h, profile, st, psz, dev)


def encode_part(chunks, groups, offs, n3, B, w, h, profile, st, psz, dev):
    gs = sorted(groups, key=len, reverse=True)[:3]
    A, Bg, Cg = gs
    streams = [torch.cuda.Stream(device=dev) for _ in range(3)]
    c0 = make_ctx()
    for k in range(3):
        c0.synth_frames_device(chunks[A[k]].data_ptr(), n3, B, w, h, 1, 20 * k)
    torch.cuda.synchronize()
    c0.close()
    yo, uo, vo = offs
    # inputs in A; Y planes in B (three slots of one chunk), U / V in C
    plp = [[chunks[Bg[0]].data_ptr() + k * (512 << 20), chunks[Cg[0]].data_ptr() + k * (512 << 20),
            chunks[Cg[0]].data_ptr() + k * (512 << 20) + (256 << 20)] for k in range(3)]
    print("encode: inputs in A, Y in B, U / V in C")
    for mode, grid, ns in (("one after the other, full grid (2048)", None, 1), ("2 streams, 1024 workgroups each", 1024, 2),
                           ("3 streams, 768 each", 768, 3), ("3 streams, 1024 each", 1024, 3), ("2 streams, 1536 each", 1536, 2)):
        cs = [make_ctx(grid) for _ in range(3)]
        for k in range(3):
            cs[k].set_stream(streams[k % ns].cuda_stream if grid else torch.cuda.current_stream().cuda_stream)
        ts = []
        for rep in range(7):
            torch.cuda.synchronize()
            t0 = time.perf_counter()
            for it in range(4):
                for k in range(3):
                    cs[k].encode_frames_device(chunks[A[k]].data_ptr(), n3, B, w, h, 1.0, profile, plp[k], st, psz)
            torch.cuda.synchronize()
            ts.append((time.perf_counter() - t0) / 12 * 1e3)
        v = sorted(ts)[len(ts) // 2]
        print("    %-40s %.4f ms per 20-frame launch  (%.0f Gpixel/s)" % (mode, v, B * w * h / v / 1e6))
        for c in cs:
            c.set_stream(None)
            c.close()


if __name__ == "__main__":
    main()
