#!/usr/bin/env python3
"""tools/experiments/settle_exp.py -- does freeing a lot of device memory disturb what runs next?
(1) the chunk pool takes ~280 GB, keeps 33 chunks and frees ~100 (200 GB): time series of the encode rate right afterwards;
(2) a process frees 200 GB, then tools/facade_hostfed (latency-bound: one host frame per call) is run repeatedly.
-> profiles/r03_settle.txt"""
import os
import subprocess
import sys
import time

ROOT = os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
sys.path.insert(0, ROOT)
import torch  # noqa: E402

import lumahdrv_amd as L  # noqa: E402
from lumahdrv_amd import capi  # noqa: E402

HF = os.path.join(ROOT, "lumahdrv_amd", "bin", "facade_hostfed")


def hostfed(n=12):
    import json
    out = subprocess.run([HF, "3840", "2160", str(n)], capture_output=True, text=True).stdout
    d = json.loads([l for l in out.splitlines() if l.startswith("{")][-1])
    return d["LumaEncoder_encode_pageable_frame"], d["LumaEncoder_encode_registered_frame"], d["lumahip_encode_frames_host_pinned"]


def main():
    print("hostfed on the idle box (pageable, registered, batched pinned Mpixel/s):", hostfed(), flush=True)
    w, h, B = 3840, 2160, 20
    n1, n3 = w * h, 3 * w * h
    _, hs, st, _ = L.plane_geometry(w, h, 2)
    psz = [hs[p] * st[p] for p in range(3)]
    ctx = L.Context(0)
    ctx.set_quantizer(L.PTF_PQ, 11, L.CS_LUV, 8, 1e4, 0.005, L.build_lut(L.PTF_PQ, 11))
    t0 = time.perf_counter()
    pool = capi.Pool(ctx, 25, 5, 3)
    t_pool = time.perf_counter() - t0
    print("pool:", t_pool, "s", pool.stats().get("groups"), flush=True)
    fl = [pool.alloc(0) for _ in range(25)]
    ych = [pool.alloc(1) for _ in range(5)]
    uvch = [pool.alloc(2) for _ in range(3)]
    yslot = (B * psz[0] + (64 << 20) - 1) // (64 << 20) * (64 << 20)
    vo = (B * psz[1] + (1 << 20) - 1) // (1 << 20) * (1 << 20)
    uvslot = (vo + B * psz[2] + (64 << 20) - 1) // (64 << 20) * (64 << 20)
    ypc, uvpc = (2 << 30) // yslot, (2 << 30) // uvslot

    def planes(b):
        u = uvch[b // uvpc] + (b % uvpc) * uvslot
        return [ych[b // ypc] + (b % ypc) * yslot, u, u + vo]
    for b in range(25):
        ctx.synth_frames_device(fl[b], n3, B, w, h, 1, b * B)
    ctx.sync()
    t_ready = time.perf_counter() - t0
    print("== encode rate after the pool has freed its rejected chunks (t = s since pool creation began; 25 launches, 2 lanes) ==")
    series = []
    while time.perf_counter() - t0 < t_ready + 25.0:
        ts = time.perf_counter()
        ctx.begin_unordered(2)
        for b in range(25):
            ctx.encode_frames_device(fl[b], n3, B, w, h, 1.0, 2, planes(b), st, psz)
        ctx.end_unordered()
        ctx.sync()
        te = time.perf_counter()
        series.append((ts - t0, (te - ts) / 25 * 1e3))
    k = max(1, len(series) // 50)
    for i in range(0, len(series), k):
        seg = sorted(x[1] for x in series[i:i + k])
        print("  t=%6.2f s  median %.4f ms/launch (%.0f Mpixel/s)" % (series[i][0], seg[len(seg) // 2], B * n1 / seg[len(seg) // 2] / 1e3), flush=True)
    pool.close()
    ctx.close()
    print("== hostfed after this process freed the pool's chunks (pageable, registered, batched pinned) ==")
    t1 = time.perf_counter()
    for i in range(14):
        print("  t=%5.1f s after the free:" % (time.perf_counter() - t1), hostfed(8), flush=True)
        if i >= 6:
            time.sleep(5)


if __name__ == "__main__":
    main()
