"""BASELINE configs[4]: ONE stream of F frames block-sharded over the GPUs -- one process per GPU under torch.distributed
(run_stream) or one process driving every GPU through the C ABI's many-GPU layer (run_stream_multi)."""
from __future__ import annotations

import json
import time

import numpy as np
import torch
import torch.distributed as dist

from .plan import SEED, WORKLOADS, config_block, stream_shard_plan
from .resident import make_pool
from .timing import Timer


def frame_digests(planes, psz, nfr, dev):
    """one int64 per frame over its three planes (position-weighted sums of the 8-byte words)"""
    d = torch.zeros(nfr, dtype=torch.int64, device=dev)
    for p in range(3):
        v = planes[p][:nfr * psz[p]].view(nfr, psz[p])
        words = v.view(torch.int64) if psz[p] % 8 == 0 else v.to(torch.int64)
        wgt = (torch.arange(words.shape[1], dtype=torch.int64, device=dev) % 1000003) * 2 + 1
        for f0 in range(0, nfr, 16):
            d[f0:f0 + 16] += (words[f0:f0 + 16] * wgt).sum(dim=1) * (p + 1)
    return d


def run_stream(L, args, rank, world, local_rank, use_dist, dev):
    """BASELINE configs[4]: ONE stream of F frames, block-sharded (lumahdrv_amd.sharding.shard_range: 2000 -> 250 per
    GPU at N = 8), each rank's shard resident in its HBM; a timed region = every rank encodes its whole shard once."""
    from lumahdrv_amd.sharding import broadcast_quantizer, gather_in_stream_order, shard_range
    name = args.workload
    ptf, bits, cs, bitsC, maxLum, minLum, sc, profile, desc, xf_desc, kname = WORKLOADS[name]
    w, h, B, F = args.width, args.height, args.frames_per_step, args.stream_frames
    cfg0 = lut0 = None
    if rank == 0:
        cfg0 = (ptf, bits, cs, bitsC, maxLum, minLum, sc, profile)
        lut0 = L.build_lut(ptf, bits, maxLum, minLum)
    cfg, lut = broadcast_quantizer(cfg0, lut0, dev)
    ptf, bits, cs, bitsC, maxLum, minLum, sc, profile = cfg
    ctx = L.Context(local_rank)
    ctx.set_stream(torch.cuda.current_stream().cuda_stream)
    ctx.set_quantizer(ptf, bits, cs, bitsC, maxLum, minLum, lut)
    mine = shard_range(F, rank, world)
    n3 = 3 * w * h
    _, hs, st, _ = L.plane_geometry(w, h, profile)
    psz = [hs[p] * st[p] for p in range(3)]
    free, _total = torch.cuda.mem_get_info(dev)
    plan = stream_shard_plan(F, rank, world, w, h, B, free, args.placement, profile)   # (what --plan-only prints)
    # A shard that does not fit the free HBM at once (the whole 2000-frame stream on ONE GPU with less than ~262 GB free) is encoded
    # in `segments` consecutive blocks, each resident while it is timed; every rank takes the same number of segments
    nseg_t = torch.tensor([plan["segments"]], dtype=torch.int64, device=dev)
    if use_dist:
        dist.all_reduce(nseg_t, op=dist.ReduceOp.MAX)
    nseg = int(nseg_t.item())
    lanes = max(0, args.lanes)
    dig, seg_te, pool_stats, K_total = [], [], None, 0
    for sg in range(nseg):
        blk = shard_range(len(mine), sg, nseg)                      # this segment's frames, relative to the shard
        first_frame, nfr = mine.start + blk.start, len(blk)
        steps = (nfr + B - 1) // B
        free, _total = torch.cuda.mem_get_info(dev)
        splan = stream_shard_plan(nfr, 0, 1, w, h, B, free, args.placement, profile) if nseg > 1 else plan
        pool = None
        if splan["pool"] is not None and steps:
            pool = make_pool(L, args, dev, local_rank, w, h, B, nbatches=steps, with_output=False)
        if pool is not None:
            from lumahdrv_amd.placement import CHUNK_BYTES, slots
            ypc, yslot = slots(CHUNK_BYTES, B * psz[0])
            uvpc, uvslot = slots(CHUNK_BYTES, B * psz[1] + (1 << 20) + B * psz[2])
            if len(pool.float) < steps or len(pool.y) < -(-steps // ypc) or len(pool.uv) < -(-steps // uvpc):
                pool.close()                       # (the driver gave fewer chunks than the plan asked for: plain allocations)
                pool = None
        src = planes = None
        if pool is not None:
            # step k's frames in chunk k of the pool's float chunks; Y and U / V planes in their own chunks (placement.py)
            src_c, y_c, uv_c = pool.take_float(steps), pool.take_y(-(-steps // ypc)), pool.take_uv(-(-steps // uvpc))
            vo = (B * psz[1] + (1 << 20) - 1) // (1 << 20) * (1 << 20)

            def where(k, src_c=src_c, y_c=y_c, uv_c=uv_c, ypc=ypc, yslot=yslot, uvpc=uvpc, uvslot=uvslot, vo=vo):      # (input pointer, plane pointers) of step k
                u = uv_c[k // uvpc].data_ptr() + (k % uvpc) * uvslot
                return src_c[k].data_ptr(), [y_c[k // ypc].data_ptr() + (k % ypc) * yslot, u, u + vo]

            def plane_views(k, nb, y_c=y_c, uv_c=uv_c, ypc=ypc, yslot=yslot, uvpc=uvpc, uvslot=uvslot, vo=vo):
                yo, uo = (k % ypc) * yslot, (k % uvpc) * uvslot
                return [y_c[k // ypc][yo:yo + nb * psz[0]], uv_c[k // uvpc][uo:uo + nb * psz[1]],
                        uv_c[k // uvpc][uo + vo:uo + vo + nb * psz[2]]]
        else:
            src = torch.empty(max(nfr, 1) * n3, dtype=torch.float32, device=dev)
            planes = [torch.zeros(max(nfr, 1) * psz[p], dtype=torch.uint8, device=dev) for p in range(3)]

            def where(k, src=src, planes=planes):
                return src.data_ptr() + k * B * n3 * 4, [planes[p].data_ptr() + k * B * psz[p] for p in range(3)]

            def plane_views(k, nb, planes=planes):
                return [planes[p][k * B * psz[p]:(k * B + nb) * psz[p]] for p in range(3)]
        for k in range(steps):
            ctx.synth_frames_device(where(k)[0], n3, min(B, nfr - k * B), w, h, SEED, first_frame + k * B)

        def enc(i, steps=steps, nfr=nfr, where=where):
            k = i % max(steps, 1)
            nb = min(B, nfr - k * B)
            if nb > 0:
                s_, pl_ = where(k)
                ctx.encode_frames_device(s_, n3, nb, w, h, sc, profile, pl_, st, psz)

        ksteps = torch.tensor([steps], dtype=torch.int64, device=dev)
        if use_dist:
            dist.all_reduce(ksteps, op=dist.ReduceOp.MAX)
        K = int(ksteps.item())                      # every rank issues K step calls (empty ones past its shard)
        K_total += K
        tm = Timer(K, 0, use_dist, dev, args.min_seconds / nseg, args.max_repeats, ctx, lanes)
        enc(0)                                      # warm-up: one step
        seg_te.append(tm.run(enc))
        torch.cuda.synchronize()
        # in-order reassembly bookkeeping: per-frame digests, gathered in STREAM order below
        for k in range(steps):
            nb = min(B, nfr - k * B)
            dig += frame_digests(plane_views(k, nb), psz, nb, dev).cpu().tolist()
        if pool is not None:
            pool_stats = pool.stats
            del src_c, y_c, uv_c
            pool.close()
        del src, planes, where, plane_views, enc
        torch.cuda.empty_cache()
    K = K_total
    te = {"wall_median": sum(t["wall_median"] for t in seg_te), "wall_min": sum(t["wall_min"] for t in seg_te),
          "wall_max": sum(t["wall_max"] for t in seg_te), "repeats": min(t["repeats"] for t in seg_te),
          "seconds": sum(t["seconds"] for t in seg_te)}
    pool = None
    allv = gather_in_stream_order(dig, F, dev)
    if args.dump_digests and rank == 0:
        json.dump(allv, open(args.dump_digests, "w"))
    checked = 0
    if rank == 0:
        one = torch.empty(n3, dtype=torch.float32, device=dev)
        pl1 = [torch.zeros(psz[p], dtype=torch.uint8, device=dev) for p in range(3)]
        for r in range(world):
            rg = shard_range(F, r, world)
            for f in sorted({rg.start, rg.stop - 1} if len(rg) else ()):
                ctx.synth_frames_device(one.data_ptr(), n3, 1, w, h, SEED, f)
                ctx.encode_frames_device(one.data_ptr(), n3, 1, w, h, sc, profile, [t.data_ptr() for t in pl1], st, psz)
                torch.cuda.synchronize()
                got = int(frame_digests(pl1, psz, 1, dev)[0].item()) & 0x7FFFFFFFFFFFFFFF
                if got != allv[f]:
                    raise SystemExit("stream frame %d (rank %d's shard): gathered digest differs from rank 0's re-encode" % (f, r))
                checked += 1
    px = float(F) * w * h
    res = {"metric": "Mpixels/s HDR quantize (4K PQ Lu'v' 11-bit), %d-frame stream block-sharded over the GPUs" % F,
           "value": round(px / te["wall_median"] / 1e6, 1), "unit": "Mpixels/s", "n_gpus": world, "steps": K, "warmup": 1,
           "ms_per_step": round(1e3 * te["wall_median"] / max(K, 1), 4), "higher_is_better": True, "scaling": "strong",
           "vs_baseline": None, "dtype": "f32", "data": "synthetic",
           "config": dict(config_block(args, world), world_size_reported_by="torch.distributed (RCCL)" if use_dist else "single process"),
           "repeats": te["repeats"], "timed_seconds": round(te["seconds"], 3),
           "ms_per_region_min_median_max": [round(1e3 * te[k], 3) for k in ("wall_min", "wall_median", "wall_max")],
           "digests": {"gathered_in_stream_order": len(allv), "spot_checked_by_rank0": checked,
                       "stream_digest": "%016x" % (sum((i + 1) * v for i, v in enumerate(allv)) & 0xFFFFFFFFFFFFFFFF)},
           "segments": nseg,
           "segments_note": ("the shard of every rank is resident in its HBM at once" if nseg == 1 else
                             "the shard did not fit the free HBM at once: encoded in %d consecutive blocks, each resident while it is "
                             "timed; value = all frames / the sum of the blocks' median times" % nseg),
           "placement": dict({"mode": args.placement if pool_stats is not None else "off (plain allocations)"},
                             **(pool_stats if pool_stats is not None else {}))}
    exp = res["config"].get("expected_stream_digest")
    if exp is not None:
        res["digests"]["equals_expected"] = res["digests"]["stream_digest"] == exp
    ctx.close()
    return res


def run_stream_multi(L, args):
    """BASELINE configs[4] through the C ABI's many-GPU layer (lumahip_multi_*): ONE process, one shard (context + stream)
    per GPU, the table built once on the host and broadcast to the GPUs with RCCL from C++, the F-frame stream block-sharded
    with lumahip_shard_range, every shard resident in its GPU's HBM.  A timed region = every shard encodes its block once;
    the host enqueues step k on every GPU before step k+1 (launches are asynchronous), then waits for all of them."""
    from lumahdrv_amd import capi
    name = args.workload
    ptf, bits, cs, bitsC, maxLum, minLum, sc, profile, desc, xf_desc, kname = WORKLOADS[name]
    w, h, B, F = args.width, args.height, args.frames_per_step, args.stream_frames
    have = torch.cuda.device_count()
    if have < args.gpus:
        raise SystemExit("bench.py --gpus %d: only %d GPU(s) visible" % (args.gpus, have))
    ns = args.gpus
    m = capi.Multi(list(range(ns)))
    m.set_quantizer(ptf, bits, cs, bitsC, maxLum, minLum, L.build_lut(ptf, bits, maxLum, minLum))
    n3 = 3 * w * h
    _, hs, st, _ = L.plane_geometry(w, h, profile)
    psz = [hs[p] * st[p] for p in range(3)]
    shards = [capi.shard_range(F, s, ns) for s in range(ns)]
    # shards that do not fit their GPU's free HBM at once are encoded in consecutive blocks (as run_stream does)
    nseg = 1
    for s in range(ns):
        free, _t = torch.cuda.mem_get_info(torch.device("cuda", s))
        need = (n3 * 4 + sum(psz)) * max(len(shards[s]), 1)
        if need > free * 0.9:
            nseg = max(nseg, int(-(-need // int(free * 0.8))))
    lanes = max(0, args.lanes)
    dig_s = [[] for _ in range(ns)]
    seg_walls, steps_total = [], 0
    for sg in range(nseg):
        blocks = [capi.shard_range(len(shards[s]), sg, nseg) for s in range(ns)]      # relative to each shard
        src, planes = [], []
        for s in range(ns):
            dev = torch.device("cuda", s)
            nfr = len(blocks[s])
            src.append(torch.empty(max(nfr, 1) * n3, dtype=torch.float32, device=dev))
            planes.append([torch.zeros(max(nfr, 1) * psz[p], dtype=torch.uint8, device=dev) for p in range(3)])
            c = m.ctx(s)
            for k in range(0, nfr, B):
                c.synth_frames_device(src[s].data_ptr() + k * n3 * 4, n3, min(B, nfr - k), w, h, SEED, shards[s].start + blocks[s].start + k)
        m.sync()
        steps = max((len(r) + B - 1) // B for r in blocks)
        steps_total += steps

        def one_pass(steps=steps, blocks=blocks, src=src, planes=planes):
            # the steps of a pass are independent batches: every shard runs them inside one unordered section (two lanes)
            if lanes:
                for s in range(ns):
                    m.ctx(s).begin_unordered(lanes)
            for k in range(steps):
                counts = [max(0, min(B, len(blocks[s]) - k * B)) for s in range(ns)]
                m.encode_frames_device([src[s].data_ptr() + k * B * n3 * 4 for s in range(ns)], n3, counts, w, h, sc, profile,
                                       [[planes[s][p].data_ptr() + k * B * psz[p] for p in range(3)] for s in range(ns)], st, psz)
            if lanes:
                for s in range(ns):
                    m.ctx(s).end_unordered()
            m.sync()

        one_pass()                                  # warm-up
        walls = []
        while sum(walls) < args.min_seconds / nseg and len(walls) < args.max_repeats:
            t0 = time.perf_counter()
            one_pass()
            walls.append(time.perf_counter() - t0)
        seg_walls.append(walls)
        for s in range(ns):
            nfr = len(blocks[s])
            if nfr:
                dig_s[s] += [int(x) & 0x7FFFFFFFFFFFFFFF for x in frame_digests(planes[s], psz, nfr, torch.device("cuda", s)).cpu().tolist()]
        del src, planes, one_pass
        torch.cuda.empty_cache()
    dig = [v for s in range(ns) for v in dig_s[s]]
    steps = steps_total
    walls = [sum(float(np.median(wl)) for wl in seg_walls)]       # one figure: the sum of the blocks' median pass times
    all_walls = [x for wl in seg_walls for x in wl]
    if args.dump_digests:
        json.dump(dig, open(args.dump_digests, "w"))
    wall = float(np.median(walls))
    px = float(F) * w * h
    res = {"metric": "Mpixels/s HDR quantize (4K PQ Lu'v' 11-bit), %d-frame stream block-sharded over the GPUs" % F,
           "value": round(px / wall / 1e6, 1), "unit": "Mpixels/s", "n_gpus": ns, "steps": steps, "warmup": 1,
           "ms_per_step": round(1e3 * wall / max(steps, 1), 4), "higher_is_better": True, "scaling": "strong",
           "vs_baseline": None, "dtype": "f32", "data": "synthetic",
           "config": dict(config_block(args, ns),
                          driver="one process, lumahip_multi_* (C ABI): one context + stream per GPU, table broadcast with RCCL: %s" % m.used_rccl()),
           "repeats": min(len(wl) for wl in seg_walls), "timed_seconds": round(sum(all_walls), 3), "segments": nseg,
           "ms_per_region_min_median_max": [round(1e3 * sum(min(wl) for wl in seg_walls), 3), round(1e3 * wall, 3),
                                            round(1e3 * sum(max(wl) for wl in seg_walls), 3)],
           "digests": {"gathered_in_stream_order": len(dig),
                       "stream_digest": "%016x" % (sum((i + 1) * v for i, v in enumerate(dig)) & 0xFFFFFFFFFFFFFFFF)},
           "placement": {"mode": "off (plain allocations)"}}
    exp = res["config"].get("expected_stream_digest")
    if exp is not None:
        res["digests"]["equals_expected"] = res["digests"]["stream_digest"] == exp
    m.close()
    return res
