"""One workload of the bench line: its resident stream, the timed legs (encode, decode, round trip, packed-layout decode,
float inputs, picture-like decode) and the roofline blocks of its encode and decode kernels."""
from __future__ import annotations

import json
import os

import numpy as np
import torch

from .plan import BYTES_PER_PIXEL, READ_BYTES_PER_PIXEL, CLOCK_GHZ, HBM_PEAK_GBS, N_SIMD, PACKED_RING, SEED, WORKLOADS, resident_frames, workload_text
from lumahdrv_amd.placement import CHUNK_BYTES

from .resident import ResidentStream
from .timing import Timer


def load_profile(path, workload, px_step, sha):
    """a committed rocprofv3-derived figure is only reported when it was captured from THESE kernel sources"""
    try:
        with open(path) as f:
            tj = json.load(f)
        ent = tj.get(workload) if workload in tj else tj
        if ent.get("kernel_source_sha") == sha and ent.get("workload", workload) == workload:
            return ent
    except Exception:
        pass
    return None


def run_workload(L, args, name, w, h, B, K, Wm, rank, world, local_rank, use_dist, dev, main, sha, pool=None, legs="full", nbatch=None):
    """one workload: resident synthetic stream, encode timed (plus decode / round trip for the main one), roofline blocks.
    legs: "full" = every leg; "encode" = the encode leg only, two lanes and ordered (the placement-off and small-pool comparisons).
    nbatch: batches of the resident stream when it is not the plan's (the small-pool leg: 2)."""
    from lumahdrv_amd.sharding import broadcast_quantizer
    ptf, bits, cs, bitsC, maxLum, minLum, sc, profile, desc, xf_desc, kname = WORKLOADS[name]
    cfg0 = lut0 = None
    if rank == 0:   # rank 0 builds the table on its host; RCCL broadcast of table + parameters over xGMI
        cfg0 = (ptf, bits, cs, bitsC, maxLum, minLum, sc, profile)
        lut0 = L.build_lut(ptf, bits, maxLum, minLum)
    cfg, lut = broadcast_quantizer(cfg0, lut0, dev)
    ptf, bits, cs, bitsC, maxLum, minLum, sc, profile = cfg
    ctx = L.Context(local_rank)
    ctx.set_stream(torch.cuda.current_stream().cuda_stream)
    ctx.set_quantizer(ptf, bits, cs, bitsC, maxLum, minLum, lut)

    n1 = w * h
    n3 = 3 * n1
    _, hs, st, _ = L.plane_geometry(w, h, profile)
    psz = [hs[p] * st[p] for p in range(3)]
    # the stream is the PLAN's (benchlib/plan.py: a function of the arguments); the pool only decides where its batches live
    if nbatch is None:
        nbatch = resident_frames(w, h, B, main) // B
    rs = ResidentStream(dev, w, h, B, nbatch, psz, pool, want_output=(legs == "full"),
                        striped_ok=(legs == "full" and args.decode_layout == "auto"), allow_short=args.allow_short_stream)
    nbatch, ptrs, striped, out_fs = rs.nbatch, rs.ptrs, rs.striped, rs.out_fs     # (nbatch is smaller only with --allow-short-stream)
    nfr = nbatch * B
    first = rank * nfr                                     # each rank has its own stream (weak scaling)
    for b in range(nbatch):
        ctx.synth_frames_device(ptrs(b)[0], n3, B, w, h, SEED, first + b * B)
    torch.cuda.synchronize()

    def enc(i):
        s, _, pl = ptrs(i % nbatch)
        ctx.encode_frames_device(s, n3, B, w, h, sc, profile, pl, st, psz)

    def dec(i):
        _, o, pl = ptrs(i % nbatch)
        ctx.decode_frames_device_planar(pl, st, psz, B, w, h, profile, sc, o, out_fs)

    lanes = max(0, args.lanes)
    tm = Timer(K, Wm, use_dist, dev, args.min_seconds, args.max_repeats, ctx, lanes)
    px_step = float(B) * w * h
    te = tm.run(enc)

    def rate(t):
        return world * K * px_step / t / 1e6

    def per_rank(t):
        v = sorted(1e3 * x / K for x in t["rank_wall_medians"])
        return {"min": round(v[0], 4), "median": round(float(np.median(v)), 4), "max": round(v[-1], 4)}

    r = {"value": round(rate(te["wall_median"]), 1), "unit": "Mpixels/s", "ms_per_step": round(1e3 * te["wall_median"] / K, 4),
         "ms_per_step_min": round(1e3 * te["wall_min"] / K, 4), "ms_per_step_max": round(1e3 * te["wall_max"] / K, 4),
         "ms_per_step_over_ranks": per_rank(te),
         "repeats": te["repeats"], "timed_seconds": round(te["seconds"], 3), "lanes": lanes,
         "workload": workload_text(name, w, h, B, nfr), "resident_frames": nfr, "config_degraded": rs.degraded,
         "resident_stream": rs.placement_report(),
         "frames_per_step": B, "width": w, "height": h, "preScaling": sc, "profile": profile,
         "distinct_input_GB_per_gpu": round(nfr * n3 * 4 / 1e9, 2)}
    if legs == "encode":
        if lanes:
            r["value_ordered"] = round(rate(tm.run(enc, lanes=0)["wall_median"]), 1)   # the same launches back to back on one stream
            r["frac_ordered"] = round(BYTES_PER_PIXEL * r["value_ordered"] * 1e6 / world / (HBM_PEAK_GBS * 1e9), 4)
        ctx.close()
        rs.close()
        return r, cfg
    td = tm.run(dec)                                       # (the planes every batch holds are the encode leg's)
    rb_after_random = ctx.rb_table_info(sc) if cs == 2 else None
    r["decode_mpix_s"] = round(rate(td["wall_median"]), 1)
    r["decode_output_layout"] = ("R, G, B planes of a batch in three HBM region groups (lumahip_decode_frames_device_planar)"
                                 if striped else "packed LumaFrame layout")
    if striped and rs.placed < nbatch:
        r["decode_output_layout"] += "; %d of %d batches in plain allocations (same layout, no group placement)" % (nbatch - rs.placed, nbatch)
    teo = tdo = None
    if lanes:
        teo = tm.run(enc, lanes=0)                         # the same launches back to back on one stream
        tdo = tm.run(dec, lanes=0)
        r["value_ordered"] = round(rate(teo["wall_median"]), 1)
        r["decode_mpix_s_ordered"] = round(rate(tdo["wall_median"]), 1)
    if main:
        # encode batch i, decode batch i: dependent, so ordered on one stream
        trt = tm.run(lambda i: (enc(i), dec(i)), lanes=0)
        r["roundtrip_mpix_s"] = round(rate(trt["wall_median"]), 1)
    if main and striped and pool is not None:
        # The reference's decoder returns the PACKED LumaFrame (include/luma/luma_frame.h:84-87: channel c at buffer + c*h*w), not
        # the three-buffer layout the legs above write: the same decode launches into packed frames, a ring of PACKED_RING
        # batches (>> the 256 MB MALL), once in chunks of the pool (the fastest float chunks left) and once in a plain allocation.
        pk = {}
        # "pool_rotating": batch b's packed frames in the b-th chunk the pool hands out in its ROTATING mode (region groups 0, 1,
        # 2, 0, ...), so that the launches in flight on the two lanes write different groups (profiles/r03_layout_lab.txt: 0.75)
        # "frame_rotating": the FRAMES of a batch rotate over three chunks of three region groups (frame f in chunk f % 3; every
        # frame still a packed LumaFrame) and the launch interleaves its tiles over the frames, so ONE launch writes all three
        # groups (lumahip_decode_frames_device_rotating): what the ordered figure of the packed layout can be
        # "library_ring": the caller lets the LIBRARY allocate (lumahip_decoded_ring_create: its own small pool, the frames of a batch
        # rotating over three region groups) -- the default a device-resident caller of the decoder gets; "caller_buffer": one
        # caller-owned plain allocation per ring of batches, what lumahip_decode_frames_device is handed otherwise
        for how in ("pool_placed", "pool_rotating", "frame_rotating", "library_ring", "caller_buffer"):
            ring = rot = frot = lring = None
            if how == "pool_placed":
                ring = pool.take_float(min(PACKED_RING, len(pool.float)))
                if len(ring) < 3:
                    pool.give_back(ring, [], [])
                    continue
            elif how == "pool_rotating":
                if min(len(g) for g in pool.striped) < PACKED_RING // 3:
                    continue
                rot = ring = pool.take_rotating(PACKED_RING)       # lumahip_pool_alloc(LUMAHIP_POOL_ROTATING): no group arithmetic here
            elif how == "frame_rotating":
                per = -(-B // 3)                                    # frames of a batch per chunk
                if min(len(g) for g in pool.striped) < PACKED_RING // 3 or 3 * per * n3 * 4 > CHUNK_BYTES:
                    continue
                frot = pool.take_striped(PACKED_RING // 3)          # [[group 0 chunks], [group 1 chunks], [group 2 chunks]]
            elif how == "library_ring":
                from lumahdrv_amd import capi
                try:
                    lring = capi.DecodedRing(ctx, PACKED_RING, B, w, h)
                except Exception:
                    continue
            plain = torch.empty(PACKED_RING * B * n3, dtype=torch.float32, device=dev) if how == "caller_buffer" else None
            nring = len(ring) if ring is not None else PACKED_RING

            def dec_packed(i, ring=ring, plain=plain, nring=nring, frot=frot, lring=lring):
                if lring is not None:
                    lring.decode(ptrs(i % nbatch)[2], st, psz, B, profile, sc, i % nring)
                    return
                if frot is not None:
                    k = i % nring
                    bases = [frot[g][k // 3].data_ptr() + (k % 3) * (-(-B // 3)) * n3 * 4 for g in range(3)]
                    ctx.decode_frames_device_rotating(ptrs(i % nbatch)[2], st, psz, B, w, h, profile, sc, bases, n3)
                    return
                o = ring[i % nring].data_ptr() if ring is not None else plain.data_ptr() + (i % nring) * B * n3 * 4
                ctx.decode_frames_device_planar(ptrs(i % nbatch)[2], st, psz, B, w, h, profile, sc, [o + k * n1 * 4 for k in range(3)], n3)

            tp = tm.run(dec_packed)
            tpo = tm.run(dec_packed, lanes=0) if lanes else tp

            def fr(ms):
                return round(BYTES_PER_PIXEL * px_step / (ms * 1e-3) / 1e9 / HBM_PEAK_GBS, 4)
            pk[how] = {"value": round(rate(tp["wall_median"]), 1), "value_ordered": round(rate(tpo["wall_median"]), 1),
                       "kernel_ms": round(tp["dev_ms_median"] / K, 4), "kernel_ms_ordered": round(tpo["dev_ms_median"] / K, 4),
                       "frac_ordered": fr(tpo["dev_ms_median"] / K), "frac_overlapped": fr(tp["dev_ms_median"] / K), "batches_in_ring": nring}
            if lring is not None:
                pk[how].update(placed_in_region_groups=lring.placed, groups_found=lring.groups)
                lring.close()
            if frot is not None:
                pool.give_back([], [], [], frot)
            elif rot is not None:
                pool.give_back_rotating(rot)
            elif ring is not None:
                pool.give_back(ring, [], [])
            del plain
            torch.cuda.empty_cache()
        pk["layout"] = "packed LumaFrame (include/luma/luma_frame.h:84-87), what LumaDecoder::decode() returns"
        pk["unit"] = "Mpixels/s; frac = algorithmic bytes / kernel_ms / 8 TB/s"
        r["decode_packed_layout"] = pk

    if rank == 0:
        # ---- roofline of the dominant kernel: hipEvents over the timed regions (median region / K), rank 0
        nprobe = max(5, min(25, nbatch))
        avg_ms = te["dev_ms_median"] / K
        iso = [ctx.time_launches(0, 1, ptrs(i % nbatch)[0], n3, B, w, h, sc, profile, ptrs(i % nbatch)[2], st, psz)
               for i in range(nprobe)]
        probe_ms = dprobe_ms = None
        if profile == 2 and w % 4 == 0:
            # the same loads and stores with no arithmetic: what the memory system gives each traffic mix on THIS box.
            # (the decode probe overwrites the decoded frames, the encode probe the planes; nothing reads them afterwards)
            dprobe_ms = float(np.median([ctx.probe_decode_traffic(ptrs(i % nbatch)[2], st, psz, B, w, h, ptrs(i % nbatch)[1], out_fs)
                                         for i in range(nprobe)]))
            probe_ms = float(np.median([ctx.probe_encode_traffic(ptrs(i % nbatch)[0], n3, B, w, h, ptrs(i % nbatch)[2], st, psz)
                                        for i in range(nprobe)]))
        tr = load_profile(os.path.join(args.profile_dir, "traffic_latest.json"), name, px_step, sha)
        if tr and tr.get("pixels_per_launch") != px_step:
            tr = None                                     # captured for a different launch size: not this launch's bytes

        def hbm_block(ms, ms_ordered, ms_iso, probe, traffic_key, kern):
            # `achieved` / `frac` price the kernel's OWN average launch duration: K launches back to back on one stream
            # (kernel_ms_ordered), which is what a rocprofv3 kernel trace of `--lanes 0` reports per launch
            # (profiles/*_kernel_stats_ordered.csv).  The default timed region overlaps launches on `lanes` streams; its window / K
            # is a throughput figure (it is what `value` is made of) and is reported as *_overlapped.
            over = BYTES_PER_PIXEL * px_step / (ms * 1e-3) / 1e9
            achieved = over if ms_ordered is None else BYTES_PER_PIXEL * px_step / (ms_ordered * 1e-3) / 1e9
            blk = {"bound": "hbm", "achieved": round(achieved, 1), "peak": HBM_PEAK_GBS, "unit": "GB/s",
                   "frac": round(achieved / HBM_PEAK_GBS, 4),
                   "frac_is": "algorithmic bytes / kernel_ms_ordered / peak" if ms_ordered is not None else "algorithmic bytes / kernel_ms / peak",
                   "achieved_overlapped": round(over, 1), "frac_overlapped": round(over / HBM_PEAK_GBS, 4),
                   "algorithmic_bytes_per_launch": BYTES_PER_PIXEL * px_step,
                   "kernel": kern, "kernel_ms": round(ms, 4),
                   "kernel_ms_is": ("hipEvent window over the K launches of a region / K; the launches overlap on %d streams "
                                    "(lumahip_begin_unordered)" % lanes) if lanes else "hipEvent window over K back-to-back launches / K",
                   "kernel_ms_ordered": None if ms_ordered is None else round(ms_ordered, 4),
                   "frac_ordered": None if ms_ordered is None else round(BYTES_PER_PIXEL * px_step / (ms_ordered * 1e-3) / 1e9 / HBM_PEAK_GBS, 4),
                   "kernel_ms_isolated_launch": None if ms_iso is None else round(ms_iso, 4),
                   "traffic": tr.get(traffic_key) if tr else None,
                   "traffic_source": ("rocprofv3 PMC passes of tools/profile_round.sh (FETCH_SIZE / WRITE_SIZE calibrated on the "
                                      "traffic-only probe of the same access pattern), captured from these kernel sources at commit "
                                      "%s: %s" % (tr.get("commit", "?"), tr.get("tag", "?")))
                                     if tr else "no PMC capture of the current kernel sources in profiles/ (null, not a stale figure)",
                   "traffic_only_ms": None if probe is None else round(probe, 4),
                   "frac_of_traffic_only_rate": None if probe is None else round(probe / (ms_ordered if ms_ordered else ms), 3)}
            # north_star words its target as a fraction of the HBM-READ roofline: the bytes the launch READS (encode: the 12 B/pixel
            # of float input, decode: the 3 B/pixel of planes) over the same durations and the same 8 TB/s
            rb = READ_BYTES_PER_PIXEL[0 if "k_encode" in kern else 1] * px_step
            blk["read_only"] = {"bytes_per_launch": rb, "frac": round(rb / ((ms_ordered if ms_ordered else ms) * 1e-3) / 1e9 / HBM_PEAK_GBS, 4),
                                "frac_overlapped": round(rb / (ms * 1e-3) / 1e9 / HBM_PEAK_GBS, 4)}
            return blk

        enc_blk = hbm_block(avg_ms, teo["dev_ms_median"] / K if teo else None, float(np.median(iso)), probe_ms,
                            "hbm_bytes_per_launch", kname)
        dkname = kname.replace("k_encode", "k_decode").replace("LDS threshold records", "table in LDS")
        diso = [ctx.time_launches(1, 1, ptrs(i % nbatch)[1][0], n3, B, w, h, sc, profile, ptrs(i % nbatch)[2], st, psz)
                for i in range(nprobe)] if not striped else None
        dec_blk = hbm_block(td["dev_ms_median"] / K, tdo["dev_ms_median"] / K if tdo else None,
                            float(np.median(diso)) if diso else None, dprobe_ms, "decode_hbm_bytes_per_launch", dkname)
        dec_blk["output_layout"] = r["decode_output_layout"]
        mix = load_profile(os.path.join(args.profile_dir, "valu_mix_latest.json"), name, px_step, sha)
        def valu_block(m, ms, hbm_blk):
            """VALU-issue roofline of a YCbCr kernel from the PMC instruction mix of these kernel sources (None without one)"""
            peak = N_SIMD * CLOCK_GHZ
            blk = {"bound": "valu", "achieved": None, "peak": round(peak, 1), "unit": "G SIMD-issue-cycles/s", "frac": None, "hbm": hbm_blk}
            if m:
                cyc = m["issue_cycles_per_launch"] * (px_step / m["pixels_per_launch"])
                ach = cyc / (ms * 1e-3) / 1e9
                blk.update({"achieved": round(ach, 1), "frac": round(ach / peak, 4), "valu_instructions_per_pixel": m.get("valu_per_pixel"),
                            "fp64_instructions_per_pixel": m.get("fp64_per_pixel"),
                            "frac_is": "PMC class counters x measured issue costs / kernel time, against 1024 SIMDs x 2.4 GHz (nominal clock)"})
            else:
                blk["note"] = "no instruction-mix capture of the current kernel sources in profiles/"
            return blk
        dec_ms_own = (tdo["dev_ms_median"] if tdo else td["dev_ms_median"]) / K
        half = ctx.half_table_info(sc) if cs == 2 else None
        if cs == 2 and half["used"] and half["table_launches"] > 0 and half["backoff_launches"] == 0:
            # YCbCr encode on the half-input table (the synthetic stream, like every EXR frame of the reference, holds binary16
            # values): three LDS gathers instead of six powf per pixel -- HBM-bound like the Lu'v' kernels.  Decode has no such
            # table (its powf arguments depend on (Y', Cr) / (Y', Cb) pairs) and stays VALU-bound.
            r["roofline"] = enc_blk
            r["roofline"]["kernel"] = "lh::k_encode<CS_YCBCR,4:2:0,VW=4,LM=6: composite records + half-input table in LDS>"
            r["roofline"]["half_input_table"] = half
            if mix:
                r["roofline"]["valu_instructions_per_pixel"] = mix.get("valu_per_pixel")
                r["roofline"]["fp64_instructions_per_pixel"] = mix.get("fp64_per_pixel")
            r["roofline"]["decode_achieved_GBs"] = dec_blk["achieved"]
            r["decode_roofline"] = valu_block(mix.get("decode") if mix else None, dec_ms_own, dec_blk)
        elif cs == 2:
            # YCbCr without the table: VALU-issue-bound.  Issue cycles per pixel = sum over instruction classes of (PMC instruction count x
            # issue cost measured by tools/bench/valu_bench.hip: fp32 / int32 2 cycles per wave64 instruction, fp64 4,
            # conversions / compares / selects / min / max 4, transcendental 8); peak = every SIMD issuing every cycle.
            peak = N_SIMD * CLOCK_GHZ                                  # G SIMD-cycles / s
            common = {k: enc_blk[k] for k in ("kernel", "kernel_ms", "kernel_ms_is", "kernel_ms_ordered", "kernel_ms_isolated_launch",
                                              "traffic", "traffic_source", "traffic_only_ms", "frac_of_traffic_only_rate")}
            hbm = {k: enc_blk[k] for k in ("achieved", "peak", "unit", "frac", "algorithmic_bytes_per_launch")}
            if mix:
                cyc = mix["issue_cycles_per_launch"] * (px_step / mix["pixels_per_launch"])   # SIMD-cycles of VALU issue
                ach = cyc / (avg_ms * 1e-3) / 1e9                      # G SIMD-cycles / s actually spent issuing VALU
                r["roofline"] = dict({"bound": "valu", "achieved": round(ach, 1), "peak": round(peak, 1),
                                      "unit": "G SIMD-issue-cycles/s", "frac": round(ach / peak, 4),
                                      "valu_instructions_per_pixel": mix.get("valu_per_pixel"),
                                      "fp64_instructions_per_pixel": mix.get("fp64_per_pixel"), "hbm": hbm}, **common)
            else:
                r["roofline"] = dict({"bound": "valu", "achieved": None, "peak": round(peak, 1), "unit": "G SIMD-issue-cycles/s",
                                      "frac": None, "note": "no instruction-mix capture of the current kernel sources in profiles/",
                                      "hbm": hbm}, **common)
            r["decode_roofline"] = valu_block(mix.get("decode") if mix else None, dec_ms_own, dec_blk)
        else:
            r["roofline"] = enc_blk
            r["roofline"]["decode_achieved_GBs"] = dec_blk["achieved"]
            r["decode_roofline"] = dec_blk
    if cs == 2 and rank == 0 and world == 1 and not (args.no_float_inputs and args.no_decode_coherent):
        # ---- the same stream when its values are NOT binary16 (the reference's PFS pipe hands the encoder arbitrary floats,
        # src/pfs_interface.cpp:57-113): full-precision mantissas in every value, and in 1e-3 of the pixels, through the DEFAULT
        # policy (lumahip_tune half_table 1; lumahip_core.hip half_policy).  Last leg of the workload: it rewrites the stream.
        from lumahdrv_amd.placement import as_tensor
        gen = torch.Generator(device=dev)
        gen.manual_seed(SEED)
        mixf = mix.get("encode_float") if mix else None

        def perturb(frac):
            for b in range(nbatch):
                v = as_tensor(ptrs(b)[0], B * n3 * 4, dev).view(torch.int32).view(B, 3, n1)
                for f in range(B):                                  # per frame: small temporaries
                    noise = torch.randint(1, 1 << 13, (3, n1), device=dev, dtype=torch.int32, generator=gen)
                    if frac < 1.0:
                        noise *= (torch.rand(n1, device=dev, generator=gen) < frac).to(torch.int32)[None]
                    v[f] |= noise
            torch.cuda.synchronize()

        def float_leg(frac, what):
            perturb(frac)
            ctx.tune("half_table", 1)                               # the policy starts afresh, as for a new stream
            i0 = ctx.half_table_info(sc)
            tf = tm.run(enc)
            tfo = tm.run(enc, lanes=0) if lanes else tf
            i1 = ctx.half_table_info(sc)
            ms_own = tfo["dev_ms_median"] / K
            blk = {"value": round(rate(tf["wall_median"]), 1), "value_ordered": round(rate(tfo["wall_median"]), 1), "unit": "Mpixels/s",
                   "inputs": what, "policy": "default (lumahip_tune half_table 1): table launches report float data, the per-pixel "
                                               "kernel k_encode<CS_YCBCR,4:2:0,VW=4,LM=5> takes the launches of a back-off",
                   "kernel_ms": round(tf["dev_ms_median"] / K, 4), "kernel_ms_ordered": round(ms_own, 4),
                   "table_launches": i1["table_launches"] - i0["table_launches"],
                   "backoff_launches": i1["backoff_launches"] - i0["backoff_launches"],
                   "hbm_frac": round(BYTES_PER_PIXEL * px_step / (ms_own * 1e-3) / 1e9 / HBM_PEAK_GBS, 4)}
            return blk, ms_own

        if legs == "full" and not args.no_decode_coherent:
            # ---- decode of a PICTURE-like stream.  The legs above decode the planes of the synthetic stream, whose pixels are
            # unrelated: there no wave finds its codes local, the red / blue tables are never read (six powf per pixel,
            # and the launch-level policy soon picks the kernels without the test).  The same stream low-pass filtered in the log
            # domain (32 x 32 box, bilinear up: neighbouring pixels, neighbouring codes -- what video looks like) takes the tables.
            import torch.nn.functional as F
            for b in range(nbatch):
                v = as_tensor(ptrs(b)[0], B * n3 * 4, dev).view(torch.float32).view(B * 3, 1, h, w)
                for i in range(B * 3):
                    lo = F.avg_pool2d(torch.log(v[i:i + 1]), 32)
                    v[i:i + 1] = torch.exp(F.interpolate(lo, size=(h, w), mode="bilinear", align_corners=False))
                enc(b)
            torch.cuda.synchronize()
            ctx.tune("ycbcr_rb_tables", 1)                          # the policy starts afresh, as for a new stream
            j0 = ctx.rb_table_info(sc)
            tc = tm.run(dec)
            tco = tm.run(dec, lanes=0) if lanes else tc
            j1 = ctx.rb_table_info(sc)
            cms = tco["dev_ms_median"] / K
            r["decode_coherent"] = {
                "value": round(rate(tc["wall_median"]), 1), "value_ordered": round(rate(tco["wall_median"]), 1), "unit": "Mpixels/s",
                "inputs": "the planes of the same stream low-pass filtered in the log domain (32 x 32 box): a picture's statistics",
                "kernel": "lh::k_decode<CS_YCBCR,4:2:0,VW=4,y table in LDS,red / blue tables in global memory>: a wave whose codes are "
                          "local reads red and blue (two 4-byte gathers) and computes green (two powf); others compute all three",
                "kernel_ms": round(tc["dev_ms_median"] / K, 4), "kernel_ms_ordered": round(cms, 4),
                "hbm_frac": round(BYTES_PER_PIXEL * px_step / (cms * 1e-3) / 1e9 / HBM_PEAK_GBS, 4),
                "rb_table_bytes": j1["bytes"], "table_launches": j1["table_launches"] - j0["table_launches"],
                "backoff_launches": j1["backoff_launches"] - j0["backoff_launches"]}
            r["decode_random_rb_policy"] = {k: rb_after_random[k] for k in ("table_launches", "backoff_launches")}
            for b in range(nbatch):
                ctx.synth_frames_device(ptrs(b)[0], n3, B, w, h, SEED, first + b * B)
        if not args.no_float_inputs:
            fblk, fms = float_leg(1.0, "every value with a full-precision mantissa (13 random low bits): no binary16 value in the stream")
            fblk["roofline"] = valu_block(mixf, fms, {k: enc_blk[k] for k in ("peak", "unit", "algorithmic_bytes_per_launch")})
            fblk["roofline"]["hbm"]["achieved"] = round(BYTES_PER_PIXEL * px_step / (fms * 1e-3) / 1e9, 1)
            fblk["roofline"]["hbm"]["frac"] = fblk["hbm_frac"]
            r["float_inputs"] = fblk
            # (the stream above is all floats already; a fresh synthetic stream for the 1e-3 point)
            for b in range(nbatch):
                ctx.synth_frames_device(ptrs(b)[0], n3, B, w, h, SEED, first + b * B)
            mblk, _ = float_leg(1e-3, "1e-3 of the pixels with full-precision mantissas in all three channels, the rest binary16 values")
            r["mixed_inputs_1e-3"] = mblk
    ctx.close()
    rs.close()
    return r, cfg
