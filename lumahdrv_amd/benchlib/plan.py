"""The bench workload as a pure function of the command line (bench.py `--plan-only` prints it).

Everything a run's `config` block says -- which BASELINE configuration, how many frames every rank keeps resident in HBM,
how they are cut into steps -- is decided HERE, from the arguments alone, before any GPU, pool or process-group work.  Free
memory never changes the workload: a run that cannot hold its plan fails loudly (bench.py exit code 3) unless
`--allow-short-stream` was given, and then the line carries `"config_degraded": true`.  Where buffers live (chunk pool or
plain allocations) is an optimisation decided at run time and reported under `placement`, never under `config`.
"""
from __future__ import annotations

import json

W4K, H4K = 3840, 2160
W8K, H8K = 7680, 4320
SEED = 20250929
BYTES_PER_PIXEL = 15.0      # 12 B read (3 x fp32) + 3 B written (Y 2 B + U 0.5 B + V 0.5 B), SURVEY.md 8(d)
READ_BYTES_PER_PIXEL = (12.0, 3.0)   # of those, what a launch READS: encode 3 x fp32, decode Y 2 B + U 0.5 B + V 0.5 B (profile 2)
HBM_PEAK_GBS = 8000.0       # MI355X_MICROARCH.md: 8.0 TB/s spec
N_SIMD, CLOCK_GHZ = 1024, 2.4   # 256 CUs x 4 SIMDs, max clock (MI355X_MICROARCH.md chip-level parameters)
PACKED_RING = 6             # batches of packed LumaFrames the packed-layout decode leg cycles through (6 x 20 x 4K = 12 GB)
MAIN_STREAM_FRAMES = 500    # BASELINE configs[1]: "500-frame synthetic stream"
MAIN_STREAM_BYTES = 50e9    # larger frames than 4K: as many as the same input bytes hold
OTHER_STREAM_BYTES = 4e9    # the other workloads of the line: >= 4 GB of distinct input (>> the 256 MB MALL), >= 8 steps

WORKLOADS = {
    # name: (ptf, bits, cs, bitsC, maxLum, minLum, preScaling, profile, description, transform, kernel)
    "pq11_luv": (1, 11, 0, 8, 1e4, 0.005, 1.0, 2, "PQ 11-bit Lu'v' 8-bit chroma, profile 2 (4:2:0 16-bit)",
                 "RGB->XYZ->Lu'v'", "lh::k_encode<CS_LUV,4:2:0,VW=4,LDS threshold records>"),
    "pq10_ycbcr": (1, 10, 2, 10, 1000.0, 0.01, 20.0, 2,
                   "HDR10 recipe: PQ 10-bit YCbCr BT.2020 10-bit chroma, max/min 1000/0.01, preScaling 20",
                   "RGB->PQ->Y'CbCr (binary16 inputs: R'G'B' from the half-input table in LDS, luminance code from the composite "
                   "records; HBM-bound)",
                   "lh::k_encode<CS_YCBCR,4:2:0,VW=4,LDS threshold records>"),
    "log12_luv": (2, 12, 0, 8, 1e4, 0.005, 1.0, 2, "LOG 12-bit Lu'v' 8-bit chroma, profile 2",
                  "RGB->XYZ->Lu'v'", "lh::k_encode<CS_LUV,4:2:0,VW=4,LDS threshold records>"),
}

# the other single-GPU configurations of BASELINE.json carried by the default line: key -> (workload, width, height, frames/step)
OTHER_WORKLOADS = {"pq10_ycbcr_4k": ("pq10_ycbcr", W4K, H4K, 20), "log12_luv_8k": ("log12_luv", W8K, H8K, 5)}

# stream digests that are known constants of (workload, width, height, frames, SEED): independent of N and of the driver
# (profiles/r02_stream2000_n1.json; tests/test_gpu_baseline_configs.py checks four of its frames against the oracle)
KNOWN_STREAM_DIGESTS = {("pq11_luv", W4K, H4K, 2000): "54051a63ee9b1773"}


def geometry(w, h, profile=2):
    """(floats per frame, bytes of the Y / U / V planes of one frame) -- the plane layout of vpx_img_alloc(fmt(profile), w, h, 32)"""
    from lumahdrv_amd import capi
    _, hs, st, _ = capi.plane_geometry(w, h, profile)
    return 3 * w * h, [hs[p] * st[p] for p in range(3)]


def resident_frames(w, h, B, main):
    """frames of the resident stream one rank holds: configs[1]'s 500 for the headline workload (at larger frames what the same
    input bytes hold), >= 8 steps and >= 4 GB of input for the others; always whole steps of B frames"""
    n3 = 3 * w * h
    if main:
        want = min(MAIN_STREAM_FRAMES, max(B, int(MAIN_STREAM_BYTES // (n3 * 4)) // B * B))
    else:
        want = max(8 * B, int(OTHER_STREAM_BYTES // (n3 * 4)) // B * B)
    return max(1, want // B) * B


def pool_request(w, h, B, decode_layout="auto", nbatches=None, with_output=True):
    """what a resident stream of `nbatches` batches (default: the headline workload's) asks of the chunk pool: chunks for float
    frames / Y planes / U+V planes / per region group for striped decode output.  None when a batch does not fit a chunk (the
    2 GiB chunks hold a 20-frame 4K batch; larger batches use plain allocations)."""
    from lumahdrv_amd.placement import CHUNK_BYTES, slots
    n3, psz = geometry(w, h)
    n1 = n3 // 3
    ypc, _ = slots(CHUNK_BYTES, B * psz[0])
    uvpc, _ = slots(CHUNK_BYTES, B * psz[1] + (1 << 20) + B * psz[2])
    spc, _ = slots(CHUNK_BYTES, B * n1 * 4)
    if B * n3 * 4 > CHUNK_BYTES or ypc < 1 or uvpc < 1:
        return None
    nb = nbatches if nbatches else resident_frames(w, h, B, True) // B
    stripe = with_output and decode_layout == "auto" and spc >= 1
    return {"batches": nb, "n_float": nb * (2 if (with_output and not stripe) else 1) + (PACKED_RING if stripe else 0),   # + the packed-layout decode leg
            "n_y": -(-nb // ypc), "n_uv": -(-nb // uvpc), "n_striped": (-(-nb // spc) + PACKED_RING // 3 if stripe else 0),
            "chunk_bytes": CHUNK_BYTES, "striped_output": bool(stripe)}


def workload_plan(name, w, h, B, main, with_output=True):
    """the resident stream of one workload on one rank: frames, steps per pass, bytes (input + decoded output + planes)"""
    profile = WORKLOADS[name][7]
    n3, psz = geometry(w, h, profile)
    frames = resident_frames(w, h, B, main)
    per_frame = n3 * 4 * (2 if with_output else 1) + sum(psz)
    return {"workload": name, "width": w, "height": h, "frames_per_step": B, "resident_frames": frames, "steps_per_pass": frames // B,
            "bytes_resident": frames * per_frame, "input_bytes": frames * n3 * 4, "plane_bytes": frames * sum(psz)}


def workload_text(name, w, h, B, frames):
    desc, xf = WORKLOADS[name][8], WORKLOADS[name][9]
    return ("%dx%d %s encode (%s, LUT quantize, 4:2:0 16-bit pack), %d frames/step, %d-frame resident stream per GPU"
            % (w, h, desc, xf, B, frames))


def config_block(args, n_gpus):
    """the `config` object of the JSON line -- a function of the arguments and the GPU count only.  Scalar keys first
    (`resident_frames`, `stream_frames`: a consumer that truncates the `workload` text still sees the stream sizes)."""
    w, h, B, name = args.width, args.height, args.frames_per_step, args.workload
    sc, profile = WORKLOADS[name][6], WORKLOADS[name][7]
    n3 = 3 * w * h
    if args.stream_frames > 0:
        from lumahdrv_amd.sharding import shard_range
        F = args.stream_frames
        per_rank = [len(shard_range(F, r, n_gpus)) for r in range(n_gpus)]
        return {"resident_frames": per_rank[0], "stream_frames": F, "resident_frames_per_rank": per_rank,
                "workload": "%dx%d %s encode, ONE %d-frame stream sharded in contiguous blocks (%d frames on rank 0), %d frames/step"
                            % (w, h, WORKLOADS[name][8], F, per_rank[0], B),
                "frames_per_step": B, "width": w, "height": h, "preScaling": sc, "profile": profile,
                "parallelism": "frame-sharded x%d" % n_gpus, "scaling": "strong",
                "expected_stream_digest": KNOWN_STREAM_DIGESTS.get((name, w, h, F))}
    frames = resident_frames(w, h, B, True)
    return {"resident_frames": frames, "stream_frames": frames * n_gpus, "resident_frames_per_rank": [frames] * n_gpus,
            "workload": workload_text(name, w, h, B, frames),
            "timed": "the quantize (encode) pass: K steps per region, region repeated until >= %.1f s of device time; "
                     "value / ms_per_step = the median region; decode and encode+decode round trip are timed "
                     "the same way and reported as decode_mpix_s / roundtrip_mpix_s" % args.min_seconds,
            "frames_per_step": B, "width": w, "height": h, "preScaling": sc, "profile": profile,
            "parallelism": "frame-sharded x%d" % n_gpus, "scaling": "weak",
            "distinct_input_GB_per_gpu": round(frames * n3 * 4 / 1e9, 2)}


def stream_shard_plan(F, rank, world, w, h, B, free, placement="auto", profile=2):
    """--stream-frames mode (BASELINE configs[4]): what `rank` holds of the ONE F-frame stream -- its block, the bytes resident in
    its HBM, whether they fit, and whether the shard is carved from the chunk pool (run_stream takes every decision from here).
    The block is a function of (F, rank, world) only; `free` decides in how many consecutive resident segments it is encoded."""
    from lumahdrv_amd.sharding import shard_range
    n3, psz = geometry(w, h, profile)
    mine = shard_range(F, rank, world)
    nfr = len(mine)
    per_frame = n3 * 4 + sum(psz)
    resident = per_frame * max(nfr, 1)
    steps = (nfr + B - 1) // B
    # the pool when the shard is worth probing for (a few GB are not) and its chunks fit what is free: a 2000-frame shard (N = 1:
    # 130 of the ~140 chunks of a 288 GB GPU) takes its float chunks from all three region groups -- half of its batches then read
    # where their planes are written, which is still no worse than what plain allocations pair at random
    req = pool_request(w, h, B, nbatches=steps, with_output=False) if (placement == "auto" and steps and per_frame * nfr >= 8e9) else None
    if req is not None:
        need = req["n_float"] + req["n_y"] + req["n_uv"]
        if need * req["chunk_bytes"] > free - (8 << 30):
            req = None
    fits = resident <= free * 0.9
    segments = 1 if fits else int(-(-resident // int(free * 0.8)))      # blocks of at most 80 % of the free HBM
    if segments > 1:
        req = None                                                        # (each block plans for itself when its turn comes)
    return {"rank": rank, "first_frame": mine.start, "frames": nfr, "steps": steps, "bytes_resident": resident,
            "input_bytes": n3 * 4 * nfr, "plane_bytes": sum(psz) * nfr, "free_bytes": int(free), "fits": fits, "segments": segments,
            "one_batch_fits": per_frame * min(B, max(nfr, 1)) <= free * 0.8,
            "pool": req, "placement": ("chunk pool" if req else "plain allocations") if segments == 1 else "per block"}


def run_plan(args, free=280e9):
    """what `bench.py <args>` would run, as a dict: the `config` block of its JSON line and every rank's resident stream.
    `free` (bytes of free HBM per GPU) only says whether the plan FITS and where its buffers would come from."""
    w, h, B, N = args.width, args.height, args.frames_per_step, args.gpus
    cfg = config_block(args, N)
    ranks = []
    if args.stream_frames > 0:
        ranks = [stream_shard_plan(args.stream_frames, r, N, w, h, B, free, args.placement) for r in range(N)]
        mode = "ONE %d-frame stream block-sharded over %d rank(s) (strong scaling)" % (args.stream_frames, N)
    else:
        wp = workload_plan(args.workload, w, h, B, True)
        n3 = 3 * w * h
        req = pool_request(w, h, B, args.decode_layout) if args.placement == "auto" else None
        frames = wp["resident_frames"]
        resident = wp["bytes_resident"] + ((PACKED_RING * B * n3 * 4) if req and req["striped_output"] else 0)
        pool_bytes = (req["n_float"] + req["n_y"] + req["n_uv"] + 3 * req["n_striped"]) * req["chunk_bytes"] if req else 0
        if pool_bytes > free - (6 << 30):
            req, pool_bytes = None, 0                   # (the pool would not get its chunks: the same frames in plain allocations)
        for r in range(N):
            ranks.append({"rank": r, "first_frame": r * frames, "frames": frames, "steps_per_pass": frames // B,
                          "bytes_resident": resident, "pool_bytes": pool_bytes, "free_bytes": int(free),
                          "fits": max(resident, pool_bytes) <= free * 0.9, "pool": req,
                          "placement": "chunk pool" if req else "plain allocations"})
        mode = "every rank its own %d-frame stream (weak scaling)" % frames
    ok = all(r["fits"] or r.get("one_batch_fits") for r in ranks)      # (a stream shard that does not fit at once is encoded in blocks)
    return {"plan_only": True, "n_gpus": N, "mode": mode, "scaling": cfg["scaling"], "width": w, "height": h, "frames_per_step": B,
            "resident_frames": cfg["resident_frames"], "stream_frames": cfg["stream_frames"], "config": cfg,
            "expected_stream_digest": cfg.get("expected_stream_digest"),
            "collective": "one broadcast of the table (2^bits floats) + an 8-value parameter block from rank 0; none on the data path",
            "fits": ok, "ranks": ranks}


def plan_only(args):
    """`bench.py --gpus N --plan-only`: the per-rank plan of the run the same command line would make, without touching a GPU.
    The launcher's environment (RANK / WORLD_SIZE / LOCAL_RANK) is not read: the plan is the arguments'."""
    if args.hbm_free_gb > 0:
        free = args.hbm_free_gb * 1e9
    else:
        import torch
        free = float(torch.cuda.mem_get_info(0)[0]) if torch.cuda.is_available() else 280e9
    d = run_plan(args, free)
    print(json.dumps(d))
    return 0 if d["fits"] else 1
