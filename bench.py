#!/usr/bin/env python3
"""bench.py -- Mpixels/s of the HDR quantize hot path (4K PQ 11-bit Lu'v', VP9 profile 2) on MI355X.

    python bench.py --gpus N --steps K --warmup W          (N > 1 without WORLD_SIZE: re-launches itself under
                                                             torch.distributed.run with N ranks, one per GPU)
    python -m torch.distributed.run --nnodes=1 --nproc-per-node N --master-addr 127.0.0.1 --master-port P \
        bench.py --gpus N --steps K --warmup W

A *step* is one pass of the fused encode kernel (RGB -> XYZ -> Lu'v' -> PQ quantize -> 4:2:0 16-bit planes,
= LumaEncoder::encode minus the VP9 stage) over one batch of `--frames-per-step` synthetic 3840x2160 frames that
are already resident in HBM.  The resident stream is BASELINE.json configs[1]'s 500 frames (25 batches of 20;
fewer when memory is short); steps walk it cyclically, so every step reads 2 GB that the previous ~24 steps evicted
from L2 / Infinity Cache.  W warm-up steps, then EXACTLY K steps between barrier + synchronize, MAX over ranks.  That
K-step region is repeated until >= --min-seconds of device time has been spent (sustained clocks, not a burst);
`ms_per_step` / `value` are the MEDIAN region, min / max are reported next to it.  Rank 0 prints ONE JSON line.

Multi-GPU: frames are independent, so rank r owns its own block of frame indices (weak scaling, no data-path
collective); the only communication is one RCCL broadcast of the transfer-function table and the quantizer
parameters from rank 0 before the timed region.  `--stream-frames F` switches to BASELINE configs[4]'s mode: ONE
F-frame stream (default use: 2000) block-sharded over the ranks (250 per GPU at N = 8, strong scaling), per-frame
plane digests all_gathered in stream order and spot-checked by rank 0.

The K launches of a region are independent batches, so they run inside ONE unordered section of the C ABI
(lumahip_begin_unordered ... lumahip_end_unordered, `--lanes`, default 2: successive batches alternate between two
internal streams, so one batch's ramp-up and tail overlap its neighbours' steady state).  The hipEvent pair brackets the
whole section on the context's stream; `kernel_ms` = that window / K.  `--lanes 0` (and the `*_ordered` fields of the default
run) is the same K launches back to back on one stream.

Extra fields in the same line (not part of `value`): decode (its own `decode_roofline` block with a decode-shaped
traffic-only probe) and encode+decode round trip, the roofline of the encode kernel (HIP events on the launch stream, live),
`value_placement_off` (the same kernels on plainly allocated buffers), `other_workloads` (BASELINE configs[2] HDR10/YCbCr at
4K and configs[3] LOG-12 at 7680x4320, each with its own roofline block and its decode rate; N = 1 only), `facade_hostfed`
(LumaEncoder::encode(LumaFrame*) end to end on host frames -- PCIe-bound, never `value`) and the CPU reference timed on this
host (`cpu_baseline`, N = 1 only).  The YCbCr workload's `value` is measured on the synthetic stream, whose values are
binary16-exact like every EXR frame of the reference (half-input table, HBM-bound); `float_inputs` and `mixed_inputs_1e-3` of the
same block are the same stream with full-precision mantissas (what the reference's PFS pipe delivers; per-pixel powf, VALU-bound)
and with 1e-3 of the pixels so, through the default kernel policy.  `--gpus N [--stream-frames F] --plan-only` prints every rank's
shard, resident bytes and pool chunks without touching a GPU.
"""
import argparse
import json
import os
import socket
import subprocess
import sys
import time

ROOT = os.path.dirname(os.path.abspath(__file__))
if ROOT not in sys.path:
    sys.path.insert(0, ROOT)

import numpy as np  # noqa: E402
import torch  # noqa: E402
import torch.distributed as dist  # noqa: E402

W4K, H4K = 3840, 2160
W8K, H8K = 7680, 4320
SEED = 20250929
BYTES_PER_PIXEL = 15.0      # 12 B read (3 x fp32) + 3 B written (Y 2 B + U 0.5 B + V 0.5 B), SURVEY.md 8(d)
HBM_PEAK_GBS = 8000.0       # MI355X_MICROARCH.md: 8.0 TB/s spec
N_SIMD, CLOCK_GHZ = 1024, 2.4   # 256 CUs x 4 SIMDs, max clock (MI355X_MICROARCH.md chip-level parameters)


def parse():
    ap = argparse.ArgumentParser()
    ap.add_argument("--gpus", type=int, default=1)
    ap.add_argument("--steps", type=int, default=25, help="steps per timed region (default 25 = one pass over the 500-frame stream)")
    ap.add_argument("--warmup", type=int, default=3)
    ap.add_argument("--frames-per-step", type=int, default=20)
    ap.add_argument("--width", type=int, default=W4K)
    ap.add_argument("--height", type=int, default=H4K)
    ap.add_argument("--workload", default="pq11_luv", choices=["pq11_luv", "pq10_ycbcr", "log12_luv"])
    ap.add_argument("--min-seconds", type=float, default=1.0,
                    help="repeat the K-step timed region until this much device time has been measured (per leg)")
    ap.add_argument("--max-repeats", type=int, default=400)
    ap.add_argument("--stream-frames", type=int, default=0,
                    help="BASELINE configs[4] mode: ONE stream of this many frames (2000) block-sharded over the ranks")
    ap.add_argument("--lanes", type=int, default=2,
                    help="streams of the unordered section the K launches of a region run in (0 = one stream, launches back to back)")
    ap.add_argument("--driver", default="torch", choices=["torch", "multi"],
                    help="--stream-frames mode: 'torch' = one process per GPU under torch.distributed (RCCL); 'multi' = ONE process "
                         "driving every visible GPU through the C ABI's lumahip_multi_* layer (table broadcast with RCCL from C++)")
    ap.add_argument("--dump-digests", default="", help="--stream-frames mode: write the per-frame digests (stream order) to this JSON file")
    ap.add_argument("--decode-layout", default="auto", choices=["auto", "packed"],
                    help="auto: with --placement auto the decoded frames are written with their R, G, B planes in three HBM region "
                         "groups (lumahip_decode_frames_device_planar); packed: the LumaFrame layout always")
    ap.add_argument("--no-facade-hostfed", action="store_true")
    ap.add_argument("--no-other-workloads", action="store_true")
    ap.add_argument("--no-placement-off", action="store_true", help="skip the value_placement_off leg (the same kernels on plain allocations)")
    ap.add_argument("--no-cpu-baseline", action="store_true")
    ap.add_argument("--no-decode-coherent", action="store_true",
                    help="skip the YCbCr workload's decode leg on a picture-like stream (the red / blue tables are read)")
    ap.add_argument("--no-float-inputs", action="store_true",
                    help="skip the float-input legs of the YCbCr workload (the same stream with full-precision mantissas / 1e-3 of them)")
    ap.add_argument("--cpu-frames", type=int, default=16, help="frames the CPU baseline encodes (bounded sample, ~10 s on 1 thread)")
    ap.add_argument("--placement", default="auto", choices=["auto", "off"],
                    help="auto: device memory is taken in 2 GiB chunks, their region groups found with traffic-only launches, and the "
                         "Y planes of the resident stream live in another group than its other buffers (lumahdrv_amd/placement.py); "
                         "off: plain allocations")
    ap.add_argument("--plan-only", action="store_true",
                    help="print what every rank of `--gpus N` would hold (shard, resident bytes, pool chunks) as one JSON line and exit; "
                         "needs no GPU (then --hbm-free-gb says how much memory a rank has); exit code 1 when some rank does not fit")
    ap.add_argument("--hbm-free-gb", type=float, default=0.0,
                    help="--plan-only: free HBM per GPU in GB (default: what GPU 0 reports, or 280 of the MI355X's 288 GB without a GPU)")
    ap.add_argument("--profile-dir", default=os.path.join(ROOT, "profiles"),
                    help="where traffic_latest.json / valu_mix_latest.json (tools/summarize_profile.py) live")
    return ap.parse_args()


PACKED_RING = 6          # batches of packed LumaFrames the packed-layout decode leg cycles through (6 x 20 x 4K = 12 GB)

WORKLOADS = {
    # name: (ptf, bits, cs, bitsC, maxLum, minLum, preScaling, profile, description, transform, kernel)
    "pq11_luv": (1, 11, 0, 8, 1e4, 0.005, 1.0, 2, "PQ 11-bit Lu'v' 8-bit chroma, profile 2 (4:2:0 16-bit)",
                 "RGB->XYZ->Lu'v'", "lh::k_encode<CS_LUV,4:2:0,VW=4,LDS threshold records>"),
    "pq10_ycbcr": (1, 10, 2, 10, 1000.0, 0.01, 20.0, 2, "HDR10 recipe: PQ 10-bit YCbCr BT.2020 10-bit chroma, max/min 1000/0.01, preScaling 20",
                   "RGB->PQ->Y'CbCr (binary16 inputs: R'G'B' from the half-input table in LDS, luminance code from the composite records; HBM-bound)",
                   "lh::k_encode<CS_YCBCR,4:2:0,VW=4,LDS threshold records>"),
    "log12_luv": (2, 12, 0, 8, 1e4, 0.005, 1.0, 2, "LOG 12-bit Lu'v' 8-bit chroma, profile 2",
                  "RGB->XYZ->Lu'v'", "lh::k_encode<CS_LUV,4:2:0,VW=4,LDS threshold records>"),
}


def free_port():
    s = socket.socket()
    s.bind(("127.0.0.1", 0))
    p = s.getsockname()[1]
    s.close()
    return p


def respawn_command(gpus, argv, port):
    """the launcher line `python bench.py --gpus N` turns itself into: one rank per GPU of ONE node, rendezvous on 127.0.0.1"""
    return [sys.executable, "-m", "torch.distributed.run", "--nnodes=1", "--nproc-per-node", str(gpus),
            "--master-addr", "127.0.0.1", "--master-port", str(port), os.path.abspath(__file__)] + list(argv)


def respawn(args):
    """`python bench.py --gpus N` with no launcher environment: run the same command line under torch.distributed.run,
    one rank per GPU, and pass its single JSON line through."""
    have = torch.cuda.device_count() if torch.cuda.is_available() else 0
    if have < args.gpus:
        raise SystemExit("bench.py --gpus %d: only %d GPU(s) visible; refusing to report a %d-GPU number from fewer devices"
                         % (args.gpus, have, args.gpus))
    cmd = respawn_command(args.gpus, sys.argv[1:], free_port())
    env = dict(os.environ)
    env.setdefault("HSA_ENABLE_IPC_MODE_LEGACY", "0")
    raise SystemExit(subprocess.call(cmd, env=env))


class Timer:
    """W warm-up steps, then EXACTLY K steps between barrier + synchronize, wall-clock MAX over ranks; that K-step region
    is repeated (warm-up only before the first) until min_seconds of device time is accumulated.  The kernels run on
    torch's current stream (ctx.set_stream) or, with `lanes`, in an unordered section that forks from and joins into it, so
    the torch.cuda.Event pair around the K launches is a hipEvent pair on the launch stream bracketing all of them:
    `dev_ms` = device time of each region.  The wall clock is read after the device has drained and BEFORE the trailing
    barrier (a RCCL barrier is a kernel launch plus a host sync that only N > 1 would pay); the MAX over ranks is taken from
    the per-rank times afterwards."""

    def __init__(self, K, Wm, use_dist, dev, min_seconds, max_repeats, ctx=None, lanes=0):
        self.K, self.Wm, self.use_dist, self.dev = K, Wm, use_dist, dev
        self.min_seconds, self.max_repeats = min_seconds, max_repeats
        self.ctx, self.lanes = ctx, lanes
        self.cuda = torch.device(dev).type == "cuda"      # (the gloo tests drive the same loop on the CPU: wall clock only)

    def _sync(self):
        if self.cuda:
            torch.cuda.synchronize()

    def region(self, fn, first, lanes):
        self._sync()
        if self.use_dist:
            dist.barrier()
        self._sync()
        e0 = e1 = None
        if self.cuda:
            e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
        t0 = time.perf_counter()
        if self.cuda:
            e0.record()
        if lanes:
            self.ctx.begin_unordered(lanes)
        for i in range(self.K):
            fn(first + i)
        if lanes:
            self.ctx.end_unordered()
        if self.cuda:
            e1.record()
        self._sync()
        wall = time.perf_counter() - t0
        if self.use_dist:
            dist.barrier()
        return wall, (e0.elapsed_time(e1) if self.cuda else 1e3 * wall)

    def run(self, fn, lanes=None):
        lanes = self.lanes if lanes is None else lanes
        # warm-up in the mode that is timed: the lane streams of an unordered section are created, and get their first launch, here
        # (the first launch of a large-LDS kernel on a fresh stream has been seen to take seconds, once)
        if lanes and self.Wm > 0:
            self.ctx.begin_unordered(lanes)
        for i in range(self.Wm):
            fn(i)
        if lanes and self.Wm > 0:
            self.ctx.end_unordered()
        walls, devs = [], []
        step = self.Wm
        while True:
            wall, dev_ms = self.region(fn, step, lanes)
            step += self.K
            walls.append(wall)
            devs.append(dev_ms)
            # every rank must take the same decision: rank 0 decides
            more = torch.tensor([1 if (sum(devs) * 1e-3 < self.min_seconds and len(devs) < self.max_repeats) else 0],
                                dtype=torch.int32, device=self.dev)
            if self.use_dist:
                dist.broadcast(more, src=0)
            if int(more.item()) == 0:
                break
        mine = float(np.median(walls))                           # this rank's median region
        t = torch.tensor(walls, dtype=torch.float64, device=self.dev)
        ranks = [mine]
        if self.use_dist:
            dist.all_reduce(t, op=dist.ReduceOp.MAX)          # per region: the slowest rank
            g = [torch.zeros(1, dtype=torch.float64, device=self.dev) for _ in range(dist.get_world_size())]
            dist.all_gather(g, torch.tensor([mine], dtype=torch.float64, device=self.dev))
            ranks = [float(x.item()) for x in g]
        walls = t.cpu().numpy()
        return {"wall_median": float(np.median(walls)), "wall_min": float(walls.min()), "wall_max": float(walls.max()),
                "dev_ms_median": float(np.median(devs)), "repeats": len(devs), "seconds": float(walls.sum()),
                "rank_wall_medians": ranks}


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


def run_workload(L, args, name, w, h, B, K, Wm, rank, world, local_rank, use_dist, dev, main, sha, pool=None, legs="full"):
    """one workload: resident synthetic stream, encode timed (plus decode / round trip for the main one), roofline blocks.
    legs: "full" = every leg; "encode" = the encode leg only (the placement-off comparison)."""
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
    per_frame = n3 * 4 * 2 + sum(psz)                      # input + decoded output + planes
    free, _total = torch.cuda.mem_get_info(dev)
    # main: configs[1]'s 500 frames (49.8 GB of input at 4K; larger frames: as many as the same bytes hold); others: >= 4 GB of
    # distinct input, >> the 256 MB MALL
    want_frames = min(500, max(B, int(50e9 // (n3 * 4)) // B * B)) if main else max(8 * B, int(4e9 // (n3 * 4)) // B * B)
    striped = False
    if pool is not None:
        from lumahdrv_amd.placement import CHUNK_BYTES, slots
        ypc, yslot = slots(CHUNK_BYTES, B * psz[0])
        uvpc, uvslot = slots(CHUNK_BYTES, B * psz[1] + (1 << 20) + B * psz[2])
        spc, sslot = slots(CHUNK_BYTES, B * n1 * 4)       # one colour plane of one batch per slot
        if B * n3 * 4 > CHUNK_BYTES or ypc < 1 or uvpc < 1:
            pool = None
    if pool is not None:
        # float frames: one chunk per batch of input; decoded output either packed (one chunk per batch) or, when the pool
        # kept chunks of three region groups for it, with the R, G and B planes of a batch in three different groups
        # (lumahip_decode_frames_device_planar); Y planes of `ypc` batches per chunk of one group; U and V planes of `uvpc`
        # batches per chunk of another (lumahdrv_amd/csrc/lumahip_pool.hip)
        striped = legs == "full" and args.decode_layout == "auto" and spc >= 1 and min(len(g) for g in pool.striped) >= 1
        nbatch = max(1, want_frames // B)
        while nbatch > 1 and (nbatch * (1 if striped else 2) > len(pool.float) or -(-nbatch // uvpc) > len(pool.uv)
                              or -(-nbatch // ypc) > len(pool.y) or (striped and -(-nbatch // spc) > min(len(g) for g in pool.striped))):
            nbatch -= 1
        src_c = pool.take_float(nbatch)                    # fastest first: the input gets the best chunks
        out_c = [] if striped else pool.take_float(nbatch)
        rgb_c = pool.take_striped(-(-nbatch // spc)) if striped else None
        uv_c = pool.take_uv(-(-nbatch // uvpc))
        y_c = pool.take_y(-(-nbatch // ypc))
        for c in uv_c + y_c:
            c.zero_()
        vo = (B * psz[1] + (1 << 20) - 1) // (1 << 20) * (1 << 20)

        def ptrs(b):
            u = uv_c[b // uvpc].data_ptr() + (b % uvpc) * uvslot
            if striped:
                o = [rgb_c[k][b // spc].data_ptr() + (b % spc) * sslot for k in range(3)]
            else:
                o = [out_c[b].data_ptr() + k * n1 * 4 for k in range(3)]
            return (src_c[b].data_ptr(), o, [y_c[b // ypc].data_ptr() + (b % ypc) * yslot, u, u + vo])
    else:
        nbatch = max(1, min(want_frames // B, int(free * 0.8 // per_frame) // B))
        src = torch.empty(nbatch * B * n3, dtype=torch.float32, device=dev)
        out = torch.empty(nbatch * B * n3 if legs == "full" else 4, dtype=torch.float32, device=dev)
        planes = [torch.zeros(nbatch * B * psz[p], dtype=torch.uint8, device=dev) for p in range(3)]

        def ptrs(b):
            o = out.data_ptr() + b * B * n3 * 4
            return (src.data_ptr() + b * B * n3 * 4, [o + k * n1 * 4 for k in range(3)],
                    [planes[p].data_ptr() + b * B * psz[p] for p in range(3)])
    out_fs = n1 if striped else n3                         # frame stride of the decoded output (floats)
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
         "workload": "%dx%d %s encode (%s, LUT quantize, 4:2:0 16-bit pack), %d frames/step, %d-frame resident stream per GPU"
                     % (w, h, desc, xf_desc, B, nfr),
         "frames_per_step": B, "width": w, "height": h, "preScaling": sc, "profile": profile,
         "distinct_input_GB_per_gpu": round(nfr * n3 * 4 / 1e9, 2)}
    if legs == "encode":
        ctx.close()
        if pool is not None:
            pool.give_back(src_c + out_c, y_c, uv_c, rgb_c)
        else:
            del src, out, planes
            torch.cuda.empty_cache()
        return r, cfg
    td = tm.run(dec)                                       # (the planes every batch holds are the encode leg's)
    rb_after_random = ctx.rb_table_info(sc) if cs == 2 else None
    r["decode_mpix_s"] = round(rate(td["wall_median"]), 1)
    r["decode_output_layout"] = ("R, G, B planes of a batch in three HBM region groups (lumahip_decode_frames_device_planar)"
                                 if striped else "packed LumaFrame layout")
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
    if main and striped:
        # The reference's decoder returns the PACKED LumaFrame (include/luma/luma_frame.h:84-87: channel c at buffer + c*h*w), not
        # the three-buffer layout the legs above write: the same decode launches into packed frames, a ring of PACKED_RING
        # batches (>> the 256 MB MALL), once in chunks of the pool (the fastest float chunks left) and once in a plain allocation.
        pk = {}
        # "pool_rotating": batch b's packed frames in the b-th chunk the pool hands out in its ROTATING mode (region groups 0, 1,
        # 2, 0, ...), so that the launches in flight on the two lanes write different groups (profiles/r03_layout_lab.txt: 0.75)
        # "frame_rotating": the FRAMES of a batch rotate over three chunks of three region groups (frame f in chunk f % 3; every
        # frame still a packed LumaFrame) and the launch interleaves its tiles over the frames, so ONE launch writes all three
        # groups (lumahip_decode_frames_device_rotating): what the ordered figure of the packed layout can be
        for how in ("pool_placed", "pool_rotating", "frame_rotating", "plain"):
            ring = rot = frot = None
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
            plain = torch.empty(PACKED_RING * B * n3, dtype=torch.float32, device=dev) if how == "plain" else None
            nring = len(ring) if ring is not None else PACKED_RING

            def dec_packed(i, ring=ring, plain=plain, nring=nring, frot=frot):
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
    if pool is not None:
        pool.give_back(src_c + out_c, y_c, uv_c, rgb_c)
    else:
        del src, out, planes
        torch.cuda.empty_cache()
    return r, cfg


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
           "config": {"workload": "%dx%d %s encode, ONE %d-frame stream sharded in contiguous blocks (%d frames on rank 0), "
                                  "%d frames/step" % (w, h, desc, F, len(shard_range(F, 0, world)), B),
                      "frames_per_step": B, "width": w, "height": h, "parallelism": "frame-sharded x%d" % world,
                      "world_size_reported_by": "torch.distributed (RCCL)" if use_dist else "single process"},
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
           "config": {"workload": "%dx%d %s encode, ONE %d-frame stream sharded in contiguous blocks (%d frames on shard 0), "
                                  "%d frames/step" % (w, h, desc, F, len(shards[0]), B),
                      "frames_per_step": B, "width": w, "height": h, "parallelism": "frame-sharded x%d" % ns,
                      "driver": "one process, lumahip_multi_* (C ABI): one context + stream per GPU, table broadcast with RCCL: %s"
                                % m.used_rccl()},
           "repeats": min(len(wl) for wl in seg_walls), "timed_seconds": round(sum(all_walls), 3), "segments": nseg,
           "ms_per_region_min_median_max": [round(1e3 * sum(min(wl) for wl in seg_walls), 3), round(1e3 * wall, 3),
                                            round(1e3 * sum(max(wl) for wl in seg_walls), 3)],
           "digests": {"gathered_in_stream_order": len(dig),
                       "stream_digest": "%016x" % (sum((i + 1) * v for i, v in enumerate(dig)) & 0xFFFFFFFFFFFFFFFF)},
           "placement": {"mode": "off (plain allocations)"}}
    m.close()
    return res


def cpu_baseline(args, cfg, w, h):
    """CPU reference on this host, bounded sample.  "reference": the real LumaQuantizer of the reference
    (oracle/_ref/libluma_ref.so, compiled unmodified in the build container and shipped prebuilt) under the harness's
    plane loop, 1 thread = the reference's behaviour; falls back to "port" (oracle/luma_oracle.c) when the prebuilt
    reference library is absent.  The all-core figure is always the port (row-sharded over pthreads)."""
    ptf, bits, cs, bitsC, maxLum, minLum, sc, profile = cfg
    try:
        from oracle import oracle_py as o
        nf = max(1, args.cpu_frames)
        cores = os.cpu_count() or 1
        orc = o.Oracle(ptf, bits, cs, bitsC, maxLum, minLum)
        kind, impl = "port", "oracle/luma_oracle.c (gcc -O2 -ffp-contract=off)"
        runner = lambda f: orc.encode(f, sc, profile, threads=1)  # noqa: E731
        if o.have_ref() and ptf in (o.PTF_PQ, o.PTF_LOG, o.PTF_LINEAR):
            try:
                ref = o.RefQuantizer(ptf, bits, cs, bitsC, maxLum, minLum)
                if hasattr(ref.L, "ref_encode_frame"):
                    kind, impl = "reference", ("reference LumaQuantizer (src/luma_quantizer.cpp, g++ -O2) under the "
                                               "plane loop of src/luma_encoder.cpp:260-317 restated in oracle/ref_harness.cpp")
                    runner = lambda f: ref.encode(f, sc, profile)  # noqa: E731
            except Exception:
                pass
        fr = [o.synth_frame(w, h, SEED, i) for i in range(nf)]
        t0 = time.perf_counter()
        for f in fr:
            runner(f)
        t1 = time.perf_counter() - t0
        fr = [o.synth_frame(w, h, SEED, i) for i in range(nf)]
        t0 = time.perf_counter()
        for f in fr:
            orc.encode(f, sc, profile, threads=cores)
        tn = time.perf_counter() - t0
        return {"value": round(nf * w * h / t1 / 1e6, 2), "unit": "Mpixels/s", "cores": 1, "kind": kind,
                "sample": "%d synthetic %dx%d frames, encode transform, %s, 1 thread = the reference's behaviour" % (nf, w, h, impl),
                "all_cores": {"value": round(nf * w * h / tn / 1e6, 2), "cores": cores, "kind": "port"}}
    except Exception as e:  # the baseline is reporting only; never fail the bench on it
        return {"value": None, "unit": "Mpixels/s", "cores": 0, "kind": "port", "sample": "failed: %r" % (e,)}


def geometry(w, h, profile=2):
    """(floats per frame, bytes of the Y / U / V planes of one frame) -- the plane layout of vpx_img_alloc(fmt(profile), w, h, 32)"""
    from lumahdrv_amd import capi
    _, hs, st, _ = capi.plane_geometry(w, h, profile)
    return 3 * w * h, [hs[p] * st[p] for p in range(3)]


def pool_request(w, h, B, decode_layout="auto", nbatches=None, with_output=True):
    """what the resident stream of `nbatches` batches (default: configs[1]'s 500 frames or 50 GB of input, whichever is smaller)
    asks of the chunk pool: chunks for float frames / Y planes / U+V planes / per region group for striped decode output.
    None when a batch does not fit a chunk (the 2 GiB chunks hold a 20-frame 4K batch; larger batches use plain allocations)."""
    from lumahdrv_amd.placement import CHUNK_BYTES, slots
    n3, psz = geometry(w, h)
    n1 = n3 // 3
    ypc, _ = slots(CHUNK_BYTES, B * psz[0])
    uvpc, _ = slots(CHUNK_BYTES, B * psz[1] + (1 << 20) + B * psz[2])
    spc, _ = slots(CHUNK_BYTES, B * n1 * 4)
    if B * n3 * 4 > CHUNK_BYTES or ypc < 1 or uvpc < 1:
        return None
    nb = nbatches if nbatches else min(500 // B, max(1, int(50e9 // (B * n3 * 4))))
    stripe = with_output and decode_layout == "auto" and spc >= 1
    return {"batches": nb, "n_float": nb * (2 if (with_output and not stripe) else 1) + (PACKED_RING if stripe else 0),   # + the packed-layout decode leg
            "n_y": -(-nb // ypc), "n_uv": -(-nb // uvpc), "n_striped": (-(-nb // spc) + PACKED_RING // 3 if stripe else 0),
            "chunk_bytes": CHUNK_BYTES, "striped_output": bool(stripe)}


def stream_shard_plan(F, rank, world, w, h, B, free, placement="auto", profile=2):
    """--stream-frames mode (BASELINE configs[4]): what `rank` holds of the ONE F-frame stream -- its block, the bytes resident in
    its HBM, whether they fit, and whether the shard is carved from the chunk pool (run_stream takes every decision from here)"""
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


def plan_only(args):
    """`bench.py --gpus N --plan-only`: the per-rank plan of the run the same command line would make, without touching a GPU"""
    if args.hbm_free_gb > 0:
        free = args.hbm_free_gb * 1e9
    elif torch.cuda.is_available():
        free = float(torch.cuda.mem_get_info(0)[0])
    else:
        free = 280e9
    w, h, B, N = args.width, args.height, args.frames_per_step, args.gpus
    ranks = []
    if args.stream_frames > 0:
        ranks = [stream_shard_plan(args.stream_frames, r, N, w, h, B, free, args.placement) for r in range(N)]
        mode = "ONE %d-frame stream block-sharded over %d rank(s) (strong scaling)" % (args.stream_frames, N)
    else:
        n3, psz = geometry(w, h)
        req = pool_request(w, h, B, args.decode_layout) if args.placement == "auto" else None
        frames = (req["batches"] * B) if req else min(500, max(B, int(50e9 // (n3 * 4)) // B * B))
        resident = frames * (2 * n3 * 4 + sum(psz)) + ((PACKED_RING * B * n3 * 4) if req and req["striped_output"] else 0)
        pool_bytes = (req["n_float"] + req["n_y"] + req["n_uv"] + 3 * req["n_striped"]) * req["chunk_bytes"] if req else 0
        for r in range(N):
            ranks.append({"rank": r, "first_frame": r * frames, "frames": frames, "steps_per_pass": frames // B,
                          "bytes_resident": resident, "pool_bytes": pool_bytes, "free_bytes": int(free),
                          "fits": max(resident, pool_bytes) <= free * 0.9, "pool": req,
                          "placement": "chunk pool" if req else "plain allocations"})
        mode = "every rank its own %d-frame stream (weak scaling)" % frames
    ok = all(r["fits"] or r.get("one_batch_fits") for r in ranks)      # (a stream shard that does not fit at once is encoded in blocks)
    print(json.dumps({"plan_only": True, "n_gpus": N, "mode": mode, "width": w, "height": h, "frames_per_step": B,
                      "collective": "one broadcast of the table (2^bits floats) + an 8-value parameter block from rank 0; none on the data path",
                      "fits": ok, "ranks": ranks}))
    return 0 if ok else 1


def make_pool(L, args, dev, local_rank, w, h, B, nbatches=None, with_output=True):
    """--placement auto: the chunk pool (C ABI lumahip_pool_*) the resident streams are carved from (None: plain allocations)"""
    if args.placement != "auto":
        return None
    try:
        from lumahdrv_amd.placement import HbmChunkPool
        req = pool_request(w, h, B, args.decode_layout, nbatches, with_output)
        if req is None:
            return None
        n_float, n_y, n_uv, n_striped = req["n_float"], req["n_y"], req["n_uv"], req["n_striped"]
        ctx = L.Context(local_rank)
        ctx.set_stream(torch.cuda.current_stream().cuda_stream)
        ctx.set_quantizer(L.PTF_PQ, 11, L.CS_LUV, 8, 1e4, 0.005, L.build_lut(L.PTF_PQ, 11, 1e4, 0.005))
        pool = HbmChunkPool(ctx, dev, n_float, n_y, n_uv, n_striped)
        ctx.close()
        if pool.float and pool.y and pool.uv:
            return pool
        pool.close()
        return None
    except Exception as e:      # placement is an optimisation, never a reason to fail the bench
        sys.stderr.write("bench.py: chunk pool unavailable (%r), plain allocations\n" % (e,))
        torch.cuda.empty_cache()
        return None


def facade_hostfed(w, h, runs=3):
    """LumaEncoder::encode(LumaFrame*) end to end on HOST frames (tools/facade_hostfed.cpp, the C++ facade): H2D + kernel +
    D2H per call -- the reference's drop-in call as its own callers make it.  PCIe-bound; reported beside, never as, `value`.
    Its own process, run BEFORE this process allocates device memory: for a few seconds after a process has handed tens of GB
    back to the driver (the chunk pool does, and so does the end of every leg), latency-bound work like one frame per call
    runs ~35 % slower (profiles/r03_settle.txt).  Median of `runs` short runs: the GPU boxes' hosts are shared."""
    exe = os.path.join(ROOT, "lumahdrv_amd", "bin", "facade_hostfed")
    try:
        got = []
        for _ in range(runs):
            out = subprocess.run([exe, str(w), str(h), "16"], capture_output=True, text=True, timeout=300)
            got.append(json.loads([l for l in out.stdout.splitlines() if l.startswith("{")][-1]))
        d = dict(got[0])
        for k in got[0]:
            if isinstance(got[0][k], float):
                d[k] = float(np.median([g[k] for g in got]))
        d["frames"] = got[0]["frames"]
        d["runs"] = runs
        d["pageable_runs"] = [g["LumaEncoder_encode_pageable_frame"] for g in got]
        d["what"] = ("host frames through the C++ facade, one frame per call, synchronous (H2D 12 B/px + fused kernel + D2H 3 B/px); "
                     "pageable = plain new float[] as the reference's LumaFrame, staged by the context's copy threads; median of "
                     "%d runs made before this process touched the GPU" % runs)
        return d
    except Exception as e:
        return {"error": repr(e)}


def main():
    args = parse()
    if args.plan_only:
        raise SystemExit(plan_only(args))
    multi_driver = args.stream_frames > 0 and args.driver == "multi"
    if args.gpus > 1 and "WORLD_SIZE" not in os.environ and not multi_driver:
        respawn(args)
    # The contract is ONE JSON line on stdout.  Libraries (RCCL with NCCL_DEBUG=VERSION, the ROCm runtime) print
    # banners to the C-level stdout, flushed at exit -- i.e. after anything Python prints.  So file descriptor 1 is
    # pointed at stderr for the whole run and the JSON line is written to the saved real stdout at the end.
    sys.stdout.flush()
    real_stdout = os.dup(1)
    os.dup2(2, 1)
    if multi_driver:
        # ONE process drives every GPU through the C ABI's many-GPU layer: no torch.distributed, no ranks
        if not torch.cuda.is_available():
            raise SystemExit("bench.py needs an MI355X (torch.cuda.is_available() is False); there is no CPU fallback")
        import lumahdrv_amd as L
        res = run_stream_multi(L, args)
        os.write(real_stdout, (json.dumps(res) + "\n").encode())
        return
    world = int(os.environ.get("WORLD_SIZE", "1"))
    rank = int(os.environ.get("RANK", "0"))
    local_rank = int(os.environ.get("LOCAL_RANK", "0"))
    if world != args.gpus:
        raise SystemExit("bench.py --gpus %d launched with WORLD_SIZE=%d: refusing to mislabel the result" % (args.gpus, world))
    if not torch.cuda.is_available():
        raise SystemExit("bench.py needs an MI355X (torch.cuda.is_available() is False); there is no CPU fallback")
    if local_rank >= torch.cuda.device_count():
        raise SystemExit("rank %d: LOCAL_RANK %d but only %d GPU(s) visible" % (rank, local_rank, torch.cuda.device_count()))
    torch.cuda.set_device(local_rank)
    dev = torch.device("cuda", local_rank)
    # LUMAHIP_BENCH_FORCE_DIST=1 exercises the RCCL code path (init, broadcast, barrier, all_reduce) with one rank too
    use_dist = world > 1 or os.environ.get("LUMAHIP_BENCH_FORCE_DIST") == "1"
    if use_dist:
        os.environ.setdefault("MASTER_ADDR", "127.0.0.1")
        os.environ.setdefault("MASTER_PORT", "29533")
        dist.init_process_group(backend="nccl", rank=rank, world_size=world, device_id=dev)   # "nccl" is RCCL on ROCm
        if dist.get_world_size() != args.gpus:
            raise SystemExit("RCCL reports world size %d, --gpus says %d" % (dist.get_world_size(), args.gpus))
    n_gpus = dist.get_world_size() if use_dist else 1

    import lumahdrv_amd as L   # after torch: one HIP runtime in the process
    from lumahdrv_amd import capi
    sha = capi.kernel_source_sha()

    if args.stream_frames > 0:
        res = run_stream(L, args, rank, n_gpus, local_rank, use_dist, dev)
    else:
        w, h, B, K, Wm = args.width, args.height, args.frames_per_step, args.steps, args.warmup
        # the host-fed leg first: its own process on a GPU this process has not allocated anything on yet (facade_hostfed says why)
        hostfed = facade_hostfed(w, h) if (rank == 0 and n_gpus == 1 and not args.no_facade_hostfed) else None
        pool = make_pool(L, args, dev, local_rank, w, h, B)
        r, cfg = run_workload(L, args, args.workload, w, h, B, K, Wm, rank, n_gpus, local_rank, use_dist, dev, True, sha, pool)
        res = {
            "metric": "Mpixels/s HDR quantize (4K PQ Lu'v' 11-bit)" if args.workload == "pq11_luv" and (w, h) == (W4K, H4K)
                      else "Mpixels/s HDR quantize (%s %dx%d)" % (args.workload, w, h),
            "value": r["value"], "unit": "Mpixels/s", "n_gpus": n_gpus, "steps": K, "warmup": Wm,
            "ms_per_step": r["ms_per_step"], "higher_is_better": True, "scaling": "weak",
            "vs_baseline": None, "dtype": "f32", "data": "synthetic",
            "config": {"workload": r["workload"],
                       "timed": "the quantize (encode) pass: K steps per region, region repeated until >= %.1f s of device time; "
                                "value / ms_per_step = the median region; decode and encode+decode round trip are timed "
                                "the same way and reported as decode_mpix_s / roundtrip_mpix_s" % args.min_seconds,
                       "frames_per_step": B, "width": w, "height": h, "preScaling": r["preScaling"], "profile": r["profile"],
                       "parallelism": "frame-sharded x%d" % n_gpus,
                       "world_size_reported_by": "torch.distributed (RCCL)" if use_dist else "single process",
                       "distinct_input_GB_per_gpu": r["distinct_input_GB_per_gpu"]},
            "repeats": r["repeats"], "timed_seconds": r["timed_seconds"],
            "ms_per_step_min": r["ms_per_step_min"], "ms_per_step_max": r["ms_per_step_max"],
            "ms_per_step_over_ranks": r["ms_per_step_over_ranks"], "lanes": r["lanes"],
            "value_ordered": r.get("value_ordered"), "decode_mpix_s_ordered": r.get("decode_mpix_s_ordered"),
            "decode_mpix_s": r["decode_mpix_s"], "decode_output_layout": r["decode_output_layout"],
            "roundtrip_mpix_s": r["roundtrip_mpix_s"],
            "decode_packed_layout": r.get("decode_packed_layout"),
            # the reference's own decode layout (LumaDecoder::decode() returns a packed LumaFrame) at its best placement: the frames of
            # a batch rotating over three chunks of three region groups (lumahip_decode_frames_device_rotating), two launches in flight
            "decode_packed_mpix_s": ((r.get("decode_packed_layout") or {}).get("frame_rotating") or {}).get("value"),
            "decode_packed_frac": ((r.get("decode_packed_layout") or {}).get("frame_rotating") or {}).get("frac_ordered"),
            "kernel_source_sha": sha,
            "placement": dict({"mode": args.placement}, **(pool.stats if pool is not None else {})),
        }
        if rank == 0:
            res["roofline"] = r["roofline"]
            res["decode_roofline"] = r["decode_roofline"]
            for k in ("float_inputs", "mixed_inputs_1e-3", "decode_coherent", "decode_random_rb_policy"):     # (--workload pq10_ycbcr: the legs other_workloads.pq10_ycbcr_4k carries by default)
                if k in r:
                    res[k] = r[k]
        if pool is not None and args.workload == "pq11_luv" and not args.no_placement_off:
            # the same kernels on plainly allocated buffers (a 200-frame resident stream, 20 GB >> the 256 MB MALL):
            # what a caller that does not place its buffers gets
            ro, _ = run_workload(L, args, args.workload, w, h, B, K, Wm, rank, n_gpus, local_rank, use_dist, dev, False, sha, None,
                                 legs="encode")
            res["value_placement_off"] = ro["value"]
        # ---- the other single-GPU configurations of BASELINE.json, same run, each with its own roofline (N = 1 only)
        if n_gpus == 1 and not args.no_other_workloads and args.workload == "pq11_luv" and (w, h) == (W4K, H4K):
            others = {}
            for key, (nm, ow, oh, ob) in {"pq10_ycbcr_4k": ("pq10_ycbcr", W4K, H4K, 20),
                                           "log12_luv_8k": ("log12_luv", W8K, H8K, 5)}.items():
                ro, _ = run_workload(L, args, nm, ow, oh, ob, K, Wm, rank, n_gpus, local_rank, use_dist, dev, False, sha, pool)
                others[key] = ro
            res["other_workloads"] = others
        if pool is not None:
            pool.close()
        if hostfed is not None:
            res["facade_hostfed"] = hostfed
        if rank == 0 and n_gpus == 1 and not args.no_cpu_baseline:
            res["cpu_baseline"] = cpu_baseline(args, cfg, w, h)

    if rank == 0:
        os.write(real_stdout, (json.dumps(res) + "\n").encode())
    if use_dist:
        dist.barrier()
        dist.destroy_process_group()


if __name__ == "__main__":
    main()
