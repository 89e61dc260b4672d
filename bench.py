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

from lumahdrv_amd.benchlib.legs import load_profile, run_workload  # noqa: E402,F401
from lumahdrv_amd.benchlib.plan import (H4K, OTHER_WORKLOADS, SEED, W4K, config_block, plan_only, pool_request,  # noqa: E402,F401
                                        stream_shard_plan)
from lumahdrv_amd.benchlib.resident import StreamDoesNotFit, make_pool, make_small_pool  # noqa: E402
from lumahdrv_amd.benchlib.stream import frame_digests, run_stream, run_stream_multi  # noqa: E402,F401
from lumahdrv_amd.benchlib.timing import Timer, ranks_seen  # noqa: E402,F401


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
    ap.add_argument("--allow-short-stream", action="store_true",
                    help="when the resident stream of the configuration does not fit the GPU's free HBM: run a shorter one and report "
                         "\"config_degraded\": true (default: fail with exit code 3 -- the workload is a function of the arguments only)")
    ap.add_argument("--plan-only", action="store_true",
                    help="print what every rank of `--gpus N` would hold (shard, resident bytes, pool chunks) as one JSON line and exit; "
                         "needs no GPU (then --hbm-free-gb says how much memory a rank has); exit code 1 when some rank does not fit")
    ap.add_argument("--hbm-free-gb", type=float, default=0.0,
                    help="--plan-only: free HBM per GPU in GB (default: what GPU 0 reports, or 280 of the MI355X's 288 GB without a GPU)")
    ap.add_argument("--profile-dir", default=os.path.join(ROOT, "profiles"),
                    help="where traffic_latest.json / valu_mix_latest.json (tools/summarize_profile.py) live")
    return ap.parse_args()



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



EXIT_DOES_NOT_FIT = 3        # the plan's resident stream does not fit this GPU's free HBM (and --allow-short-stream was not given)


def main():
    args = parse()
    if args.plan_only:
        raise SystemExit(plan_only(args))
    multi_driver = args.stream_frames > 0 and args.driver == "multi"
    if args.gpus > 1 and "WORLD_SIZE" not in os.environ and not multi_driver:
        respawn(args)
    # The workload -- the `config` block of the line -- is fixed here, from the arguments alone (benchlib/plan.py), before any
    # process-group, pool or free-memory work: the launcher form at N = 1 and the plain run are the same configuration by
    # construction, on any box.
    config = config_block(args, args.gpus)
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
    rccl_ranks_seen = None
    if use_dist:
        os.environ.setdefault("MASTER_ADDR", "127.0.0.1")
        os.environ.setdefault("MASTER_PORT", "29533")
        dist.init_process_group(backend="nccl", rank=rank, world_size=world, device_id=dev)   # "nccl" is RCCL on ROCm
        if dist.get_world_size() != args.gpus:
            raise SystemExit("RCCL reports world size %d, --gpus says %d" % (dist.get_world_size(), args.gpus))
        rccl_ranks_seen = ranks_seen(dev)
        if rccl_ranks_seen != args.gpus:
            raise SystemExit("an RCCL all_reduce of ones over the group gives %d, --gpus says %d" % (rccl_ranks_seen, args.gpus))
    n_gpus = dist.get_world_size() if use_dist else 1

    import lumahdrv_amd as L   # after torch: one HIP runtime in the process
    from lumahdrv_amd import capi
    sha = capi.kernel_source_sha()

    try:
        if args.stream_frames > 0:
            res = run_stream(L, args, rank, n_gpus, local_rank, use_dist, dev)
        else:
            res = run_default(L, args, config, rank, n_gpus, local_rank, use_dist, dev, sha)
    except StreamDoesNotFit as e:
        sys.stderr.write("bench.py: %s\n" % (e,))
        os._exit(EXIT_DOES_NOT_FIT)        # (no JSON line; os._exit: peers blocked in a collective must not keep this rank alive)
    res["rccl_ranks_seen"] = rccl_ranks_seen
    res["kernel_source_sha"] = sha

    if rank == 0:
        os.write(real_stdout, (json.dumps(res) + "\n").encode())
    if use_dist:
        dist.barrier()
        dist.destroy_process_group()


def run_default(L, args, config, rank, n_gpus, local_rank, use_dist, dev, sha):
    """the default line: the headline workload on every rank (weak scaling) + at N = 1 the other single-GPU configurations"""
    w, h, B, K, Wm = args.width, args.height, args.frames_per_step, args.steps, args.warmup
    # the host-fed leg first: its own process on a GPU this process has not allocated anything on yet (facade_hostfed says why)
    hostfed = facade_hostfed(w, h) if (rank == 0 and n_gpus == 1 and not args.no_facade_hostfed) else None
    pool = make_pool(L, args, dev, local_rank, w, h, B)
    r, cfg = run_workload(L, args, args.workload, w, h, B, K, Wm, rank, n_gpus, local_rank, use_dist, dev, True, sha, pool)
    config = dict(config, world_size_reported_by="torch.distributed (RCCL)" if use_dist else "single process")
    degraded = bool(r["config_degraded"])
    if degraded:     # --allow-short-stream and short memory: say what actually ran, and say that it is not the configuration asked for
        config.update(resident_frames=r["resident_frames"], stream_frames=r["resident_frames"] * n_gpus,
                      resident_frames_per_rank=[r["resident_frames"]] * n_gpus, workload=r["workload"],
                      distinct_input_GB_per_gpu=r["distinct_input_GB_per_gpu"])
    elif r["resident_frames"] != config["resident_frames"]:
        raise SystemExit("bench.py: internal error: %d frames resident, the plan says %d" % (r["resident_frames"], config["resident_frames"]))
    res = {
        "metric": "Mpixels/s HDR quantize (4K PQ Lu'v' 11-bit)" if args.workload == "pq11_luv" and (w, h) == (W4K, H4K)
                  else "Mpixels/s HDR quantize (%s %dx%d)" % (args.workload, w, h),
        "value": r["value"], "unit": "Mpixels/s", "n_gpus": n_gpus, "steps": K, "warmup": Wm,
        "ms_per_step": r["ms_per_step"], "higher_is_better": True, "scaling": config["scaling"],
        "vs_baseline": None, "dtype": "f32", "data": "synthetic",
        "config": config, "config_degraded": degraded,
        "repeats": r["repeats"], "timed_seconds": r["timed_seconds"],
        "ms_per_step_min": r["ms_per_step_min"], "ms_per_step_max": r["ms_per_step_max"],
        "ms_per_step_over_ranks": r["ms_per_step_over_ranks"], "lanes": r["lanes"],
        "value_ordered": r.get("value_ordered"), "decode_mpix_s_ordered": r.get("decode_mpix_s_ordered"),
        "decode_mpix_s": r["decode_mpix_s"], "decode_output_layout": r["decode_output_layout"],
        "roundtrip_mpix_s": r["roundtrip_mpix_s"],
        "decode_packed_layout": r.get("decode_packed_layout"),
        # the reference's own decode layout (LumaDecoder::decode() returns a packed LumaFrame) in buffers the LIBRARY allocates
        # (lumahip_decoded_ring_*: the frames of a batch rotating over three region groups), two launches in flight / ordered
        "decode_packed_mpix_s": ((r.get("decode_packed_layout") or {}).get("library_ring") or {}).get("value"),
        "decode_packed_frac": ((r.get("decode_packed_layout") or {}).get("library_ring") or {}).get("frac_ordered"),
        "placement": dict({"mode": args.placement, "resident_stream": r["resident_stream"]}, **(pool.stats if pool is not None else {})),
    }
    if rank == 0:
        res["roofline"] = r["roofline"]
        res["decode_roofline"] = r["decode_roofline"]
        for k in ("float_inputs", "mixed_inputs_1e-3", "decode_coherent", "decode_random_rb_policy"):     # (--workload pq10_ycbcr: the legs other_workloads.pq10_ycbcr_4k carries by default)
            if k in r:
                res[k] = r[k]
    if pool is not None and args.workload == "pq11_luv" and not args.no_placement_off:
        # the same kernels on plainly allocated buffers (a 160-frame resident stream, 16 GB >> the 256 MB MALL):
        # what a caller that does not place its buffers gets
        ro, _ = run_workload(L, args, args.workload, w, h, B, K, Wm, rank, n_gpus, local_rank, use_dist, dev, False, sha, None,
                             legs="encode")
        res["value_placement_off"] = ro["value"]
        res["placement_off"] = {"value": ro["value"], "value_ordered": ro.get("value_ordered"), "frac_ordered": ro.get("frac_ordered"),
                                "resident_frames": ro["resident_frames"]}
    # ---- the other single-GPU configurations of BASELINE.json, same run, each with its own roofline (N = 1 only)
    if n_gpus == 1 and not args.no_other_workloads and args.workload == "pq11_luv" and (w, h) == (W4K, H4K):
        others = {}
        for key, (nm, ow, oh, ob) in OTHER_WORKLOADS.items():
            ro, _ = run_workload(L, args, nm, ow, oh, ob, K, Wm, rank, n_gpus, local_rank, use_dist, dev, False, sha, pool)
            others[key] = ro
        res["other_workloads"] = others
    if pool is not None:
        pool.close()
    if args.placement == "auto" and args.workload == "pq11_luv" and not args.no_placement_off and (w, h) == (W4K, H4K):
        # ... and what a caller gets that keeps a FEW GB resident and shares the GPU: lumahip_pool_create_small (a dozen chunks probed
        # for milliseconds, 2 float + 1 Y + 1 U/V chunks = 8 GB kept), a 2-batch resident stream (4 GB of input >> the 256 MB MALL)
        sp = make_small_pool(L, dev, local_rank)
        if sp is not None:
            rsm, _ = run_workload(L, args, args.workload, w, h, B, K, Wm, rank, n_gpus, local_rank, use_dist, dev, False, sha, sp,
                                  legs="encode", nbatch=2)
            res["value_small_pool"] = rsm["value"]
            res["small_pool"] = dict({"value": rsm["value"], "value_ordered": rsm.get("value_ordered"), "frac_ordered": rsm.get("frac_ordered"),
                                      "resident_frames": rsm["resident_frames"], "resident_stream": rsm["resident_stream"],
                                      "pool_GB_kept": round(4 * sp.pool.chunk_bytes / 1e9, 1)}, **sp.stats)
            sp.close()
    if hostfed is not None:
        res["facade_hostfed"] = hostfed
    if rank == 0 and n_gpus == 1 and not args.no_cpu_baseline:
        res["cpu_baseline"] = cpu_baseline(args, cfg, w, h)
    return res


if __name__ == "__main__":
    main()
