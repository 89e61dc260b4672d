#!/usr/bin/env python3
"""bench.py -- Mpixels/s of the HDR quantize hot path (4K PQ 11-bit Lu'v', VP9 profile 2) on MI355X.

    python bench.py --gpus N --steps K --warmup W
    python -m torch.distributed.run --nnodes=1 --nproc-per-node N --master-addr 127.0.0.1 --master-port P \
        bench.py --gpus N --steps K --warmup W

A *step* is one pass of the fused encode kernel (RGB -> XYZ -> Lu'v' -> PQ-LUT quantize -> 4:2:0 16-bit
planes, = LumaEncoder::encode minus the VP9 stage) over one batch of `--frames-per-step` synthetic
3840x2160 frames that are already resident in HBM.  The default K=25 steps x 20 frames is BASELINE.json
configs[1]'s 500-frame stream; every step reads frames no earlier step touched (a ring of distinct
batches when memory is short), so nothing is served from L2 / Infinity Cache.  W warm-up steps, then exactly K
steps between barrier + synchronize, MAX over ranks; rank 0 prints ONE JSON line.

Multi-GPU: frames are independent, so rank r owns frame indices [r*K*B, (r+1)*K*B) (weak scaling, no
data-path collective); the only communication is one RCCL broadcast of the transfer-function table and the
quantizer parameters from rank 0 before the timed region.

Extra legs reported in the same line (not part of `value`): decode and encode+decode round trip, the
roofline of the encode kernel (HIP events on the launch stream, live), and the CPU oracle timed on this
host (`cpu_baseline`).
"""
import argparse
import json
import os
import sys
import time

ROOT = os.path.dirname(os.path.abspath(__file__))
if ROOT not in sys.path:
    sys.path.insert(0, ROOT)

import numpy as np  # noqa: E402
import torch  # noqa: E402
import torch.distributed as dist  # noqa: E402

W4K, H4K = 3840, 2160
SEED = 20250929
BYTES_PER_PIXEL = 15.0      # 12 B read (3 x fp32) + 3 B written (Y 2 B + U 0.5 B + V 0.5 B), SURVEY.md 8(d)
HBM_PEAK_GBS = 8000.0       # MI355X_MICROARCH.md: 8.0 TB/s spec


def parse():
    ap = argparse.ArgumentParser()
    ap.add_argument("--gpus", type=int, default=1)
    ap.add_argument("--steps", type=int, default=25)
    ap.add_argument("--warmup", type=int, default=3)
    ap.add_argument("--frames-per-step", type=int, default=20)
    ap.add_argument("--width", type=int, default=W4K)
    ap.add_argument("--height", type=int, default=H4K)
    ap.add_argument("--workload", default="pq11_luv", choices=["pq11_luv", "pq10_ycbcr", "log12_luv"])
    ap.add_argument("--no-cpu-baseline", action="store_true")
    ap.add_argument("--cpu-frames", type=int, default=16, help="frames the CPU baseline encodes (bounded sample, ~10 s on 1 thread)")
    ap.add_argument("--traffic-json", default=os.path.join(ROOT, "profiles", "traffic_latest.json"),
                    help="per-launch HBM bytes from the rocprofv3 PMC passes (tools/profile_round.sh + tools/summarize_profile.py); optional")
    return ap.parse_args()


WORKLOADS = {
    # name: (ptf, bits, cs, bitsC, maxLum, minLum, preScaling, profile, description)
    "pq11_luv": (1, 11, 0, 8, 1e4, 0.005, 1.0, 2, "PQ 11-bit Lu'v' 8-bit chroma, profile 2 (4:2:0 16-bit)",
                 "RGB->XYZ->Lu'v'", "lh::k_encode<CS_LUV,4:2:0,VW=4,LDS threshold records>"),
    "pq10_ycbcr": (1, 10, 2, 10, 1000.0, 0.01, 20.0, 2, "HDR10 recipe: PQ 10-bit YCbCr BT.2020 10-bit chroma, max/min 1000/0.01, preScaling 20",
                   "RGB->PQ->Y'CbCr (8 glibc-exact powf per pixel: fp64-VALU-bound, not HBM-bound)",
                   "lh::k_encode<CS_YCBCR,4:2:0,VW=4,LDS threshold records>"),
    "log12_luv": (2, 12, 0, 8, 1e4, 0.005, 1.0, 2, "LOG 12-bit Lu'v' 8-bit chroma, profile 2",
                  "RGB->XYZ->Lu'v'", "lh::k_encode<CS_LUV,4:2:0,VW=4,LDS threshold records>"),
}


def main():
    args = parse()
    # The contract is ONE JSON line on stdout.  Libraries (RCCL with NCCL_DEBUG=VERSION, the ROCm runtime) print
    # banners to the C-level stdout, flushed at exit -- i.e. after anything Python prints.  So file descriptor 1 is
    # pointed at stderr for the whole run and the JSON line is written to the saved real stdout at the end.
    sys.stdout.flush()
    real_stdout = os.dup(1)
    os.dup2(2, 1)
    world = int(os.environ.get("WORLD_SIZE", "1"))
    rank = int(os.environ.get("RANK", "0"))
    local_rank = int(os.environ.get("LOCAL_RANK", "0"))
    if not torch.cuda.is_available():
        raise SystemExit("bench.py needs an MI355X (torch.cuda.is_available() is False); there is no CPU fallback")
    torch.cuda.set_device(local_rank)
    dev = torch.device("cuda", local_rank)
    # LUMAHIP_BENCH_FORCE_DIST=1 exercises the RCCL code path (init, broadcast, barrier, all_reduce) with one rank too
    use_dist = world > 1 or os.environ.get("LUMAHIP_BENCH_FORCE_DIST") == "1"
    if use_dist:
        os.environ.setdefault("MASTER_ADDR", "127.0.0.1")
        os.environ.setdefault("MASTER_PORT", "29533")
        dist.init_process_group(backend="nccl", rank=rank, world_size=world, device_id=dev)   # "nccl" is RCCL on ROCm
    n_gpus = world

    import lumahdrv_amd as L   # after torch: one HIP runtime in the process

    ptf, bits, cs, bitsC, maxLum, minLum, sc, profile, desc, xf_desc, kname = WORKLOADS[args.workload]
    w, h, B, K, Wm = args.width, args.height, args.frames_per_step, args.steps, args.warmup

    # ---- quantizer: rank 0 builds the table on its host, RCCL-broadcasts it and the parameters over xGMI
    from lumahdrv_amd.sharding import broadcast_quantizer
    cfg0 = lut0 = None
    if rank == 0:
        cfg0 = (ptf, bits, cs, bitsC, maxLum, minLum, sc, profile)
        lut0 = L.build_lut(ptf, bits, maxLum, minLum)
    cfg, lut = broadcast_quantizer(cfg0, lut0, dev)
    ptf, bits, cs, bitsC, maxLum, minLum, sc, profile = cfg
    ctx = L.Context(local_rank)
    ctx.set_stream(torch.cuda.current_stream().cuda_stream)
    ctx.set_quantizer(ptf, bits, cs, bitsC, maxLum, minLum, lut)

    # ---- resident synthetic stream: as many distinct batches as the step count needs (or memory allows)
    n3 = 3 * w * h
    _, hs, st, _ = L.plane_geometry(w, h, profile)
    psz = [hs[p] * st[p] for p in range(3)]
    per_frame = n3 * 4 * 2 + sum(psz)                      # input + decoded output + planes
    free, _total = torch.cuda.mem_get_info(dev)
    max_frames = max(B, int(free * 0.8 // per_frame))
    nbatch = max(1, min(K + Wm, max_frames // B))
    ring_bytes = nbatch * B * n3 * 4
    nfr = nbatch * B
    src = torch.empty(nfr * n3, dtype=torch.float32, device=dev)
    out = torch.empty(nfr * n3, dtype=torch.float32, device=dev)
    planes = [torch.zeros(nfr * psz[p], dtype=torch.uint8, device=dev) for p in range(3)]
    first = rank * (K + Wm) * B
    for b in range(nbatch):
        ctx.synth_frames_device(src.data_ptr() + b * B * n3 * 4, n3, B, w, h, SEED, first + b * B)
    torch.cuda.synchronize()

    def ptrs(b):
        return (src.data_ptr() + b * B * n3 * 4, out.data_ptr() + b * B * n3 * 4,
                [planes[p].data_ptr() + b * B * psz[p] for p in range(3)])

    def enc(b):
        s, _, pl = ptrs(b % nbatch)
        ctx.encode_frames_device(s, n3, B, w, h, sc, profile, pl, st, psz)

    def dec(b):
        _, o, pl = ptrs(b % nbatch)
        ctx.decode_frames_device(pl, st, psz, B, w, h, profile, sc, o, n3)

    dev_ms = {}

    def timed(fn, tag=None):
        """W warm-up steps, then exactly K steps between barrier + synchronize; MAX over ranks (seconds).
        The kernels run on torch's current stream (ctx.set_stream above), so a pair of torch.cuda.Events around the
        K launches is a pair of hipEvents on the launch stream: dev_ms[tag] = device time of the timed region."""
        for i in range(Wm):
            fn(i)
        torch.cuda.synchronize()
        if use_dist:
            dist.barrier()
        torch.cuda.synchronize()
        e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
        t0 = time.perf_counter()
        e0.record()
        for i in range(K):
            fn(Wm + i)
        e1.record()
        torch.cuda.synchronize()
        if tag:
            dev_ms[tag] = e0.elapsed_time(e1)
        if use_dist:
            dist.barrier()
        torch.cuda.synchronize()
        dt = time.perf_counter() - t0
        if use_dist:
            t = torch.tensor([dt], dtype=torch.float64, device=dev)
            dist.all_reduce(t, op=dist.ReduceOp.MAX)
            dt = float(t.item())
        return dt

    px_step = float(B) * w * h
    t_enc = timed(enc, "enc")                                # the metric: quantize
    t_dec = timed(dec, "dec")
    t_rt = timed(lambda i: (enc(i), dec(i)))

    value = n_gpus * K * px_step / t_enc / 1e6
    res = {
        "metric": "Mpixels/s HDR quantize (4K PQ Lu'v' 11-bit)" if args.workload == "pq11_luv" and (w, h) == (W4K, H4K)
                  else "Mpixels/s HDR quantize (%s %dx%d)" % (args.workload, w, h),
        "value": round(value, 1), "unit": "Mpixels/s", "n_gpus": n_gpus, "steps": K, "warmup": Wm,
        "ms_per_step": round(1e3 * t_enc / K, 4), "higher_is_better": True, "scaling": "weak",
        "vs_baseline": None, "dtype": "f32", "data": "synthetic",
        "config": {"workload": "%dx%d %s encode (%s, LUT quantize, 4:2:0 16-bit pack), %d frames/step, "
                               "%d-frame resident stream per GPU" % (w, h, desc, xf_desc, B, nfr),
                   "timed": "the quantize (encode) pass; decode and encode+decode round trip are timed separately "
                            "and reported as decode_mpix_s / roundtrip_mpix_s",
                   "frames_per_step": B, "width": w, "height": h, "preScaling": sc, "profile": profile,
                   "parallelism": "frame-sharded x%d" % n_gpus, "distinct_input_GB_per_gpu": round(ring_bytes / 1e9, 2)},
        "decode_mpix_s": round(n_gpus * K * px_step / t_dec / 1e6, 1),
        "roundtrip_mpix_s": round(n_gpus * K * px_step / t_rt / 1e6, 1),
    }

    # ---- roofline of the dominant kernel (encode): HIP events on the launch stream, live, rank 0
    if rank == 0:
        s, o, pl = ptrs(0)
        iters = max(5, min(K, nbatch))
        # cycle over distinct batches so the working set never fits the caches
        ms = []
        for i in range(iters):
            s_i, _, pl_i = ptrs(i % nbatch)
            ms.append(ctx.time_launches(0, 1, s_i, n3, B, w, h, sc, profile, pl_i, st, psz))
        iso_ms = float(np.mean(ms))                  # isolated launches, one hipEvent pair each
        avg_ms = dev_ms["enc"] / K                   # hipEvents over the timed region: K back-to-back launches
        achieved = BYTES_PER_PIXEL * px_step / (avg_ms * 1e-3) / 1e9
        probe_ms = None
        if profile == 2 and w % 4 == 0:
            # the same loads and stores with no arithmetic: what the memory system gives this traffic mix here.
            # (overwrites the planes of these batches; they are re-encoded by nothing afterwards)
            pm = []
            for i in range(iters):
                s_i, _, pl_i = ptrs(i % nbatch)
                pm.append(ctx.probe_encode_traffic(s_i, n3, B, w, h, pl_i, st, psz))
            probe_ms = float(np.mean(pm))
        traffic = None
        try:
            with open(args.traffic_json) as f:
                tj = json.load(f)
            if tj.get("pixels_per_launch") == px_step and tj.get("workload") == args.workload and (w, h) == (W4K, H4K):
                traffic = tj.get("hbm_bytes_per_launch")
        except Exception:
            pass
        res["roofline"] = {"bound": "hbm", "achieved": round(achieved, 1), "peak": HBM_PEAK_GBS, "unit": "GB/s",
                           "frac": round(achieved / HBM_PEAK_GBS, 4), "traffic": traffic,
                           "kernel": kname, "kernel_ms": round(avg_ms, 4), "kernel_ms_isolated_launch": round(iso_ms, 4),
                           "algorithmic_bytes_per_launch": BYTES_PER_PIXEL * px_step,
                           "traffic_only_ms": None if probe_ms is None else round(probe_ms, 4),
                           "frac_of_traffic_only_rate": None if probe_ms is None else round(probe_ms / iso_ms, 3),
                           "decode_achieved_GBs": round(BYTES_PER_PIXEL * px_step / (dev_ms["dec"] / K * 1e-3) / 1e9, 1)}

    # ---- CPU baseline on this host, bounded sample.  "reference": the real LumaQuantizer of the reference
    # (oracle/_ref/libluma_ref.so, compiled unmodified in the build container and shipped prebuilt) under the harness's
    # plane loop, 1 thread = the reference's behaviour; falls back to "port" (oracle/luma_oracle.c) when the prebuilt
    # reference library is absent.  The all-core figure is always the port (row-sharded over pthreads).
    if rank == 0 and n_gpus == 1 and not args.no_cpu_baseline:
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
            res["cpu_baseline"] = {"value": round(nf * w * h / t1 / 1e6, 2), "unit": "Mpixels/s", "cores": 1, "kind": kind,
                                   "sample": "%d synthetic %dx%d frames, encode transform, %s, 1 thread = the reference's "
                                             "behaviour" % (nf, w, h, impl),
                                   "all_cores": {"value": round(nf * w * h / tn / 1e6, 2), "cores": cores, "kind": "port"}}
        except Exception as e:  # the baseline is reporting only; never fail the bench on it
            res["cpu_baseline"] = {"value": None, "unit": "Mpixels/s", "cores": 0, "kind": "port", "sample": "failed: %r" % (e,)}

    if rank == 0:
        os.write(real_stdout, (json.dumps(res) + "\n").encode())
    if use_dist:
        dist.barrier()
        dist.destroy_process_group()


if __name__ == "__main__":
    main()
